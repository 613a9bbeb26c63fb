// ops_small_scan.hip — fused expansion + scan (parallel-in-time) backward pass of the small models (k_scan.h).
#include "ops.h"

namespace to {
void fill_ops_small_scan(ModelOps* t) {
  t[0].expand_backward_scan = op_expand_backward_scan<DoubleIntegratorModel<1>>;
  t[1].expand_backward_scan = op_expand_backward_scan<DoubleIntegratorModel<2>>;
  t[3].expand_backward_scan = op_expand_backward_scan<CartpoleModel>;
  t[7].expand_backward_scan = op_expand_backward_scan<HybridDoubleIntegratorModel>;
}
}  // namespace to
