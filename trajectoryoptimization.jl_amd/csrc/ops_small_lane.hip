// ops_small_lane.hip — the one-lane-per-trajectory expansion kernels of the small models (k_expand_lane and the fused
// k_expand_backward_lane: the large-batch throughput path), compiled with -fno-honor-nans -fno-honor-infinities -fno-signed-zeros so
// that the structural zeros of the chunk-mode dual numbers fold away (ops_lane.h, build.py).
#include "ops_lane.h"

namespace to {
template <class M>
static void fill_one(ModelOps& o) {
  if constexpr (M::lane_backward && !M::lie) {
    o.expand_lane_k = op_expand_lane<M>;
    o.expand_backward = op_expand_backward<M>;
  }
}
void fill_ops_small_lane(ModelOps* t) {
  fill_one<DoubleIntegratorModel<1>>(t[0]);
  fill_one<DoubleIntegratorModel<2>>(t[1]);
  fill_one<DoubleIntegratorModel<3>>(t[2]);
  fill_one<CartpoleModel>(t[3]);
}
}  // namespace to
