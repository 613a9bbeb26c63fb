// ops_small_forward2.hip — two-wave forward-pass variants (k_forward2) of the small models.
#include "ops.h"

namespace to {
void fill_ops_small_forward2(ModelOps* t) {
  fill_forward2<DoubleIntegratorModel<1>, 0, 16>(t[0]);
  fill_forward2<DoubleIntegratorModel<2>, 0, 16>(t[1]);
  fill_forward2<DoubleIntegratorModel<3>, 0, 16>(t[2]);
  fill_forward2<CartpoleModel, 0, 16>(t[3]);
  fill_forward2<HybridDoubleIntegratorModel, 0, 16>(t[7]);
}
}  // namespace to
