// ops_small_forward.hip — forward-pass variants of the small models.
#include "ops.h"

namespace to {
void fill_ops_small_forward(ModelOps* t) {
  fill_forward<DoubleIntegratorModel<1>, 0, 16>(t[0]);
  fill_forward<DoubleIntegratorModel<2>, 0, 16>(t[1]);
  fill_forward<DoubleIntegratorModel<3>, 0, 16>(t[2]);
  fill_forward<CartpoleModel, 0, 16>(t[3]);
}
}  // namespace to
