// ops_small_forward.hip — forward-pass variants of the small models.
#include "ops.h"

namespace to {
void fill_ops_small_forward(ModelOps* t) {
  fill_forward<DoubleIntegratorModel<1>, 0, 16>(t[0]);
  fill_forward<DoubleIntegratorModel<2>, 0, 16>(t[1]);
  fill_forward<DoubleIntegratorModel<3>, 0, 16>(t[2]);
  fill_forward<CartpoleModel, 0, 16>(t[3]);
  // full-chip batch steps store the candidates' controls only and re-roll the accepted ones (k_accept_roll: same translation unit,
  // same -ffp-contract=on, bit-identical states)
  t[0].accept_roll = op_accept_roll<DoubleIntegratorModel<1>>;
  t[1].accept_roll = op_accept_roll<DoubleIntegratorModel<2>>;
  t[2].accept_roll = op_accept_roll<DoubleIntegratorModel<3>>;
  t[3].accept_roll = op_accept_roll<CartpoleModel>;
}
}  // namespace to
