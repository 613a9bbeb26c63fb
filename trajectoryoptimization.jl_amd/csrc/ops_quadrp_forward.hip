// ops_quadrp_forward.hip — Quadrotor{RodriguesParam}: the two general forward-pass variants.
#include "ops.h"

namespace to {
void fill_ops_quadrp_forward(ModelOps* t) {
  t[6].accept_roll = op_accept_roll<QuadrotorAttModel<ATT_RP>>;
  fill_forward<QuadrotorAttModel<ATT_RP>, 8, 9>(t[6]);
  fill_forward<QuadrotorAttModel<ATT_RP>, 10, 11>(t[6]);
  fill_forward2<QuadrotorAttModel<ATT_RP>, 8, 9>(t[6]);
  fill_forward2<QuadrotorAttModel<ATT_RP>, 10, 11>(t[6]);
}
}  // namespace to
