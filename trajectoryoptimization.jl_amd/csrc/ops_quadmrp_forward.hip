// ops_quadmrp_forward.hip — Quadrotor{MRP}: the two general forward-pass variants (any cost kind; without / with constraints).
// launch_forward falls back to them from the specialised variants the quaternion model has.
#include "ops.h"

namespace to {
void fill_ops_quadmrp_forward(ModelOps* t) {
  t[5].accept_roll = op_accept_roll<QuadrotorAttModel<ATT_MRP>>;
  fill_forward<QuadrotorAttModel<ATT_MRP>, 8, 9>(t[5]);
  fill_forward<QuadrotorAttModel<ATT_MRP>, 10, 11>(t[5]);
  fill_forward2<QuadrotorAttModel<ATT_MRP>, 8, 9>(t[5]);
  fill_forward2<QuadrotorAttModel<ATT_MRP>, 10, 11>(t[5]);
}
}  // namespace to
