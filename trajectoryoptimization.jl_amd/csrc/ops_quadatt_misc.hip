// ops_quadatt_misc.hip — Quadrotor with a three-parameter attitude (MRP key 5, RodriguesParam key 6): rollout, cost, AL outer
// update, per-knot API kernels, backward Riccati pass.
#include "ops.h"

namespace to {
void fill_ops_quadatt_misc(ModelOps* t) {
  fill_misc<QuadrotorAttModel<ATT_MRP>>(t[5]);
  fill_misc<QuadrotorAttModel<ATT_RP>>(t[6]);
  t[5].backward = op_backward<QuadrotorAttModel<ATT_MRP>>;
  t[6].backward = op_backward<QuadrotorAttModel<ATT_RP>>;
}
}  // namespace to
