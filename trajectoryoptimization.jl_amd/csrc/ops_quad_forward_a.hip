// ops_quad_forward_a.hip — Quadrotor: forward-pass variants without constraints.
#include "ops.h"

namespace to {
void fill_ops_quad_forward_a(ModelOps* t) {
  t[4].accept_roll = op_accept_roll<QuadrotorModel>;
  fill_forward<QuadrotorModel, 0, 2>(t[4]);
  fill_forward<QuadrotorModel, 8, 10>(t[4]);
}
}  // namespace to
