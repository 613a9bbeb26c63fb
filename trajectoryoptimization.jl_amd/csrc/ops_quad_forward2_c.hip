// ops_quad_forward2_c.hip — Quadrotor: two-wave forward-pass variants constrained, cached control constraints are unit SOCs (MODE bit4).
#include "ops.h"

namespace to {
void fill_ops_quad_forward2_c(ModelOps* t) {
  fill_forward2<QuadrotorModel, 18, 20>(t[4]);
  fill_forward2<QuadrotorModel, 26, 28>(t[4]);
}
}  // namespace to
