// ops_quad_forward_c.hip — Quadrotor: constrained forward-pass variants whose cached control constraints are unit SOCs (MODE bit4).
#include "ops.h"

namespace to {
void fill_ops_quad_forward_c(ModelOps* t) {
  fill_forward<QuadrotorModel, 18, 20>(t[4]);
  fill_forward<QuadrotorModel, 26, 28>(t[4]);
}
}  // namespace to
