// ops_quad_forward2_a.hip — Quadrotor: two-wave forward-pass variants without constraints.
#include "ops.h"

namespace to {
void fill_ops_quad_forward2_a(ModelOps* t) {
  fill_forward2<QuadrotorModel, 0, 2>(t[4]);
  fill_forward2<QuadrotorModel, 8, 10>(t[4]);
}
}  // namespace to
