// ops_quad_forward2_b.hip — Quadrotor: two-wave forward-pass variants with constraints (AL terms).
#include "ops.h"

namespace to {
void fill_ops_quad_forward2_b(ModelOps* t) {
  fill_forward2<QuadrotorModel, 2, 4>(t[4]);
  fill_forward2<QuadrotorModel, 10, 12>(t[4]);
}
}  // namespace to
