// ops_quad_forward_b.hip — Quadrotor: forward-pass variants with constraints (AL terms).
#include "ops.h"

namespace to {
void fill_ops_quad_forward_b(ModelOps* t) {
  fill_forward<QuadrotorModel, 2, 4>(t[4]);
  fill_forward<QuadrotorModel, 10, 12>(t[4]);
}
}  // namespace to
