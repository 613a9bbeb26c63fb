// ops_quad_backward.hip — Quadrotor: backward Riccati pass (MFMA and cooperative variants).
#include "ops.h"

namespace to {
void fill_ops_quad_backward(ModelOps* t) { t[4].backward = op_backward<QuadrotorModel>; }
}  // namespace to
