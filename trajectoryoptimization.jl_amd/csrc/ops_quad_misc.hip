// ops_quad_misc.hip — Quadrotor: rollout, cost, AL outer update, per-knot API kernels.
#include "ops.h"

namespace to {
void fill_ops_quad_misc(ModelOps* t) { fill_misc<QuadrotorModel>(t[4]); }
}  // namespace to
