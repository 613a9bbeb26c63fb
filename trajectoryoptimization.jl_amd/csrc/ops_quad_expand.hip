// ops_quad_expand.hip — Quadrotor: expansion variants.
#include "ops.h"

namespace to {
void fill_ops_quad_expand(ModelOps* t) { t[4].expand = op_expand<QuadrotorModel>; t[4].expand_const = op_expand_const<QuadrotorModel>; }
}  // namespace to
