// ops_quadmrp_expand.hip — Quadrotor{MRP}: expansion variants.
#include "ops.h"

namespace to {
void fill_ops_quadmrp_expand(ModelOps* t) { t[5].expand = op_expand<QuadrotorAttModel<ATT_MRP>>; t[5].expand_const = op_expand_const<QuadrotorAttModel<ATT_MRP>>; }
}  // namespace to
