// ops_quadrp_expand.hip — Quadrotor{RodriguesParam}: expansion variants.
#include "ops.h"

namespace to {
void fill_ops_quadrp_expand(ModelOps* t) { t[6].expand = op_expand<QuadrotorAttModel<ATT_RP>>; t[6].expand_const = op_expand_const<QuadrotorAttModel<ATT_RP>>; }
}  // namespace to
