// ops_hybrid.hip — the hybrid double integrator (model vector with a dimension change, test/hybrid_dynamics_model.jl): every kernel of
// the small-model paths, taking its time step through model_step.
#include "ops_lane.h"

namespace to {
void fill_ops_hybrid(ModelOps* t) {
  using M = HybridDoubleIntegratorModel;
  fill_misc<M>(t[7]);
  t[7].expand = op_expand<M>;
  t[7].backward = op_backward<M>;
  t[7].expand_lane_k = op_expand_lane<M>;
  t[7].expand_backward = op_expand_backward<M>;
  t[7].expand_backward_coop = op_expand_backward_coop<M>;
  fill_forward<M, 0, 16>(t[7]);
  t[7].accept_roll = op_accept_roll<M>;
}
}  // namespace to
