// ops_vector.hip — the general model vector (TO_MODEL_VECTOR; Problem(models::Vector, ...), src/problem.jl:36-73, src/dynamics.jl:15-31):
// one model per time step out of a table in device memory (models.h ModelVectorModel), stored at (6, 3).  Split expansion
// (one lane per column) + cooperative backward pass + both forward kernels, every time step through model_step.
#include "ops.h"

namespace to {
void fill_ops_vector(ModelOps* t) {
  using M = ModelVectorModel;
  fill_misc<M>(t[8]);
  t[8].expand = op_expand<M>;
  t[8].backward = op_backward<M>;
  fill_forward<M, 0, 16>(t[8]);
  t[8].accept_roll = op_accept_roll<M>;
  fill_forward2<M, 0, 16>(t[8]);
}
}  // namespace to
