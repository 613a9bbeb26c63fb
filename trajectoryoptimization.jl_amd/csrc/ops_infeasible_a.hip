// ops_infeasible_a.hip — Altro's InfeasibleModel (models.h; TO_MODEL_INFEASIBLE, the state augmentation of ALTRO's infeasible start)
// over the double integrators: every kernel of the split small-model path (column expansion, cooperative backward pass, both forward
// kernels), each time step through model_step.
#include "ops.h"

namespace to {
template <class M>
int op_infeasible_controls(to_handle* h) {
  hipLaunchKernelGGL(k_infeasible_controls<M>, grid_b(h, h->a.P.N - 1), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
static void fill_one(ModelOps& o) {
  fill_misc<M>(o);
  o.expand = op_expand<M>;
  o.backward = op_backward<M>;
  o.accept_roll = op_accept_roll<M>;
  o.infeasible_controls = op_infeasible_controls<M>;
  fill_forward<M, 0, 16>(o);
  fill_forward2<M, 0, 16>(o);
}
void fill_ops_infeasible_a(ModelOps* t) {
  fill_one<InfeasibleModel<DoubleIntegratorModel<1>>>(t[9]);
  fill_one<InfeasibleModel<DoubleIntegratorModel<2>>>(t[10]);
}
}  // namespace to
