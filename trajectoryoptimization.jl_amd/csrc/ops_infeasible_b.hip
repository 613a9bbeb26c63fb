// ops_infeasible_b.hip — Altro's InfeasibleModel over the Cartpole (see ops_infeasible_a.hip).
#include "ops.h"

namespace to {
template <class M>
int op_infeasible_controls_b(to_handle* h) {
  hipLaunchKernelGGL(k_infeasible_controls<M>, grid_b(h, h->a.P.N - 1), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
void fill_ops_infeasible_b(ModelOps* t) {
  using M = InfeasibleModel<CartpoleModel>;
  fill_misc<M>(t[11]);
  t[11].expand = op_expand<M>;
  t[11].backward = op_backward<M>;
  t[11].accept_roll = op_accept_roll<M>;
  t[11].infeasible_controls = op_infeasible_controls_b<M>;
  fill_forward<M, 0, 16>(t[11]);
  fill_forward2<M, 0, 16>(t[11]);
}
}  // namespace to
