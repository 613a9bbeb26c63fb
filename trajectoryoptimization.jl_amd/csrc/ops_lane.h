// ops_lane.h — launchers of the one-lane-per-trajectory expansion kernels (k_expand_lane, k_expand_backward_lane).  Their
// translation units (ops_small_lane.hip; ops_hybrid.hip for the hybrid model) are the only ones that instantiate them:
// ops_small_lane.hip is compiled with -fno-honor-nans -fno-honor-infinities -fno-signed-zeros (build.py), which lets the compiler
// fold the structural zeros of chunk-mode dual numbers (x * 0.0, y + 0.0: IEEE forbids it otherwise — a third of the Cartpole
// expansion's FP64 instructions were products with a literal zero).
#pragma once
#include "ops.h"

namespace to {

template <class M, int FI>
int op_expand_lane_fi(to_handle* h) {
  if constexpr (M::lane_backward && !M::lie) {
    const DevProblem& P = h->a.P;
    const int var = P.expand_variant == 0 ? 0 : (P.expand_variant == 2 ? 2 : 7);
    const dim3 lgrid(P.Bp / BLOCK, P.N);
    if (var == 0) hipLaunchKernelGGL((k_expand_lane<M, FI, 0>), lgrid, dim3(BLOCK), 0, h->stream, h->a);
    else if (var == 2) hipLaunchKernelGGL((k_expand_lane<M, FI, 2>), lgrid, dim3(BLOCK), 0, h->stream, h->a);
    else hipLaunchKernelGGL((k_expand_lane<M, FI, 7>), lgrid, dim3(BLOCK), 0, h->stream, h->a);
    HIPCHECK(hipGetLastError());
    return TO_OK;
  }
  return fail(TO_ERR_UNSUPPORTED, "lane expansion not compiled for this model");
}
template <class M>
int op_expand_lane(to_handle* h) {
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) return op_expand_lane_fi<M, INTEG_RK4>(h);
  }
  return op_expand_lane_fi<M, -1>(h);
}

// large batches of the small models: expansion fused into the one-lane-per-trajectory backward pass (k_expand.h)
template <class M, int FI>
int op_expand_backward_fi(to_handle* h) {
  if constexpr (M::lane_backward && !M::lie) {
    const DevProblem& P = h->a.P;
    const int var = P.expand_variant == 0 ? 0 : (P.expand_variant == 2 ? 2 : 7);
    const dim3 grid(P.Bp / BLOCK);
    if (var == 0) hipLaunchKernelGGL((k_expand_backward_lane<M, FI, 0>), grid, dim3(BLOCK), 0, h->stream, h->a);
    else if (var == 2) hipLaunchKernelGGL((k_expand_backward_lane<M, FI, 2>), grid, dim3(BLOCK), 0, h->stream, h->a);
    else hipLaunchKernelGGL((k_expand_backward_lane<M, FI, 7>), grid, dim3(BLOCK), 0, h->stream, h->a);
    HIPCHECK(hipGetLastError());
    return TO_OK;
  }
  return fail(TO_ERR_UNSUPPORTED, "fused lane expansion + backward pass not compiled for this model");
}
template <class M>
int op_expand_backward(to_handle* h) {
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) return op_expand_backward_fi<M, INTEG_RK4>(h);
  }
  return op_expand_backward_fi<M, -1>(h);
}

}  // namespace to
