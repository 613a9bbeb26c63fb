// ops.h — launchers of the model-templated kernels; instantiated per model by the ops_*.hip translation units.
#pragma once
#include "handle.h"
#include "k_backward.h"
#include "k_expand.h"
#include "k_forward.h"
#include "k_scan.h"
#include "k_misc.h"

namespace to {

template <class M>
void fill_traits(ModelOps& o) {
  o.write_through = M::accept_write_through;
  o.mfma_backward = M::mfma_backward;
  o.coop_backward = M::coop_backward;
  o.lane_backward = M::lane_backward;
  o.lds_gains = M::lds_gains;
  o.expand_knots = M::expand_knots;
  o.ls_first_round = M::ls_first_round;
  o.gains_lds_pieces = Gains<M>::RSK / 2;
  if constexpr (M::mfma_backward) {  // tangent-matrix layout traits (one 16 x 16 tile per knot: models with m <= 4)
    o.nep = Tm<M>::NEP; o.rs = Tm<M>::RS;
    for (int g = 0; g < 4; ++g)
      for (int c = 0; c < 16; ++c) o.crow[g * 16 + c] = compact_row<M>(g, c);
  }
}

template <class M>
int op_rollout(to_handle* h) {
  bool done = false;
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) { hipLaunchKernelGGL((k_rollout<M, INTEG_RK4>), grid_b(h), dim3(BLOCK), 0, h->stream, h->a); done = true; }
  }
  if (!done) hipLaunchKernelGGL((k_rollout<M, -1>), grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_cost(to_handle* h, int with_al, double* out, double* Jk) {
  hipLaunchKernelGGL(k_cost<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a, with_al, out, Jk);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_violation(to_handle* h, double* out) {
  hipLaunchKernelGGL(k_violation<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a, out);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_dual_update(to_handle* h) {
  hipLaunchKernelGGL(k_dual_update<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
// AL outer update of the trajectories whose inner solve ended in this batch step (k_misc.h, k_outer_*)
template <class M>
int op_outer(to_handle* h) {
  const int N = h->a.P.N;
  hipLaunchKernelGGL(k_outer_violation<M>, grid_b(h, N), dim3(BLOCK), 0, h->stream, h->a);
  hipLaunchKernelGGL(k_outer_decide<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  hipLaunchKernelGGL(k_outer_update<M>, grid_b(h, N), dim3(BLOCK), 0, h->stream, h->a);
  hipLaunchKernelGGL(k_outer_finish<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_cost_derivs(to_handle* h, double* dg, double* dh) {
  hipLaunchKernelGGL(k_cost_derivs<M>, grid_b(h, h->a.P.N), dim3(BLOCK), 0, h->stream, h->a, dg, dh);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_discrete_jacobian(to_handle* h, double* F) {
  const DevProblem& P = h->a.P;
  hipLaunchKernelGGL(k_discrete_jacobian<M>, grid_b(h, P.N - 1, P.n + P.m), dim3(BLOCK), 0, h->stream, h->a, F);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_constraint_eval(to_handle* h, int ci, double* vals, double* jac) {
  const DevCon& c = h->cons[ci];
  hipLaunchKernelGGL(k_constraint_eval<M>, grid_b(h, c.k2 - c.k1 + 1), dim3(BLOCK), 0, h->stream, h->a, ci, vals, jac);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

template <class M>
int op_constraint_hessian(to_handle* h, int ci, const double* lambda, double* H) {
  const DevCon& c = h->cons[ci];
  hipLaunchKernelGGL(k_constraint_hessian<M>, grid_b(h, c.k2 - c.k1 + 1), dim3(BLOCK), 0, h->stream, h->a, ci, lambda, H);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

// expansion variants compiled: 0 = diagonal-kind costs, no constraints; 2 = + selector / SOC-selector constraints;
// 7 = everything.  Layout (k_expand.h LAY): column layout for the cooperative backward pass, tangent-matrix layout
// (full or compact cost block) for the MFMA one, lane layout for the one-lane-per-trajectory one.
template <class M, int FI>
int op_expand_fi(to_handle* h) {
  const DevProblem& P = h->a.P;
  const int var = P.expand_variant == 0 ? 0 : (P.expand_variant == 2 ? 2 : 7);
  const int kc = var == 0 ? expand_kc<M, 0>() : expand_kc<M, 2>();
  const dim3 grid((P.B + h->G - 1) / h->G, (P.N + kc - 1) / kc);
  const int lay = h->a.bwd_lane ? 3 : !h->a.bwd_mfma ? 0 : (h->a.h_compact ? 2 : 1);
  // lane layout: one lane per (trajectory, knot), all columns at once (k_expand_lane; instantiated in the lane translation units, ops_lane.h)
  if (lay == 3 && h->expand_lane && h->ops->expand_lane_k) return h->ops->expand_lane_k(h);
#define TO_EXPAND_CASE(V, LY) \
  if (var == V && lay == LY) { hipLaunchKernelGGL((k_expand<M, FI, V, LY>), grid, dim3(BLOCK), 0, h->stream, h->a); HIPCHECK(hipGetLastError()); return TO_OK; }
  if constexpr (M::mfma_backward && ExpandPack<M>::ok) {  // compact cost block of the quaternion rigid body: the packed expansion (k_expand.h)
    if (lay == 2 && h->expand_pack && (var == 0 || var == 2)) {
      const dim3 pgrid((P.B + EXPAND_PACK_G - 1) / EXPAND_PACK_G, (P.N + kc - 1) / kc);
      if (var == 0) hipLaunchKernelGGL((k_expand<M, FI, 0, 2, true>), pgrid, dim3(BLOCK), 0, h->stream, h->a);
      else hipLaunchKernelGGL((k_expand<M, FI, 2, 2, true>), pgrid, dim3(BLOCK), 0, h->stream, h->a);
      HIPCHECK(hipGetLastError());
      return TO_OK;
    }
  }
  if constexpr (M::mfma_backward) {
    TO_EXPAND_CASE(0, 1) TO_EXPAND_CASE(0, 2) TO_EXPAND_CASE(2, 1) TO_EXPAND_CASE(2, 2) TO_EXPAND_CASE(7, 1)
  }
  if constexpr (!M::mfma_backward || M::coop_backward) {
    TO_EXPAND_CASE(0, 0) TO_EXPAND_CASE(2, 0) TO_EXPAND_CASE(7, 0)
  }
  if constexpr (M::lane_backward) {
    TO_EXPAND_CASE(0, 3) TO_EXPAND_CASE(2, 3) TO_EXPAND_CASE(7, 3)
  }
#undef TO_EXPAND_CASE
  return fail(TO_ERR_UNSUPPORTED, "expansion variant not compiled for this model");
}
template <class M>
int op_expand_const(to_handle* h) {
  const DevProblem& P = h->a.P;
  hipLaunchKernelGGL(k_expand_const_columns<M>, dim3(P.B, (P.N - 1 + 3) / 4), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M>
int op_expand(to_handle* h) {
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) return op_expand_fi<M, INTEG_RK4>(h);
  }
  return op_expand_fi<M, -1>(h);
}
template <class M>
int op_backward(to_handle* h) {
  const DevProblem& P = h->a.P;
  if constexpr (M::mfma_backward) {
    if (h->a.bwd_mfma) {
      if (h->a.h_compact) hipLaunchKernelGGL((k_backward_mfma<M, true>), dim3(P.B), dim3(BLOCK), 0, h->stream, h->a);
      else hipLaunchKernelGGL((k_backward_mfma<M, false>), dim3(P.B), dim3(BLOCK), 0, h->stream, h->a);
      HIPCHECK(hipGetLastError());
      return TO_OK;
    }
  }
  if constexpr (M::lane_backward) {
    if (h->a.bwd_lane) {
      hipLaunchKernelGGL(k_backward_lane<M>, dim3((P.B + 63) / 64), dim3(BLOCK), 0, h->stream, h->a);
      HIPCHECK(hipGetLastError());
      return TO_OK;
    }
  }
  if constexpr (!M::mfma_backward || M::coop_backward) {
    if (h->a.h_diag) hipLaunchKernelGGL((k_backward_coop<M, true>), dim3((P.B + h->G - 1) / h->G), dim3(BLOCK), 0, h->stream, h->a);
    else hipLaunchKernelGGL((k_backward_coop<M, false>), dim3((P.B + h->G - 1) / h->G), dim3(BLOCK), 0, h->stream, h->a);
    HIPCHECK(hipGetLastError());
    return TO_OK;
  }
  return fail(TO_ERR_UNSUPPORTED, "backward-pass variant not compiled for this model");
}

// small batches of the small models with diagonal cost blocks: expansion fused into the cooperative backward pass (k_expand.h)
template <class M, int FI>
int op_expand_backward_coop_fi(to_handle* h) {
  if constexpr (!M::lie && Coop<M>::R <= 8 && (!M::mfma_backward || M::coop_backward)) {
    const DevProblem& P = h->a.P;
    const dim3 grid((P.B + h->G - 1) / h->G);
    const bool mg = h->a.coop_merge != 0;
    if (P.expand_variant == 0 && mg) hipLaunchKernelGGL((k_expand_backward_coop<M, FI, 0, true>), grid, dim3(128), 0, h->stream, h->a);
    else if (P.expand_variant == 0) hipLaunchKernelGGL((k_expand_backward_coop<M, FI, 0, false>), grid, dim3(128), 0, h->stream, h->a);
    else if (P.expand_variant == 2 && mg) hipLaunchKernelGGL((k_expand_backward_coop<M, FI, 2, true>), grid, dim3(128), 0, h->stream, h->a);
    else if (P.expand_variant == 2) hipLaunchKernelGGL((k_expand_backward_coop<M, FI, 2, false>), grid, dim3(128), 0, h->stream, h->a);
    else return fail(TO_ERR_UNSUPPORTED, "fused cooperative pass needs diagonal cost blocks");
    HIPCHECK(hipGetLastError());
    return TO_OK;
  }
  return fail(TO_ERR_UNSUPPORTED, "fused expansion + cooperative backward pass not compiled for this model");
}
template <class M>
int op_expand_backward_coop(to_handle* h) {
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) return op_expand_backward_coop_fi<M, INTEG_RK4>(h);
  }
  return op_expand_backward_coop_fi<M, -1>(h);
}

// fused expansion + scan backward pass (k_scan.h): unconstrained problems with diagonal cost blocks, one wave per trajectory
template <class M>
int op_expand_backward_scan(to_handle* h) {
  if constexpr (!M::lie && M::ne <= 4 && M::m <= 2) {
    const DevProblem& P = h->a.P;
    if (P.expand_variant != 0 || !h->a.h_diag || P.N > 126) return fail(TO_ERR_UNSUPPORTED, "scan backward pass: outside its scope");
    const dim3 grid(P.B);
    if (M::pin_rk4 && P.integrator == INTEG_RK4) {
      if constexpr (M::pin_rk4) hipLaunchKernelGGL((k_expand_backward_scan<M, INTEG_RK4, 0>), grid, dim3(64), 0, h->stream, h->a);
    } else hipLaunchKernelGGL((k_expand_backward_scan<M, -1, 0>), grid, dim3(64), 0, h->stream, h->a);
    HIPCHECK(hipGetLastError());
    return TO_OK;
  }
  return fail(TO_ERR_UNSUPPORTED, "scan backward pass not compiled for this model");
}

// forward pass (line search + state machine) of kernel variant MODE: grid = one wave per TW = 64 / CW trajectories
template <class M, int MODE>
int op_forward(to_handle* h) {
  const KArgs& a = h->a;
  const int TW = a.TW;
  const size_t lds = M::lds_gains ? sizeof(double) * (2 * gains_lds_doubles<M>(TW) + StageCostLds<M::n, M::m>::size) : 0;  // two gains buffers + the stage-cost table
  hipLaunchKernelGGL((k_forward<M, MODE>), dim3((a.P.Bp + TW - 1) / TW), dim3(BLOCK), lds, h->stream, a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

// accepted steps re-rolled from their stored controls (k_accept_roll; models without write-through)
template <class M>
int op_accept_roll(to_handle* h) {
  hipLaunchKernelGGL(k_accept_gather_u<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  bool done = false;
  if constexpr (M::pin_rk4) {
    if (h->a.P.integrator == INTEG_RK4) { hipLaunchKernelGGL((k_accept_roll<M, INTEG_RK4>), grid_b(h), dim3(BLOCK), 0, h->stream, h->a); done = true; }
  }
  if (!done) hipLaunchKernelGGL((k_accept_roll<M, -1>), grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

template <class M, int MODE>
int op_forward2(to_handle* h) {
  const KArgs& a = h->a;
  const int TW = a.TW;
  hipLaunchKernelGGL((k_forward2<M, MODE>), dim3((a.P.Bp + TW - 1) / TW), dim3(128), sizeof(double) * fwd2_lds_doubles<M>(TW), h->stream, a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
template <class M, int LO, int HI>
void fill_forward2(ModelOps& o) {
  if constexpr (LO < HI) {
    if constexpr (M::pin_rk4 || (LO & 4) == 0) o.forward2[LO] = op_forward2<M, LO>;
    fill_forward2<M, LO + 1, HI>(o);
  }
}

template <class M>
void fill_misc(ModelOps& o) {
  fill_traits<M>(o);
  o.rollout = op_rollout<M>; o.cost = op_cost<M>; o.violation = op_violation<M>; o.dual_update = op_dual_update<M>;
  o.outer = op_outer<M>; o.cost_derivs = op_cost_derivs<M>; o.discrete_jacobian = op_discrete_jacobian<M>;
  o.constraint_eval = op_constraint_eval<M>; o.constraint_hessian = op_constraint_hessian<M>;
}
// forward variants [LO, HI): models that do not pin RK4 never run the bit-2 variants
template <class M, int LO, int HI>
void fill_forward(ModelOps& o) {
  if constexpr (LO < HI) {
    if constexpr (M::pin_rk4 || (LO & 4) == 0) o.forward[LO] = op_forward<M, LO>;
    fill_forward<M, LO + 1, HI>(o);
  }
}

}  // namespace to
