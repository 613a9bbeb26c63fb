// kernels.h — HIP kernels of the batched iLQR / AL hot path for gfx950 (MI355X).
//
// Data layout: array-of-structures-of-arrays with a 64-trajectory tile = one wavefront.
// Element e (e.g. e = k*n + i) of trajectory b of an array with L elements per trajectory lives at
//     base[((b/64)*L + e)*64 + (b%64)]
// so (1) the 64 lanes of a wave read/write 512 contiguous bytes per access (fully coalesced), and (2) what one
// wave touches as it walks the knots is ONE contiguous stream of L*512 bytes (sequential pages, prefetcher- and
// TLB-friendly) instead of 32 KB-strided lines.  One lane == one trajectory for the sequential recursions
// (rollout, backward Riccati, forward line search); the expansion kernel adds two more grid axes
// (knot, direction) because it is embarrassingly parallel.  Workgroup = one wave = one tile.
//
// One batch step of a solve = k_expand -> k_backward -> [k_forward (all step sizes of a round concurrently) -> k_select]
// per line-search round -> k_accept (large models) -> k_outer_* (AL).  Per-model choices live in models.h (traits).
#pragma once
#include <hip/hip_runtime.h>

#include "models.h"
#include "problem_dev.h"

namespace to {

struct KArgs {
  DevProblem P;
  double* Xs;     // (T+1) slots of L = N*n: slot 0 holds the nominal trajectory, slot t+1 line-search candidate t of the round
  double* Us;     // (T+1) slots of L = (N-1)*m
  size_t slotX, slotU;  // doubles per slot (L * Bp)
  int T;          // candidate slots = most line-search candidates evaluated concurrently in one round
  double* x0;     // L = n
  int* acc;       // [Bp] slot of the candidate accepted in this forward pass (0 = none): copied onto slot 0 by k_accept, or
                  //      written through by the next k_expand (M::accept_write_through)
  double *candJ, *candG;  // [T][Bp] cost and gradient metric of each candidate of the current round
  int* candOk;            // [T][Bp] 1: rollout stayed within the state/control limits
  int* ls_round;          // [Bp] next line-search round of this trajectory; -1: resolved for this iteration
  double *Mc, *Hc, *gc;               // column layout (see "column layout" below): [Ā B̄], Q-function cost blocks, gradient
  double *K, *d;                      // L = (N-1)*m*ne, (N-1)*m
  double *lam, *mu;                   // L = n_duals, n_cons
  double *J, *dJ, *grad, *rho, *drho, *dV, *cmax, *Jout;  // [Bp] plain (dV: [2][Bp])
  int *status, *iterations, *it_inner, *outer, *dJzero, *ls_index, *active, *budget, *bpfail;
  int* counter;   // [steps] number of trajectories still active after each batch step
  int *oflag, *ost;   // [Bp] AL outer update pending (1: evaluate, 2: update duals) and the inner solve's status
  double* knotbuf;    // tiled, L = N: per-knot scratch of the outer update (violations, then AL cost terms)
  double* mu_next;    // tiled, L = n_cons: penalties after the pending outer update
  int* list;          // [Bp] trajectories still searching, compacted by k_select for the next line-search round
  int* nlist;         // [steps][lstride] length of that list per (batch step, round)
  int lstride;
  int round;      // line-search round being launched: step sizes cand0 .. cand0+Tr-1 (grid.y of k_forward = Tr)
  int cand0, Tr;
  int al_mode;    // 0: iLQR, 1: AL-iLQR
  int control;    // 1: run the solver state machine at the end of the forward pass; 0: phase API
  int step;
};

// per-lane pointer to element 0 of this lane's trajectory in a tiled array with L elements per trajectory;
// element e is then p[e*64] (e wave-uniform -> scalar address arithmetic, immediate offsets for small constants)
#define TILE_PTR(base, L) ((base) + ((size_t)tile * (size_t)(L)) * 64 + lane)
#define EL(p, e) (p)[(size_t)(e) * 64]
#define TILE_LANE() const int tile = blockIdx.x, lane = threadIdx.x, b = tile * 64 + lane
#define XSLOT(a, c) ((a).Xs + (size_t)(c) * (a).slotX)
#define USLOT(a, c) ((a).Us + (size_t)(c) * (a).slotU)

// objective (+AL) value of one knot.  u must be zeros at the terminal knot (the reference evaluates the
// terminal cost with the knot's zero control; src/cost_functions.jl:92-94, test/objective_tests.jl:129).
// lam0 / mu0: this lane's pointers to dual row 0 / penalty 0 (tiled arrays).
// GEN = false: no dense QuadraticCost and no non-selector constraint in the tables (those branches are compiled out)
template <class M, bool GEN = true>
__device__ __forceinline__ double knot_cost(const DevProblem& P, int k, const double* x, const double* u, const double* lam0,
                                            const double* mu0, bool with_al) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  double Jk = cost_eval<n, m, GEN>(P.costs[P.cost_index[k]], x, u);
  if (P.opts.cost_dt_scaling && k < P.N - 1) Jk *= P.dt[k];
  if (with_al && P.n_cons > 0) {
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    double Ja = 0.0;
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (k < K.k1 || k > K.k2) continue;
      const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
      Ja += al_term<n, m, GEN>(K, z, lam, (size_t)64, EL(mu0, ci));
    }
    Jk += Ja;
  }
  return Jk;
}

// AL penalty terms of one knot only
// AL terms of one stage knot with up to two register-cached control-block constraints (ConStage); the others take the
// descriptor-table path.  Terms are summed in constraint order, like knot_al.
template <class M, bool GEN>
__device__ __forceinline__ double knot_al_cached(const DevProblem& P, int k, const double* x, const double* u, const double* lam0,
                                                 const double* mu0, int ncs, const ConStage<M::n, M::m>& c0,
                                                 const ConStage<M::n, M::m>& c1) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  double Ja = 0.0;
  for (int ci = 0; ci < P.n_cons; ++ci) {
    if (ncs > 0 && ci == c0.ci) { Ja += c0.term(u); continue; }
    if (ncs > 1 && ci == c1.ci) { Ja += c1.term(u); continue; }
    ConC& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
    Ja += al_term<n, m, GEN>(K, z, lam, (size_t)64, EL(mu0, ci));
  }
  return Ja;
}

template <class M, bool GEN = true>
__device__ __forceinline__ double knot_al(const DevProblem& P, int k, const double* x, const double* u, const double* lam0, const double* mu0) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = u[i];
  double Ja = 0.0;
  for (int ci = 0; ci < P.n_cons; ++ci) {
    ConC& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
    Ja += al_term<n, m, GEN>(K, z, lam, (size_t)64, EL(mu0, ci));
  }
  return Ja;
}

template <class M>
__device__ __forceinline__ double knot_violation(const DevProblem& P, int k, const double* x, const double* u) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = u[i];
  double vmax = 0.0;
  for (int ci = 0; ci < P.n_cons; ++ci) {
    ConC& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    const double v = con_violation<nz>(K, z);
    if (!(v <= vmax)) vmax = v;
  }
  return vmax;
}

// whole-trajectory pass over the NOMINAL trajectory: cost (with or without AL), max violation, optional dual update
template <class M>
__device__ __forceinline__ void trajectory_pass(const KArgs& a, int tile, int lane, bool with_al, bool do_dual_update, double* J_out,
                                                double* cmax_out, int c = 0) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const DevProblem& P = a.P;
  const int N = P.N;
  const double* X = TILE_PTR(XSLOT(a, c), N * n);
  const double* U = TILE_PTR(USLOT(a, c), (N - 1) * m);
  double* lam0 = TILE_PTR(a.lam, P.n_duals);
  double* mu0 = TILE_PTR(a.mu, P.n_cons);
  double J = 0.0, cmax = 0.0;
  for (int k = 0; k < N; ++k) {
    double x[n], u[m];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
    if (do_dual_update) {
      double z[nz];
#pragma unroll
      for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
      for (int i = 0; i < m; ++i) z[n + i] = u[i];
      for (int ci = 0; ci < P.n_cons; ++ci) {
        ConC& K = P.cons[ci];
        if (k < K.k1 || k > K.k2) continue;
        double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
        con_dual_update<nz>(K, z, lam, (size_t)64, EL(mu0, ci), P.opts.dual_max);
      }
    }
    if (cmax_out && P.n_cons > 0) { const double v = knot_violation<M>(P, k, x, u); if (!(v <= cmax)) cmax = v; }
    if (J_out) J += knot_cost<M>(P, k, x, u, lam0, mu0, with_al);
  }
  if (J_out) *J_out = J;
  if (cmax_out) *cmax_out = cmax;
}

// ------------------------------------------------------------------------------------------------ rollout!
template <class M, int FIXED_INTEG>
__global__ void __launch_bounds__(64) k_rollout(KArgs a) {  // src/problem.jl:334-340 — open-loop simulate from x0
  constexpr int n = M::n, m = M::m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  constexpr int c = 0;  // nominal slot
  double* X = TILE_PTR(XSLOT(a, c), P.N * n);
  const double* U = TILE_PTR(USLOT(a, c), (P.N - 1) * m);
  const double* x0 = TILE_PTR(a.x0, n);
  double x[n], u[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) { x[i] = EL(x0, i); EL(X, i) = x[i]; }
  for (int k = 0; k < P.N - 1; ++k) {
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = EL(U, k * m + i);
    rk_step<M, double, FIXED_INTEG>(P.mp, P.integrator, x, u, P.dt[k], xn);
#pragma unroll
    for (int i = 0; i < n; ++i) { x[i] = xn[i]; EL(X, (k + 1) * n + i) = x[i]; }
  }
}

// ------------------------------------------------------------------------------------------------ cost
// out[b] = total (AL) cost, or — when Jk is given — per-knot objective values in a tiled array with L = N
template <class M>
__global__ void __launch_bounds__(64) k_cost(KArgs a, int with_al, double* out, double* Jk) {
  constexpr int n = M::n, m = M::m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int N = P.N;
  if (Jk) {
    constexpr int c = 0;  // nominal slot
    const double* X = TILE_PTR(XSLOT(a, c), N * n);
    const double* U = TILE_PTR(USLOT(a, c), (N - 1) * m);
    double* o = TILE_PTR(Jk, N);
    for (int k = 0; k < N; ++k) {
      double x[n], u[m];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
      EL(o, k) = knot_cost<M>(P, k, x, u, nullptr, nullptr, false);
    }
    return;
  }
  double J;
  trajectory_pass<M>(a, tile, lane, with_al != 0, false, &J, nullptr);
  out[b] = J;
}

template <class M>
__global__ void __launch_bounds__(64) k_violation(KArgs a, double* out) {
  TILE_LANE();
  if (b >= a.P.B) return;
  double cm;
  trajectory_pass<M>(a, tile, lane, false, false, nullptr, &cm);
  out[b] = cm;
}

template <class M>
__global__ void __launch_bounds__(64) k_dual_update(KArgs a) {
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  trajectory_pass<M>(a, tile, lane, false, true, nullptr, nullptr);
  double* mu0 = TILE_PTR(a.mu, P.n_cons);
  for (int ci = 0; ci < P.n_cons; ++ci) EL(mu0, ci) = fmin(EL(mu0, ci) * P.opts.penalty_scaling, P.opts.penalty_max);
}

// ------------------------------------------------------------------------------------------------ column layout
// The expansion and the backward pass work on COLUMNS of the per-knot blocks: direction j of the error-state tangent
// space [δx (ne); δu (m)] is owned by one lane.  R = ne+m rounded up to a power of two lanes form one trajectory's
// group, G = 64/R trajectories share a wave.  Column arrays hold E entries per (trajectory, column):
//     base[((b/G)*E + e)*64 + (b%G)*R + j]            (64 consecutive doubles = the G x R lanes of one wave)
//   Mc: E = (N-1)*ne      Mc[k*ne + i]       = [Ā B̄]_k[i][j]
//   Hc: E = N*(ne+m)      Hc[k*(ne+m) + i]   = Q-function cost block [Qxx Qxu; Qux Quu]_k[i][j]  (cost + AL, projected)
//   gc: E = N             gc[k]              = [qx; qu]_k[j]
template <class M>
struct Coop {
  static constexpr int ne = M::ne, m = M::m, nc = ne + m;
  static constexpr int R = nc <= 4 ? 4 : nc <= 8 ? 8 : 16;
  static constexpr int G = 64 / R;
};
#define COL_PTR(base, E) ((base) + ((size_t)gtile * (size_t)(E)) * 64 + lane)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ------------------------------------------------------------------------------------------------ expansion
// grid (ceil(B/G), N): lane (g, j) of a wave = trajectory gtile*G+g, direction j.  It produces column j of everything
// the backward pass needs at knot k:
//   [Ā B̄][:,j] = G(x_{k+1})ᵀ · ∂(RK step)/∂z · v_j            (forward-mode dual through all RK stages)
//   H[:,j]      = projected Hessian-vector product of (cost + AL) with v_j,   g[j] = projected gradient component
// with v_j = [G(x_k) e_j; 0] (j<ne) or [0; e_{j-ne}].
// VAR bit0: dense QuadraticCost possible; bit1: constraints present; bit2: non-selector constraints possible.  Code the
// problem cannot reach is compiled out: the all-purpose Quadrotor kernel needed 256 VGPRs + 140 AGPRs + 480 B of scratch.
// one knot of the expansion for lane (g, j): x = x_k, u = u_k (zeros at the terminal knot), x1 = x_{k+1}
template <class M, int FIXED_INTEG, int VAR>
__device__ __forceinline__ void expand_knot(const KArgs& a, int gtile, int lane, int tile, int lane64, int j, int k, bool valid,
                                            const double* x, const double* u, const double* x1) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m, nc = ne + m;
  const DevProblem& P = a.P;
  const int N = P.N;
  const bool terminal = (k == N - 1);
  double v[nz];
  {
    double vx[n];
    errstate_col<M>(x, j < ne ? j : 0, vx);
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (j < ne) ? vx[i] : 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) v[n + i] = (i == j - ne) ? 1.0 : 0.0;
  }
  // duals of (up to two) control-block constraints of this knot are fetched NOW, ahead of the dynamics column: the AL
  // terms below would otherwise pay a memory round trip of their own (the stores in between pin their loads in place)
  constexpr int LR = m + 1;
  double l0[LR], l1[LR], mu0r = 0.0, mu1r = 0.0;
  int lci0 = -1, lci1 = -1;
  if ((VAR & 2) != 0 && P.n_cons > 0) {
    const double* lam0 = a.lam + ((size_t)tile * (size_t)P.n_duals) * 64 + lane64;
    const double* mu0 = a.mu + ((size_t)tile * (size_t)P.n_cons) * 64 + lane64;
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (k < K.k1 || k > K.k2 || K.fast != 2 || K.p > LR) continue;
      const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
      if (lci0 < 0) {
        lci0 = ci; mu0r = EL(mu0, ci);
#pragma unroll
        for (int r = 0; r < LR; ++r) l0[r] = (r < K.p) ? lam[r * 64] : 0.0;
      } else if (lci1 < 0) {
        lci1 = ci; mu1r = EL(mu0, ci);
#pragma unroll
        for (int r = 0; r < LR; ++r) l1[r] = (r < K.p) ? lam[r * 64] : 0.0;
      }
    }
  }
  // ---- dynamics column
  if (!terminal) {
    Dual xd[n], ud[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xd[i] = Dual(x[i], v[i]);
#pragma unroll
    for (int i = 0; i < m; ++i) ud[i] = Dual(u[i], v[n + i]);
    rk_step<M, Dual, FIXED_INTEG>(P.mp, P.integrator, xd, ud, P.dt[k], xn);
    double t[n], col[ne];
#pragma unroll
    for (int i = 0; i < n; ++i) t[i] = xn[i].d;
    errstate_tmul<M>(x1, t, col);
    double* Mc = COL_PTR(a.Mc, (N - 1) * ne);
    if (valid) {
#pragma unroll
      for (int i = 0; i < ne; ++i) EL(Mc, k * ne + i) = col[i];
    }
  }
  // ---- cost (+AL) gradient and Hessian-vector product on the full state
  double gr[nz], y[nz];
  cost_grad_hvp<n, m, (VAR & 1) != 0>(P.costs[P.cost_index[k]], x, u, terminal, v, gr, y);
  if (P.opts.cost_dt_scaling && !terminal) {
    const double h = P.dt[k];
#pragma unroll
    for (int i = 0; i < nz; ++i) { gr[i] *= h; y[i] *= h; }
  }
  if ((VAR & 2) != 0 && P.n_cons > 0) {
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    const double* lam0 = a.lam + ((size_t)tile * (size_t)P.n_duals) * 64 + lane64;
    const double* mu0 = a.mu + ((size_t)tile * (size_t)P.n_cons) * 64 + lane64;
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (k < K.k1 || k > K.k2) continue;
      if (ci == lci0) { al_grad_hvp<n, m, (VAR & 4) != 0, LR>(K, z, l0, 1, mu0r, v, gr, y); continue; }
      if (ci == lci1) { al_grad_hvp<n, m, (VAR & 4) != 0, LR>(K, z, l1, 1, mu1r, v, gr, y); continue; }
      const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
      al_grad_hvp<n, m, (VAR & 4) != 0>(K, z, lam, (size_t)64, EL(mu0, ci), v, gr, y);
    }
  }
  double col[ne], qxe[ne];
  errstate_tmul<M>(x, y, col);
  errstate_tmul<M>(x, gr, qxe);
  if constexpr (M::lie) {  // second-order term of the attitude map: −I₃ (qᵀ ∂J/∂q) on the attitude diagonal
    const double b1 = x[3] * gr[3] + x[4] * gr[4] + x[5] * gr[5] + x[6] * gr[6];
#pragma unroll
    for (int i = 3; i < 6; ++i) col[i] -= (i == j) ? b1 : 0.0;
  }
  double gj = 0.0;
#pragma unroll
  for (int i = 0; i < ne; ++i) gj = (i == j) ? qxe[i] : gj;
#pragma unroll
  for (int r = 0; r < m; ++r) gj = (ne + r == j) ? gr[n + r] : gj;
  if (!valid) return;
  double* Hc = COL_PTR(a.Hc, N * nc);
#pragma unroll
  for (int i = 0; i < ne; ++i) EL(Hc, k * nc + i) = col[i];
#pragma unroll
  for (int r = 0; r < m; ++r) EL(Hc, k * nc + ne + r) = terminal ? 0.0 : y[n + r];
  double* gc = COL_PTR(a.gc, N);
  EL(gc, k) = gj;
}

// A wave walks M::expand_knots consecutive knots and fetches the next knot's state/control while it works on the
// current one: with one wave per SIMD (Quadrotor) nothing else hides the load round trip, which was half of the wave's
// life (rocprof: SQ_WAIT_ANY 49 % of SQ_WAVE_CYCLES, 60 % with AL terms).  x_{k+1} is shared between neighbours.
template <class M, int FIXED_INTEG, int VAR>
__global__ void __launch_bounds__(64) k_expand(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m, KC = M::expand_knots;
  constexpr int R = Coop<M>::R, G = Coop<M>::G;
  const int gtile = blockIdx.x, lane = threadIdx.x;
  const int g = lane / R, j = lane % R;
  const int b = gtile * G + g;
  const DevProblem& P = a.P;
  const int N = P.N;
  const int k0 = blockIdx.y * KC;
  // idle lanes (padding columns, finished trajectories) compute along with EXEC full — partially masked FP64 issues
  // ~1.3x slower on gfx950 — and only their stores are predicated; a wave without any work leaves
  const bool lane_ok = b < P.B && j < nc && a.active[b];
  if (__ballot(lane_ok) == 0) return;
  // Inside a solve the step accepted by the previous forward pass still sits in its candidate slot (acc != 0): the
  // expansion reads it there and writes it through to slot 0, so the separate k_accept copy (and its re-read of every
  // candidate line that any lane of a tile accepted) disappears from the iteration.  Outside a solve acc is 0.
  // (Small models only: for the Quadrotor the gathered reads cost the expansion what k_accept costs — measured.)
  const int c = (M::accept_write_through && b < P.B) ? a.acc[b] : 0;
  const int tile = b >> 6, lane64 = b & 63;
  const double* X = XSLOT(a, c) + ((size_t)tile * (N * n)) * 64 + lane64;
  const double* U = USLOT(a, c) + ((size_t)tile * ((N - 1) * m)) * 64 + lane64;
  double x[n], u[m], x1[n], x2[n], un[m];
#pragma unroll
  for (int i = 0; i < n; ++i) { x[i] = EL(X, k0 * n + i); x1[i] = (k0 + 1 < N) ? EL(X, (k0 + 1) * n + i) : 0.0; }
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = (k0 < N - 1) ? EL(U, k0 * m + i) : 0.0;
#pragma unroll
  for (int kk = 0; kk < KC; ++kk) {
    const int k = k0 + kk;
    if (k >= N) break;
    const bool terminal = (k == N - 1);
    if (KC > 1 && kk + 1 < KC && k + 1 < N) {  // next knot's operands (x_{k+1} is already here)
#pragma unroll
      for (int i = 0; i < n; ++i) x2[i] = (k + 2 < N) ? EL(X, (k + 2) * n + i) : 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) un[i] = (k + 1 < N - 1) ? EL(U, (k + 1) * m + i) : 0.0;
    }
    const bool valid = lane_ok && !(terminal && j >= ne);
    if (M::accept_write_through && c != 0 && j == 0 && valid) {
      double* X0 = XSLOT(a, 0) + ((size_t)tile * (N * n)) * 64 + lane64;
      double* U0 = USLOT(a, 0) + ((size_t)tile * ((N - 1) * m)) * 64 + lane64;
#pragma unroll
      for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
      if (!terminal) {
#pragma unroll
        for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
      }
    }
    expand_knot<M, FIXED_INTEG, VAR>(a, gtile, lane, tile, lane64, j, k, valid, x, u, x1);
    if (KC > 1) {
#pragma unroll
      for (int i = 0; i < n; ++i) { x[i] = x1[i]; x1[i] = x2[i]; }
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = un[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------ regularisation
__device__ __forceinline__ void reg_increase(const to_solver_opts& o, double& rho, double& drho) {
  const double f = o.bp_reg_increase_factor;
  drho = fmax(drho * f, f);
  rho = fmax(rho * drho, o.bp_reg_min);
}
__device__ __forceinline__ void reg_decrease(const to_solver_opts& o, double& rho, double& drho) {
  const double f = o.bp_reg_increase_factor;
  drho = fmin(drho / f, 1.0 / f);
  const double r = rho * drho;
  rho = (r > o.bp_reg_min) ? r : 0.0;
}

// ------------------------------------------------------------------------------------------------ backward pass
// Riccati recursion, cooperative: the R lanes of a trajectory each own one column of
// [Ā B̄] / of the Q-function Hessian; the small dense products exchange operands through LDS (all lanes of a group
// read the same word: broadcast, conflict-free; groups are padded onto different banks).  Control regularisation
// Quu + ρI with a per-trajectory restart on Cholesky failure.  One wave per workgroup, G trajectories per wave.
template <class M>
struct BwdLds {
  static constexpr int ne = M::ne, m = M::m, nc = ne + m, R = Coop<M>::R;
  static constexpr int oS = 0;                 // S[i][r]          ne*ne
  static constexpr int oM = oS + ne * ne;      // Mx[i][j]         ne*R   (also S_new staging)
  static constexpr int oH = oM + ne * R;       // Hu[r][j]         m*R    rows ne.. of the Q-function Hessian = [Qux Quu]
  static constexpr int oK = oH + m * R;        // Kf[r][j]         m*ne
  static constexpr int oG = oK + m * ne;       // g[j]             R
  static constexpr int os = oG + R;            // s[i]             ne
  static constexpr int raw = os + ne;
  static constexpr int stride = raw + ((34 - (raw % 32)) % 32);  // stride % 32 == 2 doubles: groups land on distinct banks
};

template <class M>
__global__ void __launch_bounds__(64) k_backward(KArgs a) {
  constexpr int m = M::m, ne = M::ne, nc = ne + m;
  constexpr int R = Coop<M>::R, G = Coop<M>::G;
  using L = BwdLds<M>;
  __shared__ double lds[G * L::stride];
  const int gtile = blockIdx.x, lane = threadIdx.x;
  const int g = lane / R, j = lane % R;
  const int b = gtile * G + g;
  const DevProblem& P = a.P;
  const int N = P.N;
  // gfx950 issues FP64 VALU ~1.3x slower when EXEC is not all ones (tools/fp64_issue_probe.hip): padding lanes
  // (j >= nc) and finished trajectories run the arithmetic along with everybody else and only their stores are
  // predicated.  glive: this group's trajectory takes part; live: this lane owns one of its columns.
  const bool glive = (b < P.B) && a.active[b < P.B ? b : 0];
  const bool live = glive && (j < nc);
  if (__ballot(live) == 0) return;
  const int jx = j < ne ? j : ne - 1;  // in-range state column for the lanes that own none
  double* S_ = lds + g * L::stride + L::oS;
  double* Mx = lds + g * L::stride + L::oM;
  double* Hu = lds + g * L::stride + L::oH;
  double* Kf = lds + g * L::stride + L::oK;
  double* gl = lds + g * L::stride + L::oG;
  double* sl = lds + g * L::stride + L::os;
  const double* Mc = COL_PTR(a.Mc, (N - 1) * ne);
  const double* Hc = COL_PTR(a.Hc, N * nc);
  const double* gc = COL_PTR(a.gc, N);
  const int tile = b >> 6, lane64 = b & 63;
  double* pK = a.K + ((size_t)tile * ((N - 1) * m * ne)) * 64 + lane64;
  double* pd = a.d + ((size_t)tile * ((N - 1) * m)) * 64 + lane64;
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  int k = N - 2;
  bool init = true, fresh = true;
  double Mn[ne], Hn[nc], gn = 0.0;  // prefetched column of the next knot
  const double *pMk = Mc, *pHk = Hc, *pgk = gc;
  double *pKk = pK, *pdk = pd;
  while (true) {
    if (init) {  // (re)start: S = Qxx_N, s = qx_N
      fresh = true;
      {
        double Sc[ne];
#pragma unroll
        for (int i = 0; i < ne; ++i) Sc[i] = EL(Hc, (N - 1) * nc + i);
        const double s0 = EL(gc, N - 1);
        if (j < ne) {
#pragma unroll
          for (int i = 0; i < ne; ++i) S_[i * ne + j] = Sc[i];
          sl[j] = s0;
        }
      }
      dV0 = 0.0; dV1 = 0.0; k = N - 2; init = false;
      // per-knot pointers walk backwards with the recursion: constant offsets instead of 64-bit address arithmetic per load
      pMk = Mc + (size_t)(N - 2) * ne * 64; pHk = Hc + (size_t)(N - 2) * nc * 64; pgk = gc + (size_t)(N - 2) * 64;
      pKk = pK + (size_t)(N - 2) * m * ne * 64; pdk = pd + (size_t)(N - 2) * m * 64;
      WAVE_SYNC();
    }
    if (k < 0) break;
    // 1. own column of [Ā B̄] and of the cost blocks (fetched one knot ahead: nothing else hides the load latency)
    double Mj[ne], Hj[nc], gj;
    if (fresh) {
#pragma unroll
      for (int i = 0; i < ne; ++i) Mn[i] = EL(pMk, i);
#pragma unroll
      for (int i = 0; i < nc; ++i) Hn[i] = EL(pHk, i);
      gn = EL(pgk, 0);
      fresh = false;
    }
#pragma unroll
    for (int i = 0; i < ne; ++i) Mj[i] = Mn[i];
#pragma unroll
    for (int i = 0; i < nc; ++i) Hj[i] = Hn[i];
    gj = gn;
    if constexpr (ne > 6) fresh = true;  // large models: the extra live registers cost more than the latency they hide
    else if (k > 0) {
#pragma unroll
      for (int i = 0; i < ne; ++i) Mn[i] = (pMk - ne * 64)[(size_t)i * 64];
#pragma unroll
      for (int i = 0; i < nc; ++i) Hn[i] = (pHk - nc * 64)[(size_t)i * 64];
      gn = (pgk - 64)[0];
    }
#pragma unroll
    for (int i = 0; i < ne; ++i) Mx[i * R + j] = Mj[i];
    WAVE_SYNC();
    // 2. T = S M[:,j];   H[:,j] += Mᵀ T;   g_j += M[:,j]·s
    double Tj[ne];
#pragma unroll
    for (int i = 0; i < ne; ++i) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < ne; ++r) t += S_[i * ne + r] * Mj[r];
      Tj[i] = t;
    }
#pragma unroll
    for (int i = 0; i < nc; ++i) {
      double t = Hj[i];
#pragma unroll
      for (int r = 0; r < ne; ++r) t += Mx[r * R + i] * Tj[r];
      Hj[i] = t;
    }
#pragma unroll
    for (int r = 0; r < ne; ++r) gj += Mj[r] * sl[r];
    // 3. publish the control rows [Qux Quu] and the gradient
#pragma unroll
    for (int r = 0; r < m; ++r) Hu[r * R + j] = Hj[ne + r];
    gl[j] = gj;
    WAVE_SYNC();
    // 4. every lane factors Quu + ρI (m x m) redundantly
    double Quu[m][m], Lc[m][m], Qu[m];
#pragma unroll
    for (int r = 0; r < m; ++r) {
#pragma unroll
      for (int q = 0; q < m; ++q) Quu[r][q] = Hu[r * R + ne + q];
      Qu[r] = gl[ne + r];
    }
    bool pd_ok = true;
    double iL[m];  // reciprocals of the Cholesky diagonal: every later division becomes a product
#pragma unroll
    for (int r = 0; r < m; ++r)
#pragma unroll
      for (int q = 0; q < m; ++q) Lc[r][q] = Quu[r][q] + ((r == q) ? rho : 0.0);
#pragma unroll
    for (int q = 0; q < m; ++q) {
      double sj = Lc[q][q];
#pragma unroll
      for (int r = 0; r < q; ++r) sj -= Lc[q][r] * Lc[q][r];
      if (!(sj > 0.0) && glive) pd_ok = false;  // groups that only ride along never restart
      const double l = sqrt(sj);
      Lc[q][q] = l;
      iL[q] = rcp_fast(l);
#pragma unroll
      for (int i = q + 1; i < m; ++i) {
        double t = Lc[i][q];
#pragma unroll
        for (int r = 0; r < q; ++r) t -= Lc[i][r] * Lc[q][r];
        Lc[i][q] = t * iL[q];
      }
    }
    if (!pd_ok) {  // same decision in every lane of the group
      reg_increase(P.opts, rho, drho);
      if (rho > P.opts.bp_reg_max) { failed = true; break; }
      init = true;
      continue;
    }
    // 5. gains: own column of K = −(LLᵀ)⁻¹ Qux, and d = −(LLᵀ)⁻¹ Qu (redundant)
    double Kj[m], dk[m];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      double col[m];
#pragma unroll
      for (int i = 0; i < m; ++i) col[i] = pass ? Qu[i] : Hj[ne + i];
#pragma unroll
      for (int i = 0; i < m; ++i) { double t = col[i];
#pragma unroll
        for (int r = 0; r < i; ++r) t -= Lc[i][r] * col[r];
        col[i] = t * iL[i]; }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) { double t = col[i];
#pragma unroll
        for (int r = i + 1; r < m; ++r) t -= Lc[r][i] * col[r];
        col[i] = t * iL[i]; }
#pragma unroll
      for (int i = 0; i < m; ++i) { if (pass) dk[i] = -col[i]; else Kj[i] = -col[i]; }
    }
    if (j < ne) {
#pragma unroll
      for (int r = 0; r < m; ++r) { if constexpr (m > 1) Kf[r * ne + j] = Kj[r]; if (glive) EL(pKk, r * ne + j) = Kj[r]; }
    }
    if (j == 0 && glive) {
#pragma unroll
      for (int r = 0; r < m; ++r) EL(pdk, r) = dk[r];
    }
    // single-input models: every lane rebuilds the other columns' gains from the published Qux row (same two products
    // as the owner lane, bit for bit) instead of exchanging K through LDS — one barrier round less per knot
    if constexpr (m > 1) WAVE_SYNC();
    // 6. cost-to-go with the un-regularised Quu:  S' = Qxx + Kᵀ(Quu K + Qux) + Quxᵀ K,  s' = Qx + Kᵀ(Quu d + Qu) + Quxᵀ d
    double Snew[ne], snew = 0.0;
    {
      double Wj[m], qd[m];
#pragma unroll
      for (int r = 0; r < m; ++r) {
        double t = Hj[ne + r], t2 = Qu[r];
#pragma unroll
        for (int q = 0; q < m; ++q) { t += Quu[r][q] * Kj[q]; t2 += Quu[r][q] * dk[q]; }
        Wj[r] = t; qd[r] = t2;
      }
#pragma unroll
      for (int i = 0; i < ne; ++i) {
        double t = Hj[i];
#pragma unroll
        for (int r = 0; r < m; ++r) t += ((m > 1) ? Kf[r * ne + i] : -((Hu[i] * iL[0]) * iL[0])) * Wj[r];
#pragma unroll
        for (int r = 0; r < m; ++r) t += Hu[r * R + i] * Kj[r];
        Snew[i] = t;
      }
      snew = gj;
#pragma unroll
      for (int r = 0; r < m; ++r) snew += Kj[r] * qd[r];
#pragma unroll
      for (int r = 0; r < m; ++r) snew += Hj[ne + r] * dk[r];
#pragma unroll
      for (int i = 0; i < ne; ++i) Mx[i * R + j] = Snew[i];  // stage S' for the symmetrisation
    }
    double dv1 = 0.0, dv2 = 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r) {
      dv1 += dk[r] * Qu[r];
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < m; ++q) t += Quu[r][q] * dk[q];
      dv2 += dk[r] * t;
    }
    dV0 += dv1;
    dV1 += 0.5 * dv2;
    WAVE_SYNC();
    {
      double Ss[ne];
#pragma unroll
      for (int i = 0; i < ne; ++i) Ss[i] = 0.5 * (Snew[i] + Mx[jx * R + i]);
      if (j < ne) {
#pragma unroll
        for (int i = 0; i < ne; ++i) S_[i * ne + j] = Ss[i];
        sl[j] = snew;
      }
    }
    WAVE_SYNC();
    --k;
    pMk -= ne * 64; pHk -= nc * 64; pgk -= 64; pKk -= m * ne * 64; pdk -= m * 64;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  if (j == 0 && glive) {
    a.rho[b] = rho;
    a.drho[b] = drho;
    a.dV[b] = dV0;
    a.dV[(size_t)P.Bp + b] = dV1;
    a.bpfail[b] = failed ? 1 : 0;
    a.ls_round[b] = 0;  // arm the line search of this iteration
  }
}

// ------------------------------------------------------------------------------------------------ forward pass
template <class M, bool WITHK>
struct FwdKnot {  // nominal state/control and gains of one knot, fetched one knot ahead of their use
  static constexpr int n = M::n, m = M::m, ne = M::ne;
  double x[n], u[m], K[WITHK ? m : 1][WITHK ? ne : 1], d[m];
  // pointers are already at this knot (the caller walks them): constant offsets, no address arithmetic per load
  __device__ __forceinline__ void load(const double* pXk, const double* pUk, const double* pKk, const double* pdk) {
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = EL(pXk, i);
#pragma unroll
    for (int j = 0; j < m; ++j) {
      u[j] = EL(pUk, j);
      d[j] = EL(pdk, j);
      if constexpr (WITHK) {
#pragma unroll
        for (int i = 0; i < ne; ++i) K[j][i] = EL(pKk, j * ne + i);
      }
    }
  }
};

// Gains of one knot, global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, the wave keeps computing).
// The m*ne rows of a knot are contiguous in a tile (512 B each); one instruction moves two rows: lane L fetches 16 B at
// rows + 16 L and the hardware writes them at kbuf + 16 L.  ktile = this tile's K base (wave-uniform).
template <class M>
__device__ __forceinline__ void stage_gains(const double* ktile, int k, double* kbuf, int lane) {
  constexpr int rows = M::m * M::ne;
  static_assert(rows % 2 == 0, "two 512-byte rows per DMA instruction");
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* src = (const char*)(ktile + (size_t)k * rows * 64) + lane * 16;
#pragma unroll
  for (int r2 = 0; r2 < rows / 2; ++r2)
    __builtin_amdgcn_global_load_lds((gptr_t)(src + r2 * 1024), (lptr_t)((char*)kbuf + r2 * 1024), 16, 0, 0);
}

// Line-search candidate: closed-loop rollout with step alpha = decrease^(round*T + t) of trajectory b into slot t+1,
// its cost and gradient metric.  grid = (tiles, T): every step size of the round is evaluated
// CONCURRENTLY by its own wave — a sequential backtracking search would cost (deepest search in the batch) x one
// rollout per iteration, while the machine idles.
// MODE bit0: simple_stage (stage cost preloaded into registers, uniform dt); bit1: constraints present (AL terms);
// bit2: RK4 fixed at compile time; bit3: dense costs / non-selector constraints possible (else compiled out).
// One line-search candidate for the 64 trajectories (tile, lane) of a wave: closed-loop rollout with step size alpha
// into slot cs, its (AL) cost J, gradient metric gsum/(N-1) and admissibility ok.  Lanes with live == false roll out as
// well (see below) but store nothing.  kbuf: the wave's LDS buffer for DMA-staged gains (KLDS models).
template <class M, int MODE>
__device__ __forceinline__ void forward_candidate(const KArgs& a, int tile, int lane, bool live, double alpha, int cs, double* kbuf,
                                                  double& J_out, double& g_out, bool& ok_out) {
  constexpr int n = M::n, m = M::m, ne = M::ne;
  constexpr bool SIMPLE = (MODE & 1) != 0, CONS = (MODE & 2) != 0, GEN = (MODE & 8) != 0, LISTED = (MODE & 16) != 0;
  constexpr bool KLDS = M::lds_gains && !LISTED;
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  const int N = P.N;
  constexpr int c = 0;  // nominal slot
  const double* Xc = TILE_PTR(XSLOT(a, c), N * n);
  const double* Uc = TILE_PTR(USLOT(a, c), (N - 1) * m);
  double* Xn = TILE_PTR(XSLOT(a, cs), N * n);
  double* Un = TILE_PTR(USLOT(a, cs), (N - 1) * m);
  const double* pK = TILE_PTR(a.K, (N - 1) * m * ne);
  const double* pd = TILE_PTR(a.d, (N - 1) * m);
  const double* px0 = TILE_PTR(a.x0, n);
  const double* lam0 = TILE_PTR(a.lam, P.n_duals);
  const double* mu0 = TILE_PTR(a.mu, P.n_cons);
  // everything wave-uniform the loop needs is fetched ONCE: an in-order wave stalls on every scalar-load round trip
  double mp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) mp[i] = in_vgpr(P.mp[i]);
  const int integrator = P.integrator;
  const bool dt_scaling = P.opts.cost_dt_scaling != 0;
  const double max_x = o.max_state_value, max_u = o.max_control_value;
  StageCostDiag<n, m> sc;
  double h0 = 0.0;
  if constexpr (SIMPLE) { sc.load(P.costs[P.cost_index[0]]); h0 = P.dt[0]; }
  // stage constraints on the control block (norm / SOC / one-sided bounds on u) are cached in registers once
  ConStage<n, m> cs0, cs1;
  int ncs = 0, uncached = 0;
  cs0.ci = -1; cs1.ci = -1;
  if constexpr (CONS) {
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (K.fast == 2 && K.k1 == 0 && K.k2 >= N - 2 && K.p <= m + 1 && ncs < 2) {
        if (ncs == 0) cs0.load(K, ci, lam0, mu0); else cs1.load(K, ci, lam0, mu0);
        ++ncs;
      } else if (K.k1 <= N - 2) ++uncached;  // applies to some stage knot: needs the descriptor-table path
    }
  }
  double xb[n], J = 0.0, gsum = 0.0;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < n; ++i) { xb[i] = EL(px0, i); if (live) EL(Xn, i) = xb[i]; }
  const double* ktile = a.K + ((size_t)tile * ((N - 1) * m * ne)) * 64;
  if constexpr (KLDS) stage_gains<M>(ktile, 0, kbuf, lane);
  FwdKnot<M, !KLDS> nxt;
  nxt.load(Xc, Uc, pK, pd);
  const double *pXn = Xc + n * 64, *pUn = Uc + m * 64, *pKn = pK + m * ne * 64, *pdn = pd + m * 64;  // knot k+1 of the nominal
  double *pXo = Xn + n * 64, *pUo = Un;                                                              // where knot k's results go
  const bool all_cached = CONS && uncached == 0;  // wave-uniform: the loop then never touches the descriptor table
  if (ncs > 0) cs0.prefetch(0);
  if (ncs > 1) cs1.prefetch(0);
  for (int k = 0; k < N - 1; ++k) {
    const FwdKnot<M, !KLDS> cur = nxt;
    if (ncs > 0) cs0.advance();
    if (ncs > 1) cs1.advance();
    // software prefetch of the next knot.  With DMA-staged gains the wave waits for vmcnt(0) below, which would also
    // wait for loads issued here: those models issue the whole next knot AFTER that wait (one knot of compute to land).
    if constexpr (!KLDS) {
      if (k + 1 < N - 1) {
        nxt.load(pXn, pUn, pKn, pdn);
        if (ncs > 0) cs0.prefetch(k + 1);
        if (ncs > 1) cs1.prefetch(k + 1);
      }
      pXn += n * 64; pUn += m * 64; pKn += m * ne * 64; pdn += m * 64;
    }
    double dx[ne], ub[m], xn[n];
    state_diff<M>(xb, cur.x, dx);
    if constexpr (KLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this knot's gains have landed in LDS
    double gk = 0.0, du[m];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      du[j] = cur.d[j] * alpha;
#pragma unroll
      for (int i = 0; i < ne; ++i) du[j] += (KLDS ? kbuf[(j * ne + i) * 64 + lane] : cur.K[KLDS ? 0 : j][KLDS ? 0 : i]) * dx[i];
    }
    if constexpr (KLDS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane has read the buffer before the DMA refills it
      if (k + 1 < N - 1) {
        stage_gains<M>(ktile, k + 1, kbuf, lane);
        nxt.load(pXn, pUn, pKn, pdn);
        if (ncs > 0) cs0.prefetch(k + 1);
        if (ncs > 1) cs1.prefetch(k + 1);
      }
      pXn += n * 64; pUn += m * 64; pKn += m * ne * 64; pdn += m * 64;
    }
#pragma unroll
    for (int j = 0; j < m; ++j) {
      ub[j] = cur.u[j] + du[j];
      if (live) EL(pUo, j) = ub[j];
      gk = fmax(gk, fabs(cur.d[j]) * rcp_fast(fabs(ub[j]) + 1.0));
    }
    gsum += gk;
    const double h = SIMPLE ? h0 : P.dt[k];
    double Jk = SIMPLE ? sc.eval(xb, ub) : cost_eval<n, m, GEN>(P.costs[P.cost_index[k]], xb, ub);
    if (dt_scaling) Jk *= h;
    if constexpr (CONS) {
      if (all_cached) {
        double Ja = 0.0;
        if (ncs > 0) Ja += cs0.term(ub);
        if (ncs > 1) Ja += cs1.term(ub);
        Jk += Ja;
      } else Jk += knot_al_cached<M, GEN>(P, k, xb, ub, lam0, mu0, ncs, cs0, cs1);
    }
    J += Jk;
    rk_step<M, double, (MODE & 4) ? INTEG_RK4 : -1>(mp, integrator, xb, ub, h, xn);
    double mx = 0.0, mu_ = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) { xb[i] = xn[i]; if (live) EL(pXo, i) = xn[i]; const double v = fabs(xn[i]); if (!(v <= mx)) mx = v; }
    pXo += n * 64; pUo += m * 64;
#pragma unroll
    for (int j = 0; j < m; ++j) { const double v = fabs(ub[j]); if (!(v <= mu_)) mu_ = v; }
    // a rollout that left the admissible box is rejected; its lane keeps stepping (values are never used) so that the
    // wave stays converged, and the wave stops once no live lane is inside the box any more
    if (!(mx <= max_x) || !(mu_ <= max_u)) ok = false;
    if (__ballot(live && ok) == 0) break;
  }
  {
    double u0[m];
#pragma unroll
    for (int j = 0; j < m; ++j) u0[j] = 0.0;
    J += knot_cost<M, GEN>(P, N - 1, xb, u0, lam0, mu0, true);
  }
  J_out = J; g_out = gsum / (N - 1); ok_out = ok;
}

template <class M, int MODE>
__global__ void __launch_bounds__(64) k_forward(KArgs a) {
  constexpr int m = M::m, ne = M::ne;
  constexpr bool LISTED = (MODE & 16) != 0;
  constexpr bool KLDS = M::lds_gains && !LISTED;  // gains staged through LDS by DMA instead of prefetch registers
  __shared__ double kbuf[KLDS ? m * ne * 64 : 1];
  const DevProblem& P = a.P;
  const int t = blockIdx.y;
  const int idx = a.cand0 + t;
  const to_solver_opts& o = P.opts;
  if (idx >= o.iterations_linesearch) return;
  // Round 0: lane = trajectory of tile blockIdx.x.  Later rounds: the few trajectories that rejected every step size so
  // far were compacted into a list by k_select; wave v takes entries 64v..64v+63 (gathered loads, but a handful of
  // waves instead of one per tile that still holds a searching trajectory).
  int b = blockIdx.x * 64 + threadIdx.x;
  bool listed = true;
  if constexpr (LISTED) {
    const int cnt = a.nlist[a.step * a.lstride + a.round];
    if ((int)blockIdx.x * 64 >= cnt) return;
    listed = b < cnt;
    b = a.list[listed ? b : cnt - 1];
  }
  const int tile = b >> 6, lane = b & 63;
  // gfx950 issues FP64 VALU ~1.3x slower whenever EXEC is not all ones (tools/fp64_issue_probe.hip), so lanes that have
  // nothing to do are NOT masked off: they roll out their own (valid) trajectory as well and only their stores are
  // predicated.  The wave leaves only when no lane needs the candidate.
  const bool live = listed && b < P.B && a.active[b] && !a.bpfail[b] && a.ls_round[b] == a.round;
  if (__ballot(live) == 0) return;
  double alpha = 1.0;
  for (int i = 0; i < idx; ++i) alpha *= o.line_search_decrease_factor;  // same product the sequential search forms
  double J, gm;
  bool ok;
  forward_candidate<M, MODE>(a, tile, lane, live, alpha, t + 1, kbuf, J, gm, ok);
  if (!live) return;
  const size_t ci = (size_t)t * P.Bp + b;
  a.candJ[ci] = J;
  a.candG[ci] = gm;
  a.candOk[ci] = ok ? 1 : 0;
}

// Picks the FIRST accepted step size of the round (identical to sequential backtracking), then — when
// a.control — runs the per-trajectory solver state machine: convergence test (row S3) and the AL outer update (row S4).
// Small models (M::tail_in_select): the step sizes beyond the first round are not given launches of their own (for the
// Cartpole workload they never had any work) — the rare trajectory that rejects the whole first round finishes its
// search right here, sequentially, with the same rollout code (MODE as in k_forward; 0 for the other models).
template <class M, int MODE>
__global__ void __launch_bounds__(64) k_select(KArgs a) {
  constexpr int m = M::m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  if (!a.active[b]) return;        // a finished trajectory keeps its last accepted slot marked until the final k_accept
  if (a.round == 0) a.acc[b] = 0;  // (the previous step's slot was written through to slot 0 by this step's expansion)
  const int N = P.N;
  const to_solver_opts& o = P.opts;
  const bool bpfail = a.bpfail[b] != 0;
  if (!bpfail && a.ls_round[b] != a.round) return;  // already resolved in an earlier round of this iteration
  constexpr int c = 0;  // nominal slot
  int acc = 0;          // slot of the accepted candidate: the new nominal trajectory until k_accept has copied it to slot 0
  double* mu0 = TILE_PTR(a.mu, P.n_cons);
  const double Jprev = a.J[b];
  double rho = a.rho[b], drho = a.drho[b];
  double Jnew = Jprev, grad = 0.0;
  int accepted = -1;
  bool zero_step = false;  // stationary point: the zero step was "accepted" (ls_index 0, J unchanged)
  if (!bpfail) {
    const double dV0 = a.dV[b], dV1 = a.dV[(size_t)P.Bp + b];
    double alpha = 1.0;
    for (int i = 0; i < a.cand0; ++i) alpha *= o.line_search_decrease_factor;
    bool exhausted = false;
    // stationary point (predicted improvement ~ rounding noise): take the zero step, dJ = 0 => converged
    const bool stationary = -(dV0 + dV1) <= 1e-12 * (1.0 + fabs(Jprev));
    if (stationary) {
      accepted = 0; zero_step = true;
      const double* Uc = TILE_PTR(USLOT(a, c), (N - 1) * m);
      const double* pd = TILE_PTR(a.d, (N - 1) * m);
      double gs = 0.0;
      for (int k = 0; k < N - 1; ++k) {
        double gk = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) gk = fmax(gk, fabs(EL(pd, k * m + j)) * rcp_fast(fabs(EL(Uc, k * m + j)) + 1.0));
        gs += gk;
      }
      grad = gs / (N - 1);
    }
    for (int t = 0; t < a.Tr && !stationary; ++t) {
      const int idx = a.cand0 + t;
      if (idx >= o.iterations_linesearch) { exhausted = true; break; }
      const size_t ci = (size_t)t * P.Bp + b;
      if (a.candOk[ci]) {
        const double J = a.candJ[ci];
        const double expected = -alpha * (dV0 + alpha * dV1);
        const double z = (expected > 0.0) ? (Jprev - J) / expected : -1.0;
        if (z >= o.line_search_lower_bound && z <= o.line_search_upper_bound) {
          accepted = idx; Jnew = J; grad = a.candG[ci];
          acc = t + 1;
          break;
        }
      }
      alpha *= o.line_search_decrease_factor;
    }
    if constexpr (M::tail_in_select) {
      // sequential tail of the search (exactly what backtracking does); candidates go to slot 1: all of this
      // trajectory's first-round slots hold rejected steps
      bool need = accepted < 0 && !exhausted;
      for (int idx = a.cand0 + a.Tr; idx < o.iterations_linesearch && __ballot(need) != 0; ++idx) {
        double J, gm, kdummy[1];
        bool ok;
        forward_candidate<M, MODE>(a, tile, lane, need, alpha, 1, kdummy, J, gm, ok);
        if (need && ok) {
          const double expected = -alpha * (dV0 + alpha * dV1);
          const double z = (expected > 0.0) ? (Jprev - J) / expected : -1.0;
          if (z >= o.line_search_lower_bound && z <= o.line_search_upper_bound) {
            accepted = idx; Jnew = J; grad = gm; acc = 1; need = false;
          }
        }
        alpha *= o.line_search_decrease_factor;
      }
      exhausted = true;
    }
    if (accepted < 0) {
      if (!exhausted && a.cand0 + a.Tr < o.iterations_linesearch) {  // next round: join the compacted list
        a.ls_round[b] = a.round + 1;
        // one atomic per wave: the tile's entries stay adjacent, so the gathered loads of the next round still share lines
        const unsigned long long grp = __ballot(1);
        const int leader = __ffsll((long long)grp) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&a.nlist[a.step * a.lstride + a.round + 1], __popcll(grp));
        base = __shfl(base, leader);
        a.list[base + __popcll(grp & ((1ull << lane) - 1ull))] = b;
        return;
      }
      // line search failed: gradient metric on the unchanged nominal controls, regularise harder
      const double* Uc = TILE_PTR(USLOT(a, c), (N - 1) * m);
      const double* pd = TILE_PTR(a.d, (N - 1) * m);
      double gs = 0.0;
      for (int k = 0; k < N - 1; ++k) {
        double gk = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) gk = fmax(gk, fabs(EL(pd, k * m + j)) * rcp_fast(fabs(EL(Uc, k * m + j)) + 1.0));
        gs += gk;
      }
      grad = gs / (N - 1);
      reg_increase(o, rho, drho);
      rho += o.bp_reg_fp;
    }
  }
  a.ls_round[b] = -1;
  a.ls_index[b] = accepted;
  a.acc[b] = acc;
  if (!a.control) {  // phase API: report and leave the state machine alone
    a.Jout[b] = Jnew;
    a.rho[b] = rho; a.drho[b] = drho;
    if (accepted >= 0) a.J[b] = Jnew;
    return;
  }
  // ---------------- solver state machine ----------------
  int st = TO_UNSOLVED;
  bool inner_done = false;
  const double cost_tol = a.al_mode ? o.cost_tolerance_intermediate : o.cost_tolerance;
  if (bpfail) { st = TO_REGULARIZATION_MAX; inner_done = true; }
  else {
    const bool ls_failed = accepted < 0;
    const double dJ = Jprev - Jnew;
    int dz = a.dJzero[b];
    dz = (ls_failed || zero_step) ? dz + 1 : 0;  // a zero step makes no progress either
    a.dJzero[b] = dz;
    a.dJ[b] = dJ; a.grad[b] = grad; a.J[b] = Jnew;
    const int its = a.iterations[b] + 1, iti = a.it_inner[b] + 1;
    a.iterations[b] = its; a.it_inner[b] = iti;
    if (rho > o.bp_reg_max) { st = TO_REGULARIZATION_MAX; inner_done = true; }
    else if (dJ >= 0.0 && dJ < cost_tol && grad < o.gradient_tolerance && !ls_failed) { st = TO_SOLVE_SUCCEEDED; inner_done = true; }
    else if (iti >= a.budget[b]) { st = TO_MAX_ITERATIONS; inner_done = true; }
    else if (dz > o.dJ_counter_limit) { st = TO_NO_PROGRESS; inner_done = true; }
    else if (!(Jnew <= o.max_cost_value)) { st = TO_MAXIMUM_COST; inner_done = true; }
  }
  bool still_active = true;
  if (inner_done) {
    if (!a.al_mode) { a.status[b] = st; still_active = false; }
    else {  // AL outer update: whole-trajectory passes, run knot-parallel by the k_outer_* kernels after k_accept
      a.ost[b] = st; a.oflag[b] = 1;
      a.rho[b] = rho; a.drho[b] = drho;
      return;
    }
  }
  a.rho[b] = rho; a.drho[b] = drho;
  if (!still_active) a.active[b] = 0;
  else atomicAdd(&a.counter[a.step], 1);
}

// ---- AL outer update of the trajectories whose inner solve just ended (oflag = 1), knot-parallel:
//   k_outer_violation (tiles, N): constraint violation of every knot -> knotbuf
//   k_outer_decide    (tiles)   : c_max, termination tests; trajectories that go on get oflag = 2 and their new penalties
//   k_outer_update    (tiles, N): dual update with the OLD penalties, then the knot's AL cost with the new duals/penalties
//   k_outer_finish    (tiles)   : J = sum of the knot terms in knot order (same sum as a sequential pass), restart the inner solve
// One lane per trajectory walking all knots three times was the largest kernel of the constrained solves.
template <class M>
__global__ void __launch_bounds__(64) k_outer_violation(KArgs a) {
  constexpr int n = M::n, m = M::m;
  TILE_LANE();
  const DevProblem& P = a.P;
  const bool want = b < P.B && a.oflag[b] == 1;
  if (__ballot(want) == 0) return;
  if (!want) return;
  const int N = P.N, k = blockIdx.y;
  const int sl = a.acc[b];  // the step accepted in this iteration (0: none, nominal unchanged)
  const double* X = TILE_PTR(XSLOT(a, sl), N * n);
  const double* U = TILE_PTR(USLOT(a, sl), (N - 1) * m);
  double x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
  EL(TILE_PTR(a.knotbuf, N), k) = (P.n_cons > 0) ? knot_violation<M>(P, k, x, u) : 0.0;
}

template <class M>
__global__ void __launch_bounds__(64) k_outer_decide(KArgs a) {
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B || a.oflag[b] != 1) return;
  const to_solver_opts& o = P.opts;
  const int N = P.N, st = a.ost[b];
  const int outer = a.outer[b] + 1;
  a.outer[b] = outer;
  const double* vb = TILE_PTR(a.knotbuf, N);
  double cm = 0.0;
  for (int k = 0; k < N; ++k) { const double v = EL(vb, k); if (!(v <= cm)) cm = v; }
  a.cmax[b] = cm;
  const int its = a.iterations[b];
  bool go_on = false;
  if (st != TO_SOLVE_SUCCEEDED && st != TO_MAX_ITERATIONS && st != TO_NO_PROGRESS) a.status[b] = st;
  else if (cm < o.constraint_tolerance) a.status[b] = TO_SOLVE_SUCCEEDED;
  else if (its >= o.iterations_total) a.status[b] = TO_MAX_ITERATIONS;
  else if (outer >= o.iterations_outer) a.status[b] = TO_MAX_ITERATIONS_OUTER;
  else go_on = true;
  if (!go_on) { a.active[b] = 0; a.oflag[b] = 0; return; }
  a.oflag[b] = 2;
  const double* mu0 = TILE_PTR(a.mu, P.n_cons);
  double* mn0 = TILE_PTR(a.mu_next, P.n_cons);
  for (int ci = 0; ci < P.n_cons; ++ci) EL(mn0, ci) = fmin(EL(mu0, ci) * o.penalty_scaling, o.penalty_max);
}

template <class M>
__global__ void __launch_bounds__(64) k_outer_update(KArgs a) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  TILE_LANE();
  const DevProblem& P = a.P;
  const bool want = b < P.B && a.oflag[b] == 2;
  if (__ballot(want) == 0) return;
  if (!want) return;
  const int N = P.N, k = blockIdx.y;
  const int sl = a.acc[b];
  const double* X = TILE_PTR(XSLOT(a, sl), N * n);
  const double* U = TILE_PTR(USLOT(a, sl), (N - 1) * m);
  double* lam0 = TILE_PTR(a.lam, P.n_duals);
  const double* mu0 = TILE_PTR(a.mu, P.n_cons);
  const double* mn0 = TILE_PTR(a.mu_next, P.n_cons);
  double x[n], u[m], z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) { x[i] = EL(X, k * n + i); z[i] = x[i]; }
#pragma unroll
  for (int i = 0; i < m; ++i) { u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0; z[n + i] = u[i]; }
  for (int ci = 0; ci < P.n_cons; ++ci) {
    ConC& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
    con_dual_update<nz>(K, z, lam, (size_t)64, EL(mu0, ci), P.opts.dual_max);
  }
  EL(TILE_PTR(a.knotbuf, N), k) = knot_cost<M>(P, k, x, u, lam0, mn0, true);
}

template <class M>
__global__ void __launch_bounds__(64) k_outer_finish(KArgs a) {
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B || a.oflag[b] != 2) return;
  const to_solver_opts& o = P.opts;
  const int N = P.N;
  const double* jb = TILE_PTR(a.knotbuf, N);
  double J = 0.0;
  for (int k = 0; k < N; ++k) J += EL(jb, k);
  a.J[b] = J;
  double* mu0 = TILE_PTR(a.mu, P.n_cons);
  const double* mn0 = TILE_PTR(a.mu_next, P.n_cons);
  for (int ci = 0; ci < P.n_cons; ++ci) EL(mu0, ci) = EL(mn0, ci);
  a.rho[b] = o.bp_reg_initial; a.drho[b] = 0.0;
  a.dJzero[b] = 0; a.it_inner[b] = 0;
  const int rem = o.iterations_total - a.iterations[b];
  a.budget[b] = rem < o.iterations ? rem : o.iterations;
  a.status[b] = TO_UNSOLVED;
  a.oflag[b] = 0;
  atomicAdd(&a.counter[a.step], 1);
}

// Accepting a step = copying the accepted candidate's slot onto slot 0, so the nominal trajectory of every lane sits in
// ONE slot and every kernel reads it with full-line coalesced loads (a per-trajectory slot index turned each nominal
// load of a wave into up to 64 separate lines: measured 2x on the quadrotor forward pass).  Runs once per forward pass:
// later rounds only store into the slots of lanes that are still searching.  grid (tiles, 1, chunks): wave (tile, z)
// copies chunk z; every lane reads ITS accepted slot (a gather: as many lines as a per-slot pass would touch) and the
// stores to slot 0 are whole 512-byte rows.
__global__ void __launch_bounds__(64) k_accept(KArgs a) {
  TILE_LANE();
  const DevProblem& P = a.P;
  const int s = b < P.B ? a.acc[b] : 0;
  if (__ballot(s != 0) == 0) return;
  const int Lx = P.N * P.n, Lu = (P.N - 1) * P.m;
  const int per = (Lx + Lu + gridDim.z - 1) / gridDim.z;
  const int e0 = blockIdx.z * per, e1 = min(Lx + Lu, e0 + per);
  if (s == 0) return;
  const double* sx = TILE_PTR(XSLOT(a, s), Lx);
  double* dx = TILE_PTR(XSLOT(a, 0), Lx);
  const double* su = TILE_PTR(USLOT(a, s), Lu);
  double* du = TILE_PTR(USLOT(a, 0), Lu);
#pragma unroll 16
  for (int e = e0; e < min(e1, Lx); ++e) EL(dx, e) = EL(sx, e);
#pragma unroll 16
  for (int e = max(e0, Lx) - Lx; e < e1 - Lx; ++e) EL(du, e) = EL(su, e);
}

__global__ void k_clear_acc(KArgs a) {
  TILE_LANE();
  if (b < a.P.Bp) a.acc[b] = 0;
}

// start of a solve: reset the per-trajectory solver state.  J must already hold the (AL) cost of the rollout.
__global__ void k_solve_init(KArgs a, int reset_duals) {
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.Bp) return;
  const bool live = b < P.B;
  a.rho[b] = P.opts.bp_reg_initial; a.drho[b] = 0.0;
  a.dJzero[b] = 0; a.it_inner[b] = 0; a.iterations[b] = 0; a.outer[b] = 0;
  a.status[b] = TO_UNSOLVED; a.ls_index[b] = -1; a.bpfail[b] = 0; a.ls_round[b] = -1; a.acc[b] = 0; a.oflag[b] = 0;
  a.dJ[b] = 0.0; a.grad[b] = 0.0; a.cmax[b] = 0.0;
  const int tot = a.al_mode ? P.opts.iterations_total : P.opts.iterations;
  a.budget[b] = tot < P.opts.iterations ? tot : P.opts.iterations;
  a.active[b] = (live && a.budget[b] > 0) ? 1 : 0;
  if (live && a.budget[b] <= 0) a.status[b] = TO_MAX_ITERATIONS;
  if (reset_duals) {
    double* lam0 = TILE_PTR(a.lam, P.n_duals);
    double* mu0 = TILE_PTR(a.mu, P.n_cons);
    for (long long r = 0; r < P.n_duals; ++r) EL(lam0, r) = 0.0;
    for (int ci = 0; ci < P.n_cons; ++ci) EL(mu0, ci) = P.opts.penalty_initial;
  }
}

__global__ void k_set_active(KArgs a, int value, int clear_bpfail) {
  TILE_LANE();
  if (b >= a.P.Bp) return;
  a.active[b] = (b < a.P.B) ? value : 0;
  if (clear_bpfail == 1) a.bpfail[b] = 0;
  if (clear_bpfail) a.ls_round[b] = 0;  // 2: re-arm the line search only
}

__global__ void k_penalty_max(KArgs a, double* out) {
  TILE_LANE();
  if (b >= a.P.B) return;
  const double* mu0 = TILE_PTR(a.mu, a.P.n_cons);
  double mx = 0.0;
  for (int ci = 0; ci < a.P.n_cons; ++ci) mx = fmax(mx, EL(mu0, ci));
  out[b] = mx;
}

// ------------------------------------------------------------------------------------------------ layout transposes
// host layout h[e + L*b] (e = i + dim*k, column-major (dim, K, B))  <->  tiled device layout, L elements per trajectory.
// grid: (tiles, L).  Host-side access is strided, device-side coalesced; these are API-boundary copies, not hot.
// (cnt host elements per trajectory map to device elements e0 .. e0+cnt-1 of an array with L per trajectory)
__global__ void k_to_device(const double* __restrict__ h, double* __restrict__ d, int L, int e0, int cnt, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  EL(TILE_PTR(d, L), e0 + e) = h[(size_t)e + (size_t)cnt * b];
}
__global__ void k_to_host(const double* __restrict__ d, double* __restrict__ h, int L, int e0, int cnt, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  h[(size_t)e + (size_t)cnt * b] = EL(TILE_PTR(d, L), e0 + e);
}
__global__ void k_fill_uniform(double* d, const double* u, int dim, int L, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  EL(TILE_PTR(d, L), e) = u[e % dim];
}
// matrices: device blocks are row-major [k][r][c]; host wants column-major h[r + R*(c + Cc*(k + K*b))].  grid (tiles, K*R*Cc)
__global__ void k_mat_to_host(const double* __restrict__ d, double* __restrict__ h, int R, int Cc, int K, int B) {
  TILE_LANE();
  const int e = blockIdx.y;  // (k*R + r)*Cc + c
  if (b >= B) return;
  const int c = e % Cc, r = (e / Cc) % R, k = e / (Cc * R);
  h[(size_t)r + (size_t)R * (c + (size_t)Cc * (k + (size_t)K * b))] = EL(TILE_PTR(d, R * Cc * K), e);
}

// column-layout arrays -> host column-major blocks: h[r + Rr*(c + Cc*(k + K*b))] = col_array(b, column c0+c, entry k*rows_per_knot + r0 + r)
// grid (ceil(B/64), K*Rr*Cc), one thread per trajectory.
__global__ void k_col_to_host(const double* __restrict__ src, double* __restrict__ h, int E, int rows_per_knot, int r0, int Rr, int c0,
                              int Cc, int K, int B, int R, int G) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;  // (k*Cc + c)*Rr + r
  if (b >= B) return;
  const int r = e % Rr, c = (e / Rr) % Cc, k = e / (Rr * Cc);
  const int gtile = b / G, g = b % G;
  const size_t idx = ((size_t)gtile * E + (size_t)k * rows_per_knot + r0 + r) * 64 + g * R + (c0 + c);
  h[(size_t)r + (size_t)Rr * (c + (size_t)Cc * (k + (size_t)K * b))] = src[idx];
}

// ------------------------------------------------------------------------------------------------ per-knot API kernels
// RD.gradient!/RD.hessian! of the objective on the full state (no AL, no error-state projection):
// grad[(n+m), N, B], hess[(n+m),(n+m),N,B] column-major host layout, written directly.
template <class M>
__global__ void __launch_bounds__(64) k_cost_derivs(KArgs a, double* grad, double* hess) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int N = P.N, k = blockIdx.y;
  const bool terminal = (k == N - 1);
  constexpr int c = 0;  // nominal slot
  const double* X = TILE_PTR(XSLOT(a, c), N * n);
  const double* U = TILE_PTR(USLOT(a, c), (N - 1) * m);
  double x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = terminal ? 0.0 : EL(U, k * m + i);
  const size_t kb = (size_t)k + (size_t)N * b;
  for (int j = 0; j < nz; ++j) {
    double v[nz], g[nz], y[nz];
#pragma unroll
    for (int i = 0; i < nz; ++i) v[i] = (i == j) ? 1.0 : 0.0;
    cost_grad_hvp<n, m>(P.costs[P.cost_index[k]], x, u, terminal, v, g, y);
    const double sc = (P.opts.cost_dt_scaling && !terminal) ? P.dt[k] : 1.0;
#pragma unroll
    for (int i = 0; i < nz; ++i) {
      if (hess) hess[(size_t)i + nz * ((size_t)j + nz * kb)] = sc * y[i];
      if (grad && j == 0) grad[(size_t)i + nz * kb] = sc * g[i];
    }
  }
}

// RD.jacobian! of the discretised dynamics on the full state: F[n, n+m, N-1, B]
template <class M>
__global__ void __launch_bounds__(64) k_discrete_jacobian(KArgs a, double* F) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int N = P.N, k = blockIdx.y, j = blockIdx.z;
  constexpr int c = 0;  // nominal slot
  const double* X = TILE_PTR(XSLOT(a, c), N * n);
  const double* U = TILE_PTR(USLOT(a, c), (N - 1) * m);
  Dual xd[n], ud[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) xd[i] = Dual(EL(X, k * n + i), (i == j) ? 1.0 : 0.0);
#pragma unroll
  for (int i = 0; i < m; ++i) ud[i] = Dual(EL(U, k * m + i), (n + i == j) ? 1.0 : 0.0);
  rk_step<M, Dual>(P.mp, P.integrator, xd, ud, P.dt[k], xn);
  const size_t kb = (size_t)k + (size_t)(N - 1) * b;
#pragma unroll
  for (int i = 0; i < n; ++i) F[(size_t)i + n * ((size_t)j + nz * kb)] = xn[i].d;
}

// evaluate_constraints! / constraint_jacobians! for one constraint over its knot range, host layout output
template <class M>
__global__ void __launch_bounds__(64) k_constraint_eval(KArgs a, int ci, double* vals, double* jac) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  ConC& K = P.cons[ci];
  const int N = P.N, kk = blockIdx.y, k = K.k1 + kk, nk = K.k2 - K.k1 + 1;
  constexpr int c = 0;  // nominal slot
  const double* X = TILE_PTR(XSLOT(a, c), N * n);
  const double* U = TILE_PTR(USLOT(a, c), (N - 1) * m);
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
  const size_t kb = (size_t)kk + (size_t)nk * b;
  const int p = K.p, w = K.width;
  double coef[nz];
  for (int r = 0; r < p; ++r) {
    double cval;
    if (K.selector) cval = sel_row<nz>(K, z, r); else cval = con_row<nz>(K, z, r, coef);
    if (vals) vals[(size_t)r + p * kb] = cval;
    if (jac) {
      for (int col = 0; col < w; ++col) {
        double g = 0.0;
        if (K.selector) g = (K.sidx[r] == col) ? K.ssgn[r] : 0.0;
        else {
#pragma unroll
          for (int t = 0; t < nz; ++t) if (t < K.d.n_inds && K.d.inds[t] - 1 == col) g = coef[t];
        }
        jac[(size_t)r + p * ((size_t)col + (size_t)w * kb)] = g;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ cones (src/cones.jl), stateless
// x[dim,count] column-major; one thread per vector.
__global__ void k_cone_projection(int cone, int dim, long long count, const double* x, double* px, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  double* o = px + t * dim;
  int st = 0;
  if (cone == TO_CONE_IDENTITY) { for (int i = 0; i < dim; ++i) o[i] = v[i]; }
  else if (cone == TO_CONE_ZERO) { for (int i = 0; i < dim; ++i) o[i] = 0.0; }
  else if (cone == TO_CONE_NEGATIVE_ORTHANT) { for (int i = 0; i < dim; ++i) o[i] = fmin(0.0, v[i]); }
  else if (cone == TO_CONE_POSITIVE_ORTHANT) { for (int i = 0; i < dim; ++i) o[i] = fmax(0.0, v[i]); }
  else {
    const double s = v[dim - 1];
    double a2 = 0.0;
    for (int i = 0; i < dim - 1; ++i) a2 += v[i] * v[i];
    const double a = sqrt(a2);
    if (a <= -s) { for (int i = 0; i < dim; ++i) o[i] = 0.0; st = 0; }
    else if (a <= s) { for (int i = 0; i < dim; ++i) o[i] = v[i]; st = 1; }
    else if (a >= fabs(s)) { const double c = 0.5 * (1 + s / a); for (int i = 0; i < dim - 1; ++i) o[i] = v[i] * c; o[dim - 1] = a * c; st = 2; }
    else st = -1;  // NaN input: src/cones.jl:124 throws
  }
  if (status) status[t] = st;
}

__global__ void k_cone_jacobian(int cone, int dim, long long count, const double* x, double* jac, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  double* J = jac + t * dim * dim;  // column-major: J[r + dim*c]
  for (int i = 0; i < dim * dim; ++i) J[i] = 0.0;
  int st = 0;
  if (cone == TO_CONE_IDENTITY) { for (int i = 0; i < dim; ++i) J[i + dim * i] = 1.0; }
  else if (cone == TO_CONE_NEGATIVE_ORTHANT) { for (int i = 0; i < dim; ++i) J[i + dim * i] = v[i] <= 0 ? 1.0 : 0.0; }
  else if (cone == TO_CONE_POSITIVE_ORTHANT) { for (int i = 0; i < dim; ++i) J[i + dim * i] = v[i] >= 0 ? 1.0 : 0.0; }
  else if (cone == TO_CONE_SECOND_ORDER) {
    const int nn = dim;
    const double s = v[nn - 1];
    double a2 = 0.0;
    for (int i = 0; i < nn - 1; ++i) a2 += v[i] * v[i];
    const double a = sqrt(a2);
    if (a <= -s) st = 0;
    else if (a <= s) { for (int i = 0; i < nn; ++i) J[i + nn * i] = 1.0; st = 1; }
    else if (a >= fabs(s)) {
      const double c = 0.5 * (1 + s / a);
      for (int i = 0; i < nn - 1; ++i)
        for (int j = 0; j < nn - 1; ++j) J[i + nn * j] = -0.5 * s / (a * a * a) * v[i] * v[j] + ((i == j) ? c : 0.0);
      for (int i = 0; i < nn - 1; ++i) J[i + nn * (nn - 1)] = 0.5 * v[i] / a;
      for (int i = 0; i < nn - 1; ++i) J[(nn - 1) + nn * i] = ((-0.5 * s / (a * a)) + c / a) * v[i];
      J[(nn - 1) + nn * (nn - 1)] = 0.5;
      st = 2;
    } else st = -1;
  }
  if (status) status[t] = st;
}

__global__ void k_cone_hessian(int cone, int dim, long long count, const double* x, const double* bvec, double* hess, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  const double* bb = bvec + t * dim;
  double* H = hess + t * dim * dim;
  for (int i = 0; i < dim * dim; ++i) H[i] = 0.0;
  int st = 0;
  if (cone == TO_CONE_SECOND_ORDER) {
    const int nn = dim - 1;
    const double s = v[nn], bs = bb[nn];
    double a2 = 0.0, vbv = 0.0;
    for (int i = 0; i < nn; ++i) { a2 += v[i] * v[i]; vbv += v[i] * bb[i]; }
    const double a = sqrt(a2);
    if (a <= -s) st = 0;
    else if (a <= s) st = 1;
    else if (a > fabs(s)) {
      for (int i = 0; i < nn; ++i) {
        double hi = 0.0;
        for (int j = 0; j < nn; ++j) hi += (-v[i] * v[j] / (a * a) + ((i == j) ? 1.0 : 0.0)) * bb[j];
        H[i + dim * nn] = hi / (2 * a);
        H[nn + dim * i] = H[i + dim * nn];
        for (int j = 0; j <= i; ++j) {
          const double vij = v[i] * v[j];
          const double H1 = hi * v[j] * (-s / (a * a * a));
          double H2 = vij * (2 * vbv) / (a * a * a * a) - v[i] * bb[j] / (a * a);
          double H3 = -vij / (a * a);
          if (i == j) { H2 -= vbv / (a * a); H3 += 1; }
          H2 *= s / a;
          H3 *= bs / a;
          H[i + dim * j] = (H1 + H2 + H3) / 2;
          H[j + dim * i] = H[i + dim * j];
        }
      }
      st = 2;
    } else st = -1;
  }
  if (status) status[t] = st;
}

}  // namespace to
