// common.h — kernel argument block, layout macros and per-knot helpers shared by the kernels of the batched
// iLQR / AL hot path for gfx950 (MI355X).  See DESIGN.md §3 for the layouts.
//
// Tiled arrays (X, U, duals): array-of-structures-of-arrays with a 64-trajectory tile.  Element e (e.g. e = k*n + i) of
// trajectory b of an array with L elements per trajectory lives at
//     base[((b/64)*L + e)*64 + (b%64)]
// so the lanes of a wave that hold neighbouring trajectories read/write contiguous bytes and what a wave touches as it
// walks the knots is ONE contiguous stream.  Gains (K, d) and the expansion blocks are TRAJECTORY-major (below).
#pragma once
#include <hip/hip_runtime.h>

#include "models.h"
#include "problem_dev.h"

namespace to {

struct KArgs {
  DevProblem P;
  double* Xs;     // nominal trajectory, tiled, L = N*n        ("slot 0")
  double* Us;     // nominal controls, tiled, L = (N-1)*m
  double *Xc, *Uc;  // line-search candidates ("slot q+1" = candidate q), FORWARD-WAVE-major: the forward wave w = b / TW owns the
                    // contiguous block [w][e][64] and its hardware lane q*TW + (b % TW) holds candidate q of trajectory b, so
                    // every candidate store of a wave is one 512-byte row of one stream (slot-major candidates made each
                    // store touch CW streams 85 MB apart: 4x slower forward pass, TLB- and partial-line-bound)
  int dump_wave;  // index of the spare block behind the last wave's in Xc / Uc: where lanes without a candidate store
  int repack_block0;  // first of the second set of candidate blocks (wave w: repack_block0 + w), used by a repacked last line-search
                      // round (k_forward.h LsRound); 0: no repacking
  int CW, TW;     // a forward wave holds CW line-search candidates x TW trajectories, CW*TW <= 64 (hardware lane q*TW + t); candidate
                  // "slot" q+1 of trajectory b lives in the block of wave b / TW (below).  Two shapes are in use: the base one (CW a
                  // power of two, TW = 64/CW) and, once the active trajectories fit the chip that way, CW = the whole search depth
  double* x0;     // L = n
  int* acc;       // [Bp] slot of the candidate accepted in this forward pass (0 = none): copied onto slot 0 by k_accept, or
                  //      written through by the next k_expand (M::accept_write_through)
  int* accp;      // [Bp] where that candidate sits: forward wave * 64 + hardware lane (slot_ptr)
  // Active-list compaction (fused lane path of the small models at large batches, MFMA path of the Quadrotor): after batch step s
  // k_compact lists the trajectories that go on, IN INDEX ORDER, in alist[(s+1)&1]; the kernels of step s+1 take their
  // trajectories from that list, so the waves stay full while the batch drains (tile-granular skipping left most lanes of
  // most waves idle: C3 runs 141 batch steps for a mean of 52 iterations).  Which lane works on which trajectory changes
  // nothing in its arithmetic.  Index order keeps neighbouring lanes on neighbouring trajectories: the nominal arrays are
  // batch-fastest, and an arbitrary order (atomic appends) turned every nominal load into one 64-byte sector per lane (C5:
  // forward fetch traffic x2.5).
  int coop_merge; // fused cooperative pass: symmetrise the cost-to-go Hessian where it is read (coop_knot; TRAJOPT_COOP_MERGE)
  int compact;    // 1: on
  int* alist;     // [2][Bp]
  int* acount;    // [2]
  int* ccount;    // [256] per-workgroup counts of the two-launch compaction (large batches)
  double *Mc, *Hc, *gc;               // column layout (see "column layout" below): [Ā B̄], Q-function cost blocks, gradient
  double* Kt;                         // gains, trajectory-major rows: Kt[(b*(N-1) + k)*RSK + r*(ne+1) + i] = K_k[r][i], i = ne: d_k[r]
  double *Mt, *Ht, *gt;               // tangent-matrix layout of the expansion for the MFMA backward pass (k_backward.h)
  int bwd_mfma;                       // 1: expansion writes Mt/Ht/gt and the MFMA backward pass runs; 0: column layout + cooperative pass
  int bwd_lane;                       // 1: expansion writes the lane layout into Mc/Hc/gc and the one-lane-per-trajectory backward pass runs
  int h_compact;                      // Ht holds one row per knot (block-diagonal Qxx, Quu, no Qux): see k_expand.h
  int h_diag;                         // column layout: Hc holds ONE row per knot, the diagonal entry of the lane's own column (diagonal
                                      // costs, goal / bound constraints only: every off-diagonal entry of the cost block is an exact zero)
  double *lam, *mu;                   // L = n_duals, n_cons
  double *J, *dJ, *grad, *rho, *drho, *dV, *cmax, *Jout;  // [Bp] plain (dV: [2][Bp])
  int *status, *iterations, *it_inner, *outer, *dJzero, *ls_index, *active, *budget, *bpfail;
  int* counter;   // [steps] number of trajectories still active after each batch step
  int *oflag, *ost;   // [Bp] AL outer update pending (1: evaluate, 2: update duals) and the inner solve's status
  int *olist, *ocount;  // trajectories whose inner solve ended in batch step s: olist[(s&1)*Bp + 0 .. ocount[s&1]) (appended by k_forward,
                        // consumed by the k_outer_* kernels of the same step, which also clear the other parity's count)
  double* knotbuf;    // tiled, L = N: per-knot scratch of the outer update (violations, then AL cost terms)
  double* mu_next;    // tiled, L = n_cons: penalties after the pending outer update
  int* it_pn;         // [Bp] projection solves of the projected-Newton polish (k_pn.h)
  double* pn_cmax;    // [Bp] violation the polish ended with (dynamics defects included)
  int al_mode;    // 0: iLQR, 1: AL-iLQR
  int control;    // 1: run the solver state machine at the end of the forward pass; 0: phase API
  int step;
  // Two-launch line search (large dense batches of the small models): the forward pass as launch A — ONE round of CW_A step sizes for
  // every active trajectory; whoever accepted one (or has nothing to search) finishes there, the others are flagged — a compaction of
  // the flags, and launch B over the flagged trajectories only, which continues at step size ls_c0 with rounds of its own width.  Which
  // launch evaluates a candidate changes nothing in its value, and the first accepted step size is taken as before: bit-identical.
  int ls_phase;   // 0: one launch, every round in-kernel; 1: launch A; 2: launch B
  int ls_c0;      // launch B: index of its first step size (= CW of launch A)
  int blk0;       // first candidate block of this launch (launch B's blocks sit behind launch A's)
  int* pending;   // [Bp] launch A: this trajectory's search goes on in launch B
  int* plist;     // [Bp] the flagged trajectories in index order ...
  int* pcount;    // [1]  ... and their number (k_flags_count / k_flags_write)
  int store_x;    // 1: the forward pass stores every candidate's states (k_accept copies the accepted ones); 0: only their controls —
                  // k_accept_roll (k_forward.h) then re-rolls the accepted candidates.  Set per batch step by the solve loop.
};

// gains row of one knot of one trajectory: m rows of (ne gains + 1 feed-forward) doubles
template <class M> struct Gains { static constexpr int RSK = M::m * (M::ne + 1); };

// per-lane pointer to element 0 of this lane's trajectory in a tiled array with L elements per trajectory;
// element e is then p[e*64] (e wave-uniform -> scalar address arithmetic, immediate offsets for small constants)
#define TILE_PTR(base, L) ((base) + ((size_t)tile * (size_t)(L)) * 64 + lane)
#define EL(p, e) (p)[(size_t)(e) * 64]
#define TILE_LANE() const int tile = blockIdx.x, lane = threadIdx.x, b = tile * 64 + lane
// pointer to element 0 of trajectory b in slot sl of a trajectory array with L elements (sl = 0: nominal, otherwise the line-search
// candidate the last forward pass ACCEPTED for b); element e is p[e*64] in both layouts.  Where the accepted candidate sits —
// forward wave w, hardware lane l — is recorded by the forward pass itself as accp[b] = w*64 + l, so that readers need not
// know the wave shape (CW x TW) or, with active-list compaction, which wave happened to process the trajectory.
__device__ __forceinline__ double* slot_ptr(double* nominal, double* cand, const int* accp, int b, int sl, int L) {
  if (sl == 0) return nominal + ((size_t)(b >> 6) * (size_t)L) * 64 + (b & 63);
  const int pos = accp[b];
  return cand + ((size_t)(pos >> 6) * (size_t)L) * 64 + (pos & 63);
}
#define X_SLOT_PTR(a, b, sl) slot_ptr((a).Xs, (a).Xc, (a).accp, b, sl, (a).P.N * (a).P.n)
#define U_SLOT_PTR(a, b, sl) slot_ptr((a).Us, (a).Uc, (a).accp, b, sl, ((a).P.N - 1) * (a).P.m)

// objective (+AL) value of one knot.  u must be zeros at the terminal knot (the reference evaluates the
// terminal cost with the knot's zero control; src/cost_functions.jl:92-94, test/objective_tests.jl:129).
// lam0 / mu0: this lane's pointers to dual row 0 / penalty 0 (tiled arrays).
// GEN = false: no dense QuadraticCost and no non-selector constraint in the tables (those branches are compiled out)
// gl0: this lane's pointer into DevProblem::gl (per-trajectory linear cost terms; nullptr: none)
template <class M, bool GEN = true>
__device__ __forceinline__ double knot_cost(const DevProblem& P, int k, const double* x, const double* u, const double* lam0,
                                            const double* mu0, bool with_al, const double* gl0 = nullptr, const double* cp0 = nullptr) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const int cidx = P.cost_index[k];
  double Jk = cost_eval<n, m, GEN>(P.costs[cidx], x, u);
  if constexpr (GEN) {
    if (P.gl) Jk += goal_lin_cost<n, m>(gl0, cidx, x, u);  // (P.gl: wave-uniform; gl0 is only meaningful with it)
  }
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
      if constexpr (GEN) {  // (per-trajectory constraint parameters: general variants only)
        double zc[nz];
#pragma unroll
        for (int i = 0; i < nz; ++i) zc[i] = z[i];
        con_shift<nz>(P, K, cp0, zc);
        Ja += al_term<n, m, GEN>(K, zc, lam, (size_t)64, EL(mu0, ci));
      } else
      Ja += al_term<n, m, GEN>(K, z, lam, (size_t)64, EL(mu0, ci));
    }
    Jk += Ja;
  }
  return Jk;
}

// AL penalty terms of one knot only
// AL terms of one stage knot with up to two register-cached control-block constraints (ConStage); the others take the
// descriptor-table path.  Terms are summed in constraint order, like knot_al.
template <class M, bool GEN, class CS>
__device__ __forceinline__ double knot_al_cached(const DevProblem& P, int k, const double* x, const double* u, const double* lam0,
                                                 const double* mu0, int ncs, const CS& c0, const CS& c1, const double* cp0 = nullptr) {
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
    if constexpr (GEN) con_shift<nz>(P, K, cp0, z);  // (z is rebuilt for every table constraint)
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
__device__ __forceinline__ double knot_violation(const DevProblem& P, int k, const double* x, const double* u, const double* cp0 = nullptr) {
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
    double zc[nz];
#pragma unroll
    for (int i = 0; i < nz; ++i) zc[i] = z[i];
    con_shift<nz>(P, K, cp0, zc);
    const double v = con_violation<nz>(K, zc);
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
  const double* X = X_SLOT_PTR(a, tile * 64 + lane, c);
  const double* U = U_SLOT_PTR(a, tile * 64 + lane, c);
  double* lam0 = TILE_PTR(a.lam, P.n_duals);
  double* mu0 = TILE_PTR(a.mu, P.n_cons);
  const double* cp0 = TILE_PTR(P.cp, P.n_cp);
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
        double zc[nz];
#pragma unroll
        for (int i = 0; i < nz; ++i) zc[i] = z[i];
        con_shift<nz>(P, K, cp0, zc);
        con_dual_update<nz>(K, zc, lam, (size_t)64, EL(mu0, ci), P.opts.dual_max);
      }
    }
    if (cmax_out && P.n_cons > 0) { const double v = knot_violation<M>(P, k, x, u, cp0); if (!(v <= cmax)) cmax = v; }
    if (J_out) J += knot_cost<M>(P, k, x, u, lam0, mu0, with_al, TILE_PTR(P.gl, P.n_costs * nz), cp0);
  }
  if (J_out) *J_out = J;
  if (cmax_out) *cmax_out = cmax;
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

// ------------------------------------------------------------------------------------------------ regularisation
// !(x > 0) on the bit pattern: true for zero, negatives and NaN.  The positive-definiteness tests of the backward passes use it so
// that they keep their meaning in the translation units compiled with -fno-honor-nans (ops_lane.h), where the compiler may turn
// !(x > 0) into x <= 0 and a NaN pivot would pass.
__device__ __forceinline__ bool not_positive(double x) {
  const long long b = __double_as_longlong(x);
  return !(b > 0 && b <= 0x7ff0000000000000LL);
}
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

}  // namespace to
