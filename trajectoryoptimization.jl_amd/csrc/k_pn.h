// k_pn.h — projected-Newton polish (Altro.jl ProjectedNewtonSolver, the last stage of ALTRO): ONE WAVE per trajectory.
//
// What it computes is defined by oracle/oracle_pn.h (Altro is out of tree): Newton steps onto the active
// constraints — initial condition, dynamics defects f(x_k, u_k) (-) x_{k+1}, equality rows, inequality rows within
// active_set_tolerance_pn, second-order cones through the row |v| - s — in the metric of the diagonal objective Hessian,
//   dZ = -W D'(D W D' + rho I)^-1 d,  W = 1 / (max(diag H, 0) + rho_primal),
// in error-state coordinates, with Altro's reg_solve refinement, backtracking on |d|_inf and convergence-rate loops.
//
// Mapping to the hardware.  Rows grouped by knot — group k = [defect arriving at knot k (ne rows); active constraint rows of
// knot k] — make S = D W D' BLOCK-TRIDIAGONAL with blocks of nb_k = ne + pa_k <= 12 + 9 rows (C5), so its Cholesky factor is one
// sweep over the horizon: L_k,k-1 = S_k,k-1 L_k-1^-T (only the ne defect rows couple backwards), L_k = chol(S_kk + rho I -
// L_k,k-1 L_k,k-1').  That sweep — like the Riccati recursion — is sequential in k and far too small per knot for more than
// one wave, while the batch supplies the parallelism: a workgroup of one wave owns one trajectory, keeps the two live diagonal
// blocks, the coupling block and the knot's Jacobian rows in LDS (14 KB for C5), and its 64 lanes own ENTRIES of those blocks.
// Residuals, the step and the line search in between are parallel over the knots or over (knot, column) pairs of the same
// wave.  Per-trajectory state lives in a trajectory-major workspace (a header and one contiguous record per knot: the sweeps
// stream it in order), sized on the host from the constraint list.  One projection round of the batch is three launches:
//   k_pn_begin    (wave per trajectory)  fresh active set + violation; done / budget spent -> write the trajectory back, status
//   k_pn_lin_col / k_pn_lin_knot  (wave per 64 (knot, column) pairs / knots)  Jacobian columns of the dynamics by forward-mode dual numbers through the
//                 RK stages, active constraint rows, the metric — the register-hungry part (Quadrotor duals), fully parallel
//   k_pn_project  (wave per trajectory)  factorisation, reg_solve refinement, line search, convergence-rate loop
// and the host enqueues n_steps + 2 rounds back to back (a trajectory that is done costs its waves one header load).
//
// The kernel is written as a sequence of PHASES: `PN_FOR(i, n) { ... }` runs its body for i = 0..n-1 with the iterations
// spread over the lanes, `PN_SYNC()` separates phases; no iteration of a phase reads what another iteration of the same phase
// writes, and everything a later phase needs goes through LDS or the workspace.  Values computed outside PN_FOR are
// wave-uniform (every lane computes them from the same memory).  With TO_PN_HOST defined the same source compiles with g++ —
// PN_FOR becomes a plain loop — and the CPU test-suite runs THIS code against the oracle (tests/test_pn_host.py) before a GPU
// is involved; on the GPU the -m gpu tests compare it with the oracle again.
#pragma once
#include "common.h"

namespace to {

constexpr int PN_MAX_ROWS = 64;      // candidate constraint rows of one knot (one bit each in the active mask)
constexpr int PN_NB_LIMIT = 44;      // largest block: ne + active rows of a knot (LDS: 2 NB^2 + ... doubles)
constexpr int PN_REFINEMENTS = 10;   // Altro _projection_solve!: max_refinements
constexpr int PN_LS_TRIALS = 10;     // Altro _projection_linesearch!
constexpr int PN_REG_SOLVE_ITERS = 25;
constexpr double PN_REG_SOLVE_TOL = 1e-8;

struct PnArgs {
  KArgs a;
  const int* pak;         // [N] candidate rows of knot k (upper bound of its active rows; host table)
  const long long* koff;  // [N+1] offset of knot k's record in a trajectory's workspace, koff[N] = doubles per trajectory
  double* ws;             // workspace of the trajectories of this launch
  const int* list;        // trajectories to polish
  int base;               // this launch polishes list[base + workgroup]
  int nbmax;              // max_k (ne + pak[k])
  int* it_pn;             // [Bp] projection solves
  double* cmax_out;       // [Bp] violation at the end (defects included)
};

#ifdef TO_PN_HOST
#define PN_FOR(i, n) for (int i = 0; i < (n); ++i)
#define PN_SYNC() do { } while (0)
#define PN_SYNC_LDS() do { } while (0)
#define PN_FN inline
#define PN_HD inline
#else
// The trip count depends on n only and EVERY lane runs every pass: a lane beyond the end repeats the last item (same loads, same
// arithmetic, the same values stored to the same addresses, read-modify-writes included: the lanes of a wave read before any of
// them writes).  A wave whose EXEC mask stays full has no divergent regions for the register allocator to place spills around
// (hipcc may put a VGPR->AGPR spill ahead of the exec restore of a join block), and FP64 instructions issue faster.
#define PN_FOR(i, n)                                                             \
  for (int i##_b = 0, i##_n = (n); i##_b < i##_n; i##_b += 64)                   \
    for (int i = (i##_b + lane < i##_n ? i##_b + lane : i##_n - 1), i##_1 = 1; i##_1; i##_1 = 0)
#define PN_SYNC() __syncthreads()   // the workgroup IS the wave: LDS and workspace writes of a phase become visible to the next
// ... where the next phase only needs the LDS writes (the horizon sweeps: their workspace stores are read again after the sweep)
#define PN_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PN_FN __device__ inline
#define PN_HD __host__ __device__ inline
#endif

PN_FN void pn_upd_max(double& m, double v) { if (v > m || v != v) m = v; }  // NaN sticks (v > NaN is false)
PN_FN int pn_popc(unsigned long long m) { int c = 0; while (m) { m &= m - 1; ++c; } return c; }

// header of a trajectory's workspace
enum { PN_H_STATE = 0, PN_H_STEPS = 1, PN_H_VIOL = 2, PN_H_FAILED = 3, PN_H_TRAJ = 4 /* the trajectory this slot polishes */, PN_HEADER = 8 };
enum { PN_FRESH = 0, PN_ACTIVE = 1, PN_DONE = 2 };

// vectors of a knot record: d (rhs on the active set), dn (candidate's), dl (multiplier step), r (residual), t (scratch)
enum { PN_VD = 0, PN_VDN = 1, PN_VDL = 2, PN_VR = 3, PN_VT = 4, PN_NVEC = 5 };

template <class M>
struct PnRec {
  double *Z, *Zb, *dZ, *W, *F, *C, *vec, *Mi, *Nf, *Pb, *loc;
  unsigned long long* mask;
  int nbm;  // ne + pak[k]: stride of the record's vectors
  PN_FN double* v(int which) const { return vec + which * nbm; }
};
// doubles of knot k's record (host: workspace sizing; must match pn_rec)
template <class M>
PN_HD long long pn_rec_size(int pa, int pa_prev, bool first) {
  constexpr int ne = M::ne, nc = M::ne + M::m, nz = M::n + M::m;
  const int nb = ne + pa, nbp = first ? 0 : ne + pa_prev;
  return 2 * nz + 2 * nc + ne * nc + (long long)pa * nc + PN_NVEC * nb + (long long)nb * nb + (long long)nb * nbp + (long long)nbp * ne + 4;
}
template <class M>
PN_FN PnRec<M> pn_rec(const PnArgs& q, double* w, int k) {
  constexpr int ne = M::ne, nc = M::ne + M::m, nz = M::n + M::m;
  const int pa = q.pak[k], nb = ne + pa, nbp = k > 0 ? ne + q.pak[k - 1] : 0;
  PnRec<M> r;
  double* p = w + q.koff[k];
  r.Z = p; p += nz; r.Zb = p; p += nz; r.dZ = p; p += nc; r.W = p; p += nc; r.F = p; p += ne * nc; r.C = p; p += pa * nc;
  r.vec = p; p += PN_NVEC * nb; r.Mi = p; p += nb * nb; r.Nf = p; p += nb * nbp; r.Pb = p; p += nbp * ne;
  r.mask = reinterpret_cast<unsigned long long*>(p); r.loc = p + 1;
  r.nbm = nb;
  return r;
}

// LDS of one wave (doubles): three diagonal blocks (inverse factors of the previous and the current knot, work), the coupling block, the knot's Jacobian rows and metric (and the rows times the metric), two vectors, a
// 64-entry reduction buffer
struct PnLds {
  double *LA, *LB, *LW, *Lo, *F, *Ca, *Cb, *FW, *CW, *Wa, *Wb, *va, *vb, *red;
  int NB;
};
template <class M>
PN_HD long long pn_lds_doubles(int NB) {
  constexpr int ne = M::ne, nc = M::ne + M::m, ncp = nc | 1;  // rows of F and C at an odd stride: lanes on different rows, different banks
  return 3LL * NB * NB + (long long)ne * NB + 2 * ne * ncp + 3LL * (NB - ne) * ncp + 2 * nc + 2 * NB + 64;
}
template <class M>
PN_FN PnLds pn_lds(double* p, int NB) {
  constexpr int ne = M::ne, nc = M::ne + M::m, ncp = nc | 1;
  PnLds l; l.NB = NB;
  l.LA = p; p += NB * NB; l.LB = p; p += NB * NB; l.LW = p; p += NB * NB; l.Lo = p; p += ne * NB; l.F = p; p += ne * ncp;
  l.Ca = p; p += (NB - ne) * ncp; l.Cb = p; p += (NB - ne) * ncp; l.FW = p; p += ne * ncp; l.CW = p; p += (NB - ne) * ncp;
  l.Wa = p; p += nc; l.Wb = p; p += nc;
  l.va = p; p += NB; l.vb = p; p += NB; l.red = p;
  return l;
}

// ---- candidate rows of a knot: f(q, value, gradient over z = [x; u] (nz), equality?) for every row that may take part.
// A second-order cone [v; s] contributes the ONE row |v| - s.
// lane pointer into DevProblem::cp (per-trajectory constraint parameters) of the trajectory whose workspace w is
PN_FN const double* pn_cp0(const DevProblem& P, const double* w) {
  if (!P.cp) return nullptr;
  const int b = (int)w[PN_H_TRAJ];
  return P.cp + ((size_t)(b >> 6) * (size_t)P.n_cp) * 64 + (b & 63);
}
template <class M, bool GRAD, class Fn>
PN_FN void pn_for_candidates(const DevProblem& P, int k, const double* zin, const double* cp0, Fn&& f) {
  constexpr int nz = M::n + M::m;
  int qi = 0;
  double gz[nz], coef[nz], z[nz];
  for (int ci = 0; ci < P.n_cons; ++ci) {
    ConC& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    for (int i = 0; i < nz; ++i) z[i] = zin[i];
    con_shift<nz>(P, K, cp0, z);  // the state as this constraint sees it (a shift: values change, gradients do not)
    const int p = K.p;
    if (K.d.sense == TO_CONE_SECOND_ORDER) {
      double a2 = 0.0;
      for (int r = 0; r < p - 1; ++r) { const double c = sel_row<nz>(K, z, r); a2 += c * c; }
      const double a = sqrt(a2), s = sel_row<nz>(K, z, p - 1);
      if (GRAD) {
        for (int j = 0; j < nz; ++j) gz[j] = 0.0;
        if (a > 0.0)
          for (int r = 0; r < p - 1; ++r) { const int j = K.sidx[r]; if (j >= 0) gz[j] += (sel_row<nz>(K, z, r) / a) * K.ssgn[r]; }
        const int js = K.sidx[p - 1];
        if (js >= 0) gz[js] -= K.ssgn[p - 1];
      }
      f(qi, a - s, gz, false);
      ++qi;
    } else if (K.selector) {
      for (int r = 0; r < p; ++r) {
        if (GRAD) { for (int j = 0; j < nz; ++j) gz[j] = 0.0; const int j = K.sidx[r]; if (j >= 0) gz[j] = K.ssgn[r]; }
        f(qi, sel_row<nz>(K, z, r), gz, K.d.sense == TO_CONE_ZERO);
        ++qi;
      }
    } else {
      for (int r = 0; r < p; ++r) {
        for (int j = 0; j < nz; ++j) coef[j] = 0.0;
        const double c = con_row<nz>(K, z, r, coef);
        if (GRAD) {
          for (int j = 0; j < nz; ++j) gz[j] = 0.0;
          for (int t = 0; t < K.d.n_inds && t < nz; ++t) gz[K.d.inds[t] - 1] += coef[t];
        }
        f(qi, c, gz, K.d.sense == TO_CONE_ZERO);
        ++qi;
      }
    }
  }
}

// gradient of a row in the coordinates the step moves in: [G(x)' g_x; g_u] (no control at the terminal knot)
template <class M>
PN_FN void pn_project_row(const double* x, const double* gz, bool terminal, double* ge) {
  constexpr int n = M::n, m = M::m, ne = M::ne;
  errstate_tmul<M>(x, gz, ge);
  for (int j = 0; j < m; ++j) ge[ne + j] = terminal ? 0.0 : gz[n + j];
}

// defect arriving at knot k (ne): the initial condition at k = 0, f(x_{k-1}, u_{k-1}) (-) x_k otherwise
template <class M>
PN_FN void pn_defect(const DevProblem& P, int k, const double* zprev, const double* z, const double* x0, double* e) {
  constexpr int n = M::n;
  // branch-free in k (the lanes of a wave sit on different knots): the step is always taken, the operands selected
  const int kp = k > 0 ? k - 1 : 0;
  double f[n], a[n], b[n];
  model_step<M, double>(P.mp, P.integrator, kp, zprev, zprev + n, P.dt[kp], f);
  for (int i = 0; i < n; ++i) { a[i] = (k == 0) ? z[i] : f[i]; b[i] = (k == 0) ? x0[i] : z[i]; }
  state_diff<M>(a, b, e);
}

#ifdef TO_PN_HOST
#define PN_LANE_PARAM
#define PN_LANE_ARG
#else
#define PN_LANE_PARAM , int lane
#define PN_LANE_ARG , lane
#endif

// -DTO_PN_TIMING (a diagnostic build of ops_pn.hip, never the product): the first workgroup of a k_pn_project launch prints where
// its trajectory's time went, in microseconds of the 100 MHz wall clock.
#if defined(TO_PN_TIMING) && !defined(TO_PN_HOST)
__device__ long long pn_tacc[8];
#define PN_TIC() const long long pn_tic_ = wall_clock64()
#define PN_TOC(slot) do { if (blockIdx.x == 0 && lane == 0) pn_tacc[slot] += wall_clock64() - pn_tic_; } while (0)
#define PN_MARK_DECL() long long pn_mark_ = wall_clock64()
#define PN_MARK(slot) do { const long long t_ = wall_clock64(); if (blockIdx.x == 0 && lane == 0) pn_tacc[slot] += t_ - pn_mark_; pn_mark_ = t_; } while (0)
#else
#define PN_MARK_DECL() do { } while (0)
#define PN_MARK(slot) do { } while (0)
#define PN_TIC() do { } while (0)
#define PN_TOC(slot) do { } while (0)
#endif

// ---- staged sweeps.  A phase that reads knot k's blocks straight from the workspace pays one memory round trip per knot of a
// sweep that is nothing but a chain of such phases (~10 ms per trajectory and round on C5).  Blocks of up to 64 * PN_PF entries
// (nbmax <= 21: every BASELINE problem) instead travel  workspace -> registers -> LDS  one knot AHEAD: the loads of knot k+1 are
// issued before knot k is computed out of LDS, raw (record layout, sized by the host's candidate counts, so that nothing about the
// active set has to be known to issue them), and the sweep's own stores are never waited for inside the sweep (PN_SYNC_LDS).
// TO_PN_HOST runs the same index arithmetic with the "registers" being a plain array.
#ifndef TO_PN_PF
#define TO_PN_PF 7   // 0: no staging (diagnostic builds)
#endif
constexpr int PN_PF = TO_PN_PF, PN_PF_REGS = TO_PN_PF > 0 ? TO_PN_PF : 1;
#ifdef TO_PN_HOST
struct PnStage { double r[64 * PN_PF_REGS]; };
PN_FN void pn_fetch(PnStage& s, const double* src, int cnt) { for (int e = 0; e < cnt; ++e) s.r[e] = src[e]; }
PN_FN void pn_put(double* dst, const PnStage& s, int cnt) { for (int e = 0; e < cnt; ++e) dst[e] = s.r[e]; }
PN_FN void pn_put_rows(double* dst, const PnStage& s, int cnt, int nc, int ncp) { for (int e = 0; e < cnt; ++e) dst[(e / nc) * ncp + e % nc] = s.r[e]; }
PN_FN unsigned long long pn_uniform(unsigned long long v) { return v; }
#else
struct PnStage { double r[PN_PF_REGS]; };
PN_FN void pn_fetch(PnStage& s, const double* src, int cnt, int lane) {
#pragma unroll
  for (int j = 0; j < PN_PF; ++j) if (j * 64 < cnt) { const int e = j * 64 + lane; s.r[j] = src[e < cnt ? e : cnt - 1]; }
}
PN_FN void pn_put(double* dst, const PnStage& s, int cnt, int lane) {
#pragma unroll
  for (int j = 0; j < PN_PF; ++j) if (j * 64 < cnt) { const int e = j * 64 + lane; dst[e < cnt ? e : cnt - 1] = s.r[j]; }
}
PN_FN void pn_put_rows(double* dst, const PnStage& s, int cnt, int nc, int ncp, int lane) {  // rows of nc entries land at stride ncp
#pragma unroll
  for (int j = 0; j < PN_PF; ++j) if (j * 64 < cnt) { int e = j * 64 + lane; e = e < cnt ? e : cnt - 1; dst[(e / nc) * ncp + e % nc] = s.r[j]; }
}
PN_FN unsigned long long pn_uniform(unsigned long long v) {  // every lane loaded the same word: make the compiler know it
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
#endif

// max over the knots of loc[0] (NaN-aware): wave-uniform result
template <class M>
PN_FN double pn_reduce_max(const PnArgs& q, double* w, const PnLds& L PN_LANE_PARAM) {
  const int N = q.a.P.N;
  PN_FOR(j, 64) {
    double mx = 0.0;
    for (int k = j; k < N; k += 64) pn_upd_max(mx, pn_rec<M>(q, w, k).loc[0]);
    L.red[j] = mx;
  }
  PN_SYNC();
  double mx = 0.0;
  for (int j = 0; j < 64; ++j) pn_upd_max(mx, L.red[j]);
  PN_SYNC();
  return mx;
}

// d on the active set of the point Z (cand = false) or Zb (true), into d or dn.  refresh: choose the active set from the values
// first.  Returns |d|_inf.
template <class M>
PN_FN double pn_eval(const PnArgs& q, double* w, const PnLds& L, const double* x0, bool cand, bool refresh, int dst_vec PN_LANE_PARAM) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m, nc = ne + m;
  const DevProblem& P = q.a.P;
  const int N = P.N;
  const double tol_a = P.opts.active_set_tolerance_pn;
  PN_FOR(k, N) {
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const double* z = cand ? R.Zb : R.Z;
    const double* zp = z;
    if (k > 0) { const PnRec<M> Rp = pn_rec<M>(q, w, k - 1); zp = cand ? Rp.Zb : Rp.Z; }
    double* dst = R.v(dst_vec);
    double e[ne], mx = 0.0;
    pn_defect<M>(P, k, zp, z, x0, e);
    for (int i = 0; i < ne; ++i) { dst[i] = e[i]; pn_upd_max(mx, fabs(e[i])); }
    unsigned long long mask = refresh ? 0ull : *R.mask;
    int na = 0;
    const bool terminal = (k == N - 1);
    if (refresh) {
      pn_for_candidates<M, true>(P, k, z, pn_cp0(P, w), [&](int qi, double val, const double* gz, bool eq) {
        if (!(eq || val >= -tol_a)) return;
        double ge[nc], g2 = 0.0;
        pn_project_row<M>(z, gz, terminal, ge);
        for (int j = 0; j < nc; ++j) g2 += ge[j] * ge[j];
        if (!(g2 > 0.0)) return;
        mask |= 1ull << qi;
        dst[ne + na++] = val; pn_upd_max(mx, fabs(val));
      });
      *R.mask = mask;
    } else {
      pn_for_candidates<M, false>(P, k, z, pn_cp0(P, w), [&](int qi, double val, const double*, bool) {
        if (!(mask >> qi & 1ull)) return;
        dst[ne + na++] = val; pn_upd_max(mx, fabs(val));
      });
    }
    R.loc[0] = mx;
  }
  PN_SYNC();
  return pn_reduce_max<M>(q, w, L PN_LANE_ARG);
}

// inverse metric of knot k, entry j of nc: 1 / (max(diag of the error-state OBJECTIVE Hessian, 0) + rho_primal)
template <class M>
PN_FN double pn_metric_dir(const DevProblem& P, int k, const double* z, bool terminal, int j) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m;
  CostC& C = P.costs[P.cost_index[k]];
  const double sc = (P.opts.cost_dt_scaling && !terminal) ? P.dt[k] : 1.0;
  double v[nz], g[nz], y[nz];
  errstate_col<M>(z, j < ne ? j : 0, v);
  for (int i = 0; i < n; ++i) v[i] = (j < ne) ? v[i] : 0.0;
  for (int i = n; i < nz; ++i) v[i] = (i == n + j - ne) ? 1.0 : 0.0;
  cost_grad_hvp<n, m, true>(C, z, z + n, terminal, v, g, y);
  double dj = 0.0;
  for (int i = 0; i < nz; ++i) dj += v[i] * y[i];
  if constexpr (M::att == ATT_QUAT) {  // second-order term of the attitude map: -I3 (q' dJ/dq)
    double b1 = 0.0;
    for (int i = 0; i < 4; ++i) b1 += z[3 + i] * g[3 + i];
    dj -= (j >= 3 && j < 6) ? b1 : 0.0;
  } else if constexpr (M::att == ATT_MRP || M::att == ATT_RP) {
    double H2[9];
    att_differential2<M::att>(z + 3, g + 3, H2);
    const int jj = (j >= 3 && j < 6) ? j - 3 : 0;
    dj += (j >= 3 && j < 6) ? H2[4 * jj] : 0.0;
  }
  return (terminal && j >= ne) ? 0.0 : 1.0 / (fmax(dj * sc, 0.0) + P.opts.rho_primal);
}

// Linearisation, item by item (one lane per item).  k_pn_lin_col, items [0, (N-1) nc): column j of the error-state Jacobian [A B]
// of step k by a dual number through the RK stages (ForwardDiff-equivalent, like k_expand), stored in the record of the knot the
// defect arrives at.  k_pn_lin_knot, one wave per knot: metric (lane = entry) and active constraint rows of the knot.
template <class M>
PN_FN void pn_lin_column(const PnArgs& q, double* w, int it) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m, nc = ne + m;
  const DevProblem& P = q.a.P;
  const int k = it / nc, j = it % nc;
  const PnRec<M> R = pn_rec<M>(q, w, k), Rn = pn_rec<M>(q, w, k + 1);
  double v[nz];
  errstate_col<M>(R.Z, j < ne ? j : 0, v);
  for (int i = 0; i < n; ++i) v[i] = (j < ne) ? v[i] : 0.0;
  for (int i = n; i < nz; ++i) v[i] = (i == n + j - ne) ? 1.0 : 0.0;
  Dual xd[n], ud[m], xn[n];
  for (int i = 0; i < n; ++i) xd[i] = Dual(R.Z[i], v[i]);
  for (int i = 0; i < m; ++i) ud[i] = Dual(R.Z[n + i], v[n + i]);
  model_step<M, Dual>(P.mp, P.integrator, k, xd, ud, P.dt[k], xn);
  double y[n], col[ne];
  for (int i = 0; i < n; ++i) y[i] = xn[i].d;
  errstate_invmul<M>(Rn.Z, y, col);
  for (int i = 0; i < ne; ++i) Rn.F[i * nc + j] = col[i];
}
// active constraint rows of knot k in error-state coordinates
template <class M>
PN_FN void pn_lin_rows(const PnArgs& q, double* w, int k) {
  constexpr int nc = M::ne + M::m;
  const DevProblem& P = q.a.P;
  const PnRec<M> R = pn_rec<M>(q, w, k);
  const bool terminal = (k == P.N - 1);
  const unsigned long long mask = *R.mask;
  int na = 0;
  pn_for_candidates<M, true>(P, k, R.Z, pn_cp0(P, w), [&](int qi, double, const double* gz, bool) {
    if (!(mask >> qi & 1ull)) return;
    pn_project_row<M>(R.Z, gz, terminal, R.C + (size_t)na * nc);
    ++na;
  });
}
#ifdef TO_PN_HOST
template <class M>
PN_FN void pn_lin_knot(const PnArgs& q, double* w, int k) {
  constexpr int nc = M::ne + M::m;
  const PnRec<M> R = pn_rec<M>(q, w, k);
  for (int j = 0; j < nc; ++j) R.W[j] = pn_metric_dir<M>(q.a.P, k, R.Z, k == q.a.P.N - 1, j);
  pn_lin_rows<M>(q, w, k);
}
#endif

// e / d for 0 <= e < 4096, 2 <= d <= 64 by one multiplication: r = ceil(2^20 / d), e / d = (e r) >> 20  (the error e (r - 2^20 / d) < 4096
// stays below 2^20 / d)
PN_FN unsigned pn_recip(int d) { return ((1u << 20) + (unsigned)d - 1u) / (unsigned)d; }
PN_FN int pn_div(int e, unsigned r) { return (int)(((unsigned)e * r) >> 20); }
// entry e of a lower triangle stored row after row: row i = the largest i with i (i + 1) / 2 <= e  (e < 2^20: the float estimate is off by one at most)
PN_FN int pn_tri_row(int e) {
  int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  if ((i + 1) * (i + 2) / 2 <= e) ++i;
  if (i * (i + 1) / 2 > e) --i;
  return i;
}

#ifndef TO_PN_HOST
// Cholesky factor of an nb x nb block (nb <= PN_NBR; lower triangle in Lc, row stride NB) and the inverse of the factor into Mc, in
// REGISTERS: lane i owns row i of L, then column i of L^-1; what the other lanes need of a row travels by v_readlane (the row
// index is the loop counter: uniform).  The generic path below does the same arithmetic in the same order (column after column,
// the terms of every entry in ascending order) through ~45 LDS phases per block — a third of a projection's time on C5.
constexpr int PN_NBR = 21;
PN_FN double pn_bcast(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
PN_FN bool pn_chol_inv_regs(const double* Lc, double* Mc, int nb, int NB, int lane) {
  const int i = lane < nb ? lane : nb - 1;  // (a lane beyond the block repeats its last row / column)
  double a[PN_NBR], x[PN_NBR];
#pragma unroll
  for (int t = 0; t < PN_NBR; ++t) {
    const double val = Lc[i * NB + (t < nb ? t : 0)];
    a[t] = (t <= i && t < nb) ? val : 0.0;
  }
  // column j of L, then row j of L^-1 (it needs rows <= j of L only): two chains that do not depend on each other until the division by
  // L_jj, in one block so that the scheduler can interleave the square root / division sequence of the one with the dot products of the other
#pragma unroll
  for (int j = 0; j < PN_NBR; ++j) {
    if (j < nb) {
      double v = a[j];
#pragma unroll
      for (int t = 0; t < j; ++t) v -= a[t] * pn_bcast(a[t], j);
      double w = (j == i) ? 1.0 : 0.0;
#pragma unroll
      for (int t = 0; t < j; ++t) w -= pn_bcast(a[t], j) * x[t];
      const double pj = pn_bcast(v, j);
      if (!(pj > 0.0)) return false;
      const double lj = sqrt(pj);
      a[j] = (i == j) ? lj : (i > j ? v / lj : 0.0);
      x[j] = (j < i) ? 0.0 : w / lj;
      Mc[j * NB + i] = x[j];
    }
  }
  return true;
}
#endif

// Block-tridiagonal Cholesky factor of S + rho I, S = D W D': L_k,k-1 = S_k,k-1 L_k-1^-T, L_k = chol(S_kk + rho I - L_k,k-1 L_k,k-1').
// What the records keep is what the SOLVES need — they run several times per factorisation and walk the horizon knot after
// knot, so a knot must cost them one phase, not one per column: the INVERSE of every diagonal factor block,
//   Mi_k = L_k^-1,   Nf_k = Mi_k[:, 0:ne] L_k,k-1   (forward: y_k = Mi_k b_k - Nf_k y_k-1),
//   Pb_k = Mi_k-1' L_k,k-1'                          (backward: x_k-1 = Mi_k-1' y_k-1 - Pb_k x_k[0:ne]).
// (Triangular blocks of 13...21 rows with condition numbers around 1e3: the explicit inverse costs ~1e-13 against substitution,
// far inside what reg_solve's refinement corrects.)  false: a pivot was not positive.
template <class M>
PN_FN bool pn_factor(const PnArgs& q, double* w, const PnLds& L, double rho PN_LANE_PARAM) {
  constexpr int ne = M::ne, nc = M::ne + M::m, ncp = nc | 1;
  const int N = q.a.P.N, NB = L.NB;
  double *Mc = L.LA, *Mp = L.LB, *Lc = L.LW, *Cc = L.Ca, *Cp = L.Cb, *Wc = L.Wa, *Wp = L.Wb;
  // The knot's Jacobian rows and metric come one knot ahead through registers when they fit (see PnStage), and no phase inside the
  // sweep waits for the sweep's own stores (Pb, Mi, Nf are read by the solves, behind the barrier at the end).
  const bool staged = ne * nc <= 64 * PN_PF && (NB - ne) * nc <= 64 * PN_PF;
  PnStage sF, sC, sW;
  unsigned long long rmask = 0;
  if (staged) {
    const PnRec<M> R = pn_rec<M>(q, w, 0);
    pn_fetch(sC, R.C, (R.nbm - ne) * nc PN_LANE_ARG);
    pn_fetch(sW, R.W, nc PN_LANE_ARG);
    rmask = *R.mask;
  }
  int nbp = 0;
  PN_MARK_DECL();
  for (int k = 0; k < N; ++k) {
    const PnRec<M> R = pn_rec<M>(q, w, k);
    int pa;
    if (staged) {
      pa = pn_popc(pn_uniform(rmask));
      if (k > 0) pn_put_rows(L.F, sF, ne * nc, nc, ncp PN_LANE_ARG);
      else PN_FOR(e, ne * ncp) L.F[e] = 0.0;
      pn_put_rows(Cc, sC, (R.nbm - ne) * nc, nc, ncp PN_LANE_ARG);
      pn_put(Wc, sW, nc PN_LANE_ARG);
      if (k + 1 < N) {
        const PnRec<M> Rn = pn_rec<M>(q, w, k + 1);
        pn_fetch(sF, Rn.F, ne * nc PN_LANE_ARG);
        pn_fetch(sC, Rn.C, (Rn.nbm - ne) * nc PN_LANE_ARG);
        pn_fetch(sW, Rn.W, nc PN_LANE_ARG);
        rmask = *Rn.mask;
      }
    } else {
      pa = pn_popc(*R.mask);
      PN_FOR(e, ne * nc) L.F[(e / nc) * ncp + e % nc] = (k > 0) ? R.F[e] : 0.0;
      PN_FOR(e, pa * nc) Cc[(e / nc) * ncp + e % nc] = R.C[e];
      PN_FOR(e, nc) Wc[e] = R.W[e];
    }
    const int nb = ne + pa;
    const unsigned rnb = pn_recip(nb), rnbp = pn_recip(nbp > 0 ? nbp : 1);  // e / nb, e / nbp without the integer-division sequence
    PN_SYNC_LDS();
    PN_MARK(5);
    const double sg = (k == 0) ? 1.0 : -1.0, sgp = (k == 1) ? 1.0 : -1.0;  // coefficient of dx_k in its own arriving-defect row
    // rows times the metric, once: every product below is (row entry * metric entry) * row entry, left to right — FW / CW hold the
    // rounded first product, so the sums are those of the three-factor expression, with two LDS reads per term instead of three
    const int pap = nbp - ne;  // active constraint rows of the previous knot
    if (k > 0) PN_FOR(e, ne * nc) { const int i = e / nc, c = e % nc; L.FW[i * ncp + c] = L.F[i * ncp + c] * Wp[c]; }
    PN_FOR(e, pa * nc) { const int i = e / nc, c = e % nc; L.CW[i * ncp + c] = Cc[i * ncp + c] * Wc[c]; }
    PN_SYNC_LDS();
    // S_kk (lower triangle) region by region, so that no pass mixes the 16-term sums with the one-term entries: defect x defect,
    // constraint x defect, constraint x constraint
    PN_FOR(e, ne * (ne + 1) / 2) {
      const int i = pn_tri_row(e), j = e - i * (i + 1) / 2;
      double v = 0.0;
      if (k > 0) for (int c = 0; c < nc; ++c) v += L.FW[i * ncp + c] * L.F[j * ncp + c];
      if (i == j) v += Wc[i];
      Lc[i * NB + j] = v + (i == j ? rho : 0.0);
    }
    PN_FOR(e, pa * ne) {
      const int i = e / ne, j = e % ne;
      Lc[(ne + i) * NB + j] = sg * L.CW[i * ncp + j] + 0.0;
    }
    PN_FOR(e, pa * (pa + 1) / 2) {
      const int i = pn_tri_row(e), j = e - i * (i + 1) / 2;
      double v = 0.0;
      for (int c = 0; c < nc; ++c) v += L.CW[i * ncp + c] * Cc[j * ncp + c];
      Lc[(ne + i) * NB + ne + j] = v + (i == j ? rho : 0.0);
    }
    if (k > 0) {  // S_k,k-1 (its ne defect rows) into R.Nf's place-holder in LDS: Mc is free until the inverse is formed
      PN_FOR(e, ne * ne) { const int i = e / ne, j = e % ne; Mc[i * NB + j] = sgp * L.FW[i * ncp + j]; }
      const unsigned rpap = pn_recip(pap > 0 ? pap : 1);
      PN_FOR(e, ne * pap) {
        const int i = pn_div(e, rpap), j = e - i * pap;
        double v = 0.0;
        for (int c = 0; c < nc; ++c) v += L.FW[i * ncp + c] * Cp[j * ncp + c];
        Mc[i * NB + ne + j] = v;
      }
    }
    PN_SYNC_LDS();
    if (k > 0) {
      PN_FOR(e, ne * nbp) {  // L_k,k-1 = S_k,k-1 Mi_k-1'  (Mi lower triangular: column j of Mi' is row j of Mi, entries t <= j)
        const int i = pn_div(e, rnbp), j = e - i * nbp;
        double v = 0.0;
        for (int t = 0; t <= j; ++t) v += Mc[i * NB + t] * Mp[j * NB + t];
        L.Lo[i * NB + j] = v;
      }
      PN_SYNC_LDS();
      PN_FOR(e, ne * (ne + 1) / 2) {
        const int i = pn_tri_row(e), j = e - i * (i + 1) / 2;
        double v = 0.0;
        for (int t = 0; t < nbp; ++t) v += L.Lo[i * NB + t] * L.Lo[j * NB + t];
        Lc[i * NB + j] -= v;
      }
      PN_FOR(e, nbp * ne) {  // Pb_k = Mi_k-1' L_k,k-1'  (nbp x ne)
        const int j = e / ne, i = e % ne;
        double v = 0.0;
        for (int t = j; t < nbp; ++t) v += Mp[t * NB + j] * L.Lo[i * NB + t];
        R.Pb[e] = v;
      }
      PN_SYNC_LDS();
    }
    PN_MARK(6);
#ifndef TO_PN_HOST
    if (NB <= PN_NBR) {
      if (!pn_chol_inv_regs(Lc, Mc, nb, NB, lane)) return false;
    } else
#endif
    {
      for (int j = 0; j < nb; ++j) {  // right-looking Cholesky of the block
        const double pj = Lc[j * NB + j];
        if (!(pj > 0.0)) return false;
        const double lj = sqrt(pj);
        PN_FOR(i, nb - j) { const int ii = j + i; Lc[ii * NB + j] = (i == 0) ? lj : Lc[ii * NB + j] / lj; }
        PN_SYNC_LDS();
        const int cnt = nb - j - 1;
        PN_FOR(e, cnt * cnt) {
          const int i = j + 1 + e / cnt, c = j + 1 + e % cnt;
          if (c <= i) Lc[i * NB + c] -= Lc[i * NB + j] * Lc[c * NB + j];
        }
        PN_SYNC_LDS();
      }
      PN_FOR(j, nb) {  // Mi_k = L_k^-1, one column per lane (forward substitution on e_j)
        for (int i = 0; i < nb; ++i) {
          double v = (i == j) ? 1.0 : 0.0;
          for (int t = j; t < i; ++t) v -= Lc[i * NB + t] * Mc[t * NB + j];
          Mc[i * NB + j] = (i < j) ? 0.0 : v / Lc[i * NB + i];
        }
      }
    }
    PN_SYNC_LDS();
    PN_MARK(7);
    PN_FOR(e, nb * nb) { const int i = pn_div(e, rnb); R.Mi[e] = Mc[i * NB + (e - i * nb)]; }
    if (k > 0) PN_FOR(e, nb * nbp) {  // Nf_k = Mi_k[:, 0:ne] L_k,k-1
      const int i = pn_div(e, rnbp), j = e - i * nbp;
      double v = 0.0;
      for (int t = 0; t < ne && t <= i; ++t) v += Mc[i * NB + t] * L.Lo[t * NB + j];
      R.Nf[e] = v;
    }
    PN_SYNC_LDS();
    PN_MARK(5);
    double* t;
    t = Mc; Mc = Mp; Mp = t; t = Cc; Cc = Cp; Cp = t; t = Wc; Wc = Wp; Wp = t;
    nbp = nb;
  }
  PN_SYNC();
  return true;
}

// The sweeps of pn_chol_solve (below) with the blocks staged one knot ahead (see PnStage); same arithmetic in the same order
template <class M>
PN_FN void pn_chol_solve_staged(const PnArgs& q, double* w, const PnLds& L, int which PN_LANE_PARAM) {
  constexpr int ne = M::ne;
  const int N = q.a.P.N;
  double *yc = L.va, *yp = L.vb, *bM = L.LA, *bN = L.LB, *bv = L.LW;
  PnStage sM, sN, sv;
  unsigned long long rmask;
  {
    const PnRec<M> R = pn_rec<M>(q, w, 0);
    pn_fetch(sM, R.Mi, R.nbm * R.nbm PN_LANE_ARG);
    pn_fetch(sv, R.v(which), R.nbm PN_LANE_ARG);
    rmask = *R.mask;
  }
  int nbp = 0, nbpm = 0;
  for (int k = 0; k < N; ++k) {  // forward: y_k = Mi_k b_k - Nf_k y_k-1
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const int nbm = R.nbm, nb = ne + pn_popc(pn_uniform(rmask));
    pn_put(bM, sM, nbm * nbm PN_LANE_ARG);
    pn_put(bN, sN, nbm * nbpm PN_LANE_ARG);
    pn_put(bv, sv, nbm PN_LANE_ARG);
    if (k + 1 < N) {
      const PnRec<M> Rn = pn_rec<M>(q, w, k + 1);
      pn_fetch(sM, Rn.Mi, Rn.nbm * Rn.nbm PN_LANE_ARG);
      pn_fetch(sN, Rn.Nf, Rn.nbm * nbm PN_LANE_ARG);
      pn_fetch(sv, Rn.v(which), Rn.nbm PN_LANE_ARG);
      rmask = *Rn.mask;
    }
    PN_SYNC_LDS();
    double* v = R.v(which);
    PN_FOR(i, nb) {
      double s = 0.0;
      for (int t = 0; t <= i; ++t) s += bM[i * nb + t] * bv[t];
      if (k > 0) for (int t = 0; t < nbp; ++t) s -= bN[i * nbp + t] * yp[t];
      yc[i] = s;
      v[i] = s;
    }
    double* t = yc; yc = yp; yp = t;
    nbp = nb; nbpm = nbm;
  }
  PN_SYNC();
  {
    const PnRec<M> R = pn_rec<M>(q, w, N - 1);
    pn_fetch(sM, R.Mi, R.nbm * R.nbm PN_LANE_ARG);
    pn_fetch(sv, R.v(which), R.nbm PN_LANE_ARG);
    rmask = *R.mask;
  }
  for (int k = N - 1; k >= 0; --k) {  // backward: x_k = Mi_k' y_k - Pb_k+1 x_k+1[0:ne]
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const int nbm = R.nbm, nb = ne + pn_popc(pn_uniform(rmask));
    pn_put(bM, sM, nbm * nbm PN_LANE_ARG);
    if (k < N - 1) pn_put(bN, sN, nbm * ne PN_LANE_ARG);
    pn_put(bv, sv, nbm PN_LANE_ARG);
    if (k > 0) {
      const PnRec<M> Rn = pn_rec<M>(q, w, k - 1);
      pn_fetch(sM, Rn.Mi, Rn.nbm * Rn.nbm PN_LANE_ARG);
      pn_fetch(sN, R.Pb, Rn.nbm * ne PN_LANE_ARG);  // Pb_k couples knot k-1 to this one
      pn_fetch(sv, Rn.v(which), Rn.nbm PN_LANE_ARG);
      rmask = *Rn.mask;
    }
    PN_SYNC_LDS();
    double* v = R.v(which);
    PN_FOR(i, nb) {
      double s = 0.0;
      for (int t = i; t < nb; ++t) s += bM[t * nb + i] * bv[t];
      if (k < N - 1) for (int t = 0; t < ne; ++t) s -= bN[i * ne + t] * yp[t];
      yc[i] = s;
      v[i] = s;
    }
    double* t = yc; yc = yp; yp = t;
  }
  PN_SYNC();
}

// (L L') x = b in place on vector `which` of the records: one phase per knot and direction (see pn_factor); inside a sweep only the
// LDS traffic is ordered (the stores to the workspace vector are read again by the NEXT sweep, behind a full barrier).
template <class M>
PN_FN void pn_chol_solve(const PnArgs& q, double* w, const PnLds& L, int which PN_LANE_PARAM) {
  constexpr int ne = M::ne;
  const int N = q.a.P.N;
  if (L.NB * L.NB <= 64 * PN_PF) { pn_chol_solve_staged<M>(q, w, L, which PN_LANE_ARG); return; }
  double *yc = L.va, *yp = L.vb;
  int nbp = 0;
  for (int k = 0; k < N; ++k) {  // forward: y_k = Mi_k b_k - Nf_k y_k-1
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const int nb = ne + pn_popc(*R.mask);
    double* v = R.v(which);
    PN_FOR(i, nb) {
      double s = 0.0;
      for (int t = 0; t <= i; ++t) s += R.Mi[i * nb + t] * v[t];
      if (k > 0) for (int t = 0; t < nbp; ++t) s -= R.Nf[i * nbp + t] * yp[t];
      yc[i] = s;
    }
    PN_SYNC();   // (the rows above read v of the whole knot before any lane overwrites its entry)
    PN_FOR(i, nb) v[i] = yc[i];
    double* t = yc; yc = yp; yp = t;
    nbp = nb;
  }
  PN_SYNC();
  for (int k = N - 1; k >= 0; --k) {  // backward: x_k = Mi_k' y_k - Pb_k+1 x_k+1[0:ne]
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const int nb = ne + pn_popc(*R.mask);
    double* v = R.v(which);
    const double* Pb = (k < N - 1) ? pn_rec<M>(q, w, k + 1).Pb : R.Pb;
    PN_FOR(i, nb) {
      double s = 0.0;
      for (int t = i; t < nb; ++t) s += R.Mi[t * nb + i] * v[t];
      if (k < N - 1) for (int t = 0; t < ne; ++t) s -= Pb[i * ne + t] * yp[t];
      yc[i] = s;
    }
    PN_SYNC();
    PN_FOR(i, nb) v[i] = yc[i];
    double* t = yc; yc = yp; yp = t;
  }
  PN_SYNC();
}

// dZ = -W D' lambda (lambda = vector `which`)
template <class M>
PN_FN void pn_step(const PnArgs& q, double* w, int which PN_LANE_PARAM) {
  constexpr int ne = M::ne, nc = M::ne + M::m;
  const int N = q.a.P.N;
  PN_FOR(it, N * nc) {
    const int k = it / nc, c = it % nc;
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const double* lam = R.v(which);
    const int pa = pn_popc(*R.mask);
    double s = 0.0;
    if (k < N - 1) {
      const PnRec<M> Rn = pn_rec<M>(q, w, k + 1);
      const double* ln = Rn.v(which);
      for (int i = 0; i < ne; ++i) s += Rn.F[i * nc + c] * ln[i];
    }
    if (c < ne) s += (k == 0 ? 1.0 : -1.0) * lam[c];
    for (int a = 0; a < pa; ++a) s += R.C[a * nc + c] * lam[ne + a];
    R.dZ[c] = -(R.W[c] * s);
  }
  PN_SYNC();
}

// r = b + D dZ  (= b - S lambda for the lambda dZ was formed from); returns |r|_2
template <class M>
PN_FN double pn_residual(const PnArgs& q, double* w, const PnLds& L, int vb, int vr PN_LANE_PARAM) {
  constexpr int ne = M::ne, nc = M::ne + M::m;
  const int N = q.a.P.N, NB = L.NB;
  PN_FOR(it, N * NB) {
    const int k = it / NB, i = it % NB;
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const int nb = ne + pn_popc(*R.mask);
    if (i < nb) {
      double s = 0.0;
      if (i < ne) {
        if (k > 0) { const PnRec<M> Rp = pn_rec<M>(q, w, k - 1); for (int c = 0; c < nc; ++c) s += R.F[i * nc + c] * Rp.dZ[c]; }
        s += (k == 0 ? 1.0 : -1.0) * R.dZ[i];
      } else for (int c = 0; c < nc; ++c) s += R.C[(i - ne) * nc + c] * R.dZ[c];
      R.v(vr)[i] = R.v(vb)[i] + s;
    }
  }
  PN_SYNC();
  PN_FOR(j, 64) {
    double s = 0.0;
    for (int k = j; k < N; k += 64) {
      const PnRec<M> R = pn_rec<M>(q, w, k);
      const int nb = ne + pn_popc(*R.mask);
      const double* r = R.v(vr);
      for (int i = 0; i < nb; ++i) s += r[i] * r[i];
    }
    L.red[j] = s;
  }
  PN_SYNC();
  double s = 0.0;
  for (int j = 0; j < 64; ++j) s += L.red[j];
  PN_SYNC();
  return sqrt(s);
}

// Altro reg_solve: dl = (S + rho I)^-1 d refined against the unregularised S; leaves dZ = -W D' dl
template <class M>
PN_FN void pn_reg_solve(const PnArgs& q, double* w, const PnLds& L PN_LANE_PARAM) {
  constexpr int ne = M::ne;
  const int N = q.a.P.N, NB = L.NB;
  PN_FOR(it, N * NB) { const int k = it / NB, i = it % NB; const PnRec<M> R = pn_rec<M>(q, w, k); if (i < R.nbm) R.v(PN_VDL)[i] = R.v(PN_VD)[i]; }
  PN_SYNC();
  { PN_TIC(); pn_chol_solve<M>(q, w, L, PN_VDL PN_LANE_ARG); PN_TOC(1); }
  for (int it = 0;; ++it) {
    { PN_TIC(); pn_step<M>(q, w, PN_VDL PN_LANE_ARG); PN_TOC(3); }
    if (it >= PN_REG_SOLVE_ITERS) break;
    double nr;
    { PN_TIC(); nr = pn_residual<M>(q, w, L, PN_VD, PN_VR PN_LANE_ARG); PN_TOC(4); }
    if (nr < PN_REG_SOLVE_TOL) break;
    { PN_TIC(); pn_chol_solve<M>(q, w, L, PN_VR PN_LANE_ARG); PN_TOC(1); }
    PN_FOR(e, N * NB) { const int k = e / NB, i = e % NB; const PnRec<M> R = pn_rec<M>(q, w, k); if (i < ne + pn_popc(*R.mask)) R.v(PN_VDL)[i] += R.v(PN_VR)[i]; }
    PN_SYNC();
  }
}

// Round `round` of the polish of trajectory b, first part: (round 0: fetch the trajectory;) fresh active set and its violation; if
// that is within constraint_tolerance, the budget of n_steps + 1 linearisations is spent (Altro projection_solve!: while count
// <= n_steps) or the last factorisation failed: evaluate what the trajectory violates now (constraints as max_violation reports
// them, and defects), write it back, set the status.  Otherwise the trajectory is ACTIVE for k_pn_lin / k_pn_project.
template <class M>
PN_FN void pn_begin(const PnArgs& q, int b, double* w, double* lds_mem, int round PN_LANE_PARAM) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m;
  const KArgs& a = q.a;
  const DevProblem& P = a.P;
  const int N = P.N;
  const to_solver_opts& o = P.opts;
  if (round > 0 && w[PN_H_STATE] == (double)PN_DONE) return;
  const PnLds L = pn_lds<M>(lds_mem, q.nbmax);
  const int tile = b >> 6, tl = b & 63;
  double* Xn = a.Xs + ((size_t)tile * (size_t)(N * n)) * 64 + tl;
  double* Un = a.Us + ((size_t)tile * (size_t)((N - 1) * m)) * 64 + tl;
  const double* x0t = a.x0 + ((size_t)tile * (size_t)n) * 64 + tl;
  double x0[n];
  for (int i = 0; i < n; ++i) x0[i] = EL(x0t, i);
  if (round == 0) {
    PN_FOR(e, N * nz) {
      const int k = e / nz, i = e % nz;
      pn_rec<M>(q, w, k).Z[i] = i < n ? EL(Xn, k * n + i) : (k < N - 1 ? EL(Un, k * m + (i - n)) : 0.0);
    }
    PN_FOR(j, 1) { w[PN_H_STEPS] = 0.0; w[PN_H_FAILED] = 0.0; w[PN_H_TRAJ] = (double)b; }
    PN_SYNC();
  }
  const bool failed = w[PN_H_FAILED] != 0.0;
  const double viol = pn_eval<M>(q, w, L, x0, false, true, PN_VD PN_LANE_ARG);
  if (!(viol <= o.constraint_tolerance || round > o.n_steps || failed)) {
    PN_FOR(j, 1) { w[PN_H_STATE] = (double)PN_ACTIVE; w[PN_H_VIOL] = viol; }
    PN_SYNC();
    return;
  }
  PN_FOR(k, N) {
    const PnRec<M> R = pn_rec<M>(q, w, k);
    const double* zp = pn_rec<M>(q, w, k > 0 ? k - 1 : 0).Z;
    double e[ne], mx = 0.0;
    pn_defect<M>(P, k, zp, R.Z, x0, e);
    for (int i = 0; i < ne; ++i) pn_upd_max(mx, fabs(e[i]));
    if (P.n_cons > 0) pn_upd_max(mx, knot_violation<M>(P, k, R.Z, R.Z + n, pn_cp0(P, w)));
    R.loc[0] = mx;
  }
  PN_SYNC();
  const double cmax = pn_reduce_max<M>(q, w, L PN_LANE_ARG);
  PN_FOR(e, N * nz) {
    const int k = e / nz, i = e % nz;
    const double v = pn_rec<M>(q, w, k).Z[i];
    if (i < n) EL(Xn, k * n + i) = v; else if (k < N - 1) EL(Un, k * m + (i - n)) = v;
  }
  PN_FOR(j, 1) {
    q.it_pn[b] = (int)w[PN_H_STEPS];
    q.cmax_out[b] = cmax;
    a.status[b] = (cmax <= o.constraint_tolerance) ? TO_SOLVE_SUCCEEDED : TO_PROJECTION_FAIL;
    w[PN_H_STATE] = (double)PN_DONE;
  }
  PN_SYNC();
}

// ... second part, after the linearisation: Altro _projection_solve! on the frozen active set
template <class M>
PN_FN void pn_project(const PnArgs& q, int b, double* w, double* lds_mem PN_LANE_PARAM) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m;
  const KArgs& a = q.a;
  const DevProblem& P = a.P;
  const int N = P.N;
  const to_solver_opts& o = P.opts;
  if (w[PN_H_STATE] != (double)PN_ACTIVE) return;
  const PnLds L = pn_lds<M>(lds_mem, q.nbmax);
  const int NB = L.NB;
  const int tile = b >> 6, tl = b & 63;
  const double* x0t = a.x0 + ((size_t)tile * (size_t)n) * 64 + tl;
  double x0[n];
  for (int i = 0; i < n; ++i) x0[i] = EL(x0t, i);
  const double viol = w[PN_H_VIOL];
  PN_SYNC();  // every lane has read the header before lane 0 rewrites it
  PN_FOR(j, 1) w[PN_H_STEPS] += 1.0;
#if defined(TO_PN_TIMING) && !defined(TO_PN_HOST)
  const long long pn_t_all = wall_clock64();
  if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 8; ++i) pn_tacc[i] = 0;
#endif
  bool factored;
  { PN_TIC(); factored = pn_factor<M>(q, w, L, o.rho_chol PN_LANE_ARG); PN_TOC(0); }
  if (!factored) {
    PN_FOR(j, 1) w[PN_H_FAILED] = 1.0;
    PN_SYNC();
    return;
  }
  double viol_prev = viol;
  for (int count = 0; count < PN_REFINEMENTS; ++count) {
    pn_reg_solve<M>(q, w, L PN_LANE_ARG);
    double alpha = 1.0, v = 0.0;
    bool accepted = false;
    for (int ls = 0; ls < PN_LS_TRIALS; ++ls) {
      PN_FOR(k, N) {
        const PnRec<M> R = pn_rec<M>(q, w, k);
        double stp[ne];
        for (int i = 0; i < ne; ++i) stp[i] = alpha * R.dZ[i];
        state_add<M>(R.Z, stp, R.Zb);
        for (int j = 0; j < m; ++j) R.Zb[n + j] = (k < N - 1) ? R.Z[n + j] + alpha * R.dZ[ne + j] : 0.0;
      }
      PN_SYNC();
      { PN_TIC(); v = pn_eval<M>(q, w, L, x0, true, false, PN_VDN PN_LANE_ARG); PN_TOC(2); }
      if (v < viol_prev) { accepted = true; break; }
      alpha *= 0.5;
    }
    if (!accepted) break;
    PN_FOR(e, N * nz) { const PnRec<M> R = pn_rec<M>(q, w, e / nz); R.Z[e % nz] = R.Zb[e % nz]; }
    PN_FOR(e, N * NB) { const int k = e / NB, i = e % NB; const PnRec<M> R = pn_rec<M>(q, w, k); if (i < R.nbm) R.v(PN_VD)[i] = R.v(PN_VDN)[i]; }
    PN_SYNC();
    const double before = viol_prev;
    viol_prev = v;
    if (v < o.constraint_tolerance) break;
    if (before < 1.0) { if (log10(v) / log10(before) < o.r_threshold) break; }
    else if (!(v < 0.5 * before)) break;
  }
#if defined(TO_PN_TIMING) && !defined(TO_PN_HOST)
  if (blockIdx.x == 0 && lane == 0)
    printf("pn_project us: all %.1f factor %.1f (stage+store %.1f, products %.1f, chol+inverse %.1f) chol_solve %.1f eval %.1f step %.1f residual %.1f\n",
           0.01 * (wall_clock64() - pn_t_all), 0.01 * pn_tacc[0], 0.01 * pn_tacc[5], 0.01 * pn_tacc[6], 0.01 * pn_tacc[7], 0.01 * pn_tacc[1], 0.01 * pn_tacc[2],
           0.01 * pn_tacc[3], 0.01 * pn_tacc[4]);
#endif
}

#ifndef TO_PN_HOST
template <class M>
__global__ void __launch_bounds__(64) k_pn_begin(PnArgs q, int round) {
  extern __shared__ double pn_lds_mem[];
  const int b = q.list[q.base + blockIdx.x];
  pn_begin<M>(q, b, q.ws + (size_t)blockIdx.x * (size_t)q.koff[q.a.P.N], pn_lds_mem, round, (int)threadIdx.x);
}
template <class M>
__global__ void __launch_bounds__(64) k_pn_project(PnArgs q) {
  extern __shared__ double pn_lds_mem[];
  const int b = q.list[q.base + blockIdx.x];
  pn_project<M>(q, b, q.ws + (size_t)blockIdx.x * (size_t)q.koff[q.a.P.N], pn_lds_mem, (int)threadIdx.x);
}
// grid (trajectories of the launch, ceil(items / 64)), EXEC full (a lane beyond the end repeats the last item).  Two kernels:
// the dual-number RK step of the Quadrotor takes every register there is, the knot items (cost / constraint descriptors that
// differ between the lanes' knots: lane-divergent branches) must not share its allocation (spill placement).
template <class M>
__global__ void __launch_bounds__(64) k_pn_lin_col(PnArgs q) {
  constexpr int nc = M::ne + M::m;
  double* w = q.ws + (size_t)blockIdx.x * (size_t)q.koff[q.a.P.N];
  if (w[PN_H_STATE] != (double)PN_ACTIVE) return;
  pn_lin_column<M>(q, w, min((int)(blockIdx.y * 64 + threadIdx.x), (q.a.P.N - 1) * nc - 1));
}
// one WAVE per knot (grid.y = N): the knot's cost and constraint descriptors are wave-uniform then — scalar loads, uniform
// branches (with a lane per knot the dense-cost branches of lanes on differently-costed knots diverged around the register-hungry
// ErrorQuadratic duals, exactly where hipcc's spill placement is unsafe: DESIGN.md §6).  Lane j < nc owns metric entry j; the
// active rows are built by every lane alike (same values to the same addresses).
template <class M>
__global__ void __launch_bounds__(64) k_pn_lin_knot(PnArgs q) {
  constexpr int nc = M::ne + M::m;
  double* w = q.ws + (size_t)blockIdx.x * (size_t)q.koff[q.a.P.N];
  if (w[PN_H_STATE] != (double)PN_ACTIVE) return;
  const int k = blockIdx.y, j = min((int)threadIdx.x, nc - 1);
  const PnRec<M> R = pn_rec<M>(q, w, k);
  R.W[j] = pn_metric_dir<M>(q.a.P, k, R.Z, k == q.a.P.N - 1, j);
  pn_lin_rows<M>(q, w, k);
}

// max |x_1 (-) x0|, |f(x_k, u_k) (-) x_{k+1}| of the nominal trajectory (to_dynamics_defect): one lane per trajectory
template <class M>
__global__ void __launch_bounds__(64) k_defect(KArgs a, double* out) {
  constexpr int n = M::n, m = M::m, ne = M::ne;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const double* X = TILE_PTR(a.Xs, P.N * n);
  const double* U = TILE_PTR(a.Us, (P.N - 1) * m);
  const double* x0t = TILE_PTR(a.x0, n);
  double zp[n + m], z[n + m], x0[n], e[ne], mx = 0.0;
  for (int i = 0; i < n; ++i) { x0[i] = EL(x0t, i); z[i] = EL(X, i); }
  pn_defect<M>(P, 0, z, z, x0, e);
  for (int i = 0; i < ne; ++i) pn_upd_max(mx, fabs(e[i]));
  for (int k = 1; k < P.N; ++k) {
    for (int i = 0; i < n; ++i) zp[i] = z[i];
    for (int i = 0; i < m; ++i) zp[n + i] = EL(U, (k - 1) * m + i);
    for (int i = 0; i < n; ++i) z[i] = EL(X, k * n + i);
    pn_defect<M>(P, k, zp, z, x0, e);
    for (int i = 0; i < ne; ++i) pn_upd_max(mx, fabs(e[i]));
  }
  out[b] = mx;
}
#endif

}  // namespace to
