// k_forward.h — forward pass: concurrent line search + per-trajectory solver state machine.
//
// One wave holds CW line-search candidates x TW = 64/CW trajectories: hardware lane h = q*TW + t evaluates step size
// alpha = decrease^(c0 + q) for trajectory b0 + t.  The CW candidates of a trajectory therefore sit in ONE wave:
//   * they share the trajectory's gains: a knot's K,d row (trajectory-major, Kt) is brought into LDS once per wave by
//     DMA and read with CW-way broadcasts — the first version gave every candidate its own wave, which re-read the
//     gains 16 times through L2 (5 GB per Quadrotor round) and issued 24 DMA instructions per knot per wave;
//   * the first accepted step size (identical to sequential backtracking) is found with a ballot at the end of the
//     rollout, and the convergence / AL state machine runs right there: no candidate cost arrays, no k_select launch;
//   * a trajectory that rejected all CW step sizes continues with the next CW inside the same kernel (only the waves
//     that still hold a searching trajectory keep running): no compacted lists, no further launches.
#pragma once
#include <type_traits>

#include "common.h"
#include "ls_round.h"

#ifndef TO_FWD_WAVES
#define TO_FWD_WAVES 1
#endif
namespace to {

template <class M, bool WITHK>
struct FwdKnot {  // nominal state/control of one knot (+ its gains row for models that do not stage gains through LDS)
  static constexpr int n = M::n, m = M::m, RSK = Gains<M>::RSK;
  double x[n], u[m], kd[WITHK ? RSK : 1];
  // pointers are already at this knot (the caller walks them): constant offsets, no address arithmetic per load
  __device__ __forceinline__ void load(const double* pXk, const double* pUk, const double* pKk) {
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = EL(pXk, i);
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = EL(pUk, j);
    if constexpr (WITHK) {
#pragma unroll
      for (int i = 0; i < RSK; ++i) kd[i] = pKk[i];
    }
  }
};

// Gains rows of knot k of the wave's TW trajectories, global -> LDS by DMA (global_load_lds_dwordx4: no staging
// registers, the wave keeps computing).  A row is RSK doubles = PCS 16-byte pieces; piece GL = i*64 + lane of
// instruction i belongs to trajectory GL / PCS and lands at kbuf + 16 GL, i.e. the rows sit back to back in LDS
// (row stride RSK doubles: 416 B for the Quadrotor, so the TW rows read together fall on distinct banks).
template <class M>
__device__ __forceinline__ void stage_gains(const double* Kt, int b, int TW, int k, int N, double* kbuf, int hw) {
  constexpr int RSK = Gains<M>::RSK;
  static_assert(RSK % 2 == 0, "gains rows must be whole 16-byte pieces");
  constexpr int PCS = RSK / 2;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int pieces = TW * PCS;
  for (int i0 = 0; i0 < pieces; i0 += 64) {
    const int gl = i0 + hw;
    int t = gl / PCS;
    const int off = gl - t * PCS;
    t = t < TW ? t : TW - 1;  // lanes past the last row re-fetch it (their LDS pieces are never read)
    const int bt = __shfl(b, t);  // trajectory of row t: lane t of the wave holds it (q = 0), whatever maps lanes to trajectories
    const char* src = (const char*)(Kt + ((size_t)bt * (N - 1) + k) * RSK) + off * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)((char*)kbuf + (size_t)i0 * 16), 16, 0, 0);
  }
}
template <class M>
__host__ __device__ inline int gains_lds_doubles(int TW) {  // per buffer, whole DMA instructions
  constexpr int PCS = Gains<M>::RSK / 2;
  return ((TW * PCS + 63) / 64) * 64 * 2;
}

// One line-search candidate per lane: closed-loop rollout of trajectory (tile, lane) with step size alpha, stored in lane hw of
// candidate block wblock; its (AL) cost J, gradient metric gsum/(N-1) and admissibility ok.  Lanes with live == false roll out as well (a
// partially masked wave issues FP64 ~1.3x slower on gfx950) but store nothing.
// MODE bit0: simple_stage (stage cost preloaded into registers, uniform dt); bit1: constraints present (AL terms);
// bit2: RK4 fixed at compile time; bit3: dense costs / non-selector constraints possible (else compiled out);
// bit4: the register-cached control constraints are unit SOCs (problem_dev.h unit_soc_desc).
// kbuf: the wave's two LDS buffers for DMA-staged gains (M::lds_gains); krow: this lane's row offset in a buffer.
template <class M, int MODE>
__device__ __forceinline__ void forward_candidate(const KArgs& a, int tile, int lane, int b, bool live, double alpha, int wblock, double* kbuf,
                                                  int kbuf_len, int krow, int b0, int TW, int hw, double& J_out, double& g_out, bool& ok_out) {
  constexpr int n = M::n, m = M::m, ne = M::ne, RSK = Gains<M>::RSK;
  constexpr bool SIMPLE = (MODE & 1) != 0, CONS = (MODE & 2) != 0, GEN = (MODE & 8) != 0;
  constexpr bool KLDS = M::lds_gains;
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  const int N = P.N;
  const double* Xc = TILE_PTR(a.Xs, N * n);  // the nominal (slot 0)
  const double* Uc = TILE_PTR(a.Us, (N - 1) * m);
  // candidates: forward-wave-major (common.h); a wave's candidate stores are whole 512-byte rows.  Lanes
  // without a candidate store as well — into the dump block behind the last wave's, never read — so that the rollout loop
  // runs with EXEC full throughout (no lane-divergent region: see StageCostLds::load for what one cost here)
  const size_t cblock = live ? (size_t)wblock : (size_t)a.dump_wave;
  double* Xn = a.Xc + (cblock * (size_t)(N * n)) * 64 + hw;
  double* Un = a.Uc + (cblock * (size_t)((N - 1) * m)) * 64 + hw;
  const double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
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
  // stage cost: in VGPRs for the small models, in LDS for the models whose rollout loop has no register to spare
  typename std::conditional<KLDS, StageCostLds<n, m>, StageCostDiag<n, m>>::type sc;
  const bool has_gl = P.gl != nullptr;  // wave-uniform (a kernel argument): the branches on it are scalar
  const double* gl0 = TILE_PTR(P.gl, P.n_costs * (n + m));
  const double* cp0 = TILE_PTR(P.cp, P.n_cp);  // per-trajectory constraint parameters (general variants)
  const int ci0 = P.cost_index[0];
  double h0 = 0.0;
  if constexpr (SIMPLE) {
    if constexpr (KLDS) { sc.load(P.costs[P.cost_index[0]], kbuf + 2 * (size_t)kbuf_len, hw); WAVE_SYNC(); }
    else sc.load(P.costs[P.cost_index[0]]);
    h0 = P.dt[0];
  }
  // stage constraints on the control block (norm / SOC / one-sided bounds on u) are cached in registers once
  ConStage<n, m, (MODE & 16) != 0> cs0, cs1;  // bit4: every cacheable control constraint is a unit SOC (DevProblem::unit_soc)
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
  for (int i = 0; i < n; ++i) xb[i] = EL(px0, i);
  if constexpr (KLDS) stage_gains<M>(a.Kt, b, TW, 0, N, kbuf, hw);
  FwdKnot<M, !KLDS> nxt;
  nxt.load(Xc, Uc, pK);
  const double *pXn = Xc + n * 64, *pUn = Uc + m * 64, *pKn = pK + RSK;  // knot k+1 of the nominal
  double *pXo = Xn, *pUo = Un;                                          // where knot k's candidate state / control go
  const bool all_cached = CONS && uncached == 0;  // wave-uniform: the loop then never touches the descriptor table
  if (ncs > 0) cs0.prefetch(0);
  if (ncs > 1) cs1.prefetch(0);
  for (int k = 0; k < N - 1; ++k) {
    // this knot's gains have landed in LDS, its nominal is in registers; the stores of the previous knot were issued a
    // whole knot ago, so waiting for everything costs nothing extra
    if constexpr (KLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const FwdKnot<M, !KLDS> cur = nxt;
    if (ncs > 0) cs0.advance();
    if (ncs > 1) cs1.advance();
    // The next knot's DMA / loads go out first (the compiler orders an LDS-DMA behind every earlier VMEM operation, so it
    // must not follow the stores), then x̄_k: nothing issued here is needed before the next wait, a whole knot away
    const double* kcur = kbuf + (size_t)(k & 1) * kbuf_len + krow;
    if (k + 1 < N - 1) {
      if constexpr (KLDS) stage_gains<M>(a.Kt, b, TW, k + 1, N, kbuf + (size_t)((k + 1) & 1) * kbuf_len, hw);
      nxt.load(pXn, pUn, pKn);
      if (ncs > 0) cs0.prefetch(k + 1);
      if (ncs > 1) cs1.prefetch(k + 1);
    }
    pXn += n * 64; pUn += m * 64; pKn += RSK;
    // candidate states are 3/4 of what this kernel moves, and with the chip full its duration follows the bytes it stores (C5: 560 /
    // 950 / 1800 us per launch at 4 / 8 / 16 candidates per trajectory): the solve loop then has only the controls stored
    if (a.store_x) {
#pragma unroll
      for (int i = 0; i < n; ++i) EL(pXo, i) = xb[i];
      pXo += n * 64;
    }
    double dx[ne], ub[m], xn[n];
    state_diff<M>(xb, cur.x, dx);
    double gk = 0.0;
#pragma unroll
    for (int j = 0; j < m; ++j) {
      double kr[ne + 1];  // gains row j: all of it is requested before the first product (one LDS round trip per row, not per pair)
#pragma unroll
      for (int i = 0; i <= ne; ++i) kr[i] = KLDS ? kcur[j * (ne + 1) + i] : cur.kd[KLDS ? 0 : j * (ne + 1) + i];
      if constexpr (KLDS) __builtin_amdgcn_sched_barrier(0);
      const double dj = kr[ne];
      double du = dj * alpha;
#pragma unroll
      for (int i = 0; i < ne; ++i) du += kr[i] * dx[i];
      ub[j] = cur.u[j] + du;
      EL(pUo, j) = ub[j];
      gk = fmax(gk, fabs(dj) * rcp_fast(fabs(ub[j]) + 1.0));
    }
    pUo += m * 64;
    gsum += gk;
    const double h = SIMPLE ? h0 : P.dt[k];
    double Jk = SIMPLE ? sc.eval(xb, ub) : cost_eval<n, m, GEN>(P.costs[P.cost_index[k]], xb, ub);
    if constexpr (GEN) {  // per-trajectory q, r: in the general variants only (launch_forward picks one when DevProblem::gl is set), so that
      if (has_gl) Jk += goal_lin_cost<n, m>(gl0, SIMPLE ? ci0 : (int)P.cost_index[k], xb, ub);  // the default kernels keep their loops as they were
    }
    if (dt_scaling) Jk *= h;
    if constexpr (CONS) {
      if (all_cached) {
        double Ja = 0.0;
        if (ncs > 0) Ja += cs0.term(ub);
        if (ncs > 1) Ja += cs1.term(ub);
        Jk += Ja;
      } else Jk += knot_al_cached<M, GEN>(P, k, xb, ub, lam0, mu0, ncs, cs0, cs1, cp0);
    }
    J += Jk;
    model_step<M, double, (MODE & 4) ? INTEG_RK4 : -1>(mp, integrator, k, xb, ub, h, xn);
    double mx = 0.0, mu_ = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) { xb[i] = xn[i]; const double v = fabs(xn[i]); if (!(v <= mx)) mx = v; }
#pragma unroll
    for (int j = 0; j < m; ++j) { const double v = fabs(ub[j]); if (!(v <= mu_)) mu_ = v; }
    // a rollout that left the admissible box is rejected; its lane keeps stepping (values are never used) so that the
    // wave stays converged, and the wave stops once no live lane is inside the box any more
    if (!(mx <= max_x) || !(mu_ <= max_u)) ok = false;
    if (__ballot(live && ok) == 0) break;
  }
  if (a.store_x && __ballot(live && ok) != 0) {  // wave-uniform: the loop ran to its end, pXo has walked to the terminal knot
#pragma unroll
    for (int i = 0; i < n; ++i) EL(pXo, i) = xb[i];  // x̄_N (a rejected candidate's is never read)
  }
  {
    double u0[m];
#pragma unroll
    for (int j = 0; j < m; ++j) u0[j] = 0.0;
    J += knot_cost<M, GEN>(P, N - 1, xb, u0, lam0, mu0, true, gl0, cp0);
  }
  if constexpr (KLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still be in flight when the next pass refills the buffers
  J_out = J; g_out = gsum / (N - 1); ok_out = ok;
}

// Accepting a step whose candidate states were not stored (KArgs::store_x == 0), in two launches over the trajectories of this step
// (one lane per trajectory; with active-list compaction: the entries of this step's list, in index order):
//   k_accept_gather_u  copies the accepted candidate's CONTROLS onto slot 0 — gathered loads (every lane reads its own accepted
//                      slot), nothing depends on them inside the wave, so dozens are in flight per lane;
//   k_accept_roll      rolls the nominal controls out again from x0 onto the nominal states.  Same step function, same operands and —
//                      this translation unit is compiled with -ffp-contract=on — the same fused operations as in forward_candidate:
//                      the states are bit-identical to the ones the line search evaluated (tests/test_gpu_parity.py::
//                      test_accept_by_rollout* compare whole solves for equality).
// One kernel did both at first (gathered control loads four knots ahead of the rollout): with the chip full of stores the gathered
// loads took ~10 us to come back, the wave's counted vmcnt waits — one counter for loads and stores on gfx950 — turned that into a
// stall per group of four knots, and the kernel ran at 5.4 us per knot where the same step takes 0.8 us in k_forward (r05 traces,
// Cartpole at B = 1 048 576: 2.2 ms per launch against 1.05 ms in the first batch step, where every trajectory accepts the same
// step size and the gather happens to be coalesced).
template <class M>
__device__ __forceinline__ bool accept_lane(const KArgs& a, int& b, int& s) {
  const DevProblem& P = a.P;
  bool inrange;
  if (a.compact) {
    const int cnt = a.acount[a.step & 1];
    if ((int)blockIdx.x * 64 >= cnt) return false;  // wave-uniform
    const int li = blockIdx.x * 64 + threadIdx.x;
    inrange = li < cnt;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + (inrange ? li : cnt - 1)];
  } else {
    b = blockIdx.x * 64 + threadIdx.x;
    inrange = b < P.B;
    if (!inrange) b = P.B - 1;
  }
  s = inrange ? a.acc[b] : 0;
  return __ballot(s != 0) != 0;
}
template <class M>
__global__ void __launch_bounds__(64) k_accept_gather_u(KArgs a) {
  constexpr int m = M::m;
  int b, s;
  if (!accept_lane<M>(a, b, s)) return;
  const int Lu = (a.P.N - 1) * m;
  const double* su = U_SLOT_PTR(a, b, s);
  double* du = U_SLOT_PTR(a, b, 0);
  if (s == 0) return;
  int e = 0;
  for (; e + 16 <= Lu; e += 16) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = EL(su, e + i);
#pragma unroll
    for (int i = 0; i < 16; ++i) EL(du, e + i) = v[i];
  }
  for (; e < Lu; ++e) EL(du, e) = EL(su, e);
}
template <class M, int FIXED_INTEG>
__global__ void __launch_bounds__(64) k_accept_roll(KArgs a) {
  constexpr int n = M::n, m = M::m;
  const DevProblem& P = a.P;
  int b, s;
  if (!accept_lane<M>(a, b, s)) return;
  const int tile = b >> 6, lane = b & 63;
  const int N = P.N;
  // Lanes that accepted nothing roll their own nominal alongside (EXEC stays full) and store into the spare tile behind the batch
  // (Xs is allocated one tile longer): the stores sit in straight-line code, no per-lane branch around them
  const bool st = s != 0;
  const int dtile = st ? tile : P.Bp / 64;
  double* dX = a.Xs + ((size_t)dtile * (size_t)(N * n)) * 64 + lane;
  const double* pU = TILE_PTR(a.Us, (N - 1) * m);  // the accepted controls, already on the nominal (k_accept_gather_u): coalesced rows
  const double* px0 = TILE_PTR(a.x0, n);
  double mp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) mp[i] = in_vgpr(P.mp[i]);
  const int integrator = P.integrator;
  double x[n], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(px0, i);
  // controls and time steps are fetched D knots at a time, the next group while the current one is stepped through
  constexpr int D = 4;
  double ua[D][m], ub[D][m], ha[D], hb[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int kk = d < N - 1 ? d : N - 2;
#pragma unroll
    for (int j = 0; j < m; ++j) ua[d][j] = EL(pU, kk * m + j);
    ha[d] = P.dt[kk];
  }
  for (int k0 = 0; k0 < N - 1; k0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kk = k0 + D + d < N - 1 ? k0 + D + d : N - 2;  // past the horizon: re-read the last knot (never used)
#pragma unroll
      for (int j = 0; j < m; ++j) ub[d][j] = EL(pU, kk * m + j);
      hb[d] = P.dt[kk];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int k = k0 + d;
      if (k >= N - 1) break;  // wave-uniform
#pragma unroll
      for (int i = 0; i < n; ++i) EL(dX, k * n + i) = x[i];
      model_step<M, double, FIXED_INTEG>(mp, integrator, k, x, ua[d], ha[d], xn);
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = xn[i];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int j = 0; j < m; ++j) ua[d][j] = ub[d][j];
      ha[d] = hb[d];
    }
  }
#pragma unroll
  for (int i = 0; i < n; ++i) EL(dX, (N - 1) * n + i) = x[i];
  if (st) a.acc[b] = 0;  // settled: every later reader finds the step on the nominal
}

// gradient metric of the UNCHANGED nominal controls (zero step / failed line search): mean_k max_j |d_kj| / (|u_kj| + 1)
template <class M>
__device__ __forceinline__ double nominal_gradient(const KArgs& a, int tile, int lane, int b) {
  constexpr int m = M::m, ne = M::ne, RSK = Gains<M>::RSK;
  const int N = a.P.N;
  const double* Uc = U_SLOT_PTR(a, b, 0);
  const double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
  double gs = 0.0;
  for (int k = 0; k < N - 1; ++k) {
    double gk = 0.0;
#pragma unroll
    for (int j = 0; j < m; ++j) gk = fmax(gk, fabs(pK[(size_t)k * RSK + j * (ne + 1) + ne]) * rcp_fast(fabs(EL(Uc, k * m + j)) + 1.0));
    gs += gk;
  }
  return gs / (N - 1);
}

template <class M> struct LsRepack { static constexpr bool value = !M::accept_write_through; };  // the models that repack (and have the second block)

// End of a forward pass, one lane per trajectory (q == 0): failed-search regularisation, the solver state machine (rows S3, S4)
// and — with active-list compaction — the settling of accepted steps of trajectories that leave the plain iteration path.
template <class M>
__device__ __forceinline__ void forward_finish(const KArgs& a, int tile, int lane, int b, int hw, int q, int t, int TW, bool act, bool bpfail,
                                               bool zero_step, int accepted, int acc, int accpos, double Jprev, double Jnew, double grad) {
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  // one lane per trajectory finishes the iteration; `settle`: the trajectory leaves the plain next-iteration path (it is done, or
  // its inner solve ended and the AL outer update takes over) while its accepted step still only exists as a candidate
  bool settle = false;
  if (q == 0 && act) {
    double rho = a.rho[b], drho = a.drho[b];
    if (!bpfail) {
      if (zero_step) grad = nominal_gradient<M>(a, tile, lane, b);
      else if (accepted < 0) {  // line search failed: gradient metric on the unchanged nominal controls, regularise harder
        grad = nominal_gradient<M>(a, tile, lane, b);
        reg_increase(o, rho, drho);
        rho += o.bp_reg_fp;
      }
    }
    a.ls_index[b] = accepted;
    a.acc[b] = acc;
    if (acc) a.accp[b] = accpos;  // candidate block and hardware lane that hold the accepted candidate (slot_ptr)
    if (!a.control) {  // phase API: report and leave the state machine alone
      a.Jout[b] = Jnew;
      a.rho[b] = rho; a.drho[b] = drho;
      if (accepted >= 0) a.J[b] = Jnew;
    } else {
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
      a.rho[b] = rho; a.drho[b] = drho;
      if (!inner_done) {
        atomicAdd(&a.counter[a.step], 1);  // (with compaction k_compact rebuilds the list of active trajectories after this step)
      } else {
        settle = a.compact && M::accept_write_through && acc != 0 && a.store_x;  // (models without write-through run k_accept after every forward pass;
                                                                                // without stored states k_accept_roll follows this launch)
        if (!a.al_mode) { a.status[b] = st; a.active[b] = 0; }
        else {  // AL outer update: whole-trajectory passes, run knot-parallel by the k_outer_* kernels
          a.ost[b] = st; a.oflag[b] = 1;
          a.olist[(size_t)(a.step & 1) * P.Bp + atomicAdd(&a.ocount[a.step & 1], 1)] = b;  // order is irrelevant: the outer update is per trajectory
        }
      }
    }
  }
  // Active-list compaction: the lanes of this wave hold OTHER trajectories in the next step, so an accepted step that is not
  // going to be written through by the next expansion is settled now — the whole wave copies the candidate (its own block,
  // lane sl) onto the trajectory's nominal slot, 64 elements per pass, and the slot index is cleared.
  if (a.compact) {
    unsigned long long todo = __ballot(settle);
    const int Lx = P.N * M::n, Lu = (P.N - 1) * M::m;
    if (todo) __threadfence();  // the candidate was stored by other lanes of this wave: make those stores visible to the loads below
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int bs = __shfl(b, src), pos = __shfl(accpos, src);  // where the accepted candidate sits: block * 64 + hardware lane
      const double* cx = a.Xc + ((size_t)(pos >> 6) * (size_t)Lx) * 64 + (pos & 63);
      const double* cu = a.Uc + ((size_t)(pos >> 6) * (size_t)Lu) * 64 + (pos & 63);
      double* nx = a.Xs + ((size_t)(bs >> 6) * (size_t)Lx) * 64 + (bs & 63);
      double* nu = a.Us + ((size_t)(bs >> 6) * (size_t)Lu) * 64 + (bs & 63);
      for (int e = hw; e < Lx; e += 64) EL(nx, e) = EL(cx, e);
      for (int e = hw; e < Lu; e += 64) EL(nu, e) = EL(cu, e);
      if (hw == src) a.acc[bs] = 0;
    }
  }
}

// Forward pass of one iLQR iteration.  grid = Bp / TW waves.  Line search: round r evaluates step sizes r*CW .. r*CW+CW-1
// concurrently (one per lane group) and takes the FIRST accepted one — identical to sequential backtracking
// (SURVEY.md row S2); then — when a.control — the per-trajectory solver state machine runs: convergence test (row S3)
// and the hand-over to the AL outer update (row S4, k_outer_*).
template <class M, int MODE>
__global__ void __launch_bounds__(64, TO_FWD_WAVES) k_forward(KArgs a) {
  extern __shared__ double kbuf[];  // M::lds_gains: two buffers of gains_lds_doubles(TW)
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  const int hw = threadIdx.x;
  const int CW = a.CW, TW = a.TW;
  const int q = hw / TW, t = hw - q * TW;      // lanes with q >= CW (64 is not a multiple of TW) ride along without a candidate
  const int b0 = blockIdx.x * TW;
  // trajectory of this lane: position b0 + t of the batch, or — with active-list compaction — of this step's list
  int b, inrange;
  if (a.ls_phase == 2) {  // launch B of the two-launch line search (common.h): the trajectories launch A flagged, in index order
    const int cnt = a.pcount[0];
    if (b0 >= cnt) return;  // wave-uniform
    inrange = (b0 + t) < cnt;
    b = a.plist[inrange ? b0 + t : cnt - 1];
  } else if (a.compact) {
    const int cnt = a.acount[a.step & 1];
    if (b0 >= cnt) return;  // wave-uniform
    inrange = (b0 + t) < cnt;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + (inrange ? b0 + t : cnt - 1)];
  } else {
    inrange = (b0 + t) < P.B;
    b = (b0 + t) < P.Bp ? b0 + t : P.Bp - 1;  // clamped: the last wave may reach past the batch
  }
  const int tile = b >> 6, lane = b & 63;
  // gfx950 issues FP64 VALU ~1.3x slower whenever EXEC is not all ones (tools/fp64_issue_probe.hip), so lanes that have
  // nothing to do are NOT masked off: they roll out their own (valid) trajectory as well and only their stores are
  // predicated.  The wave leaves only when no lane needs anything.
  const bool act = inrange && a.active[b] != 0;
  if (__ballot(act) == 0) return;
  const bool bpfail = act && a.bpfail[b] != 0;
  const int total = o.iterations_linesearch;
  const double Jprev = a.J[b];
  const double dV0 = a.dV[b], dV1 = a.dV[(size_t)P.Bp + b];
  // stationary point (predicted improvement ~ rounding noise): take the zero step, dJ = 0 => converged
  const bool zero_step = act && !bpfail && (-(dV0 + dV1) <= 1e-12 * (1.0 + fabs(Jprev)));
  bool need = act && !bpfail && !zero_step;
  int accepted = zero_step ? 0 : -1, acc = 0, accpos = 0;
  double Jnew = Jprev, grad = 0.0;
  const double f = o.line_search_decrease_factor;
  const int kbuf_len = M::lds_gains ? gains_lds_doubles<M>(TW) : 0;
  if constexpr (!LsRepack<M>::value) {
    // static lane map in every round (the models whose accepted steps are written through by the next expansion: their search
    // rarely leaves the first round, and the general loop below cost the Cartpole kernel 2.3 us per launch)
    const int krow = t * Gains<M>::RSK;
    const int wblk = a.blk0 + (int)blockIdx.x;  // this wave's candidate block
    for (int c0 = (a.ls_phase == 2 ? a.ls_c0 : 0); c0 < total; c0 += CW) {
      if (__ballot(need) == 0) break;
      const double alpha = ls_alpha(f, c0 + q, total);
      const bool cand = need && q < CW && (c0 + q) < total;
      double J, gm;
      bool ok;
      forward_candidate<M, MODE>(a, tile, lane, b, cand, alpha, wblk, kbuf, kbuf_len, krow, b0, TW, hw, J, gm, ok);
      bool accept = false;
      if (cand && ok) {
        const double expected = -alpha * (dV0 + alpha * dV1);
        const double z = (expected > 0.0) ? (Jprev - J) / expected : -1.0;
        accept = z >= o.line_search_lower_bound && z <= o.line_search_upper_bound;
      }
      const unsigned long long am = __ballot(accept);
      int qs = -1;  // first accepted candidate of this lane's trajectory (bits qq*TW + t)
      for (int qq = CW - 1; qq >= 0; --qq) qs = ((am >> (qq * TW + t)) & 1ull) ? qq : qs;
      const int src = (qs >= 0 ? qs : 0) * TW + t;
      const double Js = __shfl(J, src), gs = __shfl(gm, src);
      if (need && qs >= 0) { Jnew = Js; grad = gs; accepted = c0 + qs; acc = qs + 1; accpos = wblk * 64 + src; need = false; }
      if (a.ls_phase == 1) break;  // launch A: one round
    }
    // launch A: a trajectory that has accepted nothing yet (and still has step sizes to try) goes on in launch B and is finished there
    const bool later = a.ls_phase == 1 && need && CW < total;
    if (later && q == 0) a.pending[b] = 1;
    forward_finish<M>(a, tile, lane, b, hw, q, t, TW, act && !later, bpfail, zero_step, accepted, acc, accpos, Jprev, Jnew, grad);
    return;
  }
  const unsigned long long tmask = TW >= 64 ? ~0ull : (1ull << TW) - 1ull;  // lanes 0 .. TW-1 (q = 0): one per trajectory of the wave
  for (int c0 = 0; c0 < total;) {
    const unsigned long long nm = __ballot(need) & tmask;
    if (nm == 0) break;
    const LsRound R = ls_round(nm, c0, total, CW, TW, q, t, hw, a.repack_block0 != 0);
    // the trajectory this lane works for in this round (its own one under the static map)
    const int bR = __shfl(b, R.ts);
    const double JprevR = __shfl(Jprev, R.ts), dV0R = __shfl(dV0, R.ts), dV1R = __shfl(dV1, R.ts);
    const double alpha = ls_alpha(f, c0 + R.qc, total);
    const bool cand = R.has && R.qc < R.cw && (c0 + R.qc) < total;
    const int wblock = R.repacked ? a.repack_block0 + (int)blockIdx.x : (int)blockIdx.x;
    double J, gm;
    bool ok;
    forward_candidate<M, MODE>(a, bR >> 6, bR & 63, bR, cand, alpha, wblock, kbuf, kbuf_len, R.tr * Gains<M>::RSK, b0, R.tw, hw, J, gm, ok);
    bool accept = false;
    if (cand && ok) {
      const double expected = -alpha * (dV0R + alpha * dV1R);
      const double z = (expected > 0.0) ? (JprevR - J) / expected : -1.0;
      accept = z >= o.line_search_lower_bound && z <= o.line_search_upper_bound;
    }
    const unsigned long long am = __ballot(accept);
    int qs = -1;  // first accepted candidate of this lane's own trajectory (bits qq*tw + j)
    for (int qq = R.cw - 1; qq >= 0; --qq) qs = ((am >> ((qq * R.tw + R.j) & 63)) & 1ull) ? qq : qs;
    const int src = ((qs >= 0 ? qs : 0) * R.tw + R.j) & 63;
    const double Js = __shfl(J, src), gs = __shfl(gm, src);
    if (need && qs >= 0) { Jnew = Js; grad = gs; accepted = c0 + qs; acc = qs + 1; accpos = wblock * 64 + src; need = false; }
    c0 += R.cw;
  }
  forward_finish<M>(a, tile, lane, b, hw, q, t, TW, act, bpfail, zero_step, accepted, acc, accpos, Jprev, Jnew, grad);
}

// ------------------------------------------------------------------------------------------------ two-wave forward pass
// The Quadrotor-class rollout is a latency chain: ~1 000 instructions per knot on ONE wave, and for most batch steps of a solve
// the chip holds fewer forward waves than SIMDs (the batch has drained), so the step takes exactly as long as that chain.  Only
// about two thirds of it IS the recurrence (state difference, gains, RK stages); the rest — candidate stores, stage cost, AL
// terms, admissibility limits, gradient metric — merely consumes (x_k, u_k).  k_forward2 gives a workgroup TWO waves with the
// same lane -> (candidate, trajectory) map:
//   wave 0, the roller: nominal + gains (LDS-DMA) -> δx, u_k, RK step; publishes (x_k, u_k, d_k) in an LDS ring slot per knot;
//   wave 1, the accountant: one knot behind, stores the candidate, accumulates J / gradient metric / limits from the slot, and
//           after the rollout runs the acceptance test and the state machine exactly as k_forward does.
// One workgroup barrier per knot orders the two-slot ring (the roller refills slot k&1 two knots later, i.e. after the barrier
// the accountant reaches only when it has read it).  Every expression and every summation order is k_forward's, and the forward
// translation units are compiled with -ffp-contract=on (build.py: fused multiply-adds formed per source expression, not per
// inlining context), so the two kernels are BIT-IDENTICAL — they have to be: the solve loop picks between them per batch step
// from the number of active trajectories, and a trajectory's result must not depend on the rest of its batch
// (tests/test_gpu_parity.py::test_two_wave_forward_pass asserts equality).  The early exit of a wave whose candidates have all left the admissible box is not
// replicated (it only saves time in a case that is rejected anyway).
// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the accountant's candidate stores and dual
// prefetches (vmcnt) at every knot.  Both waves sit on one CU; nothing but LDS is exchanged between them.
#define FWD2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <class M>
struct Fwd2Ring {
  static constexpr int n = M::n, m = M::m;
  static constexpr int NV = n + 2 * m, PAIRS = (NV + 1) / 2, SLOT = PAIRS * 128;  // doubles per slot: 64 lanes x PAIRS 16-byte pieces
};

template <class M, int MODE>
__device__ __forceinline__ void fwd2_roll(const KArgs& a, int tile, int lane, int b, double alpha, double* kbuf, int kbuf_len, int krow,
                                          int TW, int hw, double* ring) {
  constexpr int n = M::n, m = M::m, ne = M::ne, RSK = Gains<M>::RSK;
  constexpr bool SIMPLE = (MODE & 1) != 0;
  constexpr bool KLDS = M::lds_gains;
  using R = Fwd2Ring<M>;
  const DevProblem& P = a.P;
  const int N = P.N;
  const double* Xc = TILE_PTR(a.Xs, N * n);
  const double* Uc = TILE_PTR(a.Us, (N - 1) * m);
  const double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
  const double* px0 = TILE_PTR(a.x0, n);
  double mp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) mp[i] = in_vgpr(P.mp[i]);
  const int integrator = P.integrator;
  const double h0 = SIMPLE ? P.dt[0] : 0.0;
  double xb[n];
#pragma unroll
  for (int i = 0; i < n; ++i) xb[i] = EL(px0, i);
  if constexpr (KLDS) stage_gains<M>(a.Kt, b, TW, 0, N, kbuf, hw);
  FwdKnot<M, !KLDS> nxt;
  nxt.load(Xc, Uc, pK);
  const double *pXn = Xc + n * 64, *pUn = Uc + m * 64, *pKn = pK + RSK;
  for (int k = 0; k < N - 1; ++k) {
    if constexpr (KLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const FwdKnot<M, !KLDS> cur = nxt;
    const double* kcur = kbuf + (size_t)(k & 1) * kbuf_len + krow;
    if constexpr (!KLDS) {  // gains row in registers: plain loads, the compiler waits for exactly what it uses — issue them first
      if (k + 1 < N - 1) nxt.load(pXn, pUn, pKn);
      pXn += n * 64; pUn += m * 64; pKn += RSK;
    }
    double vals[2 * R::PAIRS];  // [x_k | u_k | d_k | pad]
#pragma unroll
    for (int i = 0; i < n; ++i) vals[i] = xb[i];
    double dx[ne], ub[m], xn[n];
    state_diff<M>(xb, cur.x, dx);
    {
      // the roller has registers to spare (the accountant holds the cost / AL state): the whole gains block of the knot is
      // requested in one go — one LDS round trip per knot instead of one per row (C3 forward phase 604.8 -> 599.5 us per step)
      double kr[m][ne + 1];
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int i = 0; i <= ne; ++i) kr[j][i] = KLDS ? kcur[j * (ne + 1) + i] : cur.kd[KLDS ? 0 : j * (ne + 1) + i];
      if constexpr (KLDS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < m; ++j) {
        const double dj = kr[j][ne];
        double du = dj * alpha;
#pragma unroll
        for (int i = 0; i < ne; ++i) du += kr[j][i] * dx[i];
        ub[j] = cur.u[j] + du;
        vals[n + j] = ub[j];
        vals[n + m + j] = dj;
      }
    }
    if (2 * R::PAIRS > R::NV) vals[2 * R::PAIRS - 1] = 0.0;
    double2* slot = (double2*)(ring + (size_t)(k & 1) * R::SLOT) + hw;
#pragma unroll
    for (int pr = 0; pr < R::PAIRS; ++pr) slot[pr * 64] = make_double2(vals[2 * pr], vals[2 * pr + 1]);
    // LDS-staged gains: the next knot's DMA / loads go out only NOW.  hipcc orders every LDS write of a wave behind its
    // outstanding LDS-DMAs (it cannot tell the ring from the gains buffers) and a vmcnt wait drains the nominal loads with
    // them — issued at the top of the knot, as in k_forward, the ring writes above stalled for a full memory round trip per
    // knot.  They land during the RK stages.
    if constexpr (KLDS) {
      if (k + 1 < N - 1) {
        stage_gains<M>(a.Kt, b, TW, k + 1, N, kbuf + (size_t)((k + 1) & 1) * kbuf_len, hw);
        nxt.load(pXn, pUn, pKn);
      }
      pXn += n * 64; pUn += m * 64; pKn += RSK;
    }
    const double h = SIMPLE ? h0 : P.dt[k];
    model_step<M, double, (MODE & 4) ? INTEG_RK4 : -1>(mp, integrator, k, xb, ub, h, xn);
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = xn[i];
    FWD2_BARRIER();
  }
  {  // terminal state
    double2* slot = (double2*)(ring + (size_t)((N - 1) & 1) * R::SLOT) + hw;
#pragma unroll
    for (int pr = 0; pr < (n + 1) / 2; ++pr) slot[pr * 64] = make_double2(xb[2 * pr], (2 * pr + 1 < n) ? xb[2 * pr + 1] : 0.0);
    FWD2_BARRIER();
  }
  if constexpr (KLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class M, int MODE>
__device__ __forceinline__ void fwd2_account(const KArgs& a, int tile, int lane, int b, bool live, int wblock, double* ctab, int hw, const double* ring,
                                             double& J_out, double& g_out, bool& ok_out) {
  constexpr int n = M::n, m = M::m;
  constexpr bool SIMPLE = (MODE & 1) != 0, CONS = (MODE & 2) != 0, GEN = (MODE & 8) != 0;
  using R = Fwd2Ring<M>;
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  const int N = P.N;
  const size_t cblock = live ? (size_t)wblock : (size_t)a.dump_wave;
  double* pXo = a.Xc + (cblock * (size_t)(N * n)) * 64 + hw;
  double* pUo = a.Uc + (cblock * (size_t)((N - 1) * m)) * 64 + hw;
  const double* lam0 = TILE_PTR(a.lam, P.n_duals);
  const double* mu0 = TILE_PTR(a.mu, P.n_cons);
  const bool dt_scaling = P.opts.cost_dt_scaling != 0;
  const double max_x = o.max_state_value, max_u = o.max_control_value;
  typename std::conditional<M::lds_gains, StageCostLds<n, m>, StageCostDiag<n, m>>::type sc;
  const bool has_gl = P.gl != nullptr;  // wave-uniform (a kernel argument): the branches on it are scalar
  const double* gl0 = TILE_PTR(P.gl, P.n_costs * (n + m));
  const double* cp0 = TILE_PTR(P.cp, P.n_cp);  // per-trajectory constraint parameters (general variants)
  const int ci0 = P.cost_index[0];
  double h0 = 0.0;
  if constexpr (SIMPLE) {
    if constexpr (M::lds_gains) { sc.load(P.costs[P.cost_index[0]], ctab, hw); WAVE_SYNC(); }
    else sc.load(P.costs[P.cost_index[0]]);
    h0 = P.dt[0];
  }
  ConStage<n, m, (MODE & 16) != 0> cs0, cs1;
  int ncs = 0, uncached = 0;
  cs0.ci = -1; cs1.ci = -1;
  if constexpr (CONS) {
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (K.fast == 2 && K.k1 == 0 && K.k2 >= N - 2 && K.p <= m + 1 && ncs < 2) {
        if (ncs == 0) cs0.load(K, ci, lam0, mu0); else cs1.load(K, ci, lam0, mu0);
        ++ncs;
      } else if (K.k1 <= N - 2) ++uncached;
    }
  }
  const bool all_cached = CONS && uncached == 0;
  if (ncs > 0) cs0.prefetch(0);
  if (ncs > 1) cs1.prefetch(0);
  double J = 0.0, gsum = 0.0;
  bool ok = true;
  for (int k = 0; k < N - 1; ++k) {
    FWD2_BARRIER();  // the roller has published knot k
    if (ncs > 0) cs0.advance();
    if (ncs > 1) cs1.advance();
    if (k + 1 < N - 1) {
      if (ncs > 0) cs0.prefetch(k + 1);
      if (ncs > 1) cs1.prefetch(k + 1);
    }
    double vals[2 * R::PAIRS];
    const double2* slot = (const double2*)(ring + (size_t)(k & 1) * R::SLOT) + hw;
#pragma unroll
    for (int pr = 0; pr < R::PAIRS; ++pr) { const double2 v = slot[pr * 64]; vals[2 * pr] = v.x; vals[2 * pr + 1] = v.y; }
    const double* xb = vals;
    const double* ub = vals + n;
    const double* dk = vals + n + m;
#pragma unroll
    for (int i = 0; i < n; ++i) EL(pXo, i) = xb[i];
    pXo += n * 64;
#pragma unroll
    for (int j = 0; j < m; ++j) EL(pUo, j) = ub[j];
    pUo += m * 64;
    double gk = 0.0;
#pragma unroll
    for (int j = 0; j < m; ++j) gk = fmax(gk, fabs(dk[j]) * rcp_fast(fabs(ub[j]) + 1.0));
    gsum += gk;
    const double h = SIMPLE ? h0 : P.dt[k];
    double Jk = SIMPLE ? sc.eval(xb, ub) : cost_eval<n, m, GEN>(P.costs[P.cost_index[k]], xb, ub);
    if constexpr (GEN) {  // per-trajectory q, r: in the general variants only (launch_forward picks one when DevProblem::gl is set), so that
      if (has_gl) Jk += goal_lin_cost<n, m>(gl0, SIMPLE ? ci0 : (int)P.cost_index[k], xb, ub);  // the default kernels keep their loops as they were
    }
    if (dt_scaling) Jk *= h;
    if constexpr (CONS) {
      if (all_cached) {
        double Ja = 0.0;
        if (ncs > 0) Ja += cs0.term(ub);
        if (ncs > 1) Ja += cs1.term(ub);
        Jk += Ja;
      } else Jk += knot_al_cached<M, GEN>(P, k, xb, ub, lam0, mu0, ncs, cs0, cs1, cp0);
    }
    J += Jk;
    // admissibility (k_forward checks x_{k+1} and u_k in iteration k: the same set of values, seen one knot later here)
    double mx = 0.0, mu_ = 0.0;
    if (k > 0) {
#pragma unroll
      for (int i = 0; i < n; ++i) { const double v = fabs(xb[i]); if (!(v <= mx)) mx = v; }
    }
#pragma unroll
    for (int j = 0; j < m; ++j) { const double v = fabs(ub[j]); if (!(v <= mu_)) mu_ = v; }
    if (!(mx <= max_x) || !(mu_ <= max_u)) ok = false;
  }
  FWD2_BARRIER();  // terminal state
  {
    double xb[n + 1];
    const double2* slot = (const double2*)(ring + (size_t)((N - 1) & 1) * R::SLOT) + hw;
#pragma unroll
    for (int pr = 0; pr < (n + 1) / 2; ++pr) { const double2 v = slot[pr * 64]; xb[2 * pr] = v.x; if (2 * pr + 1 < n + 1) xb[2 * pr + 1] = v.y; }
    double mx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) { const double v = fabs(xb[i]); if (!(v <= mx)) mx = v; }
    if (!(mx <= max_x)) ok = false;
#pragma unroll
    for (int i = 0; i < n; ++i) EL(pXo, i) = xb[i];
    double u0[m];
#pragma unroll
    for (int j = 0; j < m; ++j) u0[j] = 0.0;
    J += knot_cost<M, GEN>(P, N - 1, xb, u0, lam0, mu0, true, gl0, cp0);
  }
  J_out = J; g_out = gsum / (N - 1); ok_out = ok;
}

#ifndef TO_FWD2_WAVES
#define TO_FWD2_WAVES 1
#endif
// dynamic LDS: [gains buffer 0 | gains buffer 1 | stage-cost table | ring slot 0 | ring slot 1 | need mask]
template <class M>
__host__ __device__ inline size_t fwd2_lds_doubles(int TW) {
  return (M::lds_gains ? 2 * (size_t)gains_lds_doubles<M>(TW) + StageCostLds<M::n, M::m>::size : 0) + 2 * (size_t)Fwd2Ring<M>::SLOT + 2;
}
template <class M, int MODE>
__global__ void __launch_bounds__(128, TO_FWD2_WAVES) k_forward2(KArgs a) {
  extern __shared__ double kbuf[];
  const DevProblem& P = a.P;
  const to_solver_opts& o = P.opts;
  const int hw = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0: roller, 1: accountant (wave-uniform, a scalar branch)
  const int CW = a.CW, TW = a.TW;
  const int q = hw / TW, t = hw - q * TW;
  const int b0 = blockIdx.x * TW;
  int b, inrange;
  if (a.compact) {
    const int cnt = a.acount[a.step & 1];
    if (b0 >= cnt) return;  // both waves of the workgroup
    inrange = (b0 + t) < cnt;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + (inrange ? b0 + t : cnt - 1)];
  } else {
    inrange = (b0 + t) < P.B;
    b = (b0 + t) < P.Bp ? b0 + t : P.Bp - 1;
  }
  const int tile = b >> 6, lane = b & 63;
  const bool act = inrange && a.active[b] != 0;
  if (__ballot(act) == 0) return;  // nothing of this workgroup's state changes before its last barrier: both waves decide alike
  const bool bpfail = act && a.bpfail[b] != 0;
  const int total = o.iterations_linesearch;
  const double Jprev = a.J[b];
  const double dV0 = a.dV[b], dV1 = a.dV[(size_t)P.Bp + b];
  const bool zero_step = act && !bpfail && (-(dV0 + dV1) <= 1e-12 * (1.0 + fabs(Jprev)));
  bool need = act && !bpfail && !zero_step;
  int accepted = zero_step ? 0 : -1, acc = 0, accpos = 0;
  double Jnew = Jprev, grad = 0.0;
  const double f = o.line_search_decrease_factor;
  const int kbuf_len = M::lds_gains ? gains_lds_doubles<M>(TW) : 0;
  const unsigned long long tmask = TW >= 64 ? ~0ull : (1ull << TW) - 1ull;
  double* ctab = kbuf + 2 * (size_t)kbuf_len;
  double* ring = ctab + (M::lds_gains ? StageCostLds<M::n, M::m>::size : 0);
  unsigned long long* needmask = (unsigned long long*)(ring + 2 * (size_t)Fwd2Ring<M>::SLOT);
  if constexpr (!LsRepack<M>::value) {  // static lane map in every round (k_forward)
    const int krow = t * Gains<M>::RSK;
    unsigned long long nm = __ballot(need);  // identical in both waves here; afterwards the accountant's word
    for (int c0 = 0; c0 < total; c0 += CW) {
      if (nm == 0) break;
      const double alpha = ls_alpha(f, c0 + q, total);
      if (role == 0) {
        fwd2_roll<M, MODE>(a, tile, lane, b, alpha, kbuf, kbuf_len, krow, TW, hw, ring);
      } else {
        const bool cand = need && q < CW && (c0 + q) < total;
        double J, gm;
        bool ok;
        fwd2_account<M, MODE>(a, tile, lane, b, cand, (int)blockIdx.x, ctab, hw, ring, J, gm, ok);
        bool accept = false;
        if (cand && ok) {
          const double expected = -alpha * (dV0 + alpha * dV1);
          const double z = (expected > 0.0) ? (Jprev - J) / expected : -1.0;
          accept = z >= o.line_search_lower_bound && z <= o.line_search_upper_bound;
        }
        const unsigned long long am = __ballot(accept);
        int qs = -1;
        for (int qq = CW - 1; qq >= 0; --qq) qs = ((am >> (qq * TW + t)) & 1ull) ? qq : qs;
        const int src = (qs >= 0 ? qs : 0) * TW + t;
        const double Js = __shfl(J, src), gs = __shfl(gm, src);
        if (need && qs >= 0) { Jnew = Js; grad = gs; accepted = c0 + qs; acc = qs + 1; accpos = (int)blockIdx.x * 64 + src; need = false; }
        const unsigned long long left = __ballot(need);
        if (hw == 0) *needmask = left;
      }
      FWD2_BARRIER();
      {  // (the word is rewritten a whole rollout — N barriers — later)
        const unsigned long long w = *needmask;
        nm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)w);
      }
    }
    if (role == 0) return;
    forward_finish<M>(a, tile, lane, b, hw, q, t, TW, act, bpfail, zero_step, accepted, acc, accpos, Jprev, Jnew, grad);
    return;
  }
  unsigned long long nm = __ballot(need) & tmask;  // identical in both waves here; afterwards the accountant's word
  for (int c0 = 0; c0 < total;) {
    if (nm == 0) break;
    const LsRound R = ls_round(nm, c0, total, CW, TW, q, t, hw, a.repack_block0 != 0);  // both waves: the same map (k_forward)
    const int bR = __shfl(b, R.ts);
    const double alpha = ls_alpha(f, c0 + R.qc, total);
    if (role == 0) {
      fwd2_roll<M, MODE>(a, bR >> 6, bR & 63, bR, alpha, kbuf, kbuf_len, R.tr * Gains<M>::RSK, R.tw, hw, ring);
    } else {
      const double JprevR = __shfl(Jprev, R.ts), dV0R = __shfl(dV0, R.ts), dV1R = __shfl(dV1, R.ts);
      const bool cand = R.has && R.qc < R.cw && (c0 + R.qc) < total;
      const int wblock = R.repacked ? a.repack_block0 + (int)blockIdx.x : (int)blockIdx.x;
      double J, gm;
      bool ok;
      fwd2_account<M, MODE>(a, bR >> 6, bR & 63, bR, cand, wblock, ctab, hw, ring, J, gm, ok);
      bool accept = false;
      if (cand && ok) {
        const double expected = -alpha * (dV0R + alpha * dV1R);
        const double z = (expected > 0.0) ? (JprevR - J) / expected : -1.0;
        accept = z >= o.line_search_lower_bound && z <= o.line_search_upper_bound;
      }
      const unsigned long long am = __ballot(accept);
      int qs = -1;
      for (int qq = R.cw - 1; qq >= 0; --qq) qs = ((am >> ((qq * R.tw + R.j) & 63)) & 1ull) ? qq : qs;
      const int src = ((qs >= 0 ? qs : 0) * R.tw + R.j) & 63;
      const double Js = __shfl(J, src), gs = __shfl(gm, src);
      if (need && qs >= 0) { Jnew = Js; grad = gs; accepted = c0 + qs; acc = qs + 1; accpos = wblock * 64 + src; need = false; }
      const unsigned long long left = __ballot(need) & tmask;
      if (hw == 0) *needmask = left;
    }
    FWD2_BARRIER();
    {  // (the word is rewritten a whole rollout — N barriers — later)
      const unsigned long long w = *needmask;
      nm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)w);
    }
    c0 += R.cw;
  }
  if (role == 0) return;
  forward_finish<M>(a, tile, lane, b, hw, q, t, TW, act, bpfail, zero_step, accepted, acc, accpos, Jprev, Jnew, grad);
}

}  // namespace to
