// k_expand.h — dynamics / cost / constraint expansion.
#pragma once
#include "common.h"
#include "k_backward.h"  // Tm<M>, compact_row: the layouts the backward pass reads

namespace to {

// ------------------------------------------------------------------------------------------------ expansion
// grid (ceil(B/G), N): lane (g, j) of a wave = trajectory gtile*G+g, direction j.  It produces column j of everything
// the backward pass needs at knot k:
//   [Ā B̄][:,j] = G(x_{k+1})ᵀ · ∂(RK step)/∂z · v_j            (forward-mode dual through all RK stages)
//   H[:,j]      = projected Hessian-vector product of (cost + AL) with v_j,   g[j] = projected gradient component
// with v_j = [G(x_k) e_j; 0] (j<ne) or [0; e_{j-ne}].
// VAR bit0: dense QuadraticCost possible; bit1: constraints present; bit2: non-selector constraints possible.  Code the
// problem cannot reach is compiled out: the all-purpose Quadrotor kernel needed 256 VGPRs + 140 AGPRs + 480 B of scratch.
// one knot of the expansion for lane (g, j): x = x_k, u = u_k (zeros at the terminal knot), x1 = x_{k+1}
// LAY: where the columns go.  0: column layout (cooperative backward pass); 1: tangent-matrix layout, full cost block;
// 2: tangent-matrix layout, compact cost block; 3: lane layout (one lane per trajectory backward pass) — all in k_backward.h.
// PACK (rigid body — quaternion, MRP or Rodrigues attitude —, compact cost block): six of the sixteen columns of [Ā B̄] are CONSTANTS — the dynamics do
// not read the position, and the (world-frame) velocity only as ṙ = v, so ∂x⁺/∂r = [I; 0; 0; 0] and ∂x⁺/∂v = [h I; 0; I; 0] for every
// Runge-Kutta scheme — and live in Mt from k_expand_const_columns on.  A wave then holds SIX trajectories x the TEN differentiated
// columns (attitude, ω, controls) instead of four x sixteen; the lanes of the first six of them also deliver the cost entries of one
// constant column each (jc >= 0): the cost (+AL) Hessian-vector product is taken with the direction of the own column PLUS e_jc — the
// compact cost block is block-diagonal with r and v on 1 x 1 blocks (KArgs::h_compact), so entry jc of the product is exactly the diagonal
// entry of column jc and every other entry is what the own direction alone gives — and the gradient does not depend on the direction.
template <class M> struct ExpandPack { static constexpr bool ok = M::lie && (M::att == ATT_QUAT || M::att == ATT_MRP || M::att == ATT_RP) && M::ne == 12 && M::m == 4 && Tm<M>::fits; };
template <class M, int FIXED_INTEG, int VAR, int LAY, bool PACK = false>
__device__ __forceinline__ void expand_knot(const KArgs& a, int gtile, int lane, int tile, int lane64, int b, int j, int k, bool valid,
                                            const double* x, const double* u, const double* x1, const ConExp<M::m>& ce0,
                                            const ConExp<M::m>& ce1, bool table_cons, int jc = -1, bool valid_c = false) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m, nc = ne + m;
  static_assert(!PACK || (LAY == 2 && ExpandPack<M>::ok), "packed expansion: compact tangent-matrix layout of the rigid body");
  constexpr int NEP = Tm<M>::NEP, RS = Tm<M>::RS, NR = Tm<M>::NR;
  const int ct = j < ne ? j : NEP + (j - ne);  // tangent index of this lane's column
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
  const bool on0 = (VAR & 2) != 0 && ce0.at(k), on1 = (VAR & 2) != 0 && ce1.at(k);  // wave-uniform (descriptors cached by k_expand)
  if ((VAR & 2) != 0) {
    const double* lam0 = a.lam + ((size_t)tile * (size_t)P.n_duals) * 64 + lane64;
    const double* mu0 = a.mu + ((size_t)tile * (size_t)P.n_cons) * 64 + lane64;
    if (on0) {
      const double* lam = lam0 + (size_t)(ce0.dual_off + (long long)(k - ce0.k1) * ce0.p) * 64;
      mu0r = EL(mu0, ce0.ci);
#pragma unroll
      for (int r = 0; r < LR; ++r) l0[r] = (r < ce0.p) ? lam[r * 64] : 0.0;
    }
    if (on1) {
      const double* lam = lam0 + (size_t)(ce1.dual_off + (long long)(k - ce1.k1) * ce1.p) * 64;
      mu1r = EL(mu0, ce1.ci);
#pragma unroll
      for (int r = 0; r < LR; ++r) l1[r] = (r < ce1.p) ? lam[r * 64] : 0.0;
    }
  }
  // ---- dynamics column
  if (!terminal) {
    Dual xd[n], ud[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xd[i] = Dual(x[i], v[i]);
#pragma unroll
    for (int i = 0; i < m; ++i) ud[i] = Dual(u[i], v[n + i]);
    model_step<M, Dual, FIXED_INTEG>(P.mp, P.integrator, k, xd, ud, P.dt[k], xn);
    double t[n], col[ne];
#pragma unroll
    for (int i = 0; i < n; ++i) t[i] = xn[i].d;
    errstate_invmul<M>(x1, t, col);
    if constexpr (LAY == 0) {
      double* Mc = COL_PTR(a.Mc, (N - 1) * ne);
      if (valid) {
#pragma unroll
        for (int i = 0; i < ne; ++i) EL(Mc, k * ne + i) = col[i];
      }
    } else if constexpr (LAY == 3) {
      double* Ml = a.Mc + (((size_t)tile * (size_t)(N - 1) + k) * (ne * nc) + j) * 64 + lane64;
      if (valid) {
#pragma unroll
        for (int i = 0; i < ne; ++i) EL(Ml, i * nc) = col[i];
      }
    } else {
      double* Mt = a.Mt + (((size_t)b * (N - 1) + k) * RS) * 64 + ct;
      if (valid) {
#pragma unroll
        for (int i = 0; i < ne; ++i) Mt[(i / 4) * 64 + (i % 4) * 16] = col[i];
      }
    }
  }
  // ---- cost (+AL) gradient and Hessian-vector product on the full state
  double gr[nz], y[nz];
  const int sc = (n == 13 && jc >= 3) ? jc + 1 : jc;  // state index of the constant column this lane serves (quaternion states: error index 6..8 = state 7..9); PACK only
  if constexpr (PACK) {  // (from here on v carries e_sc as well; the own column's dot products below skip that entry)
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (jc >= 0 && i == sc) ? 1.0 : v[i];
  }
  cost_grad_hvp<n, m, (VAR & 1) != 0>(P.costs[P.cost_index[k]], x, u, terminal, v, gr, y);
  if (P.gl) goal_lin_grad<n, m>(P.gl + ((size_t)tile * (size_t)(P.n_costs * nz)) * 64 + lane64, P.cost_index[k], terminal, gr);  // per-trajectory q, r
  if (P.opts.cost_dt_scaling && !terminal) {
    const double h = P.dt[k];
#pragma unroll
    for (int i = 0; i < nz; ++i) { gr[i] *= h; y[i] *= h; }
  }
  if ((VAR & 2) != 0) {
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    // constraints in the order of the list (the sums below are order-sensitive in the last bits): the cached ones from
    // registers, the rest — only where table_cons says one applies to this wave's knots — through the descriptor table
    if (!table_cons) {
      if (on0) al_grad_hvp_ctrl<n, m>(ce0, z, l0, mu0r, v, gr, y);
      if (on1) al_grad_hvp_ctrl<n, m>(ce1, z, l1, mu1r, v, gr, y);
    } else {
      const double* lam0 = a.lam + ((size_t)tile * (size_t)P.n_duals) * 64 + lane64;
      const double* mu0 = a.mu + ((size_t)tile * (size_t)P.n_cons) * 64 + lane64;
      for (int ci = 0; ci < P.n_cons; ++ci) {
        if (ci == ce0.ci) { if (on0) al_grad_hvp_ctrl<n, m>(ce0, z, l0, mu0r, v, gr, y); continue; }
        if (ci == ce1.ci) { if (on1) al_grad_hvp_ctrl<n, m>(ce1, z, l1, mu1r, v, gr, y); continue; }
        ConC& K = P.cons[ci];
        if (k < K.k1 || k > K.k2) continue;
        const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
        if constexpr ((VAR & 4) != 0) {  // per-trajectory constraint parameters (DevProblem::cp): the general variant only
          double zc[nz];
#pragma unroll
          for (int i = 0; i < nz; ++i) zc[i] = z[i];
          con_shift<nz>(P, K, P.cp + ((size_t)tile * (size_t)P.n_cp) * 64 + lane64, zc);
          al_grad_hvp<n, m, true>(K, zc, lam, (size_t)64, EL(mu0, ci), v, gr, y, P.opts.al_full_newton != 0);
        } else
        al_grad_hvp<n, m, (VAR & 4) != 0>(K, z, lam, (size_t)64, EL(mu0, ci), v, gr, y, P.opts.al_full_newton != 0);
      }
    }
  }
  if constexpr (LAY == 2 && M::lie) {
    // Compact cost block (diagonal + 3x3 attitude block, KArgs::h_compact): a lane needs its column's DIAGONAL entry, the
    // attitude lanes rows 3..5, the control lanes the control rows — and entry j of the projected gradient.  With v = column j of
    // G(x) (or e_{n+r}) already in registers these are dot products, (G'y)[j] = v·y and (G'g)[j] = v·g: the zeros of v add
    // exactly nothing, so the values are those of picking entry j out of G'y / G'g — without the 16-deep select chains per
    // picked entry that were half of this kernel's non-FP64 instructions (profiles/r03: 49 % of its VALU instructions).
    double gj = 0.0, dj = 0.0;
    if constexpr (PACK) {  // the own direction has a zero at sc: its terms are left out, not multiplied by the 1 planted there
#pragma unroll
      for (int i = 0; i < nz; ++i) gj += ((jc >= 0 && i == sc) ? 0.0 : v[i]) * gr[i];
#pragma unroll
      for (int i = 0; i < n; ++i) dj += ((jc >= 0 && i == sc) ? 0.0 : v[i]) * y[i];
      if (jc >= 0 && valid_c) {  // cost entries of the constant column jc: diagonal entry of the block, gradient component
        const double dc = pick<n>(y, sc), gc = pick<n>(gr, sc);
        a.Ht[((size_t)b * N + k) * 64 + (jc & 3) * 16 + jc] = dc;
        a.gt[((size_t)b * N + k) * 16 + jc] = gc;
      }
    } else {
#pragma unroll
    for (int i = 0; i < nz; ++i) gj += v[i] * gr[i];
#pragma unroll
    for (int i = 0; i < n; ++i) dj += v[i] * y[i];
    }
    double col[ne];
    errstate_tmul<M>(x, y, col);  // (only the attitude rows 3..5 are used below)
    if constexpr (M::att == ATT_QUAT) {  // second-order term of the attitude map: −I₃ (qᵀ ∂J/∂q) on the attitude diagonal
      const double b1 = x[3] * gr[3] + x[4] * gr[4] + x[5] * gr[5] + x[6] * gr[6];
#pragma unroll
      for (int i = 3; i < 6; ++i) col[i] -= (i == j) ? b1 : 0.0;
    } else if constexpr (M::att == ATT_MRP || M::att == ATT_RP) {  // ... of a three-parameter attitude: ∇²differential(p, ∂J/∂p)
      double H2[9];
      att_differential2<M::att>(x + 3, gr + 3, H2);
#pragma unroll
      for (int i = 0; i < 3; ++i) col[3 + i] += (j == 3) ? H2[3 * i] : (j == 4) ? H2[3 * i + 1] : (j == 5) ? H2[3 * i + 2] : 0.0;
    }
    if (!valid) return;
    const bool isctl = ct >= NEP, isatt = (ct >= 3 && ct < 6);
    double* Ht = a.Ht + ((size_t)b * N + k) * 64 + ct;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // compact_row(g, ct), by lane class
      constexpr int arow[4] = {4, 5, 3, 3};  // attitude lanes: g = 0 -> row 4, 1 -> 5, 3 -> 3 (g = 2: none)
      const double yc = (g < m && !terminal) ? y[n + (g < m ? g : 0)] : 0.0;
      const double va = col[arow[g]];
      const double val = isctl ? yc : isatt ? va : dj;
      const bool ok = isctl ? (g < m) : isatt ? (g != 2) : ((ct & 3) == g);
      if (ok) Ht[g * 16] = val;
    }
    a.gt[((size_t)b * N + k) * 16 + ct] = gj;
    return;
  }
  double col[ne], qxe[ne];
  errstate_tmul<M>(x, y, col);
  errstate_tmul<M>(x, gr, qxe);
  if constexpr (M::att == ATT_QUAT) {  // second-order term of the attitude map: −I₃ (qᵀ ∂J/∂q) on the attitude diagonal
    const double b1 = x[3] * gr[3] + x[4] * gr[4] + x[5] * gr[5] + x[6] * gr[6];
#pragma unroll
    for (int i = 3; i < 6; ++i) col[i] -= (i == j) ? b1 : 0.0;
  } else if constexpr (M::att == ATT_MRP || M::att == ATT_RP) {  // ... of a three-parameter attitude: ∇²differential(p, ∂J/∂p), a full 3x3 block
    double H2[9];
    att_differential2<M::att>(x + 3, gr + 3, H2);
#pragma unroll
    for (int i = 0; i < 3; ++i) col[3 + i] += (j == 3) ? H2[3 * i] : (j == 4) ? H2[3 * i + 1] : (j == 5) ? H2[3 * i + 2] : 0.0;
  }
  double gj = 0.0;
#pragma unroll
  for (int i = 0; i < ne; ++i) gj = (i == j) ? qxe[i] : gj;
#pragma unroll
  for (int r = 0; r < m; ++r) gj = (ne + r == j) ? gr[n + r] : gj;
  if (!valid) return;
  if constexpr (LAY == 0) {
    if (a.h_diag) {  // the column is zero off its diagonal entry (KArgs::h_diag): one row per knot instead of nc
      double dj = 0.0;
#pragma unroll
      for (int i = 0; i < ne; ++i) dj = (i == j) ? col[i] : dj;
#pragma unroll
      for (int r = 0; r < m; ++r) dj = (ne + r == j) ? (terminal ? 0.0 : y[n + r]) : dj;
      EL(COL_PTR(a.Hc, N), k) = dj;
    } else {
      double* Hc = COL_PTR(a.Hc, N * nc);
#pragma unroll
      for (int i = 0; i < ne; ++i) EL(Hc, k * nc + i) = col[i];
#pragma unroll
      for (int r = 0; r < m; ++r) EL(Hc, k * nc + ne + r) = terminal ? 0.0 : y[n + r];
    }
    double* gc = COL_PTR(a.gc, N);
    EL(gc, k) = gj;
  } else if constexpr (LAY == 3) {  // upper triangle of the symmetric block, column j from its own lane
    using L = LaneLay<M>;
    double* Hl = a.Hc + (((size_t)tile * (size_t)N + k) * L::NS + (size_t)(j * (j + 1) / 2)) * 64 + lane64;
#pragma unroll
    for (int i = 0; i < ne; ++i)
      if (i <= j) EL(Hl, i) = col[i];
#pragma unroll
    for (int r = 0; r < m; ++r)
      if (ne + r <= j) EL(Hl, ne + r) = terminal ? 0.0 : y[n + r];
    double* gl = a.gc + (((size_t)tile * (size_t)N + k) * nc + j) * 64 + lane64;
    gl[0] = gj;
  } else {
    if constexpr (LAY == 1) {
      double* Ht = a.Ht + (((size_t)b * N + k) * NR) * 64 + ct;
#pragma unroll
      for (int i = 0; i < ne; ++i) Ht[(i / 4) * 64 + (i % 4) * 16] = col[i];
#pragma unroll
      for (int r = 0; r < m; ++r) Ht[((NEP + r) / 4) * 64 + ((NEP + r) % 4) * 16] = terminal ? 0.0 : y[n + r];
    } else {  // one entry per lane group: the row compact_row(g, column) of this column
      double* Ht = a.Ht + ((size_t)b * N + k) * 64 + ct;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = compact_row<M>(g, ct);
        if (i < 0) continue;
        double v = 0.0;
        if (i >= NEP) v = terminal ? 0.0 : pick<m>(y + n, i - NEP); else v = pick<ne>(col, i);
        Ht[g * 16] = v;
      }
    }
    a.gt[((size_t)b * N + k) * 16 + ct] = gj;
  }
}

#ifndef TO_EXPAND_KC_CONS
#define TO_EXPAND_KC_CONS 4  // knots per wave of the CONSTRAINED variants of models that pipeline at all (0: the model's expand_knots for every variant)
#endif
template <class M, int VAR>
__host__ __device__ constexpr int expand_kc() {
  return ((VAR & 2) != 0 && TO_EXPAND_KC_CONS > 0 && M::expand_knots > 1) ? TO_EXPAND_KC_CONS : M::expand_knots;
}

// A wave walks M::expand_knots consecutive knots and fetches the next knot's state/control while it works on the
// current one: with one wave per SIMD (Quadrotor) nothing else hides the load round trip, which was half of the wave's
// life (rocprof: SQ_WAIT_ANY 49 % of SQ_WAVE_CYCLES, 60 % with AL terms).  x_{k+1} is shared between neighbours.
#ifndef TO_EXPAND_WAVES
#define TO_EXPAND_WAVES 1  // minimum waves per SIMD k_expand is compiled for (register cap 512 / waves)
#endif
constexpr int EXPAND_PACK_G = 6, EXPAND_PACK_R = 10;  // packed expansion: trajectories per wave x differentiated columns (lanes 60..63 idle)
template <class M, int FIXED_INTEG, int VAR, int LAY, bool PACK = false>
__global__ void __launch_bounds__(64, TO_EXPAND_WAVES) k_expand(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m, KC = expand_kc<M, VAR>();
  constexpr int R = PACK ? EXPAND_PACK_R : Coop<M>::R, G = PACK ? EXPAND_PACK_G : Coop<M>::G;
  const int gtile = blockIdx.x, lane = threadIdx.x;
  // PACK: lane jj of a trajectory's ten owns column 3..5 (attitude), 9..11 (ω) or 12..15 (controls); the first six also serve the cost
  // entries of constant column 0..2 (r) / 6..8 (v); lanes 60..63 ride along on the last trajectory without storing anything
  const int g = PACK ? (lane < G * R ? lane / R : G - 1) : lane / R, jj = PACK ? (lane < G * R ? lane % R : R - 1) : lane % R;
  const int j = PACK ? (jj < 3 ? 3 + jj : jj < 6 ? 6 + jj : 6 + jj) : jj;   // 0,1,2 -> 3,4,5;  3,4,5 -> 9,10,11;  6..9 -> 12..15
  const int jc = PACK ? (jj < 3 ? jj : jj < 6 ? 3 + jj : -1) : -1;            // 0,1,2 -> 0,1,2;  3,4,5 -> 6,7,8
  const bool lane_live = !PACK || lane < G * R;
  const DevProblem& P = a.P;
  const int N = P.N;
  const int k0 = blockIdx.y * KC;
  // trajectory of this lane group: position gtile*G + g of the batch, or — active-list compaction (tangent-matrix layouts,
  // which address everything by trajectory) — of this step's list, so that the waves stay full while the batch drains
  int b;
  bool inrange;
  if (LAY != 0 && LAY != 3 && a.compact) {
    const int cnt = a.acount[a.step & 1];
    if (gtile * G >= cnt) return;  // wave-uniform
    inrange = (gtile * G + g) < cnt;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + (inrange ? gtile * G + g : cnt - 1)];
  } else {
    b = gtile * G + g;
    inrange = b < P.B;
  }
  // idle lanes (padding columns, finished trajectories) compute along with EXEC full — partially masked FP64 issues
  // ~1.3x slower on gfx950 — and only their stores are predicated; a wave without any work leaves
  const bool lane_ok = lane_live && inrange && j < nc && a.active[inrange ? b : 0];
  if (__ballot(lane_ok) == 0) return;
  // Inside a solve the step accepted by the previous forward pass still sits in its candidate slot (acc != 0): the
  // expansion reads it there and writes it through to slot 0, so the separate k_accept copy (and its re-read of every
  // candidate line that any lane of a tile accepted) disappears from the iteration.  Outside a solve acc is 0.
  // (Small models only: for the Quadrotor the gathered reads cost the expansion what k_accept costs — measured.)
  const int c = (M::accept_write_through && inrange) ? a.acc[b] : 0;
  const int tile = b >> 6, lane64 = b & 63;
  // constraint descriptors, once per wave: up to two control-block constraints go to registers; table_cons: some other
  // constraint applies to one of this wave's knots (for the usual goal constraint: only the wave at the terminal knot)
  ConExp<m> ce0, ce1;
  bool table_cons = false;
  if constexpr ((VAR & 2) != 0) {
    for (int ci = 0; ci < P.n_cons; ++ci) {
      ConC& K = P.cons[ci];
      if (K.fast == 2 && K.p <= m + 1 && ce1.ci < 0) { if (ce0.ci < 0) ce0.load(K, ci); else ce1.load(K, ci); }
      else if (K.k1 <= k0 + KC - 1 && K.k2 >= k0) table_cons = true;
    }
  }
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
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
      double* X0 = X_SLOT_PTR(a, b, 0);
      double* U0 = U_SLOT_PTR(a, b, 0);
#pragma unroll
      for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
      if (!terminal) {
#pragma unroll
        for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
      }
    }
    expand_knot<M, FIXED_INTEG, VAR, LAY, PACK>(a, gtile, lane, tile, lane64, b, j, k, valid, x, u, x1, ce0, ce1, table_cons, jc, lane_ok);
    if (KC > 1) {
#pragma unroll
      for (int i = 0; i < n; ++i) { x[i] = x1[i]; x1[i] = x2[i]; }
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = un[i];
    }
  }
}

// The constant columns of [Ā B̄] of the packed expansion (above), written ONCE per handle (the time steps never change): position columns
// e_c, velocity columns [h_k e_d; 0; e_d; 0], every other row zero.  grid (B, ceil((N-1)/4)), 64 lanes = 4 knots x 16 rows/columns slots.
template <class M>
__global__ void __launch_bounds__(64) k_expand_const_columns(KArgs a) {
  constexpr int ne = M::ne, RS = Tm<M>::RS;
  const DevProblem& P = a.P;
  const int b = blockIdx.x, k = blockIdx.y * 4 + (threadIdx.x >> 4), t = threadIdx.x & 15;
  if (b >= P.B || k >= P.N - 1 || t >= ne) return;
  const int i = t;  // row
  double* Mt = a.Mt + (((size_t)b * (P.N - 1) + k) * RS) * 64 + (i / 4) * 64 + (i % 4) * 16;
  const double h = P.dt[k];
  // ∂r⁺/∂v is h in exact arithmetic; the value stored is the one the dual numbers produce when they run through rk_step (models.h) —
  // k_i.d = h, acc = k1 + 2 k2 + 2 k3 (RK3: k1 + 4 k2), x + (acc + k_last) (1/6), with the roundings of those operations — so that the packed
  // kernel and the 4 x 16 one agree to the last bit (the zig-zag solve is chaotic enough to amplify one ulp of this entry to 1e-6)
  double dv = h;
  if (P.integrator == INTEG_RK3) { double acc = h; acc = fma(4.0, h, acc); dv = (acc + h) * (1.0 / 6.0); }
  else if (P.integrator != INTEG_EULER) { double acc = h; acc = fma(2.0, h, acc); acc = fma(2.0, h, acc); dv = (acc + h) * (1.0 / 6.0); }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Mt[c] = (i == c) ? 1.0 : 0.0;                                   // ∂x⁺/∂r_c
    Mt[6 + c] = (i == c) ? dv : (i == 6 + c) ? 1.0 : 0.0;           // ∂x⁺/∂v_c
  }
}

// ------------------------------------------------------------------------------------------------ lane expansion
// Small vector-space models with the one-lane-per-trajectory backward pass (large batches): ONE LANE per (trajectory, knot)
// produces the whole expansion of its knot — all nc = ne + m columns of [A B] from a single pass of the RK stages in
// chunk-mode dual numbers (MDual<nc>, models.h), then the cost (+AL) columns.  The column-per-lane kernel above spends 8
// lanes (5 useful on the Cartpole) per (trajectory, knot), each re-evaluating the VALUE part of the dynamics — 0.65 MFLOP
// issued per Cartpole trajectory-iteration for a 0.1 MFLOP expansion — this one evaluates it once:
// 1 971 instead of 8 x 1 037 lane-instructions per Cartpole knot.
//
// expand_lane_knot: the expansion of knot k of this lane's trajectory INTO REGISTERS, in the lane layout's own order
// (k_backward.h, LaneLay): Mk[i*nc + j] = [A B][i][j] (not touched at the terminal knot), H[sym(i, j)] = upper triangle of
// the cost (+AL) block, g[j] = gradient.  Same calls and the same order of the constraint terms as expand_knot's table path.
template <class M, int FIXED_INTEG, int VAR>
__device__ __forceinline__ void expand_lane_knot(const KArgs& a, int tile, int lane, int k, const double* x, const double* u,
                                                 double* Mk, double* H, double* g) {
  static_assert(!M::lie, "lane expansion: vector-space models only (identity error-state maps)");
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m, nc = ne + m;
  const DevProblem& P = a.P;
  const bool terminal = (k == P.N - 1);
  if constexpr (has_stage_jac<M>::value && FIXED_INTEG == INTEG_RK4) {
    if (!terminal) cartpole_rk4_jac(P.mp, x, u, P.dt[k], Mk);  // chain rule over hand-derived stage partials (models.h)
  } else
  if (!terminal) {  // every column of [A B] in one pass
    MDual<nc> xd[n], ud[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { xd[i].v = x[i]; xd[i].d[i] = 1.0; }
#pragma unroll
    for (int i = 0; i < m; ++i) { ud[i].v = u[i]; ud[i].d[ne + i] = 1.0; }
    model_step<M, MDual<nc>, FIXED_INTEG>(P.mp, P.integrator, k, xd, ud, P.dt[k], xn);
#pragma unroll
    for (int i = 0; i < ne; ++i)
#pragma unroll
      for (int j = 0; j < nc; ++j) Mk[i * nc + j] = xn[i].d[j];
  }
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = u[i];
  const double* lam0 = a.lam + ((size_t)tile * (size_t)P.n_duals) * 64 + lane;
  const double* mu0 = a.mu + ((size_t)tile * (size_t)P.n_cons) * 64 + lane;
  if constexpr (VAR == 0) {
    // Plain DiagonalCost without constraints (C2 and the large-batch sweep): the block is diag(Q, R) and the gradient Q x + q, R u + r —
    // ONE batch of scalar descriptor loads per knot.  The general loop below asks cost_grad_hvp for one column at a time, and each
    // call re-reads the descriptor (five groups of scalar loads with a wait each per Cartpole knot: ~6 % of a knot at one wave per SIMD).
    // Same products and sums as the general path computes for unit vectors (its multiplications by 0 and 1 are exact).
    CostC& C = P.costs[P.cost_index[k]];
    if (C.kind == TO_COST_DIAGONAL) {
      const double sc = (P.opts.cost_dt_scaling && !terminal) ? P.dt[k] : 1.0;
      double gr[nz];
#pragma unroll
      for (int i = 0; i < n; ++i) gr[i] = C.Q[i] * x[i] + C.q[i];
#pragma unroll
      for (int i = 0; i < m; ++i) gr[n + i] = terminal ? 0.0 : C.R[i] * u[i] + C.r[i];
      if (P.gl) goal_lin_grad<n, m>(P.gl + ((size_t)tile * (size_t)(P.n_costs * nz)) * 64 + lane, P.cost_index[k], terminal, gr);
#pragma unroll
      for (int j = 0; j < nc; ++j) {
#pragma unroll
        for (int i = 0; i <= j; ++i) H[j * (j + 1) / 2 + i] = 0.0;
        H[j * (j + 1) / 2 + j] = (j < ne) ? C.Q[j] * sc : (terminal ? 0.0 : C.R[j - ne] * sc);
        g[j] = gr[j < ne ? j : n + (j - ne)] * sc;
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < nc; ++j) {
    double v[nz], gr[nz], y[nz];
#pragma unroll
    for (int i = 0; i < nz; ++i) v[i] = (i == (j < ne ? j : n + (j - ne))) ? 1.0 : 0.0;
    cost_grad_hvp<n, m, (VAR & 1) != 0>(P.costs[P.cost_index[k]], x, u, terminal, v, gr, y);
    if (P.gl) goal_lin_grad<n, m>(P.gl + ((size_t)tile * (size_t)(P.n_costs * nz)) * 64 + lane, P.cost_index[k], terminal, gr);  // per-trajectory q, r
    if (P.opts.cost_dt_scaling && !terminal) {
      const double h = P.dt[k];
#pragma unroll
      for (int i = 0; i < nz; ++i) { gr[i] *= h; y[i] *= h; }
    }
    if constexpr ((VAR & 2) != 0) {
      for (int ci = 0; ci < P.n_cons; ++ci) {
        ConC& K = P.cons[ci];
        if (k < K.k1 || k > K.k2) continue;
        const double* lam = lam0 + (size_t)(K.dual_off + (long long)(k - K.k1) * K.p) * 64;
        if constexpr ((VAR & 4) != 0) {  // per-trajectory constraint parameters (DevProblem::cp): the general variant only
          double zc[nz];
#pragma unroll
          for (int i = 0; i < nz; ++i) zc[i] = z[i];
          con_shift<nz>(P, K, P.cp + ((size_t)tile * (size_t)P.n_cp) * 64 + lane, zc);
          al_grad_hvp<n, m, true>(K, zc, lam, (size_t)64, EL(mu0, ci), v, gr, y, P.opts.al_full_newton != 0);
        } else
        al_grad_hvp<n, m, (VAR & 4) != 0>(K, z, lam, (size_t)64, EL(mu0, ci), v, gr, y, P.opts.al_full_newton != 0);
      }
    }
#pragma unroll
    for (int i = 0; i < ne; ++i)
      if (i <= j) H[j * (j + 1) / 2 + i] = y[i];
#pragma unroll
    for (int r = 0; r < m; ++r)
      if (ne + r <= j) H[j * (j + 1) / 2 + ne + r] = terminal ? 0.0 : y[n + r];
    g[j] = gr[j < ne ? j : n + (j - ne)];
  }
}

// the expansion as a kernel of its own (phase API, and the solve loop when the fused backward pass is switched off):
// writes the lane layout as whole 512-byte rows, lane = trajectory.  grid (Bp / 64, N).
template <class M, int FIXED_INTEG, int VAR>
__global__ void __launch_bounds__(64) k_expand_lane(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m;
  using L = LaneLay<M>;
  const DevProblem& P = a.P;
  const int N = P.N;
  const int tile = blockIdx.x, lane = threadIdx.x, b = tile * 64 + lane, k = blockIdx.y;  // b < Bp always
  // lanes without a trajectory to expand compute along on their own (valid) data; only their stores are predicated
  const bool live = (b < P.B) && a.active[b] != 0;
  if (__ballot(live) == 0) return;
  const bool terminal = (k == N - 1);
  const int c = (M::accept_write_through && b < P.B) ? a.acc[b] : 0;  // the step the last forward pass accepted (k_expand)
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  double x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = terminal ? 0.0 : EL(U, k * m + i);
  if (M::accept_write_through && c != 0 && live) {
    double* X0 = X_SLOT_PTR(a, b, 0);
    double* U0 = U_SLOT_PTR(a, b, 0);
#pragma unroll
    for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
    if (!terminal) {
#pragma unroll
      for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
    }
  }
  double Mk[ne * nc], H[L::NS], g[nc];
  expand_lane_knot<M, FIXED_INTEG, VAR>(a, tile, lane, k, x, u, Mk, H, g);
  if (!live) return;
  if (!terminal) {
    double* Ml = a.Mc + (((size_t)tile * (size_t)(N - 1) + k) * (ne * nc)) * 64 + lane;
#pragma unroll
    for (int e = 0; e < ne * nc; ++e) EL(Ml, e) = Mk[e];
  }
  double* Hl = a.Hc + (((size_t)tile * (size_t)N + k) * L::NS) * 64 + lane;
  double* gl = a.gc + (((size_t)tile * (size_t)N + k) * nc) * 64 + lane;
#pragma unroll
  for (int j = 0; j < nc; ++j) {
    if (terminal && j >= ne) continue;  // no control directions at the terminal knot
#pragma unroll
    for (int i = 0; i <= j; ++i) EL(Hl, j * (j + 1) / 2 + i) = H[j * (j + 1) / 2 + i];
    EL(gl, j) = g[j];
  }
}

// One knot of the lane Riccati recursion — k_backward_lane's knot, verbatim — on operands in registers: [A B] of the knot (Mk), its cost
// block and gradient in the lane layout (H, g), the cost-to-go (S, s: updated in place).  Stores the gains row at pKk if `live`.
// false: Quu + rho I is not positive definite (nothing stored, S and s untouched).
template <class M>
__device__ __forceinline__ bool lane_riccati_knot(const double (&Mk)[M::ne][M::ne + M::m], const double* H, const double* g, double (&S)[M::ne][M::ne],
                                                  double (&s)[M::ne], double rho, bool live, double* pKk, double& dV0, double& dV1) {
  constexpr int m = M::m, ne = M::ne, nc = ne + m;
  using L = LaneLay<M>;
    double T[ne][nc];
#pragma unroll
    for (int i = 0; i < ne; ++i)
#pragma unroll
      for (int j = 0; j < nc; ++j) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < ne; ++r) t += S[i][r] * Mk[r][j];
        T[i][j] = t;
      }
    double Qxx[ne][ne], Qux[m][ne], Quu[m][m], gq[nc];
#pragma unroll
    for (int j = 0; j < ne; ++j) {
#pragma unroll
      for (int i = 0; i < nc; ++i) {
        double t = H[L::sym(i, j)];
#pragma unroll
        for (int r = 0; r < ne; ++r) t += Mk[r][i] * T[r][j];
        if (i < ne) Qxx[i][j] = t; else Qux[i - ne][j] = t;
      }
    }
#pragma unroll
    for (int q = 0; q < m; ++q)
#pragma unroll
      for (int p = 0; p < m; ++p) {
        double t = H[L::sym(ne + p, ne + q)];
#pragma unroll
        for (int r = 0; r < ne; ++r) t += Mk[r][ne + p] * T[r][ne + q];
        Quu[p][q] = t;
      }
#pragma unroll
    for (int j = 0; j < nc; ++j) {
      double t = g[j];
#pragma unroll
      for (int r = 0; r < ne; ++r) t += Mk[r][j] * s[r];
      gq[j] = t;
    }
    double Lc[m][m], iL[m];
    bool pd_ok = true;
#pragma unroll
    for (int r = 0; r < m; ++r)
#pragma unroll
      for (int q = 0; q < m; ++q) Lc[r][q] = Quu[r][q] + ((r == q) ? rho : 0.0);
#pragma unroll
    for (int q = 0; q < m; ++q) {
      double sj = Lc[q][q];
#pragma unroll
      for (int r = 0; r < q; ++r) sj -= Lc[q][r] * Lc[q][r];
      if (not_positive(sj) && live) pd_ok = false;
      iL[q] = rsqrt_fast(sj);
      Lc[q][q] = sj * iL[q];
#pragma unroll
      for (int i = q + 1; i < m; ++i) {
        double t = Lc[i][q];
#pragma unroll
        for (int r = 0; r < q; ++r) t -= Lc[i][r] * Lc[q][r];
        Lc[i][q] = t * iL[q];
      }
    }
    if (!pd_ok) return false;
    double Kg[m][ne], dk[m];
#pragma unroll
    for (int cc = 0; cc <= ne; ++cc) {
      double col[m];
#pragma unroll
      for (int i = 0; i < m; ++i) col[i] = (cc < ne) ? Qux[i][cc < ne ? cc : 0] : gq[ne + i];
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
      for (int i = 0; i < m; ++i) { if (cc < ne) Kg[i][cc < ne ? cc : 0] = -col[i]; else dk[i] = -col[i]; }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < m; ++r) {
#pragma unroll
        for (int j = 0; j < ne; ++j) pKk[r * (ne + 1) + j] = Kg[r][j];
        pKk[r * (ne + 1) + ne] = dk[r];
      }
    }
    double W[m][ne], qd[m];
#pragma unroll
    for (int r = 0; r < m; ++r) {
#pragma unroll
      for (int j = 0; j < ne; ++j) {
        double t = Qux[r][j];
#pragma unroll
        for (int q = 0; q < m; ++q) t += Quu[r][q] * Kg[q][j];
        W[r][j] = t;
      }
      double t2 = gq[ne + r];
#pragma unroll
      for (int q = 0; q < m; ++q) t2 += Quu[r][q] * dk[q];
      qd[r] = t2;
    }
    double Sn[ne][ne], sn[ne];
#pragma unroll
    for (int j = 0; j < ne; ++j) {
#pragma unroll
      for (int i = 0; i < ne; ++i) {
        double t = Qxx[i][j];
#pragma unroll
        for (int r = 0; r < m; ++r) t += Kg[r][i] * W[r][j];
#pragma unroll
        for (int r = 0; r < m; ++r) t += Qux[r][i] * Kg[r][j];
        Sn[i][j] = t;
      }
      double t = gq[j];
#pragma unroll
      for (int r = 0; r < m; ++r) t += Kg[r][j] * qd[r];
#pragma unroll
      for (int r = 0; r < m; ++r) t += Qux[r][j] * dk[r];
      sn[j] = t;
    }
    double dv1 = 0.0, dv2 = 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r) {
      dv1 += dk[r] * gq[ne + r];
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < m; ++q) t += Quu[r][q] * dk[q];
      dv2 += dk[r] * t;
    }
    dV0 += dv1;
    dV1 += 0.5 * dv2;
#pragma unroll
    for (int i = 0; i < ne; ++i) {
#pragma unroll
      for (int j = 0; j < ne; ++j) S[i][j] = 0.5 * (Sn[i][j] + Sn[j][i]);
      s[i] = sn[i];
    }
  return true;
}

// ------------------------------------------------------------------------------------------------ fused lane expansion + Riccati
// Large batches of the small models are HBM-bound (measured at B = 32 768: 13.9 M trajectory-iterations/s whichever
// expansion kernel, backward-pass flavour or line-search width runs — 0.85 ms per batch step for ~3 GB of traffic), and two
// thirds of that traffic is the expansion itself: 40 doubles per Cartpole knot written by the expansion and read straight
// back by the backward pass.  Here the lane that walks a trajectory's Riccati recursion expands each knot in its own
// registers right before it consumes it — [A B], the cost block and the gradient never exist in memory.  What is left per
// knot: x, u (read), the accepted step written through (5 doubles) and the gains row (5 doubles).
// Same arithmetic, same order as k_expand_lane followed by k_backward_lane (bit-identical gains).
#ifndef TO_FUSED_LANE_WAVES
#define TO_FUSED_LANE_WAVES 1  // minimum waves per SIMD the fused lane kernel is compiled for (register cap 512 / waves)
#endif
template <class M, int FIXED_INTEG, int VAR>
__global__ void __launch_bounds__(64, TO_FUSED_LANE_WAVES) k_expand_backward_lane(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m, RSK = Gains<M>::RSK;
  using L = LaneLay<M>;
  constexpr int NS = L::NS;
  const DevProblem& P = a.P;
  const int N = P.N;
  // this lane's trajectory: lane of the tile, or — with active-list compaction — entry blockIdx.x*64 + lane of this step's list
  int b, inrange;
  if (a.compact) {
    const int cnt = a.acount[a.step & 1];
    if ((int)blockIdx.x * 64 >= cnt) return;  // wave-uniform
    const int li = blockIdx.x * 64 + threadIdx.x;
    inrange = li < cnt;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + (inrange ? li : cnt - 1)];
  } else {
    b = blockIdx.x * 64 + threadIdx.x;  // b < Bp always
    inrange = b < P.B;
  }
  const int tile = b >> 6, lane = b & 63;
  const bool live = inrange && a.active[b] != 0;
  if (__ballot(live) == 0) return;
  const int c = (M::accept_write_through && inrange) ? a.acc[b] : 0;
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  double* X0 = X_SLOT_PTR(a, b, 0);
  double* U0 = U_SLOT_PTR(a, b, 0);
  const bool wt = M::accept_write_through && c != 0 && live;
  double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  double S[ne][ne], s[ne];
  while (true) {  // one pass of the recursion; a Cholesky failure raises rho and starts over
    {  // terminal knot: S = Qxx_N, s = qx_N
      double x[n], u[m], Mk[1], H[NS], g[nc];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = EL(X, (N - 1) * n + i);
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = 0.0;
      if (wt) {
#pragma unroll
        for (int i = 0; i < n; ++i) EL(X0, (N - 1) * n + i) = x[i];
      }
      expand_lane_knot<M, FIXED_INTEG, VAR>(a, tile, lane, N - 1, x, u, Mk, H, g);
#pragma unroll
      for (int i = 0; i < ne; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) S[i][j] = H[L::sym(i, j)];
        s[i] = g[i];
      }
    }
    dV0 = 0.0; dV1 = 0.0;
    bool restart = false;
    double xn_[n], un_[m];  // knot k's state / control, fetched one knot ahead
#pragma unroll
    for (int i = 0; i < n; ++i) xn_[i] = EL(X, (N - 2) * n + i);
#pragma unroll
    for (int i = 0; i < m; ++i) un_[i] = EL(U, (N - 2) * m + i);
    for (int kv = N - 2; kv >= 0; --kv) {
      // Lanes leave this loop one by one (a Cholesky failure breaks out for ITS lane), so the compiler treats the counter as lane-
      // dependent and fetched everything indexed by it — P.dt[k], P.cost_index[k], the cost descriptor behind it — with VECTOR loads
      // and a vmcnt(0) wait each: three dependent memory round trips per knot that also drained the prefetch of the next knot's
      // nominal (r05 counters: 44 % of the wave cycles waiting at one wave per SIMD).  Every lane still in the loop holds the same
      // value: read it from the first one, and the tables come through scalar loads again.
      const int k = __builtin_amdgcn_readfirstlane(kv);
      double x[n], u[m];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = xn_[i];
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = un_[i];
      if (k > 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) xn_[i] = EL(X, (k - 1) * n + i);
#pragma unroll
        for (int i = 0; i < m; ++i) un_[i] = EL(U, (k - 1) * m + i);
      }
      if (wt) {
#pragma unroll
        for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
#pragma unroll
        for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
      }
      double Me[ne * nc], H[NS], g[nc];
      expand_lane_knot<M, FIXED_INTEG, VAR>(a, tile, lane, k, x, u, Me, H, g);
      double Mk[ne][nc];
#pragma unroll
      for (int i = 0; i < ne; ++i)
#pragma unroll
        for (int j = 0; j < nc; ++j) Mk[i][j] = Me[i * nc + j];
      if (!lane_riccati_knot<M>(Mk, H, g, S, s, rho, live, pK + (size_t)k * RSK, dV0, dV1)) {
        reg_increase(P.opts, rho, drho);
        if (rho > P.opts.bp_reg_max) failed = true; else restart = true;
        break;
      }
    }
    if (!restart) break;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  if (live) {
    a.rho[b] = rho;
    a.drho[b] = drho;
    a.dV[b] = dV0;
    a.dV[(size_t)P.Bp + b] = dV1;
    a.bpfail[b] = failed ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------ fused expansion + cooperative Riccati
// Small batches of the small models (C2: B = 1024) are latency-bound: a batch step is expansion (20 us, launch-bound: all
// knots in parallel) -> cooperative backward pass (73 us: 100 sequential knots) -> forward pass (100 us), every trajectory
// needs ~110 of them one after the other, and the chip is nearly empty.  Here the expansion leaves the critical path: a
// workgroup of TWO waves owns the G trajectories of one column-layout group; wave 1 expands — one lane per (trajectory, knot),
// chunk-mode dual numbers (expand_lane_knot), KB = 64/G knots per pass, walking the horizon backwards — into a double-buffered
// LDS ring, wave 0 runs the Riccati recursion of k_backward_coop on the chunk expanded one pass earlier (a pass of the
// expander takes 3.3 us on the Cartpole, the 8 knots it feeds take the recursion 5.6 us).  [A B], the cost blocks and the
// gradients never exist in memory (55 of the 71 MB a C2 batch step moved).  A Cholesky failure in any trajectory restarts the
// whole workgroup with that trajectory's larger rho; the others redo exactly what they did (deterministic), so every
// trajectory ends with the result of its own restart sequence, as with the split kernels.
// Diagonal cost blocks only (KArgs::h_diag) and R <= 8 lanes per trajectory: what fits the ring.
template <class M>
struct FusedCoop {
  static constexpr int ne = M::ne, m = M::m, nc = ne + m, R = Coop<M>::R, G = Coop<M>::G;
  static constexpr int KB = 64 / G;            // knots per chunk: the expander's 64 lanes = KB knots x G trajectories
  static constexpr int EW = ne + 2;            // per (knot, trajectory, column): the column of [A B], the cost diagonal, the gradient
  static constexpr int chunk = KB * G * R * EW;  // doubles per ring buffer
};

// one knot of the cooperative recursion for lane (g, j) — the arithmetic of k_backward_coop's loop body, operands handed in:
// Mj = column j of [A B], Hj = column j of the cost block, gj = gradient entry.  false: Quu + rho I is not positive definite
// (nothing has been stored for this knot).
// merge: S_ holds the RAW cost-to-go Hessian S' of the previous knot and is symmetrised where it is read (the same expression
// ½(S'[i][r] + S'[r][i]), so bit-identical) — one LDS exchange round per knot less than staging S' and publishing ½(S' + S'ᵀ).
template <class M, bool merge>
__device__ __forceinline__ bool coop_knot(double* S_, double* Mx, double* Hu, double* Kf, double* gl, double* sl, int j, int jx, bool glive,
                                          const double* Mj, double* Hj, double gj, double rho, double* pKk, double& dV0, double& dV1,
                                          const double* Mring, int EWs) {
  // Mring (merge only): the group's [A B] of this knot as it sits in the expansion ring, M[r][i] = Mring[i*EWs + r] — the other
  // lanes' columns are read there instead of being exchanged through Mx (one more round gone)
  constexpr int m = M::m, ne = M::ne, nc = ne + m, R = Coop<M>::R;
  if (!merge) {
#pragma unroll
    for (int i = 0; i < ne; ++i) Mx[i * R + j] = Mj[i];
    WAVE_SYNC();
  }
  double Tj[ne];
#pragma unroll
  for (int i = 0; i < ne; ++i) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < ne; ++r) t += (merge ? 0.5 * (S_[i * ne + r] + S_[r * ne + i]) : S_[i * ne + r]) * Mj[r];
    Tj[i] = t;
  }
#pragma unroll
  for (int i = 0; i < nc; ++i) {
    double t = Hj[i];
#pragma unroll
    for (int r = 0; r < ne; ++r) t += (merge ? Mring[i * EWs + r] : Mx[r * R + i]) * Tj[r];
    Hj[i] = t;
  }
#pragma unroll
  for (int r = 0; r < ne; ++r) gj += Mj[r] * sl[r];
#pragma unroll
  for (int r = 0; r < m; ++r) Hu[r * R + j] = Hj[ne + r];
  gl[j] = gj;
  WAVE_SYNC();
  double Quu[m][m], Lc[m][m], Qu[m];
#pragma unroll
  for (int r = 0; r < m; ++r) {
#pragma unroll
    for (int q = 0; q < m; ++q) Quu[r][q] = Hu[r * R + ne + q];
    Qu[r] = gl[ne + r];
  }
  bool pd_ok = true;
  double iL[m];
#pragma unroll
  for (int r = 0; r < m; ++r)
#pragma unroll
    for (int q = 0; q < m; ++q) Lc[r][q] = Quu[r][q] + ((r == q) ? rho : 0.0);
#pragma unroll
  for (int q = 0; q < m; ++q) {
    double sj = Lc[q][q];
#pragma unroll
    for (int r = 0; r < q; ++r) sj -= Lc[q][r] * Lc[q][r];
    if (!(sj > 0.0) && glive) pd_ok = false;
    iL[q] = rsqrt_fast(sj);
    Lc[q][q] = sj * iL[q];
#pragma unroll
    for (int i = q + 1; i < m; ++i) {
      double t = Lc[i][q];
#pragma unroll
      for (int r = 0; r < q; ++r) t -= Lc[i][r] * Lc[q][r];
      Lc[i][q] = t * iL[q];
    }
  }
  if (!pd_ok) return false;  // same decision in every lane of the group
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
    for (int r = 0; r < m; ++r) { if constexpr (m > 1) Kf[r * ne + j] = Kj[r]; if (glive) pKk[r * (ne + 1) + j] = Kj[r]; }
  }
  if (j == 0 && glive) {
#pragma unroll
    for (int r = 0; r < m; ++r) pKk[r * (ne + 1) + ne] = dk[r];
  }
  if constexpr (m > 1) WAVE_SYNC();
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
    if (!merge) {
#pragma unroll
      for (int i = 0; i < ne; ++i) Mx[i * R + j] = Snew[i];
    }
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
  WAVE_SYNC();  // every lane has read this knot's S_, Hu, Mx
  if (merge) {  // publish the raw column; the next knot symmetrises as it reads
    if (j < ne) {
#pragma unroll
      for (int i = 0; i < ne; ++i) S_[i * ne + j] = Snew[i];
      sl[j] = snew;
    }
  } else {
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
  return true;
}

template <class M, int FIXED_INTEG, int VAR, bool MERGE>
__global__ void __launch_bounds__(128) k_expand_backward_coop(KArgs a) {
  static_assert(!M::lie && Coop<M>::R <= 8, "fused cooperative pass: vector-space models with at most 8 directions");
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m, RSK = Gains<M>::RSK;
  using F = FusedCoop<M>;
  using L = BwdLds<M>;
  using LL = LaneLay<M>;
  constexpr int R = F::R, G = F::G, KB = F::KB, EW = F::EW;
  __shared__ double ring[2 * F::chunk];
  __shared__ double lds[G * L::stride];
  __shared__ int restart_flag;
  const DevProblem& P = a.P;
  const int N = P.N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gtile = blockIdx.x;
  const int nch = (N + KB - 1) / KB;  // chunk c holds knots N-1 - c*KB - kk, kk = 0..KB-1 (the horizon backwards)
  // ---- roles.  Backward wave: lane (g, j) as in k_backward_coop.  Expander wave: lane (kk, ge) = knot kk of the chunk, trajectory ge.
  const int g = lane / R, j = lane % R;
  const int b = gtile * G + g;
  const bool glive = (b < P.B) && a.active[b < P.B ? b : 0];
  const int kk = lane / G, ge = lane % G;
  const int be = gtile * G + ge;
  const bool elive = (be < P.B) && a.active[be < P.B ? be : 0];
  {  // workgroup-uniform exit: nothing to do for any of the G trajectories
    __shared__ int any_live;
    if (threadIdx.x == 0) any_live = 0;
    __syncthreads();
    if (wave == 0 && glive && j == 0) any_live = 1;
    __syncthreads();
    if (!any_live) return;
  }
  // padding columns of the ring (j >= nc) are read by the riding-along lanes of the backward wave: keep them finite
  for (int i = threadIdx.x; i < 2 * F::chunk; i += 128) ring[i] = 0.0;
  const int jx = j < ne ? j : ne - 1;
  double* S_ = lds + g * L::stride + L::oS;
  double* Mx = lds + g * L::stride + L::oM;
  double* Hu = lds + g * L::stride + L::oH;
  double* Kf = lds + g * L::stride + L::oK;
  double* gl = lds + g * L::stride + L::oG;
  double* sl = lds + g * L::stride + L::os;
  double* pK = a.Kt + ((size_t)(b < P.B ? b : 0) * (N - 1)) * RSK;
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  // expander: where this lane's trajectory lives (the step accepted by the last forward pass is written through, as in k_expand)
  const int ce = (M::accept_write_through && be < P.B) ? a.acc[be] : 0;
  const double* Xe = X_SLOT_PTR(a, be < P.B ? be : 0, (be < P.B) ? ce : 0);
  const double* Ue = U_SLOT_PTR(a, be < P.B ? be : 0, (be < P.B) ? ce : 0);
  double* X0 = X_SLOT_PTR(a, be < P.B ? be : 0, 0);
  double* U0 = U_SLOT_PTR(a, be < P.B ? be : 0, 0);
  const int etile = (be < P.B ? be : 0) >> 6, elane = (be < P.B ? be : 0) & 63;
  while (true) {  // one pass over the horizon; repeated when a trajectory had to raise its regularisation
    if (threadIdx.x == 0) restart_flag = 0;
    __syncthreads();
    bool gstop = failed;  // this group takes no further part in the pass (it failed for good, or it asked for the restart)
    dV0 = 0.0; dV1 = 0.0;
    // the expander fetches the state / control of its NEXT knot while it works on the current one: a pass that starts with a
    // global load round trip took 7 us instead of the 3.3 us of its arithmetic, and the lock-step made the Riccati wave wait for it
    double xq[n], uq[m];
    if (wave == 1) {
      const int k = N - 1 - kk;
#pragma unroll
      for (int i = 0; i < n; ++i) xq[i] = (k >= 0) ? EL(Xe, k * n + i) : 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) uq[i] = (k >= 0 && k < N - 1) ? EL(Ue, k * m + i) : 0.0;
    }
    for (int s = 0; s <= nch; ++s) {
      if (wave == 1 && s < nch) {  // ---- expand chunk s into ring[s & 1]
        const int k = N - 1 - s * KB - kk;
        double x[n], u[m];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xq[i];
#pragma unroll
        for (int i = 0; i < m; ++i) u[i] = uq[i];
        {
          const int kn = k - KB;  // this lane's knot in the next chunk
#pragma unroll
          for (int i = 0; i < n; ++i) xq[i] = (kn >= 0) ? EL(Xe, (kn >= 0 ? kn : 0) * n + i) : 0.0;
#pragma unroll
          for (int i = 0; i < m; ++i) uq[i] = (kn >= 0) ? EL(Ue, (kn >= 0 ? kn : 0) * m + i) : 0.0;
        }
        if (k >= 0) {  // (wave-divergent only in the last chunk)
          const bool terminal = (k == N - 1);
          if (M::accept_write_through && ce != 0 && elive) {
#pragma unroll
            for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
            if (!terminal) {
#pragma unroll
              for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
            }
          }
          double Mk[ne * nc], H[LL::NS], gq[nc];
#pragma unroll
          for (int e = 0; e < ne * nc; ++e) Mk[e] = 0.0;
          expand_lane_knot<M, FIXED_INTEG, VAR>(a, etile, elane, k, x, u, Mk, H, gq);
          double* dst = ring + (size_t)(s & 1) * F::chunk + ((size_t)(kk * G + ge) * R) * EW;
#pragma unroll
          for (int jj = 0; jj < nc; ++jj) {
#pragma unroll
            for (int i = 0; i < ne; ++i) dst[jj * EW + i] = Mk[i * nc + jj];
            dst[jj * EW + ne] = (terminal && jj >= ne) ? 0.0 : H[LL::sym(jj, jj)];
            dst[jj * EW + ne + 1] = (terminal && jj >= ne) ? 0.0 : gq[jj];
          }
        }
      }
      if (wave == 0 && s >= 1 && !gstop) {  // ---- Riccati recursion over chunk s-1
        const double* src = ring + (size_t)((s - 1) & 1) * F::chunk;
        for (int q = 0; q < KB; ++q) {
          const int k = N - 1 - (s - 1) * KB - q;
          if (k < 0) break;
          const double* e = src + ((size_t)(q * G + g) * R + j) * EW;
          if (k == N - 1) {  // terminal knot: S = Qxx_N, s = qx_N
            const double hd = e[ne], s0 = e[ne + 1];
            if (j < ne) {
#pragma unroll
              for (int i = 0; i < ne; ++i) S_[i * ne + j] = (i == j) ? hd : 0.0;
              sl[j] = s0;
            }
            WAVE_SYNC();
            continue;
          }
          double Mj[ne], Hj[nc];
#pragma unroll
          for (int i = 0; i < ne; ++i) Mj[i] = e[i];
          const double hd = e[ne];
#pragma unroll
          for (int i = 0; i < nc; ++i) Hj[i] = (i == j) ? hd : 0.0;
          const double gj = e[ne + 1];
          if (!coop_knot<M, MERGE>(S_, Mx, Hu, Kf, gl, sl, j, jx, glive, Mj, Hj, gj, rho, pK + (size_t)k * RSK, dV0, dV1,
                                   src + ((size_t)(q * G + g) * R) * EW, EW)) {
            reg_increase(P.opts, rho, drho);
            if (rho > P.opts.bp_reg_max) failed = true;
            else restart_flag = 1;  // every lane of the group writes the same value
            gstop = true;
            break;
          }
        }
      }
      __syncthreads();
      if (restart_flag) break;  // workgroup-uniform: read after the barrier
    }
    if (!restart_flag) break;
    __syncthreads();  // everybody has seen the flag before thread 0 clears it for the next pass
  }
  if (wave == 0) {
    if (!failed) reg_decrease(P.opts, rho, drho);
    if (j == 0 && glive) {
      a.rho[b] = rho;
      a.drho[b] = drho;
      a.dV[b] = dV0;
      a.dV[(size_t)P.Bp + b] = dV1;
      a.bpfail[b] = failed ? 1 : 0;
    }
  }
}

}  // namespace to
