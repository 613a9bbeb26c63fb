// k_backward.h — backward Riccati recursion.
#pragma once
#include "common.h"

namespace to {

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

// HD: diagonal cost blocks (KArgs::h_diag) — one row of Hc per knot, the lane's own diagonal entry.
template <class M, bool HD>
__global__ void __launch_bounds__(64) k_backward_coop(KArgs a) {
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
  constexpr int HR = HD ? 1 : nc;  // rows of Hc per knot
  const double* Hc = COL_PTR(a.Hc, N * HR);
  const double* gc = COL_PTR(a.gc, N);
  constexpr int RSK = Gains<M>::RSK;
  double* pK = a.Kt + ((size_t)(b < P.B ? b : 0) * (N - 1)) * RSK;  // this trajectory's gains rows (trajectory-major)
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  int k = N - 2;
  bool init = true, fresh = true;
  double Mn[ne], Hn[HR], gn = 0.0;  // prefetched column of the next knot
  const double *pMk = Mc, *pHk = Hc, *pgk = gc;
  double* pKk = pK;
  while (true) {
    if (init) {  // (re)start: S = Qxx_N, s = qx_N
      fresh = true;
      {
        double Sc[ne];
        if constexpr (HD) {
          const double hd = EL(Hc, N - 1);
#pragma unroll
          for (int i = 0; i < ne; ++i) Sc[i] = (i == j) ? hd : 0.0;
        } else {
#pragma unroll
          for (int i = 0; i < ne; ++i) Sc[i] = EL(Hc, (N - 1) * nc + i);
        }
        const double s0 = EL(gc, N - 1);
        if (j < ne) {
#pragma unroll
          for (int i = 0; i < ne; ++i) S_[i * ne + j] = Sc[i];
          sl[j] = s0;
        }
      }
      dV0 = 0.0; dV1 = 0.0; k = N - 2; init = false;
      // per-knot pointers walk backwards with the recursion: constant offsets instead of 64-bit address arithmetic per load
      pMk = Mc + (size_t)(N - 2) * ne * 64; pHk = Hc + (size_t)(N - 2) * HR * 64; pgk = gc + (size_t)(N - 2) * 64;
      pKk = pK + (size_t)(N - 2) * RSK;
      WAVE_SYNC();
    }
    if (k < 0) break;
    // 1. own column of [Ā B̄] and of the cost blocks (fetched one knot ahead: nothing else hides the load latency)
    double Mj[ne], Hj[nc], gj;
    if (fresh) {
#pragma unroll
      for (int i = 0; i < ne; ++i) Mn[i] = EL(pMk, i);
#pragma unroll
      for (int i = 0; i < HR; ++i) Hn[i] = EL(pHk, i);
      gn = EL(pgk, 0);
      fresh = false;
    }
#pragma unroll
    for (int i = 0; i < ne; ++i) Mj[i] = Mn[i];
#pragma unroll
    for (int i = 0; i < nc; ++i) Hj[i] = HD ? ((i == j) ? Hn[0] : 0.0) : Hn[HD ? 0 : i];
    gj = gn;
    if constexpr (ne > 6) fresh = true;  // large models: the extra live registers cost more than the latency they hide
    else if (k > 0) {
#pragma unroll
      for (int i = 0; i < ne; ++i) Mn[i] = (pMk - ne * 64)[(size_t)i * 64];
#pragma unroll
      for (int i = 0; i < HR; ++i) Hn[i] = (pHk - HR * 64)[(size_t)i * 64];
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
      iL[q] = rsqrt_fast(sj);  // the diagonal of L is only ever divided by: 10 dependent operations instead of the ~30
      Lc[q][q] = sj * iL[q];   // of an IEEE sqrt followed by a reciprocal (as in the MFMA kernel)
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
      for (int r = 0; r < m; ++r) { if constexpr (m > 1) Kf[r * ne + j] = Kj[r]; if (glive) pKk[r * (ne + 1) + j] = Kj[r]; }
    }
    if (j == 0 && glive) {
#pragma unroll
      for (int r = 0; r < m; ++r) pKk[r * (ne + 1) + ne] = dk[r];
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
    pMk -= ne * 64; pHk -= HR * 64; pgk -= 64; pKk -= RSK;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  if (j == 0 && glive) {
    a.rho[b] = rho;
    a.drho[b] = drho;
    a.dV[b] = dV0;
    a.dV[(size_t)P.Bp + b] = dV1;
    a.bpfail[b] = failed ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------ lane backward pass
// Small models (ne + m <= 6): ONE LANE per trajectory, the whole recursion in that lane's registers — no LDS, no
// barrier, no operand exchanged between lanes.  The cooperative kernel above spends a knot of a 4-state model mostly in
// its four LDS exchange rounds (0.78 us per knot on the Cartpole at one wave per SIMD); a single lane runs the same
// ~230 FMAs back to back.  Latency, not throughput, is what a batch of 1 024 needs: it is 16 waves either way.
// "Lane layout" of the expansion (k_expand.h LAY = 3), tile = b >> 6, l = b & 63, kept in the Mc / Hc / gc allocations:
//   M[i][j]        at Mc[((tile*(N-1) + k)*ne*nc + i*nc + j)*64 + l]
//   H[i][j], i<=j  at Hc[((tile*N + k)*NS + j*(j+1)/2 + i)*64 + l]      NS = nc(nc+1)/2 (upper triangle, column by column)
//   g[j]           at gc[((tile*N + k)*nc + j)*64 + l]
// so a knot's operands are nc*(ne + (nc+1)/2 + 1) fully coalesced 512-byte rows per wave.
template <class M>
struct LaneLay {
  static constexpr int ne = M::ne, m = M::m, nc = ne + m, NS = nc * (nc + 1) / 2, EM = ne * nc;
  __host__ __device__ static constexpr int sym(int i, int j) { return i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }
};

template <class M>
__global__ void __launch_bounds__(64) k_backward_lane(KArgs a) {
  constexpr int m = M::m, ne = M::ne, nc = ne + m, RSK = Gains<M>::RSK;
  using L = LaneLay<M>;
  constexpr int NS = L::NS, EM = L::EM;
  const DevProblem& P = a.P;
  const int N = P.N;
  const int tile = blockIdx.x, lane = threadIdx.x, b = tile * 64 + lane;  // b < Bp always (Bp is a multiple of 64)
  // lanes without a trajectory to solve run along on their own (valid) data and only their stores are predicated
  const bool live = (b < P.B) && a.active[b] != 0;
  if (__ballot(live) == 0) return;
  const double* Ml = a.Mc + ((size_t)tile * (size_t)(N - 1) * EM) * 64 + lane;
  const double* Hl = a.Hc + ((size_t)tile * (size_t)N * NS) * 64 + lane;
  const double* gl = a.gc + ((size_t)tile * (size_t)N * nc) * 64 + lane;
  double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false, init = true, fresh = true;
  int k = N - 2;
  double S[ne][ne], s[ne];
  double Mn[EM], Hn[NS], gn[nc];  // the next knot's operands, fetched one knot ahead
  const double *pMk = Ml, *pHk = Hl, *pgk = gl;
  double* pKk = pK;
  while (true) {
    if (init) {  // (re)start: S = Qxx_N, s = qx_N
#pragma unroll
      for (int i = 0; i < ne; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) S[i][j] = EL(Hl, (size_t)(N - 1) * NS + L::sym(i, j));
        s[i] = EL(gl, (size_t)(N - 1) * nc + i);
      }
      dV0 = 0.0; dV1 = 0.0; k = N - 2; init = false; fresh = true;
      pMk = Ml + (size_t)(N - 2) * EM * 64; pHk = Hl + (size_t)(N - 2) * NS * 64; pgk = gl + (size_t)(N - 2) * nc * 64;
      pKk = pK + (size_t)(N - 2) * RSK;
    }
    if (k < 0) break;
    if (fresh) {
#pragma unroll
      for (int e = 0; e < EM; ++e) Mn[e] = EL(pMk, e);
#pragma unroll
      for (int e = 0; e < NS; ++e) Hn[e] = EL(pHk, e);
#pragma unroll
      for (int e = 0; e < nc; ++e) gn[e] = EL(pgk, e);
      fresh = false;
    }
    double Mk[ne][nc], H[NS], g[nc];
#pragma unroll
    for (int i = 0; i < ne; ++i)
#pragma unroll
      for (int j = 0; j < nc; ++j) Mk[i][j] = Mn[i * nc + j];
#pragma unroll
    for (int e = 0; e < NS; ++e) H[e] = Hn[e];
#pragma unroll
    for (int e = 0; e < nc; ++e) g[e] = gn[e];
    if (k > 0) {
#pragma unroll
      for (int e = 0; e < EM; ++e) Mn[e] = (pMk - (size_t)EM * 64)[(size_t)e * 64];
#pragma unroll
      for (int e = 0; e < NS; ++e) Hn[e] = (pHk - (size_t)NS * 64)[(size_t)e * 64];
#pragma unroll
      for (int e = 0; e < nc; ++e) gn[e] = (pgk - (size_t)nc * 64)[(size_t)e * 64];
    }
    // T = S M;  Q = H + M'T (state columns in full, of the control columns only the Quu block);  gq = g + M's
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
    // Cholesky of Quu + rho I
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
      if (!(sj > 0.0) && live) pd_ok = false;  // lanes that only ride along never restart
      iL[q] = rsqrt_fast(sj);  // the diagonal of L is only ever divided by: 10 dependent operations instead of the ~30
      Lc[q][q] = sj * iL[q];   // of an IEEE sqrt followed by a reciprocal (as in the MFMA kernel)
#pragma unroll
      for (int i = q + 1; i < m; ++i) {
        double t = Lc[i][q];
#pragma unroll
        for (int r = 0; r < q; ++r) t -= Lc[i][r] * Lc[q][r];
        Lc[i][q] = t * iL[q];
      }
    }
    if (!pd_ok) {
      reg_increase(P.opts, rho, drho);
      if (rho > P.opts.bp_reg_max) { failed = true; break; }
      init = true;
      continue;
    }
    // gains K = -(LL')^-1 Qux (column by column), d = -(LL')^-1 Qu
    double Kg[m][ne], dk[m];
#pragma unroll
    for (int c = 0; c <= ne; ++c) {
      double col[m];
#pragma unroll
      for (int i = 0; i < m; ++i) col[i] = (c < ne) ? Qux[i][c < ne ? c : 0] : gq[ne + i];
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
      for (int i = 0; i < m; ++i) { if (c < ne) Kg[i][c < ne ? c : 0] = -col[i]; else dk[i] = -col[i]; }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < m; ++r) {
#pragma unroll
        for (int j = 0; j < ne; ++j) pKk[r * (ne + 1) + j] = Kg[r][j];
        pKk[r * (ne + 1) + ne] = dk[r];
      }
    }
    // cost-to-go with the un-regularised Quu:  S' = Qxx + K'(Quu K + Qux) + Qux'K,  s' = Qx + K'(Quu d + Qu) + Qux'd
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
    --k;
    pMk -= (size_t)EM * 64; pHk -= (size_t)NS * 64; pgk -= (size_t)nc * 64; pKk -= RSK;
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

// ------------------------------------------------------------------------------------------------ MFMA backward pass
// One WAVE per trajectory; the per-knot blocks live in the result layout of v_mfma_f64_16x16x4_f64 ("tangent-matrix
// layout"): a 16x16 matrix X sits in 4 registers, lane (g, c) = hardware lane g*16 + c holds X[g + 4r][c] in register r.
// Tangent index of a state direction i is i, of a control direction j it is NEP + j (NEP = ne rounded up to 4), so the
// control rows [Qux Quu] are exactly register RS = NEP/4 with control j in lane group j.  Register r of a matrix in this
// layout IS K-slice r of it as an MFMA operand (B operand directly, A operand as its transpose), hence
//     T  = S M        = sum_s mfma(S_s, M_s)       (S symmetric: its slice s as A operand is register s)
//     Hq = H + M' T   = sum_s mfma(M_s, T_s)       (T comes out of the first product already in operand form)
//     S' = Hq_xx + K' W + Qux' K                    (two more MFMAs with one-register operands)
// with no LDS traffic for the two large products; the m x m factorisation and the vectors go through two small LDS
// exchanges per knot.  FP64 MFMA runs at the vector rate on gfx950 — the point is not peak but that a trajectory's
// 12x16 blocks are spread over 64 lanes (3-4 registers each) instead of 16 lanes holding 400 doubles and exchanging
// every operand through LDS: 256 VGPR + 248 AGPR at one wave per SIMD before, ~100 registers now.
typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef TO_BWD_WAVES
#define TO_BWD_WAVES 1  // minimum waves per SIMD the MFMA backward kernel is compiled for (register cap 512 / waves)
#endif

template <class M>
struct Tm {
  static constexpr int ne = M::ne, m = M::m;
  static constexpr int NEP = (ne + 3) / 4 * 4;  // tangent index of control 0
  static constexpr int RS = NEP / 4;            // registers that hold state rows
  static constexpr int NR = RS + 1;             // ... plus the control rows
  static constexpr bool fits = NEP + m <= 16 && m <= 4;  // one 16 x 16 tile per knot (asserted where the layout is used: k_backward_mfma,
                                                         // the tangent-matrix expansions; models with more controls never instantiate those)
};

// Compact cost block (KArgs::h_compact): when the Q-function cost block is block-diagonal — diagonal Qxx (plus the 3x3
// attitude block of Lie-group models), no Qux, any Quu — every lane of the layout needs at most ONE entry of its column:
// row compact_row(g, c) (-1: none).  One 64-lane row per knot instead of NR.
template <class M>
__host__ __device__ inline int compact_row(int g, int c) {
  constexpr int ne = Tm<M>::ne, m = Tm<M>::m, NEP = Tm<M>::NEP;
  if (c >= NEP) return (c < NEP + m && g < m) ? NEP + g : -1;
  if (c >= ne) return -1;
  if (M::lie && c >= 3 && c < 6) { const int i = (g == 3) ? 3 : (g == 0) ? 4 : (g == 1) ? 5 : -1; return i; }
  return (c % 4 == g) ? c : -1;
}

template <class M>
struct MfmaLds {  // doubles; rows padded to 17 so that transposed reads spread over the banks
  static constexpr int oC = 0;             // control rows [Qux Quu][a][c]   4 x 17
  static constexpr int oG = oC + 4 * 17;   // Q-function gradient [c]         16
  static constexpr int oS = oG + 16;       // S' staging [i][c]               16 x 17
  static constexpr int oV = oS + 16 * 17;  // s' [c]                          16
  static constexpr int size = oV + 16;
};

template <class M, bool HC>
__global__ void __launch_bounds__(64, TO_BWD_WAVES) k_backward_mfma(KArgs a) {
  constexpr int m = M::m, ne = M::ne, NEP = Tm<M>::NEP, RS = Tm<M>::RS, NR = Tm<M>::NR, RSK = Gains<M>::RSK;
  static_assert(Tm<M>::fits, "one 16x16 tile per knot");
  constexpr int HR = HC ? 1 : NR;  // rows of the cost block per knot
  using L = MfmaLds<M>;
  __shared__ double lds[L::size];
  const DevProblem& P = a.P;
  const int N = P.N;
  const int hw = threadIdx.x, g = hw >> 4, c = hw & 15;
  int b = blockIdx.x;  // one trajectory per wave: number blockIdx.x of the batch, or — active-list compaction — of this step's list
  if (a.compact) {
    if ((int)blockIdx.x >= a.acount[a.step & 1]) return;
    b = a.alist[(size_t)(a.step & 1) * P.Bp + blockIdx.x];
  }
  if (b >= P.B || !a.active[b]) return;  // wave-uniform
  double* Cl = lds + L::oC;
  double* Gl = lds + L::oG;
  double* Sl = lds + L::oS;
  double* Vl = lds + L::oV;
  const double* Mt = a.Mt + ((size_t)b * (N - 1)) * RS * 64 + hw;
  const double* Ht = a.Ht + ((size_t)b * N) * HR * 64 + hw;
  const double* gt = a.gt + ((size_t)b * N) * 16;
  double* Kt = a.Kt + ((size_t)b * (N - 1)) * RSK;
  const bool ccol = c < ne;            // this lane's column is a state direction
  bool rowok[RS];                      // register r of this lane holds a state row
#pragma unroll
  for (int r = 0; r < RS; ++r) rowok[r] = (g + 4 * r) < ne;
  const int csel = compact_row<M>(g, c);
  const int rsel = csel < 0 ? -1 : csel / 4;
  double rho = a.rho[b], drho = a.drho[b];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  auto load_cost = [&](int k, v4d& H) {  // cost block of knot k in the layout (padding lanes hold zeros in memory)
    if constexpr (HC) {
      const double v = Ht[(size_t)k * 64];
      H = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < NR; ++r) H[r] = (r == rsel) ? v : 0.0;
    } else {
      H = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < NR; ++r) H[r] = Ht[((size_t)k * NR + r) * 64];
    }
  };
  while (true) {
    // (re)start: S = Qxx_N, s = qx_N
    double S[RS], srow[RS];
    {
      v4d H;
      load_cost(N - 1, H);
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        S[r] = (rowok[r] && ccol) ? H[r] : 0.0;
        srow[r] = rowok[r] ? gt[(size_t)(N - 1) * 16 + g + 4 * r] : 0.0;
      }
    }
    dV0 = 0.0; dV1 = 0.0;
    bool restart = false;
    // operands of the knot being processed are fetched one knot ahead (registers are plentiful in this layout)
    double Mn[RS], gn, Hn[HR];  // the cost block is prefetched raw (one value per lane when compact) and expanded at its use
#pragma unroll
    for (int r = 0; r < RS; ++r) Mn[r] = Mt[((size_t)(N - 2) * RS + r) * 64];
#pragma unroll
    for (int r = 0; r < HR; ++r) Hn[r] = Ht[((size_t)(N - 2) * HR + r) * 64];
    gn = gt[(size_t)(N - 2) * 16 + c];
    for (int k = N - 2; k >= 0; --k) {
      double Mr[RS];
#pragma unroll
      for (int r = 0; r < RS; ++r) Mr[r] = Mn[r];
      v4d Hq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < NR; ++r) Hq[r] = HC ? ((r == rsel) ? Hn[0] : 0.0) : Hn[HC ? 0 : r];
      const double gcol = gn;
      if (k > 0) {
#pragma unroll
        for (int r = 0; r < RS; ++r) Mn[r] = Mt[((size_t)(k - 1) * RS + r) * 64];
#pragma unroll
        for (int r = 0; r < HR; ++r) Hn[r] = Ht[((size_t)(k - 1) * HR + r) * 64];
        gn = gt[(size_t)(k - 1) * 16 + c];
      }
      // 1. T = S M,  Hq = H + M' T  (registers of T beyond RS stay zero: S has no rows there)
      v4d T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < RS; ++s) T = __builtin_amdgcn_mfma_f64_16x16x4f64(S[s], Mr[s], T, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < RS; ++s) Hq = __builtin_amdgcn_mfma_f64_16x16x4f64(Mr[s], T[s], Hq, 0, 0, 0);
      // 2. Q-function gradient: g_c + sum_i M[i][c] s[i]; the rows of a column are spread over the four lane groups
      double pg = 0.0;
#pragma unroll
      for (int r = 0; r < RS; ++r) pg += Mr[r] * srow[r];
      pg += __shfl_xor(pg, 16);
      pg += __shfl_xor(pg, 32);
      const double Qg = gcol + pg;
      // 3. publish the control rows and the gradient; every lane reads Quu, Qu (wave-uniform) and its own column of Qux
      const double hctl = Hq[RS];  // [Qux Quu][g][c]
      Cl[g * 17 + c] = hctl;
      if (g == 0) Gl[c] = Qg;
      WAVE_SYNC();
      double Quu[m][m], Qu[m], qx[m];
#pragma unroll
      for (int r = 0; r < m; ++r) {
#pragma unroll
        for (int q = 0; q <= r; ++q) { Quu[r][q] = Cl[r * 17 + NEP + q]; Quu[q][r] = Quu[r][q]; }  // lower triangle (Hq is symmetric up to rounding)
        Qu[r] = Gl[NEP + r];
        qx[r] = ccol ? Cl[r * 17 + c] : 0.0;
      }
      // 4. factor Quu + rho I (same arithmetic in every lane), restart with more regularisation when not positive definite
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
        if (!(sj > 0.0)) pd_ok = false;
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
      if (!pd_ok) {  // wave-uniform
        reg_increase(P.opts, rho, drho);
        if (rho > P.opts.bp_reg_max) failed = true; else restart = true;
        break;
      }
      // 5. gains: own column of K = -(LL')^-1 Qux.  The feed-forward d = -(LL')^-1 Qu is one more column of the same
      // solve: the lanes of tangent column NEP (first control column: no K column of their own) carry Qu through it, and
      // everybody picks the result up with v_readlane — the solve is not run a second time in all 64 lanes.
      const bool dcol = (c == NEP);
      double Kc[m], dk[m];
      {
        double col[m];
#pragma unroll
        for (int i = 0; i < m; ++i) col[i] = dcol ? Qu[i] : qx[i];
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
        for (int i = 0; i < m; ++i) {
          const double v = -col[i];
          dk[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), NEP), __builtin_amdgcn_readlane(__double2loint(v), NEP));
          Kc[i] = ccol ? v : 0.0;
        }
      }
      const double kown = pick<m>(Kc, g);  // K[g][c]: K as MFMA operand (B: K, A: K')
      if (g < m && c <= ne) Kt[(size_t)k * RSK + g * (ne + 1) + c] = ccol ? kown : pick<m>(dk, g);  // the knot's whole gains row
      // 6. cost-to-go with the un-regularised Quu:  S' = Qxx + K'(Quu K + Qux) + Qux' K,  s' = Qx + K'(Quu d + Qu) + Qux' d
      double Wc[m], qd[m], Qd[m];  // Qd = Quu d serves both Quu d + Qu and d' Quu d
#pragma unroll
      for (int r = 0; r < m; ++r) {
        double t = qx[r], t2 = 0.0;
#pragma unroll
        for (int q = 0; q < m; ++q) { t += Quu[r][q] * Kc[q]; t2 += Quu[r][q] * dk[q]; }
        Wc[r] = t; Qd[r] = t2; qd[r] = t2 + Qu[r];
      }
      const double wown = pick<m>(Wc, g);                  // W[g][c]
      const double qown = (g < m && ccol) ? hctl : 0.0;    // Qux[g][c]
      v4d Sn = Hq;
      Sn = __builtin_amdgcn_mfma_f64_16x16x4f64(kown, wown, Sn, 0, 0, 0);  // += K' W
      Sn = __builtin_amdgcn_mfma_f64_16x16x4f64(qown, kown, Sn, 0, 0, 0);  // += Qux' K
      double snew = Qg;
#pragma unroll
      for (int r = 0; r < m; ++r) snew += Kc[r] * qd[r];
#pragma unroll
      for (int r = 0; r < m; ++r) snew += qx[r] * dk[r];
      double dv1 = 0.0, dv2 = 0.0;
#pragma unroll
      for (int r = 0; r < m; ++r) {
        dv1 += dk[r] * Qu[r];
        dv2 += dk[r] * Qd[r];
      }
      dV0 += dv1;
      dV1 += 0.5 * dv2;
      // 7. S <- (S' + S'^T)/2 and s <- s' re-indexed by rows, through LDS
#pragma unroll
      for (int r = 0; r < RS; ++r) Sl[(g + 4 * r) * 17 + c] = Sn[r];
      if (g == 0) Vl[c] = snew;
      WAVE_SYNC();
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        const double st = Sl[c * 17 + g + 4 * r];
        S[r] = (rowok[r] && ccol) ? 0.5 * (Sn[r] + st) : 0.0;
        srow[r] = rowok[r] ? Vl[g + 4 * r] : 0.0;
      }
      WAVE_SYNC();
    }
    if (!restart) break;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  if (hw == 0) {
    a.rho[b] = rho;
    a.drho[b] = drho;
    a.dV[b] = dV0;
    a.dV[(size_t)P.Bp + b] = dV1;
    a.bpfail[b] = failed ? 1 : 0;
  }
}

}  // namespace to
