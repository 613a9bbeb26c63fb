// k_scan.h — backward pass parallel in TIME: fused expansion + block-parallel (associative scan) Riccati for the small models.
//
// At the headline shape (Cartpole, B = 1024) a batch step is latency: 100 sequential Riccati knots (0.88 us each on the
// cooperative kernel) and 100 sequential rollout knots, on a chip that holds one wave per SIMD.  The Riccati recursion is a
// scan: the conditional value function of a span of knots i..j, V(x_i, x_j), is parametrised by (A, b, C, eta, J) and two
// adjacent spans combine associatively (Särkkä & García-Fernández, "Temporal parallelization of dynamic programming and
// linear quadratic control", IEEE TAC 2023, Lemma 10; sign convention here V(x) = ½x'Jx + eta'x):
//   M = I + C1 J2
//   A = A2 M⁻¹ A1                     b = A2 M⁻¹ (b1 − C1 eta2) + b2           C = A2 M⁻¹ C1 A2' + C2
//   eta = A1' M⁻ᵀ (eta2 + J2 b1) + eta1                                        J = A1' M⁻ᵀ J2 A1 + J1
// One WAVE owns a trajectory; lane l owns the block of knots 2l, 2l+1:
//   1. expands its two knots in its own registers (chunk-mode duals, expand_lane_knot — the accepted step is written through);
//   2. builds their elements — stage knot: A = Ā, b = −B̄ R⁻¹ r, C = B̄ R⁻¹ B̄', eta = q, J = Q (diagonal cost blocks: what the
//      kernel is instantiated for; the terminal knot is (0, 0, 0, q_N, Q_N)) — and combines them into the block's element;
//   3. six rounds of a Hillis-Steele suffix scan across the lanes (ds_bpermute): lane l ends with the element of knots 2l..N-1,
//      whose (J, eta) IS the cost-to-go (S, s) at knot 2l;
//   4. takes (S, s) of the NEXT block from lane l+1 and walks its own two knots with the sequential recursion — the very
//      expressions of k_backward_lane — which yields the gains, the expected improvement and nothing else.
// 7 combines + 2 expansions + 2 Riccati steps per lane instead of 100 dependent knots: 30 us instead of 88 us per C2 step.
// The cost-to-go at the block boundaries carries the scan's rounding instead of the sequential recursion's: gains agree to
// 2e-15 (max-norm, relative); solving the C2 batch of 1024 with this arithmetic on the CPU leaves every iteration count and
// status unchanged and moves the converged states by <= 2.8e-7 — inside the 1e-6 band.
// Scope: rho == 0 on entry (no control regularisation: with rho > 0 the recursion S = Qxx + K'QuuK + ... uses the UNregularised
// Quu with gains of the regularised one and is no Riccati recursion any more), diagonal cost blocks, unconstrained problems,
// N <= 126, ne <= 4, m <= 2.  A trajectory with rho > 0, or whose Quu turns out not positive definite, takes the sequential pass
// in this same wave (every knot's expansion is parked in LDS; all lanes walk the horizon on broadcast operands): the arithmetic
// and the restart rule of k_backward_lane, no second launch.
#pragma once
#include "common.h"
#include "k_backward.h"
#include "k_expand.h"

namespace to {

template <int n>
struct ScanEl {
  static constexpr int NSY = n * (n + 1) / 2;
  static constexpr int NV = n * n + 2 * n + 2 * NSY;  // doubles per element
  double A[n * n], b[n], C[NSY], eta[n], J[NSY];
  __device__ __forceinline__ static constexpr int sy(int i, int j) { return i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }
  __device__ __forceinline__ void identity() {
#pragma unroll
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int j = 0; j < n; ++j) A[i * n + j] = (i == j) ? 1.0 : 0.0;
      b[i] = 0.0; eta[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NSY; ++i) { C[i] = 0.0; J[i] = 0.0; }
  }
};

// in-place solve M X = R (M n x n general, R n x r), no pivoting: M = I + (PSD)(PSD) has eigenvalues >= 1
template <int n, int r>
__device__ __forceinline__ void scan_lu_solve(double (&Mx)[n][n], double (&R)[n][r]) {
  double piv[n];
#pragma unroll
  for (int c = 0; c < n; ++c) {
    piv[c] = rcp_fast(Mx[c][c]);
#pragma unroll
    for (int i = c + 1; i < n; ++i) {
      const double f = Mx[i][c] * piv[c];
#pragma unroll
      for (int j = c + 1; j < n; ++j) Mx[i][j] -= f * Mx[c][j];
#pragma unroll
      for (int j = 0; j < r; ++j) R[i][j] -= f * R[c][j];
    }
  }
#pragma unroll
  for (int c = n - 1; c >= 0; --c) {
#pragma unroll
    for (int j = 0; j < r; ++j) {
      double v = R[c][j];
#pragma unroll
      for (int i = c + 1; i < n; ++i) v -= Mx[c][i] * R[i][j];
      R[c][j] = v * piv[c];
    }
  }
}

// o = e1 (earlier span) combined with e2 (the span right after it)
template <int n>
__device__ __forceinline__ void scan_combine(const ScanEl<n>& e1, const ScanEl<n>& e2, ScanEl<n>& o) {
  using E = ScanEl<n>;
  double Mm[n][n], Mt[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int t = 0; t < n; ++t) v += e1.C[E::sy(i, t)] * e2.J[E::sy(t, j)];
      Mm[i][j] = v;
      Mt[j][i] = v;
    }
  double R[n][2 * n + 1];
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < n; ++j) R[i][j] = e1.A[i * n + j];
    double v = e1.b[i];
#pragma unroll
    for (int t = 0; t < n; ++t) v -= e1.C[E::sy(i, t)] * e2.eta[t];
    R[i][n] = v;
#pragma unroll
    for (int j = 0; j < n; ++j) R[i][n + 1 + j] = e1.C[E::sy(i, j)];
  }
  scan_lu_solve<n, 2 * n + 1>(Mm, R);
  double XC[n][n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double v = 0.0, w = 0.0;
#pragma unroll
      for (int t = 0; t < n; ++t) { v += e2.A[i * n + t] * R[t][j]; w += e2.A[i * n + t] * R[t][n + 1 + j]; }
      o.A[i * n + j] = v;
      XC[i][j] = w;
    }
    double v = e2.b[i];
#pragma unroll
    for (int t = 0; t < n; ++t) v += e2.A[i * n + t] * R[t][n];
    o.b[i] = v;
  }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      double cij = e2.C[E::sy(i, j)], cji = cij;
#pragma unroll
      for (int t = 0; t < n; ++t) { cij += XC[i][t] * e2.A[j * n + t]; cji += XC[j][t] * e2.A[i * n + t]; }
      o.C[E::sy(i, j)] = 0.5 * (cij + cji);
    }
  double Y[n][n + 1];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double v = e2.eta[i];
#pragma unroll
    for (int t = 0; t < n; ++t) v += e2.J[E::sy(i, t)] * e1.b[t];
    Y[i][0] = v;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double w = 0.0;
#pragma unroll
      for (int t = 0; t < n; ++t) w += e2.J[E::sy(i, t)] * e1.A[t * n + j];
      Y[i][1 + j] = w;
    }
  }
  scan_lu_solve<n, n + 1>(Mt, Y);
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double v = e1.eta[i];
#pragma unroll
    for (int t = 0; t < n; ++t) v += e1.A[t * n + i] * Y[t][0];
    o.eta[i] = v;
  }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      double a = e1.J[E::sy(i, j)], c = a;
#pragma unroll
      for (int t = 0; t < n; ++t) { a += e1.A[t * n + i] * Y[t][1 + j]; c += e1.A[t * n + j] * Y[t][1 + i]; }
      o.J[E::sy(i, j)] = 0.5 * (a + c);
    }
}

__device__ __forceinline__ double scan_shfl(double v, int src_lane) {  // value of lane src_lane (0..63)
  return __shfl(v, src_lane);
}

// element of one knot from its expansion (diagonal cost block: Hd = its nc diagonal entries)
template <class M>
__device__ __forceinline__ bool scan_element(const double* Mk, const double* Hd, const double* g, bool terminal, bool beyond,
                                             ScanEl<M::ne>& e) {
  constexpr int m = M::m, ne = M::ne, nc = ne + m;
  using E = ScanEl<ne>;
  bool ok = true;
  double iR[m];
#pragma unroll
  for (int c = 0; c < m; ++c) { if (!terminal && not_positive(Hd[ne + c])) ok = false; iR[c] = rcp_fast(terminal ? 1.0 : Hd[ne + c]); }
#pragma unroll
  for (int i = 0; i < ne; ++i) {
#pragma unroll
    for (int j = 0; j < ne; ++j) e.A[i * ne + j] = terminal ? 0.0 : Mk[i * nc + j];
    double bb = 0.0;
#pragma unroll
    for (int c = 0; c < m; ++c) bb -= Mk[i * nc + ne + c] * (g[ne + c] * iR[c]);
    e.b[i] = terminal ? 0.0 : bb;
    e.eta[i] = g[i];
  }
#pragma unroll
  for (int j = 0; j < ne; ++j)
#pragma unroll
    for (int i = 0; i <= j; ++i) {
      double cc = 0.0;
#pragma unroll
      for (int c = 0; c < m; ++c) cc += Mk[i * nc + ne + c] * (Mk[j * nc + ne + c] * iR[c]);
      e.C[E::sy(i, j)] = terminal ? 0.0 : cc;
      e.J[E::sy(i, j)] = (i == j) ? Hd[i] : 0.0;
    }
  {  // a knot past the horizon: the neutral element (selects, not a branch: the lanes stay converged)
    E id;
    id.identity();
    double* pe = (double*)&e;
    const double* pi = (const double*)&id;
#pragma unroll
    for (int i = 0; i < E::NV; ++i) pe[i] = beyond ? pi[i] : pe[i];
  }
  return ok || beyond;
}

// One sequential Riccati step (k_backward_lane's knot, verbatim) from (S, s) = cost-to-go at knot k+1: gains row to pKk,
// expected-improvement terms, (S, s) <- cost-to-go at knot k.  false: Quu + rho I is not positive definite.
template <class M>
__device__ __forceinline__ bool scan_riccati_step(const double* Me, const double* Hd, const double* g, double (&S)[M::ne][M::ne], double (&s)[M::ne],
                                                  double rho, double* pKk, bool store, double& dv1o, double& dv2o) {
  constexpr int m = M::m, ne = M::ne, nc = ne + m;
  double Mk[ne][nc];
#pragma unroll
  for (int i = 0; i < ne; ++i)
#pragma unroll
    for (int j = 0; j < nc; ++j) Mk[i][j] = Me[i * nc + j];
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
      double t = (i == j) ? Hd[i] : 0.0;
#pragma unroll
      for (int r = 0; r < ne; ++r) t += Mk[r][i] * T[r][j];
      if (i < ne) Qxx[i][j] = t; else Qux[i - ne][j] = t;
    }
  }
#pragma unroll
  for (int q = 0; q < m; ++q)
#pragma unroll
    for (int p = 0; p < m; ++p) {
      double t = (p == q) ? Hd[ne + p] : 0.0;
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
    if (not_positive(sj)) pd_ok = false;
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
  if (store) {
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
  dv1o = dv1; dv2o = 0.5 * dv2;
#pragma unroll
  for (int i = 0; i < ne; ++i) {
#pragma unroll
    for (int j = 0; j < ne; ++j) S[i][j] = 0.5 * (Sn[i][j] + Sn[j][i]);
    s[i] = sn[i];
  }
  return pd_ok;
}

// grid = B workgroups of one wave: trajectory b = blockIdx.x.  LDS: the two knots' expansions of every lane, parked while the
// scan runs (registers hold two elements and the solver's workspace then).
template <class M, int FIXED_INTEG, int VAR>
__global__ void __launch_bounds__(64, 1) k_expand_backward_scan(KArgs a) {
  static_assert(!M::lie && M::ne <= 4 && M::m <= 2, "scan backward pass: small vector-space models");
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = ne + m, RSK = Gains<M>::RSK;
  constexpr int PK = ne * nc + 2 * nc;  // parked per knot: [A B], cost diagonal, gradient
  using E = ScanEl<ne>;
  using LL = LaneLay<M>;
  __shared__ double park[2 * PK * 64];
  const DevProblem& P = a.P;
  const int N = P.N;
  const int b = blockIdx.x, l = threadIdx.x;
  const int tile = b >> 6, lane = b & 63;
  double rho = a.rho[b], drho = a.drho[b];
  if (!a.active[b]) return;
  const int c = M::accept_write_through ? a.acc[b] : 0;
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  double* X0 = X_SLOT_PTR(a, b, 0);
  double* U0 = U_SLOT_PTR(a, b, 0);
  const bool wt = M::accept_write_through && c != 0;
  bool ok = true;
  E el;
  // ---- 1, 2: expansions of knots 2l, 2l+1 (lanes past the horizon ride along on the terminal knot), elements, block element
  {
    E e0, e1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kr = 2 * l + kk;
      const bool beyond = kr > N - 1;
      const int k = beyond ? N - 1 : kr;
      const bool terminal = (k == N - 1);
      double x[n], u[m];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = terminal ? 0.0 : EL(U, k * m + i);
      if (wt && !beyond) {
#pragma unroll
        for (int i = 0; i < n; ++i) EL(X0, k * n + i) = x[i];
        if (!terminal) {
#pragma unroll
          for (int i = 0; i < m; ++i) EL(U0, k * m + i) = u[i];
        }
      }
      double Mk[ne * nc], H[LL::NS], g[nc], Hd[nc];
#pragma unroll
      for (int i = 0; i < ne * nc; ++i) Mk[i] = 0.0;
      expand_lane_knot<M, FIXED_INTEG, VAR>(a, tile, lane, k, x, u, Mk, H, g);
#pragma unroll
      for (int j = 0; j < nc; ++j) { Hd[j] = (terminal && j >= ne) ? 1.0 : H[LL::sym(j, j)]; if (terminal && j >= ne) g[j] = 0.0; }
      double* pk = park + (size_t)kk * PK * 64 + l;
#pragma unroll
      for (int i = 0; i < ne * nc; ++i) pk[i * 64] = Mk[i];
#pragma unroll
      for (int j = 0; j < nc; ++j) { pk[(ne * nc + j) * 64] = Hd[j]; pk[(ne * nc + nc + j) * 64] = g[j]; }
      const bool eok = scan_element<M>(Mk, Hd, g, terminal, beyond, kk == 0 ? e0 : e1);
      if (!eok && !beyond) ok = false;
    }
    scan_combine<ne>(e0, e1, el);
  }
  // ---- the scan path proper: no regularisation pending (wave-uniform)
  double dV0 = 0.0, dV1 = 0.0;
  double* pK = a.Kt + ((size_t)b * (N - 1)) * RSK;
  bool done = false, failed = false;
  // (a control cost entry that is not positive admits no element — C = B R⁻¹ B' — but the sequential pass may still find
  // Quu = R + B'SB positive definite: such a trajectory walks sequentially with its rho untouched)
  const bool elements_ok = __ballot(!ok) == 0;
  if (rho == 0.0 && elements_ok) {
    // ---- 3: suffix scan across the lanes: after round d lane l holds the element of blocks l .. l + 2d - 1
    const int NB = (N + 1) / 2;  // blocks on the horizon
  #pragma unroll 1
    for (int d = 1; d < NB; d <<= 1) {
      E pe;
      const int src = l + d;
      const bool has = src < 64;
      const int sl = has ? src : 63;
      {
        const double* pv = (const double*)&el;
        double* po = (double*)&pe;
  #pragma unroll
        for (int i = 0; i < E::NV; ++i) po[i] = scan_shfl(pv[i], sl);
      }
      if (!has) pe.identity();
      E out;
      scan_combine<ne>(el, pe, out);
      el = out;
    }
    // ---- 4: cost-to-go behind this lane's block, then its own knots sequentially
    double S[ne][ne], s[ne];
    {
      const int src = (l + 1 < 64) ? l + 1 : 63;
      double Jn[E::NSY], en[ne];
  #pragma unroll
      for (int i = 0; i < E::NSY; ++i) Jn[i] = scan_shfl(el.J[i], src);
  #pragma unroll
      for (int i = 0; i < ne; ++i) en[i] = scan_shfl(el.eta[i], src);
  #pragma unroll
      for (int i = 0; i < ne; ++i) {
  #pragma unroll
        for (int j = 0; j < ne; ++j) S[i][j] = Jn[E::sy(i, j)];
        s[i] = en[i];
      }
    }
  #pragma unroll
    for (int kk = 1; kk >= 0; --kk) {
      const int k = 2 * l + kk;
      double Mk[ne * nc], Hd[nc], g[nc];
      const double* pk = park + (size_t)kk * PK * 64 + l;
  #pragma unroll
      for (int i = 0; i < ne * nc; ++i) Mk[i] = pk[i * 64];
  #pragma unroll
      for (int j = 0; j < nc; ++j) { Hd[j] = pk[(ne * nc + j) * 64]; g[j] = pk[(ne * nc + nc + j) * 64]; }
      if (k == N - 1) {  // the terminal knot opens the recursion of its block: S = Q_N, s = q_N
  #pragma unroll
        for (int i = 0; i < ne; ++i) {
  #pragma unroll
          for (int j = 0; j < ne; ++j) S[i][j] = (i == j) ? Hd[i] : 0.0;
          s[i] = g[i];
        }
      }
      const bool stage = k < N - 1;  // (lanes / knots past the horizon compute along on finite data and store nothing)
      double dv1, dv2;
      double St[ne][ne], st[ne];
  #pragma unroll
      for (int i = 0; i < ne; ++i) {
  #pragma unroll
        for (int j = 0; j < ne; ++j) St[i][j] = S[i][j];
        st[i] = s[i];
      }
      const bool pd = scan_riccati_step<M>(Mk, Hd, g, St, st, 0.0, pK + (size_t)(stage ? k : 0) * RSK, stage, dv1, dv2);
      if (stage) {
        if (!pd) ok = false;
        dV0 += dv1; dV1 += dv2;
  #pragma unroll
        for (int i = 0; i < ne; ++i) {
  #pragma unroll
          for (int j = 0; j < ne; ++j) S[i][j] = St[i][j];
          s[i] = st[i];
        }
      }
    }
    // expected improvement: sum over the knots (fixed butterfly order)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { dV0 += __shfl_xor(dV0, o); dV1 += __shfl_xor(dV1, o); }
    if (__ballot(!ok) == 0) done = true;
    else {  // some Quu is not positive definite: what the sequential pass does at its first failure, then its restart below
      reg_increase(P.opts, rho, drho);
      if (rho > P.opts.bp_reg_max) failed = true;
    }
  }
  // ---- regularisation pending (rho > 0: the regularised recursion is no Riccati recursion, no scan) or raised just now: the
  // sequential pass, in this same wave.  Every knot's expansion is parked in LDS; ALL lanes walk the horizon together on the same
  // (broadcast) operands — identical values in every lane, EXEC full, no exchange — lane 0 stores.  k_backward_lane's loop,
  // restarts included; ~95 us for 100 knots, on a path the C2 solve never takes.
  WAVE_SYNC();  // the parked expansions of every lane, visible to every lane
  while (!done && !failed) {
    double S[ne][ne], s[ne];
    bool restart = false;
    dV0 = 0.0; dV1 = 0.0;
    for (int k = N - 1; k >= 0; --k) {
      double Mk[ne * nc], Hd[nc], g[nc];
      const double* pk = park + (size_t)(k & 1) * PK * 64 + (k >> 1);
#pragma unroll
      for (int i = 0; i < ne * nc; ++i) Mk[i] = pk[i * 64];
#pragma unroll
      for (int j = 0; j < nc; ++j) { Hd[j] = pk[(ne * nc + j) * 64]; g[j] = pk[(ne * nc + nc + j) * 64]; }
      if (k == N - 1) {
#pragma unroll
        for (int i = 0; i < ne; ++i) {
#pragma unroll
          for (int j = 0; j < ne; ++j) S[i][j] = (i == j) ? Hd[i] : 0.0;
          s[i] = g[i];
        }
        continue;
      }
      double dv1, dv2;
      const bool pd = scan_riccati_step<M>(Mk, Hd, g, S, s, rho, pK + (size_t)k * RSK, l == 0, dv1, dv2);
      if (!pd) {  // (wave-uniform: every lane holds the same numbers)
        reg_increase(P.opts, rho, drho);
        if (rho > P.opts.bp_reg_max) failed = true; else restart = true;
        break;
      }
      dV0 += dv1; dV1 += dv2;
    }
    if (!restart) done = !failed;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  if (l == 0) {
    a.rho[b] = rho; a.drho[b] = drho;
    a.dV[b] = dV0; a.dV[(size_t)P.Bp + b] = dV1;
    a.bpfail[b] = failed ? 1 : 0;
  }
}

}  // namespace to
