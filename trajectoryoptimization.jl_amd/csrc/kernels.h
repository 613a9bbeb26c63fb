// kernels.h — HIP kernels of the batched iLQR / AL hot path for gfx950 (MI355X).
//
// Data layout: batch-fastest structure-of-arrays.  Element (k, i) of a per-knot vector of
// trajectory b lives at base[(k*dim + i)*Bp + b]; a wave's 64 lanes are 64 consecutive trajectories, so
// every load/store below is one fully coalesced 512-byte transaction.  One lane == one trajectory for
// the sequential recursions (rollout, backward Riccati, forward line search); the expansion kernel adds
// two more grid axes (knot, direction) because it is embarrassingly parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "models.h"
#include "problem_dev.h"

namespace to {

struct KArgs {
  DevProblem P;
  double* X[2];   // [N][n][Bp]   double-buffered; cur[b] selects the nominal trajectory
  double* U[2];   // [N-1][m][Bp]
  double* x0;     // [n][Bp]
  int* cur;       // [Bp]
  double *A, *Bm;                  // [N-1][ne][ne][Bp], [N-1][ne][m][Bp]
  double *Qxx, *Quu, *Qux, *qx, *qu;  // [N][ne][ne], [N][m][m], [N][m][ne], [N][ne], [N][m]   (x Bp)
  double *K, *d;                   // [N-1][m][ne][Bp], [N-1][m][Bp]
  double *lam, *mu;                // [n_duals][Bp], [n_cons][Bp]
  double *J, *dJ, *grad, *rho, *drho, *dV, *cmax, *Jout;  // [Bp] (dV: [2][Bp])
  int *status, *iterations, *it_inner, *outer, *dJzero, *ls_index, *active, *budget, *bpfail;
  int* counter;   // [steps] number of trajectories still active after each batch step
  int al_mode;    // 0: iLQR, 1: AL-iLQR
  int control;    // 1: run the solver state machine at the end of the forward pass; 0: phase API
  int step;
};

#define TO_IDX(k, dim, i) (((size_t)(k) * (dim) + (i)) * Bp + b)

template <class M>
__device__ __forceinline__ void load_vec(const double* base, int k, int dim_total, int Bp, int b, double* out, int count) {
  for (int i = 0; i < count; ++i) out[i] = base[TO_IDX(k, dim_total, i)];
}

// objective (+AL) value of one knot.  u must be zeros at the terminal knot (the reference evaluates the
// terminal cost with the knot's zero control; src/cost_functions.jl:92-94, test/objective_tests.jl:129).
template <class M>
__device__ __forceinline__ double knot_cost(const KArgs& a, int k, const double* x, const double* u, int b, bool with_al) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const DevProblem& P = a.P;
  const int Bp = P.Bp;
  double Jk = cost_eval<n, m>(P.costs[P.cost_index[k]], x, u);
  if (P.opts.cost_dt_scaling && k < P.N - 1) Jk *= P.dt[k];
  if (with_al && P.n_cons > 0) {
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    double Ja = 0.0;
    for (int ci = 0; ci < P.n_cons; ++ci) {
      const DevCon& K = P.cons[ci];
      if (k < K.k1 || k > K.k2) continue;
      const double* lam = a.lam + ((size_t)(K.dual_off + (long long)(k - K.k1) * K.p)) * Bp + b;
      Ja += al_term<nz>(K, z, lam, (size_t)Bp, a.mu[(size_t)ci * Bp + b]);
    }
    Jk += Ja;
  }
  return Jk;
}

template <class M>
__device__ __forceinline__ double knot_violation(const KArgs& a, int k, const double* x, const double* u) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const DevProblem& P = a.P;
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = u[i];
  double vmax = 0.0;
  for (int ci = 0; ci < P.n_cons; ++ci) {
    const DevCon& K = P.cons[ci];
    if (k < K.k1 || k > K.k2) continue;
    const double v = con_violation<nz>(K, z);
    if (!(v <= vmax)) vmax = v;
  }
  return vmax;
}

// whole-trajectory pass over the NOMINAL trajectory: cost (with or without AL), max violation, optional dual update
template <class M>
__device__ __forceinline__ void trajectory_pass(const KArgs& a, int b, bool with_al, bool do_dual_update, double* J_out, double* cmax_out) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const DevProblem& P = a.P;
  const int Bp = P.Bp, N = P.N;
  const int c = a.cur[b];
  const double* X = a.X[c];
  const double* U = a.U[c];
  double J = 0.0, cmax = 0.0;
  for (int k = 0; k < N; ++k) {
    double x[n], u[m];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = X[TO_IDX(k, n, i)];
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? U[TO_IDX(k, m, i)] : 0.0;
    if (do_dual_update) {
      double z[nz];
#pragma unroll
      for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
      for (int i = 0; i < m; ++i) z[n + i] = u[i];
      for (int ci = 0; ci < P.n_cons; ++ci) {
        const DevCon& K = P.cons[ci];
        if (k < K.k1 || k > K.k2) continue;
        double* lam = a.lam + ((size_t)(K.dual_off + (long long)(k - K.k1) * K.p)) * Bp + b;
        con_dual_update<nz>(K, z, lam, (size_t)Bp, a.mu[(size_t)ci * Bp + b], P.opts.dual_max);
      }
    }
    if (cmax_out && P.n_cons > 0) { const double v = knot_violation<M>(a, k, x, u); if (!(v <= cmax)) cmax = v; }
    if (J_out) J += knot_cost<M>(a, k, x, u, b, with_al);
  }
  if (J_out) *J_out = J;
  if (cmax_out) *cmax_out = cmax;
}

// ------------------------------------------------------------------------------------------------ rollout!
template <class M>
__global__ void __launch_bounds__(64) k_rollout(KArgs a) {  // src/problem.jl:334-340 — open-loop simulate from x0
  constexpr int n = M::n, m = M::m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int Bp = P.Bp;
  const int c = a.cur[b];
  double* X = a.X[c];
  const double* U = a.U[c];
  double x[n], u[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) { x[i] = a.x0[(size_t)i * Bp + b]; X[TO_IDX(0, n, i)] = x[i]; }
  for (int k = 0; k < P.N - 1; ++k) {
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = U[TO_IDX(k, m, i)];
    rk_step<M, double>(P.mp, P.integrator, x, u, P.dt[k], xn);
#pragma unroll
    for (int i = 0; i < n; ++i) { x[i] = xn[i]; X[TO_IDX(k + 1, n, i)] = x[i]; }
  }
}

// ------------------------------------------------------------------------------------------------ cost
// mode bit0: include AL terms; bit1: write per-knot objective values to Jk[N][Bp] instead
template <class M>
__global__ void __launch_bounds__(64) k_cost(KArgs a, int with_al, double* out, double* Jk) {
  constexpr int n = M::n, m = M::m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int Bp = P.Bp, N = P.N;
  if (Jk) {
    const int c = a.cur[b];
    for (int k = 0; k < N; ++k) {
      double x[n], u[m];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = a.X[c][TO_IDX(k, n, i)];
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? a.U[c][TO_IDX(k, m, i)] : 0.0;
      Jk[(size_t)k * Bp + b] = knot_cost<M>(a, k, x, u, b, false);
    }
    return;
  }
  double J;
  trajectory_pass<M>(a, b, with_al != 0, false, &J, nullptr);
  out[b] = J;
}

template <class M>
__global__ void __launch_bounds__(64) k_violation(KArgs a, double* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.P.B) return;
  double cm;
  trajectory_pass<M>(a, b, false, false, nullptr, &cm);
  out[b] = cm;
}

template <class M>
__global__ void __launch_bounds__(64) k_dual_update(KArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  trajectory_pass<M>(a, b, false, true, nullptr, nullptr);
  for (int ci = 0; ci < P.n_cons; ++ci) {
    double* mu = &a.mu[(size_t)ci * P.Bp + b];
    *mu = fmin(*mu * P.opts.penalty_scaling, P.opts.penalty_max);
  }
}

// ------------------------------------------------------------------------------------------------ expansion
// One thread = one trajectory b (x), one knot k (y), one direction j (z) of the error-state tangent space
// [δx (ne); δu (m)].  It produces column j of everything the backward pass needs:
//   [Ā B̄][:,j]   = G(x_{k+1})ᵀ · ∂(RK step)/∂z · v_j          (forward-mode dual through all RK stages)
//   [Qxx;Qux][:,j] or Quu[:,j-ne] = projected Hessian-vector product of (cost + AL) with v_j
//   qx, qu (thread j==0 only)
// with v_j = [G(x_k) e_j; 0] (j<ne) or [0; e_{j-ne}].
template <class M>
__global__ void __launch_bounds__(64) k_expand(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nz = n + m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  if (!a.active[b]) return;
  const int Bp = P.Bp, N = P.N;
  const int k = blockIdx.y, j = blockIdx.z;
  const bool terminal = (k == N - 1);
  if (terminal && j >= ne) return;
  const int c = a.cur[b];
  const double* X = a.X[c];
  const double* U = a.U[c];
  double x[n], u[m], v[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = X[TO_IDX(k, n, i)];
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = terminal ? 0.0 : U[TO_IDX(k, m, i)];
  if (j < ne) {
    errstate_col<M>(x, j, v);
#pragma unroll
    for (int i = 0; i < m; ++i) v[n + i] = 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) v[n + i] = (i == j - ne) ? 1.0 : 0.0;
  }
  // ---- dynamics column
  if (!terminal) {
    Dual xd[n], ud[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xd[i] = Dual(x[i], v[i]);
#pragma unroll
    for (int i = 0; i < m; ++i) ud[i] = Dual(u[i], v[n + i]);
    rk_step<M, Dual>(P.mp, P.integrator, xd, ud, P.dt[k], xn);
    double x1[n], t[n], col[ne];
#pragma unroll
    for (int i = 0; i < n; ++i) { x1[i] = X[TO_IDX(k + 1, n, i)]; t[i] = xn[i].d; }
    errstate_tmul<M>(x1, t, col);
    if (j < ne) {
#pragma unroll
      for (int i = 0; i < ne; ++i) a.A[(((size_t)k * ne + i) * ne + j) * Bp + b] = col[i];
    } else {
#pragma unroll
      for (int i = 0; i < ne; ++i) a.Bm[(((size_t)k * ne + i) * m + (j - ne)) * Bp + b] = col[i];
    }
  }
  // ---- cost (+AL) gradient and Hessian-vector product on the full state
  double g[nz], y[nz];
  cost_grad_hvp<n, m>(P.costs[P.cost_index[k]], x, u, terminal, v, g, y);
  if (P.opts.cost_dt_scaling && !terminal) {
    const double h = P.dt[k];
#pragma unroll
    for (int i = 0; i < nz; ++i) { g[i] *= h; y[i] *= h; }
  }
  if (P.n_cons > 0) {
    double z[nz];
#pragma unroll
    for (int i = 0; i < n; ++i) z[i] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) z[n + i] = u[i];
    for (int ci = 0; ci < P.n_cons; ++ci) {
      const DevCon& K = P.cons[ci];
      if (k < K.k1 || k > K.k2) continue;
      const double* lam = a.lam + ((size_t)(K.dual_off + (long long)(k - K.k1) * K.p)) * Bp + b;
      al_grad_hvp<nz>(K, z, lam, (size_t)Bp, a.mu[(size_t)ci * Bp + b], v, g, y);
    }
  }
  if (j < ne) {
    double col[ne];
    errstate_tmul<M>(x, y, col);
    if constexpr (M::lie) {  // second-order term of the attitude map: −I₃ (qᵀ ∂J/∂q)
      const double b1 = x[3] * g[3] + x[4] * g[4] + x[5] * g[5] + x[6] * g[6];
      if (j >= 3 && j < 6) {
#pragma unroll
        for (int i = 3; i < 6; ++i) col[i] -= (i == j) ? b1 : 0.0;
      }
    }
#pragma unroll
    for (int i = 0; i < ne; ++i) a.Qxx[(((size_t)k * ne + i) * ne + j) * Bp + b] = col[i];
    if (!terminal) {
#pragma unroll
      for (int r = 0; r < m; ++r) a.Qux[(((size_t)k * m + r) * ne + j) * Bp + b] = y[n + r];
    }
    if (j == 0) {
      double qxe[ne];
      errstate_tmul<M>(x, g, qxe);
#pragma unroll
      for (int i = 0; i < ne; ++i) a.qx[TO_IDX(k, ne, i)] = qxe[i];
      if (!terminal) {
#pragma unroll
        for (int r = 0; r < m; ++r) a.qu[TO_IDX(k, m, r)] = g[n + r];
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < m; ++r) a.Quu[(((size_t)k * m + r) * m + (j - ne)) * Bp + b] = y[n + r];
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
// Riccati recursion, one lane per trajectory.  Control regularisation Quu + ρI with
// restart on Cholesky failure.
template <class M>
__global__ void __launch_bounds__(64) k_backward(KArgs a) {
  constexpr int m = M::m, ne = M::ne;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  if (!a.active[b]) return;
  const int Bp = P.Bp, N = P.N;
  double rho = a.rho[b], drho = a.drho[b];
  double S[ne][ne], s[ne];
  double dV0 = 0.0, dV1 = 0.0;
  bool failed = false;
  while (true) {
    bool restart = false;
    dV0 = 0.0; dV1 = 0.0;
#pragma unroll
    for (int i = 0; i < ne; ++i) {
#pragma unroll
      for (int j = 0; j < ne; ++j) S[i][j] = a.Qxx[(((size_t)(N - 1) * ne + i) * ne + j) * Bp + b];
      s[i] = a.qx[TO_IDX(N - 1, ne, i)];
    }
    for (int k = N - 2; k >= 0; --k) {
      double A[ne][ne], Bk[ne][m], SA[ne][ne], SB[ne][m];
#pragma unroll
      for (int i = 0; i < ne; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) A[i][j] = a.A[(((size_t)k * ne + i) * ne + j) * Bp + b];
#pragma unroll
        for (int j = 0; j < m; ++j) Bk[i][j] = a.Bm[(((size_t)k * ne + i) * m + j) * Bp + b];
      }
#pragma unroll
      for (int i = 0; i < ne; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) { double t = 0.0;
#pragma unroll
          for (int r = 0; r < ne; ++r) t += S[i][r] * A[r][j];
          SA[i][j] = t; }
#pragma unroll
        for (int j = 0; j < m; ++j) { double t = 0.0;
#pragma unroll
          for (int r = 0; r < ne; ++r) t += S[i][r] * Bk[r][j];
          SB[i][j] = t; }
      }
      double Qx[ne], Qu[m], Qxx[ne][ne], Quu[m][m], Qux[m][ne];
#pragma unroll
      for (int i = 0; i < ne; ++i) { double t = a.qx[TO_IDX(k, ne, i)];
#pragma unroll
        for (int r = 0; r < ne; ++r) t += A[r][i] * s[r];
        Qx[i] = t; }
#pragma unroll
      for (int j = 0; j < m; ++j) { double t = a.qu[TO_IDX(k, m, j)];
#pragma unroll
        for (int r = 0; r < ne; ++r) t += Bk[r][j] * s[r];
        Qu[j] = t; }
#pragma unroll
      for (int i = 0; i < ne; ++i)
#pragma unroll
        for (int j = 0; j < ne; ++j) { double t = a.Qxx[(((size_t)k * ne + i) * ne + j) * Bp + b];
#pragma unroll
          for (int r = 0; r < ne; ++r) t += A[r][i] * SA[r][j];
          Qxx[i][j] = t; }
#pragma unroll
      for (int i = 0; i < m; ++i) {
#pragma unroll
        for (int j = 0; j < m; ++j) { double t = a.Quu[(((size_t)k * m + i) * m + j) * Bp + b];
#pragma unroll
          for (int r = 0; r < ne; ++r) t += Bk[r][i] * SB[r][j];
          Quu[i][j] = t; }
#pragma unroll
        for (int j = 0; j < ne; ++j) { double t = a.Qux[(((size_t)k * m + i) * ne + j) * Bp + b];
#pragma unroll
          for (int r = 0; r < ne; ++r) t += Bk[r][i] * SA[r][j];
          Qux[i][j] = t; }
      }
      // Cholesky of Quu + ρI (lower, in L)
      double L[m][m];
      bool pd = true;
#pragma unroll
      for (int i = 0; i < m; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) L[i][j] = Quu[i][j] + ((i == j) ? rho : 0.0);
#pragma unroll
      for (int j = 0; j < m; ++j) {
        double sj = L[j][j];
#pragma unroll
        for (int r = 0; r < j; ++r) sj -= L[j][r] * L[j][r];
        if (!(sj > 0.0)) pd = false;
        const double l = sqrt(sj);
        L[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          double t = L[i][j];
#pragma unroll
          for (int r = 0; r < j; ++r) t -= L[i][r] * L[j][r];
          L[i][j] = t / l;
        }
      }
      if (!pd) {
        reg_increase(P.opts, rho, drho);
        if (rho > P.opts.bp_reg_max) failed = true;
        restart = true;
        break;
      }
      // gains: K = −(LLᵀ)⁻¹ Qux, d = −(LLᵀ)⁻¹ Qu
      double Kk[m][ne], dk[m];
#pragma unroll
      for (int j = 0; j <= ne; ++j) {
        double col[m];
#pragma unroll
        for (int i = 0; i < m; ++i) col[i] = (j < ne) ? Qux[i][j < ne ? j : 0] : Qu[i];
#pragma unroll
        for (int i = 0; i < m; ++i) { double t = col[i];
#pragma unroll
          for (int r = 0; r < i; ++r) t -= L[i][r] * col[r];
          col[i] = t / L[i][i]; }
#pragma unroll
        for (int i = m - 1; i >= 0; --i) { double t = col[i];
#pragma unroll
          for (int r = i + 1; r < m; ++r) t -= L[r][i] * col[r];
          col[i] = t / L[i][i]; }
#pragma unroll
        for (int i = 0; i < m; ++i) { if (j < ne) Kk[i][j < ne ? j : 0] = -col[i]; else dk[i] = -col[i]; }
      }
#pragma unroll
      for (int i = 0; i < m; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) a.K[(((size_t)k * m + i) * ne + j) * Bp + b] = Kk[i][j];
        a.d[TO_IDX(k, m, i)] = dk[i];
      }
      // cost-to-go with the un-regularised Quu
      double KtQuu[ne][m];
#pragma unroll
      for (int i = 0; i < ne; ++i)
#pragma unroll
        for (int j = 0; j < m; ++j) { double t = 0.0;
#pragma unroll
          for (int r = 0; r < m; ++r) t += Kk[r][i] * Quu[r][j];
          KtQuu[i][j] = t; }
      double snew[ne];
#pragma unroll
      for (int i = 0; i < ne; ++i) {
        double t = Qx[i];
#pragma unroll
        for (int j = 0; j < m; ++j) t += KtQuu[i][j] * dk[j];
#pragma unroll
        for (int j = 0; j < m; ++j) t += Kk[j][i] * Qu[j];
#pragma unroll
        for (int j = 0; j < m; ++j) t += Qux[j][i] * dk[j];
        snew[i] = t;
      }
#pragma unroll
      for (int i = 0; i < ne; ++i)
#pragma unroll
        for (int j = 0; j < ne; ++j) {
          double t = Qxx[i][j];
#pragma unroll
          for (int r = 0; r < m; ++r) t += KtQuu[i][r] * Kk[r][j];
#pragma unroll
          for (int r = 0; r < m; ++r) t += Kk[r][i] * Qux[r][j];
#pragma unroll
          for (int r = 0; r < m; ++r) t += Qux[r][i] * Kk[r][j];
          SA[i][j] = t;  // reuse SA as S_new
        }
#pragma unroll
      for (int i = 0; i < ne; ++i) {
#pragma unroll
        for (int j = 0; j < ne; ++j) S[i][j] = 0.5 * (SA[i][j] + SA[j][i]);
        s[i] = snew[i];
      }
      double dv1 = 0.0, dv2 = 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) {
        dv1 += dk[i] * Qu[i];
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) t += Quu[i][j] * dk[j];
        dv2 += dk[i] * t;
      }
      dV0 += dv1;
      dV1 += 0.5 * dv2;
    }
    if (!restart || failed) break;
  }
  if (!failed) reg_decrease(P.opts, rho, drho);
  a.rho[b] = rho;
  a.drho[b] = drho;
  a.dV[b] = dV0;
  a.dV[(size_t)Bp + b] = dV1;
  a.bpfail[b] = failed ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ forward pass
// Closed-loop rollout + backtracking line search, then — when a.control — the per-
// trajectory solver state machine: convergence test (row S3) and the AL outer update (row S4).
template <class M>
__global__ void __launch_bounds__(64) k_forward(KArgs a) {
  constexpr int n = M::n, m = M::m, ne = M::ne;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  if (!a.active[b]) return;
  const int Bp = P.Bp, N = P.N;
  const to_solver_opts& o = P.opts;
  const int c = a.cur[b];
  const double* Xc = a.X[c];
  const double* Uc = a.U[c];
  double* Xn = a.X[1 - c];
  double* Un = a.U[1 - c];
  const double Jprev = a.J[b];
  double rho = a.rho[b], drho = a.drho[b];
  const bool bpfail = a.bpfail[b] != 0;
  double Jnew = Jprev, grad = 0.0, grad_nominal = 0.0;
  int accepted = -1;
  if (!bpfail) {
    const double dV0 = a.dV[b], dV1 = a.dV[(size_t)Bp + b];
    double alpha = 1.0;
    for (int it = 0; it < o.iterations_linesearch; ++it) {
      double xb[n], J = 0.0, gsum = 0.0, gnom = 0.0;
      bool ok = true;
#pragma unroll
      for (int i = 0; i < n; ++i) { xb[i] = a.x0[(size_t)i * Bp + b]; Xn[TO_IDX(0, n, i)] = xb[i]; }
      for (int k = 0; k < N - 1; ++k) {
        double xk[n], dx[ne], ub[m], xn[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xk[i] = Xc[TO_IDX(k, n, i)];
        state_diff<M>(xb, xk, dx);
        double gk = 0.0, gk_nom = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) {
          const double dj = a.d[TO_IDX(k, m, j)];
          double du = dj * alpha;
#pragma unroll
          for (int i = 0; i < ne; ++i) du += a.K[(((size_t)k * m + j) * ne + i) * Bp + b] * dx[i];
          const double uk = Uc[TO_IDX(k, m, j)];
          ub[j] = uk + du;
          Un[TO_IDX(k, m, j)] = ub[j];
          gk = fmax(gk, fabs(dj) / (fabs(ub[j]) + 1.0));
          gk_nom = fmax(gk_nom, fabs(dj) / (fabs(uk) + 1.0));
        }
        gsum += gk; gnom += gk_nom;
        J += knot_cost<M>(a, k, xb, ub, b, true);
        rk_step<M, double>(P.mp, P.integrator, xb, ub, P.dt[k], xn);
        double mx = 0.0, mu_ = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { xb[i] = xn[i]; Xn[TO_IDX(k + 1, n, i)] = xn[i]; const double v = fabs(xn[i]); if (!(v <= mx)) mx = v; }
#pragma unroll
        for (int j = 0; j < m; ++j) { const double v = fabs(ub[j]); if (!(v <= mu_)) mu_ = v; }
        if (!(mx <= o.max_state_value) || !(mu_ <= o.max_control_value)) { ok = false; break; }
      }
      if (it == 0) grad_nominal = gnom / (N - 1);  // only complete (and only used) when the first rollout ran through; see below
      if (ok) {
        double u0[m];
#pragma unroll
        for (int j = 0; j < m; ++j) u0[j] = 0.0;
        J += knot_cost<M>(a, N - 1, xb, u0, b, true);
        const double expected = -alpha * (dV0 + alpha * dV1);
        const double z = (expected > 0.0) ? (Jprev - J) / expected : -1.0;
        if (z >= o.line_search_lower_bound && z <= o.line_search_upper_bound) {
          accepted = it; Jnew = J; grad = gsum / (N - 1);
          break;
        }
      }
      alpha *= o.line_search_decrease_factor;
    }
    if (accepted < 0) {
      // gradient metric on the unchanged nominal controls (exact: recompute, the in-loop value may be partial)
      double gs = 0.0;
      for (int k = 0; k < N - 1; ++k) {
        double gk = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) gk = fmax(gk, fabs(a.d[TO_IDX(k, m, j)]) / (fabs(Uc[TO_IDX(k, m, j)]) + 1.0));
        gs += gk;
      }
      grad = gs / (N - 1);
      (void)grad_nominal;
      reg_increase(o, rho, drho);
      rho += o.bp_reg_fp;
    } else {
      a.cur[b] = 1 - c;
    }
  }
  a.ls_index[b] = accepted;
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
    dz = ls_failed ? dz + 1 : 0;
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
    else {
      const int outer = a.outer[b] + 1;
      a.outer[b] = outer;
      double cm;
      trajectory_pass<M>(a, b, false, false, nullptr, &cm);
      a.cmax[b] = cm;
      const int its = a.iterations[b];
      if (st != TO_SOLVE_SUCCEEDED && st != TO_MAX_ITERATIONS && st != TO_NO_PROGRESS) { a.status[b] = st; still_active = false; }
      else if (cm < o.constraint_tolerance) { a.status[b] = TO_SOLVE_SUCCEEDED; still_active = false; }
      else if (its >= o.iterations_total) { a.status[b] = TO_MAX_ITERATIONS; still_active = false; }
      else if (outer >= o.iterations_outer) { a.status[b] = TO_MAX_ITERATIONS_OUTER; still_active = false; }
      else {
        // dual + penalty update, then start the next inner solve on the same trajectory
        trajectory_pass<M>(a, b, false, true, nullptr, nullptr);
        for (int ci = 0; ci < P.n_cons; ++ci) {
          double* mu = &a.mu[(size_t)ci * Bp + b];
          *mu = fmin(*mu * o.penalty_scaling, o.penalty_max);
        }
        double Jal;
        trajectory_pass<M>(a, b, true, false, &Jal, nullptr);
        a.J[b] = Jal;
        rho = o.bp_reg_initial; drho = 0.0;
        a.dJzero[b] = 0; a.it_inner[b] = 0;
        const int rem = o.iterations_total - its;
        a.budget[b] = rem < o.iterations ? rem : o.iterations;
        a.status[b] = TO_UNSOLVED;
      }
    }
  }
  a.rho[b] = rho; a.drho[b] = drho;
  if (!still_active) a.active[b] = 0;
  else atomicAdd(&a.counter[a.step], 1);
}

// start of a solve: reset the per-trajectory solver state.  J must already hold the (AL) cost of the rollout.
__global__ void k_solve_init(KArgs a, int reset_duals) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.Bp) return;
  const bool live = b < P.B;
  a.rho[b] = P.opts.bp_reg_initial; a.drho[b] = 0.0;
  a.dJzero[b] = 0; a.it_inner[b] = 0; a.iterations[b] = 0; a.outer[b] = 0;
  a.status[b] = TO_UNSOLVED; a.ls_index[b] = -1; a.bpfail[b] = 0;
  a.dJ[b] = 0.0; a.grad[b] = 0.0; a.cmax[b] = 0.0;
  const int tot = a.al_mode ? P.opts.iterations_total : P.opts.iterations;
  a.budget[b] = tot < P.opts.iterations ? tot : P.opts.iterations;
  a.active[b] = (live && a.budget[b] > 0) ? 1 : 0;
  if (live && a.budget[b] <= 0) a.status[b] = TO_MAX_ITERATIONS;
  if (reset_duals) {
    for (long long r = 0; r < P.n_duals; ++r) a.lam[(size_t)r * P.Bp + b] = 0.0;
    for (int ci = 0; ci < P.n_cons; ++ci) a.mu[(size_t)ci * P.Bp + b] = P.opts.penalty_initial;
  }
}

__global__ void k_set_active(KArgs a, int value) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.P.Bp) return;
  a.active[b] = (b < a.P.B) ? value : 0;
  a.bpfail[b] = 0;
}

__global__ void k_penalty_max(KArgs a, double* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.P.B) return;
  double mx = 0.0;
  for (int ci = 0; ci < a.P.n_cons; ++ci) mx = fmax(mx, a.mu[(size_t)ci * a.P.Bp + b]);
  out[b] = mx;
}

// ------------------------------------------------------------------------------------------------ layout transposes
// host layout  h[i + dim*(k + K*b)]   <->   device layout d[(k*dim + i)*Bp + b]
__global__ void k_to_device(const double* __restrict__ h, double* __restrict__ d, int dim, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ki = blockIdx.y;  // k*dim + i
  if (b >= B) return;
  d[(size_t)ki * Bp + b] = h[(size_t)ki + (size_t)dim * K * b];
}
__global__ void k_to_host(const double* __restrict__ d, double* __restrict__ h, int dim, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ki = blockIdx.y;
  if (b >= B) return;
  h[(size_t)ki + (size_t)dim * K * b] = d[(size_t)ki * Bp + b];
}
// same with the nominal buffer chosen per trajectory (X/U double buffering)
__global__ void k_to_host_cur(const double* __restrict__ d0, const double* __restrict__ d1, const int* __restrict__ cur,
                              double* __restrict__ h, int dim, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ki = blockIdx.y;
  if (b >= B) return;
  const double* d = cur[b] ? d1 : d0;
  h[(size_t)ki + (size_t)dim * K * b] = d[(size_t)ki * Bp + b];
}
__global__ void k_to_device_cur(const double* __restrict__ h, double* __restrict__ d0, double* __restrict__ d1,
                                const int* __restrict__ cur, int dim, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ki = blockIdx.y;
  if (b >= B) return;
  double* d = cur[b] ? d1 : d0;
  d[(size_t)ki * Bp + b] = h[(size_t)ki + (size_t)dim * K * b];
}
__global__ void k_fill_uniform(double* d0, double* d1, const int* cur, const double* u, int dim, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int ki = blockIdx.y;
  if (b >= B) return;
  double* d = cur[b] ? d1 : d0;
  d[(size_t)ki * Bp + b] = u[ki % dim];
}
// matrices: host h[r + R*(c + Cc*(k + K*b))] (column-major) <-> device d[((k*R + r)*Cc + c)*Bp + b]
__global__ void k_mat_to_host(const double* __restrict__ d, double* __restrict__ h, int R, int Cc, int K, int B, int Bp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;  // (k*R + r)*Cc + c
  if (b >= B) return;
  const int c = e % Cc, r = (e / Cc) % R, k = e / (Cc * R);
  h[(size_t)r + (size_t)R * (c + (size_t)Cc * (k + (size_t)K * b))] = d[(size_t)e * Bp + b];
}

// ------------------------------------------------------------------------------------------------ per-knot API kernels
// RD.gradient!/RD.hessian! of the objective on the full state (no AL, no error-state projection):
// grad[(n+m), N, B], hess[(n+m),(n+m),N,B] column-major host layout, written directly.
template <class M>
__global__ void __launch_bounds__(64) k_cost_derivs(KArgs a, double* grad, double* hess) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int Bp = P.Bp, N = P.N, k = blockIdx.y;
  const bool terminal = (k == N - 1);
  const int c = a.cur[b];
  double x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.X[c][TO_IDX(k, n, i)];
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = terminal ? 0.0 : a.U[c][TO_IDX(k, m, i)];
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
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const int Bp = P.Bp, N = P.N, k = blockIdx.y, j = blockIdx.z;
  const int c = a.cur[b];
  Dual xd[n], ud[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) xd[i] = Dual(a.X[c][TO_IDX(k, n, i)], (i == j) ? 1.0 : 0.0);
#pragma unroll
  for (int i = 0; i < m; ++i) ud[i] = Dual(a.U[c][TO_IDX(k, m, i)], (n + i == j) ? 1.0 : 0.0);
  rk_step<M, Dual>(P.mp, P.integrator, xd, ud, P.dt[k], xn);
  const size_t kb = (size_t)k + (size_t)(N - 1) * b;
#pragma unroll
  for (int i = 0; i < n; ++i) F[(size_t)i + n * ((size_t)j + nz * kb)] = xn[i].d;
}

// evaluate_constraints! / constraint_jacobians! for one constraint over its knot range, host layout output
template <class M>
__global__ void __launch_bounds__(64) k_constraint_eval(KArgs a, int ci, double* vals, double* jac) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  const DevCon& K = P.cons[ci];
  const int Bp = P.Bp, N = P.N, kk = blockIdx.y, k = K.k1 + kk, nk = K.k2 - K.k1 + 1;
  const int c = a.cur[b];
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = a.X[c][TO_IDX(k, n, i)];
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = (k < N - 1) ? a.U[c][TO_IDX(k, m, i)] : 0.0;
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
