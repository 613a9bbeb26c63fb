// k_misc.h — rollout, cost, violation, dual update, the knot-parallel AL outer update and the per-knot API kernels.
#pragma once
#include "common.h"

namespace to {

// ------------------------------------------------------------------------------------------------ rollout!
template <class M, int FIXED_INTEG>
__global__ void __launch_bounds__(64) k_rollout(KArgs a) {  // src/problem.jl:334-340 — open-loop simulate from x0
  constexpr int n = M::n, m = M::m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  constexpr int c = 0;  // nominal slot
  double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  const double* x0 = TILE_PTR(a.x0, n);
  double x[n], u[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) { x[i] = EL(x0, i); EL(X, i) = x[i]; }
  int lim = 0;  // first knot beyond max_state_value / max_control_value: the state it arrives at first, then the control (Altro's rollout!)
  const double max_x = P.opts.max_state_value, max_u = P.opts.max_control_value;
  for (int k = 0; k < P.N - 1; ++k) {
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = EL(U, k * m + i);
    model_step<M, double, FIXED_INTEG>(P.mp, P.integrator, k, x, u, P.dt[k], xn);
    double mx = 0.0, mu = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) { x[i] = xn[i]; EL(X, (k + 1) * n + i) = x[i]; const double v = fabs(x[i]); mx = !(v <= mx) ? v : mx; }
#pragma unroll
    for (int i = 0; i < m; ++i) { const double v = fabs(u[i]); mu = !(v <= mu) ? v : mu; }
    if (lim == 0) lim = !(mx <= max_x) ? TO_STATE_LIMIT : !(mu <= max_u) ? TO_CONTROL_LIMIT : 0;
  }
  // the initial rollout of a solve (KArgs::control): a trajectory beyond the limits ends right here — TO_STATE_LIMIT / TO_CONTROL_LIMIT,
  // no iteration performed (inside a line search such a candidate is simply rejected, k_forward.h); the phase API only simulates
  if (a.control && lim != 0 && a.active[b] != 0) {
    a.status[b] = lim; a.active[b] = 0;
    if (a.al_mode) a.outer[b] = 1;  // (the AL loop counts the outer iteration its first inner solve belongs to)
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
    const double* X = X_SLOT_PTR(a, b, c);
    const double* U = U_SLOT_PTR(a, b, c);
    double* o = TILE_PTR(Jk, N);
    for (int k = 0; k < N; ++k) {
      double x[n], u[m];
#pragma unroll
      for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
      for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
      EL(o, k) = knot_cost<M>(P, k, x, u, nullptr, nullptr, false, TILE_PTR(P.gl, P.n_costs * (n + m)));
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


// ---- AL outer update of the trajectories whose inner solve just ended (oflag = 1), knot-parallel:
//   k_outer_violation (waves, N): constraint violation of every knot -> knotbuf
//   k_outer_decide    (waves)   : c_max, termination tests; trajectories that go on get oflag = 2 and their new penalties
//   k_outer_update    (waves, N): dual update with the OLD penalties, then the knot's AL cost with the new duals/penalties
//   k_outer_finish    (waves)   : J = sum of the knot terms in knot order (same sum as a sequential pass), restart the inner solve
// One lane per trajectory walking all knots three times was the largest kernel of the constrained solves.  The lanes of a
// wave take their trajectories from the COMPACTED list k_forward appended them to (a.olist): in C5 some 6 % of the batch
// ends an inner solve in any given batch step, scattered over nearly every 64-trajectory tile — mapped tile by tile, every
// (tile, knot) wave ran for three or four useful lanes (the four kernels: 236 us per step, 5 % of the solve).
#define OUTER_LANE(FLAG)                                                                      \
  const DevProblem& P = a.P;                                                                  \
  const int cnt_ = a.ocount[a.step & 1];                                                      \
  if ((int)blockIdx.x * 64 >= cnt_) return;                                                   \
  const int li_ = blockIdx.x * 64 + threadIdx.x;                                              \
  const int b = a.olist[(size_t)(a.step & 1) * P.Bp + (li_ < cnt_ ? li_ : cnt_ - 1)];         \
  const int tile = b >> 6, lane = b & 63;                                                     \
  const bool want = li_ < cnt_ && a.oflag[b] == (FLAG)

template <class M>
__global__ void __launch_bounds__(64) k_outer_violation(KArgs a) {
  constexpr int n = M::n, m = M::m;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.ocount[(a.step + 1) & 1] = 0;  // next step's list starts empty
  OUTER_LANE(1);
  if (__ballot(want) == 0) return;
  if (!want) return;
  const int N = P.N, k = blockIdx.y;
  const int sl = a.acc[b];  // the step accepted in this iteration (0: none, nominal unchanged)
  const double* X = X_SLOT_PTR(a, b, sl);
  const double* U = U_SLOT_PTR(a, b, sl);
  double x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
  EL(TILE_PTR(a.knotbuf, N), k) = (P.n_cons > 0) ? knot_violation<M>(P, k, x, u, TILE_PTR(P.cp, P.n_cp)) : 0.0;
}

template <class M>
__global__ void __launch_bounds__(64) k_outer_decide(KArgs a) {
  OUTER_LANE(1);
  if (!want) return;
  const to_solver_opts& o = P.opts;
  const int N = P.N, st = a.ost[b];
  const int outer = a.outer[b] + 1;
  a.outer[b] = outer;
  const double* vb = TILE_PTR(a.knotbuf, N);
  double cm = 0.0;
#pragma unroll 8
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
  OUTER_LANE(2);
  if (__ballot(want) == 0) return;
  if (!want) return;
  const int N = P.N, k = blockIdx.y;
  const int sl = a.acc[b];
  const double* X = X_SLOT_PTR(a, b, sl);
  const double* U = U_SLOT_PTR(a, b, sl);
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
    double zc[nz];
#pragma unroll
    for (int i = 0; i < nz; ++i) zc[i] = z[i];
    con_shift<nz>(P, K, TILE_PTR(P.cp, P.n_cp), zc);
    con_dual_update<nz>(K, zc, lam, (size_t)64, EL(mu0, ci), P.opts.dual_max);
  }
  EL(TILE_PTR(a.knotbuf, N), k) = knot_cost<M>(P, k, x, u, lam0, mn0, true, TILE_PTR(P.gl, P.n_costs * nz), TILE_PTR(P.cp, P.n_cp));
}

template <class M>
__global__ void __launch_bounds__(64) k_outer_finish(KArgs a) {
  OUTER_LANE(2);
  if (!want) return;
  const to_solver_opts& o = P.opts;
  const int N = P.N;
  const double* jb = TILE_PTR(a.knotbuf, N);
  double J = 0.0;
#pragma unroll 8
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
#undef OUTER_LANE

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
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
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
    if (P.gl) goal_lin_grad<n, m>(TILE_PTR(P.gl, P.n_costs * nz), P.cost_index[k], terminal, g);
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
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  Dual xd[n], ud[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) xd[i] = Dual(EL(X, k * n + i), (i == j) ? 1.0 : 0.0);
#pragma unroll
  for (int i = 0; i < m; ++i) ud[i] = Dual(EL(U, k * m + i), (n + i == j) ? 1.0 : 0.0);
  model_step<M, Dual>(P.mp, P.integrator, k, xd, ud, P.dt[k], xn);
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
  const double* X = X_SLOT_PTR(a, b, c);
  const double* U = U_SLOT_PTR(a, b, c);
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
  con_shift<nz>(P, K, TILE_PTR(P.cp, P.n_cp), z);
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

// ∇jacobian! (src/abstract_constraint.jl:255-280): H[w,w,nk,B] += sum_r lambda_r * Hessian of c_r at the nominal trajectory, for
// one constraint over its knot range; lambda[p,nk,B], both in host layout on the device.  Closed forms (the reference
// differentiates the Jacobian with ForwardDiff; Goal :70-73 and Bound :767-770 are zero there too): affine rows contribute
// nothing, |z_I|^2 - a^2 gives 2 I on I, circle / sphere -2 I on the centre coordinates, collision -2 [I -I; -I I],
// QuatVecEq the second derivative of q/|q|.
template <class M>
__global__ void __launch_bounds__(64) k_constraint_hessian(KArgs a, int ci, const double* lambda, double* H) {
  constexpr int n = M::n, m = M::m, nz = n + m;
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.B) return;
  ConC& K = P.cons[ci];
  const int N = P.N, kk = blockIdx.y, k = K.k1 + kk, nk = K.k2 - K.k1 + 1, w = K.width, p = K.p;
  const double* X = X_SLOT_PTR(a, b, 0);
  const double* U = U_SLOT_PTR(a, b, 0);
  double z[nz];
#pragma unroll
  for (int i = 0; i < n; ++i) z[i] = EL(X, k * n + i);
#pragma unroll
  for (int i = 0; i < m; ++i) z[n + i] = (k < N - 1) ? EL(U, k * m + i) : 0.0;
  const size_t kb = (size_t)kk + (size_t)nk * b;
  const double* lam = lambda + (size_t)p * kb;
  double* Hk = H + (size_t)w * w * kb;
  auto add = [&](int i, int j, double v) { Hk[(size_t)i + (size_t)w * j] += v; };
  const int kind = K.d.kind;
  if (kind == TO_CON_NORM) {
    if (K.d.sense != TO_CONE_SECOND_ORDER) for (int t = 0; t < K.d.n_inds; ++t) add(K.d.inds[t] - 1, K.d.inds[t] - 1, 2.0 * lam[0]);
  } else if (kind == TO_CON_CIRCLE || kind == TO_CON_SPHERE) {
    const int D = kind == TO_CON_CIRCLE ? 2 : 3;
    for (int i = 0; i < p; ++i) for (int t = 0; t < D; ++t) add(K.d.inds[t] - 1, K.d.inds[t] - 1, -2.0 * lam[i]);
  } else if (kind == TO_CON_COLLISION) {
    const int D = K.d.n_inds / 2;
    for (int t = 0; t < D; ++t) {
      const int ia = K.d.inds[t] - 1, ib = K.d.inds[D + t] - 1;
      add(ia, ia, -2.0 * lam[0]); add(ib, ib, -2.0 * lam[0]); add(ia, ib, 2.0 * lam[0]); add(ib, ia, 2.0 * lam[0]);
    }
  } else if (kind == TO_CON_QUATVEC) {
    double q[4], s2 = 0.0;
    for (int t = 0; t < 4; ++t) { q[t] = pick<nz>(z, K.d.inds[t] - 1); s2 += q[t] * q[t]; }
    const double s = sqrt(s2), s3 = s2 * s, s5 = s3 * s2;
    for (int r = 0; r < 3; ++r) {
      const int i = r + 1;
      for (int j = 0; j < 4; ++j)
        for (int kq = 0; kq < 4; ++kq) {
          const double v = -((i == j ? q[kq] : 0.0) + (i == kq ? q[j] : 0.0) + (j == kq ? q[i] : 0.0)) / s3 + 3.0 * q[i] * q[j] * q[kq] / s5;
          add(K.d.inds[j] - 1, K.d.inds[kq] - 1, lam[r] * v);
        }
    }
  }
}

// Altro's infeasible_controls for an InfeasibleModel (models.h): w_k = x_{k+1} - f_d(x_k, u_k[0 .. m0)) from the current nominal states,
// one lane per (trajectory, knot), written into the slack entries of the nominal controls.  grid (tiles, N-1).
template <class M>
__global__ void __launch_bounds__(64) k_infeasible_controls(KArgs a) {
  constexpr int n = M::n, m = M::m, m0 = M::m0;
  TILE_LANE();
  const DevProblem& P = a.P;
  const int N = P.N, k = blockIdx.y;
  if (b >= P.B) return;
  const double* X = TILE_PTR(a.Xs, N * n);
  double* U = TILE_PTR(a.Us, (N - 1) * m);
  double x[n], u[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = EL(X, k * n + i);
#pragma unroll
  for (int j = 0; j < m; ++j) u[j] = j < m0 ? EL(U, k * m + j) : 0.0;
  double mp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) mp[i] = P.mp[i];
  model_step<M, double, -1>(mp, P.integrator, k, x, u, P.dt[k], xn);
#pragma unroll
  for (int i = 0; i < n; ++i) EL(U, k * m + m0 + i) = EL(X, (k + 1) * n + i) - xn[i];
}

}  // namespace to
