/*
 * oracle_pn.h — TEST INFRASTRUCTURE ONLY (part of the CPU oracle; included by trajopt_oracle.cpp inside its anonymous namespace).
 *
 * Projected-Newton polish: the last stage of ALTRO (Altro.jl ProjectedNewtonSolver — out of tree; the reference's own
 * constrained results are produced with it: examples/Cartpole.ipynb cells 17-19 "Terminal constraint violation 3.4e-9",
 * examples/Quadrotor.ipynb cell 20 "7.6e-10").  No reference artefact pins its internals, so — like the AL stage (SURVEY.md row
 * S4) — its rules are DEFINED here, shaped after Altro's solver (solve! -> projection_solve! -> _projection_solve! ->
 * _projection_linesearch! -> reg_solve), and the GPU must match this file:
 *
 *   primal        Z = (x_1, u_1, ..., x_N) moved along ERROR-STATE steps: x (+) dx (state_add), u + du
 *   rows          initial condition x_1 (-) x0, dynamics defects f(x_k, u_k) (-) x_{k+1}, and per knot the ACTIVE rows of the
 *                 constraint list: every row of an equality constraint, the rows of an inequality constraint with
 *                 c >= -active_set_tolerance_pn, a second-order cone [v; s] through the ONE scalar row |v| - s (>= -tol);
 *                 rows whose gradient vanishes (a control bound at the terminal knot) never take part
 *   metric        H = diag of the error-state OBJECTIVE Hessian (no AL terms; Altro: "assume constant, diagonal cost
 *                 Hessian"), clipped at 0, + rho_primal
 *   one step      dZ = -H^-1 D' dl,  (D H^-1 D' + rho_chol I) dl ~ d  by Cholesky + iterative refinement against the
 *                 unregularised S (reg_solve: at most 25 rounds, until |r|_2 < 1e-8)
 *   line search   alpha = 1, 1/2, ... (10 trials) until |d(Z (+) alpha dZ)|_inf < |d(Z)|_inf on the frozen active set;
 *                 no trial accepted: the refinement loop ends (Altro keeps the last trial; keeping a worse point helps nobody)
 *   refinement    at most 10 steps on one linearisation, until viol < constraint_tolerance or the convergence rate
 *                 log10(viol)/log10(viol_prev) drops below r_threshold (viol_prev >= 1: until viol >= viol_prev / 2)
 *   outer         at most n_steps + 1 linearisations (Altro projection_solve!: `while count <= n_steps`), each preceded by a fresh
 *                 active set; done when |d|_inf <= constraint_tolerance
 *
 * S is assembled and factorised here as ONE BANDED matrix (row-oriented Cholesky over rows ordered knot by knot); the GPU
 * factorises the same matrix as block-tridiagonal knot blocks.  ORACLE_PN_DENSE=1 widens the band to the full matrix (dense
 * Cholesky) for the cross-check in tests/test_oracle_pn.py.
 */

constexpr int PN_MAX_ROWS = 64;      /* candidate constraint rows per knot (one bit each in the active mask) */
constexpr int PN_REFINEMENTS = 10;   /* Altro _projection_solve!: max_refinements */
constexpr int PN_LS_TRIALS = 10;     /* Altro _projection_linesearch!: count > 10 */
constexpr int PN_REG_SOLVE_ITERS = 25;
constexpr double PN_REG_SOLVE_TOL = 1e-8;

struct PnRow {
  int v0 = 0, len = 0;               /* window of primal variables the row touches */
  double coef[2 * MAXZ];
};

struct PnSys {
  int N = 0, ne = 0, m = 0, nc = 0, nv = 0, M = 0, bw = 0;
  std::vector<double> W;             /* [nv] inverse metric */
  std::vector<uint64_t> mask;        /* [N] active candidate rows of each knot */
  std::vector<int> goff;             /* [N+1] first row of group k = [arriving defect (ne); active constraint rows] */
  std::vector<PnRow> rows;
  std::vector<double> d, S, L;
  double& s_at(int i, int j) { return S[(size_t)i * (bw + 1) + (j - i + bw)]; } /* j in [i-bw, i] */
  double& l_at(int i, int j) { return L[(size_t)i * (bw + 1) + (j - i + bw)]; }
};

/* number of candidate rows of knot k */
int pn_candidate_count(const Problem& P, int k) {
  int q = 0;
  for (const ConInfo& ci : P.cons) {
    if (k < ci.k1 || k > ci.k2) continue;
    q += (ci.d.sense == TO_CONE_SECOND_ORDER) ? 1 : ci.p;
  }
  return q;
}

/* values (and gradients with respect to z = [x; u], row-major [q][nz]) of every candidate row of knot k */
int pn_candidates(const Problem& P, const Traj& t, const double* X, const double* U, int k, double* val, double* grad) {
  const int n = P.n, m = P.m, nz = n + m;
  double z[MAXZ], c[TO_MAX_P], jac[TO_MAX_P * MAXZ];
  knot_z(P, X, U, k, z);
  int q = 0;
  for (size_t ic = 0; ic < P.cons.size(); ++ic) {
    const ConInfo& ci = P.cons[ic];
    if (k < ci.k1 || k > ci.k2) continue;
    EffDesc ed;
    constraint_evaluate(ed.get(t, ic, ci), n, m, z, c, grad ? jac : nullptr);
    if (ci.d.sense == TO_CONE_SECOND_ORDER) { /* [v; s] in the cone <=> |v| - s <= 0 */
      const int p = ci.p;
      double a2 = 0.0; for (int i = 0; i < p - 1; ++i) a2 += c[i] * c[i];
      const double a = std::sqrt(a2);
      val[q] = a - c[p - 1];
      if (grad) for (int j = 0; j < nz; ++j) {
        double g = 0.0;
        if (a > 0.0) for (int i = 0; i < p - 1; ++i) g += (c[i] / a) * jac[i * nz + j];
        grad[q * nz + j] = g - jac[(p - 1) * nz + j];
      }
      ++q;
    } else {
      for (int r = 0; r < ci.p; ++r) {
        val[q] = c[r];
        if (grad) for (int j = 0; j < nz; ++j) grad[q * nz + j] = jac[r * nz + j];
        ++q;
      }
    }
  }
  return q;
}

/* sense of candidate row q of knot k: true = equality (always active) */
void pn_candidate_senses(const Problem& P, int k, bool* eq) {
  int q = 0;
  for (const ConInfo& ci : P.cons) {
    if (k < ci.k1 || k > ci.k2) continue;
    const int rows = (ci.d.sense == TO_CONE_SECOND_ORDER) ? 1 : ci.p;
    for (int r = 0; r < rows; ++r) eq[q++] = (ci.d.sense == TO_CONE_ZERO);
  }
}

/* defect arriving at knot k (ne): k = 0 the initial condition x_1 (-) x0, otherwise f(x_{k-1}, u_{k-1}) (-) x_k */
void pn_defect(const Problem& P, const Traj& t, const double* X, const double* U, int k, double* e) {
  const int n = P.n, m = P.m;
  if (k == 0) { state_diff(P.M, X, t.x0.data(), e); return; }
  double f[MAXN];
  knot_step(P.M, P.integrator, k - 1, &X[(size_t)(k - 1) * n], &U[(size_t)(k - 1) * m], P.dt[k - 1], f);
  state_diff(P.M, f, &X[(size_t)k * n], e);
}

/* max |defect| over the horizon (the dynamics infeasibility of a trajectory that is not a rollout) */
double dynamics_defect(const Problem& P, const Traj& t) {
  double mx = 0.0, e[MAXN];
  for (int k = 0; k < P.N; ++k) {
    pn_defect(P, t, t.X.data(), t.U.data(), k, e);
    for (int i = 0; i < P.ne; ++i) { const double v = std::fabs(e[i]); if (v > mx || std::isnan(v)) mx = v; }
  }
  return mx;
}

/* d on the active set; refresh: choose the active set from the values at (X, U) first.  Returns |d|_inf (NaN-aware). */
double pn_residual(const Problem& P, const Traj& t, const double* X, const double* U, PnSys& s, bool refresh, std::vector<double>& d) {
  const int N = P.N, ne = P.ne, n = P.n, m = P.m, nz = n + m;
  const double tol_a = P.opts.active_set_tolerance_pn;
  double val[PN_MAX_ROWS], grad[PN_MAX_ROWS * MAXZ], G[MAXN * MAXN];
  bool eq[PN_MAX_ROWS];
  if (refresh) {
    s.mask.assign(N, 0); s.goff.assign(N + 1, 0);
    for (int k = 0; k < N; ++k) {
      const int nq = pn_candidates(P, t, X, U, k, val, grad);
      pn_candidate_senses(P, k, eq);
      errstate_jacobian(P.M, &X[(size_t)k * n], G);
      uint64_t mk = 0; int pa = 0;
      for (int q = 0; q < nq; ++q) {
        if (!(eq[q] || val[q] >= -tol_a)) continue;
        /* gradient in the coordinates the step moves in: [G' g_x; g_u] (no control at the terminal knot) */
        double g2 = 0.0;
        for (int j = 0; j < ne; ++j) { double v = 0.0; for (int r = 0; r < n; ++r) v += G[r * ne + j] * grad[q * nz + r]; g2 += v * v; }
        if (k < N - 1) for (int j = 0; j < m; ++j) g2 += grad[q * nz + n + j] * grad[q * nz + n + j];
        if (!(g2 > 0.0)) continue;
        mk |= (uint64_t)1 << q; ++pa;
      }
      s.mask[k] = mk; s.goff[k + 1] = s.goff[k] + ne + pa;
    }
    s.M = s.goff[N];
  }
  d.assign(s.M, 0.0);
  double viol = 0.0;
  for (int k = 0; k < N; ++k) {
    double* dk = &d[s.goff[k]];
    pn_defect(P, t, X, U, k, dk);
    const int nq = pn_candidates(P, t, X, U, k, val, nullptr);
    int a = 0;
    for (int q = 0; q < nq; ++q) if (s.mask[k] >> q & 1) dk[ne + a++] = val[q];
    for (int i = 0; i < ne + a; ++i) { const double v = std::fabs(dk[i]); if (v > viol || std::isnan(v)) viol = v; }
  }
  return viol;
}

/* Jacobian rows of the active set at (X, U) (mask as chosen by the last refresh) and the inverse metric */
void pn_linearise(const Problem& P, const Traj& t, const double* X, const double* U, PnSys& s) {
  const int N = P.N, ne = P.ne, n = P.n, m = P.m, nz = n + m, nc = ne + m;
  s.N = N; s.ne = ne; s.m = m; s.nc = nc; s.nv = (N - 1) * nc + ne;
  s.rows.assign(s.M, PnRow());
  s.W.assign(s.nv, 0.0);
  std::vector<double> Ae(ne * ne), Be(ne * m), Qxx(ne * ne), Quu(m * m), Qux(m * ne), qx(ne), qu(m);
  double val[PN_MAX_ROWS], grad[PN_MAX_ROWS * MAXZ], G[MAXN * MAXN];
  for (int k = 0; k < N; ++k) {
    /* metric: objective only */
    cost_blocks(P, t, X, U, /*with_al*/ false, k, Qxx.data(), Quu.data(), Qux.data(), qx.data(), qu.data());
    for (int i = 0; i < ne; ++i) s.W[(size_t)k * nc + i] = 1.0 / (std::fmax(Qxx[i * ne + i], 0.0) + P.opts.rho_primal);
    if (k < N - 1) for (int j = 0; j < m; ++j) s.W[(size_t)k * nc + ne + j] = 1.0 / (std::fmax(Quu[j * m + j], 0.0) + P.opts.rho_primal);
    /* arriving defect */
    PnRow* R = &s.rows[s.goff[k]];
    if (k == 0) {
      for (int i = 0; i < ne; ++i) { R[i].v0 = 0; R[i].len = ne; for (int j = 0; j < ne; ++j) R[i].coef[j] = (i == j) ? 1.0 : 0.0; }
    } else {
      dynamics_blocks(P, X, U, k - 1, Ae.data(), Be.data());
      for (int i = 0; i < ne; ++i) {
        R[i].v0 = (k - 1) * nc; R[i].len = nc + ne;
        for (int j = 0; j < ne; ++j) R[i].coef[j] = Ae[i * ne + j];
        for (int j = 0; j < m; ++j) R[i].coef[ne + j] = Be[i * m + j];
        for (int j = 0; j < ne; ++j) R[i].coef[nc + j] = (i == j) ? -1.0 : 0.0;
      }
    }
    /* active constraint rows, in error-state coordinates */
    const int nq = pn_candidates(P, t, X, U, k, val, grad);
    errstate_jacobian(P.M, &X[(size_t)k * n], G);
    int a = 0;
    for (int q = 0; q < nq; ++q) {
      if (!(s.mask[k] >> q & 1)) continue;
      PnRow& r = R[ne + a++];
      r.v0 = k * nc; r.len = (k < N - 1) ? nc : ne;
      for (int j = 0; j < ne; ++j) { double v = 0.0; for (int i = 0; i < n; ++i) v += G[i * ne + j] * grad[q * nz + i]; r.coef[j] = v; }
      if (k < N - 1) for (int j = 0; j < m; ++j) r.coef[ne + j] = grad[q * nz + n + j];
    }
  }
}

/* S = D W D' (banded, lower) and the Cholesky factor of S + rho I */
bool pn_factor(PnSys& s, double rho) {
  const int M = s.M;
  int bw = 0;
  for (int k = 1; k < s.N; ++k) bw = std::max(bw, s.goff[k + 1] - s.goff[k - 1] - 1);
  bw = std::max(bw, s.goff[1] - 1);
  if (std::getenv("ORACLE_PN_DENSE")) bw = M - 1;
  s.bw = bw;
  s.S.assign((size_t)M * (bw + 1), 0.0); s.L.assign((size_t)M * (bw + 1), 0.0);
  for (int i = 0; i < M; ++i) {
    const PnRow& ri = s.rows[i];
    for (int j = std::max(0, i - bw); j <= i; ++j) {
      const PnRow& rj = s.rows[j];
      const int lo = std::max(ri.v0, rj.v0), hi = std::min(ri.v0 + ri.len, rj.v0 + rj.len);
      double v = 0.0;
      for (int c = lo; c < hi; ++c) v += ri.coef[c - ri.v0] * s.W[c] * rj.coef[c - rj.v0];
      s.s_at(i, j) = v;
    }
  }
  for (int i = 0; i < M; ++i)
    for (int j = std::max(0, i - bw); j <= i; ++j) {
      double v = s.s_at(i, j) + (i == j ? rho : 0.0);
      for (int k = std::max(0, i - bw); k < j; ++k) if (k >= j - bw) v -= s.l_at(i, k) * s.l_at(j, k);
      if (i == j) { if (!(v > 0.0)) return false; s.l_at(i, i) = std::sqrt(v); }
      else s.l_at(i, j) = v / s.l_at(j, j);
    }
  return true;
}
void pn_chol_solve(PnSys& s, std::vector<double>& b) { /* (L L') x = b in place */
  const int M = s.M, bw = s.bw;
  for (int i = 0; i < M; ++i) { double v = b[i]; for (int k = std::max(0, i - bw); k < i; ++k) v -= s.l_at(i, k) * b[k]; b[i] = v / s.l_at(i, i); }
  for (int i = M - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k <= std::min(M - 1, i + bw); ++k) v -= s.l_at(k, i) * b[k]; b[i] = v / s.l_at(i, i); }
}
void pn_S_mul(PnSys& s, const std::vector<double>& x, std::vector<double>& y) { /* y = S x through D W D' (S itself is only stored lower) */
  std::vector<double> tz(s.nv, 0.0);
  for (int i = 0; i < s.M; ++i) { const PnRow& r = s.rows[i]; for (int c = 0; c < r.len; ++c) tz[r.v0 + c] += r.coef[c] * x[i]; }
  for (int c = 0; c < s.nv; ++c) tz[c] *= s.W[c];
  y.assign(s.M, 0.0);
  for (int i = 0; i < s.M; ++i) { const PnRow& r = s.rows[i]; double v = 0.0; for (int c = 0; c < r.len; ++c) v += r.coef[c] * tz[r.v0 + c]; y[i] = v; }
}
/* Altro reg_solve: x = (S + rho I)^-1 b refined against the unregularised S */
void pn_reg_solve(PnSys& s, const std::vector<double>& b, std::vector<double>& x) {
  x = b; pn_chol_solve(s, x);
  std::vector<double> r, Sx;
  for (int it = 0; it < PN_REG_SOLVE_ITERS; ++it) {
    pn_S_mul(s, x, Sx);
    r.resize(s.M);
    double n2 = 0.0;
    for (int i = 0; i < s.M; ++i) { r[i] = b[i] - Sx[i]; n2 += r[i] * r[i]; }
    if (std::sqrt(n2) < PN_REG_SOLVE_TOL) break;
    pn_chol_solve(s, r);
    for (int i = 0; i < s.M; ++i) x[i] += r[i];
  }
}

struct PnResult { bool ran = false, failed = false; double viol = 0.0; };

/* Altro _projection_solve! on the working copy (X, U); the active set is the one pn_residual(refresh) left in s */
PnResult pn_projection(const Problem& P, const Traj& t, std::vector<double>& X, std::vector<double>& U, PnSys& s, double viol0) {
  const int N = P.N, n = P.n, m = P.m, ne = P.ne, nc = ne + m;
  PnResult res; res.ran = true; res.viol = viol0;
  pn_linearise(P, t, X.data(), U.data(), s);
  if (!pn_factor(s, P.opts.rho_chol)) { res.failed = true; return res; }
  std::vector<double> dl, dZ(s.nv), Xb(X.size()), Ub(U.size()), dn;
  double viol_prev = viol0;
  for (int count = 0; count < PN_REFINEMENTS; ++count) {
    pn_reg_solve(s, s.d, dl);
    std::fill(dZ.begin(), dZ.end(), 0.0);
    for (int i = 0; i < s.M; ++i) { const PnRow& r = s.rows[i]; for (int c = 0; c < r.len; ++c) dZ[r.v0 + c] += r.coef[c] * dl[i]; }
    for (int c = 0; c < s.nv; ++c) dZ[c] = -(s.W[c] * dZ[c]);
    double alpha = 1.0, v = 0.0; bool accepted = false;
    for (int ls = 0; ls < PN_LS_TRIALS; ++ls) {
      double step[MAXN];
      for (int k = 0; k < N; ++k) {
        for (int i = 0; i < ne; ++i) step[i] = alpha * dZ[(size_t)k * nc + i];
        state_add(P.M, &X[(size_t)k * n], step, &Xb[(size_t)k * n]);
        if (k < N - 1) for (int j = 0; j < m; ++j) Ub[(size_t)k * m + j] = U[(size_t)k * m + j] + alpha * dZ[(size_t)k * nc + ne + j];
      }
      v = pn_residual(P, t, Xb.data(), Ub.data(), s, false, dn);
      if (std::getenv("ORACLE_PN_VERBOSE")) {
        double mz = 0; int iz = 0; for (int c = 0; c < s.nv; ++c) if (std::fabs(dZ[c]) > mz) { mz = std::fabs(dZ[c]); iz = c; }
        int ir = 0; for (int i = 0; i < s.M; ++i) if (std::fabs(dn[i]) > std::fabs(dn[ir])) ir = i;
        int kr = 0; while (kr + 1 < N && s.goff[kr + 1] <= ir) ++kr;
        double ml = 0; for (int i = 0; i < s.M; ++i) ml = std::fmax(ml, std::fabs(dl[i]));
        std::fprintf(stderr, "    trial alpha %.3g: viol %.3e  max|dZ| %.3e at knot %d entry %d  worst row: knot %d local %d  max|dl| %.3e\n", alpha, v, mz, iz / nc, iz % nc, kr, ir - s.goff[kr], ml);
      }
      if (v < viol_prev) { accepted = true; break; }
      alpha *= 0.5;
    }
    if (std::getenv("ORACLE_PN_VERBOSE")) std::fprintf(stderr, "  pn refine %d: viol %.3e -> %.3e alpha %.3g accepted %d rows %d\n", count, viol_prev, v, alpha, (int)accepted, s.M);
    if (!accepted) break;
    X = Xb; U = Ub; s.d = dn;
    const double before = viol_prev;
    viol_prev = v;
    if (v < P.opts.constraint_tolerance) break;
    if (before < 1.0) { if (std::log10(v) / std::log10(before) < P.opts.r_threshold) break; }
    else if (!(v < 0.5 * before)) break;
  }
  res.viol = viol_prev;
  return res;
}

/* Altro solve!(::ProjectedNewtonSolver) on the trajectory held by t */
void pn_solve(const Problem& P, Traj& t) {
  PnSys s;
  std::vector<double> X = t.X, U = t.U;
  t.iterations_pn = 0;
  bool failed = false;
  for (int k = 0; k < P.N && !failed; ++k) if (pn_candidate_count(P, k) > PN_MAX_ROWS) failed = true;
  for (int step = 0; !failed; ++step) {
    const double viol = pn_residual(P, t, X.data(), U.data(), s, true, s.d);
    if (viol <= P.opts.constraint_tolerance || step > P.opts.n_steps) break; /* Altro projection_solve!: while count <= n_steps */
    if (std::getenv("ORACLE_PN_VERBOSE")) std::fprintf(stderr, "pn step %d: viol %.3e rows %d\n", step, viol, s.M);
    const PnResult r = pn_projection(P, t, X, U, s, viol);
    t.iterations_pn++;
    if (r.failed) failed = true;
  }
  t.X = X; t.U = U;
  const double cm = P.cons.empty() ? 0.0 : max_violation(P, t);
  const double df = dynamics_defect(P, t);
  t.c_max = (std::isnan(cm) || std::isnan(df)) ? std::numeric_limits<double>::quiet_NaN() : std::fmax(cm, df);
  t.status = (t.c_max <= P.opts.constraint_tolerance) ? TO_SOLVE_SUCCEEDED : TO_PROJECTION_FAIL;
  t.J = total_cost(P, t, t.X.data(), t.U.data(), false);
}
