/*
 * trajopt_oracle.cpp — TEST INFRASTRUCTURE ONLY.  CPU oracle for the batched iLQR / AL hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * library; the product (libtrajopt_hip.so) never links or falls back to it.
 *
 * It restates, one trajectory at a time (OpenMP over the batch), the reference's in-tree arithmetic
 * (costs src/cost_functions.jl + src/lie_costs.jl, constraints src/constraints.jl, cones src/cones.jl,
 * rollout!/cost loops src/problem.jl:330-340, src/objective.jl:89-106) and the out-of-tree pieces the
 * reference delegates to RobotDynamics / RobotZoo / Altro.
 *
 * Parity pinning (tests/test_oracle_golden.py): G1 bit-level Quadrotor RK4 rollout
 * (examples/Internal API.ipynb cell 6), G2 error-state A/B entries (cell 12), G3 Cartpole iLQR 84
 * iterations / J=1.44974 on the legacy RK3+dt-scaled stack (examples/Cartpole.ipynb cell 25), G5 J_1,
 * G6 error-state cost Hessian block (cell 35) and the closed-form known-answer tests of the
 * reference's test suite.  The Altro-side constants that no reference artefact pins (regularisation
 * schedule, AL update rules) are defined HERE and the GPU path must match them.
 *
 * API: oracle_<name> mirrors to_<name> of include/trajopt_hip.h (same descriptors, same host layouts).
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "oracle_math.h"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

struct ConInfo {
  to_constraint_desc d;
  int p = 0;
  int width = 0;     /* n for state constraints, n+m for stage constraints */
  int k1 = 0, k2 = 0; /* 0-based inclusive */
  size_t dual_off = 0;
};

struct Problem {
  Model M;
  std::vector<to_step_model> steps;  /* TO_MODEL_VECTOR */
  int integrator = TO_RK4;
  int n = 0, m = 0, ne = 0, N = 0, B = 0;
  std::vector<double> dt;
  std::vector<to_cost_desc> costs;
  std::vector<int> cost_index;
  std::vector<ConInfo> cons;
  size_t n_duals = 0;
  to_solver_opts opts;
};

struct Traj {
  std::vector<double> x0, X, U, Xb, Ub;
  std::vector<double> A, Bm;                 /* [(N-1)] ne*ne, ne*m row-major */
  std::vector<double> Qxx, Quu, Qux, qx, qu;  /* [N] */
  std::vector<double> K, d;                   /* [(N-1)] m*ne, m */
  std::vector<double> Sall, sall;             /* [N] ne*ne, ne: cost-to-go of the last backward pass (oracle_get_cost_to_go) */
  std::vector<double> gl;                     /* per-trajectory linear cost terms, [n_costs] (n + m): what q, r differ by from the descriptor (empty: none) */
  std::vector<double> lambda, mu;             /* duals (n_duals), penalties (ncons) */
  std::vector<std::vector<double>> cpar;      /* per-trajectory constraint parameters (oracle_set_constraint_params_batch): cpar[i] replaces
                                                 parameters of constraint i's descriptor for THIS trajectory — a GoalConstraint's target (the
                                                 leading p), a LinearConstraint's b (behind A) — (empty: the shared ones) */
  double dV[2] = {0, 0};
  double rho = 0, drho = 0;
  double J = 0, dJ = 0, grad = 0, c_max = 0;
  int iterations = 0, iterations_outer = 0, iterations_pn = 0, status = TO_UNSOLVED, ls_index = -1;
  int dJ_zero_counter = 0;
  bool ls_failed = false, zero_step = false;
};

}  // namespace

struct oracle_handle {
  Problem P;
  std::vector<Traj> T;
  int threads = 1;
};

namespace {

void default_opts(to_solver_opts* o) {
  std::memset(o, 0, sizeof(*o));
  o->cost_tolerance = 1e-4; o->gradient_tolerance = 10.0; o->iterations = 300; o->dJ_counter_limit = 10;
  o->iterations_linesearch = 20; o->line_search_lower_bound = 1e-8; o->line_search_upper_bound = 10.0;
  o->line_search_decrease_factor = 0.5; o->bp_reg_initial = 0.0; o->bp_reg_increase_factor = 1.6;
  o->bp_reg_min = 1e-8; o->bp_reg_max = 1e8; o->bp_reg_fp = 10.0; o->max_cost_value = 1e8;
  o->max_state_value = 1e8; o->max_control_value = 1e8; o->constraint_tolerance = 1e-6;
  o->cost_tolerance_intermediate = 1e-4; o->penalty_initial = 1.0; o->penalty_scaling = 10.0;
  o->penalty_max = 1e8; o->dual_max = 1e8; o->iterations_outer = 30; o->cost_dt_scaling = 0;
  o->iterations_total = 1000;
  o->projected_newton_tolerance = 1e-3; o->active_set_tolerance_pn = 1e-3; o->rho_chol = 1e-8; o->rho_primal = 1e-8;
  o->r_threshold = 1.1; o->n_steps = 2; o->projected_newton = 1;
}

/* ------------------------------------------------------------------ descriptor validation */
int validate_opts(const to_solver_opts& o) {
  auto bad = [](const char* what) { return fail(TO_ERR_ARGUMENT, std::string("solver option out of range: ") + what); };
  if (!(o.cost_tolerance >= 0) || !(o.cost_tolerance_intermediate >= 0) || !(o.gradient_tolerance >= 0) || !(o.constraint_tolerance >= 0))
    return bad("tolerances must be >= 0");
  if (o.iterations < 0 || o.iterations_outer < 0 || o.iterations_total < 0 || o.dJ_counter_limit < 0) return bad("iteration counts must be >= 0");
  if (o.iterations_linesearch < 1 || o.iterations_linesearch > 64) return bad("iterations_linesearch must be in 1..64");
  if (!(o.line_search_decrease_factor > 0.0 && o.line_search_decrease_factor < 1.0)) return bad("line_search_decrease_factor must be in (0,1)");
  if (!(o.line_search_lower_bound >= 0.0) || !(o.line_search_upper_bound > o.line_search_lower_bound)) return bad("line-search bounds must satisfy 0 <= lower < upper");
  if (!(o.bp_reg_increase_factor > 1.0)) return bad("bp_reg_increase_factor must be > 1");
  if (!(o.bp_reg_initial >= 0.0) || !(o.bp_reg_min >= 0.0) || !(o.bp_reg_max > o.bp_reg_min) || !(o.bp_reg_fp >= 0.0)) return bad("regularisation bounds");
  if (!(o.penalty_initial > 0.0) || !(o.penalty_scaling >= 1.0) || !(o.penalty_max >= o.penalty_initial) || !(o.dual_max > 0.0)) return bad("penalty parameters");
  if (!(o.max_cost_value > 0.0) || !(o.max_state_value > 0.0) || !(o.max_control_value > 0.0)) return bad("max_*_value must be > 0");
  if (o.cost_dt_scaling != 0 && o.cost_dt_scaling != 1) return bad("cost_dt_scaling must be 0 or 1");
  if (o.al_full_newton != 0 && o.al_full_newton != 1) return bad("al_full_newton must be 0 or 1");
  if (!(o.projected_newton_tolerance >= 0) || !(o.active_set_tolerance_pn >= 0)) return bad("projected-Newton tolerances must be >= 0");
  if (!(o.rho_chol >= 0) || !(o.rho_primal > 0) || !(o.r_threshold > 0)) return bad("rho_chol >= 0, rho_primal > 0, r_threshold > 0");
  if (o.n_steps < 0 || (o.projected_newton != 0 && o.projected_newton != 1)) return bad("n_steps must be >= 0, projected_newton 0 or 1");
  return TO_OK;
}

int validate_constraint(const Problem& P, const to_constraint_desc& d, ConInfo* out) {
  const int n = P.n, m = P.m, nz = n + m;
  ConInfo ci; ci.d = d;
  if (d.k_first < 1 || d.k_last > P.N || d.k_first > d.k_last)
    return fail(TO_ERR_ASSERTION, "constraint knot range outside 1:N");
  if (d.n_inds < 0 || d.n_inds > TO_MAX_CON_INDS || d.n_params < 0 || d.n_params > TO_MAX_CON_PARAMS)
    return fail(TO_ERR_ARGUMENT, "constraint inds/params count out of range");
  for (int i = 0; i < d.n_inds; ++i)
    for (int j = i + 1; j < d.n_inds; ++j)
      if (d.inds[i] == d.inds[j]) return fail(TO_ERR_ARGUMENT, "constraint indices must be distinct");
  switch (d.kind) {
    case TO_CON_GOAL:
      if (d.sense != TO_CONE_ZERO) return fail(TO_ERR_ARGUMENT, "GoalConstraint sense must be Equality");
      if (d.n_params != d.n_inds) return fail(TO_ERR_DIMENSION_MISMATCH, "GoalConstraint: length(xf) != length(inds)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "GoalConstraint index outside state");
      ci.width = n; break;
    case TO_CON_BOUND:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "BoundConstraint sense must be Inequality");
      if (d.n_params != 2 * nz) return fail(TO_ERR_DIMENSION_MISMATCH, "BoundConstraint needs z_max and z_min of length n+m");
      for (int i = 0; i < nz; ++i)
        if (!(d.params[i] >= d.params[nz + i])) /* src/constraints.jl:708-713 */
          return fail(TO_ERR_ARGUMENT, "Upper bounds must be greater than or equal to lower bounds");
      ci.width = nz; break;
    case TO_CON_NORM:
      if (d.n_params != 1) return fail(TO_ERR_ARGUMENT, "NormConstraint needs one parameter (val)");
      if (!(d.params[0] >= 0)) return fail(TO_ERR_ASSERTION, "Value must be greater than or equal to zero");
      if (d.sense != TO_CONE_ZERO && d.sense != TO_CONE_NEGATIVE_ORTHANT && d.sense != TO_CONE_SECOND_ORDER)
        return fail(TO_ERR_ARGUMENT, "NormConstraint sense must be Equality, Inequality or SecondOrderCone");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > nz) return fail(TO_ERR_DIMENSION_MISMATCH, "NormConstraint index outside [x;u]");
      ci.width = nz; break;
    case TO_CON_CIRCLE:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "CircleConstraint sense must be Inequality");
      if (d.n_inds != 2 || d.n_params % 3 != 0 || d.n_params == 0) return fail(TO_ERR_ASSERTION, "Lengths of xc, yc, and radius must be equal");
      for (int i = 0; i < 2; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "CircleConstraint index outside state");
      ci.width = n; break;
    case TO_CON_SPHERE:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "SphereConstraint sense must be Inequality");
      if (d.n_inds != 3 || d.n_params % 4 != 0 || d.n_params == 0) return fail(TO_ERR_ASSERTION, "Lengths of xc, yc, zc, and radius must be equal");
      for (int i = 0; i < 3; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "SphereConstraint index outside state");
      ci.width = n; break;
    case TO_CON_LINEAR:
      if (d.sense != TO_CONE_ZERO && d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "LinearConstraint sense must be Equality or Inequality");
      if (d.n_inds < 1 || d.n_params % (d.n_inds + 1) != 0 || d.n_params == 0) return fail(TO_ERR_ASSERTION, "size(A,1) == length(b)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > nz) return fail(TO_ERR_DIMENSION_MISMATCH, "LinearConstraint index outside [x;u]");
      ci.width = nz; break;
    case TO_CON_COLLISION:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "CollisionConstraint sense must be Inequality");
      if (d.n_inds < 2 || d.n_inds % 2 != 0) return fail(TO_ERR_ASSERTION, "Position dimensions must be of equal length"); /* src/constraints.jl:349 */
      if (d.n_params != 1) return fail(TO_ERR_ARGUMENT, "CollisionConstraint needs one parameter (radius)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "CollisionConstraint index outside state");
      ci.width = n; break;
    case TO_CON_QUATVEC:
      if (d.sense != TO_CONE_ZERO) return fail(TO_ERR_ARGUMENT, "QuatVecEq sense must be Equality");
      if (d.n_inds != 4 || d.n_params != 4) return fail(TO_ERR_ARGUMENT, "QuatVecEq needs 4 quaternion indices and a 4-vector qf");
      for (int i = 0; i < 4; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "QuatVecEq index outside state");
      ci.width = n; break;
    default: return fail(TO_ERR_UNSUPPORTED, "unknown constraint kind");
  }
  ci.p = constraint_output_dim(d, n, m);
  if (ci.p < 1 || ci.p > TO_MAX_P) return fail(TO_ERR_UNSUPPORTED, "constraint output dimension outside 1..TO_MAX_P");
  if (d.p != 0 && d.p != ci.p) return fail(TO_ERR_DIMENSION_MISMATCH, "constraint output dimension mismatch");
  ci.k1 = d.k_first - 1; ci.k2 = d.k_last - 1;
  *out = ci;
  return TO_OK;
}

int validate_cost(const Problem& P, const to_cost_desc& c) {
  if (c.kind != TO_COST_DIAGONAL && c.kind != TO_COST_QUADRATIC && c.kind != TO_COST_DIAGONAL_QUAT && c.kind != TO_COST_ERROR_QUADRATIC)
    return fail(TO_ERR_UNSUPPORTED, "unknown cost kind");
  const int rot = P.M.rot();
  if (c.kind == TO_COST_ERROR_QUADRATIC) {  // needs the rigid-body state layout [r; attitude; v; w]
    if (rot < 0) return fail(TO_ERR_ARGUMENT, "ErrorQuadratic needs a rigid-body model");
    if ((int)c.w != rot || c.w != (double)rot) return fail(TO_ERR_ARGUMENT, "ErrorQuadratic: w must name the model's attitude representation (to_rotation)");
    if (rot == TO_ROT_QUATERNION)
      for (int i = 0; i < 4; ++i) if (c.q_ind[i] != 4 + i) return fail(TO_ERR_UNSUPPORTED, "ErrorQuadratic: q_ind must be 4:7");
  }
  if (c.kind == TO_COST_DIAGONAL_QUAT) {
    if (rot > TO_ROT_QUATERNION) return fail(TO_ERR_ARGUMENT, "DiagonalQuatCost needs a state that carries a unit quaternion");
    for (int i = 0; i < 4; ++i) if (c.q_ind[i] < 1 || c.q_ind[i] > P.n) return fail(TO_ERR_DIMENSION_MISMATCH, "quat_ind outside state");
  }
  return TO_OK;
}

int build_problem(const to_problem_desc* desc, const to_solver_opts* opts, Problem* P) {
  if (!desc) return fail(TO_ERR_NULL, "null descriptor");
  if (desc->abi_version != TO_ABI_VERSION) return fail(TO_ERR_ARGUMENT, "ABI version mismatch");
  int n, m, ne;
  if (model_dims(desc->model, desc->model_params, &n, &m, &ne) != 0) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  if (desc->n != n || desc->m != m) return fail(TO_ERR_DIMENSION_MISMATCH, "Model and problem dimensions are inconsistent");
  if (opts) { int r = validate_opts(*opts); if (r) return r; }
  if (desc->N < 2) return fail(TO_ERR_ASSERTION, "N must be at least 2");
  if (desc->B < 1) return fail(TO_ERR_ARGUMENT, "batch must be positive");
  if (!(desc->tf > desc->t0)) return fail(TO_ERR_ASSERTION, "Final time must be greater than initial time"); /* src/problem.jl:50 */
  if (desc->integrator < TO_RK4 || desc->integrator > TO_EULER) return fail(TO_ERR_UNSUPPORTED, "unknown integrator");
  if (desc->model == TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) {
    const double S = desc->model_params[1];
    if (!(S >= 1.0) || !(S <= (double)(desc->N - 2)) || S != std::floor(S))
      return fail(TO_ERR_ARGUMENT, "hybrid double integrator: params[1] (time steps of the first model) must be an integer in 1 .. N-2");
  }
  if (desc->model == TO_MODEL_VECTOR) { /* RD.dims(models), src/dynamics.jl:15-31 */
    if (!desc->step_models) return fail(TO_ERR_NULL, "TO_MODEL_VECTOR needs step_models[N-1]");
    P->steps.assign(desc->step_models, desc->step_models + desc->N - 1);
    for (int k = 0; k < desc->N - 1; ++k) {
      const to_step_model& s = P->steps[k];
      if (s.n < 1 || s.n > TO_VECTOR_N || s.m < 1 || s.m > TO_VECTOR_M || s.n_out < 1 || s.n_out > TO_VECTOR_N) return fail(TO_ERR_UNSUPPORTED, "model vector: step dimensions outside (6, 3)");
      if (s.kind == TO_STEP_DOUBLE_INTEGRATOR) { if (s.n != 2 * s.m || s.n_out != s.n) return fail(TO_ERR_DIMENSION_MISMATCH, "double integrator step: n = 2D, m = D, n_out = n"); if (!(s.params[0] > 0.0)) return fail(TO_ERR_ARGUMENT, "double integrator step: mass must be positive"); }
      else if (s.kind == TO_STEP_CARTPOLE) { if (s.n != 4 || s.m != 1 || s.n_out != 4) return fail(TO_ERR_DIMENSION_MISMATCH, "Cartpole step: (n, m, n_out) = (4, 1, 4)"); }
      else if (s.kind != TO_STEP_LINEAR_MAP) return fail(TO_ERR_UNSUPPORTED, "unknown step model kind");
      if (k + 1 < desc->N - 1 && P->steps[k + 1].n != s.n_out) return fail(TO_ERR_DIMENSION_MISMATCH, "Model mismatch at time step " + std::to_string(k + 1));
    }
  }
  P->M.id = desc->model; P->M.n = n; P->M.m = m; P->M.ne = ne;
  std::memcpy(P->M.p, desc->model_params, sizeof(P->M.p));
  P->integrator = desc->integrator; P->n = n; P->m = m; P->ne = ne; P->N = desc->N; P->B = desc->B;
  P->dt.resize(desc->N - 1);
  if (desc->dt) {
    double s = 0;
    for (int k = 0; k < desc->N - 1; ++k) { P->dt[k] = desc->dt[k]; s += desc->dt[k]; if (!(desc->dt[k] > 0)) return fail(TO_ERR_ASSERTION, "dt must be positive"); }
    if (std::fabs(s - (desc->tf - desc->t0)) > 1e-8 * std::fmax(1.0, std::fabs(desc->tf - desc->t0)))
      return fail(TO_ERR_ASSERTION, "Time steps are inconsistent with the final time"); /* test/problems_tests.jl:85 */
  } else {
    for (int k = 0; k < desc->N - 1; ++k) P->dt[k] = (desc->tf - desc->t0) / (desc->N - 1);
  }
  if (desc->n_costs < 1 || !desc->costs) return fail(TO_ERR_ARGUMENT, "objective needs at least one cost function");
  P->costs.assign(desc->costs, desc->costs + desc->n_costs);
  for (auto& c : P->costs) { int r = validate_cost(*P, c); if (r) return r; }
  P->cost_index.resize(desc->N);
  if (desc->cost_index) {
    for (int k = 0; k < desc->N; ++k) {
      if (desc->cost_index[k] < 0 || desc->cost_index[k] >= desc->n_costs) return fail(TO_ERR_DIMENSION_MISMATCH, "cost_index outside costs"); /* src/problem.jl:66 */
      P->cost_index[k] = desc->cost_index[k];
    }
  } else {
    if (desc->n_costs < 2) return fail(TO_ERR_ARGUMENT, "Objective(stage, terminal, N) needs two cost functions");
    for (int k = 0; k < desc->N; ++k) P->cost_index[k] = (k == desc->N - 1) ? 1 : 0;
  }
  P->cons.clear(); P->n_duals = 0;
  if (desc->n_constraints < 0 || (desc->n_constraints > 0 && !desc->constraints)) return fail(TO_ERR_ARGUMENT, "bad constraint list");
  for (int i = 0; i < desc->n_constraints; ++i) {
    ConInfo ci; int r = validate_constraint(*P, desc->constraints[i], &ci); if (r) return r;
    ci.dual_off = P->n_duals;
    P->n_duals += (size_t)ci.p * (ci.k2 - ci.k1 + 1);
    P->cons.push_back(ci);
  }
  if (opts) P->opts = *opts; else default_opts(&P->opts);
  return TO_OK;
}

void alloc_traj(const Problem& P, Traj& t) {
  const int n = P.n, m = P.m, ne = P.ne, N = P.N;
  t.x0.assign(n, 0.0);
  t.X.assign((size_t)N * n, std::numeric_limits<double>::quiet_NaN()); /* X0 = NaN default, src/problem.jl:83 */
  t.U.assign((size_t)(N - 1) * m, 0.0);                                  /* U0 = 0 default, src/problem.jl:84 */
  t.Xb = t.X; t.Ub = t.U;
  t.A.assign((size_t)(N - 1) * ne * ne, 0.0); t.Bm.assign((size_t)(N - 1) * ne * m, 0.0);
  t.Qxx.assign((size_t)N * ne * ne, 0.0); t.Quu.assign((size_t)N * m * m, 0.0); t.Qux.assign((size_t)N * m * ne, 0.0);
  t.qx.assign((size_t)N * ne, 0.0); t.qu.assign((size_t)N * m, 0.0);
  t.K.assign((size_t)(N - 1) * m * ne, 0.0); t.d.assign((size_t)(N - 1) * m, 0.0);
  t.lambda.assign(P.n_duals, 0.0); t.mu.assign(P.cons.size(), P.opts.penalty_initial);
  t.rho = P.opts.bp_reg_initial; t.drho = 0.0;
}

/* ------------------------------------------------------------------ per-trajectory operators */
void rollout(const Problem& P, Traj& t) { /* src/problem.jl:334-340 */
  const int n = P.n, m = P.m;
  for (int i = 0; i < n; ++i) t.X[i] = t.x0[i];
  for (int k = 0; k < P.N - 1; ++k)
    knot_step(P.M, P.integrator, k, &t.X[(size_t)k * n], &t.U[(size_t)k * m], P.dt[k], &t.X[(size_t)(k + 1) * n]);
}

inline void knot_z(const Problem& P, const double* X, const double* U, int k, double* z) {
  const int n = P.n, m = P.m;
  for (int i = 0; i < n; ++i) z[i] = X[(size_t)k * n + i];
  for (int j = 0; j < m; ++j) z[n + j] = (k < P.N - 1) ? U[(size_t)k * m + j] : 0.0; /* terminal control = 0 */
}

/* gl: the trajectory's per-trajectory linear cost terms (oracle_set_cost_linear_batch; set_LQR_goal! with one goal per trajectory,
 * src/cost_functions.jl:249-258) or nullptr */
double objective_knot(const Problem& P, const double* X, const double* U, int k, const double* gl = nullptr) {
  double z[MAXZ]; knot_z(P, X, U, k, z);
  const to_cost_desc& C = P.costs[P.cost_index[k]];
  double J = cost_evaluate(C, P.n, P.m, z, z + P.n);
  if (gl) { const double* g = gl + (size_t)P.cost_index[k] * (P.n + P.m); for (int i = 0; i < P.n + P.m; ++i) J += g[i] * z[i]; }
  if (P.opts.cost_dt_scaling && k < P.N - 1) J *= P.dt[k];
  return J;
}

/* The descriptor constraint i has for trajectory t: the shared one, or — set_goal_state!(prob, Xf; constraint = true) with one goal per
 * trajectory, src/problem.jl:303-309 — a copy whose leading parameters are the trajectory's own. */
struct EffDesc {
  to_constraint_desc tmp;
  const to_constraint_desc& get(const Traj& t, size_t i, const ConInfo& ci) {
    if (i >= t.cpar.size() || t.cpar[i].empty()) return ci.d;
    tmp = ci.d;
    const size_t off = (ci.d.kind == TO_CON_LINEAR) ? (size_t)ci.p * ci.d.n_inds : 0;
    for (size_t r = 0; r < t.cpar[i].size(); ++r) tmp.params[off + r] = t.cpar[i][r];
    return tmp;
  }
};

/* AL penalty of constraint ci at one knot; lambda points at the p duals of that knot */
double al_term(const ConInfo& ci, const to_constraint_desc& D, int n, int m, const double* z, const double* lambda, double mu) {
  double c[TO_MAX_P];
  constraint_evaluate(D, n, m, z, c, nullptr);
  const int p = ci.p;
  double J = 0.0;
  if (ci.d.sense == TO_CONE_ZERO) {
    for (int i = 0; i < p; ++i) J += lambda[i] * c[i] + 0.5 * mu * c[i] * c[i];
  } else if (ci.d.sense == TO_CONE_NEGATIVE_ORTHANT) {
    for (int i = 0; i < p; ++i) {
      bool active = (c[i] >= 0.0) || (lambda[i] > 0.0);
      J += lambda[i] * c[i] + (active ? 0.5 * mu * c[i] * c[i] : 0.0);
    }
  } else { /* SecondOrderCone */
    double lb[TO_MAX_P], lp[TO_MAX_P];
    for (int i = 0; i < p; ++i) lb[i] = lambda[i] - mu * c[i];
    cone_projection(TO_CONE_SECOND_ORDER, lb, lp, p);
    double a = 0.0, b = 0.0;
    for (int i = 0; i < p; ++i) { a += lp[i] * lp[i]; b += lambda[i] * lambda[i]; }
    J = (a - b) / (2.0 * mu);
  }
  return J;
}

double al_knot(const Problem& P, const Traj& t, const double* X, const double* U, int k) {
  double z[MAXZ]; knot_z(P, X, U, k, z);
  double J = 0.0;
  for (size_t i = 0; i < P.cons.size(); ++i) {
    const ConInfo& ci = P.cons[i];
    if (k < ci.k1 || k > ci.k2) continue;
    EffDesc ed;
    J += al_term(ci, ed.get(t, i, ci), P.n, P.m, z, &t.lambda[ci.dual_off + (size_t)(k - ci.k1) * ci.p], t.mu[i]);
  }
  return J;
}

double total_cost(const Problem& P, const Traj& t, const double* X, const double* U, bool with_al) {
  double J = 0.0;
  for (int k = 0; k < P.N; ++k) {
    double Jk = objective_knot(P, X, U, k, t.gl.empty() ? nullptr : t.gl.data());
    if (with_al && !P.cons.empty()) Jk += al_knot(P, t, X, U, k);
    J += Jk;
  }
  return J;
}

double max_violation(const Problem& P, const Traj& t) {
  double cmax = 0.0, z[MAXZ], c[TO_MAX_P], pc[TO_MAX_P];
  EffDesc ed;
  for (size_t ic = 0; ic < P.cons.size(); ++ic) {
    const ConInfo& ci = P.cons[ic];
    const to_constraint_desc& D = ed.get(t, ic, ci);
    for (int k = ci.k1; k <= ci.k2; ++k) {
      knot_z(P, t.X.data(), t.U.data(), k, z);
      constraint_evaluate(D, P.n, P.m, z, c, nullptr);
      for (int i = 0; i < ci.p; ++i) {
        double v;
        if (ci.d.sense == TO_CONE_ZERO) v = std::fabs(c[i]);
        else if (ci.d.sense == TO_CONE_NEGATIVE_ORTHANT) v = std::fmax(0.0, c[i]);
        else { if (i == 0) cone_projection(TO_CONE_SECOND_ORDER, c, pc, ci.p); v = std::fabs(c[i] - pc[i]); }
        if (v > cmax || std::isnan(v)) cmax = v;
      }
    }
  }
  return cmax;
}

void dual_update(const Problem& P, Traj& t) {
  double z[MAXZ], c[TO_MAX_P], lb[TO_MAX_P];
  for (size_t i = 0; i < P.cons.size(); ++i) {
    const ConInfo& ci = P.cons[i];
    double mu = t.mu[i];
    EffDesc ed;
    const to_constraint_desc& D = ed.get(t, i, ci);
    for (int k = ci.k1; k <= ci.k2; ++k) {
      knot_z(P, t.X.data(), t.U.data(), k, z);
      constraint_evaluate(D, P.n, P.m, z, c, nullptr);
      double* lam = &t.lambda[ci.dual_off + (size_t)(k - ci.k1) * ci.p];
      if (ci.d.sense == TO_CONE_ZERO) {
        for (int r = 0; r < ci.p; ++r) lam[r] = std::fmax(-P.opts.dual_max, std::fmin(P.opts.dual_max, lam[r] + mu * c[r]));
      } else if (ci.d.sense == TO_CONE_NEGATIVE_ORTHANT) {
        for (int r = 0; r < ci.p; ++r) lam[r] = std::fmin(P.opts.dual_max, std::fmax(0.0, lam[r] + mu * c[r]));
      } else {
        for (int r = 0; r < ci.p; ++r) lb[r] = lam[r] - mu * c[r];
        cone_projection(TO_CONE_SECOND_ORDER, lb, lam, ci.p);
      }
    }
    t.mu[i] = std::fmin(mu * P.opts.penalty_scaling, P.opts.penalty_max);
  }
}

/* full-state cost (+AL) expansion at knot k: grad (nz), hess (nz*nz row-major) */
void knot_expansion_full(const Problem& P, const Traj& t, const double* X, const double* U, bool with_al, int k, double* grad, double* hess) {
  const int n = P.n, m = P.m, nz = n + m;
  const bool terminal = (k == P.N - 1);
  double z[MAXZ]; knot_z(P, X, U, k, z);
  cost_expansion(P.costs[P.cost_index[k]], n, m, z, z + n, terminal, grad, hess);
  if (!t.gl.empty()) { const double* g = &t.gl[(size_t)P.cost_index[k] * nz]; for (int i = 0; i < (terminal ? n : nz); ++i) grad[i] += g[i]; }
  if (P.opts.cost_dt_scaling && !terminal) {
    for (int i = 0; i < nz; ++i) grad[i] *= P.dt[k];
    for (int i = 0; i < nz * nz; ++i) hess[i] *= P.dt[k];
  }
  double c[TO_MAX_P], jac[TO_MAX_P * MAXZ], y[TO_MAX_P], W[TO_MAX_P * TO_MAX_P];
  for (size_t i = 0; with_al && i < P.cons.size(); ++i) {
    const ConInfo& ci = P.cons[i];
    if (k < ci.k1 || k > ci.k2) continue;
    const int p = ci.p; const double mu = t.mu[i];
    const double* lam = &t.lambda[ci.dual_off + (size_t)(k - ci.k1) * p];
    EffDesc ed;
    const to_constraint_desc& D = ed.get(t, i, ci);
    constraint_evaluate(D, n, m, z, c, jac);
    std::memset(W, 0, sizeof(double) * p * p);
    if (ci.d.sense == TO_CONE_ZERO) {
      for (int r = 0; r < p; ++r) { y[r] = lam[r] + mu * c[r]; W[r * p + r] = mu; }
    } else if (ci.d.sense == TO_CONE_NEGATIVE_ORTHANT) {
      for (int r = 0; r < p; ++r) {
        bool active = (c[r] >= 0.0) || (lam[r] > 0.0);
        y[r] = lam[r] + (active ? mu * c[r] : 0.0);
        W[r * p + r] = active ? mu : 0.0;
      }
    } else { /* SOC: psi = (|Pi(lb)|^2 - |lam|^2)/(2mu), lb = lam - mu c */
      double lb[TO_MAX_P], lp[TO_MAX_P], Jp[TO_MAX_P * TO_MAX_P], Hp[TO_MAX_P * TO_MAX_P];
      for (int r = 0; r < p; ++r) lb[r] = lam[r] - mu * c[r];
      cone_projection(TO_CONE_SECOND_ORDER, lb, lp, p);
      cone_projection_jacobian(TO_CONE_SECOND_ORDER, lb, Jp, p);
      cone_projection_hessian(TO_CONE_SECOND_ORDER, lb, lp, Hp, p);
      for (int r = 0; r < p; ++r) { double s = 0.0; for (int q = 0; q < p; ++q) s += Jp[q * p + r] * lp[q]; y[r] = -s; }
      for (int r = 0; r < p; ++r) for (int q = 0; q < p; ++q) {
        double s = 0.0; for (int v = 0; v < p; ++v) s += Jp[v * p + r] * Jp[v * p + q];
        W[r * p + q] = mu * (s + Hp[r * p + q]);
      }
    }
    /* full Newton (opts.al_full_newton): + sum_r y_r d2c_r/dz2 with the multiplier estimate y; the SOC branch already carries the
       curvature of the projection and its constraints are linear in z */
    if (P.opts.al_full_newton && ci.d.sense != TO_CONE_SECOND_ORDER) constraint_hessian_add(D, n, m, z, y, hess, nz);
    /* grad += jac' y ; hess += jac' W jac */
    for (int a = 0; a < nz; ++a) { double s = 0.0; for (int r = 0; r < p; ++r) s += jac[r * nz + a] * y[r]; grad[a] += s; }
    double WJ[TO_MAX_P * MAXZ];
    for (int r = 0; r < p; ++r) for (int a = 0; a < nz; ++a) { double s = 0.0; for (int q = 0; q < p; ++q) s += W[r * p + q] * jac[q * nz + a]; WJ[r * nz + a] = s; }
    for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) { double s = 0.0; for (int r = 0; r < p; ++r) s += jac[r * nz + a] * WJ[r * nz + b]; hess[a * nz + b] += s; }
  }
  if (terminal) { /* no control at the terminal knot */
    for (int j = 0; j < m; ++j) grad[n + j] = 0.0;
    for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) if (a >= n || b >= n) hess[a * nz + b] = 0.0;
  }
}

/* error-state dynamics Jacobians of step k at (X, U): Ae (ne x ne), Be (ne x m), row-major */
void dynamics_blocks(const Problem& P, const double* X, const double* U, int k, double* Ae, double* Be) {
  const int n = P.n, m = P.m, ne = P.ne;
  static thread_local std::vector<double> A, Bf, G0, G1, T1;
  A.resize(n * n); Bf.resize(n * m); G0.resize(n * ne); G1.resize(n * ne); T1.resize(n * ne);
  const double* x = &X[(size_t)k * n]; const double* u = &U[(size_t)k * m];
  knot_step_jacobian(P.M, P.integrator, k, x, u, P.dt[k], A.data(), Bf.data());
  errstate_jacobian(P.M, x, G0.data());
  errstate_left_inverse(P.M, &X[(size_t)(k + 1) * n], G1.data()); /* E(x_{k+1}): ne x n (= G' for unit quaternions) */
  matmul(A.data(), G0.data(), T1.data(), n, n, ne); /* A G_k : n x ne */
  for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) { double s = 0.0; for (int r = 0; r < n; ++r) s += G1[i * n + r] * T1[r * ne + j]; Ae[i * ne + j] = s; }
  for (int i = 0; i < ne; ++i) for (int j = 0; j < m; ++j) { double s = 0.0; for (int r = 0; r < n; ++r) s += G1[i * n + r] * Bf[r * m + j]; Be[i * m + j] = s; }
}

/* error-state cost (+AL when with_al) blocks of knot k at (X, U) */
void cost_blocks(const Problem& P, const Traj& t, const double* X, const double* U, bool with_al, int k,
                 double* Qxx, double* Quu, double* Qux, double* qx, double* qu) {
  const int n = P.n, m = P.m, ne = P.ne, nz = n + m;
  static thread_local std::vector<double> G0, grad, hess, T2;
  G0.resize(n * ne); grad.resize(nz); hess.resize(nz * nz); T2.resize(n * ne);
  const double* x = &X[(size_t)k * n];
  knot_expansion_full(P, t, X, U, with_al, k, grad.data(), hess.data());
  errstate_jacobian(P.M, x, G0.data());
  for (int i = 0; i < ne; ++i) { double s = 0.0; for (int r = 0; r < n; ++r) s += G0[r * ne + i] * grad[r]; qx[i] = s; }
  for (int j = 0; j < m; ++j) qu[j] = grad[n + j];
  /* Qxx = G' Hxx G */
  for (int r = 0; r < n; ++r) for (int j = 0; j < ne; ++j) { double s = 0.0; for (int c = 0; c < n; ++c) s += hess[r * nz + c] * G0[c * ne + j]; T2[r * ne + j] = s; }
  for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) { double s = 0.0; for (int r = 0; r < n; ++r) s += G0[r * ne + i] * T2[r * ne + j]; Qxx[i * ne + j] = s; }
  if (P.M.id == TO_MODEL_QUADROTOR && P.M.rot() == TO_ROT_QUATERNION) { /* second-order term of the attitude map: -I3 (q' dJ/dq) (Rotations ∇differential) */
    double b1 = 0.0; for (int i = 0; i < 4; ++i) b1 += x[3 + i] * grad[3 + i];
    for (int i = 0; i < 3; ++i) Qxx[(3 + i) * ne + 3 + i] -= b1;
  } else if (P.M.id == TO_MODEL_QUADROTOR) { /* ... of a three-parameter attitude: ∇²differential(p, dJ/dp) */
    double H2[9]; att_differential2(P.M.rot(), x + 3, &grad[3], H2);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Qxx[(3 + i) * ne + 3 + j] += H2[3 * i + j];
  }
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Quu[i * m + j] = hess[(n + i) * nz + n + j];
  for (int i = 0; i < m; ++i) for (int j = 0; j < ne; ++j) { double s = 0.0; for (int c = 0; c < n; ++c) s += hess[(n + i) * nz + c] * G0[c * ne + j]; Qux[i * ne + j] = s; }
}

void expand(const Problem& P, Traj& t) {
  const int m = P.m, ne = P.ne, N = P.N;
  for (int k = 0; k < N - 1; ++k)
    dynamics_blocks(P, t.X.data(), t.U.data(), k, &t.A[(size_t)k * ne * ne], &t.Bm[(size_t)k * ne * m]);
  for (int k = 0; k < N; ++k)
    cost_blocks(P, t, t.X.data(), t.U.data(), true, k, &t.Qxx[(size_t)k * ne * ne], &t.Quu[(size_t)k * m * m], &t.Qux[(size_t)k * m * ne],
                &t.qx[(size_t)k * ne], &t.qu[(size_t)k * m]);
}

void reg_increase(const Problem& P, Traj& t) {
  const double f = P.opts.bp_reg_increase_factor;
  t.drho = std::fmax(t.drho * f, f);
  t.rho = std::fmax(t.rho * t.drho, P.opts.bp_reg_min);
}
void reg_decrease(const Problem& P, Traj& t) {
  const double f = P.opts.bp_reg_increase_factor;
  t.drho = std::fmin(t.drho / f, 1.0 / f);
  double r = t.rho * t.drho;
  t.rho = (r > P.opts.bp_reg_min) ? r : 0.0;
}

/* in-place lower Cholesky of a (m x m row-major); false if not positive definite */
bool cholesky(double* a, int m) {
  for (int j = 0; j < m; ++j) {
    double s = a[j * m + j];
    for (int k = 0; k < j; ++k) s -= a[j * m + k] * a[j * m + k];
    if (!(s > 0.0)) return false;
    double l = std::sqrt(s);
    a[j * m + j] = l;
    for (int i = j + 1; i < m; ++i) {
      double v = a[i * m + j];
      for (int k = 0; k < j; ++k) v -= a[i * m + k] * a[j * m + k];
      a[i * m + j] = v / l;
    }
  }
  return true;
}
void chol_solve(const double* L, int m, double* b) { /* solves (L L') x = b in place */
  for (int i = 0; i < m; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * m + k] * b[k]; b[i] = s / L[i * m + i]; }
  for (int i = m - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * b[k]; b[i] = s / L[i * m + i]; }
}

/* backward Riccati recursion.  Returns false when regularisation exceeded bp_reg_max. */
bool backward(const Problem& P, Traj& t) {
  const int m = P.m, ne = P.ne, N = P.N;
  static thread_local std::vector<double> S, s, SA, SB, Qxx, Quu, Qux, Qx, Qu, L, col, KtQuu, Snew, snew;
  S.resize(ne * ne); s.resize(ne); SA.resize(ne * ne); SB.resize(ne * m); Qxx.resize(ne * ne); Quu.resize(m * m); Qux.resize(m * ne);
  Qx.resize(ne); Qu.resize(m); L.resize(m * m); col.resize(m); KtQuu.resize(ne * m); Snew.resize(ne * ne); snew.resize(ne);
  while (true) {
    bool restart = false;
    t.dV[0] = t.dV[1] = 0.0;
    for (int i = 0; i < ne * ne; ++i) S[i] = t.Qxx[(size_t)(N - 1) * ne * ne + i];
    for (int i = 0; i < ne; ++i) s[i] = t.qx[(size_t)(N - 1) * ne + i];
    t.Sall.resize((size_t)N * ne * ne); t.sall.resize((size_t)N * ne);
    for (int i = 0; i < ne * ne; ++i) t.Sall[(size_t)(N - 1) * ne * ne + i] = S[i];
    for (int i = 0; i < ne; ++i) t.sall[(size_t)(N - 1) * ne + i] = s[i];
    for (int k = N - 2; k >= 0; --k) {
      const double* A = &t.A[(size_t)k * ne * ne]; const double* Bm = &t.Bm[(size_t)k * ne * m];
      const double* cQxx = &t.Qxx[(size_t)k * ne * ne]; const double* cQuu = &t.Quu[(size_t)k * m * m];
      const double* cQux = &t.Qux[(size_t)k * m * ne]; const double* cqx = &t.qx[(size_t)k * ne]; const double* cqu = &t.qu[(size_t)k * m];
      matmul(S.data(), A, SA.data(), ne, ne, ne);
      matmul(S.data(), Bm, SB.data(), ne, ne, m);
      for (int i = 0; i < ne; ++i) { double v = cqx[i]; for (int r = 0; r < ne; ++r) v += A[r * ne + i] * s[r]; Qx[i] = v; }
      for (int j = 0; j < m; ++j) { double v = cqu[j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + j] * s[r]; Qu[j] = v; }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) { double v = cQxx[i * ne + j]; for (int r = 0; r < ne; ++r) v += A[r * ne + i] * SA[r * ne + j]; Qxx[i * ne + j] = v; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double v = cQuu[i * m + j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + i] * SB[r * m + j]; Quu[i * m + j] = v; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < ne; ++j) { double v = cQux[i * ne + j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + i] * SA[r * ne + j]; Qux[i * ne + j] = v; }
      /* control regularisation */
      for (int i = 0; i < m * m; ++i) L[i] = Quu[i];
      for (int i = 0; i < m; ++i) L[i * m + i] += t.rho;
      if (!cholesky(L.data(), m)) {
        reg_increase(P, t);
        if (t.rho > P.opts.bp_reg_max) return false;
        restart = true; break;
      }
      double* K = &t.K[(size_t)k * m * ne]; double* d = &t.d[(size_t)k * m];
      for (int j = 0; j < ne; ++j) {
        for (int i = 0; i < m; ++i) col[i] = Qux[i * ne + j];
        chol_solve(L.data(), m, col.data());
        for (int i = 0; i < m; ++i) K[i * ne + j] = -col[i];
      }
      for (int i = 0; i < m; ++i) col[i] = Qu[i];
      chol_solve(L.data(), m, col.data());
      for (int i = 0; i < m; ++i) d[i] = -col[i];
      /* cost-to-go (un-regularised Quu) */
      for (int i = 0; i < ne; ++i) for (int j = 0; j < m; ++j) { double v = 0.0; for (int r = 0; r < m; ++r) v += K[r * ne + i] * Quu[r * m + j]; KtQuu[i * m + j] = v; }
      for (int i = 0; i < ne; ++i) {
        double v = Qx[i];
        for (int j = 0; j < m; ++j) v += KtQuu[i * m + j] * d[j];
        for (int j = 0; j < m; ++j) v += K[j * ne + i] * Qu[j];
        for (int j = 0; j < m; ++j) v += Qux[j * ne + i] * d[j];
        snew[i] = v;
      }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) {
        double v = Qxx[i * ne + j];
        for (int r = 0; r < m; ++r) v += KtQuu[i * m + r] * K[r * ne + j];
        for (int r = 0; r < m; ++r) v += K[r * ne + i] * Qux[r * ne + j];
        for (int r = 0; r < m; ++r) v += Qux[r * ne + i] * K[r * ne + j];
        Snew[i * ne + j] = v;
      }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) S[i * ne + j] = 0.5 * (Snew[i * ne + j] + Snew[j * ne + i]);
      for (int i = 0; i < ne; ++i) s[i] = snew[i];
      for (int i = 0; i < ne * ne; ++i) t.Sall[(size_t)k * ne * ne + i] = S[i];
      for (int i = 0; i < ne; ++i) t.sall[(size_t)k * ne + i] = s[i];
      double dv1 = 0.0, dv2 = 0.0;
      for (int i = 0; i < m; ++i) { dv1 += d[i] * Qu[i]; double v = 0.0; for (int j = 0; j < m; ++j) v += Quu[i * m + j] * d[j]; dv2 += d[i] * v; }
      t.dV[0] += dv1; t.dV[1] += 0.5 * dv2;
    }
    if (!restart) break;
  }
  reg_decrease(P, t);
  return true;
}

/* ---- Arithmetic model of the GPU's scan backward pass (csrc/k_scan.h), for tests/test_oracle_sensitivity.py only -------------
 * NOT a restatement of the reference: the reference (Altro) runs the sequential recursion above.  The GPU's k_expand_backward_scan
 * computes the cost-to-go at every second knot with an associative scan (Särkkä & García-Fernández 2023, Lemma 10) and walks the
 * two knots of each block with the sequential expressions; this function does the same on the CPU — same element definitions,
 * same Hillis-Steele order, same block size — so that the effect of the scan's rounding on whole solves can be measured
 * oracle-against-oracle (env ORACLE_RICCATI_SCAN=1 switches every backward pass of this library to it). */
namespace scanexp {
struct El { double A[16], b[4], C[16], eta[4], J[16]; };
/* solve M X = R for X (n x r), M n x n row-major general, partial pivoting-free LU (M = I + PSD*PSD: well conditioned) */
static void lu_solve(int n, double* M, double* R, int r) {
  for (int c = 0; c < n; ++c) {
    double piv = 1.0 / M[c * n + c];
    for (int i = c + 1; i < n; ++i) {
      double f = M[i * n + c] * piv;
      for (int j = c + 1; j < n; ++j) M[i * n + j] -= f * M[c * n + j];
      for (int j = 0; j < r; ++j) R[i * r + j] -= f * R[c * r + j];
    }
  }
  for (int c = n - 1; c >= 0; --c) {
    double piv = 1.0 / M[c * n + c];
    for (int j = 0; j < r; ++j) {
      double v = R[c * r + j];
      for (int i = c + 1; i < n; ++i) v -= M[c * n + i] * R[i * r + j];
      R[c * r + j] = v * piv;
    }
  }
}
static void combine(int n, const El& e1, const El& e2, El& o) {
  double M[16], R[4 * 9];
  const int r = 2 * n + 1;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = (i == j) ? 1.0 : 0.0; for (int t = 0; t < n; ++t) v += e1.C[i * n + t] * e2.J[t * n + j]; M[i * n + j] = v; }
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) R[i * r + j] = e1.A[i * n + j];
    double v = e1.b[i]; for (int t = 0; t < n; ++t) v -= e1.C[i * n + t] * e2.eta[t]; R[i * r + n] = v;
    for (int j = 0; j < n; ++j) R[i * r + n + 1 + j] = e1.C[i * n + j];
  }
  double Mc[16]; for (int i = 0; i < n * n; ++i) Mc[i] = M[i];
  lu_solve(n, Mc, R, r);
  double XC[16];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) { double v = 0; for (int t = 0; t < n; ++t) v += e2.A[i * n + t] * R[t * r + j]; o.A[i * n + j] = v; }
    { double v = e2.b[i]; for (int t = 0; t < n; ++t) v += e2.A[i * n + t] * R[t * r + n]; o.b[i] = v; }
    for (int j = 0; j < n; ++j) { double v = 0; for (int t = 0; t < n; ++t) v += e2.A[i * n + t] * R[t * r + n + 1 + j]; XC[i * n + j] = v; }
  }
  double Cn[16];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double v = e2.C[i * n + j]; for (int t = 0; t < n; ++t) v += XC[i * n + t] * e2.A[j * n + t]; Cn[i * n + j] = v; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) o.C[i * n + j] = 0.5 * (Cn[i * n + j] + Cn[j * n + i]);
  /* (I + J2 C1) = M^T */
  double Mt[16], Y[4 * 5];
  const int r2 = n + 1;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Mt[i * n + j] = M[j * n + i];
  for (int i = 0; i < n; ++i) {
    double v = e2.eta[i]; for (int t = 0; t < n; ++t) v += e2.J[i * n + t] * e1.b[t]; Y[i * r2 + 0] = v;
    for (int j = 0; j < n; ++j) { double w = 0; for (int t = 0; t < n; ++t) w += e2.J[i * n + t] * e1.A[t * n + j]; Y[i * r2 + 1 + j] = w; }
  }
  lu_solve(n, Mt, Y, r2);
  double Jn[16];
  for (int i = 0; i < n; ++i) {
    { double v = e1.eta[i]; for (int t = 0; t < n; ++t) v += e1.A[t * n + i] * Y[t * r2 + 0]; o.eta[i] = v; }
    for (int j = 0; j < n; ++j) { double v = e1.J[i * n + j]; for (int t = 0; t < n; ++t) v += e1.A[t * n + i] * Y[t * r2 + 1 + j]; Jn[i * n + j] = v; }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) o.J[i * n + j] = 0.5 * (Jn[i * n + j] + Jn[j * n + i]);
}
}  // namespace scanexp

bool backward(const Problem& P, Traj& t);
bool backward_scan(const Problem& P, Traj& t) {
  using namespace scanexp;
  const int m = P.m, ne = P.ne, N = P.N, n = ne;
  if (t.rho != 0.0 || ne > 4 || m > 2 || N > 126 || !P.cons.empty()) return backward(P, t);  /* the GPU kernel's scope */
  for (int k = 0; k < N; ++k) { /* ... diagonal cost blocks only */
    for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) if (i != j && t.Qxx[(size_t)k * ne * ne + i * ne + j] != 0.0) return backward(P, t);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) if (i != j && t.Quu[(size_t)k * m * m + i * m + j] != 0.0) return backward(P, t);
    for (int i = 0; i < m * ne; ++i) if (t.Qux[(size_t)k * m * ne + i] != 0.0) return backward(P, t);
  }
  std::vector<El> el(N);
  for (int k = 0; k < N - 1; ++k) {
    const double* A = &t.A[(size_t)k * ne * ne]; const double* Bm = &t.Bm[(size_t)k * ne * m];
    const double* Q = &t.Qxx[(size_t)k * ne * ne]; const double* R = &t.Quu[(size_t)k * m * m];
    const double* H = &t.Qux[(size_t)k * m * ne]; const double* q = &t.qx[(size_t)k * ne]; const double* r = &t.qu[(size_t)k * m];
    double L[4]; for (int i = 0; i < m * m; ++i) L[i] = R[i];
    if (!cholesky(L, m)) return backward(P, t);
    /* RiH = R^-1 H (m x ne), Rir = R^-1 r */
    double RiH[8], Rir[2], col[2];
    for (int j = 0; j < ne; ++j) { for (int i = 0; i < m; ++i) col[i] = H[i * ne + j]; chol_solve(L, m, col); for (int i = 0; i < m; ++i) RiH[i * ne + j] = col[i]; }
    for (int i = 0; i < m; ++i) col[i] = r[i]; chol_solve(L, m, col); for (int i = 0; i < m; ++i) Rir[i] = col[i];
    double RiBt[8]; /* R^-1 B^T (m x ne) */
    for (int j = 0; j < ne; ++j) { for (int i = 0; i < m; ++i) col[i] = Bm[j * m + i]; chol_solve(L, m, col); for (int i = 0; i < m; ++i) RiBt[i * ne + j] = col[i]; }
    El& e = el[k];
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        double a = A[i * n + j]; for (int c = 0; c < m; ++c) a -= Bm[i * m + c] * RiH[c * n + j]; e.A[i * n + j] = a;
        double qq = Q[i * n + j]; for (int c = 0; c < m; ++c) qq -= H[c * n + i] * RiH[c * n + j]; e.J[i * n + j] = qq;
        double cc = 0; for (int c = 0; c < m; ++c) cc += Bm[i * m + c] * RiBt[c * n + j]; e.C[i * n + j] = cc;
      }
      double bb = 0; for (int c = 0; c < m; ++c) bb -= Bm[i * m + c] * Rir[c]; e.b[i] = bb;
      double ee = q[i]; for (int c = 0; c < m; ++c) ee -= H[c * n + i] * Rir[c]; e.eta[i] = ee;
    }
  }
  { El& e = el[N - 1]; std::memset(&e, 0, sizeof(e));
    for (int i = 0; i < n * n; ++i) e.J[i] = t.Qxx[(size_t)(N - 1) * ne * ne + i];
    for (int i = 0; i < n; ++i) e.eta[i] = t.qx[(size_t)(N - 1) * ne + i]; }
  const int BL = 2, L = (N + BL - 1) / BL;
  std::vector<El> cur(L), nxt(L);
  for (int l = 0; l < L; ++l) {
    int k0 = l * BL;
    cur[l] = el[k0];
    for (int k = k0 + 1; k < std::min(N, k0 + BL); ++k) { El o; combine(n, cur[l], el[k], o); cur[l] = o; }
  }
  for (int d = 1; d < L; d *= 2) {
    for (int l = 0; l < L; ++l) { if (l + d < L) combine(n, cur[l], cur[l + d], nxt[l]); else nxt[l] = cur[l]; }
    std::swap(cur, nxt);
  }
  /* per block: sequential Riccati steps from the next block's suffix value */
  static thread_local std::vector<double> S, s, SA, SB, Qxx, Quu, Qux, Qx, Qu, Lc, col, KtQuu, Snew, snew;
  S.resize(ne * ne); s.resize(ne); SA.resize(ne * ne); SB.resize(ne * m); Qxx.resize(ne * ne); Quu.resize(m * m); Qux.resize(m * ne);
  Qx.resize(ne); Qu.resize(m); Lc.resize(m * m); col.resize(m); KtQuu.resize(ne * m); Snew.resize(ne * ne); snew.resize(ne);
  t.dV[0] = t.dV[1] = 0.0;
  std::vector<double> dv1s(N - 1), dv2s(N - 1);
  for (int l = L - 1; l >= 0; --l) {
    int k0 = l * BL, kend = std::min(N, k0 + BL);  /* knots k0 .. kend-1 */
    if (kend == N) { /* block holds the terminal knot */
      for (int i = 0; i < ne * ne; ++i) S[i] = t.Qxx[(size_t)(N - 1) * ne * ne + i];
      for (int i = 0; i < ne; ++i) s[i] = t.qx[(size_t)(N - 1) * ne + i];
      kend = N - 1;
    } else {
      for (int i = 0; i < ne * ne; ++i) S[i] = cur[l + 1].J[i];
      for (int i = 0; i < ne; ++i) s[i] = cur[l + 1].eta[i];
    }
    for (int k = kend - 1; k >= k0; --k) {
      const double* A = &t.A[(size_t)k * ne * ne]; const double* Bm = &t.Bm[(size_t)k * ne * m];
      const double* cQxx = &t.Qxx[(size_t)k * ne * ne]; const double* cQuu = &t.Quu[(size_t)k * m * m];
      const double* cQux = &t.Qux[(size_t)k * m * ne]; const double* cqx = &t.qx[(size_t)k * ne]; const double* cqu = &t.qu[(size_t)k * m];
      matmul(S.data(), A, SA.data(), ne, ne, ne);
      matmul(S.data(), Bm, SB.data(), ne, ne, m);
      for (int i = 0; i < ne; ++i) { double v = cqx[i]; for (int r = 0; r < ne; ++r) v += A[r * ne + i] * s[r]; Qx[i] = v; }
      for (int j = 0; j < m; ++j) { double v = cqu[j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + j] * s[r]; Qu[j] = v; }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) { double v = cQxx[i * ne + j]; for (int r = 0; r < ne; ++r) v += A[r * ne + i] * SA[r * ne + j]; Qxx[i * ne + j] = v; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double v = cQuu[i * m + j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + i] * SB[r * m + j]; Quu[i * m + j] = v; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < ne; ++j) { double v = cQux[i * ne + j]; for (int r = 0; r < ne; ++r) v += Bm[r * m + i] * SA[r * ne + j]; Qux[i * ne + j] = v; }
      for (int i = 0; i < m * m; ++i) Lc[i] = Quu[i];
      if (!cholesky(Lc.data(), m)) return backward(P, t);
      double* K = &t.K[(size_t)k * m * ne]; double* d = &t.d[(size_t)k * m];
      for (int j = 0; j < ne; ++j) { for (int i = 0; i < m; ++i) col[i] = Qux[i * ne + j]; chol_solve(Lc.data(), m, col.data()); for (int i = 0; i < m; ++i) K[i * ne + j] = -col[i]; }
      for (int i = 0; i < m; ++i) col[i] = Qu[i];
      chol_solve(Lc.data(), m, col.data());
      for (int i = 0; i < m; ++i) d[i] = -col[i];
      for (int i = 0; i < ne; ++i) for (int j = 0; j < m; ++j) { double v = 0.0; for (int r = 0; r < m; ++r) v += K[r * ne + i] * Quu[r * m + j]; KtQuu[i * m + j] = v; }
      for (int i = 0; i < ne; ++i) {
        double v = Qx[i];
        for (int j = 0; j < m; ++j) v += KtQuu[i * m + j] * d[j];
        for (int j = 0; j < m; ++j) v += K[j * ne + i] * Qu[j];
        for (int j = 0; j < m; ++j) v += Qux[j * ne + i] * d[j];
        snew[i] = v;
      }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) {
        double v = Qxx[i * ne + j];
        for (int r = 0; r < m; ++r) v += KtQuu[i * m + r] * K[r * ne + j];
        for (int r = 0; r < m; ++r) v += K[r * ne + i] * Qux[r * ne + j];
        for (int r = 0; r < m; ++r) v += Qux[r * ne + i] * K[r * ne + j];
        Snew[i * ne + j] = v;
      }
      for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) S[i * ne + j] = 0.5 * (Snew[i * ne + j] + Snew[j * ne + i]);
      for (int i = 0; i < ne; ++i) s[i] = snew[i];
      double dv1 = 0.0, dv2 = 0.0;
      for (int i = 0; i < m; ++i) { dv1 += d[i] * Qu[i]; double v = 0.0; for (int j = 0; j < m; ++j) v += Quu[i * m + j] * d[j]; dv2 += d[i] * v; }
      dv1s[k] = dv1; dv2s[k] = 0.5 * dv2;
    }
  }
  for (int k = N - 2; k >= 0; --k) { t.dV[0] += dv1s[k]; t.dV[1] += dv2s[k]; }
  reg_decrease(P, t);
  return true;
}

/* closed-loop rollout with step alpha into (Xb,Ub); false if a state/control limit or NaN is hit */
bool rollout_closed_loop(const Problem& P, Traj& t, double alpha) {
  const int n = P.n, m = P.m, ne = P.ne, N = P.N;
  double dx[MAXN];
  for (int i = 0; i < n; ++i) t.Xb[i] = t.x0[i];
  for (int k = 0; k < N - 1; ++k) {
    const double* xb = &t.Xb[(size_t)k * n];
    state_diff(P.M, xb, &t.X[(size_t)k * n], dx);
    const double* K = &t.K[(size_t)k * m * ne]; const double* d = &t.d[(size_t)k * m];
    double* ub = &t.Ub[(size_t)k * m];
    for (int j = 0; j < m; ++j) {
      double du = d[j] * alpha;
      for (int i = 0; i < ne; ++i) du += K[j * ne + i] * dx[i];
      ub[j] = t.U[(size_t)k * m + j] + du;
    }
    double* xn = &t.Xb[(size_t)(k + 1) * n];
    knot_step(P.M, P.integrator, k, xb, ub, P.dt[k], xn);
    double mx = 0.0, mu = 0.0;
    for (int i = 0; i < n; ++i) { double a = std::fabs(xn[i]); if (!(a <= mx)) mx = a; }
    for (int j = 0; j < m; ++j) { double a = std::fabs(ub[j]); if (!(a <= mu)) mu = a; }
    if (!(mx <= P.opts.max_state_value) || !(mu <= P.opts.max_control_value)) return false;
  }
  return true;
}

/* forward pass with backtracking line search.  On success (Xb,Ub) are copied into (X,U). Returns new J. */
double forward(const Problem& P, Traj& t, double J_prev) {
  double alpha = 1.0;
  t.ls_index = -1; t.ls_failed = false; t.zero_step = false;
  /* Stationary point: the backward pass predicts no improvement at all (an exactly solved LQ problem gives ~1e-30).
   * Every ratio z = dJ/expected would then be rounding noise; accept the zero step instead (dJ = 0 => converged).
   * Oracle-defined; keeps iteration counts deterministic and skips Altro's 10 wasted NO_PROGRESS iterations. */
  if (-(t.dV[0] + t.dV[1]) <= 1e-12 * (1.0 + std::fabs(J_prev))) { t.ls_index = 0; t.zero_step = true; return J_prev; }
  for (int it = 0; it < P.opts.iterations_linesearch; ++it) {
    bool ok = rollout_closed_loop(P, t, alpha);
    if (ok) {
      double J = total_cost(P, t, t.Xb.data(), t.Ub.data(), true);
      double expected = -alpha * (t.dV[0] + alpha * t.dV[1]);
      double z = (expected > 0.0) ? (J_prev - J) / expected : -1.0;
      if (z >= P.opts.line_search_lower_bound && z <= P.opts.line_search_upper_bound) {
        t.X = t.Xb; t.U = t.Ub; t.ls_index = it;
        return J;
      }
    }
    alpha *= P.opts.line_search_decrease_factor;
  }
  /* failure: keep the nominal trajectory, regularise harder */
  t.ls_failed = true;
  reg_increase(P, t);
  t.rho += P.opts.bp_reg_fp;
  return J_prev;
}

double gradient_metric(const Problem& P, const Traj& t) {
  const int m = P.m; double s = 0.0;
  for (int k = 0; k < P.N - 1; ++k) {
    double mx = 0.0;
    for (int j = 0; j < m; ++j) { double v = std::fabs(t.d[(size_t)k * m + j]) / (std::fabs(t.U[(size_t)k * m + j]) + 1.0); if (v > mx) mx = v; }
    s += mx;
  }
  return s / (P.N - 1);
}

/* one iLQR iteration; returns true when the inner solve is finished (status set) */
bool ilqr_step(const Problem& P, Traj& t, double cost_tol, int max_iters, double& J_prev) {
  expand(P, t);
  if (!(std::getenv("ORACLE_RICCATI_SCAN") ? backward_scan(P, t) : backward(P, t))) { t.status = TO_REGULARIZATION_MAX; return true; }
  double J = forward(P, t, J_prev);
  t.dJ = J_prev - J;
  /* a zero step (stationary point) makes no progress either: with a gradient tolerance it cannot meet, the solve ends
   * NO_PROGRESS after dJ_counter_limit repeats instead of spinning to MAX_ITERATIONS */
  if (t.ls_failed || t.zero_step) t.dJ_zero_counter++; else t.dJ_zero_counter = 0;
  t.grad = gradient_metric(P, t);
  J_prev = J; t.J = J;
  t.iterations++;
  if (t.rho > P.opts.bp_reg_max) { t.status = TO_REGULARIZATION_MAX; return true; }
  if (t.dJ >= 0.0 && t.dJ < cost_tol && t.grad < P.opts.gradient_tolerance && !t.ls_failed) { t.status = TO_SOLVE_SUCCEEDED; return true; }
  if (max_iters <= 0) { t.status = TO_MAX_ITERATIONS; return true; }
  if (t.dJ_zero_counter > P.opts.dJ_counter_limit) { t.status = TO_NO_PROGRESS; return true; }
  if (!(J <= P.opts.max_cost_value)) { t.status = TO_MAXIMUM_COST; return true; }
  return false;
}

void ilqr_solve(const Problem& P, Traj& t, double cost_tol, int max_iters) {
  t.rho = P.opts.bp_reg_initial; t.drho = 0.0; t.dJ_zero_counter = 0; t.status = TO_UNSOLVED;
  rollout(P, t);
  double J_prev = total_cost(P, t, t.X.data(), t.U.data(), true);
  t.J = J_prev;
  /* Altro's rollout! checks every knot it simulates — the state it arrives at first, then the control that took it there — and
   * reports STATE_LIMIT / CONTROL_LIMIT.  Inside a line search such a candidate is simply rejected (rollout_closed_loop); the
   * INITIAL rollout of a solve has nothing to fall back on: the solve ends there with that status, no iteration performed. */
  for (int k = 0; k < P.N - 1; ++k) {
    double mx = 0.0, mu = 0.0;
    for (int i = 0; i < P.n; ++i) { double a = std::fabs(t.X[(size_t)(k + 1) * P.n + i]); if (!(a <= mx)) mx = a; }
    for (int j = 0; j < P.m; ++j) { double a = std::fabs(t.U[(size_t)k * P.m + j]); if (!(a <= mu)) mu = a; }
    if (!(mx <= P.opts.max_state_value)) { t.status = TO_STATE_LIMIT; return; }
    if (!(mu <= P.opts.max_control_value)) { t.status = TO_CONTROL_LIMIT; return; }
  }
  if (max_iters <= 0) { t.status = TO_MAX_ITERATIONS; return; }
  int it = 0;
  while (true) {
    ++it;
    if (ilqr_step(P, t, cost_tol, max_iters - it, J_prev)) break;
  }
}

void al_solve(const Problem& P, Traj& t, double constraint_tolerance) {
  std::fill(t.lambda.begin(), t.lambda.end(), 0.0);
  std::fill(t.mu.begin(), t.mu.end(), P.opts.penalty_initial);
  t.iterations = 0; t.iterations_outer = 0; t.iterations_pn = 0;
  for (int outer = 1; outer <= P.opts.iterations_outer; ++outer) {
    int budget = std::min(P.opts.iterations, P.opts.iterations_total - t.iterations);
    ilqr_solve(P, t, P.opts.cost_tolerance_intermediate, budget);
    t.iterations_outer = outer;
    t.c_max = max_violation(P, t);
    if (t.status != TO_SOLVE_SUCCEEDED && t.status != TO_MAX_ITERATIONS && t.status != TO_NO_PROGRESS) break;
    if (t.c_max < constraint_tolerance) { t.status = TO_SOLVE_SUCCEEDED; break; }
    if (t.iterations >= P.opts.iterations_total) { t.status = TO_MAX_ITERATIONS; break; }
    if (outer == P.opts.iterations_outer) { t.status = TO_MAX_ITERATIONS_OUTER; break; }
    dual_update(P, t);
  }
}

#include "oracle_pn.h"

/* Altro solve!(::ALTROSolver): the AL stage runs to projected_newton_tolerance, the polish takes the trajectories it left
 * SOLVE_SUCCEEDED above constraint_tolerance */
void altro_solve(const Problem& P, Traj& t) {
  const bool pn = P.opts.projected_newton && !P.cons.empty();
  al_solve(P, t, pn ? P.opts.projected_newton_tolerance : P.opts.constraint_tolerance);
  if (pn && t.status == TO_SOLVE_SUCCEEDED && t.c_max > P.opts.constraint_tolerance) pn_solve(P, t);
}

template <class F>
void for_batch(oracle_handle* h, F f) {
  const int B = h->P.B;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(h->threads)
#endif
  for (int b = 0; b < B; ++b) f(h->T[b], b);
}

void fill_stats(oracle_handle* h, to_solve_stats* st, double ms, bool with_defect = false) {
  if (!st) return;
  const Problem& P = h->P;
  int64_t tot = 0;
  for (int b = 0; b < P.B; ++b) {
    Traj& t = h->T[b];
    tot += t.iterations;
    if (st->iterations) st->iterations[b] = t.iterations;
    if (st->iterations_outer) st->iterations_outer[b] = t.iterations_outer;
    if (st->status) st->status[b] = t.status;
    if (st->cost) st->cost[b] = total_cost(P, t, t.X.data(), t.U.data(), false);
    if (st->dJ) st->dJ[b] = t.dJ;
    if (st->gradient) st->gradient[b] = t.grad;
    if (st->c_max) {
      st->c_max[b] = P.cons.empty() ? 0.0 : max_violation(P, t);
      if (with_defect) { const double df = dynamics_defect(P, t); if (df > st->c_max[b] || std::isnan(df)) st->c_max[b] = df; }
    }
    if (st->iterations_pn) st->iterations_pn[b] = t.iterations_pn;
    if (st->penalty_max) { double mx = 0.0; for (double v : t.mu) mx = std::fmax(mx, v); st->penalty_max[b] = mx; }
  }
  st->total_iterations = tot; st->batch_steps = 0; st->solve_ms = ms;
}

}  // namespace

#define CHECK_H(h) do { if (!(h)) return fail(TO_ERR_NULL, "null handle"); } while (0)
#define CHECK_P(p) do { if (!(p)) return fail(TO_ERR_NULL, "null pointer"); } while (0)

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }
int oracle_default_options(to_solver_opts* o) { CHECK_P(o); default_opts(o); return TO_OK; }
int oracle_abi_version(void) { return TO_ABI_VERSION; }
int oracle_device_count(int* c) { if (c) *c = 0; return TO_OK; }
int oracle_sync(oracle_handle*) { return TO_OK; }
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int oracle_create(const to_problem_desc* desc, const to_solver_opts* opts, int /*device, ignored*/, oracle_handle** out) {
  CHECK_P(out);
  oracle_handle* h = new oracle_handle();
  int r = build_problem(desc, opts, &h->P);
  if (r) { delete h; return r; }
  h->P.M.steps = h->P.steps.empty() ? nullptr : h->P.steps.data();  /* built in place: the table lives as long as the handle */
  h->T.resize(h->P.B);
  for (auto& t : h->T) alloc_traj(h->P, t);
  h->threads = 1;
  *out = h;
  return TO_OK;
}
int oracle_destroy(oracle_handle* h) { delete h; return TO_OK; }
int oracle_set_threads(oracle_handle* h, int threads) { CHECK_H(h); h->threads = threads < 1 ? 1 : threads; return TO_OK; }
int oracle_set_options(oracle_handle* h, const to_solver_opts* o) { CHECK_H(h); CHECK_P(o); { int r = validate_opts(*o); if (r) return r; } h->P.opts = *o; return TO_OK; }
int oracle_get_options(const oracle_handle* h, to_solver_opts* o) { CHECK_H(h); CHECK_P(o); *o = h->P.opts; return TO_OK; }
int oracle_dims(const oracle_handle* h, int32_t* n, int32_t* m, int32_t* ne, int32_t* N, int32_t* B) {
  CHECK_H(h);
  if (n) *n = h->P.n; if (m) *m = h->P.m; if (ne) *ne = h->P.ne; if (N) *N = h->P.N; if (B) *B = h->P.B;
  return TO_OK;
}
int oracle_num_constraints(const oracle_handle* h, int32_t* p) { /* src/constraint_list.jl:198-206 */
  CHECK_H(h); CHECK_P(p);
  for (int k = 0; k < h->P.N; ++k) p[k] = 0;
  for (const ConInfo& ci : h->P.cons) for (int k = ci.k1; k <= ci.k2; ++k) p[k] += ci.p;
  return TO_OK;
}

int oracle_set_initial_state(oracle_handle* h, const double* x0) {
  CHECK_H(h); CHECK_P(x0);
  for (int b = 0; b < h->P.B; ++b) for (int i = 0; i < h->P.n; ++i) h->T[b].x0[i] = x0[(size_t)b * h->P.n + i];
  return TO_OK;
}
int oracle_get_initial_state(oracle_handle* h, double* x0) {
  CHECK_H(h); CHECK_P(x0);
  for (int b = 0; b < h->P.B; ++b) for (int i = 0; i < h->P.n; ++i) x0[(size_t)b * h->P.n + i] = h->T[b].x0[i];
  return TO_OK;
}
int oracle_set_controls(oracle_handle* h, const double* U) {
  CHECK_H(h); CHECK_P(U);
  size_t sz = (size_t)(h->P.N - 1) * h->P.m;
  for (int b = 0; b < h->P.B; ++b) std::memcpy(h->T[b].U.data(), U + b * sz, sz * sizeof(double));
  return TO_OK;
}
int oracle_set_controls_uniform(oracle_handle* h, const double* u) {
  CHECK_H(h); CHECK_P(u);
  for (int b = 0; b < h->P.B; ++b) for (int k = 0; k < h->P.N - 1; ++k) for (int j = 0; j < h->P.m; ++j) h->T[b].U[(size_t)k * h->P.m + j] = u[j];
  return TO_OK;
}
int oracle_set_states(oracle_handle* h, const double* X) {
  CHECK_H(h); CHECK_P(X);
  size_t sz = (size_t)h->P.N * h->P.n;
  for (int b = 0; b < h->P.B; ++b) std::memcpy(h->T[b].X.data(), X + b * sz, sz * sizeof(double));
  return TO_OK;
}
int oracle_get_states(oracle_handle* h, double* X) {
  CHECK_H(h); CHECK_P(X);
  size_t sz = (size_t)h->P.N * h->P.n;
  for (int b = 0; b < h->P.B; ++b) std::memcpy(X + b * sz, h->T[b].X.data(), sz * sizeof(double));
  return TO_OK;
}
int oracle_get_controls(oracle_handle* h, double* U) {
  CHECK_H(h); CHECK_P(U);
  size_t sz = (size_t)(h->P.N - 1) * h->P.m;
  for (int b = 0; b < h->P.B; ++b) std::memcpy(U + b * sz, h->T[b].U.data(), sz * sizeof(double));
  return TO_OK;
}
int oracle_set_cost(oracle_handle* h, int32_t id, const to_cost_desc* c) {
  CHECK_H(h); CHECK_P(c);
  if (id < 0 || id >= (int)h->P.costs.size()) return fail(TO_ERR_ARGUMENT, "cost id out of range");
  int r = validate_cost(h->P, *c); if (r) return r;
  h->P.costs[id] = *c;
  for (Traj& t : h->T) if (!t.gl.empty()) std::fill(t.gl.begin() + (size_t)id * (h->P.n + h->P.m), t.gl.begin() + (size_t)(id + 1) * (h->P.n + h->P.m), 0.0);  /* its per-trajectory terms start over */
  return TO_OK;
}
/* per-trajectory q (n, B) / r (m, B) of cost `id` (either may be NULL): set_LQR_goal!(cost, xf_b, uf_b) for every trajectory at once */
int oracle_set_cost_linear_batch(oracle_handle* h, int32_t id, const double* q, const double* r) {
  CHECK_H(h);
  Problem& P = h->P;
  if (id < 0 || id >= (int)P.costs.size()) return fail(TO_ERR_ARGUMENT, "cost id out of range");
  if (P.costs[id].kind == TO_COST_ERROR_QUADRATIC) return fail(TO_ERR_UNSUPPORTED, "per-trajectory linear terms: not for ErrorQuadratic (its q slot carries x_ref)");
  const int n = P.n, m = P.m, nz = n + m;
  for (int b = 0; b < P.B; ++b) {
    Traj& t = h->T[b];
    if (t.gl.empty()) t.gl.assign((size_t)P.costs.size() * nz, 0.0);
    double* g = &t.gl[(size_t)id * nz];
    if (q) for (int i = 0; i < n; ++i) g[i] = q[i + (size_t)n * b] - P.costs[id].q[i];
    if (r) for (int j = 0; j < m; ++j) g[n + j] = r[j + (size_t)m * b] - P.costs[id].r[j];
  }
  return TO_OK;
}
int oracle_clear_cost_linear_batch(oracle_handle* h) {
  CHECK_H(h);
  for (Traj& t : h->T) t.gl.clear();
  return TO_OK;
}
int oracle_set_constraint(oracle_handle* h, int32_t id, const to_constraint_desc* c) {
  CHECK_H(h); CHECK_P(c);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  ConInfo ci; int r = validate_constraint(h->P, *c, &ci); if (r) return r;
  const ConInfo& old = h->P.cons[id];
  if (ci.p != old.p || ci.k1 != old.k1 || ci.k2 != old.k2) return fail(TO_ERR_DIMENSION_MISMATCH, "replacement constraint must keep p and the knot range");
  ci.dual_off = old.dual_off; h->P.cons[id] = ci;
  for (Traj& t : h->T) if ((size_t)id < t.cpar.size()) t.cpar[id].clear();  /* its per-trajectory parameters start over */
  return TO_OK;
}
/* One parameter set per TRAJECTORY for constraint id (to_set_constraint_params_batch): params[p, B] column-major.  GOAL: xf[inds]; LINEAR: b. */
int oracle_set_constraint_params_batch(oracle_handle* h, int32_t id, const double* params) {
  CHECK_H(h); CHECK_P(params);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id];
  if (ci.d.kind != TO_CON_GOAL && ci.d.kind != TO_CON_LINEAR)
    return fail(TO_ERR_UNSUPPORTED, "per-trajectory constraint parameters: GoalConstraint (its target) and LinearConstraint (its b) only");
  const int np = ci.p;
  for (int b = 0; b < h->P.B; ++b) {
    Traj& t = h->T[b];
    if (t.cpar.size() < h->P.cons.size()) t.cpar.resize(h->P.cons.size());
    t.cpar[id].assign(params + (size_t)np * b, params + (size_t)np * (b + 1));
  }
  return TO_OK;
}
int oracle_clear_constraint_params_batch(oracle_handle* h) {
  CHECK_H(h);
  for (Traj& t : h->T) t.cpar.clear();
  return TO_OK;
}

int oracle_rollout(oracle_handle* h) { CHECK_H(h); for_batch(h, [&](Traj& t, int) { rollout(h->P, t); }); return TO_OK; }
int oracle_cost(oracle_handle* h, double* J) {
  CHECK_H(h); CHECK_P(J);
  for_batch(h, [&](Traj& t, int b) { J[b] = total_cost(h->P, t, t.X.data(), t.U.data(), false); });
  return TO_OK;
}
int oracle_al_cost(oracle_handle* h, double* J) {
  CHECK_H(h); CHECK_P(J);
  for_batch(h, [&](Traj& t, int b) { J[b] = total_cost(h->P, t, t.X.data(), t.U.data(), true); });
  return TO_OK;
}
int oracle_stage_costs(oracle_handle* h, double* Jk) {
  CHECK_H(h); CHECK_P(Jk);
  for_batch(h, [&](Traj& t, int b) { for (int k = 0; k < h->P.N; ++k) Jk[(size_t)b * h->P.N + k] = objective_knot(h->P, t.X.data(), t.U.data(), k, t.gl.empty() ? nullptr : t.gl.data()); });
  return TO_OK;
}
int oracle_expand(oracle_handle* h) { CHECK_H(h); for_batch(h, [&](Traj& t, int) { expand(h->P, t); }); return TO_OK; }
int oracle_backward(oracle_handle* h) {
  CHECK_H(h);
  for_batch(h, [&](Traj& t, int) { if (!(std::getenv("ORACLE_RICCATI_SCAN") ? backward_scan(h->P, t) : backward(h->P, t))) t.status = TO_REGULARIZATION_MAX; });
  return TO_OK;
}
int oracle_forward(oracle_handle* h, int32_t* ls_index, double* J_new) {
  CHECK_H(h);
  for_batch(h, [&](Traj& t, int b) {
    double Jp = total_cost(h->P, t, t.X.data(), t.U.data(), true);
    double J = forward(h->P, t, Jp);
    if (ls_index) ls_index[b] = t.ls_index;
    if (J_new) J_new[b] = J;
  });
  return TO_OK;
}
int oracle_ilqr_solve(oracle_handle* h, to_solve_stats* st) {
  CHECK_H(h);
  auto t0 = std::chrono::steady_clock::now();
  for_batch(h, [&](Traj& t, int) { t.iterations = 0; t.iterations_outer = 0; ilqr_solve(h->P, t, h->P.opts.cost_tolerance, h->P.opts.iterations); });
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fill_stats(h, st, ms);
  return TO_OK;
}
int oracle_al_solve(oracle_handle* h, to_solve_stats* st) {
  CHECK_H(h);
  auto t0 = std::chrono::steady_clock::now();
  for_batch(h, [&](Traj& t, int) { al_solve(h->P, t, h->P.opts.constraint_tolerance); });
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fill_stats(h, st, ms);
  return TO_OK;
}
int oracle_pn_solve(oracle_handle* h, to_solve_stats* st) {
  CHECK_H(h);
  auto t0 = std::chrono::steady_clock::now();
  for_batch(h, [&](Traj& t, int) { t.iterations = 0; t.iterations_outer = 0; pn_solve(h->P, t); });
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fill_stats(h, st, ms, true);
  return TO_OK;
}
int oracle_altro_solve(oracle_handle* h, to_solve_stats* st) {
  CHECK_H(h);
  auto t0 = std::chrono::steady_clock::now();
  for_batch(h, [&](Traj& t, int) { altro_solve(h->P, t); });
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fill_stats(h, st, ms, true);
  return TO_OK;
}
int oracle_dynamics_defect(oracle_handle* h, double* out) {
  CHECK_H(h); CHECK_P(out);
  for_batch(h, [&](Traj& t, int b) { out[b] = dynamics_defect(h->P, t); });
  return TO_OK;
}

/* getters: convert internal row-major blocks to the column-major host layouts of the header */
int oracle_get_dynamics_jacobians(oracle_handle* h, double* A, double* Bm) {
  CHECK_H(h);
  const int ne = h->P.ne, m = h->P.m, N = h->P.N;
  for (int b = 0; b < h->P.B; ++b) for (int k = 0; k < N - 1; ++k) {
    const Traj& t = h->T[b];
    if (A) for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) A[i + ne * (j + (size_t)ne * (k + (size_t)(N - 1) * b))] = t.A[(size_t)k * ne * ne + i * ne + j];
    if (Bm) for (int i = 0; i < ne; ++i) for (int j = 0; j < m; ++j) Bm[i + ne * (j + (size_t)m * (k + (size_t)(N - 1) * b))] = t.Bm[(size_t)k * ne * m + i * m + j];
  }
  return TO_OK;
}
int oracle_get_cost_expansion(oracle_handle* h, double* Qxx, double* Quu, double* Qux, double* qx, double* qu) {
  CHECK_H(h);
  const int ne = h->P.ne, m = h->P.m, N = h->P.N;
  for (int b = 0; b < h->P.B; ++b) for (int k = 0; k < N; ++k) {
    const Traj& t = h->T[b]; size_t kb = k + (size_t)N * b;
    if (Qxx) for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) Qxx[i + ne * (j + ne * kb)] = t.Qxx[(size_t)k * ne * ne + i * ne + j];
    if (Quu) for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Quu[i + m * (j + m * kb)] = t.Quu[(size_t)k * m * m + i * m + j];
    if (Qux) for (int i = 0; i < m; ++i) for (int j = 0; j < ne; ++j) Qux[i + m * (j + ne * kb)] = t.Qux[(size_t)k * m * ne + i * ne + j];
    if (qx) for (int i = 0; i < ne; ++i) qx[i + ne * kb] = t.qx[(size_t)k * ne + i];
    if (qu) for (int i = 0; i < m; ++i) qu[i + m * kb] = t.qu[(size_t)k * m + i];
  }
  return TO_OK;
}
int oracle_get_gains(oracle_handle* h, double* K, double* d, double* dV, double* rho) {
  CHECK_H(h);
  const int ne = h->P.ne, m = h->P.m, N = h->P.N;
  for (int b = 0; b < h->P.B; ++b) {
    const Traj& t = h->T[b];
    for (int k = 0; k < N - 1; ++k) {
      size_t kb = k + (size_t)(N - 1) * b;
      if (K) for (int i = 0; i < m; ++i) for (int j = 0; j < ne; ++j) K[i + m * (j + ne * kb)] = t.K[(size_t)k * m * ne + i * ne + j];
      if (d) for (int i = 0; i < m; ++i) d[i + m * kb] = t.d[(size_t)k * m + i];
    }
    if (dV) { dV[2 * b] = t.dV[0]; dV[2 * b + 1] = t.dV[1]; }
    if (rho) rho[b] = t.rho;
  }
  return TO_OK;
}
int oracle_get_cost_to_go(oracle_handle* h, double* S, double* s) {
  CHECK_H(h);
  const int ne = h->P.ne, N = h->P.N;
  for (int b = 0; b < h->P.B; ++b) {
    const Traj& t = h->T[b];
    if (t.Sall.size() != (size_t)N * ne * ne) return fail(TO_ERR_ARGUMENT, "no backward pass has run on this handle");
    for (int k = 0; k < N; ++k) {
      size_t kb = k + (size_t)N * b;
      if (S) for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) S[i + ne * (j + ne * kb)] = t.Sall[(size_t)k * ne * ne + i * ne + j];
      if (s) for (int i = 0; i < ne; ++i) s[i + ne * kb] = t.sall[(size_t)k * ne + i];
    }
  }
  return TO_OK;
}
/* Altro's infeasible_controls: slack controls w_k = x_{k+1} - f_d(x_k, u_k) of an InfeasibleModel from the current states */
int oracle_infeasible_controls(oracle_handle* h) {
  CHECK_H(h);
  const Problem& P = h->P;
  if (P.M.id != TO_MODEL_INFEASIBLE) return fail(TO_ERR_ARGUMENT, "to_infeasible_controls: the handle's model is not TO_MODEL_INFEASIBLE");
  const int n = P.n, m = P.m, N = P.N, m0 = m - n;
  for (int b = 0; b < P.B; ++b) {
    Traj& t = h->T[b];
    for (int k = 0; k < N - 1; ++k) {
      double u[MAXM], xn[MAXN];
      for (int j = 0; j < m; ++j) u[j] = j < m0 ? t.U[(size_t)k * m + j] : 0.0;
      knot_step(P.M, P.integrator, k, &t.X[(size_t)k * n], u, P.dt[k], xn);
      for (int i = 0; i < n; ++i) t.U[(size_t)k * m + m0 + i] = t.X[(size_t)(k + 1) * n + i] - xn[i];
    }
  }
  return TO_OK;
}
int oracle_cost_expansion(oracle_handle* h, double* grad, double* hess) {
  CHECK_H(h);
  const int n = h->P.n, m = h->P.m, nz = n + m, N = h->P.N;
  std::vector<double> g(nz), H(nz * nz);
  for (int b = 0; b < h->P.B; ++b) for (int k = 0; k < N; ++k) {
    knot_expansion_full(h->P, h->T[b], h->T[b].X.data(), h->T[b].U.data(), /*with_al (objective only)*/ false, k, g.data(), H.data());
    size_t kb = k + (size_t)N * b;
    if (grad) for (int i = 0; i < nz; ++i) grad[i + nz * kb] = g[i];
    if (hess) for (int i = 0; i < nz; ++i) for (int j = 0; j < nz; ++j) hess[i + nz * (j + nz * kb)] = H[i * nz + j];
  }
  return TO_OK;
}
int oracle_discrete_jacobian(oracle_handle* h, double* F) {
  CHECK_H(h); CHECK_P(F);
  const int n = h->P.n, m = h->P.m, nz = n + m, N = h->P.N;
  std::vector<double> A(n * n), Bf(n * m);
  for (int b = 0; b < h->P.B; ++b) for (int k = 0; k < N - 1; ++k) {
    const Traj& t = h->T[b];
    knot_step_jacobian(h->P.M, h->P.integrator, k, &t.X[(size_t)k * n], &t.U[(size_t)k * m], h->P.dt[k], A.data(), Bf.data());
    size_t kb = k + (size_t)(N - 1) * b;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) F[i + n * (j + nz * kb)] = A[i * n + j];
      for (int j = 0; j < m; ++j) F[i + n * (n + j + nz * kb)] = Bf[i * m + j];
    }
  }
  return TO_OK;
}

int oracle_knot_dims(const oracle_handle* h, int32_t* nx, int32_t* nu) {
  CHECK_H(h);
  for (int k = 0; k < h->P.N; ++k) { int a, b; knot_dims(h->P.M, k, &a, &b, h->P.N); nx[k] = a; nu[k] = b; }
  return TO_OK;
}
int oracle_constraint_info(const oracle_handle* h, int32_t id, int32_t* p, int32_t* width, int32_t* nk, int32_t* sense) {
  CHECK_H(h);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id];
  if (p) *p = ci.p; if (width) *width = ci.width; if (nk) *nk = ci.k2 - ci.k1 + 1; if (sense) *sense = ci.d.sense;
  return TO_OK;
}
int oracle_evaluate_constraints(oracle_handle* h, int32_t id, double* vals) {
  CHECK_H(h); CHECK_P(vals);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id]; const int nk = ci.k2 - ci.k1 + 1;
  double z[MAXZ], c[TO_MAX_P];
  for (int b = 0; b < h->P.B; ++b) for (int k = ci.k1; k <= ci.k2; ++k) {
    knot_z(h->P, h->T[b].X.data(), h->T[b].U.data(), k, z);
    EffDesc ed;
    constraint_evaluate(ed.get(h->T[b], id, ci), h->P.n, h->P.m, z, c, nullptr);
    for (int r = 0; r < ci.p; ++r) vals[r + ci.p * ((k - ci.k1) + (size_t)nk * b)] = c[r];
  }
  return TO_OK;
}
int oracle_constraint_jacobians(oracle_handle* h, int32_t id, double* jac) {
  CHECK_H(h); CHECK_P(jac);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id]; const int nk = ci.k2 - ci.k1 + 1, nz = h->P.n + h->P.m, w = ci.width;
  double z[MAXZ], c[TO_MAX_P], J[TO_MAX_P * MAXZ];
  for (int b = 0; b < h->P.B; ++b) for (int k = ci.k1; k <= ci.k2; ++k) {
    knot_z(h->P, h->T[b].X.data(), h->T[b].U.data(), k, z);
    EffDesc ed;
    constraint_evaluate(ed.get(h->T[b], id, ci), h->P.n, h->P.m, z, c, J);
    size_t kb = (k - ci.k1) + (size_t)nk * b;
    for (int r = 0; r < ci.p; ++r) for (int j = 0; j < w; ++j) jac[r + ci.p * (j + (size_t)w * kb)] = J[r * nz + j];
  }
  return TO_OK;
}
int oracle_constraint_hessians(oracle_handle* h, int32_t id, const double* lambda, double* H) {
  CHECK_H(h); CHECK_P(lambda); CHECK_P(H);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id]; const int nk = ci.k2 - ci.k1 + 1, w = ci.width;
  double z[MAXZ];
  for (int b = 0; b < h->P.B; ++b) for (int k = ci.k1; k <= ci.k2; ++k) {
    knot_z(h->P, h->T[b].X.data(), h->T[b].U.data(), k, z);
    const size_t kb = (k - ci.k1) + (size_t)nk * b;
    EffDesc ed;
    constraint_hessian_add(ed.get(h->T[b], id, ci), h->P.n, h->P.m, z, lambda + (size_t)ci.p * kb, H + (size_t)w * w * kb, w);
  }
  return TO_OK;
}
int oracle_max_violation(oracle_handle* h, double* c_max) {
  CHECK_H(h); CHECK_P(c_max);
  for_batch(h, [&](Traj& t, int b) { c_max[b] = max_violation(h->P, t); });
  return TO_OK;
}
int oracle_get_duals(oracle_handle* h, int32_t id, double* lambda, double* mu) {
  CHECK_H(h);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id]; size_t cnt = (size_t)ci.p * (ci.k2 - ci.k1 + 1);
  for (int b = 0; b < h->P.B; ++b) {
    if (lambda) std::memcpy(lambda + b * cnt, &h->T[b].lambda[ci.dual_off], cnt * sizeof(double));
    if (mu) mu[b] = h->T[b].mu[id];
  }
  return TO_OK;
}
int oracle_set_duals(oracle_handle* h, int32_t id, const double* lambda, const double* mu) {
  CHECK_H(h);
  if (id < 0 || id >= (int)h->P.cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const ConInfo& ci = h->P.cons[id]; size_t cnt = (size_t)ci.p * (ci.k2 - ci.k1 + 1);
  for (int b = 0; b < h->P.B; ++b) {
    if (lambda) std::memcpy(&h->T[b].lambda[ci.dual_off], lambda + b * cnt, cnt * sizeof(double));
    if (mu) h->T[b].mu[id] = mu[b];
  }
  return TO_OK;
}
int oracle_reset_duals(oracle_handle* h) {
  CHECK_H(h);
  for (auto& t : h->T) { std::fill(t.lambda.begin(), t.lambda.end(), 0.0); std::fill(t.mu.begin(), t.mu.end(), h->P.opts.penalty_initial); }
  return TO_OK;
}
int oracle_dual_update(oracle_handle* h) { CHECK_H(h); for_batch(h, [&](Traj& t, int) { dual_update(h->P, t); }); return TO_OK; }

/* cones, batched stateless; column-major outputs */
int oracle_cone_projection(int /*device*/, int32_t cone, int32_t dim, int64_t count, const double* x, double* px, int32_t* status) {
  CHECK_P(x); CHECK_P(px);
  if (dim < 1 || dim > TO_MAX_P) return fail(TO_ERR_ARGUMENT, "cone dimension out of range");
  for (int64_t i = 0; i < count; ++i) {
    int s = cone_projection(cone, x + i * dim, px + i * dim, dim);
    if (cone == TO_CONE_SECOND_ORDER && s < 0) return fail(TO_ERR_CONE, "Invalid second-order cone projection");
    if (status) status[i] = (cone == TO_CONE_SECOND_ORDER) ? s : 0;
  }
  return TO_OK;
}
int oracle_cone_projection_jacobian(int /*device*/, int32_t cone, int32_t dim, int64_t count, const double* x, double* jac) {
  CHECK_P(x); CHECK_P(jac);
  if (dim < 1 || dim > TO_MAX_P) return fail(TO_ERR_ARGUMENT, "cone dimension out of range");
  double J[TO_MAX_P * TO_MAX_P];
  for (int64_t i = 0; i < count; ++i) {
    int s = cone_projection_jacobian(cone, x + i * dim, J, dim);
    if (cone == TO_CONE_SECOND_ORDER && s < 0) return fail(TO_ERR_CONE, "Invalid second-order cone projection");
    for (int r = 0; r < dim; ++r) for (int c = 0; c < dim; ++c) jac[r + dim * (c + (size_t)dim * i)] = J[r * dim + c];
  }
  return TO_OK;
}
int oracle_cone_projection_hessian(int /*device*/, int32_t cone, int32_t dim, int64_t count, const double* x, const double* b, double* hess) {
  CHECK_P(x); CHECK_P(b); CHECK_P(hess);
  if (dim < 1 || dim > TO_MAX_P) return fail(TO_ERR_ARGUMENT, "cone dimension out of range");
  double H[TO_MAX_P * TO_MAX_P];
  for (int64_t i = 0; i < count; ++i) {
    int s = cone_projection_hessian(cone, x + i * dim, b + i * dim, H, dim);
    if (cone == TO_CONE_SECOND_ORDER && s < 0) return fail(TO_ERR_CONE, "Invalid second-order cone projection");
    for (int r = 0; r < dim; ++r) for (int c = 0; c < dim; ++c) hess[r + dim * (c + (size_t)dim * i)] = H[r * dim + c];
  }
  return TO_OK;
}

/* stand-alone model evaluation for tests: xdot = f(x,u), discrete step, state_diff */
int oracle_dynamics(int32_t model, const double* params, const double* x, const double* u, double* xdot) {
  Model M; M.id = model; std::memcpy(M.p, params, sizeof(M.p));
  if (model_dims(model, params, &M.n, &M.m, &M.ne)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  if (model == TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) return fail(TO_ERR_UNSUPPORTED, "a model vector has no single continuous dynamics");
  dynamics(M, x, u, xdot); return TO_OK;
}
int oracle_discrete_dynamics(int32_t model, const double* params, int32_t integrator, const double* x, const double* u, double h, double* xn) {
  Model M; M.id = model; std::memcpy(M.p, params, sizeof(M.p));
  if (model_dims(model, params, &M.n, &M.m, &M.ne)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  if (model == TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) return fail(TO_ERR_UNSUPPORTED, "a model vector steps per knot (rollout)");
  discrete_dynamics(M, integrator, x, u, h, xn); return TO_OK;
}
int oracle_state_diff(int32_t model, const double* params, const double* x, const double* x0, double* dx) {
  Model M; M.id = model; std::memcpy(M.p, params, sizeof(M.p));
  if (model_dims(model, params, &M.n, &M.m, &M.ne)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  state_diff(M, x, x0, dx); return TO_OK;
}
int oracle_state_add(int32_t model, const double* params, const double* x, const double* dx, double* xo) {
  Model M; M.id = model; std::memcpy(M.p, params, sizeof(M.p));
  if (model_dims(model, params, &M.n, &M.m, &M.ne)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  state_add(M, x, dx, xo); return TO_OK;
}

}  // extern "C"
