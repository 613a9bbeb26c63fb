// desc_lower.h — host-side lowering of the C-ABI descriptors (include/trajopt_hip.h): solver-option defaults and validation,
// model dimensions, constraint descriptors -> DevCon (selector tables), cost validation.  Pure host C++ (no HIP calls): shared
// by the library (trajopt_hip.hip) and by the host build of the projected-Newton kernel that the CPU test-suite runs
// (tests/host_shim/pn_harness.cpp).  Mirrors the reference's constructors: Problem src/problem.jl:44-72, add_constraint!
// src/constraint_list.jl:103-134, BoundConstraint src/constraints.jl:660-687, NormConstraint :442-455.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "models.h"
#include "problem_dev.h"

namespace to {
int fail(int code, const std::string& msg);  // records the message for to_last_error(), returns code

inline void default_opts(to_solver_opts* o) {
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

// Nonsense options used to give silent non-termination-style behaviour (every trajectory runs to MAX_ITERATIONS).
inline int validate_opts(const to_solver_opts& o) {
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

inline int model_dims(int id, const double* params, int* n, int* m, int* ne, int* key) {
  switch (id) {
    case TO_MODEL_DOUBLE_INTEGRATOR: {
      const int D = (int)params[1];
      if (D < 1 || D > 3) return -1;
      *n = 2 * D; *m = D; *ne = 2 * D; *key = D - 1; return 0;
    }
    case TO_MODEL_CARTPOLE: *n = 4; *m = 1; *ne = 4; *key = 3; return 0;
    case TO_MODEL_QUADROTOR: {  // params[10]: attitude representation of the state (to_rotation)
      const int rot = (int)params[10];
      if (rot == TO_ROT_QUATERNION) { *n = 13; *m = 4; *ne = 12; *key = 4; return 0; }
      if (rot == TO_ROT_MRP || rot == TO_ROT_RODRIGUES) { *n = 12; *m = 4; *ne = 12; *key = rot == TO_ROT_MRP ? 5 : 6; return 0; }
      return -1;
    }
    case TO_MODEL_HYBRID_DOUBLE_INTEGRATOR: *n = 4; *m = 2; *ne = 4; *key = 7; return 0;
    case TO_MODEL_VECTOR: *n = TO_VECTOR_N; *m = TO_VECTOR_M; *ne = TO_VECTOR_N; *key = 8; return 0;
    case TO_MODEL_INFEASIBLE: {  // Altro's InfeasibleModel over a small vector-space base (models.h InfeasibleModel): params[15] = base id
      const int base = (int)params[15];
      if (base == TO_MODEL_DOUBLE_INTEGRATOR) {
        const int D = (int)params[1];
        if (D < 1 || D > 2) return -1;  // (D = 3: m = 3 + 6 exceeds TO_MAX_M)
        *n = 2 * D; *m = 3 * D; *ne = 2 * D; *key = 8 + D; return 0;
      }
      if (base == TO_MODEL_CARTPOLE) { *n = 4; *m = 5; *ne = 4; *key = 11; return 0; }
      return -1;
    }
  }
  return -1;
}

// Model vector (TO_MODEL_VECTOR): every check RD.dims(models) makes (src/dynamics.jl:15-31), and the per-step table the kernels
// read (models.h ModelVectorModel: 64 doubles per step).
inline int lower_step_models(const to_step_model* sm, int N, std::vector<double>* table) {
  if (!sm) return fail(TO_ERR_NULL, "TO_MODEL_VECTOR needs step_models[N-1]");
  table->assign((size_t)64 * (N - 1), 0.0);
  for (int k = 0; k < N - 1; ++k) {
    const to_step_model& s = sm[k];
    double* r = table->data() + (size_t)64 * k;
    if (s.n < 1 || s.n > TO_VECTOR_N || s.m < 1 || s.m > TO_VECTOR_M || s.n_out < 1 || s.n_out > TO_VECTOR_N)
      return fail(TO_ERR_UNSUPPORTED, "model vector: step dimensions outside (6, 3)");
    switch (s.kind) {
      case TO_STEP_DOUBLE_INTEGRATOR:
        if (s.n != 2 * s.m || s.n_out != s.n) return fail(TO_ERR_DIMENSION_MISMATCH, "double integrator step: n = 2D, m = D, n_out = n");
        if (!(s.params[0] > 0.0)) return fail(TO_ERR_ARGUMENT, "double integrator step: mass must be positive");
        r[4] = s.params[0]; r[5] = s.m;
        break;
      case TO_STEP_CARTPOLE:
        if (s.n != 4 || s.m != 1 || s.n_out != 4) return fail(TO_ERR_DIMENSION_MISMATCH, "Cartpole step: (n, m, n_out) = (4, 1, 4)");
        for (int i = 0; i < 4; ++i) r[4 + i] = s.params[i];
        break;
      case TO_STEP_LINEAR_MAP:
        for (int i = 0; i < s.n_out; ++i) {
          for (int j = 0; j < s.n; ++j) r[8 + 6 * i + j] = s.params[i + s.n_out * j];
          for (int j = 0; j < s.m; ++j) r[44 + 3 * i + j] = s.params[s.n_out * s.n + i + s.n_out * j];
        }
        break;
      default: return fail(TO_ERR_UNSUPPORTED, "unknown step model kind");
    }
    r[0] = s.kind; r[1] = s.n; r[2] = s.m; r[3] = s.n_out;
    if (k + 1 < N - 1 && sm[k + 1].n != s.n_out)  // RD.dims: "Model mismatch at time step k"
      return fail(TO_ERR_DIMENSION_MISMATCH, "Model mismatch at time step " + std::to_string(k + 1) + ". Model " + std::to_string(k + 1) +
                                                 " has an output dimension of " + std::to_string(s.n_out) + " but model " + std::to_string(k + 2) +
                                                 " has a state dimension of " + std::to_string(sm[k + 1].n) + ".");
  }
  return TO_OK;
}

inline int validate_constraint(int n, int m, int N, const to_constraint_desc& d, DevCon* out) {
  const int nz = n + m;
  DevCon ci;
  std::memset(&ci, 0, sizeof(ci));
  ci.d = d;
  ci.cp_off = -1;  // shared parameters (to_set_constraint_params_batch flags a constraint later)
  if (d.k_first < 1 || d.k_last > N || d.k_first > d.k_last)
    return fail(TO_ERR_ASSERTION, "Invalid inds, inds[end] must be less than number of knotpoints");  // src/constraint_list.jl:112
  if (d.n_inds < 0 || d.n_inds > TO_MAX_CON_INDS || d.n_params < 0 || d.n_params > TO_MAX_CON_PARAMS)
    return fail(TO_ERR_ARGUMENT, "constraint inds/params count out of range");
  auto dimerr = [&](const char* what) {
    return fail(TO_ERR_DIMENSION_MISMATCH, std::string("New constraint not consistent with n=") + std::to_string(n) +
                                               " and m=" + std::to_string(m) + ": " + what);  // src/constraint_list.jl:109
  };
  // index lists address [x;u] positions one-to-one: a duplicate would make the reported Jacobian (last duplicate wins)
  // disagree with the one the solver accumulates
  for (int i = 0; i < d.n_inds; ++i)
    for (int j = i + 1; j < d.n_inds; ++j)
      if (d.inds[i] == d.inds[j]) return fail(TO_ERR_ARGUMENT, "constraint indices must be distinct");
  int p = 0;
  switch (d.kind) {
    case TO_CON_GOAL:
      if (d.sense != TO_CONE_ZERO) return fail(TO_ERR_ARGUMENT, "GoalConstraint sense must be Equality");
      if (d.n_params != d.n_inds) return dimerr("GoalConstraint length(xf) != length(inds)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return dimerr("GoalConstraint index outside the state");
      ci.width = n; p = d.n_inds; ci.selector = 1;
      for (int r = 0; r < p && r < TO_MAX_P; ++r) { ci.sidx[r] = d.inds[r] - 1; ci.ssgn[r] = 1.0; ci.soff[r] = d.params[r]; }
      break;
    case TO_CON_BOUND: {
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "BoundConstraint sense must be Inequality");
      if (d.n_params != 2 * nz) return dimerr("BoundConstraint needs z_max and z_min of length n+m");
      for (int i = 0; i < nz; ++i)
        if (!(d.params[i] >= d.params[nz + i])) return fail(TO_ERR_ARGUMENT, "Upper bounds must be greater than or equal to lower bounds");  // src/constraints.jl:712
      ci.width = nz; ci.selector = 1;
      for (int j = 0; j < nz; ++j) if (std::isfinite(d.params[j])) { if (p < TO_MAX_P) { ci.sidx[p] = j; ci.ssgn[p] = 1.0; ci.soff[p] = d.params[j]; } ++p; }
      for (int j = 0; j < nz; ++j) if (std::isfinite(d.params[nz + j])) { if (p < TO_MAX_P) { ci.sidx[p] = j; ci.ssgn[p] = -1.0; ci.soff[p] = d.params[nz + j]; } ++p; }
      break;
    }
    case TO_CON_NORM:
      if (d.n_params != 1) return fail(TO_ERR_ARGUMENT, "NormConstraint needs one parameter (val)");
      if (!(d.params[0] >= 0)) return fail(TO_ERR_ASSERTION, "Value must be greater than or equal to zero");  // src/constraints.jl:453
      if (d.sense != TO_CONE_ZERO && d.sense != TO_CONE_NEGATIVE_ORTHANT && d.sense != TO_CONE_SECOND_ORDER)
        return fail(TO_ERR_ARGUMENT, "NormConstraint sense must be Equality, Inequality or SecondOrderCone");
      if (d.n_inds < 1 || d.n_inds > nz) return dimerr("NormConstraint needs 1..n+m indices");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > nz) return dimerr("NormConstraint index outside [x;u]");
      ci.width = nz;
      if (d.sense == TO_CONE_SECOND_ORDER) {
        p = d.n_inds + 1; ci.selector = 1;
        for (int r = 0; r < d.n_inds; ++r) { ci.sidx[r] = d.inds[r] - 1; ci.ssgn[r] = 1.0; ci.soff[r] = 0.0; }
        ci.sidx[d.n_inds] = -1; ci.ssgn[d.n_inds] = 0.0; ci.soff[d.n_inds] = d.params[0];
      } else p = 1;
      break;
    case TO_CON_CIRCLE:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "CircleConstraint sense must be Inequality");
      if (d.n_inds != 2 || d.n_params % 3 != 0 || d.n_params == 0) return fail(TO_ERR_ASSERTION, "Lengths of xc, yc, and radius must be equal.");
      for (int i = 0; i < 2; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return dimerr("CircleConstraint index outside the state");
      ci.width = n; p = d.n_params / 3;
      break;
    case TO_CON_SPHERE:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "SphereConstraint sense must be Inequality");
      if (d.n_inds != 3 || d.n_params % 4 != 0 || d.n_params == 0) return fail(TO_ERR_ASSERTION, "Lengths of xc, yc, zc, and radius must be equal.");
      for (int i = 0; i < 3; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return dimerr("SphereConstraint index outside the state");
      ci.width = n; p = d.n_params / 4;
      break;
    case TO_CON_LINEAR:
      if (d.sense != TO_CONE_ZERO && d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "LinearConstraint sense must be Equality or Inequality");
      if (d.n_inds < 1 || d.n_inds > nz || d.n_params == 0 || d.n_params % (d.n_inds + 1) != 0) return fail(TO_ERR_ASSERTION, "size(A,1) == length(b)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > nz) return dimerr("LinearConstraint index outside [x;u]");
      ci.width = nz; p = d.n_params / (d.n_inds + 1);
      break;
    case TO_CON_COLLISION:
      if (d.sense != TO_CONE_NEGATIVE_ORTHANT) return fail(TO_ERR_ARGUMENT, "CollisionConstraint sense must be Inequality");
      if (d.n_inds < 2 || d.n_inds % 2 != 0) return fail(TO_ERR_ASSERTION, "Position dimensions must be of equal length"); /* src/constraints.jl:349 */
      if (d.n_inds > n) return dimerr("CollisionConstraint has more position indices than states");
      if (d.n_params != 1) return fail(TO_ERR_ARGUMENT, "CollisionConstraint needs one parameter (radius)");
      for (int i = 0; i < d.n_inds; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "CollisionConstraint index outside state");
      ci.width = n; p = 1; break;
    case TO_CON_QUATVEC:
      if (d.sense != TO_CONE_ZERO) return fail(TO_ERR_ARGUMENT, "QuatVecEq sense must be Equality");
      if (d.n_inds != 4 || d.n_params != 4) return fail(TO_ERR_ARGUMENT, "QuatVecEq needs 4 quaternion indices and a 4-vector qf");
      for (int i = 0; i < 4; ++i) if (d.inds[i] < 1 || d.inds[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "QuatVecEq index outside state");
      ci.width = n; p = 3; break;
    default: return fail(TO_ERR_UNSUPPORTED, "unknown constraint kind");
  }
  if (p < 1 || p > TO_MAX_P) return fail(TO_ERR_UNSUPPORTED, "constraint output dimension outside 1..TO_MAX_P");
  if (ci.selector && (d.kind == TO_CON_GOAL || (d.kind == TO_CON_NORM && d.sense == TO_CONE_SECOND_ORDER))) {
    // fast layouts: rows map to a state prefix z[r] or to the control block z[n+r] with sign +1 (see problem_dev.h)
    const int D = d.n_inds;
    bool prefix = D <= n, ctrl = D <= m;
    for (int r = 0; r < D; ++r) { prefix = prefix && d.inds[r] == r + 1; ctrl = ctrl && d.inds[r] == n + r + 1; }
    ci.fast = prefix ? 1 : ctrl ? 2 : 0;
  }
  if (d.p != 0 && d.p != p) return fail(TO_ERR_DIMENSION_MISMATCH, "constraint output dimension does not match its descriptor");
  ci.p = p; ci.k1 = d.k_first - 1; ci.k2 = d.k_last - 1;
  *out = ci;
  return TO_OK;
}

// rot: attitude representation of the model's state (to_rotation), -1 for vector-space models
inline int validate_cost(int n, int rot, const to_cost_desc& c) {
  if (c.kind != TO_COST_DIAGONAL && c.kind != TO_COST_QUADRATIC && c.kind != TO_COST_DIAGONAL_QUAT && c.kind != TO_COST_ERROR_QUADRATIC)
    return fail(TO_ERR_UNSUPPORTED, "unknown cost kind");
  if (c.kind == TO_COST_ERROR_QUADRATIC) {  // needs the rigid-body state layout [r; attitude; v; w]
    if (rot < 0) return fail(TO_ERR_ARGUMENT, "ErrorQuadratic needs a rigid-body model");
    if ((int)c.w != rot || c.w != (double)rot) return fail(TO_ERR_ARGUMENT, "ErrorQuadratic: w must name the model's attitude representation (to_rotation)");
    if (rot == TO_ROT_QUATERNION)
      for (int i = 0; i < 4; ++i) if (c.q_ind[i] != 4 + i) return fail(TO_ERR_UNSUPPORTED, "ErrorQuadratic: q_ind must be 4:7");
  }
  if (c.kind == TO_COST_DIAGONAL_QUAT) {
    if (rot > TO_ROT_QUATERNION) return fail(TO_ERR_ARGUMENT, "DiagonalQuatCost needs a state that carries a unit quaternion");
    for (int i = 0; i < 4; ++i) if (c.q_ind[i] < 1 || c.q_ind[i] > n) return fail(TO_ERR_DIMENSION_MISMATCH, "quat_ind outside the state");
  }
  return TO_OK;
}

}  // namespace to
