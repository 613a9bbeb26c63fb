/*
 * trajopt_hip.h — C-ABI of libtrajopt_hip.so: MI355X-native batched trajectory-optimisation
 * hot path (RK4 rollout -> dynamics/cost/constraint expansion -> backward Riccati ->
 * forward line-search rollout -> iLQR / augmented-Lagrangian loop) for a BATCH of
 * independent trajectories.
 *
 * This header is the drop-in boundary.  The reference (TrajectoryOptimization.jl v0.7.1,
 * pure Julia) has no FFI; the entry points below are what a Julia `ccall` shim keeping the
 * reference's type names (Problem / Objective / ConstraintList / KnotPoint) binds, one per
 * operator of the reference's solver-facing API (see INTEGRATION.md for the shim):
 *
 *   to_rollout                  <- rollout!(prob)                       src/problem.jl:330-340
 *   to_cost / to_stage_costs    <- cost(prob) / cost!(obj,Z)            src/problem.jl:321, src/objective.jl:89-106
 *   to_cost_expansion           <- RD.gradient!/RD.hessian! per knot    src/cost_functions.jl:137-233, src/lie_costs.jl:78-95
 *   to_evaluate_constraints     <- evaluate_constraints!                src/abstract_constraint.jl:200-225
 *   to_constraint_jacobians     <- constraint_jacobians!                src/abstract_constraint.jl:236-248
 *   to_cone_projection{,_jacobian,_hessian} <- projection!/∇projection!/∇²projection!  src/cones.jl:96-276
 *   to_expand / to_backward / to_forward / to_ilqr_solve / to_al_solve
 *                               <- the Altro.jl iLQR / AL loops that consume a Problem
 *                                  (out of tree; SURVEY.md §8a rows E1,S1-S4)
 *   to_set_* / to_get_*         <- initial_controls!/initial_states!/set_initial_state!/states/controls
 *                                  src/problem.jl:198-310
 *
 * Conventions
 *   - All floating point is IEEE double (reference: Problem{T<:AbstractFloat}, Float64 everywhere).
 *   - Host arrays are CALLER-OWNED, in the reference's native layout: column-major, state index
 *     fastest, then knot, then trajectory:   X[i + n*(k + N*b)],  U[j + m*(k + (N-1)*b)].
 *     Knot indices inside descriptors are 1-based inclusive ranges like Julia `inds::UnitRange`
 *     (src/constraint_list.jl:38); state/control indices (`inds`, `q_ind`) are 1-based too.
 *   - Device memory is library-owned behind the handle, batch-fastest SoA.
 *   - Every function returns int: 0 = ok, negative = error class mirroring the reference's
 *     exception (to_status_code); text via to_last_error().  Nothing throws across the boundary.
 *   - Integer outputs (iterations, status, line-search index, active flags) are int32.
 *   - A handle owns one HIP stream and is not thread-safe (like Objective.J / tmpu scratch in
 *     the reference, src/objective.jl:29, src/cost_functions.jl:431).
 *   - There is NO CPU fallback: every compute entry point fails with TO_ERR_HIP when no HIP
 *     device is usable.
 */
#ifndef TRAJOPT_HIP_H
#define TRAJOPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round 1.  2: to_solver_opts::reserved1 became al_full_newton (validated: 0 or 1), new entry points
 * to_constraint_hessians, to_comm_*, to_allgather, to_allgather_stats, to_comm_shards, to_solver_path, to_build_id; to_cost_desc gained the
 * ERROR_QUADRATIC error maps.  3: TO_MODEL_HYBRID_DOUBLE_INTEGRATOR and to_knot_dims (model vectors whose dimensions change
 * along the horizon), to_solver_path reports 8 values.  4: projected-Newton polish — to_solver_opts gained the Altro
 * ProjectedNewtonSolver options (appended), to_solve_stats gained iterations_pn, new status TO_PROJECTION_FAIL, new entry
 * points to_pn_solve, to_altro_solve, to_dynamics_defect and the asynchronous to_*_solve_async / to_solve_wait; TO_MODEL_VECTOR with
 * to_problem_desc::step_models (general model vectors).  Policy: the version changes whenever a struct field changes meaning or a symbol is added; a
 * host checks to_abi_version() == TO_ABI_VERSION right after dlopen (the Python and Julia shims do) and to_create rejects
 * a descriptor stamped with another version.  5: TO_MODEL_INFEASIBLE, to_infeasible_controls, to_set_cost_linear_batch, to_get_cost_to_go.
 * 6: to_solve_progress / to_solve_wait_below (pipelined solves over several handles), to_set_constraint_params_batch /
 * to_clear_constraint_params_batch (one GoalConstraint target per trajectory); TO_STATE_LIMIT / TO_CONTROL_LIMIT are
 * emitted; TRAJOPT_RCCL_LIB, TRAJOPT_GUARD. */
#define TO_ABI_VERSION 6

#define TO_MAX_N 16       /* max state dimension            */
#define TO_MAX_M 8        /* max control dimension          */
#define TO_MAX_P 40       /* max rows of one constraint     */
#define TO_MAX_CON_PARAMS 400
#define TO_MAX_CON_INDS 48

/* ---- return codes ---------------------------------------------------------------------- */
typedef enum {
  TO_OK = 0,
  TO_ERR_DIMENSION_MISMATCH = -1, /* Julia DimensionMismatch (src/problem.jl:64-68, src/constraint_list.jl:109) */
  TO_ERR_ARGUMENT = -2,           /* Julia ArgumentError     (src/problem.jl:88, src/constraints.jl:712)       */
  TO_ERR_ASSERTION = -3,          /* Julia AssertionError    (src/problem.jl:49-55)                            */
  TO_ERR_HIP = -4,                /* HIP runtime failure / no device                                           */
  TO_ERR_UNSUPPORTED = -5,        /* descriptor combination outside the hot-path scope                         */
  TO_ERR_NULL = -6,               /* null pointer                                                              */
  TO_ERR_CONE = -7                /* ErrorException("Invalid second-order cone projection") src/cones.jl:124    */
} to_status_code;

/* ---- per-trajectory solver status (Altro.jl TerminationStatus order) --------------------- */
typedef enum {
  TO_UNSOLVED = 0,
  TO_LINESEARCH_FAIL = 1,
  TO_SOLVE_SUCCEEDED = 2,
  TO_MAX_ITERATIONS = 3,
  TO_MAX_ITERATIONS_OUTER = 4,
  TO_MAXIMUM_COST = 5,
  TO_STATE_LIMIT = 6,   /* the INITIAL rollout of a solve left |x| <= max_state_value (checked knot by knot, the state first — Altro's   */
  TO_CONTROL_LIMIT = 7, /* rollout!) / |u| <= max_control_value: the solve ends there, no iteration performed.  Inside a line search such a
                           candidate is rejected like any other failed step size and no status is reported */
  TO_NO_PROGRESS = 8,
  TO_COST_INCREASE = 9,
  TO_REGULARIZATION_MAX = 10,
  TO_PROJECTION_FAIL = 11 /* to_pn_solve / to_altro_solve: the projected-Newton polish ended above constraint_tolerance (Altro keeps
                             SOLVE_SUCCEEDED there and only warns; here the status tells) */
} to_solver_status;

/* ---- models (RobotZoo.jl / examples, restated; SURVEY.md §8a R2,R3) ----------------------- */
typedef enum {
  TO_MODEL_DOUBLE_INTEGRATOR = 0, /* examples/quickstart.jl:11-23; n=2D, m=D; params[0]=mass, params[1]=D (1,2,3) */
  TO_MODEL_CARTPOLE = 1,          /* docs/src/model.md:20-51; params = mc, mp, l, g                               */
  TO_MODEL_QUADROTOR = 2,         /* examples/Quadrotor.ipynb cells 4,8; params = mass, Jx,Jy,Jz, gx,gy,gz,
                                     motor_dist, kf, km, rotation (to_rotation, params[10]; 0 = the notebook's
                                     Quadrotor{QuatRotation}: n = 13)                                            */
  TO_MODEL_HYBRID_DOUBLE_INTEGRATOR = 3, /* the model VECTOR of test/hybrid_dynamics_model.jl:14-52 (src/dynamics.jl:15-31): a 2-D double
                                     integrator (4, 2) for the first S = params[1] time steps, a jump map (4, 2) -> 2,
                                     x+ = [(x3 + x4)/2, (u1 + u2)/2], on step S + 1, then a 1-D double integrator (2, 1);
                                     params[0] = mass; needs 1 <= S <= N - 2.  Stored at the largest dimensions (n = 4, m = 2):
                                     states / controls of the narrower knots are zero-padded, costs and constraints of
                                     those knots are given at (4, 2) with nothing on the padding (a padded control needs a
                                     positive R entry: it then stays exactly 0).  to_knot_dims reports the live dimensions. */
  TO_MODEL_VECTOR = 4,             /* Problem(models::Vector{<:DiscreteDynamics}, ...) in general (src/problem.jl:36-73, src/dynamics.jl:15-31):
                                     one model per time step, to_problem_desc::step_models[N-1], any mix of the small compiled-in models
                                     and linear discrete maps whose dimensions chain (output dimension of step k = state dimension of
                                     step k+1, checked like RD.dims).  Stored at (n, m) = (TO_VECTOR_N, TO_VECTOR_M) = (6, 3), narrower
                                     knots zero-padded exactly as for the hybrid double integrator; model_params are unused. */
  TO_MODEL_INFEASIBLE = 5          /* Altro's InfeasibleModel — the state augmentation of ALTRO's infeasible start, what the reference's
                                     change_dimension family exists for (src/constraints.jl:820-936, src/constraint_list.jl:208-217,
                                     src/cost_functions.jl:391-401): x+ = f_d(x, u[1:m0]) + u[m0+1 : m0+n], one slack control per state on top
                                     of a base model.  model_params[15] = to_model_id of the base (DOUBLE_INTEGRATOR with D = 1, 2 or
                                     CARTPOLE), model_params[0..14] = the base model's parameters; n = n_base, m = m_base + n_base.  The
                                     host composes the rest as Altro does (costs / constraints lifted with change_dimension, R_inf on the
                                     slacks, the equality u[m0+1:] = 0 on every stage knot) and seeds the slacks from a state guess with
                                     to_infeasible_controls.  A Quadrotor base (13, 17) exceeds TO_MAX_M and the 16 x 16 tile of its
                                     backward pass: TO_ERR_UNSUPPORTED. */
} to_model_id;

/* One time step of a model vector (TO_MODEL_VECTOR).  A continuous model is discretised with the problem's integrator; a linear
 * map is a discrete map x+ = A x + B u from (n, m) to n_out states (the jump map of test/hybrid_dynamics_model.jl:31-33 is one). */
#define TO_VECTOR_N 6
#define TO_VECTOR_M 3
typedef enum {
  TO_STEP_DOUBLE_INTEGRATOR = 0, /* n = 2D, m = D, n_out = n; params[0] = mass */
  TO_STEP_CARTPOLE = 1,          /* n = 4, m = 1, n_out = 4; params = mc, mp, l, g */
  TO_STEP_LINEAR_MAP = 2         /* params = A (n_out x n, column-major), then B (n_out x m, column-major) */
} to_step_kind;
typedef struct {
  int32_t kind;   /* to_step_kind */
  int32_t n, m;   /* state / control dimension of this step (RD.dims) */
  int32_t n_out;  /* output dimension: the state dimension of the next knot */
  double params[60];
} to_step_model;

/* Attitude representation R of a RigidBody{R} state (examples/Quadrotor.ipynb cell 5: "typically one of QuatRotation{T},
 * MRP{T}, or RodriguesParam{T}"; src/lie_costs.jl:1-3).  QUATERNION: x = [r, q(w,x,y,z), v, w], n = 13.  MRP / RODRIGUES:
 * x = [r, p(3), v, w], n = 12.  The error state has 12 entries in every case (RD.state_diff with the Cayley map: the
 * Rodrigues vector of the relative rotation). */
typedef enum { TO_ROT_QUATERNION = 0, TO_ROT_MRP = 1, TO_ROT_RODRIGUES = 2 } to_rotation;

typedef enum { TO_RK4 = 0, TO_RK3 = 1, TO_EULER = 2 } to_integrator; /* default RK4: src/problem.jl:120 */

/* ---- cost functions (src/cost_functions.jl, src/lie_costs.jl) ----------------------------- */
typedef enum {
  TO_COST_DIAGONAL = 0,      /* DiagonalCost      src/cost_functions.jl:326-346: Q,R hold diagonals     */
  TO_COST_QUADRATIC = 1,     /* QuadraticCost     src/cost_functions.jl:422-453: dense Q (n x n), R (m x m), H (m x n), column-major */
  TO_COST_DIAGONAL_QUAT = 2, /* DiagonalQuatCost  src/lie_costs.jl:34-55: diagonal + w*min(1 +/- q_ref'q) */
  TO_COST_ERROR_QUADRATIC = 3 /* ErrorQuadratic  src/lie_costs.jl:178-241 (rigid bodies, n = 13): 0.5 dx'Q dx + c + 0.5 u'Ru + r'u with
                                 dx = state_diff(x, x_ref, CayleyMap) in R^12; Q[0..12) = error-state diagonal, R/r diagonal/linear
                                 control terms, q[0..n) = x_ref, q_ind = quaternion indices (4,5,6,7).  Gradient and Hessian are
                                 exact (ForwardDiff in the reference).  w = attitude representation of the model's state
                                 (to_rotation as a double): ErrorQuadratic{QuatRotation} (n = 13), {MRP}, {RodriguesParam} (n = 12). */
} to_cost_kind;

typedef struct {
  int32_t kind;                     /* to_cost_kind */
  int32_t terminal;                 /* the reference's `terminal` field; informational (terminal-ness of a knot is dt==0) */
  double Q[TO_MAX_N * TO_MAX_N];    /* diagonal kinds: first n entries; QUADRATIC: n*n col-major (ld = n) */
  double R[TO_MAX_M * TO_MAX_M];    /* diagonal kinds: first m entries; QUADRATIC: m*m col-major (ld = m) */
  double H[TO_MAX_M * TO_MAX_N];    /* QUADRATIC only: m*n col-major (ld = m); cost term u'Hx            */
  double q[TO_MAX_N];
  double r[TO_MAX_M];
  double c;
  double w;                         /* DIAGONAL_QUAT: weight of the geodesic term; ERROR_QUADRATIC: to_rotation of the state */
  double q_ref[4];                  /* DIAGONAL_QUAT: reference quaternion (w,x,y,z) */
  int32_t q_ind[4];                 /* DIAGONAL_QUAT: 1-based state indices of the quaternion (default 4:7) */
} to_cost_desc;

/* ---- cones / constraint sense (src/cones.jl:17-61) --------------------------------------- */
typedef enum {
  TO_CONE_ZERO = 0,              /* Equality         */
  TO_CONE_NEGATIVE_ORTHANT = 1,  /* Inequality c<=0  */
  TO_CONE_SECOND_ORDER = 2,      /* [v; s], |v|<=s, scalar LAST */
  TO_CONE_POSITIVE_ORTHANT = 3,
  TO_CONE_IDENTITY = 4
} to_cone;

/* ---- constraints (src/constraints.jl) ----------------------------------------------------- */
typedef enum {
  TO_CON_GOAL = 0,    /* GoalConstraint  :22-87   sense Equality; inds[p] = state indices, params[0..p) = xf[inds]          */
  TO_CON_BOUND = 1,   /* BoundConstraint :644-783 sense Inequality; params[0..n+m) = z_max, params[n+m..2(n+m)) = z_min
                         (+/-inf allowed); rows ordered [finite max rows; finite min rows]                                  */
  TO_CON_NORM = 2,    /* NormConstraint  :438-521 params[0] = val; inds[D] = indices into z=[x;u]; sense Equality /
                         Inequality (c = |z[inds]|^2 - val^2, p=1) or SecondOrder (c = [z[inds]; val], p=D+1)                */
  TO_CON_CIRCLE = 3,  /* CircleConstraint :168-233 sense Inequality; inds[0]=xi, inds[1]=yi; params = x[P], y[P], r[P]; p=P  */
  TO_CON_SPHERE = 4,  /* SphereConstraint :249-326 inds = xi,yi,zi; params = x[P], y[P], z[P], r[P]                          */
  TO_CON_LINEAR = 5,  /* LinearConstraint :103-150 params = A (p x D col-major), then b[p]; inds[D] into z; sense Eq/Ineq    */
  TO_CON_COLLISION = 6, /* CollisionConstraint :332-393 sense Inequality; inds = [x1(D); x2(D)] state indices, params[0] = radius;
                          c = r^2 - |x[x1] - x[x2]|^2, p = 1                                                                 */
  TO_CON_QUATVEC = 7   /* QuatVecEq :938-965 sense Equality; inds[4] = quaternion indices (w,x,y,z), params[0..4) = qf (unit);
                          c = vec(q/|q|) - sign(qf'q) vec(qf), p = 3; Jacobian = d/dq of the normalisation (ForwardDiff there) */
} to_con_kind;

typedef struct {
  int32_t kind;     /* to_con_kind */
  int32_t sense;    /* to_cone */
  int32_t k_first;  /* 1-based inclusive knot range (ConstraintList.inds, src/constraint_list.jl:38) */
  int32_t k_last;
  int32_t p;        /* output dimension; 0 = let the library derive it; otherwise validated */
  int32_t n_inds;
  int32_t inds[TO_MAX_CON_INDS];
  int32_t n_params;
  double params[TO_MAX_CON_PARAMS];
} to_constraint_desc;

/* ---- problem (src/problem.jl:36-73) -------------------------------------------------------- */
typedef struct {
  int32_t abi_version; /* TO_ABI_VERSION */
  int32_t model;       /* to_model_id   */
  int32_t integrator;  /* to_integrator */
  int32_t n, m;        /* state / control dims; validated against the model (RD.dims) */
  int32_t N;           /* knot points  (N-1 models, src/problem.jl:49) */
  int32_t B;           /* batch: number of independent trajectories held by this handle */
  double model_params[16];
  double t0, tf;       /* tf > t0 asserted (src/problem.jl:50); uniform dt = (tf-t0)/(N-1) unless dt given */
  const double* dt;    /* optional [N-1] step sizes (test/problems_tests.jl:78-85), must sum to tf-t0; NULL = uniform */
  int32_t n_costs;           /* number of distinct cost functions */
  const to_cost_desc* costs; /* [n_costs] */
  const int32_t* cost_index; /* [N] 0-based index into costs per knot (Objective.cost, src/objective.jl:28);
                                NULL = Objective(stage, terminal, N): costs[0] for k<N, costs[1] at k=N (src/objective.jl:74-77) */
  int32_t n_constraints;
  const to_constraint_desc* constraints; /* ConstraintList order (src/constraint_list.jl:103-134) */
  const to_step_model* step_models;      /* TO_MODEL_VECTOR: [N-1] one model per time step; NULL otherwise */
} to_problem_desc;

/* ---- solver options (names follow Altro.jl SolverOptions; examples/Cartpole.ipynb cell 17) -- */
typedef struct {
  /* iLQR */
  double cost_tolerance;              /* 1e-4 */
  double gradient_tolerance;          /* 10.0 */
  int32_t iterations;                 /* 300: max inner iterations per iLQR solve */
  int32_t dJ_counter_limit;           /* 10 */
  int32_t iterations_linesearch;      /* 20 */
  int32_t reserved0;
  double line_search_lower_bound;     /* 1e-8 */
  double line_search_upper_bound;     /* 10.0 */
  double line_search_decrease_factor; /* 0.5 */
  double bp_reg_initial;              /* 0.0 */
  double bp_reg_increase_factor;      /* 1.6 */
  double bp_reg_min;                  /* 1e-8 */
  double bp_reg_max;                  /* 1e8 */
  double bp_reg_fp;                   /* 10.0 */
  double max_cost_value;              /* 1e8 */
  double max_state_value;             /* 1e8 */
  double max_control_value;           /* 1e8 */
  /* augmented Lagrangian */
  double constraint_tolerance;        /* 1e-6 */
  double cost_tolerance_intermediate; /* 1e-4 */
  double penalty_initial;             /* 1.0 */
  double penalty_scaling;             /* 10.0 */
  double penalty_max;                 /* 1e8 */
  double dual_max;                    /* 1e8 */
  int32_t iterations_outer;           /* 30 */
  int32_t cost_dt_scaling;            /* 0 (v0.7 semantics, NEWS.md:11-12); 1 = legacy: stage costs multiplied by dt */
  int32_t iterations_total;           /* 1000: cap on inner iterations summed over AL outer loops */
  int32_t al_full_newton;             /* 0: Gauss-Newton AL Hessian (Altro's default).  1: the expansion adds the constraint curvature
                                         sum_r ybar_r * d2c_r/dz2, ybar = the multiplier estimate lambda + I_mu c, with the closed forms of
                                         to_constraint_hessians (src/abstract_constraint.jl:255-280: the nabla-jacobian! term) */
  /* projected-Newton polish (Altro.jl ProjectedNewtonSolver, the last stage of ALTRO: examples/Cartpole.ipynb cells 17-19,
   * examples/Quadrotor.ipynb cell 20; names follow Altro's SolverOptions).  to_altro_solve runs the AL stage until the violation
   * is below projected_newton_tolerance and hands the trajectory to the polish, which ends at constraint_tolerance. */
  double projected_newton_tolerance;  /* 1e-3 */
  double active_set_tolerance_pn;     /* 1e-3: an inequality row takes part in the projection when c >= -tol */
  double rho_chol;                    /* 1e-8 (Altro: 1e-2): regularisation of S = D H^-1 D' before its Cholesky factorisation */
  double rho_primal;                  /* 1e-8: added to the diagonal cost Hessian H (the metric of the projection) */
  double r_threshold;                 /* 1.1: stop refining on one linearisation once log10(viol)/log10(viol_prev) drops below */
  int32_t n_steps;                    /* 2: the polish runs at most n_steps + 1 linearisations (Altro: `while count <= n_steps`) */
  int32_t projected_newton;           /* 1: to_altro_solve polishes; 0: to_altro_solve == to_al_solve */
} to_solver_opts;

/* caller-allocated outputs of a solve; any pointer may be NULL */
typedef struct {
  int32_t* iterations;        /* [B] inner iLQR iterations performed by each trajectory */
  int32_t* iterations_outer;  /* [B] AL outer iterations (0 for to_ilqr_solve) */
  int32_t* status;            /* [B] to_solver_status */
  double* cost;               /* [B] objective cost J at the solution (cost(prob), no AL terms) */
  double* dJ;                 /* [B] last accepted cost decrease */
  double* gradient;           /* [B] last iLQR gradient metric */
  double* c_max;              /* [B] max constraint violation (0 without constraints); to_pn_solve / to_altro_solve: the dynamics and
                                 initial-condition defects count too (a polished trajectory is no longer an exact rollout) */
  double* penalty_max;        /* [B] largest penalty used */
  int32_t* iterations_pn;     /* [B] projection solves (linearisations) of the projected-Newton polish; 0 where it did not run */
  /* aggregates written by the library */
  int64_t total_iterations;   /* sum_b iterations[b] (the numerator of the headline metric) */
  int32_t batch_steps;        /* batch-synchronous device iterations executed */
  int32_t reserved;
  double solve_ms;            /* device time of the solve loop (hipEvent) */
} to_solve_stats;

typedef struct to_handle_s to_handle;

/* ---- library ------------------------------------------------------------------------------ */
int to_abi_version(void);
const char* to_build_id(void);             /* hash of the sources this binary was compiled from (build.py stamps it; tests and
                                              bench.py print it, build() recompiles when it differs from the tree) */
const char* to_last_error(void);
int to_device_count(int* count);           /* number of usable HIP devices */
int to_default_options(to_solver_opts* o); /* fill with the defaults documented above */

/* ---- handle lifecycle --------------------------------------------------------------------- */
/* Validates the descriptor exactly as the reference's constructors do (Problem inner ctor
 * src/problem.jl:44-72, add_constraint! src/constraint_list.jl:103-134, BoundConstraint ctor
 * src/constraints.jl:660-687, NormConstraint ctor :442-455) and allocates device storage on
 * `device` (HIP ordinal).  opts may be NULL (defaults). */
int to_create(const to_problem_desc* desc, const to_solver_opts* opts, int device, to_handle** out);
int to_destroy(to_handle* h);
/* Options are validated (TO_ERR_ARGUMENT for out-of-range values, here and in to_create).  The number of line-search
 * candidate slots is fixed at to_create from iterations_linesearch: raising it later works, with less concurrency than
 * a fresh handle would have. */
int to_set_options(to_handle* h, const to_solver_opts* opts);
int to_get_options(const to_handle* h, to_solver_opts* opts);
int to_sync(to_handle* h);                 /* block until the handle's stream is idle */
void* to_stream(to_handle* h);             /* the hipStream_t the handle launches on */

/* sizes */
int to_dims(const to_handle* h, int32_t* n, int32_t* m, int32_t* n_err, int32_t* N, int32_t* B);
int to_num_constraints(const to_handle* h, int32_t* p_per_knot /* [N], ConstraintList.p src/constraint_list.jl:44 */);

/* ---- trajectory I/O (host layout (n,N,B) / (m,N-1,B) column-major) ------------------------- */
int to_set_initial_state(to_handle* h, const double* x0 /* [n*B] */);  /* set_initial_state! src/problem.jl:275 */
int to_set_controls(to_handle* h, const double* U /* [m*(N-1)*B] */);  /* initial_controls!  src/problem.jl:255-268 */
int to_set_states(to_handle* h, const double* X /* [n*N*B] */);        /* initial_states!    src/problem.jl:242-253 */
int to_set_controls_uniform(to_handle* h, const double* u /* [m] */);  /* initial_controls!(prob, u0) for every k and b */
int to_get_states(to_handle* h, double* X);
int to_get_controls(to_handle* h, double* U);
int to_get_initial_state(to_handle* h, double* x0);
/* TO_MODEL_INFEASIBLE only — Altro's infeasible_controls: from the CURRENT states X (an initial_states! guess, src/problem.jl:242-253,
 * with x_1 = x0) and the base controls U[1:m0], set the slack controls w_k = x_{k+1} - f_d(x_k, u_k) so that a rollout reproduces X:
 * the guess is then what the solve starts from (every solve begins with a rollout of the controls). */
int to_infeasible_controls(to_handle* h);
/* device-to-device copies into caller-provided DEVICE buffers (same (n,N,B) layout), for RCCL all-gather */
int to_get_states_device(to_handle* h, void* dX);
int to_get_controls_device(to_handle* h, void* dU);
/* ---- multi-GPU: one process per GPU, each handle owns a contiguous shard of the batch; no collective inside
 * the solves; one RCCL all-gather (over xGMI) of the converged trajectories.  librccl.so is dlopen'ed on first use.
 *   rank 0: to_comm_unique_id(id) -> ship the 128 bytes to the other ranks (MPI, a file, torch.distributed ...)
 *   all   : to_comm_init_rank(h, nranks, rank, id); ... solve ...; to_allgather(h, dX_all, dU_all)
 * dX_all / dU_all are caller-owned DEVICE buffers of n*N*B_total / m*(N-1)*B_total doubles; the result is the host layout
 * (n, N, B_total) with trajectories in global order (rank-major).  Shards may differ in size (a batch that does not divide
 * by the number of GPUs): the sizes are exchanged at to_comm_init_rank, to_comm_shards reports them; equal shards take one
 * in-place ncclAllGather, unequal ones one grouped ncclBroadcast per rank.  to_allgather_stats is the small gather of
 * iterations / status / objective cost of every trajectory (HOST arrays of B_total entries, any may be NULL). */
int to_comm_unique_id(void* id128 /* [128] bytes */);
int to_comm_init_rank(to_handle* h, int32_t nranks, int32_t rank, const void* id128);
int to_comm_shards(const to_handle* h, int32_t* nranks, int32_t* rank, int64_t* B_total, int32_t* counts /* [nranks], nullable */);
int to_allgather(to_handle* h, void* dX_all, void* dU_all /* either may be NULL */);
int to_allgather_stats(to_handle* h, int32_t* iterations_all, int32_t* status_all, double* J_all /* HOST [B_total]; any may be NULL */);
int to_comm_destroy(to_handle* h);
/* goal / reference updates between solves (set_goal_state! src/problem.jl:294-310; set_LQR_goal! src/cost_functions.jl:249-258) */
int to_set_cost(to_handle* h, int32_t cost_id, const to_cost_desc* cost);
int to_set_constraint(to_handle* h, int32_t con_id, const to_constraint_desc* con);
/* One goal per TRAJECTORY (SURVEY.md §8b "optionally per-trajectory xf": batched MPC, goal sweeps): set_LQR_goal!(cost, xf_b, uf_b)
 * (src/cost_functions.jl:249-258 — "only changes q and r") for every trajectory of the batch at once.  q[n, B] / r[m, B] (column-major,
 * either may be NULL) REPLACE the linear terms of cost `cost_id` for trajectory b; Q, R, H, c stay the descriptor's.  The host mirrors
 * compute q_b = -Q xf_b (set_goal_state!(prob, Xf::Matrix), src/problem.jl:294-310).  Kinds DIAGONAL, QUADRATIC, DIAGONAL_QUAT (its
 * quaternion reference q_ref stays shared); TO_ERR_UNSUPPORTED for ERROR_QUADRATIC.  to_set_cost on a cost resets its per-trajectory
 * terms; to_clear_cost_linear_batch returns the whole handle to shared descriptors.  A GoalConstraint's target per trajectory:
 * to_set_constraint_params_batch (below).  The projected-Newton polish builds its metric from the
 * descriptors alone (per-trajectory q does not enter a Hessian except through the attitude term of quaternion states, which the polish
 * then takes from the shared q). */
int to_set_cost_linear_batch(to_handle* h, int32_t cost_id, const double* q /* [n*B] or NULL */, const double* r /* [m*B] or NULL */);
int to_clear_cost_linear_batch(to_handle* h);
/* One constraint-parameter set per TRAJECTORY (round 6): set_goal_state!(prob, xf; constraint = true) updates the GoalConstraints too
 * (src/problem.jl:303-309, src/constraints.jl:22-87) — here for every trajectory of the batch at once.  con_id names a GOAL constraint,
 * params[p, B] (column-major) = xf_b[inds], the target of trajectory b — or a LINEAR constraint (src/constraints.jl:103-150), params[p, B] =
 * b_b, its right-hand side for trajectory b (the rows of A must be linearly independent: TO_ERR_UNSUPPORTED otherwise).  Everything that
 * evaluates the constraint — AL terms and their expansion, violation, dual update, the projected-Newton polish, to_evaluate_constraints —
 * then uses the trajectory's own parameters (the solves run the general kernel variants while any constraint carries them).  On the
 * device the difference from the descriptor is a shift of z = [x; u] as that constraint sees it: the GoalConstraint with target xf + d is
 * the shared one at x - d; A z = b + db is the shared one at z - A'(A A')^-1 db (values agree with the direct evaluation to rounding,
 * Jacobians exactly).  to_set_constraint on the constraint returns it to shared parameters, to_clear_constraint_params_batch all of
 * them.  Other kinds: TO_ERR_UNSUPPORTED. */
int to_set_constraint_params_batch(to_handle* h, int32_t con_id, const double* params /* [p*B] */);
int to_clear_constraint_params_batch(to_handle* h);

/* ---- the hot path, phase by phase ----------------------------------------------------------- */
int to_rollout(to_handle* h);                                   /* rollout!  src/problem.jl:330-340 */
int to_cost(to_handle* h, double* J /* [B] */);                  /* cost      src/objective.jl:89-93  */
int to_stage_costs(to_handle* h, double* Jk /* [N*B], Objective.J */);
int to_expand(to_handle* h);     /* dynamics Jacobians (error-state) + cost expansion (+AL terms) at the current (X,U) */
int to_backward(to_handle* h);   /* Riccati recursion -> K, d, dV; regularises per trajectory */
/* ls_index: index of the accepted step size alpha = decrease^index; -1 = line search failed (nominal kept, regularisation
 * raised).  At a stationary point (predicted decrease <= 1e-12 (1+|J|)) the ZERO step is taken: ls_index = 0 with
 * J_new == J and an unchanged trajectory; inside a solve such a step counts towards dJ_counter_limit. */
int to_forward(to_handle* h, int32_t* ls_index /* [B], -1 = failed */, double* J_new /* [B] */);
int to_ilqr_solve(to_handle* h, to_solve_stats* stats);
int to_al_solve(to_handle* h, to_solve_stats* stats);
/* Altro's ProjectedNewtonSolver on the CURRENT trajectory (X, U) of every trajectory of the batch: Newton steps on the active
 * constraints — dynamics defects x_{k+1} (-) f(x_k, u_k), the initial condition, equality rows, inequality rows within
 * active_set_tolerance_pn of their bound, second-order cones through the scalar row |v| - s — in the metric of the diagonal
 * cost Hessian: dZ = -H^-1 D'(D H^-1 D' + rho I)^-1 d in error-state coordinates, the block-tridiagonal S factorised knot by
 * knot, with Altro's refinement / line-search / convergence-rate loops.  Status: TO_SOLVE_SUCCEEDED when the violation
 * (defects included) ends <= constraint_tolerance, TO_PROJECTION_FAIL otherwise.  stats->iterations stay 0. */
int to_pn_solve(to_handle* h, to_solve_stats* stats);
/* ALTRO (Altro.jl solve!(::ALTROSolver)): AL-iLQR with constraint_tolerance := projected_newton_tolerance, then the polish on
 * every trajectory the AL stage left SOLVE_SUCCEEDED with c_max > constraint_tolerance.  With projected_newton = 0 or no
 * constraints: to_al_solve. */
int to_altro_solve(to_handle* h, to_solve_stats* stats);
/* Asynchronous variants: enqueue the solve on the handle's stream from a worker thread owned by the handle and
 * return at once; to_solve_wait blocks until it is done and returns its code (stats are valid after that).  One solve in flight
 * per handle; any other call on the handle while one is in flight fails with TO_ERR_ARGUMENT. */
int to_ilqr_solve_async(to_handle* h, to_solve_stats* stats);
int to_al_solve_async(to_handle* h, to_solve_stats* stats);
int to_altro_solve_async(to_handle* h, to_solve_stats* stats);
int to_solve_wait(to_handle* h);
/* Pipelined solves (round 6).  A solve is batch-synchronous and its batch drains unevenly: C3 spends 52 of its 141 batch steps on a
 * handful of stragglers with the chip empty.  Trajectories are independent (one Z per problem, src/problem.jl:330-340: no cross
 * terms), so a host that has MORE work — the next MPC batch, the next shard of a sweep — starts it on a second handle while the
 * first drains; the two streams share the device.  to_solve_progress reports what the solve loop of the solve in flight last saw:
 * *active = trajectories still iterating (B right after to_*_solve_async, 0 once the iLQR / AL stage has ended — a polish may still
 * be running: to_solve_wait is the completion), *batch_steps = batch steps whose counters have been read, *in_flight = 1 until
 * to_solve_wait has returned.  to_solve_wait_below blocks until *active <= active_max (at once when nothing is in flight).  Both may
 * be called while a solve is in flight (they are the only ones besides the pure descriptor getters); any pointer may be NULL. */
int to_solve_progress(to_handle* h, int32_t* active, int32_t* batch_steps, int32_t* in_flight);
/* Note for hosts that pipeline: a handle that is no longer needed should be destroyed — HIP multiplexes streams onto a few hardware
 * queues, and idle handles' streams take slots that concurrent solves then have to share (measured: three pipelined C5 solves 1.24
 * instead of 1.40 M it/s next to three stale handles). */
int to_solve_wait_below(to_handle* h, int32_t active_max);

/* expansion / gain getters (parity + solver introspection); host layouts column-major:
 *   A[ne,ne,N-1,B]  Bm[ne,m,N-1,B]  Qxx[ne,ne,N,B] Quu[m,m,N,B] Qux[m,ne,N,B] qx[ne,N,B] qu[m,N,B]
 *   K[m,ne,N-1,B]   d[m,N-1,B]  dV[2,B]  rho[B]                                       */
int to_get_dynamics_jacobians(to_handle* h, double* A, double* Bm);
int to_get_cost_expansion(to_handle* h, double* Qxx, double* Quu, double* Qux, double* qx, double* qu);
int to_get_gains(to_handle* h, double* K, double* d, double* dV, double* rho);
/* Cost-to-go of the last backward pass's recursion, recomputed from the stored expansion and gains:
 *   S[ne,ne,N,B]  s[ne,N,B]   S_N = Qxx_N, s_N = qx_N;  S_k = Qxx + K'Quu K + K'Qux + Qux'K,  s_k = qx + K'Quu d + K'qu + Qux'd
 * with the Q-function blocks of knot k (to_get_cost_expansion holds l_xx etc.; Q = l + [A B]'S_{k+1}[A B]).  Needs to_expand + to_backward
 * (phase API) on a handle whose backward pass keeps its expansion in memory (not inside a fused solve). */
int to_get_cost_to_go(to_handle* h, double* S, double* s);

/* raw (non error-state) per-knot cost derivatives of the objective at the current trajectory
 * (RD.gradient! / RD.hessian!): grad[(n+m),N,B], hess[(n+m),(n+m),N,B] */
int to_cost_expansion(to_handle* h, double* grad, double* hess);
/* raw RK Jacobian [A B] of the discrete dynamics (RD.jacobian! on DiscretizedDynamics): F[n,(n+m),N-1,B] */
int to_discrete_jacobian(to_handle* h, double* F);

/* ---- measurement: per-kernel device time of the solve loop, from hipEvents recorded on the handle's stream ----
 * slots: 0 expansion, 1 backward pass, 2 forward pass (line-search rounds, selection / state machine, accept copy, AL
 * outer update), 3 projected-Newton polish (to_altro_solve; one "launch" = all its kernels of one solve).  A "launch" is one batch step.  Accumulates over solves until reset.  Enabling it adds four
 * event records per batch step.
 * Tuning knob (environment, read at to_create): TRAJOPT_LS_CANDIDATES = step sizes evaluated concurrently in the first
 * line-search round (default min(16, 1024 / tiles)); results do not depend on it. */
/* Which kernels a solve on this handle runs.  Batch independence: a trajectory's result does not depend on the other trajectories
 * of its batch.  The choices made PER BATCH STEP from the number of active trajectories — line-search wave shape, one- or
 * two-wave forward kernel, candidate states stored or re-rolled, active-list compaction — are between bit-identical kernels (tested).  The choices made ONCE at
 * to_create from the model, the cost / constraint kinds and the batch size B — backward-pass flavour, fused expansion, scan —
 * are between kernels that agree to rounding (gains to 2e-15), so two handles of very different B (1 000 vs
 * 100 000 trajectories) may differ in the last bits; shards of one batch have (near-)equal B and take the same kernels.
 * info[0]: backward pass 0 = cooperative (R lanes per trajectory, LDS), 1 = MFMA (one wave per
 * trajectory), 2 = lane (one lane per trajectory); info[1]: 1 = the expansion is fused into the backward-pass kernel (profile slot
 * 0 is then empty and slot 1 covers both); info[2]: 1 = active-list compaction; info[3]: step sizes tried concurrently in the
 * first line-search round; info[4]: waves per forward-pass workgroup (2: roller + accountant, k_forward2); info[5]: 1 = the backward pass runs as a scan over the
 * horizon (one wave per trajectory, k_scan.h); info[6]: 1 = batch steps that fill the chip store only the controls of the line-search
 * candidates and roll the accepted ones out again (bit-identical states; k_accept_roll); info[7]: bit 0 = the last line-search round
 * is repacked (the trajectories of a wave that are still searching share all its lanes; bit-identical), bit 1 = iLQR solves move
 * the trajectories still iterating into a dense working set as the batch converges (k_repack_*; bit-identical). */
int to_solver_path(const to_handle* h, int32_t* info /* [8] */);
/* Live state / control dimensions per knot, nx[N], nu[N] (RD.dims(models), src/dynamics.jl:15-31: the terminal knot carries the
 * last model's control dimension).  (n, m) on every knot unless the model is a hybrid model vector. */
int to_knot_dims(const to_handle* h, int32_t* nx, int32_t* nu);
#define TO_PROFILE_SLOTS 4
int to_set_profiling(to_handle* h, int enable);
int to_get_profile(to_handle* h, double* kernel_ms /* [TO_PROFILE_SLOTS] */, int64_t* launches /* [TO_PROFILE_SLOTS] */);
int to_reset_profile(to_handle* h);

/* ---- constraints --------------------------------------------------------------------------- */
/* evaluate_constraints! / constraint_jacobians! for constraint `con_id` over its knot range.
 * vals[p, nk, B], jac[p, w, nk, B] with w = n (state constraints: GOAL, CIRCLE, SPHERE, COLLISION, QUATVEC) or n+m (stage constraints),
 * nk = k_last-k_first+1.  jac is fully written (zeros included). */
int to_evaluate_constraints(to_handle* h, int32_t con_id, double* vals);
int to_constraint_jacobians(to_handle* h, int32_t con_id, double* jac);
/* ∇jacobian! (src/abstract_constraint.jl:255-280): H[w,w,nk,B] += sum_r lambda[r,k,b] * Hessian of c_r — ADDS to H like the
 * reference.  lambda[p,nk,B].  Zero for the kinds whose rows are affine (GOAL src/constraints.jl:70-73, BOUND :767-770,
 * LINEAR, NORM in SOC form); closed forms for NORM (quadratic), CIRCLE, SPHERE, COLLISION, QUATVEC.  The solver's own
 * expansion stays Gauss-Newton (it does not use this term). */
int to_constraint_hessians(to_handle* h, int32_t con_id, const double* lambda, double* H);
int to_constraint_info(const to_handle* h, int32_t con_id, int32_t* p, int32_t* width, int32_t* nk, int32_t* sense);
int to_max_violation(to_handle* h, double* c_max /* [B] */);
/* max |x_1 (-) x0|, |x_{k+1} (-) f(x_k, u_k)| over the horizon, per trajectory: exactly 0 for a rollout, the dynamics
 * infeasibility a projected-Newton polish leaves otherwise */
int to_dynamics_defect(to_handle* h, double* defect /* [B] */);
int to_get_duals(to_handle* h, int32_t con_id, double* lambda /* [p,nk,B] */, double* mu /* [B] */);
int to_set_duals(to_handle* h, int32_t con_id, const double* lambda, const double* mu);
int to_reset_duals(to_handle* h);            /* lambda = 0, mu = penalty_initial */
int to_dual_update(to_handle* h);            /* one AL dual + penalty update at the current trajectory */
int to_al_cost(to_handle* h, double* J_al /* [B] objective + AL terms */);

/* ---- cones, batched and stateless (src/cones.jl) --------------------------------------------- */
/* x[dim,count] column-major. px[dim,count]; jac[dim,dim,count]; hess[dim,dim,count] with b[dim,count].
 * status[count] (nullable): SOC branch 0=below 1=in 2=outside (cone_status src/cones.jl:278-291). */
int to_cone_projection(int device, int32_t cone, int32_t dim, int64_t count, const double* x, double* px, int32_t* status);
int to_cone_projection_jacobian(int device, int32_t cone, int32_t dim, int64_t count, const double* x, double* jac);
int to_cone_projection_hessian(int device, int32_t cone, int32_t dim, int64_t count, const double* x, const double* b, double* hess);

#ifdef __cplusplus
}
#endif
#endif /* TRAJOPT_HIP_H */
