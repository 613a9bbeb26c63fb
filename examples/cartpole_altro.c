/* A plain-C host of the drop-in boundary (include/trajopt_hip.h): the reference's constrained Cartpole swing-up
 * (examples/Cartpole.ipynb cells 3-17: N = 101, tf = 5, RK3, Q = 1e-2 I, R = 1e-1, Qf = 100 I, xf = [0, pi, 0, 0], |u| <= 3, goal at N,
 * U0 = 0.01, the notebook's solver options) for a batch of B trajectories with perturbed initial states, solved as the notebook solves it — ALTRO = AL-iLQR + the
 * projected-Newton polish — through nothing but the C-ABI.  What a Julia / C / Fortran host does, without the Python mirror:
 *
 *   gcc -I include examples/cartpole_altro.c -L trajectoryoptimization.jl_amd/csrc -ltrajopt_hip -lm -o cartpole_altro
 *   LD_LIBRARY_PATH=trajectoryoptimization.jl_amd/csrc ./cartpole_altro [B]
 *
 * Prints one line per trajectory class and the notebook's reference value (trajectory 0 starts at x0 = 0: J = 1.5525587).
 * tests/test_c_host_example.py compiles and links it on every CPU run and runs it on the GPU box. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "trajopt_hip.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != TO_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, to_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  const int n = 4, m = 1, N = 101, B = argc > 1 ? atoi(argv[1]) : 8;
  const double pi = 3.14159265358979323846, xf[4] = {0.0, pi, 0.0, 0.0};

  /* Objective(stage, terminal, N) = LQRObjective(Q, R, Qf, xf, N)  (src/objective.jl:137-183) */
  to_cost_desc costs[2];
  memset(costs, 0, sizeof(costs));
  for (int t = 0; t < 2; ++t) {
    const double qd = t == 0 ? 1e-2 : 100.0;
    costs[t].kind = TO_COST_DIAGONAL;
    costs[t].terminal = t;
    for (int i = 0; i < n; ++i) { costs[t].Q[i] = qd; costs[t].q[i] = -qd * xf[i]; costs[t].c += 0.5 * qd * xf[i] * xf[i]; }
    costs[t].R[0] = 1e-1;
  }
  /* ConstraintList: BoundConstraint(|u| <= 3) at 1:N-1, GoalConstraint(xf) at N */
  to_constraint_desc cons[2];
  memset(cons, 0, sizeof(cons));
  cons[0].kind = TO_CON_BOUND; cons[0].sense = TO_CONE_NEGATIVE_ORTHANT; cons[0].k_first = 1; cons[0].k_last = N - 1;
  cons[0].n_params = 2 * (n + m);
  for (int i = 0; i < n + m; ++i) { cons[0].params[i] = i < n ? INFINITY : 3.0; cons[0].params[n + m + i] = i < n ? -INFINITY : -3.0; }
  cons[1].kind = TO_CON_GOAL; cons[1].sense = TO_CONE_ZERO; cons[1].k_first = N; cons[1].k_last = N;
  cons[1].n_inds = n; cons[1].n_params = n;
  for (int i = 0; i < n; ++i) { cons[1].inds[i] = i + 1; cons[1].params[i] = xf[i]; }

  to_problem_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.abi_version = TO_ABI_VERSION;
  desc.model = TO_MODEL_CARTPOLE; desc.integrator = TO_RK3;  /* the notebook's TrajectoryOptimization version: RK3, costs scaled by dt */
  desc.n = n; desc.m = m; desc.N = N; desc.B = B;
  desc.model_params[0] = 1.0; desc.model_params[1] = 0.2; desc.model_params[2] = 0.5; desc.model_params[3] = 9.81;  /* mc, mp, l, g */
  desc.t0 = 0.0; desc.tf = 5.0;
  desc.n_costs = 2; desc.costs = costs;
  desc.n_constraints = 2; desc.constraints = cons;

  if (to_abi_version() != TO_ABI_VERSION) { fprintf(stderr, "library ABI %d, header %d\n", to_abi_version(), TO_ABI_VERSION); return 1; }
  to_solver_opts opts;
  CHECK(to_default_options(&opts));
  opts.cost_dt_scaling = 1;                 /* legacy stage costs (NEWS.md:11-12) */
  opts.cost_tolerance_intermediate = 1e-2;  /* examples/Cartpole.ipynb cell 17: SolverOptions(cost_tolerance_intermediate = 1e-2, */
  opts.penalty_scaling = 10.0;              /*   penalty_scaling = 10., penalty_initial = 1.0) */
  opts.penalty_initial = 1.0;
  to_handle* h = NULL;
  CHECK(to_create(&desc, &opts, 0, &h));

  /* initial states (n, B) column-major: trajectory 0 is the notebook's x0 = 0, the others start off it */
  double* x0 = calloc((size_t)n * B, sizeof(double));
  for (int b = 1; b < B; ++b) { x0[b * n + 0] = 0.4 * sin(1.7 * b); x0[b * n + 1] = 0.25 * cos(2.3 * b); }
  CHECK(to_set_initial_state(h, x0));
  const double u0 = 0.01;
  CHECK(to_set_controls_uniform(h, &u0));

  to_solve_stats st;
  memset(&st, 0, sizeof(st));
  st.iterations = malloc(sizeof(int32_t) * B); st.iterations_outer = malloc(sizeof(int32_t) * B); st.status = malloc(sizeof(int32_t) * B);
  st.iterations_pn = malloc(sizeof(int32_t) * B); st.cost = malloc(sizeof(double) * B); st.c_max = malloc(sizeof(double) * B);
  CHECK(to_altro_solve(h, &st));

  double* X = malloc(sizeof(double) * n * N * B);
  CHECK(to_get_states(h, X));
  int converged = 0;
  for (int b = 0; b < B; ++b) converged += st.status[b] == TO_SOLVE_SUCCEEDED && st.c_max[b] <= opts.constraint_tolerance;
  printf("build %s  B=%d  converged=%d  iterations=%lld  batch_steps=%d  solve_ms=%.3f\n", to_build_id(), B, converged,
         (long long)st.total_iterations, st.batch_steps, st.solve_ms);
  for (int b = 0; b < B && b < 4; ++b) {
    const double* xN = X + ((size_t)b * N + (N - 1)) * n;
    printf("trajectory %d: iLQR %d outer %d projections %d status %d J=%.9f c_max=%.2e x_N=[%.6f %.6f %.6f %.6f]\n", b, st.iterations[b],
           st.iterations_outer[b], st.iterations_pn[b], st.status[b], st.cost[b], st.c_max[b], xN[0], xN[1], xN[2], xN[3]);
  }
  printf("reference (examples/Cartpole.ipynb cell 19, ALTRO): J = 1.552558743680986\n");
  CHECK(to_destroy(h));
  return converged == B ? 0 : 2;
}
