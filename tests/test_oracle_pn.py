"""The oracle's projected-Newton polish and ALTRO driver (oracle/oracle_pn.h; Altro.jl's ProjectedNewtonSolver is out of tree).

Pins, none of which needs a GPU:
  * the reference's own ALTRO results — examples/Cartpole.ipynb cells 17-23 (cost, iteration count, control tail, the
    feasibility ALTRO reports) and examples/Quadrotor.ipynb cell 22 (feasibility);
  * an independent dense numpy restatement of the polish (vector-space models) against the oracle's banded C++ version;
  * banded vs dense factorisation inside the oracle itself; the retraction x (+) dx against state_diff;
  * feasibility and the minimum-norm property of one projection step."""
import ctypes as C
import json
import math
import os
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
from trajectoryoptimization_jl_amd import internal as I

G = json.loads((Path(__file__).parent / "golden" / "reference_goldens.json").read_text())
pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def _params(model):
    return (C.c_double * 16)(*(model.params() + [0.0] * (16 - len(model.params()))))


# ------------------------------------------------------------------------------------------------ reference pins
def test_G4_cartpole_altro_reproduces_the_notebook(oracle):
    """examples/Cartpole.ipynb cells 17-23: ALTRO (AL-iLQR + projected Newton), cost_tolerance_intermediate 1e-2, penalty 1 x 10,
    on the stack the notebook was saved with (RK3, stage costs x dt): 40 iterations, J = 1.552558743680986, violation 3.4e-9,
    u_100 = -3.000000000027 (bound active), x_1 moved off x0 by 3e-11 (the polish treats the initial condition as a constraint).
    The oracle's AL stage takes 39 iLQR iterations, then one projection: the same cost to 2e-7, the same control tail to 1e-5."""
    ga = G["G4_cartpole_altro"]
    o = T.SolverOptions(lib=oracle, cost_dt_scaling=1, cost_tolerance_intermediate=1e-2, penalty_scaling=10.0, penalty_initial=1.0)
    prob = configs.cartpole_problem(batch=1, lib=oracle, options=o, constrained=True, integration=T.RK3)
    s = T.ALTROSolver(prob).solve()
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED
    assert s.stats["cost"][0] == pytest.approx(ga["cost"], rel=1e-6)
    assert abs(int(s.stats["iterations"][0]) + int(s.stats["iterations_pn"][0] > 0) - ga["iterations"]) <= 1
    assert s.stats["c_max"][0] < 1e-8                     # the notebook reports 3.4e-9
    U, X = T.controls(prob)[0, :, 0], T.states(prob)[0]
    np.testing.assert_allclose(U[-len(ga["U_tail"]):], ga["U_tail"], atol=2e-5)   # measured: 8e-6
    np.testing.assert_allclose(U[:len(ga["U_head"])], ga["U_head"], atol=2e-4)
    assert np.all(np.abs(U) <= 3.0 + 1e-8)
    np.testing.assert_allclose(X[-1], [0, math.pi, 0, 0], atol=1e-8)
    np.testing.assert_allclose(X[0], 0.0, atol=1e-8)
    # the AL stage alone (no polish) creeps to the same tolerance with 66 iterations and lands elsewhere
    prob2 = configs.cartpole_problem(batch=1, lib=oracle, options=o, constrained=True, integration=T.RK3)
    s2 = T.ALTROSolver(prob2, projected_newton=0).solve()
    assert int(s2.stats["iterations_pn"][0]) == 0 and int(s2.stats["iterations"][0]) > int(s.stats["iterations"][0])


def test_G4_quadrotor_zigzag_altro(oracle):
    """examples/Quadrotor.ipynb cell 22: ALTRO ends with violation 7.6e-10 (J = 0.29928, 90 iterations).  The solve is chaotic in
    its start, so cost and count stay sanity pins (1 %); the feasibility is what the polish is for."""
    g = G["G4_quadrotor_altro"]
    prob, wpts, _ = configs.quadrotor_zigzag_problem(lib=oracle)
    s = T.ALTROSolver(prob).solve()
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED and s.stats["c_max"][0] < 1e-6
    assert s.stats["cost"][0] == pytest.approx(g["cost"], rel=1e-2)
    assert int(s.stats["iterations_pn"][0]) >= 1
    prob, _, _ = configs.quadrotor_zigzag_problem(lib=oracle)
    s = T.ALTROSolver(prob, constraint_tolerance=1e-10).solve()
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED and s.stats["c_max"][0] < 1e-9
    U = T.controls(prob)[0]
    assert U.min() >= -1e-9 and U.max() <= 12.0 + 1e-9
    assert np.linalg.norm(T.states(prob)[0][-1, :3] - wpts[2]) < 5e-3


# ------------------------------------------------------------------------------------------------ building blocks
@pytest.mark.parametrize("model", [T.Quadrotor(), T.Quadrotor(rotation="mrp"), T.Quadrotor(rotation="rp"), T.Cartpole()])
def test_state_add_inverts_state_diff(model, oracle):
    rng = np.random.default_rng(3)
    n, ne = model.n, (12 if model.n >= 12 else model.n)
    for _ in range(20):
        x = rng.normal(size=n) * 0.4
        if n == 13:
            x[3:7] = rng.normal(size=4); x[3:7] /= np.linalg.norm(x[3:7])
        dx = rng.normal(size=ne) * 0.3
        xo, back = np.zeros(n), np.zeros(ne)
        oracle.call("state_add", model.model_id, _params(model), pd(x), pd(dx), pd(xo))
        oracle.call("state_diff", model.model_id, _params(model), pd(xo), pd(x), pd(back))
        np.testing.assert_allclose(back, dx, rtol=1e-12, atol=1e-13)
        if n == 13:
            assert np.linalg.norm(xo[3:7]) == pytest.approx(1.0, abs=1e-14)


def test_defect_of_a_rollout_is_zero(oracle):
    for prob in (configs.cartpole_problem(batch=3, N=31, lib=oracle), configs.quadrotor_problem(batch=2, N=21, tf=1.0, lib=oracle)):
        T.rollout(prob)
        d = np.ones(prob.B)
        prob._call("dynamics_defect", prob._pd(d))
        assert np.all(d == 0.0)


def _perturbed(build, oracle, seed=5, scale=1e-3):
    """An AL solution moved off the constraint manifold: what the polish is handed."""
    prob = build()
    T.ALSolver(prob, constraint_tolerance=1e-3).solve()
    rng = np.random.default_rng(seed)
    X, U = T.states(prob), T.controls(prob)
    T.initial_states(prob, X + scale * rng.normal(size=X.shape))
    T.initial_controls(prob, U + scale * rng.normal(size=U.shape))
    o = T.SolverOptions(lib=oracle)   # options stick to the handle: back to the defaults for what follows
    prob._call("set_options", C.byref(o._o))
    return prob


def test_banded_and_dense_factorisations_agree(oracle):
    build = lambda: configs.cartpole_problem(batch=2, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=oracle)
    res = []
    for dense in (False, True):
        if dense:
            os.environ["ORACLE_PN_DENSE"] = "1"
        try:
            prob = _perturbed(build, oracle)
            s = T.ProjectedNewtonSolver(prob).solve()
        finally:
            os.environ.pop("ORACLE_PN_DENSE", None)
        assert np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED)
        res.append((T.states(prob), T.controls(prob), s.stats["iterations_pn"].copy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=1e-11)
    assert np.array_equal(res[0][2], res[1][2])


# ------------------------------------------------------------------------------------------------ numpy restatement
class NumpyPN:
    """Dense restatement of oracle_pn.h for vector-space models (x (+) dx = x + dx).  Constraint values / Jacobians and the
    dynamics Jacobians come from the oracle's phase API (pinned by the KATs and goldens), the polish itself — active set, D, the
    metric, S, Cholesky + refinement, line search, rate test — is written here from the header's description."""

    def __init__(self, prob, oracle, model, hdiag):
        self.p, self.o, self.model, self.h = prob, oracle, model, hdiag  # hdiag [N, n+m]: diagonal of the objective Hessian
        self.opts = T.SolverOptions(lib=oracle)
        prob._call("get_options", C.byref(self.opts._o))
        self.x0 = np.zeros((prob.B, prob.n)); prob._call("get_initial_state", prob._pd(self.x0))
        self.dt = np.diff(prob.gettimes())

    def step(self, x, u, k):
        xn = np.zeros(self.p.n)
        self.o.call("discrete_dynamics", self.model.model_id, _params(self.model), self.p.integration, pd(np.ascontiguousarray(x)),
                    pd(np.ascontiguousarray(u)), float(self.dt[k]), pd(xn))
        return xn

    def candidates(self, X, U, jac):
        """per knot: list of (value, gradient over [x; u], is_equality)"""
        p = self.p
        T.initial_states(p, X[None]); T.initial_controls(p, U[None])
        rows = [[] for _ in range(p.N)]
        for i, con in enumerate(p.constraints):
            a, b = p.constraints.inds[i]
            vals = T.evaluate_constraints(p, i)[0]
            J = T.constraint_jacobians(p, i)[0] if jac else None
            for kk, k in enumerate(range(a - 1, b)):
                if J is not None:
                    Jk = np.zeros((con.p, p.n + p.m)); Jk[:, :J.shape[2]] = J[kk]
                if isinstance(T.sense(con), T.SecondOrderCone):
                    v, s_ = vals[kk, :-1], vals[kk, -1]
                    a_ = np.linalg.norm(v)
                    g = (v / a_) @ Jk[:-1] - Jk[-1] if (jac and a_ > 0) else np.zeros(p.n + p.m)
                    rows[k].append((a_ - s_, g, False))
                else:
                    eq = isinstance(T.sense(con), T.ZeroCone)
                    for r in range(con.p):
                        rows[k].append((vals[kk, r], Jk[r] if jac else None, eq))
        return rows

    def residual(self, X, U, mask=None):
        p = self.p
        rows = self.candidates(X, U, jac=mask is None)
        if mask is None:
            mask = []
            for k in range(p.N):
                mk = []
                for (v, g, eq) in rows[k]:
                    gg = g[:p.n] if k == p.N - 1 else g
                    mk.append(bool((eq or v >= -self.opts.active_set_tolerance_pn) and gg @ gg > 0))
                mask.append(mk)
        d = []
        for k in range(p.N):
            e = X[0] - self.x0[0] if k == 0 else self.step(X[k - 1], U[k - 1], k - 1) - X[k]
            d.append(np.concatenate([e, [rows[k][q][0] for q in range(len(rows[k])) if mask[k][q]]]))
        return d, mask, rows

    def solve(self, X, U):
        p, o = self.p, self.opts
        n, m, N = p.n, p.m, p.N
        nc = n + m
        nv = (N - 1) * nc + n
        steps = 0
        for step in range(100):
            d, mask, rows = self.residual(X, U)
            viol = max(np.abs(np.concatenate(d)).max(), 0)
            if viol <= o.constraint_tolerance or step > o.n_steps:
                break
            steps += 1
            # linearise
            T.initial_states(p, X[None]); T.initial_controls(p, U[None])
            I.expand(p)
            A, B = I.dynamics_jacobians(p)
            W = np.zeros(nv)
            for k in range(N):
                w = 1.0 / (np.maximum(self.h[k], 0) + o.rho_primal)
                W[k * nc:k * nc + (nc if k < N - 1 else n)] = w[:nc if k < N - 1 else n]
            Drows = []
            for k in range(N):
                for i in range(n):
                    r = np.zeros(nv)
                    if k == 0:
                        r[i] = 1.0
                    else:
                        r[(k - 1) * nc:(k - 1) * nc + n] = A[0, k - 1, i]; r[(k - 1) * nc + n:k * nc] = B[0, k - 1, i]; r[k * nc + i] = -1.0
                    Drows.append(r)
                for q, (v, g, eq) in enumerate(rows[k]):
                    if mask[k][q]:
                        r = np.zeros(nv); w = nc if k < N - 1 else n
                        r[k * nc:k * nc + w] = g[:w]
                        Drows.append(r)
            D = np.array(Drows)
            S = (D * W) @ D.T
            L = np.linalg.cholesky(S + o.rho_chol * np.eye(len(S)))
            chol = lambda b: np.linalg.solve(L.T, np.linalg.solve(L, b))
            dv = np.concatenate(d)
            prev = viol
            for count in range(10):
                x = chol(dv)
                for it in range(25):
                    r = dv - S @ x
                    if np.linalg.norm(r) < 1e-8:
                        break
                    x = x + chol(r)
                dZ = -W * (D.T @ x)
                alpha, ok = 1.0, False
                for ls in range(10):
                    Xb, Ub = X.copy(), U.copy()
                    for k in range(N):
                        Xb[k] += alpha * dZ[k * nc:k * nc + n]
                        if k < N - 1:
                            Ub[k] += alpha * dZ[k * nc + n:(k + 1) * nc]
                    dn, _, _ = self.residual(Xb, Ub, mask)
                    v = np.abs(np.concatenate(dn)).max()
                    if v < prev:
                        ok = True
                        break
                    alpha *= 0.5
                if not ok:
                    break
                X, U, dv = Xb, Ub, np.concatenate(dn)
                before, prev = prev, v
                if v < o.constraint_tolerance:
                    break
                if before < 1.0:
                    if math.log10(v) / math.log10(before) < o.r_threshold:
                        break
                elif not v < 0.5 * before:
                    break
        return X, U, steps


def _cartpole_hdiag(N):
    h = np.zeros((N, 5)); h[:, :4] = 1e-2; h[:, 4] = 1e-1; h[-1, :4] = 100.0
    return h


def _quickstart_hdiag(N):
    h = np.ones((N, 6)); h[-1, :4] = N - 1
    return h


@pytest.mark.parametrize("name", ["cartpole_con", "quickstart"])
def test_polish_against_a_dense_numpy_restatement(name, oracle):
    if name == "cartpole_con":
        build = lambda: configs.cartpole_problem(batch=1, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=oracle)
        model, hd = T.Cartpole(), _cartpole_hdiag(41)
    else:
        def build():  # U0 = 0 starts on the symmetry axis of the obstacle (a saddle the AL stage never leaves): nudge it
            p = configs.quickstart_problem(batch=1, lib=oracle)
            T.initial_controls(p, np.array([0.1, 0.0]))
            return p
        model, hd = T.DoubleIntegrator(1.0, 2), _quickstart_hdiag(21)
    prob = _perturbed(build, oracle, scale=2e-3)
    X0, U0 = T.states(prob)[0].copy(), T.controls(prob)[0].copy()
    s = T.ProjectedNewtonSolver(prob).solve()
    Xo, Uo = T.states(prob)[0].copy(), T.controls(prob)[0].copy()
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED and s.stats["c_max"][0] <= 1e-6
    assert int(s.stats["iterations"][0]) == 0
    ref = NumpyPN(build(), oracle, model, hd)
    Xn, Un, steps = ref.solve(X0.copy(), U0.copy())
    assert steps == int(s.stats["iterations_pn"][0])
    np.testing.assert_allclose(Xo, Xn, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Uo, Un, rtol=0, atol=1e-9)
    # the polish moved the trajectory by about the perturbation, not more (a projection, not a re-solve)
    assert np.abs(Xo - X0).max() < 0.1 and np.abs(Uo - U0).max() < 0.5


def test_one_projection_is_the_minimum_norm_step(oracle):
    """With LINEAR dynamics and constraints (double integrator, goal + control bounds) one Newton step lands on the active
    manifold exactly and is the H-weighted least-norm step onto it: any other feasible point is farther from the start."""
    model = T.DoubleIntegrator(1.0, 2)
    n, m, N = 4, 2, 15
    xf = np.array([1.0, 2.0, 0.0, 0.0])
    def build():
        obj = T.LQRObjective(np.array([1.0, 2.0, 3.0, 4.0]), np.array([0.5, 0.25]), np.full(n, 10.0), xf, N)
        cons = T.ConstraintList(n, m, N)
        T.add_constraint(cons, T.GoalConstraint(xf), N)
        return T.Problem(model, obj, np.zeros(n), 2.0, xf=xf, constraints=cons, batch=1, lib=oracle)
    prob = build()
    rng = np.random.default_rng(11)
    T.initial_controls(prob, rng.normal(size=(1, N - 1, m)))
    T.rollout(prob)
    X0 = T.states(prob)[0] + 0.05 * rng.normal(size=(N, n))
    T.initial_states(prob, X0[None])
    U0 = T.controls(prob)[0].copy()
    s = T.ProjectedNewtonSolver(prob, constraint_tolerance=1e-10, rho_chol=1e-12).solve()
    assert int(s.stats["iterations_pn"][0]) == 1 and s.stats["c_max"][0] < 1e-10
    X1, U1 = T.states(prob)[0], T.controls(prob)[0]
    hx, hu, hf = np.array([1.0, 2.0, 3.0, 4.0]), np.array([0.5, 0.25]), np.full(n, 10.0)
    def dist(X, U):
        dX, dU = X - X0, U - U0
        return sum(dX[k] @ ((hf if k == N - 1 else hx) * dX[k]) for k in range(N)) + sum(dU[k] @ (hu * dU[k]) for k in range(N - 1))
    base = dist(X1, U1)
    # other feasible trajectories: roll out perturbed controls from x0, then fix the goal with a second polish
    for trial in range(5):
        p2 = build()
        T.initial_controls(p2, (U1 + 0.1 * rng.normal(size=U1.shape))[None]); T.rollout(p2)
        T.ProjectedNewtonSolver(p2, constraint_tolerance=1e-10, rho_chol=1e-12).solve()
        assert dist(T.states(p2)[0], T.controls(p2)[0]) > base


def test_altro_small_quadrotor_batch_converges(oracle):
    """C5's shape at a reduced horizon: the polish takes every trajectory the AL stage hands over to 1e-6."""
    build = lambda: configs.quadrotor_problem(batch=8, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=oracle)
    prob = build()
    s = T.ALTROSolver(prob).solve()
    assert np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED)
    assert s.stats["c_max"].max() < 1e-6
    d = np.zeros(8); prob._call("dynamics_defect", prob._pd(d))
    assert d.max() < 1e-6 and T.max_violation(prob).max() < 1e-6
    assert np.all(s.stats["iterations_pn"] >= 1)
    sa = T.ALSolver(build()).solve()   # the AL stage alone: 30 outer iterations on half of them, twice the iLQR iterations
    assert np.sum(sa.stats["status"] == T.capi.SOLVE_SUCCEEDED) < 8
    assert s.stats["iterations"].sum() < sa.stats["iterations"].sum()


def test_polish_of_an_unconstrained_problem_and_of_a_minimal_horizon(oracle):
    """Without a constraint list the active set is the initial condition and the dynamics defects alone: a rollout is left untouched
    (0 projections), a perturbed trajectory is projected back onto the dynamics.  N = 3: the smallest horizon with an interior knot."""
    for N, tf in ((31, 1.5), (3, 0.1)):
        prob = configs.cartpole_problem(batch=2, N=N, tf=tf, lib=oracle)
        T.rollout(prob)
        X0 = T.states(prob).copy()
        s = T.ProjectedNewtonSolver(prob).solve()
        assert np.all(s.stats["iterations_pn"] == 0) and np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED)
        np.testing.assert_array_equal(T.states(prob), X0)
        rng = np.random.default_rng(N)
        T.initial_states(prob, X0 + 1e-3 * rng.normal(size=X0.shape))
        assert T.dynamics_defect(prob).min() > 1e-5
        s = T.ProjectedNewtonSolver(prob).solve()
        assert np.all(s.stats["iterations_pn"] >= 1) and np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED)
        assert T.dynamics_defect(prob).max() <= 1e-6 and s.stats["c_max"].max() <= 1e-6
        np.testing.assert_allclose(T.states(prob)[:, 0], prob.x0 if hasattr(prob, "x0") else X0[:, 0], atol=1e-6)
