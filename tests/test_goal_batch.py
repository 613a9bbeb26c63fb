"""One goal per trajectory on one handle (SURVEY.md §8b "optionally per-trajectory xf"; VERDICT r04 missing #2): set_goal_state!(prob, Xf)
= set_LQR_goal!(cost, xf_b) for every trajectory (src/problem.jl:294-310, src/cost_functions.jl:249-258: only q changes), through
to_set_cost_linear_batch.  CPU: the oracle's batch against B single-trajectory oracle problems retargeted with the scalar
set_goal_state!.  GPU: the HIP path against the oracle."""
import numpy as np
import pytest

import trajopt_amd as T
from trajopt_amd import internal as I
from trajectoryoptimization_jl_amd import configs


def cartpole_goals(B, seed=5):
    rng = np.random.default_rng(seed)
    Xf = np.tile(np.array([0.0, np.pi, 0.0, 0.0]), (B, 1))
    Xf[:, 0] = rng.uniform(-1.0, 1.0, B)          # where the cart should stop
    Xf[0, 0] = 0.0
    return Xf


def test_per_trajectory_goals_equal_single_trajectory_problems(oracle):
    B = 6
    Xf = cartpole_goals(B)
    pb = configs.cartpole_problem(batch=B, lib=oracle)
    x0 = np.empty((B, 4)); pb._call("get_initial_state", pb._pd(x0))
    T.set_goal_state(pb, Xf)
    T.rollout(pb)
    Jb, Jkb = T.cost(pb), T.stage_costs(pb)
    I.expand(pb); I.backwardpass(pb)
    gb = I.gains(pb)
    sb = T.iLQRSolver(pb).solve()
    Xb, Ub = T.states(pb), T.controls(pb)
    for b in range(B):
        p1 = configs.cartpole_problem(batch=1, lib=oracle)
        p1.set_initial_state(x0[b])
        T.set_goal_state(p1, Xf[b])                # the reference's scalar verb
        T.rollout(p1)
        np.testing.assert_allclose(T.cost(p1)[0], Jb[b], rtol=1e-13)
        np.testing.assert_allclose(T.stage_costs(p1)[0], Jkb[b], rtol=1e-12, atol=1e-13)
        I.expand(p1); I.backwardpass(p1)
        g1 = I.gains(p1)
        np.testing.assert_allclose(g1["d"][0], gb["d"][b], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(g1["K"][0], gb["K"][b], rtol=1e-9, atol=1e-11)
        s1 = T.iLQRSolver(p1).solve()
        assert int(s1.stats["iterations"][0]) == int(sb.stats["iterations"][b]) and int(s1.stats["status"][0]) == int(sb.stats["status"][b])
        np.testing.assert_allclose(T.states(p1)[0], Xb[b], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(s1.stats["cost"][0], sb.stats["cost"][b], rtol=1e-8)
    assert np.abs(Xb[:, -1, 0] - Xf[:, 0]).max() < 0.05 and np.ptp(Xb[:, -1, 0]) > 0.5       # every cart stops at ITS goal
    # shared descriptors again
    T.clear_goal_state_batch(pb)
    ref = configs.cartpole_problem(batch=B, lib=oracle)
    T.initial_controls(pb, T.controls(ref)); T.rollout(pb); T.rollout(ref)
    np.testing.assert_array_equal(T.cost(pb), T.cost(ref))


def test_per_trajectory_goal_checks(oracle):
    p = configs.cartpole_problem(batch=3, constrained=True, lib=oracle)
    with pytest.raises(T.DimensionMismatch):
        T.set_goal_state(p, np.zeros((2, 4)))
    T.set_goal_state(p, np.zeros((3, 4)))                      # objective and GoalConstraint, one target per trajectory
    T.set_goal_state(p, np.zeros((3, 4)), constraint=False)
    with pytest.raises(T.UnsupportedError, match="GoalConstraint .its target. and LinearConstraint"):
        p._call("set_constraint_params_batch", 0, p._pd(np.zeros((3, 1))))      # constraint 0 is the control bound
    with pytest.raises(T.ArgumentError):
        p._call("set_constraint_params_batch", 7, p._pd(np.zeros((3, 4))))


def test_per_trajectory_goal_constraints_equal_single_trajectory_problems(oracle):
    """set_goal_state!(prob, xf; constraint = true) retargets the GoalConstraint as well (src/problem.jl:303-309, src/constraints.jl:22-87).
    With one goal per trajectory (to_set_constraint_params_batch) every trajectory of the batch must behave exactly like a
    single-trajectory problem retargeted with the reference's scalar verb: constraint values and Jacobians, violation, AL cost, the
    expansion's gains, whole AL and ALTRO solves (integers equal, the cart at ITS goal to the constraint tolerance)."""
    B = 5
    Xf = cartpole_goals(B, seed=11)
    Xf[:, 0] *= 0.5
    pb = configs.cartpole_problem(batch=B, constrained=True, lib=oracle)
    x0 = np.empty((B, 4)); pb._call("get_initial_state", pb._pd(x0))
    T.set_goal_state(pb, Xf)
    T.rollout(pb)
    I.dual_update(pb); I.dual_update(pb)
    gi = len(pb.constraints) - 1                               # the GoalConstraint
    cb, jb = T.evaluate_constraints(pb, gi), T.constraint_jacobians(pb, gi)
    vb, ab = T.max_violation(pb), I.al_cost(pb)
    I.expand(pb); I.backwardpass(pb)
    gb = I.gains(pb)
    singles = []
    for b in range(B):
        p1 = configs.cartpole_problem(batch=1, constrained=True, lib=oracle)
        p1.set_initial_state(x0[b])
        T.set_goal_state(p1, Xf[b])                            # the reference's scalar verb: objective + GoalConstraint
        T.rollout(p1)
        I.dual_update(p1); I.dual_update(p1)
        np.testing.assert_allclose(T.evaluate_constraints(p1, gi)[0], cb[b], rtol=1e-13, atol=1e-14)
        np.testing.assert_array_equal(T.constraint_jacobians(p1, gi)[0], jb[b])
        np.testing.assert_allclose(T.max_violation(p1)[0], vb[b], rtol=1e-13)
        np.testing.assert_allclose(I.al_cost(p1)[0], ab[b], rtol=1e-12)
        I.expand(p1); I.backwardpass(p1)
        g1 = I.gains(p1)
        np.testing.assert_allclose(g1["d"][0], gb["d"][b], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(g1["K"][0], gb["K"][b], rtol=1e-9, atol=1e-10)
        singles.append(p1)
    assert np.ptp(cb[:, 0, 0]) > 0.1                           # the targets do differ
    for Solver in (T.ALSolver, T.ALTROSolver):
        for p in [pb] + singles:
            T.initial_controls(p, np.full(1, 0.01)); I.reset_duals(p)
        sb = Solver(pb).solve()
        Xb = T.states(pb)
        for b, p1 in enumerate(singles):
            s1 = Solver(p1).solve()
            for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
                assert int(s1.stats[k][0]) == int(sb.stats[k][b]), (Solver.__name__, k, b)
            np.testing.assert_allclose(T.states(p1)[0], Xb[b], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(s1.stats["c_max"][0], sb.stats["c_max"][b], rtol=1e-5, atol=1e-12)
        assert (sb.stats["status"] == T.capi.SOLVE_SUCCEEDED).all() and sb.stats["c_max"].max() < 1e-6
        assert np.abs(Xb[:, -1, :] - Xf).max() < 1e-5          # every cart AT its own goal, to the constraint tolerance
    # to_set_constraint returns a constraint to shared parameters; the scalar verb does exactly that
    T.set_goal_state(pb, Xf[0])
    T.rollout(pb)
    c0 = T.evaluate_constraints(pb, gi)
    np.testing.assert_allclose(c0[:, 0, :], T.states(pb)[:, -1, :] - Xf[0][None, :], rtol=0, atol=1e-14)


def linear_problem(lib, B, bvals=None):
    """2-D double integrator to a goal with a half-plane keep-out  a'p <= b  on every stage knot and a coupled control row
    u1 + 2 u2 <= c  (one LinearConstraint with two rows over [x; u]); optionally one right-hand side per trajectory."""
    model = T.DoubleIntegrator(1.0, 2)
    n, m, N = 4, 2, 31
    xf = np.array([1.0, 2.0, 0.0, 0.0])
    obj = T.LQRObjective(np.ones(n), 0.1 * np.ones(m), 100 * np.ones(n), xf, N)
    cons = T.ConstraintList(n, m, N)
    A = np.array([[1.0, -0.5, 0.0, 0.0], [0.0, 0.0, 1.0, 2.0]])         # over z[inds], inds = (x1, x2, u1, u2)
    T.add_constraint(cons, T.LinearConstraint(n, m, A, np.array([0.6, 3.0]), T.Inequality(), inds=[1, 2, 5, 6]), (1, N - 1))
    T.add_constraint(cons, T.GoalConstraint(xf), N)
    p = T.Problem(model, obj, np.zeros(n), 3.0, xf=xf, constraints=cons, batch=B, lib=lib)
    x0 = np.zeros((B, n)); x0[:, :2] = np.random.default_rng(3).uniform(-0.2, 0.2, (B, 2)); x0[0] = 0
    p.set_initial_state(x0)
    if bvals is not None:
        T.set_constraint_params_batch(p, 0, bvals)
    return p, x0


def linear_rhs(B, seed=2):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(0.45, 0.9, B), rng.uniform(2.4, 3.6, B)], axis=1)      # [B, p]


def test_per_trajectory_linear_rhs_equals_single_trajectory_problems(oracle):
    """One right-hand side b per trajectory for a LinearConstraint (to_set_constraint_params_batch; src/constraints.jl:103-150): every trajectory
    of the oracle's batch against a single-trajectory problem BUILT with that b — values, Jacobians, violation, AL cost, gains, AL solves."""
    B = 5
    bv = linear_rhs(B)
    pb, x0 = linear_problem(oracle, B, bv)
    T.initial_controls(pb, np.array([0.3, 0.5])); T.rollout(pb)
    I.dual_update(pb); I.dual_update(pb)
    cb, jb, vb, ab = T.evaluate_constraints(pb, 0), T.constraint_jacobians(pb, 0), T.max_violation(pb), I.al_cost(pb)
    I.expand(pb); I.backwardpass(pb)
    gb = I.gains(pb)
    sb = T.ALSolver(pb).solve()
    Xb = T.states(pb)
    for b in range(B):
        p1, _ = linear_problem(oracle, 1)
        con = p1.constraints.constraints[0]
        con.b = bv[b].copy()                                   # the descriptor itself carries this trajectory's b
        d = con._desc(*p1.constraints.inds[0])
        p1._call("set_constraint", 0, __import__("ctypes").byref(d))
        p1.set_initial_state(x0[b])
        T.initial_controls(p1, np.array([0.3, 0.5])); T.rollout(p1)
        I.dual_update(p1); I.dual_update(p1)
        np.testing.assert_allclose(T.evaluate_constraints(p1, 0)[0], cb[b], rtol=1e-13, atol=1e-14)
        np.testing.assert_array_equal(T.constraint_jacobians(p1, 0)[0], jb[b])
        np.testing.assert_allclose(T.max_violation(p1)[0], vb[b], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(I.al_cost(p1)[0], ab[b], rtol=1e-12)
        I.expand(p1); I.backwardpass(p1)
        g1 = I.gains(p1)
        np.testing.assert_allclose(g1["K"][0], gb["K"][b], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(g1["d"][0], gb["d"][b], rtol=1e-9, atol=1e-10)
        s1 = T.ALSolver(p1).solve()
        for k in ("iterations", "iterations_outer", "status"):
            assert int(s1.stats[k][0]) == int(sb.stats[k][b]), (k, b)
        np.testing.assert_allclose(T.states(p1)[0], Xb[b], rtol=1e-6, atol=1e-7)
    assert (sb.stats["status"] == T.capi.SOLVE_SUCCEEDED).all() and np.ptp(cb[:, 0, 0]) > 0.05

@pytest.mark.gpu
@pytest.mark.parametrize("path", ["default", "lane", "fwd1"])
def test_per_trajectory_goals_cartpole_on_gpu(path, hip, oracle, monkeypatch):
    """HIP against the oracle with one goal per trajectory: every phase, then whole iLQR solves (integers bit-exact) — on the scan /
    cooperative path, on the fused lane path with compaction and accept-by-rollout, and with the one-wave forward kernel."""
    from test_gpu_parity import assert_solve_parity
    B = 96
    if path == "lane":
        monkeypatch.setenv("TRAJOPT_BACKWARD", "lane"); monkeypatch.setenv("TRAJOPT_ACCEPT_ROLL_MIN", "1"); B = 300
    if path == "fwd1":
        monkeypatch.setenv("TRAJOPT_FWD2", "0")
    Xf = cartpole_goals(B)
    ph, po = configs.cartpole_problem(batch=B, lib=hip), configs.cartpole_problem(batch=B, lib=oracle)
    for p in (ph, po):
        T.set_goal_state(p, Xf); T.rollout(p)
    np.testing.assert_allclose(T.cost(ph), T.cost(po), rtol=1e-12)
    np.testing.assert_allclose(T.stage_costs(ph), T.stage_costs(po), rtol=1e-12, atol=1e-14)
    for p in (ph, po):
        I.expand(p); I.backwardpass(p)
    gh, go = I.gains(ph), I.gains(po)
    np.testing.assert_allclose(gh["K"], go["K"], rtol=1e-7, atol=1e-9); np.testing.assert_allclose(gh["d"], go["d"], rtol=1e-7, atol=1e-9)
    lh, Jh = I.forwardpass(ph); lo, Jo = I.forwardpass(po)
    np.testing.assert_array_equal(lh, lo); np.testing.assert_allclose(Jh, Jo, rtol=1e-10)
    ph, po = configs.cartpole_problem(batch=B, lib=hip), configs.cartpole_problem(batch=B, lib=oracle)
    for p in (ph, po):
        T.set_goal_state(p, Xf)
    sh, so = T.iLQRSolver(ph).solve(), T.iLQRSolver(po).solve()
    assert_solve_parity(sh, so, ph, po, unconverged_rtol=1e-4)
    X = T.states(ph)
    done = sh.stats["status"] == T.capi.SOLVE_SUCCEEDED
    assert done.mean() > 0.9 and np.abs(X[done, -1, 0] - Xf[done, 0]).max() < 0.05


@pytest.mark.gpu
def test_per_trajectory_goals_quadrotor_and_constraints_on_gpu(hip, oracle):
    """Quadrotor (QuatLQRCost: the vector part of the goal per trajectory, the attitude reference shared), iLQR; and an AL solve of the
    double integrator with control bounds and per-trajectory goals."""
    from test_gpu_parity import assert_solve_parity
    B = 40
    rng = np.random.default_rng(8)
    def quad(lib):
        p = configs.quadrotor_problem(batch=B, N=41, tf=1.0, lib=lib)
        Xf = np.tile(p.xf, (B, 1)); Xf[:, :3] += rng0.uniform(-0.5, 0.5, (B, 3))
        T.set_goal_state(p, Xf)
        return p, Xf
    rng0 = np.random.default_rng(8); ph, Xf = quad(hip)
    rng0 = np.random.default_rng(8); po, _ = quad(oracle)
    sh, so = T.iLQRSolver(ph).solve(), T.iLQRSolver(po).solve()
    assert_solve_parity(sh, so, ph, po)
    assert np.abs(T.states(ph)[:, -1, :3] - Xf[:, :3]).max() < 0.4 and np.abs(T.states(ph)[:, -1, :3] - ph.xf[:3]).max() > 0.4   # 1 s: towards ITS goal
    from test_infeasible import di_problem
    out = []
    for lib in (hip, oracle):
        model = T.DoubleIntegrator(1.0, 2)
        obj = T.LQRObjective(np.ones(4), 0.1 * np.ones(2), 100 * np.ones(4), np.array([1.0, 2.0, 0, 0]), 31)
        cons = T.ConstraintList(4, 2, 31)
        T.add_constraint(cons, T.BoundConstraint(4, 2, u_max=1.5, u_min=-1.5), (1, 30))
        p = T.Problem(model, obj, np.zeros(4), 3.0, constraints=cons, batch=B, lib=lib)
        G = np.zeros((B, 4)); G[:, :2] = np.random.default_rng(2).uniform(-2, 2, (B, 2))
        T.set_goal_state(p, G)
        out.append((T.ALSolver(p).solve(), p, G))
    (sh, ph, G), (so, po, _) = out
    assert_solve_parity(sh, so, ph, po)
    assert np.all(sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) and np.abs(T.states(ph)[:, -1, :2] - G[:, :2]).max() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cartpole", "cartpole_lane", "quadrotor"])
def test_per_trajectory_goal_constraints_on_gpu(kind, hip, oracle, monkeypatch):
    """One GoalConstraint target per trajectory (to_set_constraint_params_batch) on the GPU against the oracle: constraint values, the AL
    expansion's gains, the forward pass, then whole AL and ALTRO solves with integers bit-exact — Cartpole on the cooperative path and on
    the lane path with compaction, the Quadrotor with the C5 constraint set (goal on position + velocities, SOC on the controls)."""
    from test_gpu_parity import assert_solve_parity
    if kind == "cartpole_lane":
        monkeypatch.setenv("TRAJOPT_BACKWARD", "lane"); monkeypatch.setenv("TRAJOPT_ACCEPT_ROLL_MIN", "1")
    def mk(lib):
        if kind == "quadrotor":
            p = configs.quadrotor_problem(batch=24, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=lib)
            Xf = np.tile(p.xf, (p.B, 1)); Xf[:, :3] += np.random.default_rng(4).uniform(-0.6, 0.6, (p.B, 3))
        else:
            p = configs.cartpole_problem(batch=300 if kind == "cartpole_lane" else 70, constrained=True, lib=lib)
            Xf = cartpole_goals(p.B, seed=3); Xf[:, 0] *= 0.5
        T.set_goal_state(p, Xf)
        return p, Xf
    (ph, Xf), (po, _) = mk(hip), mk(oracle)
    gi = len(ph.constraints) - 1 if kind != "quadrotor" else next(i for i, c in enumerate(ph.constraints.constraints) if isinstance(c, T.GoalConstraint))
    for p in (ph, po):
        T.rollout(p); I.dual_update(p); I.dual_update(p)
    np.testing.assert_allclose(T.evaluate_constraints(ph, gi), T.evaluate_constraints(po, gi), rtol=1e-11, atol=1e-12)
    np.testing.assert_array_equal(T.constraint_jacobians(ph, gi), T.constraint_jacobians(po, gi))
    np.testing.assert_allclose(T.max_violation(ph), T.max_violation(po), rtol=1e-11)
    np.testing.assert_allclose(I.al_cost(ph), I.al_cost(po), rtol=1e-11)
    for p in (ph, po):
        I.expand(p); I.backwardpass(p)
    gh, go = I.gains(ph), I.gains(po)
    np.testing.assert_allclose(gh["K"], go["K"], rtol=1e-6, atol=1e-8); np.testing.assert_allclose(gh["d"], go["d"], rtol=1e-6, atol=1e-8)
    lh, Jh = I.forwardpass(ph); lo, Jo = I.forwardpass(po)
    np.testing.assert_array_equal(lh, lo); np.testing.assert_allclose(Jh, Jo, rtol=1e-9)
    for Solver, kw in ((T.ALTROSolver, dict(n_steps=configs.C5_PN_STEPS) if kind == "quadrotor" else {}),
                       (T.ALSolver, dict(constraint_tolerance=1e-4))):
        (ph, Xf), (po, _) = mk(hip), mk(oracle)
        sh, so = Solver(ph, **kw).solve(), Solver(po, **kw).solve()
        assert_solve_parity(sh, so, ph, po, rtol=1e-5 if Solver is T.ALSolver else 1e-6)
        ok = sh.stats["status"] == T.capi.SOLVE_SUCCEEDED
        assert ok.mean() > 0.9
        inds = [j - 1 for j in ph.constraints.constraints[gi].inds]
        assert np.abs(T.states(ph)[ok][:, -1, inds] - Xf[ok][:, inds]).max() < (2e-4 if Solver is T.ALSolver else 1e-5)


@pytest.mark.gpu
def test_per_trajectory_linear_rhs_on_gpu(hip, oracle):
    """One LinearConstraint right-hand side per trajectory on the GPU (stored as the minimum-norm shift of z: A (z - A^+ db) - b) against the oracle
    (which replaces b in a per-trajectory copy of the descriptor): values to rounding, Jacobians exactly, gains, forward pass, then AL and ALTRO solves
    with integers bit-exact.  Linearly dependent rows are refused."""
    from test_gpu_parity import assert_solve_parity
    B = 40
    bv = linear_rhs(B)
    (ph, _), (po, _) = linear_problem(hip, B, bv), linear_problem(oracle, B, bv)
    for p in (ph, po):
        T.initial_controls(p, np.array([0.3, 0.5])); T.rollout(p); I.dual_update(p); I.dual_update(p)
    np.testing.assert_allclose(T.evaluate_constraints(ph, 0), T.evaluate_constraints(po, 0), rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(T.constraint_jacobians(ph, 0), T.constraint_jacobians(po, 0))
    np.testing.assert_allclose(T.max_violation(ph), T.max_violation(po), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(I.al_cost(ph), I.al_cost(po), rtol=1e-11)
    for p in (ph, po):
        I.expand(p); I.backwardpass(p)
    gh, go = I.gains(ph), I.gains(po)
    np.testing.assert_allclose(gh["K"], go["K"], rtol=1e-7, atol=1e-9); np.testing.assert_allclose(gh["d"], go["d"], rtol=1e-7, atol=1e-9)
    lh, Jh = I.forwardpass(ph); lo, Jo = I.forwardpass(po)
    np.testing.assert_array_equal(lh, lo); np.testing.assert_allclose(Jh, Jo, rtol=1e-10)
    for Solver in (T.ALSolver, T.ALTROSolver):
        (ph, _), (po, _) = linear_problem(hip, B, bv), linear_problem(oracle, B, bv)
        sh, so = Solver(ph).solve(), Solver(po).solve()
        assert_solve_parity(sh, so, ph, po)
        assert (sh.stats["status"] == T.capi.SOLVE_SUCCEEDED).all()
        X, U = T.states(ph), T.controls(ph)
        c0 = X[:, :-1, 0] - 0.5 * X[:, :-1, 1] - bv[:, None, 0]
        c1 = U[:, :, 0] + 2 * U[:, :, 1] - bv[:, None, 1]
        assert max(c0.max(), c1.max()) < 2e-6 and (np.abs(c1).min(axis=1) < 1e-3).mean() > 0.2     # every trajectory inside ITS half-planes; the control row binds for some
    model = T.DoubleIntegrator(1.0, 2)
    cons = T.ConstraintList(4, 2, 11)
    T.add_constraint(cons, T.LinearConstraint(4, 2, np.array([[1.0, 2.0], [2.0, 4.0]]), np.zeros(2), T.Inequality(), inds=[1, 2]), (1, 10))
    obj = T.LQRObjective(np.ones(4), np.ones(2), np.ones(4), np.zeros(4), 11)
    for lib in (hip,):
        pd = T.Problem(model, obj, np.zeros(4), 1.0, constraints=cons, batch=3, lib=lib)
        with pytest.raises(T.UnsupportedError, match="linearly independent"):
            T.set_constraint_params_batch(pd, 0, np.zeros((3, 2)))
