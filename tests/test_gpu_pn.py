"""GPU parity of the projected-Newton polish (csrc/k_pn.h, one wave per trajectory) and of the ALTRO driver (AL-iLQR down to
projected_newton_tolerance, then the polish) against the CPU oracle (oracle/oracle_pn.h), through the C-ABI:
to_pn_solve, to_altro_solve, to_dynamics_defect, the asynchronous solve entry points — and the reference's own published ALTRO
result (examples/Cartpole.ipynb cells 17-23) reproduced on the GPU."""
import ctypes as C
import json
import math
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
from test_gpu_parity import assert_trajectories_close

pytestmark = pytest.mark.gpu
G = json.loads((Path(__file__).parent / "golden" / "reference_goldens.json").read_text())


def _defaults(prob):
    o = T.SolverOptions(lib=prob._lib)
    prob._call("set_options", C.byref(o._o))


def _quickstart(lib, batch):
    """quickstart (examples/quickstart.jl) from starts OFF the obstacle's symmetry axis: from x0 = 0 with U0 = 0 the straight line to
    the goal runs through the circle's centre — a saddle the AL stage never leaves (oracle and GPU alike), and nudged off it by
    U0 alone the early outer iterations are so sensitive that a 2e-14 difference in J moves an iteration count."""
    p = configs.quickstart_problem(batch=batch, lib=lib)
    x0 = np.zeros((batch, 4))
    x0[:, 0] = 0.25 + 0.15 * np.arange(batch)
    x0[:, 1] = -0.1 * np.arange(batch)
    p.set_initial_state(x0)
    T.initial_controls(p, np.array([0.1, 0.0]))
    return p


PN_CASES = {
    "cartpole_bounds_goal": (lambda lib: configs.cartpole_problem(batch=70, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=lib), 1e-3),
    "quickstart_circle_soc_bound_goal": (lambda lib: _quickstart(lib, 5), 1e-3),
    "quadrotor_goal_soc": (lambda lib: configs.quadrotor_problem(batch=24, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=lib), 0.0),
    "quadrotor_goal_soc_perturbed": (lambda lib: configs.quadrotor_problem(batch=9, N=41, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=lib), 3e-4),
}


@pytest.mark.parametrize("name", list(PN_CASES))
def test_pn_solve_vs_oracle(name, hip, oracle):
    """to_pn_solve from the same (off-manifold) trajectories on both sides: projection counts and status bit-exact, X / U to 1e-8
    (measured ~1e-12: Newton contracts the rounding differences of the two factorisations)."""
    build, scale = PN_CASES[name]
    po, ph = build(oracle), build(hip)
    T.ALSolver(po, constraint_tolerance=1e-3).solve()
    _defaults(po)
    X, U = T.states(po), T.controls(po)
    if scale:
        rng = np.random.default_rng(9)
        X, U = X + scale * rng.normal(size=X.shape), U + scale * rng.normal(size=U.shape)
    for p in (po, ph):
        T.initial_states(p, X); T.initial_controls(p, U)
    np.testing.assert_allclose(T.dynamics_defect(ph), T.dynamics_defect(po), rtol=1e-9, atol=1e-13)
    sh, so = T.ProjectedNewtonSolver(ph).solve(), T.ProjectedNewtonSolver(po).solve()
    for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
        np.testing.assert_array_equal(sh.stats[k], so.stats[k], err_msg=k)
    assert np.all(sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) and np.all(sh.stats["iterations_pn"] >= 1)
    np.testing.assert_allclose(T.states(ph), T.states(po), rtol=0, atol=1e-8)
    np.testing.assert_allclose(T.controls(ph), T.controls(po), rtol=0, atol=1e-8)
    np.testing.assert_allclose(sh.stats["c_max"], so.stats["c_max"], rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(sh.stats["cost"], so.stats["cost"], rtol=1e-9)
    assert sh.stats["c_max"].max() <= 1e-6
    np.testing.assert_allclose(T.dynamics_defect(ph), T.dynamics_defect(po), rtol=1e-3, atol=1e-10)
    assert T.max_violation(ph).max() <= 1e-6


def assert_altro_parity(sh, so, ph, po, rtol=1e-6):
    for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
        np.testing.assert_array_equal(sh.stats[k], so.stats[k], err_msg=k)
    np.testing.assert_allclose(sh.stats["cost"], so.stats["cost"], rtol=rtol)
    np.testing.assert_allclose(sh.stats["c_max"], so.stats["c_max"], rtol=1e-2, atol=1e-8)
    assert_trajectories_close(T.states(ph), T.states(po), rtol, "X")
    assert_trajectories_close(T.controls(ph), T.controls(po), rtol, "U")
    assert sh.total_iterations == so.total_iterations


ALTRO_CASES = {
    "cartpole": lambda lib: configs.cartpole_problem(batch=64, constrained=True, lib=lib),
    "quickstart": lambda lib: _quickstart(lib, 3),
    "quadrotor_N61": lambda lib: configs.quadrotor_problem(batch=40, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=lib),
}


@pytest.mark.parametrize("name", list(ALTRO_CASES))
def test_altro_solve_vs_oracle(name, hip, oracle):
    """to_altro_solve: integers (iLQR iterations, outer iterations, projections, status) bit-exact, X / U / J at 1e-6."""
    ph, po = ALTRO_CASES[name](hip), ALTRO_CASES[name](oracle)
    sh, so = T.ALTROSolver(ph).solve(), T.ALTROSolver(po).solve()
    assert_altro_parity(sh, so, ph, po)
    ok = sh.stats["status"] == T.capi.SOLVE_SUCCEEDED
    assert ok.mean() >= 0.9 and np.all(sh.stats["c_max"][ok] <= 1e-6)
    assert np.any(sh.stats["iterations_pn"] > 0)
    # projected_newton = 0 is the AL stage alone (run to constraint_tolerance)
    ph2, ph3 = ALTRO_CASES[name](hip), ALTRO_CASES[name](hip)
    s2, sa = T.ALTROSolver(ph2, projected_newton=0).solve(), T.ALSolver(ph3).solve()
    for k in ("iterations", "iterations_outer", "status"):
        np.testing.assert_array_equal(s2.stats[k], sa.stats[k])
    np.testing.assert_array_equal(T.states(ph2), T.states(ph3))
    assert np.all(s2.stats["iterations_pn"] == 0)


def test_G4_cartpole_altro_on_gpu(hip, oracle):
    """examples/Cartpole.ipynb cells 17-23 reproduced ON THE GPU: ALTRO, 40 iterations, J = 1.552558743680986, violation 3.4e-9."""
    ga = G["G4_cartpole_altro"]
    def run(lib):
        o = T.SolverOptions(lib=lib, cost_dt_scaling=1, cost_tolerance_intermediate=1e-2, penalty_scaling=10.0, penalty_initial=1.0)
        prob = configs.cartpole_problem(batch=2, lib=lib, options=o, constrained=True, integration=T.RK3)
        return prob, T.ALTROSolver(prob).solve()
    prob, s = run(hip)
    _, so = run(oracle)
    for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
        np.testing.assert_array_equal(s.stats[k], so.stats[k], err_msg=k)
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED
    assert s.stats["cost"][0] == pytest.approx(ga["cost"], rel=1e-6)
    assert abs(int(s.stats["iterations"][0]) + 1 - ga["iterations"]) <= 1 and int(s.stats["iterations_pn"][0]) >= 1
    assert s.stats["c_max"][0] < 1e-8
    U = T.controls(prob)[0, :, 0]
    np.testing.assert_allclose(U[-len(ga["U_tail"]):], ga["U_tail"], atol=2e-5)
    np.testing.assert_allclose(T.states(prob)[0, -1], [0, math.pi, 0, 0], atol=1e-8)


def test_dynamics_defect_of_rollouts(hip):
    for prob in (configs.cartpole_problem(batch=70, N=31, lib=hip), configs.quadrotor_problem(batch=66, N=21, tf=1.0, lib=hip)):
        T.rollout(prob)
        # k_rollout and k_defect are separately compiled (FMA contraction may differ): rounding level, not structure
        assert T.dynamics_defect(prob).max() < 1e-13


def test_pn_three_parameter_attitudes_and_hybrid(hip, oracle):
    """the other model keys through the kernel: MRP / RodriguesParam rigid bodies, the hybrid double integrator"""
    for rot in ("mrp", "rp"):
        model = T.Quadrotor(rotation=rot)
        n, m, N = 12, 4, 31
        x0 = model.build_state([0.0, 0.0, 0.0]); xf = model.build_state([0.6, -0.4, 0.5])
        def build(lib):
            obj = T.LQRObjective(np.r_[np.ones(3), np.full(3, 0.5), np.full(6, 0.1)], np.full(m, 1e-2),
                                 np.r_[np.full(3, 100.0), np.full(3, 50.0), np.full(6, 10.0)], xf, N, uf=np.full(m, 1.22625))
            cons = T.ConstraintList(n, m, N)
            T.add_constraint(cons, T.GoalConstraint(xf, [1, 2, 3, 7, 8, 9]), N)
            T.add_constraint(cons, T.BoundConstraint(n, m, u_min=0.0, u_max=3.0), range(1, N))
            p = T.Problem(model, obj, x0, 1.5, xf=xf, constraints=cons, batch=5, lib=lib)
            T.initial_controls(p, np.full(m, 1.22625))
            return p
        ph, po = build(hip), build(oracle)
        sh, so = T.ALTROSolver(ph).solve(), T.ALTROSolver(po).solve()
        assert_altro_parity(sh, so, ph, po)
        assert np.all(sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) and np.all(sh.stats["iterations_pn"] >= 1)


def test_async_solves_match_sync_and_overlap(hip):
    """to_*_solve_async + to_solve_wait: two handles in flight at once give what the synchronous calls give."""
    def mk():
        return configs.cartpole_problem(batch=256, lib=hip), configs.quadrotor_problem(batch=64, N=61, tf=3.0, constrained=True,
                                                                                     goal_inds=configs.C5_GOAL_INDS, lib=hip)
    pa, pb = mk()
    sa, sb = T.iLQRSolver(pa).solve(), T.ALTROSolver(pb).solve()
    qa, qb = mk()
    ta, tb = T.iLQRSolver(qa), T.ALTROSolver(qb)
    ta.solve_async(); tb.solve_async()
    with pytest.raises(T.ArgumentError):
        ta.solve()                       # one solve in flight per handle
    # ... and NO other call on it (the worker owns the handle's stream, staging buffer and argument block): getters, setters and the
    # phase API all refuse until to_solve_wait; pure descriptor getters (to_dims) stay available
    for call in (lambda: T.states(qb), lambda: T.controls(qb), lambda: T.cost(qb), lambda: T.rollout(qb), lambda: T.max_violation(qb),
                 lambda: T.initial_controls(qb, np.zeros(qb.m)), lambda: qb._call("set_options", C.byref(qb._lib.default_options()))):
        with pytest.raises(T.ArgumentError, match="in flight"):
            call()
    n_ = C.c_int32(0)
    qb._call("dims", C.byref(n_), None, None, None, None)
    assert n_.value == 13
    tb.wait(); ta.wait()
    for s, t, p, q in ((sa, ta, pa, qa), (sb, tb, pb, qb)):
        for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
            np.testing.assert_array_equal(s.stats[k], t.stats[k], err_msg=k)
        np.testing.assert_array_equal(T.states(p), T.states(q))
        np.testing.assert_array_equal(T.controls(p), T.controls(q))


def test_early_polish_matches_polish_after_the_al_stage(hip, monkeypatch):
    """ALTRO solves hand the trajectories whose AL stage has ended to the polish on a second stream while the rest of the batch
    still iterates (trajopt_hip.hip, early polish).  The polish of a trajectory depends on nothing but its own (X, U), so every
    output must EQUAL the one-polish-at-the-end path (TRAJOPT_PN_EARLY=0): trajectories, status, projection counts, violation."""
    out = []
    for early, at in (("0", "4"), ("1", "4"), ("4", "2"), ("8", "1")):  # hand-overs allowed, first one when B / at are left
        monkeypatch.setenv("TRAJOPT_PN_EARLY", early)
        monkeypatch.setenv("TRAJOPT_PN_EARLY_AT", at)
        p = configs.quadrotor_problem(batch=600, N=101, tf=5.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=hip)
        s = T.ALTROSolver(p, n_steps=configs.C5_PN_STEPS).solve()
        out.append((s.stats, T.states(p), T.controls(p)))
    ref = out[0]
    assert (ref[0]["iterations_pn"] > 0).mean() > 0.5 and len(np.unique(ref[0]["iterations"])) > 20
    for st, X, U in out[1:]:
        for k in ("iterations", "iterations_outer", "iterations_pn", "status", "cost", "c_max"):
            np.testing.assert_array_equal(st[k], ref[0][k], err_msg=k)
        np.testing.assert_array_equal(X, ref[1])
        np.testing.assert_array_equal(U, ref[2])


def test_polish_in_workspace_chunks(hip, monkeypatch):
    """A workspace budget smaller than the batch (TRAJOPT_PN_WS_GB): the polish goes through the list in chunks that reuse the slots,
    and ALTRO solves fall back to one polish after the AL stage.  Same results as with one slot per trajectory."""
    out = []
    for gb in (None, "0.02"):
        if gb:
            monkeypatch.setenv("TRAJOPT_PN_WS_GB", gb)
        p = configs.quadrotor_problem(batch=150, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=hip)
        s = T.ALTROSolver(p, n_steps=configs.C5_PN_STEPS).solve()
        q = configs.cartpole_problem(batch=90, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=hip)
        sq = T.ALTROSolver(q).solve()
        out.append((s.stats, T.states(p), T.controls(p), sq.stats, T.states(q), T.controls(q)))
    a, b = out
    assert (a[0]["iterations_pn"] > 0).sum() > 100 and (a[3]["iterations_pn"] > 0).sum() > 40   # more trajectories than 0.02 GB hold
    for i in (0, 3):
        for k in ("iterations", "iterations_pn", "status", "cost", "c_max"):
            np.testing.assert_array_equal(a[i][k], b[i][k], err_msg=k)
    for i in (1, 2, 4, 5):
        np.testing.assert_array_equal(a[i], b[i])


def test_full_size_C5_altro_vs_oracle(hip, oracle):
    """BASELINE config C5 at its own shape (Quadrotor + GoalConstraint + SOC cone, N=201, B=8192) solved as the reference's stack
    solves constrained problems — ALTRO: AL-iLQR to 1e-3, projected-Newton polish to 1e-6 — against the oracle on 512 sampled
    trajectories (the first two tiles, two tiles from the middle, the last four).  HARD asserts: iLQR iterations, outer
    iterations, projection counts and status bit-exact on every sampled trajectory, X / U / J at the north-star 1e-6 — the
    band the AL-only test needs (hundreds of iterations at penalty 1e8 amplify last-bit differences) is gone with the tail."""
    kw = dict(N=201, constrained=True, goal_inds=configs.C5_GOAL_INDS)
    ph = configs.quadrotor_problem(batch=8192, lib=hip, **kw)
    sh = T.ALTROSolver(ph, n_steps=configs.C5_PN_STEPS).solve()
    Xh, Uh = T.states(ph), T.controls(ph)
    from oracle_binding import set_threads
    total = 0
    for b0, cnt in ((0, 128), (4000, 128), (8192 - 256, 256)):
        po = configs.quadrotor_problem(batch=cnt, b_offset=b0, lib=oracle, **kw)
        set_threads(po, oracle.max_threads())
        so = T.ALTROSolver(po, n_steps=configs.C5_PN_STEPS).solve()
        idx = np.arange(b0, b0 + cnt)
        for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
            np.testing.assert_array_equal(sh.stats[k][idx], so.stats[k], err_msg=f"{k} (block at {b0})")
        np.testing.assert_allclose(sh.stats["cost"][idx], so.stats["cost"], rtol=1e-6)
        assert_trajectories_close(Xh[idx], T.states(po), 1e-6, "X")
        assert_trajectories_close(Uh[idx], T.controls(po), 1e-6, "U")
        # element-wise view of the same comparison (rtol 1e-6, atol 1e-9)
        for A, R in ((Xh[idx], T.states(po)), (Uh[idx], T.controls(po))):
            frac = np.mean(np.abs(A - R) <= 1e-6 * np.abs(R) + 1e-9)
            assert frac >= 0.9999, frac
        total += cnt
    ok = (sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) & (sh.stats["c_max"] <= 1e-6)
    print(f"C5 ALTRO: converged {ok.mean():.4f} of 8192; projections {np.bincount(sh.stats['iterations_pn'])}; "
          f"iterations {sh.total_iterations}; {total} trajectories compared with the oracle")
    assert ok.mean() >= 0.99
    assert set(np.unique(sh.stats["status"])) <= {T.capi.SOLVE_SUCCEEDED, T.capi.PROJECTION_FAIL, T.capi.MAX_ITERATIONS_OUTER, T.capi.MAX_ITERATIONS}


def test_full_size_C5prime_quatvec_goal_altro_vs_oracle(hip, oracle):
    """C5' = C5 with the terminal ATTITUDE pinned too, the reference's way: QuatVecEq(qf)@N (src/constraints.jl:938-965 — the vector
    part of the normalised quaternion, feasible under RK4's norm drift where the full 13-state GoalConstraint is not) next to
    Goal(position, velocities) and the SOC cone.  B = 8192, ALTRO; 256 sampled trajectories against the oracle, hard asserts."""
    from oracle_binding import set_threads
    kw = dict(N=201, constrained=True, goal_inds=configs.C5_GOAL_INDS, quatvec_goal=True)
    ph = configs.quadrotor_problem(batch=8192, lib=hip, **kw)
    sh = T.ALTROSolver(ph, n_steps=configs.C5_PN_STEPS).solve()
    Xh, Uh = T.states(ph), T.controls(ph)
    for b0, cnt in ((0, 128), (8192 - 128, 128)):
        po = configs.quadrotor_problem(batch=cnt, b_offset=b0, lib=oracle, **kw)
        set_threads(po, oracle.max_threads())
        so = T.ALTROSolver(po, n_steps=configs.C5_PN_STEPS).solve()
        idx = np.arange(b0, b0 + cnt)
        for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
            np.testing.assert_array_equal(sh.stats[k][idx], so.stats[k], err_msg=f"{k} (block at {b0})")
        np.testing.assert_allclose(sh.stats["cost"][idx], so.stats["cost"], rtol=1e-6)
        assert_trajectories_close(Xh[idx], T.states(po), 1e-6, "X")
        assert_trajectories_close(Uh[idx], T.controls(po), 1e-6, "U")
    ok = (sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) & (sh.stats["c_max"] <= 1e-6)
    qf = np.array([math.cos(math.radians(67.5)), 0, 0, math.sin(math.radians(67.5))])
    qN = Xh[ok, -1, 3:7] / np.linalg.norm(Xh[ok, -1, 3:7], axis=1, keepdims=True)
    print(f"C5' ALTRO: converged {ok.mean():.4f}; projections {np.bincount(sh.stats['iterations_pn'])}; iterations {sh.total_iterations}")
    assert ok.mean() >= 0.98
    assert np.abs(np.abs(qN @ qf) - 1.0).max() < 1e-10      # the attitude arrived (up to the quaternion's sign)


def test_pn_without_constraints_and_minimal_horizon(hip, oracle):
    """no constraint list: the active set is the initial condition + the dynamics defects; a rollout is left untouched; N = 3"""
    for N, tf in ((31, 1.5), (3, 0.1)):
        ph, po = configs.cartpole_problem(batch=70, N=N, tf=tf, lib=hip), configs.cartpole_problem(batch=70, N=N, tf=tf, lib=oracle)
        T.rollout(ph); T.rollout(po)
        X0 = T.states(po).copy()
        s = T.ProjectedNewtonSolver(ph).solve()
        assert np.all(s.stats["iterations_pn"] == 0) and np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED)
        np.testing.assert_allclose(T.states(ph), X0, rtol=0, atol=1e-13)
        rng = np.random.default_rng(N)
        Xp = X0 + 1e-3 * rng.normal(size=X0.shape)
        for p in (ph, po):
            T.initial_states(p, Xp)
        sh, so = T.ProjectedNewtonSolver(ph).solve(), T.ProjectedNewtonSolver(po).solve()
        for k in ("iterations_pn", "status"):
            np.testing.assert_array_equal(sh.stats[k], so.stats[k])
        np.testing.assert_allclose(T.states(ph), T.states(po), rtol=0, atol=1e-9)
        np.testing.assert_allclose(T.controls(ph), T.controls(po), rtol=0, atol=1e-9)
        assert sh.stats["c_max"].max() <= 1e-6 and T.dynamics_defect(ph).max() <= 1e-6
