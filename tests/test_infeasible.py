"""ALTRO's infeasible start: Altro's InfeasibleModel — x+ = f_d(x, u) + w, one slack control per state — behind
TO_MODEL_INFEASIBLE, composed on the host the way Altro composes it with the reference's change_dimension family
(src/constraints.jl:820-936, src/constraint_list.jl:208-217, src/cost_functions.jl:391-401), and the cost-to-go getter.
CPU part: the oracle against closed forms / finite differences / the feasible-start solution.  GPU part: the HIP path against the oracle."""
import numpy as np
import pytest

import trajopt_amd as T
from trajopt_amd import internal as I
from trajectoryoptimization_jl_amd import configs


def di_problem(lib, D=1, B=5, N=31, tf=3.0, umax=2.0):
    """D-dimensional double integrator to a goal, control bounds, trajectories from perturbed starts (b = 0: the origin)."""
    model = T.DoubleIntegrator(1.0, D)
    n, m = model.dims()
    xf = np.zeros(n); xf[:D] = np.arange(1, D + 1)
    obj = T.LQRObjective(np.ones(n), 0.1 * np.ones(m), 100 * np.ones(n), xf, N)
    cons = T.ConstraintList(n, m, N)
    T.add_constraint(cons, T.GoalConstraint(xf), N)
    T.add_constraint(cons, T.BoundConstraint(n, m, u_max=umax, u_min=-umax), (1, N - 1))
    p = T.Problem(model, obj, np.zeros(n), tf, xf=xf, constraints=cons, batch=B, lib=lib)
    x0 = np.zeros((B, n)); x0[:, :D] = np.random.default_rng(3).uniform(-0.3, 0.3, (B, D)); x0[0] = 0
    p.set_initial_state(x0)
    return p, x0, xf


def line_guess(x0, xf, N):
    t = np.linspace(0, 1, N)[None, :, None]
    return x0[:, None, :] * (1 - t) + np.asarray(xf)[None, None, :] * t   # positions on a line, velocities inconsistent with them


def cartpole_infeasible(lib, B=6, amp=0.3):
    """The constrained Cartpole (bounds + goal) with a state guess its controls cannot produce."""
    p = configs.cartpole_problem(batch=B, constrained=True, lib=lib)
    T.rollout(p)
    X = T.states(p).copy()
    t = np.linspace(0, 1, p.N)
    X[:, :, 1] += amp * np.sin(np.pi * t)[None, :]
    X[:, :, 0] += 0.5 * amp * np.sin(2 * np.pi * t)[None, :]
    return T.InfeasibleProblem(p, X, R_inf=2.0), X


# ------------------------------------------------------------------------------------------------ CPU: host mirror + oracle
def test_change_dimension_of_costs():
    """change_dimension(cost, n, m, ix, iu) (src/cost_functions.jl:391-401): the lifted cost acts on x[ix], u[iu] as the original."""
    c = T.LQRCost(np.array([1.0, 2.0]), np.array([0.5]), np.array([0.3, -0.2]), np.array([0.1]))
    big = T.change_dimension(c, 5, 3, (2, 3), (3, 3))
    assert isinstance(big, T.DiagonalCost) and (big.n, big.m) == (5, 3)
    np.testing.assert_array_equal(big.Q, [0, 1, 2, 0, 0]); np.testing.assert_array_equal(big.R, [0, 0, 0.5])
    np.testing.assert_array_equal(big.q, [0, *c.q, 0, 0]); np.testing.assert_array_equal(big.r, [0, 0, *c.r])
    assert big.c == c.c and big.terminal == c.terminal
    dense = T.QuadraticCost(np.array([[2.0, 0.1], [0.1, 1.0]]), np.array([[0.3]]), np.array([[0.2, -0.1]]), [1.0, 2.0], [0.5], 0.7)
    bd = T.change_dimension(dense, 3, 2)            # leading entries, like the reference's default
    np.testing.assert_array_equal(bd.Q[:2, :2], dense.Q); assert not bd.Q[2].any() and not bd.Q[:, 2].any()
    np.testing.assert_array_equal(bd.H[:1, :2], dense.H); assert not bd.H[1].any()
    with pytest.raises(T.DimensionMismatch):
        T.change_dimension(c, 5, 3, (2, 4), (3, 3))
    obj = T.LQRObjective(np.ones(2), np.ones(1), np.ones(2), np.zeros(2), 5)
    lifted = T.change_dimension(obj, 2, 3)
    assert lifted.dims()[:2] == (2, 3) and lifted.cost[0] is lifted.cost[1] and lifted.cost[-1] is not lifted.cost[0]


def test_infeasible_model_descriptor_checks():
    assert T.InfeasibleModel(T.Cartpole()).dims() == (4, 5)
    assert T.InfeasibleModel(T.DoubleIntegrator(1.0, 2)).dims() == (4, 6)
    for bad in (T.Quadrotor(), T.DoubleIntegrator(1.0, 3)):     # (13, 17) / (6, 9): beyond the library's control dimension
        with pytest.raises(T.UnsupportedError):
            T.InfeasibleModel(bad)
    con = T.InfeasibleConstraint(4, 5)
    assert con.p == 4 and T.sense(con) == T.Equality()


def test_infeasible_controls_make_any_guess_feasible(oracle):
    p, x0, xf = di_problem(oracle, D=2)
    guess = line_guess(x0, xf, p.N)
    q = T.InfeasibleProblem(p, guess, R_inf=1.0)
    assert (q.n, q.m) == (4, 6) and len(q.constraints) == len(p.constraints) + 1
    T.rollout(q)
    np.testing.assert_allclose(T.states(q), guess, rtol=0, atol=1e-14)       # the rollout IS the guess
    U = T.controls(q)
    assert np.abs(U[:, :, 2:]).max() > 0.01                                   # ... bought with non-zero slack controls
    np.testing.assert_array_equal(U[:, :, :2], T.controls(p))                 # base controls untouched
    pc, X = cartpole_infeasible(oracle)
    T.rollout(pc)
    np.testing.assert_allclose(T.states(pc), X, rtol=0, atol=1e-13)


def test_infeasible_jacobian_is_A_B_I(oracle):
    """[A B I]: the base model's RK Jacobian next to an identity block — against central differences of the step."""
    pc, _ = cartpole_infeasible(oracle, B=2)
    F = I.discrete_jacobian(pc)                                               # [B, N-1, n, n+m]
    base = configs.cartpole_problem(batch=2, constrained=True, lib=oracle)
    T.initial_states(base, T.states(pc)); T.initial_controls(base, T.controls(pc)[:, :, :1])
    Fb = I.discrete_jacobian(base)
    np.testing.assert_allclose(F[..., :5], Fb, rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(F[..., 5:], np.broadcast_to(np.eye(4), F[..., 5:].shape))


def test_infeasible_altro_finds_the_feasible_start_optimum(oracle):
    """ALTROSolver(prob, infeasible=true) from a straight-line state guess: the slack controls end at zero, the base dynamics hold, and
    — the double integrator problem is convex — the trajectory is the one the feasible start finds."""
    for D in (1, 2):
        p, x0, xf = di_problem(oracle, D)
        T.initial_states(p, line_guess(x0, xf, p.N))
        s = T.ALTROSolver(p, infeasible=True, R_inf=1.0).solve()
        q = s.prob
        assert np.all(s.stats["status"] == T.capi.SOLVE_SUCCEEDED) and s.stats["c_max"].max() < 1e-8
        U = T.controls(q)
        assert np.abs(U[:, :, p.m:]).max() < 1e-8
        ref, _, _ = di_problem(oracle, D)
        sr = T.ALTROSolver(ref).solve()
        np.testing.assert_allclose(T.states(q), T.states(ref), atol=2e-7)
        np.testing.assert_allclose(U[:, :, :p.m], T.controls(ref), atol=2e-6)
        np.testing.assert_allclose(s.stats["cost"], sr.stats["cost"], rtol=1e-7)
        T.initial_states(ref, T.states(q)); T.initial_controls(ref, U[:, :, :p.m])
        assert T.dynamics_defect(ref).max() < 1e-8                            # feasible for the BASE model


def test_cost_to_go_of_the_oracle(oracle):
    """S_N = Qxx_N, and the quadratic model's predicted decrease along d: dV = sum_k (d'Qu, ½ d'Quu d) — here only the terminal
    identity and symmetry; the recursion itself is the backward pass the goldens G3 / G4 pin."""
    p = configs.cartpole_problem(batch=3, lib=oracle)
    T.rollout(p); I.expand(p); I.backwardpass(p)
    S, s = I.cost_to_go(p)
    Qxx = I.cost_expansion(p)["Qxx"] if hasattr(I, "cost_expansion") else None
    np.testing.assert_allclose(S, S.transpose(0, 1, 3, 2), atol=1e-12)
    np.testing.assert_allclose(S[:, -1], np.broadcast_to(np.diag(np.full(4, 100.0)), S[:, -1].shape))
    assert np.all(np.linalg.eigvalsh(S) > 0)


# ------------------------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["di1", "di2", "cartpole"])
def test_infeasible_phases_on_gpu(which, hip, oracle):
    def build(lib):
        if which == "cartpole":
            return cartpole_infeasible(lib)[0]
        p, x0, xf = di_problem(lib, 1 if which == "di1" else 2, B=70)
        return T.InfeasibleProblem(p, line_guess(x0, xf, p.N), R_inf=1.0)
    ph, po = build(hip), build(oracle)
    np.testing.assert_allclose(T.controls(ph), T.controls(po), rtol=1e-10, atol=1e-12)   # to_infeasible_controls
    for p in (ph, po):
        T.rollout(p)
    np.testing.assert_allclose(T.states(ph), T.states(po), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(T.cost(ph), T.cost(po), rtol=1e-12)
    np.testing.assert_allclose(I.al_cost(ph), I.al_cost(po), rtol=1e-12)
    np.testing.assert_allclose(T.max_violation(ph), T.max_violation(po), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(I.discrete_jacobian(ph), I.discrete_jacobian(po), rtol=1e-10, atol=1e-12)
    for p in (ph, po):
        I.expand(p); I.backwardpass(p)
    (Ah, Bh), (Ao, Bo) = I.dynamics_jacobians(ph), I.dynamics_jacobians(po)
    np.testing.assert_allclose(Ah, Ao, rtol=1e-10, atol=1e-12); np.testing.assert_allclose(Bh, Bo, rtol=1e-10, atol=1e-12)
    gh, go = I.gains(ph), I.gains(po)
    np.testing.assert_allclose(gh["K"], go["K"], rtol=1e-7, atol=1e-9); np.testing.assert_allclose(gh["d"], go["d"], rtol=1e-7, atol=1e-9)
    (Sh, sh), (So, so) = I.cost_to_go(ph), I.cost_to_go(po)                               # to_get_cost_to_go
    np.testing.assert_allclose(Sh, So, rtol=1e-8, atol=1e-9); np.testing.assert_allclose(sh, so, rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 2])
def test_infeasible_altro_double_integrator_on_gpu(D, hip, oracle):
    """ALTRO from an infeasible straight-line guess, HIP against the oracle: integers bit-exact, trajectories at 1e-6; the slack
    controls end at zero and the solution is the feasible start's."""
    from test_gpu_parity import assert_trajectories_close
    out = []
    for lib in (hip, oracle):
        p, x0, xf = di_problem(lib, D, B=70)
        T.initial_states(p, line_guess(x0, xf, p.N))
        s = T.ALTROSolver(p, infeasible=True, R_inf=1.0).solve()
        out.append((s, s.prob))
    (sh, ph), (so, po) = out
    for k in ("iterations", "iterations_outer", "iterations_pn", "status"):
        np.testing.assert_array_equal(sh.stats[k], so.stats[k], err_msg=k)
    assert_trajectories_close(T.states(ph), T.states(po), 1e-6, "X")
    assert_trajectories_close(T.controls(ph), T.controls(po), 1e-6, "U")
    np.testing.assert_allclose(sh.stats["cost"], so.stats["cost"], rtol=1e-6)
    assert np.all(sh.stats["status"] == T.capi.SOLVE_SUCCEEDED) and sh.stats["c_max"].max() < 1e-8
    assert np.abs(T.controls(ph)[:, :, D:]).max() < 1e-8
    ref, _, _ = di_problem(hip, D, B=70)
    T.ALTROSolver(ref).solve()
    np.testing.assert_allclose(T.states(ph), T.states(ref), atol=2e-7)


@pytest.mark.gpu
def test_infeasible_cartpole_solves_on_gpu(hip, oracle):
    """The nonlinear case: iLQR and AL solves of the infeasible Cartpole problem under an iteration cap, integers bit-exact."""
    from test_gpu_parity import assert_solve_parity
    ph, po = cartpole_infeasible(hip)[0], cartpole_infeasible(oracle)[0]
    sh, so = T.iLQRSolver(ph, iterations=25).solve(), T.iLQRSolver(po, iterations=25).solve()
    assert_solve_parity(sh, so, ph, po, unconverged_rtol=1e-5)
    ph, po = cartpole_infeasible(hip)[0], cartpole_infeasible(oracle)[0]
    kw = dict(iterations=20, iterations_outer=3)
    sh, so = T.ALSolver(ph, **kw).solve(), T.ALSolver(po, **kw).solve()
    assert_solve_parity(sh, so, ph, po, unconverged_rtol=1e-5)
    assert sh.stats["iterations"].min() >= 20


@pytest.mark.gpu
def test_infeasible_quadrotor_is_refused(hip):
    p = configs.quadrotor_problem(batch=2, N=11, tf=0.5, constrained=True, lib=hip)
    with pytest.raises(T.UnsupportedError):
        T.ALTROSolver(p, infeasible=True)
