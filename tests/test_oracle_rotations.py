"""RigidBody{MRP} / RigidBody{RodriguesParam} (SURVEY §8(f)3 remainder; src/lie_costs.jl:1-3, ErrorQuadratic{Rot} :178-241): the
Quadrotor with a three-parameter attitude, on the oracle.  Known answers come from the physics, which does not care how
the attitude is written down: the same physical state in the three representations has the same accelerations, the same
error vector against the same reference, the same ErrorQuadratic cost; every derivative is checked against finite
differences of the oracle's own values."""
import ctypes as C

import numpy as np
import pytest

import trajopt_amd as T
from trajopt_amd import internal as I

ROTS = ("quat", "mrp", "rp")


def qmul(a, b):
    return np.r_[a[0] * b[0] - a[1:] @ b[1:], a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])]


def to_quat(rot, att):
    if rot == "quat":
        return att / np.linalg.norm(att)
    if rot == "mrp":
        n = att @ att
        return np.r_[1 - n, 2 * att] / (1 + n)
    return np.r_[1.0, att] / np.sqrt(1 + att @ att)


def from_quat(rot, q):
    return q if rot == "quat" else (q[1:] / (1 + q[0]) if rot == "mrp" else q[1:] / q[0])


def oplus(rot, x, d):
    """x ⊕ δ: the error-state retraction with the Cayley map (attitude: q ⊗ [1, δφ]/sqrt(1+|δφ|²))."""
    na = 4 if rot == "quat" else 3
    q = qmul(to_quat(rot, x[3:3 + na]), np.r_[1.0, d[3:6]] / np.sqrt(1 + d[3:6] @ d[3:6]))
    if rot == "quat":
        q = q * np.linalg.norm(x[3:7])  # keep the (possibly non-unit) norm of the stored quaternion
    return np.r_[x[:3] + d[:3], from_quat(rot, q), x[3 + na:] + d[6:]]


def call(oracle, name, model, *arrays):
    params = (C.c_double * 16)(*(model.params() + [0.0] * (16 - len(model.params()))))
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    oracle.call(name, model.model_id, params, *[pd(a) if isinstance(a, np.ndarray) else a for a in arrays])


def physical_states(seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        if q[0] < 0.2:
            q[0] = 0.2 + abs(q[0]); q /= np.linalg.norm(q)   # keep away from the RodriguesParam singularity (q_w = 0)
        yield rng.standard_normal(3), q, rng.standard_normal(3), 0.5 * rng.standard_normal(3), rng.uniform(0.5, 2.5, 4)


def test_same_physics_in_every_representation(oracle):
    for r, q, v, w, u in physical_states(1, 6):
        acc = {}
        for rot in ROTS:
            model = T.Quadrotor(rotation=rot)
            x = model.build_state(r, q, v, w)
            xd = np.zeros(model.n)
            call(oracle, "dynamics", model, x, u, xd)
            na = 4 if rot == "quat" else 3
            np.testing.assert_allclose(xd[:3], v, rtol=0, atol=0)
            acc[rot] = (xd[3 + na:], xd[3:3 + na])
        for rot in ("mrp", "rp"):
            np.testing.assert_allclose(acc[rot][0], acc["quat"][0], rtol=1e-13, atol=1e-13)   # v̇, ω̇
            # attitude rate: d/dt of the conversion applied to q̇ = ½ q ⊗ (0, ω)
            eps = 1e-7
            qd = acc["quat"][1]
            fd = (from_quat(rot, (q + eps * qd) / np.linalg.norm(q + eps * qd)) - from_quat(rot, (q - eps * qd) / np.linalg.norm(q - eps * qd))) / (2 * eps)
            np.testing.assert_allclose(acc[rot][1], fd, rtol=1e-7, atol=1e-8)


def test_state_diff_is_the_rodrigues_vector_of_the_relative_rotation(oracle):
    states = list(physical_states(2, 4))
    for (r0, q0, v0, w0, _), (r1, q1, v1, w1, _) in zip(states[:-1], states[1:]):
        rel = qmul(np.r_[q0[0], -q0[1:]], q1)
        expect = np.r_[r1 - r0, rel[1:] / rel[0], v1 - v0, w1 - w0]
        for rot in ROTS:
            model = T.Quadrotor(rotation=rot)
            dx = np.zeros(12)
            call(oracle, "state_diff", model, model.build_state(r1, q1, v1, w1), model.build_state(r0, q0, v0, w0), dx)
            np.testing.assert_allclose(dx, expect, rtol=1e-12, atol=1e-13, err_msg=rot)


def _problem(oracle, rot, cost="lqr", N=6, batch=1, seed=3):
    model = T.Quadrotor(rotation=rot)
    n, m = model.dims()
    rng = np.random.default_rng(seed)
    (r, q, v, w, u), (rf, qf, _, _, _) = list(physical_states(seed, 2))
    x0, xf = model.build_state(r, q, v, w), model.build_state(rf, qf)
    Qe = rng.uniform(0.5, 2.0, 12)
    if cost == "errquad":
        stage = T.ErrorQuadratic(model, Qe, np.full(m, 0.1), xf, model.hover_control())
        term = T.ErrorQuadratic(model, 10 * Qe, np.full(m, 0.1), xf, model.hover_control(), terminal=True)
    else:
        Qd = rng.uniform(0.5, 2.0, n)
        stage = T.LQRCost(Qd, np.full(m, 0.1), xf, model.hover_control())
        term = T.LQRCost(10 * Qd, np.full(m, 0.1), xf, model.hover_control(), terminal=True)
    prob = T.Problem(model, T.Objective(stage, term, N), x0, 0.1 * (N - 1), xf=xf, lib=oracle, batch=batch)
    T.initial_controls(prob, np.tile(u, (batch, N - 1, 1)) + 0.2 * rng.standard_normal((batch, N - 1, m)))
    T.rollout(prob)
    return prob, model


@pytest.mark.parametrize("rot", ["mrp", "rp"])
def test_discrete_jacobian_and_error_state_jacobians_against_finite_differences(oracle, rot):
    """The hand-derived continuous Jacobians through the RK4 chain rule (raw [A B]) and the error-state [Ā B̄] = G(x⁺)ᵀ[A B]G(x):
    the latter against differences of x⁺(x ⊕ δ) ⊖ x⁺(x), i.e. straight from the definitions of ⊕ and state_diff."""
    prob, model = _problem(oracle, rot)
    n, m, h = model.n, 4, 0.1
    X, U = T.states(prob)[0], T.controls(prob)[0]
    F = I.discrete_jacobian(prob)[0]
    I.expand(prob)
    A, B = I.dynamics_jacobians(prob)
    A, B = A[0], B[0]

    def step(x, u):
        xn = np.zeros(n)
        params = (C.c_double * 16)(*(model.params() + [0.0] * (16 - len(model.params()))))
        pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        oracle.call("discrete_dynamics", model.model_id, params, T.RK4, pd(np.ascontiguousarray(x)), pd(np.ascontiguousarray(u)), C.c_double(h), pd(xn))
        return xn

    def sdiff(x, x0):
        dx = np.zeros(12)
        call(oracle, "state_diff", model, np.ascontiguousarray(x), np.ascontiguousarray(x0), dx)
        return dx

    eps = 1e-6
    for k in (0, 3):
        x, u = X[k], U[k]
        fd = np.zeros((n, n + m))
        for j in range(n + m):
            e = np.zeros(n + m); e[j] = eps
            fd[:, j] = (step(x + e[:n], u + e[n:]) - step(x - e[:n], u - e[n:])) / (2 * eps)
        np.testing.assert_allclose(F[k], fd, rtol=2e-6, atol=2e-8)
        xn = step(x, u)
        fe = np.zeros((12, 12 + m))
        for j in range(12 + m):
            e = np.zeros(12 + m); e[j] = eps
            xp = step(oplus(rot, x, e[:12]), u + e[12:]); xm = step(oplus(rot, x, -e[:12]), u - e[12:])
            fe[:, j] = (sdiff(xp, xn) - sdiff(xm, xn)) / (2 * eps)
        np.testing.assert_allclose(np.hstack([A[k], B[k]]), fe, rtol=2e-6, atol=2e-8)


@pytest.mark.parametrize("rot", ["mrp", "rp"])
@pytest.mark.parametrize("cost", ["lqr", "errquad"])
def test_error_state_cost_expansion_against_finite_differences(oracle, rot, cost):
    """q̄ = Gᵀ∇J and Q̄ = GᵀHG + ∇²differential(p, ∂J/∂p) (the second-order term of a three-parameter attitude) against first
    and second differences of δ -> J_k(x ⊕ δ), for a plain quadratic on the raw state and for ErrorQuadratic{MRP/RP}."""
    prob, model = _problem(oracle, rot, cost)
    I.expand(prob)
    E = I.cost_expansion(prob)
    X = T.states(prob)[0]
    k = 2

    def Jk(x):
        p2, _ = _problem(oracle, rot, cost)
        Xs = T.states(p2); Xs[0, k] = x
        T.initial_states(p2, Xs)
        return T.stage_costs(p2)[0, k]

    eps = 1e-4
    x = X[k]
    g = np.array([(Jk(oplus(rot, x, eps * np.eye(12)[i])) - Jk(oplus(rot, x, -eps * np.eye(12)[i]))) / (2 * eps) for i in range(12)])
    np.testing.assert_allclose(E["qx"][0, k], g, rtol=1e-6, atol=1e-7)
    H = np.zeros((12, 12))
    J0 = Jk(x)
    for i in range(12):
        for j in range(i, 12):
            ei, ej = eps * np.eye(12)[i], eps * np.eye(12)[j]
            if i == j:
                H[i, i] = (Jk(oplus(rot, x, ei)) - 2 * J0 + Jk(oplus(rot, x, -ei))) / eps ** 2
            else:
                H[i, j] = H[j, i] = (Jk(oplus(rot, x, ei + ej)) - Jk(oplus(rot, x, ei - ej)) - Jk(oplus(rot, x, ej - ei)) + Jk(oplus(rot, x, -ei - ej))) / (4 * eps ** 2)
    np.testing.assert_allclose(E["Qxx"][0, k], H, rtol=2e-4, atol=2e-5)


def test_error_quadratic_value_is_representation_independent(oracle):
    """ErrorQuadratic penalises the error STATE, which is the same 12-vector whichever way the attitude is stored:
    ErrorQuadratic{MRP}, {RodriguesParam} and {QuatRotation} give the same cost on the same physical trajectory."""
    vals = {}
    for rot in ROTS:
        model = T.Quadrotor(rotation=rot)
        (r, q, v, w, u), (rf, qf, vf, wf, _) = list(physical_states(7, 2))
        Qe = np.linspace(0.5, 2.0, 12)
        c = T.ErrorQuadratic(model, Qe, np.full(4, 0.1), model.build_state(rf, qf, vf, wf), model.hover_control())
        prob = T.Problem(model, T.Objective(c, c, 2), model.build_state(r, q, v, w), 0.1, lib=oracle)
        X = np.tile(model.build_state(r, q, v, w), (1, 2, 1))
        T.initial_states(prob, X)
        T.initial_controls(prob, u)
        vals[rot] = T.stage_costs(prob)[0, 0]
    assert vals["mrp"] == pytest.approx(vals["quat"], rel=1e-12)
    assert vals["rp"] == pytest.approx(vals["quat"], rel=1e-12)


@pytest.mark.parametrize("rot", ["mrp", "rp"])
def test_ilqr_flies_the_three_parameter_quadrotor_to_its_goal(oracle, rot):
    model = T.Quadrotor(rotation=rot)
    n, m = model.dims()
    N = 51
    x0 = model.build_state([0.0, 0.0, 0.0])
    th = np.radians(60.0) / 2
    xf = model.build_state([1.0, 1.5, 0.5], [np.cos(th), 0.0, 0.0, np.sin(th)])
    Qe = np.r_[np.ones(3), 0.5 * np.ones(3), 0.1 * np.ones(6)]
    stage = T.ErrorQuadratic(model, Qe, np.full(m, 1e-2), xf, model.hover_control())
    term = T.ErrorQuadratic(model, 100 * Qe, np.full(m, 1e-2), xf, model.hover_control(), terminal=True)
    prob = T.Problem(model, T.Objective(stage, term, N), x0, 2.5, xf=xf, lib=oracle)
    T.initial_controls(prob, model.hover_control())
    s = T.iLQRSolver(prob).solve()
    assert int(s.stats["status"][0]) == T.capi.SOLVE_SUCCEEDED
    X = T.states(prob)[0]
    np.testing.assert_allclose(X[-1, :3], xf[:3], atol=2e-2)
    np.testing.assert_allclose(X[-1, 3:6], xf[3:6], atol=2e-2)
    with pytest.raises(T.ArgumentError):   # a state without a quaternion cannot carry the quaternion geodesic cost
        T.Problem(model, T.Objective(T.DiagonalQuatCost(np.ones(n), np.ones(m)), T.DiagonalQuatCost(np.ones(n), np.ones(m)), N), x0, 2.5, lib=oracle)
