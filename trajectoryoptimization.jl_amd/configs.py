"""The BASELINE.json workloads as Problem builders with synthetic,
reproducible inputs.  Host-side input generation only; shared by tests/, bench.py and smoke()."""
from __future__ import annotations

import math

import numpy as np

from . import api as T

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_uniform(seed, index):
    """U[0,1) doubles from the counter-based splitmix64 stream: element ``i`` is output number ``i`` of
    splitmix64(seed).  Indexed by the GLOBAL trajectory number so shards draw identical inputs."""
    idx = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _shard(batch, b_offset):
    return np.arange(b_offset, b_offset + batch, dtype=np.uint64)


def cartpole_x0(batch, b_offset=0, seed=1):
    """C2: x0_b = [ξ1, ξ2, 0, 0], ξ1~U(-0.5,0.5), ξ2~U(-0.3,0.3); global b=0 is exactly zero."""
    b = _shard(batch, b_offset)
    x0 = np.zeros((batch, 4))
    x0[:, 0] = splitmix64_uniform(seed, 2 * b) - 0.5
    x0[:, 1] = (splitmix64_uniform(seed, 2 * b + np.uint64(1)) - 0.5) * 0.6
    x0[b == 0] = 0.0
    return x0


def cartpole_problem(batch=1024, N=101, tf=5.0, b_offset=0, constrained=False, integration=T.RK4,
                     u_bnd=3.0, device=0, lib=None, options=None):
    """C2 (examples/Cartpole.ipynb cells 3-15; docs/src/creating_problems.md:34-53): Q=1e-2 I, R=1e-1, Qf=100 I,
    xf=[0,π,0,0], U0≡0.01.  ``constrained`` adds the notebook's |u|≤3 bound and goal constraint."""
    model = T.Cartpole()
    n, m = model.dims()
    xf = np.array([0.0, math.pi, 0.0, 0.0])
    obj = T.LQRObjective(np.full(n, 1e-2), np.full(m, 1e-1), np.full(n, 100.0), xf, N)
    cons = T.ConstraintList(n, m, N)
    if constrained:
        T.add_constraint(cons, T.BoundConstraint(n, m, u_min=-u_bnd, u_max=u_bnd), range(1, N))
        T.add_constraint(cons, T.GoalConstraint(xf), N)
    prob = T.Problem(model, obj, np.zeros(n), tf, xf=xf, constraints=cons, batch=batch, integration=integration,
                     device=device, lib=lib, options=options)
    prob.set_initial_state(cartpole_x0(batch, b_offset))
    T.initial_controls(prob, np.full(m, 0.01))
    return prob


def quadrotor_x0(batch, b_offset=0, seed=2):
    """C3: r0 = ζ_b ~ U(-1,1)^3, q = identity, v = ω = 0; global b=0 unperturbed."""
    b = _shard(batch, b_offset)
    x0 = np.zeros((batch, 13))
    for j in range(3):
        x0[:, j] = 2.0 * splitmix64_uniform(seed, 3 * b + np.uint64(j)) - 1.0
    x0[b == 0, :3] = 0.0
    x0[:, 3] = 1.0
    return x0


# C5's GoalConstraint acts on position and both velocities (SURVEY.md §8d "optionally inds=[1,2,3,8..13]"): the full
# 13-state goal also pins the quaternion, which RK4 does not keep on the unit sphere, so it is infeasible at 1e-6.
C5_GOAL_INDS = [1, 2, 3, 8, 9, 10, 11, 12, 13]
# C5 solved as ALTRO (AL-iLQR + projected-Newton polish): the polish may linearise up to C5_PN_STEPS + 1 times (Altro's default
# n_steps = 2 allows 3).  The rotor-force clamp max(0, kf u) is a kink of the dynamics; the AL stage parks some controls at
# u ~ 0, where each Newton step can cross the kink for a few of them and the Jacobian has to be rebuilt on the other side:
# 98.4 % of the batch is done within 3 linearisations, the rest needs 4 or 5.
C5_PN_STEPS = 8


def quadrotor_problem(batch=4096, N=201, tf=5.0, b_offset=0, constrained=False, goal_inds=None, u_norm_max=6.0,
                      integration=T.RK4, device=0, lib=None, options=None, quatvec_goal=False):
    """C3/C4 (shape from test/quatcosts.jl:152-168 + src/lie_costs.jl:133-142): point-to-point with QuatLQRCost,
    xf = (r=[2,3,1], yaw 135°), stage Q=diag(1,1,1, 0,0,0,0, .1×6), R=1e-2 I, terminal Q×100, U0≡hover.
    ``constrained`` = C5: GoalConstraint(xf)@N + NormConstraint(‖u‖₂≤6, SecondOrderCone)@1..N-1.
    ``quatvec_goal`` = C5': the terminal attitude is pinned as well, the reference's way — QuatVecEq(qf)@N
    (src/constraints.jl:938-965: the vector part of the NORMALISED quaternion, so RK4's drift off the unit sphere does not make
    it infeasible the way the full 13-state GoalConstraint is)."""
    model = T.Quadrotor()
    n, m = model.dims()
    th = math.radians(135.0) / 2
    xf = np.zeros(n)
    xf[:3] = [2.0, 3.0, 1.0]
    xf[3:7] = [math.cos(th), 0.0, 0.0, math.sin(th)]
    Qd = np.array([1.0, 1, 1, 0, 0, 0, 0, .1, .1, .1, .1, .1, .1])
    Rd = np.full(m, 1e-2)
    uhover = model.hover_control()
    stage = T.QuatLQRCost(Qd, Rd, xf, uhover, w=1.0)
    term = T.QuatLQRCost(100.0 * Qd, Rd, xf, uhover, w=1.0, terminal=True)
    obj = T.Objective(stage, term, N)
    cons = T.ConstraintList(n, m, N)
    if constrained:
        T.add_constraint(cons, T.NormConstraint(n, m, u_norm_max, T.SecondOrderCone(), "control"), range(1, N))
        T.add_constraint(cons, T.GoalConstraint(xf, goal_inds), N)
        if quatvec_goal:
            T.add_constraint(cons, T.QuatVecEq(n, m, xf[3:7]), N)
    prob = T.Problem(model, obj, np.zeros(n), tf, xf=xf, constraints=cons, batch=batch, integration=integration,
                     device=device, lib=lib, options=options)
    prob.set_initial_state(quadrotor_x0(batch, b_offset))
    T.initial_controls(prob, uhover)
    return prob


def quadrotor_zigzag_problem(batch=1, legacy=True, device=0, lib=None, **optkw):
    """examples/Quadrotor.ipynb cells 10-22 (golden G4_quadrotor_altro): the quadrotor flies a zig-zag through two
    waypoints.  N=101, tf=5; x0 r=(0,-10,1), goal r=(0,10,1), identity attitude; nominal cost LQRCost(Q=diag(1e-5 r,
    1e-5 q, 1e-3 v, 1e-3 ω), R=1e-4 I, x_nom=0) on every knot except the waypoint knots 33 / 66 (Q = 1e-3·diag(1e3,1,1,1
    per block), goal r=(±10,0,1)) and the final knot 101 (Q = diag(10,100,10,10 per block)) — a distinct cost per knot
    class, src/objective.jl:27-45; controls bounded 0 ≤ u ≤ 12 on 1..N-1 (cell 18); U0 ≡ 0.5·mass/m (cell 16: the notebook
    calls it hover, it is 1/20 of it); ALTRO options penalty_scaling=100, penalty_initial=0.1 (cell 22).
    ``legacy`` = the stack the notebook was saved with (RK3, stage costs × dt), like G3."""
    model = T.Quadrotor()
    n, m = model.dims()
    N, tf = 101, 5.0

    def build_state(r, q=(1.0, 0.0, 0.0, 0.0), v=(0.0, 0.0, 0.0), w=(0.0, 0.0, 0.0)):
        return np.array([*r, *q, *v, *w], dtype=np.float64)

    def fill_state(a, b, c, d):
        return np.array([a] * 3 + [b] * 4 + [c] * 3 + [d] * 3, dtype=np.float64)

    x0, xf = build_state([0.0, -10.0, 1.0]), build_state([0.0, 10.0, 1.0])
    wpts, times = [[10.0, 0.0, 1.0], [-10.0, 0.0, 1.0], [0.0, 10.0, 1.0]], [33, 66, 101]
    R = np.full(m, 1e-4)
    cost_nom = T.LQRCost(fill_state(1e-5, 1e-5, 1e-3, 1e-3), R, build_state([0.0, 0.0, 0.0]))
    Qw, Qf = fill_state(1e3, 1.0, 1.0, 1.0), fill_state(10.0, 100.0, 10.0, 10.0)
    wcosts = [T.LQRCost(Qf if t == N else 1e-3 * Qw, R, build_state(r), terminal=(t == N)) for r, t in zip(wpts, times)]
    obj = T.Objective([wcosts[times.index(k)] if k in times else cost_nom for k in range(1, N + 1)])
    cons = T.ConstraintList(n, m, N)
    T.add_constraint(cons, T.BoundConstraint(n, m, u_min=0.0, u_max=12.0), range(1, N))
    opts = T.SolverOptions(lib=lib, cost_dt_scaling=1 if legacy else 0, penalty_scaling=100.0, penalty_initial=0.1, **optkw)
    prob = T.Problem(model, obj, x0, tf, xf=xf, constraints=cons, batch=batch, device=device, lib=lib, options=opts,
                     integration=T.RK3 if legacy else T.RK4)
    T.initial_controls(prob, np.full(m, 0.5 * model.mass / m))
    return prob, wpts, times


def quickstart_problem(N=21, tf=3.0, batch=1, device=0, lib=None, options=None):
    """C1 = examples/quickstart.jl:28-59: 2-D double integrator, Goal@N, Circle(0,1,r=.5)@2:N-1,
    Norm-SOC(5,:control)@1:N-1, Bound(|u|≤10)@1:N-1, U0≡0."""
    model = T.DoubleIntegrator(1.0, 2)
    n, m = model.dims()
    xf = np.array([0.0, 2.0, 0.0, 0.0])
    obj = T.LQRObjective(np.ones(n), np.ones(m), np.ones(n) * (N - 1), xf, N)
    cons = T.ConstraintList(n, m, N)
    T.add_constraint(cons, T.GoalConstraint(xf), N)
    T.add_constraint(cons, T.CircleConstraint(n, [0.0], [1.0], [0.5]), range(2, N))
    T.add_constraint(cons, T.NormConstraint(n, m, 5.0, T.SecondOrderCone(), "control"), range(1, N))
    T.add_constraint(cons, T.BoundConstraint(n, m, u_min=-10, u_max=10), range(1, N))
    return T.Problem(model, obj, np.zeros(n), tf, xf=xf, constraints=cons, batch=batch, device=device, lib=lib,
                     options=options)


def algorithmic_bytes_per_iteration(n, m, ne, N, duals=0):
    """FP64 bytes one trajectory-iteration must move (X,U read + X̄,Ū write, Ā/B̄ write+read,
    K/d write+read, dual reads).  C2 = 48 064 B, C3 = 835 408 B, C5 = 851 616 B."""
    return 8 * (2 * (N * n + (N - 1) * m) + 2 * (N - 1) * (ne * ne + ne * m) + 2 * (N - 1) * (m * ne + m)) + 16 * duals
