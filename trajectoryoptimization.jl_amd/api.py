"""Host-side mirror of the reference's solver-facing API, over the C-ABI.

The reference is Julia; Julia is not available in this image, so the host side is written in Python
with the reference's names and argument meaning (``!`` dropped from mutating functions).  Everything
here is descriptor plumbing: all arithmetic runs in ``libtrajopt_hip.so`` on the GPU.

Index conventions follow the reference (Julia): knot ranges and state/control indices are **1-based
and inclusive**.  A knot range may be given as an ``int`` (``N``), a ``(first, last)`` tuple, or a
Python ``range`` whose *values* are the 1-based knots (``range(1, N)`` == Julia ``1:N-1``).

Mirrored reference API (file:line in /root/reference):
  costs        DiagonalCost, QuadraticCost, LQRCost, DiagonalQuatCost, QuatLQRCost
               src/cost_functions.jl:326-346,422-453,532-547; src/lie_costs.jl:34-55,133-142
  objective    Objective, LQRObjective, TrackingObjective          src/objective.jl:27-45,137-196
  cones        Equality/ZeroCone, Inequality/NegativeOrthant, SecondOrderCone, projection, ∇projection,
               ∇²projection, cone_status                           src/cones.jl
  constraints  GoalConstraint, BoundConstraint, NormConstraint, CircleConstraint, SphereConstraint,
               CollisionConstraint, QuatVecEq, LinearConstraint, ConstraintList, add_constraint!   src/constraints.jl, src/constraint_list.jl
  problem      Problem, rollout!, cost, states, controls, initial_controls!, initial_states!,
               set_initial_state!, set_goal_state!, get_* getters   src/problem.jl
  solvers      iLQRSolver, ALSolver (=Altro's AL-iLQR), SolverOptions, solve!, iterations, status,
               max_violation (Altro.jl, out of tree; examples/Cartpole.ipynb cells 17-25)
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi as capi
from ._capi import (ArgumentError, DimensionMismatch, ConstraintDesc, CostDesc, ProblemDesc,
                    SolverOpts, SolveStats, UnsupportedError)

__all__ = [
    "clear_goal_state_batch", "set_constraint_params_batch", "InfeasibleModel", "InfeasibleConstraint", "InfeasibleProblem", "infeasible_controls",
    "DoubleIntegrator", "Cartpole", "Quadrotor", "DiscreteMap", "LinearMap", "ModelVector", "HybridDoubleIntegrator", "pad_cost", "dims", "RK4", "RK3", "Euler",
    "DiagonalCost", "QuadraticCost", "LQRCost", "DiagonalQuatCost", "ErrorQuadratic", "QuatLQRCost",
    "Objective", "LQRObjective", "TrackingObjective",
    "Equality", "ZeroCone", "Inequality", "NegativeOrthant", "SecondOrderCone", "PositiveOrthant", "IdentityCone",
    "projection", "grad_projection", "hess_projection", "cone_status", "dualcone",
    "GoalConstraint", "BoundConstraint", "NormConstraint", "CircleConstraint", "SphereConstraint", "CollisionConstraint", "QuatVecEq",
    "LinearConstraint", "StateBound", "ControlBound", "IndexedConstraint", "change_dimension",
    "ConstraintList", "add_constraint", "num_constraints", "constraint_hessians",
    "KnotPoint", "Problem", "rollout", "cost", "states", "controls", "initial_controls", "initial_states",
    "set_initial_state", "set_goal_state", "update_trajectory", "get_constraints", "get_objective", "get_model",
    "get_initial_state", "get_final_state", "get_trajectory", "gettimes",
    "SolverOptions", "iLQRSolver", "ALSolver", "ALTROSolver", "ProjectedNewtonSolver", "dynamics_defect", "solve", "SolvePipeline", "iterations", "status", "max_violation",
    "evaluate_constraints", "constraint_jacobians", "sense", "upper_bound", "lower_bound", "is_bound",
    "DimensionMismatch", "ArgumentError", "UnsupportedError",
]

RK4, RK3, Euler = capi.RK4, capi.RK3, capi.EULER


def _vec(x, n=None, name="vector"):
    a = np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel()
    if n is not None and a.size != n:
        raise DimensionMismatch(f"{name} has length {a.size}, expected {n}")
    return a


def _diag_or_vec(Q):
    """Accept Diagonal-as-vector or a square matrix; return (is_diag, array)."""
    a = np.asarray(Q, dtype=np.float64)
    if a.ndim <= 1:
        return True, np.atleast_1d(a).ravel()
    if a.shape[0] != a.shape[1]:
        raise DimensionMismatch("weight matrix must be square")
    if np.count_nonzero(a - np.diag(np.diag(a))) == 0:
        return True, np.diag(a).copy()
    return False, a.copy()


# --------------------------------------------------------------------------------------------- models
class _Model:
    model_id = -1
    n = m = 0

    def params(self):
        raise NotImplementedError

    def dims(self):
        return self.n, self.m

    @property
    def errstate_dim(self):
        return self.n


def _same_model(a, b):
    return type(a) is type(b) and a.dims() == b.dims() and getattr(a, "params", lambda: None)() == getattr(b, "params", lambda: None)()


class DiscreteMap(_Model):
    """A discrete map x⁺ = g(x, u) from (n, m) to ``output_dim`` states — the jump map of a hybrid model vector
    (test/hybrid_dynamics_model.jl:14-38).  On its own it is host-side bookkeeping (``dims(models)``, ``ConstraintList(models)``,
    the ``Problem`` validation); the map itself is compiled into the library as part of a hybrid model (``HybridDoubleIntegrator``),
    which hands out its jump map through ``.models(N)``."""

    def __init__(self, n, m, output_dim, hybrid=None):
        self.n, self.m, self._ny = int(n), int(m), int(output_dim)
        self.hybrid = hybrid  # the compiled-in hybrid model this map belongs to (None: bookkeeping only)

    @property
    def output_dim(self):
        return self._ny

    def params(self):
        return [float(self.n), float(self.m), float(self._ny)]


class LinearMap(DiscreteMap):
    """The discrete map x⁺ = A x + B u from (n, m) to ``output_dim = size(A, 1)`` states: the kind of jump map that connects the
    phases of a model vector (test/hybrid_dynamics_model.jl:31-33 is one).  A step of ``TO_MODEL_VECTOR``."""

    def __init__(self, A, B):
        A, B = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.atleast_2d(np.asarray(B, dtype=np.float64))
        if A.shape[0] != B.shape[0]:
            raise DimensionMismatch("LinearMap: A and B need the same number of rows")
        super().__init__(A.shape[1], B.shape[1], A.shape[0])
        self.A, self.B = A, B

    def params(self):
        return list(self.A.flatten(order="F")) + list(self.B.flatten(order="F"))


class ModelVector(_Model):
    """``Problem(models::Vector{<:DiscreteDynamics}, ...)`` in general (src/problem.jl:36-73, src/dynamics.jl:15-31; TO_MODEL_VECTOR):
    one model per time step — DoubleIntegrator (D = 1, 2, 3), Cartpole, LinearMap — whose dimensions chain.  The library stores
    the trajectory at (6, 3) with the narrower knots zero-padded; ``Problem`` lowers costs and constraints onto the padding
    exactly as for ``HybridDoubleIntegrator``."""
    model_id = capi.MODEL_VECTOR
    n, m = capi.TO_VECTOR_N, capi.TO_VECTOR_M

    def __init__(self, models):
        self.steps = list(models)
        for mod in self.steps:
            if not isinstance(mod, (DoubleIntegrator, Cartpole, LinearMap)):
                raise UnsupportedError(f"model vector step of type {type(mod).__name__}: the compiled-in step models are DoubleIntegrator, "
                                       "Cartpole and LinearMap")
            if mod.n > self.n or mod.m > self.m or getattr(mod, "output_dim", mod.n) > self.n:
                raise UnsupportedError("model vector: step dimensions outside (6, 3)")

    def params(self):
        return []

    def _step_descs(self):
        arr = (capi.StepModel * len(self.steps))()
        for d, mod in zip(arr, self.steps):
            d.kind = (capi.STEP_LINEAR_MAP if isinstance(mod, LinearMap) else capi.STEP_CARTPOLE if isinstance(mod, Cartpole)
                      else capi.STEP_DOUBLE_INTEGRATOR)
            d.n, d.m, d.n_out = mod.n, mod.m, getattr(mod, "output_dim", mod.n)
            p = mod.params() if not isinstance(mod, DoubleIntegrator) else [mod.mass]
            d.params[: len(p)] = p
        return arr


def dims(models):
    """RD.dims(models::Vector{<:DiscreteDynamics}) (src/dynamics.jl:15-31): state / control dimension at each of the N knots of
    an (N-1)-vector of models; the last state dimension is the last model's output dimension, the last control dimension its
    control dimension.  Raises DimensionMismatch when consecutive models do not chain."""
    models = list(models)
    nx = [mod.n for mod in models] + [getattr(models[-1], "output_dim", models[-1].n)]
    nu = [mod.m for mod in models] + [models[-1].m]
    for k, mod in enumerate(models, start=1):
        ny = getattr(mod, "output_dim", mod.n)
        if nx[k] != ny:
            raise DimensionMismatch(f"Model mismatch at time step {k}. Model {k} has an output dimension of {ny} but model {k + 1} "
                                    f"has a state dimension of {nx[k]}.")
    return nx, nu


class DoubleIntegrator(_Model):
    """examples/quickstart.jl:11-23 generalised to D dimensions (D=2 there)."""
    model_id = capi.MODEL_DOUBLE_INTEGRATOR

    def __init__(self, mass=1.0, D=2):
        if D not in (1, 2, 3):
            raise ArgumentError("DoubleIntegrator dimension must be 1, 2 or 3")
        self.mass, self.D = float(mass), int(D)
        self.n, self.m = 2 * D, D

    def params(self):
        return [self.mass, float(self.D)]


class HybridDoubleIntegrator(_Model):
    """The model vector of test/hybrid_dynamics_model.jl:14-52 as ONE compiled-in model (TO_MODEL_HYBRID_DOUBLE_INTEGRATOR): a 2-D
    double integrator (4, 2) for ``steps_2d`` time steps, the jump map (4, 2) -> 2, x⁺ = [(x₃ + x₄)/2, (u₁ + u₂)/2], then a 1-D
    double integrator (2, 1).  ``models(N)`` is the reference-shaped (N-1)-vector of per-step models to build ``ConstraintList``,
    ``Objective`` and ``Problem`` from; ``Problem`` recognises it, stores states / controls zero-padded at (4, 2) and lowers the
    per-knot costs and constraints onto the padded vectors (``pad_cost``, ``IndexedConstraint``)."""
    model_id = capi.MODEL_HYBRID_DOUBLE_INTEGRATOR
    n, m = 4, 2

    def __init__(self, mass=1.0, steps_2d=5):
        self.mass, self.steps_2d = float(mass), int(steps_2d)
        if self.steps_2d < 1:
            raise ArgumentError("HybridDoubleIntegrator needs at least one 2-D time step")
        self._first, self._second = DoubleIntegrator(self.mass, 2), DoubleIntegrator(self.mass, 1)
        self.jump = DiscreteMap(4, 2, 2, hybrid=self)

    def params(self):
        return [self.mass, float(self.steps_2d)]

    def models(self, N):
        """One model per time step for an N-knot horizon (N - 1 entries)."""
        rest = N - 2 - self.steps_2d
        if rest < 0:
            raise ArgumentError("horizon too short for the jump map: N >= steps_2d + 2")
        return [self._first] * self.steps_2d + [self.jump] + [self._second] * rest

    @staticmethod
    def match(models):
        """The HybridDoubleIntegrator whose ``models(N)`` equals this model vector, or None."""
        jumps = [k for k, mod in enumerate(models) if isinstance(mod, DiscreteMap)]
        if len(jumps) != 1 or not isinstance(models[jumps[0]].hybrid, HybridDoubleIntegrator):
            return None
        hyb = models[jumps[0]].hybrid
        want = hyb.models(len(models) + 1) if len(models) >= hyb.steps_2d + 1 else None
        if want is None or jumps[0] != hyb.steps_2d:
            return None
        return hyb if all(_same_model(a, b) for a, b in zip(models, want)) else None


def pad_cost(cost, n, m):
    """The cost function of a knot with live dimensions (cost.n, cost.m) on the zero-padded (n, m) vectors of a hybrid model:
    nothing on padded states; a padded control gets R = 1 and no linear term, so its feedforward and feedback gains are exactly
    zero and it stays at 0 (a zero R entry would make Quu singular)."""
    n0, m0 = cost.n, cost.m
    if (n0, m0) == (n, m):
        return cost
    if n0 > n or m0 > m:
        raise DimensionMismatch("cost dimensions exceed the padded model dimensions")
    if cost.kind not in (capi.COST_DIAGONAL, capi.COST_QUADRATIC):
        raise UnsupportedError("only DiagonalCost / QuadraticCost can be padded onto a hybrid model's storage dimensions")
    q, r = np.r_[cost.q, np.zeros(n - n0)], np.r_[cost.r, np.zeros(m - m0)]
    if cost.kind == capi.COST_DIAGONAL:
        return DiagonalCost(np.r_[cost.Q, np.zeros(n - n0)], np.r_[cost.R, np.ones(m - m0)], q, r, cost.c, terminal=cost.terminal)
    Q, R, H = np.zeros((n, n)), np.eye(m), np.zeros((m, n))
    Q[:n0, :n0], R[:m0, :m0], H[:m0, :n0] = cost.Q, cost.R, cost.H
    return QuadraticCost(Q, R, H, q, r, cost.c, terminal=cost.terminal)


class Cartpole(_Model):
    """RobotZoo.Cartpole, restated at docs/src/model.md:20-51."""
    model_id = capi.MODEL_CARTPOLE
    n, m = 4, 1

    def __init__(self, mc=1.0, mp=0.2, l=0.5, g=9.81):
        self.mc, self.mp, self.l, self.g = float(mc), float(mp), float(l), float(g)

    def params(self):
        return [self.mc, self.mp, self.l, self.g]


class InfeasibleModel(_Model):
    """Altro's ``InfeasibleModel`` — the state augmentation of ALTRO's infeasible start (TO_MODEL_INFEASIBLE): the base model plus
    one slack control per state, x⁺ = f_d(x, u[1:m0]) + u[m0+1 : m0+n].  With the slacks of ``infeasible_controls`` ANY state guess
    (``initial_states!``, src/problem.jl:242-253) is dynamically feasible.  The reference carries the ``change_dimension`` family for
    exactly this (src/constraints.jl:820-936, src/constraint_list.jl:208-217, src/cost_functions.jl:391-401).  Bases: the 1-D / 2-D
    double integrator and the Cartpole (a Quadrotor base, (13, 17), exceeds the library's control dimension)."""
    model_id = capi.MODEL_INFEASIBLE

    def __init__(self, model):
        if not isinstance(model, (DoubleIntegrator, Cartpole)) or model.m + model.n > capi.TO_MAX_M:
            raise UnsupportedError("InfeasibleModel: the base must be a 1-D / 2-D double integrator or a Cartpole")
        self.model = model
        self.n, self.m = model.n, model.m + model.n

    def params(self):
        p = list(self.model.params()) + [0.0] * 16
        p[15] = float(self.model.model_id)
        return p[:16]


class Quadrotor(_Model):
    """RobotZoo.Quadrotor / examples/Quadrotor.ipynb cells 4 and 8: ``Quadrotor{R} <: RigidBody{R}``.  ``rotation`` is the
    attitude representation R of the state (cell 5: QuatRotation, MRP or RodriguesParam; src/lie_costs.jl:1-3):
    "quat" -> x = [r, q(w,x,y,z), v, ω], n = 13 (the default, ``Quadrotor()``); "mrp" / "rp" -> x = [r, p(3), v, ω], n = 12.
    The error state has 12 entries either way."""
    model_id = capi.MODEL_QUADROTOR
    m = 4
    ROTATIONS = {"quat": 0, "QuatRotation": 0, "UnitQuaternion": 0, "mrp": 1, "MRP": 1, "rp": 2, "RodriguesParam": 2}

    def __init__(self, mass=0.5, J=(0.0023, 0.0023, 0.004), gravity=(0.0, 0.0, -9.81),
                 motor_dist=0.1750, kf=1.0, km=0.0245, rotation="quat"):
        self.mass, self.J, self.gravity = float(mass), tuple(map(float, J)), tuple(map(float, gravity))
        self.motor_dist, self.kf, self.km = float(motor_dist), float(kf), float(km)
        if rotation not in self.ROTATIONS:
            raise ArgumentError(f"unknown rotation {rotation!r}: one of quat, mrp, rp")
        self.rotation = self.ROTATIONS[rotation]
        self.n = 13 if self.rotation == 0 else 12

    def params(self):
        return [self.mass, *self.J, *self.gravity, self.motor_dist, self.kf, self.km, float(self.rotation)]

    @property
    def errstate_dim(self):
        return 12

    def hover_control(self):
        """``zeros(model)[2]``: thrust that cancels gravity, -g_z*m/4 per motor."""
        return np.full(4, -self.gravity[2] * self.mass / 4.0 / self.kf)

    def build_state(self, r, q=None, v=(0.0, 0.0, 0.0), w=(0.0, 0.0, 0.0)):
        """RobotDynamics.build_state(model, r, q, v, ω): ``q`` is a unit quaternion (w, x, y, z), stored as such or converted to
        the model's three-parameter attitude (MRP p = q_v/(1+q_w), RodriguesParam g = q_v/q_w)."""
        q = np.array([1.0, 0.0, 0.0, 0.0]) if q is None else _vec(q, 4, "q")
        q = q / np.linalg.norm(q)
        att = q if self.rotation == 0 else (q[1:] / (1.0 + q[0]) if self.rotation == 1 else q[1:] / q[0])
        return np.concatenate([_vec(r, 3, "r"), att, _vec(v, 3, "v"), _vec(w, 3, "ω")])


# --------------------------------------------------------------------------------------------- costs
class QuadraticCostFunction:
    """½xᵀQx + ½uᵀRu + uᵀHx + qᵀx + rᵀu + c  (src/cost_functions.jl:19-34)."""
    kind = capi.COST_QUADRATIC

    def __init__(self, Q, R, H=None, q=None, r=None, c=0.0, terminal=False):
        self.Q = np.asarray(Q, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.n, self.m = self.Q.shape[0], self.R.shape[0]
        self.H = np.zeros((self.m, self.n)) if H is None else np.asarray(H, dtype=np.float64)
        if self.H.shape != (self.m, self.n):
            raise DimensionMismatch("H must be m x n")
        self.q = np.zeros(self.n) if q is None else _vec(q, self.n, "q")
        self.r = np.zeros(self.m) if r is None else _vec(r, self.m, "r")
        self.c, self.terminal = float(c), bool(terminal)

    def state_dim(self):
        return self.n

    def control_dim(self):
        return self.m

    def is_diag(self):
        return self.kind != capi.COST_QUADRATIC

    def _desc(self):
        d = CostDesc()
        n, m = self.n, self.m
        if n > capi.TO_MAX_N or m > capi.TO_MAX_M:
            raise DimensionMismatch("cost dimensions exceed TO_MAX_N/TO_MAX_M")
        d.kind, d.terminal = self.kind, int(self.terminal)
        if self.kind == capi.COST_QUADRATIC:
            d.Q[: n * n] = list(self.Q.ravel(order="F"))
            d.R[: m * m] = list(self.R.ravel(order="F"))
            d.H[: m * n] = list(self.H.ravel(order="F"))
        else:
            d.Q[:n] = list(self.Q)
            d.R[:m] = list(self.R)
        d.q[:n] = list(self.q)
        d.r[:m] = list(self.r)
        d.c = self.c
        d.w = getattr(self, "w", 0.0)
        d.q_ref[:] = list(getattr(self, "q_ref", np.array([1.0, 0, 0, 0])))
        d.q_ind[:] = list(getattr(self, "q_ind", (4, 5, 6, 7)))
        return d


class QuadraticCost(QuadraticCostFunction):
    """src/cost_functions.jl:422-453."""


class DiagonalCost(QuadraticCostFunction):
    """src/cost_functions.jl:326-346.  Q, R given as diagonals (vectors) or diagonal matrices."""
    kind = capi.COST_DIAGONAL

    def __init__(self, Q, R, q=None, r=None, c=0.0, terminal=False):
        dq, Qd = _diag_or_vec(Q)
        dr, Rd = _diag_or_vec(R)
        if not (dq and dr):
            raise ArgumentError("DiagonalCost needs diagonal Q and R")
        self.Q, self.R = Qd, Rd
        self.n, self.m = Qd.size, Rd.size
        self.H = np.zeros((self.m, self.n))
        self.q = np.zeros(self.n) if q is None else _vec(q, self.n, "q")
        self.r = np.zeros(self.m) if r is None else _vec(r, self.m, "r")
        self.c, self.terminal = float(c), bool(terminal)


def LQRCost(Q, R, xf, uf=None, terminal=False):
    """½(x-xf)ᵀQ(x-xf) + ½(u-uf)ᵀR(u-uf)  (src/cost_functions.jl:532-547)."""
    dq, Qa = _diag_or_vec(Q)
    dr, Ra = _diag_or_vec(R)
    n, m = Qa.shape[0], Ra.shape[0]
    xf = _vec(xf, n, "xf")
    uf = np.zeros(m) if uf is None else _vec(uf, m, "uf")
    if dq and dr:
        q, r = -Qa * xf, -Ra * uf
        c = 0.5 * xf @ (Qa * xf) + 0.5 * uf @ (Ra * uf)
        return DiagonalCost(Qa, Ra, q, r, c, terminal=terminal)
    Qm = np.diag(Qa) if dq else Qa
    Rm = np.diag(Ra) if dr else Ra
    return QuadraticCost(Qm, Rm, None, -Qm @ xf, -Rm @ uf, 0.5 * xf @ Qm @ xf + 0.5 * uf @ Rm @ uf, terminal=terminal)


class DiagonalQuatCost(DiagonalCost):
    """src/lie_costs.jl:34-55: diagonal quadratic + w·min(1 ± q_refᵀ x[q_ind])."""
    kind = capi.COST_DIAGONAL_QUAT

    def __init__(self, Q, R, q=None, r=None, c=0.0, w=1.0, q_ref=(1.0, 0, 0, 0), q_ind=(4, 5, 6, 7), terminal=False):
        super().__init__(Q, R, q, r, c, terminal)
        self.w = float(w)
        self.q_ref = _vec(q_ref, 4, "q_ref")
        if len(q_ind) != 4:
            raise AssertionError("quat_ind argument must be of length 4")
        self.q_ind = tuple(int(i) for i in q_ind)


class ErrorQuadratic(QuadraticCostFunction):
    """½ (x ⊖ x_ref)ᵀ Q (x ⊖ x_ref) + c + ½uᵀRu + rᵀu with the Cayley-map error state of a rigid body
    (``ErrorQuadratic{Rot}``, src/lie_costs.jl:178-241; Rot = the model's attitude representation: QuatRotation, MRP or
    RodriguesParam).  ``Q`` is the 12-vector of error-state weights, or — on a quaternion state — a 13-vector / 13x13
    diagonal whose 4th entry is dropped like the reference constructor does (:226-229); ``u_ref`` folds into r and c (:230-231)."""
    kind = capi.COST_ERROR_QUADRATIC

    def __init__(self, model, Q, R, x_ref, u_ref=None, r=None, c=0.0, q_ind=(4, 5, 6, 7), terminal=False):
        n, m = model.dims()
        rot = getattr(model, "rotation", None)
        if rot is None or n not in (12, 13):
            raise ArgumentError("ErrorQuadratic needs a rigid-body model (13 states with a quaternion, 12 with MRP / RodriguesParam)")
        dq, Qd = _diag_or_vec(Q)
        dr, Rd = _diag_or_vec(R)
        if not (dq and dr):
            raise ArgumentError("ErrorQuadratic needs diagonal Q and R")
        if rot == 0 and Qd.size == 13:  # Rot <: QuatRotation && size(Q,1) == size(x_ref,1): drop the 4th entry (src/lie_costs.jl:226-229)
            Qd = np.delete(Qd, 3)
        if Qd.size != 12 or Rd.size != m:
            raise DimensionMismatch("ErrorQuadratic: Q must have 12 entries (or 13 on a quaternion state), R m entries")
        self.n, self.m, self.rotation = n, m, rot
        self.Q, self.R = Qd, Rd
        self.H = np.zeros((m, n))
        self.x_ref = _vec(x_ref, n, "x_ref")
        u_ref = np.zeros(m) if u_ref is None else _vec(u_ref, m, "u_ref")
        self.r = (np.zeros(m) if r is None else _vec(r, m, "r")) - Rd * u_ref
        self.c = float(c) + 0.5 * u_ref @ (Rd * u_ref)
        if rot == 0 and tuple(int(i) for i in q_ind) != (4, 5, 6, 7):
            raise ArgumentError("ErrorQuadratic: q_ind must be 4:7 (the rigid-body state layout)")
        self.q_ind = (4, 5, 6, 7)
        self.terminal = bool(terminal)

    @property
    def q(self):  # the descriptor's q slot carries x_ref for this kind (include/trajopt_hip.h)
        return self.x_ref

    def _desc(self):
        d = CostDesc()
        d.kind, d.terminal = self.kind, int(self.terminal)
        d.Q[:12] = list(self.Q)
        d.R[: self.m] = list(self.R)
        d.q[: self.n] = list(self.x_ref)
        d.r[: self.m] = list(self.r)
        d.c = self.c
        d.w = float(self.rotation)  # to_rotation of the state: ErrorQuadratic{QuatRotation} / {MRP} / {RodriguesParam}
        d.q_ref[:] = [1.0, 0.0, 0.0, 0.0]
        d.q_ind[:] = list(self.q_ind)
        return d


def QuatLQRCost(Q, R, xf, uf=None, w=1.0, quat_ind=(4, 5, 6, 7), terminal=False):
    """src/lie_costs.jl:133-142."""
    _, Qd = _diag_or_vec(Q)
    _, Rd = _diag_or_vec(R)
    xf = _vec(xf, Qd.size, "xf")
    uf = np.zeros(Rd.size) if uf is None else _vec(uf, Rd.size, "uf")
    if len(quat_ind) != 4:
        raise AssertionError("quat_ind argument must be of length 4")
    q_ref = xf[[i - 1 for i in quat_ind]]
    c = 0.5 * xf @ (Qd * xf) + 0.5 * uf @ (Rd * uf)
    return DiagonalQuatCost(Qd, Rd, -Qd * xf, -Rd * uf, c, w, q_ref, quat_ind, terminal=terminal)


class Objective:
    """Vector of N cost functions (src/objective.jl:27-45)."""

    def __init__(self, costs, terminal_cost=None, N=None):
        if isinstance(terminal_cost, (int, np.integer)):  # Objective(cost, N)
            terminal_cost, N = None, int(terminal_cost)
        if terminal_cost is not None:  # Objective(cost, cost_terminal, N)  src/objective.jl:74-77
            self.cost = [costs] * (N - 1) + [terminal_cost]
        elif isinstance(costs, QuadraticCostFunction):  # Objective(cost, N)  src/objective.jl:66-72
            self.cost = [costs] * N
        else:
            self.cost = list(costs)
        self.J = np.zeros(len(self.cost))  # (dimensions may differ from knot to knot: hybrid model vectors, src/dynamics.jl:15-31)

    def __len__(self):
        return len(self.cost)

    def __getitem__(self, k):
        return self.cost[k]

    def dims(self):
        return self.cost[0].n, self.cost[0].m

    def knot_dims(self):
        """RD.dims(obj): state / control dimension of the cost function at every knot (src/objective.jl:85-87)."""
        return [c.n for c in self.cost], [c.m for c in self.cost]

    def _descs(self):
        """Deduplicate by identity -> (list of CostDesc, cost_index[N])."""
        uniq, index = [], []
        for c in self.cost:
            for i, u in enumerate(uniq):
                if u is c:
                    index.append(i)
                    break
            else:
                uniq.append(c)
                index.append(len(uniq) - 1)
        return uniq, index


def LQRObjective(Q, R, Qf, xf, N, uf=None, checks=True):
    """src/objective.jl:137-183: stage LQRCost(Q,R,xf,uf); terminal uses Qf, the SAME R and r, cf=½xfᵀQf xf."""
    dq, Qa = _diag_or_vec(Q)
    dr, Ra = _diag_or_vec(R)
    df, Qfa = _diag_or_vec(Qf)
    n, m = Qa.shape[0], Ra.shape[0]
    xf = _vec(xf)
    assert Qa.shape[0] == xf.size and Qfa.shape[0] == xf.size
    uf = np.zeros(m) if uf is None else _vec(uf)
    assert Ra.shape[0] == uf.size
    if dq and dr and df:
        q, r = -Qa * xf, -Ra * uf
        c = 0.5 * xf @ (Qa * xf) + 0.5 * uf @ (Ra * uf)
        stage = DiagonalCost(Qa, Ra, q, r, c, terminal=False)
        term = DiagonalCost(Qfa, Ra, -Qfa * xf, r, 0.5 * xf @ (Qfa * xf), terminal=True)
    else:
        Qm = np.diag(Qa) if dq else Qa
        Rm = np.diag(Ra) if dr else Ra
        Qfm = np.diag(Qfa) if df else Qfa
        r = -Rm @ uf
        stage = QuadraticCost(Qm, Rm, None, -Qm @ xf, r, 0.5 * xf @ Qm @ xf + 0.5 * uf @ Rm @ uf)
        term = QuadraticCost(Qfm, Rm, None, -Qfm @ xf, r, 0.5 * xf @ Qfm @ xf, terminal=True)
    return Objective(stage, term, N)


def TrackingObjective(Q, R, X, U, Qf=None):
    """src/objective.jl:190-196: per-knot LQRCost tracking the reference (X[k], U[k])."""
    X = np.asarray(X, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    N = X.shape[1]
    costs = [LQRCost(Q, R, X[:, k], U[:, k] if k < U.shape[1] else None) for k in range(N)]
    costs[-1] = LQRCost(Q if Qf is None else Qf, R, X[:, -1], terminal=True)
    return Objective(costs)


# --------------------------------------------------------------------------------------------- cones
class _Cone:
    code = -1

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self))

    def __repr__(self):
        return type(self).__name__ + "()"


class ZeroCone(_Cone):
    code = capi.CONE_ZERO


class NegativeOrthant(_Cone):
    code = capi.CONE_NEGATIVE_ORTHANT


class SecondOrderCone(_Cone):
    code = capi.CONE_SECOND_ORDER


class PositiveOrthant(_Cone):
    code = capi.CONE_POSITIVE_ORTHANT


class IdentityCone(_Cone):
    code = capi.CONE_IDENTITY


Equality = ZeroCone
Inequality = NegativeOrthant


def dualcone(cone):
    """src/cones.jl:65-69."""
    return {IdentityCone: ZeroCone, ZeroCone: IdentityCone}.get(type(cone), type(cone))()


def _cone_call(name, cone, x, b=None, lib=None, device=0):
    lib = lib or capi.load_hip_library()
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    single = x.ndim == 1
    xs = x.reshape(1, -1) if single else x
    count, dim = xs.shape
    xs = np.ascontiguousarray(xs)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    if name == "cone_projection":
        out = np.empty_like(xs)
        st = np.zeros(count, dtype=np.int32)
        lib.call(name, device, cone.code, dim, count, pd(xs), pd(out), st.ctypes.data_as(C.POINTER(C.c_int32)))
        return (out[0], int(st[0])) if single else (out, st)
    out = np.empty((count, dim, dim))
    if name == "cone_projection_jacobian":
        lib.call(name, device, cone.code, dim, count, pd(xs), pd(out))
    else:
        bs = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(count, dim))
        lib.call(name, device, cone.code, dim, count, pd(xs), pd(bs), pd(out))
    out = out.transpose(0, 2, 1)  # column-major blocks -> numpy [count, row, col]
    return out[0] if single else out


def projection(cone, x, lib=None):
    """Π_K(x)  (src/cones.jl:71-127).  x: [dim] or [count, dim]."""
    return _cone_call("cone_projection", cone, x, lib=lib)[0]


def cone_status(cone, x, lib=None):
    """:below / :in / :outside  (src/cones.jl:278-291)."""
    st = _cone_call("cone_projection", cone, x, lib=lib)[1]
    names = {0: "below", 1: "in", 2: "outside"}
    return names[st] if np.isscalar(st) or isinstance(st, int) else [names[int(s)] for s in st]


def grad_projection(cone, x, lib=None):
    """∇projection!  (src/cones.jl:129-188)."""
    return _cone_call("cone_projection_jacobian", cone, x, lib=lib)


def hess_projection(cone, x, b, lib=None):
    """∇²projection!: Hessian of bᵀΠ(x)  (src/cones.jl:201-276)."""
    return _cone_call("cone_projection_hessian", cone, x, b, lib=lib)


# --------------------------------------------------------------------------------------------- constraints
class AbstractConstraint:
    kind = -1
    state_only = False

    def sense(self):
        return self._sense

    def output_dim(self):
        return self.p

    def check_dims(self, n, m):
        """src/abstract_constraint.jl:146-149."""
        if self.state_only:
            return self.n == n
        return self.n == n and self.m == m

    def _fill(self, d):
        raise NotImplementedError

    def _desc(self, k1, k2):
        d = ConstraintDesc()
        d.kind, d.sense, d.k_first, d.k_last, d.p = self.kind, self._sense.code, k1, k2, self.p
        inds, params = self._fill()
        if len(inds) > capi.TO_MAX_CON_INDS or len(params) > capi.TO_MAX_CON_PARAMS:
            raise DimensionMismatch("constraint exceeds TO_MAX_CON_INDS/TO_MAX_CON_PARAMS")
        d.n_inds, d.n_params = len(inds), len(params)
        d.inds[: len(inds)] = [int(i) for i in inds]
        d.params[: len(params)] = [float(v) for v in params]
        return d


def sense(con):
    return con.sense()


def upper_bound(con):
    """src/abstract_constraint.jl:107-123; an IndexedConstraint forwards to the constraint it wraps (src/constraints.jl:877-879)."""
    if isinstance(con, IndexedConstraint):
        return upper_bound(con.con)
    s = con.sense()
    if isinstance(con, BoundConstraint):
        return con.z_max
    return np.full(con.p, 0.0 if not isinstance(s, SecondOrderCone) else np.inf)


def lower_bound(con):
    if isinstance(con, IndexedConstraint):
        return lower_bound(con.con)
    s = con.sense()
    if isinstance(con, BoundConstraint):
        return con.z_min
    return np.full(con.p, 0.0 if isinstance(s, ZeroCone) else -np.inf)


def is_bound(con):
    if isinstance(con, IndexedConstraint):
        return is_bound(con.con)
    return isinstance(con, (GoalConstraint, BoundConstraint))


class GoalConstraint(AbstractConstraint):
    """x[inds] − xf[inds] = 0  (src/constraints.jl:22-87)."""
    kind = capi.CON_GOAL
    state_only = True

    def __init__(self, xf, inds=None):
        xf = _vec(xf)
        self.n = xf.size
        self.inds = list(range(1, self.n + 1)) if inds is None else [int(i) for i in inds]
        self.xf = xf[[i - 1 for i in self.inds]].copy()
        self.p = len(self.inds)
        self._sense = Equality()

    def _fill(self):
        return self.inds, list(self.xf)


class BoundConstraint(AbstractConstraint):
    """x_min ≤ x ≤ x_max, u_min ≤ u ≤ u_max  (src/constraints.jl:644-783)."""
    kind = capi.CON_BOUND

    def __init__(self, n, m, x_max=np.inf, x_min=-np.inf, u_max=np.inf, u_min=-np.inf):
        self.n, self.m = int(n), int(m)

        def check(k, hi, lo):  # checkBounds, src/constraints.jl:708-719
            hi = np.full(k, float(hi)) if np.isscalar(hi) else _vec(hi, k, "upper bound")
            lo = np.full(k, float(lo)) if np.isscalar(lo) else _vec(lo, k, "lower bound")
            if not np.all(hi >= lo):
                raise ArgumentError("Upper bounds must be greater than or equal to lower bounds")
            return hi, lo

        xh, xl = check(n, x_max, x_min)
        uh, ul = check(m, u_max, u_min)
        self.z_max = np.concatenate([xh, uh])
        self.z_min = np.concatenate([xl, ul])
        b = np.concatenate([-self.z_max, self.z_min])
        self.inds = [i + 1 for i in np.flatnonzero(np.isfinite(b))]
        self.p = len(self.inds)
        self._sense = Inequality()

    def _fill(self):
        return [], list(self.z_max) + list(self.z_min)


class NormConstraint(AbstractConstraint):
    """‖z[inds]‖ (=,≤) a, or the second-order-cone form [z[inds]; a] ∈ SOC  (src/constraints.jl:438-521)."""
    kind = capi.CON_NORM

    def __init__(self, n, m, val, sense, inds=None):
        self.n, self.m = int(n), int(m)
        if isinstance(inds, str):
            inds = {"state": range(1, n + 1), "control": range(n + 1, n + m + 1)}[inds]
        self.inds = list(range(1, n + m + 1)) if inds is None else [int(i) for i in inds]
        if not val >= 0:
            raise AssertionError("Value must be greater than or equal to zero")
        self.val = float(val)
        self._sense = sense
        self.p = len(self.inds) + 1 if isinstance(sense, SecondOrderCone) else 1

    def _fill(self):
        return self.inds, [self.val]


class CircleConstraint(AbstractConstraint):
    """(x−xc)²+(y−yc)² ≥ r²  (src/constraints.jl:168-233)."""
    kind = capi.CON_CIRCLE
    state_only = True

    def __init__(self, n, xc, yc, radius, xi=1, yi=2):
        self.n = int(n)
        self.x, self.y, self.radius = _vec(xc), _vec(yc), _vec(radius)
        if not (self.x.size == self.y.size == self.radius.size):
            raise AssertionError("Lengths of xc, yc, and radius must be equal.")
        self.xi, self.yi = int(xi), int(yi)
        self.p = self.x.size
        self._sense = Inequality()

    def _fill(self):
        return [self.xi, self.yi], list(self.x) + list(self.y) + list(self.radius)


class SphereConstraint(AbstractConstraint):
    """src/constraints.jl:249-326."""
    kind = capi.CON_SPHERE
    state_only = True

    def __init__(self, n, xc, yc, zc, radius, xi=1, yi=2, zi=3):
        self.n = int(n)
        self.x, self.y, self.z, self.radius = _vec(xc), _vec(yc), _vec(zc), _vec(radius)
        if not (self.x.size == self.y.size == self.z.size == self.radius.size):
            raise AssertionError("Lengths of xc, yc, zc, and radius must be equal.")
        self.xi, self.yi, self.zi = int(xi), int(yi), int(zi)
        self.p = self.x.size
        self._sense = Inequality()

    def _fill(self):
        return [self.xi, self.yi, self.zi], list(self.x) + list(self.y) + list(self.z) + list(self.radius)


class CollisionConstraint(AbstractConstraint):
    """‖x[x1] − x[x2]‖² ≥ r²  (pairwise non-self-collision, src/constraints.jl:332-393); c = r² − dᵀd, p = 1."""
    kind = capi.CON_COLLISION
    state_only = True

    def __init__(self, n, x1, x2, radius):
        self.n = int(n)
        self.x1, self.x2 = [int(i) for i in x1], [int(i) for i in x2]
        if len(self.x1) != len(self.x2):
            raise AssertionError(f"Position dimensions must be of equal length, got {len(self.x1)} and {len(self.x2)}")
        self.radius = float(radius)
        self.p = 1
        self._sense = Inequality()

    def _fill(self):
        return self.x1 + self.x2, [self.radius]


class QuatVecEq(AbstractConstraint):
    """vec(normalize(x[qind])) = ±vec(qf)  (src/constraints.jl:938-965): the vector part of the attitude matches the goal
    quaternion on the nearer hemisphere; p = 3, Equality.  qf = (w, x, y, z), normalised like QuatRotation(qf)."""
    kind = capi.CON_QUATVEC
    state_only = True

    def __init__(self, n, m, qf, qind=(4, 5, 6, 7)):
        self.n, self.m = int(n), int(m)
        qf = _vec(qf, 4, "qf")
        self.qf = qf / np.linalg.norm(qf)
        self.qind = [int(i) for i in qind]
        if len(self.qind) != 4:
            raise ValueError("qind must hold 4 indices")
        self.p = 3
        self._sense = Equality()

    def _fill(self):
        return self.qind, list(self.qf)


class LinearConstraint(AbstractConstraint):
    """A z[inds] − b (=,≤) 0  (src/constraints.jl:103-150)."""
    kind = capi.CON_LINEAR

    def __init__(self, n, m, A, b, sense, inds=None):
        self.n, self.m = int(n), int(m)
        self.A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        self.b = _vec(b)
        if self.A.shape[0] != self.b.size:
            raise AssertionError("size(A,1) == length(b)")
        self.inds = list(range(1, n + m + 1)) if inds is None else [int(i) for i in inds]
        if len(self.inds) != self.A.shape[1]:
            raise AssertionError("length(inds) == size(A,2)")
        self._sense = sense
        self.p = self.b.size

    def check_dims(self, n, m):
        return self.n == n and self.m == m

    def _fill(self):
        return self.inds, list(self.A.ravel(order="F")) + list(self.b)


def StateBound(n, m, x_max=np.inf, x_min=-np.inf):
    """Bounds on the state only (the reference's `StateBound`, src/constraints.jl:528-631, keeps a separate type over the
    same arithmetic as BoundConstraint; here it IS a BoundConstraint with infinite control bounds: same rows, same order)."""
    return BoundConstraint(n, m, x_max=x_max, x_min=x_min)


def ControlBound(n, m, u_max=np.inf, u_min=-np.inf):
    """Bounds on the control only (src/constraints.jl:528-631)."""
    return BoundConstraint(n, m, u_max=u_max, u_min=u_min)


class IndexedConstraint(AbstractConstraint):
    """A constraint written for dimensions (n0, m0) acting on a slice of a larger problem's [x; u]
    (src/constraints.jl:820-936): c(z) = con(x[ix], u[iu]), Jacobian columns scattered to ix / n+iu, zero elsewhere.
    ``ix`` / ``iu`` are 1-based unit ranges (first, last) into the NEW state / control vectors, defaulting to the
    leading entries like the reference's 3-argument constructor.  The library needs no kernel for it: every descriptor
    addresses [x; u] through index lists, so the wrapper is lowered by remapping the inner constraint's indices."""

    def __init__(self, n, m, con, ix=None, iu=None):
        self.n, self.m, self.con = int(n), int(m), con
        n0 = getattr(con, "n", None)
        m0 = getattr(con, "m", None)
        if con.state_only or m0 is None:
            m0 = self.m if iu is None else _range_len(iu)
        if n0 is None:
            n0 = self.n if ix is None else _range_len(ix)
        ix = (1, n0) if ix is None else _knot_range(ix, self.n)
        iu = (1, m0) if iu is None else _knot_range(iu, self.m)
        if ix[1] - ix[0] + 1 != n0 or (not con.state_only and iu[1] - iu[0] + 1 != m0):
            raise DimensionMismatch("IndexedConstraint: length(ix), length(iu) must equal the inner constraint's dimensions")
        if not (1 <= ix[0] and ix[1] <= self.n and 1 <= iu[0] and iu[1] <= self.m):
            raise DimensionMismatch("IndexedConstraint: ix / iu outside the new state / control vector")
        self.n0, self.m0, self.ix, self.iu = n0, m0, ix, iu
        self.p, self._sense, self.kind = con.p, con.sense(), con.kind
        self.state_only = False  # a StageConstraint in the reference: Jacobians are p x (n+m)

    def _map(self, i):  # 1-based index into the inner [x0; u0] -> 1-based index into the new [x; u]
        return self.ix[0] + i - 1 if i <= self.n0 else self.n + self.iu[0] + (i - self.n0) - 1

    def _fill(self):
        inds, params = self.con._fill()
        if self.kind == capi.CON_BOUND:  # params = [z_max; z_min] of the inner dimensions -> padded with +-inf
            zmax, zmin = np.full(self.n + self.m, np.inf), np.full(self.n + self.m, -np.inf)
            nz0 = self.n0 + self.m0
            for i in range(nz0):
                zmax[self._map(i + 1) - 1] = params[i]
                zmin[self._map(i + 1) - 1] = params[nz0 + i]
            return [], list(zmax) + list(zmin)
        return [self._map(int(i)) for i in inds], params

    def _desc(self, k1, k2):
        d = super()._desc(k1, k2)
        return d

    @property
    def lowered_state_only(self):
        """Does the LOWERED descriptor act on the state alone (the library then reports Jacobians of width n)?  Follows
        nested wrappers down to the innermost constraint (change_dimension of an already wrapped list)."""
        return getattr(self.con, "lowered_state_only", self.con.state_only)


def _range_len(r):
    a, b = _knot_range(r, 0)
    return b - a + 1


def change_dimension(obj, n, m, ix=None, iu=None):
    """change_dimension (src/constraints.jl:934-936, src/constraint_list.jl:208-217): wrap a constraint, or every
    constraint of a list, for a problem of dimensions (n, m) whose state / control contain the old ones at ix / iu."""
    if isinstance(obj, ConstraintList):
        new = ConstraintList(n, m, obj.N)
        for con, inds in zip(obj.constraints, obj.inds):
            add_constraint(new, change_dimension(con, n, m, ix, iu), inds)
        return new
    if isinstance(obj, QuadraticCostFunction):
        return _change_cost_dimension(obj, n, m, ix, iu)
    if isinstance(obj, Objective):
        lifted = {}
        return Objective([lifted.setdefault(id(c), _change_cost_dimension(c, n, m, ix, iu)) for c in obj.cost])
    return IndexedConstraint(n, m, obj, ix, iu)


def _change_cost_dimension(cost, n, m, ix=None, iu=None):
    """change_dimension(cost::DiagonalCost, n, m, ix, iu) (src/cost_functions.jl:391-401): the cost of a problem of dimensions
    (n, m) that acts on x[ix], u[iu] as ``cost`` does and ignores the rest (zero weights there).  Also for dense QuadraticCosts
    (the blocks scattered the same way)."""
    n0, m0 = cost.n, cost.m
    ix = (1, n0) if ix is None else _knot_range(ix, n)
    iu = (1, m0) if iu is None else _knot_range(iu, m)
    if ix[1] - ix[0] + 1 != n0 or iu[1] - iu[0] + 1 != m0 or ix[1] > n or iu[1] > m or ix[0] < 1 or iu[0] < 1:
        raise DimensionMismatch("change_dimension: length(ix), length(iu) must equal the cost's dimensions and lie inside (n, m)")
    sx, su = slice(ix[0] - 1, ix[1]), slice(iu[0] - 1, iu[1])
    q, r = np.zeros(n), np.zeros(m)
    q[sx], r[su] = cost.q, cost.r
    if type(cost) in (DiagonalCost,):
        Q, R = np.zeros(n), np.zeros(m)
        Q[sx], R[su] = cost.Q, cost.R
        return DiagonalCost(Q, R, q, r, cost.c, terminal=cost.terminal)
    if type(cost) is QuadraticCost:
        Q, R, H = np.zeros((n, n)), np.zeros((m, m)), np.zeros((m, n))
        Q[sx, sx], R[su, su], H[su, sx] = cost.Q, cost.R, cost.H
        return QuadraticCost(Q, R, H, q, r, cost.c, terminal=cost.terminal)
    raise UnsupportedError("change_dimension: DiagonalCost and QuadraticCost only (src/cost_functions.jl:391-401)")


def InfeasibleConstraint(n, m):
    """Altro's ``InfeasibleConstraint``: the slack controls of an ``InfeasibleModel`` of dimensions (n, m) — its last n controls —
    are zero, an equality on every stage knot."""
    m0 = m - n
    return LinearConstraint(n, m, np.eye(n), np.zeros(n), Equality(), inds=list(range(n + m0 + 1, n + m + 1)))


def infeasible_controls(prob):
    """Altro's ``infeasible_controls``: from the problem's CURRENT states (an ``initial_states!`` guess) and base controls, the slack
    controls that make the guess dynamically feasible (to_infeasible_controls)."""
    prob._call("infeasible_controls")


def InfeasibleProblem(prob, X0, R_inf=1.0, U0=None):
    """Altro's ``InfeasibleProblem(prob, Z0, R_inf)`` — what ``ALTROSolver(prob, infeasible=true)`` solves: the model wrapped in an
    ``InfeasibleModel``, costs and constraints lifted to (n, m + n) with ``change_dimension``, ½ R_inf |w|² on the slack controls,
    ``InfeasibleConstraint`` (w = 0) on knots 1..N-1, and the slack controls seeded from the state guess ``X0`` ([N, n] or
    [B, N, n]; ``U0`` = base controls, default the problem's current ones).  Altro passes ``opts.R_inf / dt``."""
    if not isinstance(prob.model, (DoubleIntegrator, Cartpole)):
        raise UnsupportedError("InfeasibleProblem: the base model must be a double integrator or a Cartpole")
    model = InfeasibleModel(prob.model)
    n, m0, N = prob.n, prob.m, prob.N
    m = m0 + n
    lifted = {}
    costs = []
    for c in prob.obj.cost:
        if id(c) not in lifted:
            cc = _change_cost_dimension(c, n, m)
            if isinstance(cc, DiagonalCost):
                cc.R[m0:] += R_inf
            else:
                cc.R[m0:, m0:] += R_inf * np.eye(n)
            lifted[id(c)] = cc
        costs.append(lifted[id(c)])
    cons = change_dimension(prob.constraints, n, m, (1, n), (1, m0))
    add_constraint(cons, InfeasibleConstraint(n, m), (1, N - 1))
    opts = SolverOptions(lib=prob._lib)
    prob._call("get_options", C.byref(opts._o))
    x0 = np.empty((prob.B, n))
    prob._call("get_initial_state", prob._pd(x0))
    new = Problem(model, Objective(costs), x0[0], prob.tf, xf=prob.xf, constraints=cons, t0=prob.t0, dt=prob._dt,
                  integration=prob.integration, batch=prob.B, options=opts, lib=prob._lib)
    new.set_initial_state(x0)
    Ub = controls(prob) if U0 is None else prob._batch(np.asarray(U0, dtype=np.float64), (m0, N - 1), "U0").reshape(prob.B, N - 1, m0)
    U = np.zeros((prob.B, N - 1, m))
    U[:, :, :m0] = Ub
    initial_controls(new, U)
    initial_states(new, X0)
    infeasible_controls(new)
    return new


def _knot_range(inds, N):
    if isinstance(inds, (int, np.integer)):
        return int(inds), int(inds)
    if isinstance(inds, range):
        if inds.step != 1 or len(inds) == 0:
            raise ArgumentError("knot range must be a non-empty unit range")
        return inds[0], inds[-1]
    a, b = inds
    return int(a), int(b)


class ConstraintList:
    """src/constraint_list.jl:35-52."""

    def __init__(self, n, m=None, N=None):
        """ConstraintList(n, m, N), ConstraintList(nx, nu) with one entry per knot, or ConstraintList(models)
        (src/constraint_list.jl:35-68)."""
        if m is None:  # ConstraintList(models)
            nx, nu = dims(n)
        elif N is None:  # ConstraintList(nx, nu)
            nx, nu = [int(v) for v in n], [int(v) for v in m]
            if len(nx) != len(nu):
                raise DimensionMismatch("nx and nu must have one entry per knot point")
        else:
            nx, nu = [int(n)] * int(N), [int(m)] * int(N)
        self.nx, self.nu, self.N = nx, nu, len(nx)
        self.n, self.m = max(nx), max(nu)  # the uniform dimensions when every knot has the same
        self.constraints, self.inds = [], []
        self.p = [0] * self.N

    @property
    def uniform(self):
        return len(set(self.nx)) == 1 and len(set(self.nu)) == 1

    def __len__(self):
        return len(self.constraints)

    def __getitem__(self, i):
        return self.constraints[i]

    def __iter__(self):
        return iter(self.constraints)

    def zip(self):
        return list(zip(self.inds, self.constraints))

    def copy(self):
        c = ConstraintList(self.nx, self.nu)
        for con, (a, b) in zip(self.constraints, self.inds):
            add_constraint(c, con, (a, b))
        return c

    def _descs(self):
        arr = (ConstraintDesc * max(1, len(self)))()
        for i, (con, (a, b)) in enumerate(zip(self.constraints, self.inds)):
            arr[i] = con._desc(a, b)
        return arr


def add_constraint(cons, con, inds, idx=-1):
    """add_constraint!  (src/constraint_list.jl:103-134).  ``idx`` is 1-based like the reference."""
    k1, k2 = _knot_range(inds, cons.N)
    for k in range(k1, min(k2, cons.N) + 1):  # every knot of the range, like the reference (dimensions may change along the horizon)
        if k >= 1 and not con.check_dims(cons.nx[k - 1], cons.nu[k - 1]):
            raise DimensionMismatch(f"New constraint not consistent with n={cons.nx[k - 1]} and m={cons.nu[k - 1]} at time step {k}.")
    assert 1 <= k1 <= k2 <= cons.N, f"Invalid inds, inds[end] must be less than number of knotpoints, {cons.N}"
    if len(cons) == 0:
        idx = -1
    if idx == -1:
        cons.constraints.append(con)
        cons.inds.append((k1, k2))
    elif 0 < idx <= len(cons):
        cons.constraints.insert(idx - 1, con)
        cons.inds.insert(idx - 1, (k1, k2))
    else:
        raise ArgumentError(f"cannot insert constraint at index={idx}. Length = {len(cons)}")
    cons.p = [0] * cons.N
    for c, (a, b) in zip(cons.constraints, cons.inds):
        for k in range(a, b + 1):
            cons.p[k - 1] += c.p
    return cons


def num_constraints(obj):
    """src/constraint_list.jl:198 / src/problem.jl:203: total constraint rows per knot."""
    return obj.p if isinstance(obj, ConstraintList) else obj.constraints.p


# --------------------------------------------------------------------------------------------- problem
class KnotPoint:
    """z=[x;u], t, dt; terminal ⇔ dt == 0 (RobotDynamics.KnotPoint as built at src/problem.jl:58-61)."""

    def __init__(self, x, u, t, dt):
        self.x, self.u = _vec(x), _vec(u)
        self.z = np.concatenate([self.x, self.u])
        self.t, self.dt = float(t), float(dt)

    def is_terminal(self):
        return self.dt == 0.0


class SolverOptions:
    """Altro.jl ``SolverOptions`` subset (examples/Cartpole.ipynb cell 17); fields of ``to_solver_opts``."""

    _names = [f[0] for f in SolverOpts._fields_ if not f[0].startswith("reserved")]

    def __init__(self, lib=None, **kw):
        lib = lib or capi.load_hip_library()
        self._o = lib.default_options()
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, k):
        if k in SolverOptions._names:
            return getattr(self._o, k)
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k in SolverOptions._names:
            setattr(self._o, k, v)
        elif k.startswith("_"):
            object.__setattr__(self, k, v)
        else:
            raise ArgumentError(f"unknown solver option {k}")


class Problem:
    """Trajectory-optimisation problem for a BATCH of trajectories (src/problem.jl:36-73).

    ``Problem(model, obj, x0, tf; xf, constraints, t0, X0, U0, dt, integration)`` as in the reference,
    plus ``batch`` (number of independent trajectories sharing model/objective/constraints) and
    ``device``.  ``x0`` may be [n] (replicated) or [n, B] / [B, n]-shaped via ``set_initial_state``.
    """

    def __init__(self, model, obj, *args, xf=None, constraints=None, t0=0.0, X0=None, U0=None, dt=None,
                 integration=RK4, batch=1, device=0, options=None, lib=None, **kwargs):
        if "x0" in kwargs:  # src/problem.jl:87-91
            raise ArgumentError("Cannot pass x0 as a keyword argument. It is now a positional argument, "
                                "and xf is a keyword argument.\n\tUse Problem(model, obj, x0, tf, xf=xf, kwargs...) instead.")
        if len(args) != 2 or kwargs:
            raise TypeError("Problem(model, obj, x0, tf; xf, constraints, t0, X0, U0, dt, integration, batch, device)")
        x0, tf = args
        self._lib = lib or capi.load_hip_library()
        # Problem(model, ...) builds N-1 copies of the model (src/problem.jl:115); Problem(models::Vector, ...) takes one model per
        # time step, whose dimensions may change along the horizon (src/problem.jl:36-73, src/dynamics.jl:15-31)
        models = list(model) if isinstance(model, (list, tuple)) else [model] * (len(obj) - 1)
        if len(models) != len(obj) - 1:
            raise AssertionError("length(models) == N-1")  # src/problem.jl:49
        nx, nu = dims(models)
        if np.asarray(x0).ndim == 1 and np.asarray(x0).size != nx[0]:
            raise AssertionError("length(x0) == nx[1]")  # src/problem.jl:46
        self.constraints = constraints if constraints is not None else ConstraintList(nx, nu)
        if self.constraints.nx != nx:
            raise DimensionMismatch("Constraint state dimensions don't match model")
        if self.constraints.nu != nu:
            raise DimensionMismatch("Constraint control dimensions don't match model")
        nx_obj, nu_obj = obj.knot_dims()
        if nx_obj != nx:
            raise DimensionMismatch("Objective state dimensions don't match model.")
        if nu_obj != nu:
            raise DimensionMismatch("Objective control dimensions don't match model.")
        self.models, self.nx, self.nu = models, nx, nu
        self.hybrid = False
        if not all(_same_model(mod, models[0]) for mod in models) or isinstance(models[0], DiscreteMap):
            # every check of the reference's constructor has passed.  The library integrates ONE compiled-in model over the whole
            # horizon; a model vector runs when it is the per-step view of a compiled-in HYBRID model (models.h, model_step)
            # (the general per-step table, TO_MODEL_VECTOR, otherwise)
            model = HybridDoubleIntegrator.match(models) or ModelVector(models)
            self.hybrid = True
            n, m = model.dims()
            # costs and constraints of the narrower knots onto the zero-padded storage vectors
            padded = {}
            obj_lowered = Objective([padded.setdefault(id(c), pad_cost(c, n, m)) for c in obj.cost])
            cons_lowered = ConstraintList(n, m, len(obj))
            for inds, con in self.constraints.zip():
                k1 = _knot_range(inds, len(obj))[0]
                n0, m0 = nx[k1 - 1], nu[k1 - 1]
                if (n0, m0) != (n, m) and not (isinstance(con, GoalConstraint) and n0 == n):
                    con = IndexedConstraint(n, m, con, ix=(1, n0), iu=(1, m0))
                add_constraint(cons_lowered, con, inds)
            x0 = np.asarray(x0, dtype=np.float64)
            x0 = np.concatenate([x0, np.zeros(x0.shape[:-1] + (n - nx[0],))], axis=-1) if nx[0] < n else x0
        else:
            model, obj_lowered, cons_lowered = models[0], obj, self.constraints
        self.model, self.obj = model, obj
        n, m = model.dims()
        self.n, self.m, self.N, self.B = n, m, len(obj), int(batch)
        self.t0, self.tf = float(t0), float(tf)
        if self.hybrid and xf is not None and np.asarray(xf).size == nx[-1] and nx[-1] < n:
            xf = np.r_[np.asarray(xf, dtype=np.float64).ravel(), np.zeros(n - nx[-1])]   # the reference's xf has length nx[end] (src/problem.jl:47)
        self.xf = np.full(n, np.nan) if xf is None else _vec(xf, n, "xf")
        self.integration = integration
        self._dt = None if dt is None else np.ascontiguousarray(_vec(dt, self.N - 1, "dt"))

        uniq, index = obj_lowered._descs()
        self._obj_lowered, self._cons_lowered = obj_lowered, cons_lowered
        self._cost_objs = uniq
        self._costs = (CostDesc * len(uniq))(*[c._desc() for c in uniq])
        self._cost_index = (C.c_int32 * self.N)(*index)
        self._cons = cons_lowered._descs()
        d = ProblemDesc()
        d.abi_version, d.model, d.integrator = capi.TO_ABI_VERSION, model.model_id, integration
        d.n, d.m, d.N, d.B = n, m, self.N, self.B
        p = model.params()
        d.model_params[: len(p)] = p
        d.t0, d.tf = self.t0, self.tf
        d.dt = self._dt.ctypes.data_as(C.POINTER(C.c_double)) if self._dt is not None else None
        d.n_costs, d.costs, d.cost_index = len(uniq), self._costs, self._cost_index
        d.n_constraints = len(cons_lowered)
        d.constraints = self._cons
        self._steps = model._step_descs() if isinstance(model, ModelVector) else None
        d.step_models = self._steps
        self._desc = d
        self._h = C.c_void_p()
        opts = options._o if isinstance(options, SolverOptions) else options
        self._lib.call("create", C.byref(d), C.byref(opts) if opts is not None else None, int(device), C.byref(self._h))
        knx, knu = self.knot_dims()
        if (knx, knu) != (nx, nu):  # the library's own view of the model vector (to_knot_dims) against RD.dims(models)
            raise DimensionMismatch(f"library knot dimensions {knx}, {knu} differ from dims(models) = {nx}, {nu}")
        self.set_initial_state(x0)
        if U0 is not None:
            initial_controls(self, U0)
        if X0 is not None:
            initial_states(self, X0)

    def knot_dims(self):
        """Live (state, control) dimension at each knot as the library sees them (to_knot_dims; RD.dims(models))."""
        nx, nu = (C.c_int32 * self.N)(), (C.c_int32 * self.N)()
        self._call("knot_dims", nx, nu)
        return list(nx), list(nu)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.raw("destroy")(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # ---- helpers
    def _call(self, name, *args):
        return self._lib.call(name, self._h, *args)

    @staticmethod
    def _pd(a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    @staticmethod
    def _pi(a):
        return a.ctypes.data_as(C.POINTER(C.c_int32))

    def _batch(self, a, inner, name):
        """Coerce to C-contiguous [B, *inner_reversed] memory == column-major (inner..., B)."""
        a = np.asarray(a, dtype=np.float64)
        tgt = (self.B,) + tuple(reversed(inner))
        if a.shape == tuple(inner):  # single trajectory in Julia (n, N) layout -> replicate
            a = np.broadcast_to(a.T, tgt)
        elif a.shape == tuple(reversed(inner)):
            a = np.broadcast_to(a, tgt)
        elif a.shape != tgt:
            raise DimensionMismatch(f"{name}: expected shape {tgt} ([B, knot, dim]) or {tuple(inner)}, got {a.shape}")
        return np.ascontiguousarray(a)

    # ---- setters / getters (src/problem.jl:198-310)
    def set_initial_state(self, x0):
        a = np.asarray(x0, dtype=np.float64)
        if a.ndim == 1:
            a = np.broadcast_to(_vec(a, self.n, "x0"), (self.B, self.n))
        elif a.shape != (self.B, self.n):
            raise DimensionMismatch(f"x0 must be [n] or [B, n]; got {a.shape}")
        a = np.ascontiguousarray(a)
        self._call("set_initial_state", self._pd(a))
        self.x0 = a.copy()

    def dims(self):
        return self.n, self.m, self.N

    @property
    def errstate_dim(self):
        return self.model.errstate_dim

    def gettimes(self):
        dt = self._dt if self._dt is not None else np.full(self.N - 1, (self.tf - self.t0) / (self.N - 1))
        return self.t0 + np.concatenate([[0.0], np.cumsum(dt)])


def initial_controls(prob, U0):
    """initial_controls!(prob, U0): U0 is [m] (all knots), [m, N-1] / [N-1, m], or [B, N-1, m]."""
    a = np.asarray(U0, dtype=np.float64)
    if a.ndim == 1 and a.size == prob.m:
        a = np.ascontiguousarray(a)
        prob._call("set_controls_uniform", prob._pd(a))
        return
    if a.ndim == 2 and prob.m == prob.N - 1 and a.shape == (prob.m, prob.N - 1):
        raise ArgumentError("ambiguous U0 shape; pass [B, N-1, m]")
    a = prob._batch(a, (prob.m, prob.N - 1), "U0")
    prob._call("set_controls", prob._pd(a))


def initial_states(prob, X0):
    """initial_states!(prob, X0) (src/problem.jl:242-253): X0 is [n, N] / [N, n] (one trajectory, replicated) or [B, N, n]."""
    a = np.asarray(X0, dtype=np.float64)
    if a.ndim == 2 and prob.n == prob.N and a.shape == (prob.n, prob.N):
        raise ArgumentError("ambiguous X0 shape (n == N); pass [B, N, n]")
    a = prob._batch(a, (prob.n, prob.N), "X0")
    prob._call("set_states", prob._pd(a))


def set_initial_state(prob, x0):
    prob.set_initial_state(x0)


def states(prob):
    """states(prob) -> [B, N, n]."""
    X = np.empty((prob.B, prob.N, prob.n))
    prob._call("get_states", prob._pd(X))
    return X


def controls(prob):
    """controls(prob) -> [B, N-1, m]."""
    U = np.empty((prob.B, prob.N - 1, prob.m))
    prob._call("get_controls", prob._pd(U))
    return U


def rollout(prob):
    """rollout!(prob)  (src/problem.jl:330-340)."""
    prob._call("rollout")


def cost(prob):
    """cost(prob) -> [B]  (src/problem.jl:321, src/objective.jl:89-93).  Also fills Objective.J (first trajectory)."""
    J = np.empty(prob.B)
    prob._call("cost", prob._pd(J))
    return J


def stage_costs(prob):
    """cost!(obj, Z): per-knot costs [B, N] (Objective.J, src/objective.jl:104-106)."""
    Jk = np.empty((prob.B, prob.N))
    prob._call("stage_costs", prob._pd(Jk))
    prob.obj.J = Jk[0].copy()
    return Jk


def get_constraints(prob):
    return prob.constraints


def get_objective(prob):
    return prob.obj


def get_model(prob, k=None):
    return prob.model


def get_initial_state(prob):
    x = np.empty((prob.B, prob.n))
    prob._call("get_initial_state", prob._pd(x))
    return x


def get_final_state(prob):
    return prob.xf


def get_trajectory(prob):
    return states(prob), controls(prob)


def gettimes(prob):
    return prob.gettimes()


def set_goal_state(prob, xf, objective=True, constraint=True):
    """set_goal_state!  (src/problem.jl:294-310): set_LQR_goal! on every cost, update GoalConstraints."""
    if getattr(prob, "hybrid", False):  # (the reference's set_goal_state! needs one state dimension for all knots as well)
        raise UnsupportedError("set_goal_state on a hybrid model vector: the knots have different state dimensions; rebuild the problem")
    if np.asarray(xf).ndim == 2:
        return _set_goal_state_batch(prob, xf, objective, constraint)
    xf = _vec(xf, prob.n, "xf")
    if objective:
        for i, c in enumerate(prob._cost_objs):
            if c.kind == capi.COST_ERROR_QUADRATIC:
                raise TypeError("set_LQR_goal! is only defined for QuadraticCostFunction (src/cost_functions.jl:249)")
            if c.kind == capi.COST_QUADRATIC:
                c.q = -c.Q @ xf  # set_LQR_goal!: only q changes (src/cost_functions.jl:249-252)
            else:
                c.q = -c.Q * xf
            d = c._desc()
            prob._call("set_cost", i, C.byref(d))
    if constraint:
        for i, (con, (a, b)) in enumerate(zip(prob.constraints.constraints, prob.constraints.inds)):
            if isinstance(con, GoalConstraint):
                con.xf = xf[[j - 1 for j in con.inds]].copy()
                d = con._desc(a, b)
                prob._call("set_constraint", i, C.byref(d))
    prob.xf = xf.copy()
    if objective:
        prob.xf_batch = None  # to_set_cost starts every cost's per-trajectory terms over: the batch goals are gone


def _set_goal_state_batch(prob, Xf, objective=True, constraint=True):
    """set_goal_state!(prob, Xf) with ONE GOAL PER TRAJECTORY, Xf [B, n] (batched MPC / goal sweeps on one handle):
    set_LQR_goal!(cost, xf_b) on every cost for every trajectory — q_b = -Q xf_b, nothing else changes (src/cost_functions.jl:249-252)
    — through to_set_cost_linear_batch, and with ``constraint=True`` the target of every GoalConstraint, xf_b[inds], through
    to_set_constraint_params_batch."""
    Xf = np.ascontiguousarray(np.asarray(Xf, dtype=np.float64))
    if Xf.shape != (prob.B, prob.n):
        raise DimensionMismatch(f"Xf must be [B, n] = {(prob.B, prob.n)}; got {Xf.shape}")
    if constraint:  # (src/problem.jl:303-309: every GoalConstraint of the list gets the new target — here trajectory by trajectory)
        for i, con in enumerate(prob.constraints.constraints):
            if isinstance(con, GoalConstraint):
                par = np.ascontiguousarray(Xf[:, [j - 1 for j in con.inds]])      # [B, p] = column-major (p, B)
                prob._call("set_constraint_params_batch", i, prob._pd(par))
    if objective:
        for i, c in enumerate(prob._cost_objs):
            if c.kind == capi.COST_ERROR_QUADRATIC:
                raise TypeError("set_LQR_goal! is only defined for QuadraticCostFunction (src/cost_functions.jl:249)")
            q = -(Xf @ c.Q.T) if c.kind == capi.COST_QUADRATIC else -(Xf * c.Q[None, :])   # [B, n] = column-major (n, B)
            prob._call("set_cost_linear_batch", i, prob._pd(np.ascontiguousarray(q)), None)
    prob.xf_batch = Xf.copy()


def set_constraint_params_batch(prob, con_id, params):
    """One parameter set per TRAJECTORY for constraint ``con_id`` (0-based position in the ConstraintList): ``params`` [B, p] — a
    GoalConstraint's target xf_b[inds], or a LinearConstraint's right-hand side b_b (to_set_constraint_params_batch).  ``set_goal_state``
    with a matrix calls this for every GoalConstraint; ``clear_goal_state_batch`` returns every constraint to its shared parameters."""
    con = prob.constraints.constraints[con_id]
    par = np.ascontiguousarray(np.asarray(params, dtype=np.float64))
    if par.shape != (prob.B, con.p):
        raise DimensionMismatch(f"params must be [B, p] = {(prob.B, con.p)}; got {par.shape}")
    prob._call("set_constraint_params_batch", int(con_id), prob._pd(par))


def clear_goal_state_batch(prob):
    """Back to the shared descriptors (to_clear_cost_linear_batch, to_clear_constraint_params_batch)."""
    prob._call("clear_cost_linear_batch")
    prob._call("clear_constraint_params_batch")
    prob.xf_batch = None


def update_trajectory(prob, X, U, start=1):
    """update_trajectory!(obj, Z, start)  (src/objective.jl:198-212): retarget a tracking objective (one cost per knot,
    see TrackingObjective) to knots start..start+N-1 of the reference (X [n, Nref], U [m, >=Nref-1]); like
    set_LQR_goal! it changes only q and r, never the constant c (src/cost_functions.jl:249-258)."""
    if getattr(prob, "hybrid", False):
        raise UnsupportedError("update_trajectory on a hybrid model vector: the knots have different dimensions; rebuild the problem")
    X = np.asarray(X, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    costs = prob.obj.cost
    if len(set(map(id, costs))) != prob.N:
        raise ValueError("update_trajectory! needs a tracking objective (a distinct cost per knot)")
    if start < 1 or start - 1 + prob.N > X.shape[1]:
        raise IndexError("reference trajectory too short for this start index")
    for i, c in enumerate(costs):
        k = start - 1 + i
        xf = _vec(X[:, k], prob.n, "xf")
        uf = _vec(U[:, k], prob.m, "uf") if k < U.shape[1] else np.zeros(prob.m)
        if c.kind == capi.COST_ERROR_QUADRATIC:
            raise TypeError("update_trajectory! needs QuadraticCostFunction costs (src/objective.jl:207)")
        if c.kind == capi.COST_QUADRATIC:
            c.q, c.r = -c.Q @ xf, -c.R @ uf
        else:
            c.q, c.r = -c.Q * xf, -c.R * uf
        d = c._desc()
        prob._call("set_cost", prob._cost_objs.index(c), C.byref(d))


def evaluate_constraints(prob, i):
    """evaluate_constraints! for constraint ``i`` (0-based list position) -> [B, nk, p]  (src/abstract_constraint.jl:200-225)."""
    con = prob.constraints[i]
    a, b = prob.constraints.inds[i]
    vals = np.empty((prob.B, b - a + 1, con.p))
    prob._call("evaluate_constraints", i, prob._pd(vals))
    return vals


def _lib_width(prob, i):
    """Jacobian width the LIBRARY reports for constraint ``i`` (to_constraint_info): n for state constraints, else n+m.
    Asked, not inferred from the Python classes, so that buffer sizes can never disagree with what the kernels write."""
    p, w, nk, sn = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    prob._call("constraint_info", i, C.byref(p), C.byref(w), C.byref(nk), C.byref(sn))
    return int(w.value)


def constraint_jacobians(prob, i):
    """constraint_jacobians! -> [B, nk, p, w]  (src/abstract_constraint.jl:236-248)."""
    con = prob.constraints[i]
    a, b = prob.constraints.inds[i]
    w = _lib_width(prob, i)
    jac = np.empty((prob.B, b - a + 1, w, con.p))
    prob._call("constraint_jacobians", i, prob._pd(jac))
    jac = jac.transpose(0, 1, 3, 2)
    if isinstance(con, IndexedConstraint) and w < prob.n + prob.m:  # the reference's wrapper is a stage constraint: p x (n+m)
        jac = np.concatenate([jac, np.zeros(jac.shape[:3] + (prob.m,))], axis=3)
    return jac


def constraint_hessians(prob, i, lam, H=None):
    """∇jacobian! over the knot range (src/abstract_constraint.jl:255-280): returns H + Σ_r lam[..., r] ∇²c_r as
    [B, nk, w, w]; ``lam`` is [B, nk, p]; ``H`` (same shape as the result) defaults to zeros — the operator ADDS."""
    con = prob.constraints[i]
    a, b = prob.constraints.inds[i]
    nk, w = b - a + 1, _lib_width(prob, i)
    nz = prob.n + prob.m
    # the reference's IndexedConstraint is a stage constraint whatever it wraps: its Jacobians are p x (n+m) and its
    # Hessians (n+m) x (n+m) (src/constraints.jl:820-936); the library works on the lowered (possibly state-only) form
    pad = isinstance(con, IndexedConstraint) and w < nz
    wo = nz if pad else w
    lam = np.ascontiguousarray(np.broadcast_to(np.asarray(lam, dtype=np.float64), (prob.B, nk, con.p)))
    if H is not None:
        H = np.asarray(H, dtype=np.float64)
        if H.shape != (prob.B, nk, wo, wo):
            raise DimensionMismatch(f"H must be [B, nk, {wo}, {wo}]")
    out = np.zeros((prob.B, nk, w, w)) if H is None else np.ascontiguousarray(H[:, :, :w, :w].transpose(0, 1, 3, 2)).copy()
    prob._call("constraint_hessians", i, prob._pd(lam), prob._pd(out))
    out = out.transpose(0, 1, 3, 2)
    if pad:
        full = np.zeros((prob.B, nk, nz, nz)) if H is None else H.copy()
        full[:, :, :w, :w] = out
        return full
    return out


# --------------------------------------------------------------------------------------------- solvers
class _Solver:
    _entry = "ilqr_solve"

    def __init__(self, prob, opts=None, **kw):
        self.prob = prob
        if opts is None:  # start from the options the Problem was created with
            opts = SolverOptions(lib=prob._lib)
            prob._call("get_options", C.byref(opts._o))
        for k, v in kw.items():
            setattr(opts, k, v)
        self.opts = opts
        B = prob.B
        self.stats = dict(
            iterations=np.zeros(B, np.int32), iterations_outer=np.zeros(B, np.int32), status=np.zeros(B, np.int32),
            cost=np.zeros(B), dJ=np.zeros(B), gradient=np.zeros(B), c_max=np.zeros(B), penalty_max=np.zeros(B),
            iterations_pn=np.zeros(B, np.int32))
        self.total_iterations = 0
        self.batch_steps = 0
        self.solve_ms = 0.0

    def solve(self):
        p = self.prob
        p._call("set_options", C.byref(self.opts._o))
        st = SolveStats()
        for k, a in self.stats.items():
            ptr = a.ctypes.data_as(C.POINTER(C.c_int32 if a.dtype == np.int32 else C.c_double))
            setattr(st, k, ptr)
        p._call(self._entry, C.byref(st))
        self.total_iterations, self.batch_steps, self.solve_ms = st.total_iterations, st.batch_steps, st.solve_ms
        return self

    def solve_async(self):
        """Enqueue the solve and return at once (to_*_solve_async); ``wait()`` blocks until it is done.  One solve in flight per
        Problem; the stats arrays are valid after ``wait()``."""
        if self._entry == "pn_solve":
            raise UnsupportedError("no asynchronous entry point for the projected-Newton solver alone")
        p = self.prob
        p._call("set_options", C.byref(self.opts._o))
        self._st = st = SolveStats()
        for k, a in self.stats.items():
            ptr = a.ctypes.data_as(C.POINTER(C.c_int32 if a.dtype == np.int32 else C.c_double))
            setattr(st, k, ptr)
        p._call(self._entry + "_async", C.byref(st))
        return self

    def wait(self):
        self.prob._call("solve_wait")
        st = self._st
        self.total_iterations, self.batch_steps, self.solve_ms = st.total_iterations, st.batch_steps, st.solve_ms
        return self

    def progress(self):
        """(active, batch_steps, in_flight) of the solve in flight as its solve loop last saw it (to_solve_progress)."""
        a, b, f = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.prob._call("solve_progress", C.byref(a), C.byref(b), C.byref(f))
        return a.value, b.value, bool(f.value)

    def wait_below(self, active_max):
        """Block until at most ``active_max`` trajectories of the solve in flight are still iterating (to_solve_wait_below)."""
        self.prob._call("solve_wait_below", int(active_max))
        return self


class iLQRSolver(_Solver):
    """Altro.iLQRSolver(prob, opts): unconstrained iLQR on the batch (examples/Cartpole.ipynb cell 25)."""
    _entry = "ilqr_solve"


class ALSolver(_Solver):
    """Altro's augmented-Lagrangian iLQR (Altro.ALSolver: the AL stage of ALTROSolver, run to ``constraint_tolerance``)."""
    _entry = "al_solve"


class ProjectedNewtonSolver(_Solver):
    """Altro.ProjectedNewtonSolver: Newton steps on the active constraints (dynamics defects included) from the problem's
    CURRENT trajectory; ``stats["iterations_pn"]`` counts the linearisations, ``stats["c_max"]`` includes the defects."""
    _entry = "pn_solve"


class ALTROSolver(_Solver):
    """Altro.ALTROSolver (examples/Cartpole.ipynb cell 17, examples/Quadrotor.ipynb cell 20): AL-iLQR down to
    ``projected_newton_tolerance``, then the projected-Newton polish down to ``constraint_tolerance``
    (``projected_newton=0``: the AL stage alone, like Altro's option of the same name)."""
    _entry = "altro_solve"

    def __init__(self, prob, opts=None, infeasible=False, R_inf=1.0, **kw):
        """``infeasible=True`` (Altro's keyword): the solver works on ``InfeasibleProblem(prob, states(prob), R_inf / dt)`` — the
        problem's current states (``initial_states!``) are the start, made dynamically feasible by slack controls that the AL stage
        and the polish drive to zero; ``solver.prob`` is that augmented problem (controls [u; w])."""
        if infeasible:
            dt = (prob.tf - prob.t0) / (prob.N - 1) if prob._dt is None else float(prob._dt[0])
            prob = InfeasibleProblem(prob, states(prob), R_inf / dt)
        super().__init__(prob, opts, **kw)


def solve(solver):
    """solve!(solver)."""
    return solver.solve()


class SolvePipeline:
    """Pipelined solves over ``depth = len(solvers)`` handles of the same shape (to_solve_progress / to_solve_wait_below).

    A solve is batch-synchronous and its batch drains unevenly (C3: 52 of 141 batch steps run on a handful of stragglers with the chip
    empty).  Trajectories are independent — one ``Z`` per problem, no cross terms (src/problem.jl:330-340) — so a host with more work
    than one batch (the next MPC batch, the next shard of a sweep) starts the next solve on another handle while the one in flight
    drains.  ``submit(prepare)``: job i runs on handle ``i % depth``; that handle's previous job is collected first (``wait``), then
    ``prepare(prob)`` sets the job's inputs (x0 / goal / initial controls), and the solve is admitted once the job in front has at most
    ``admit_below`` trajectories still iterating.  Every trajectory's result equals that of an unpipelined solve bit for bit
    (tests/test_gpu_pipeline.py).  ``results`` holds (job, total_iterations, batch_steps, stats copy or None) in completion order."""

    def __init__(self, solvers, admit_below=None, keep_stats=False, on_done=None):
        self.solvers = list(solvers)
        self.depth = len(self.solvers)
        B = self.solvers[0].prob.B
        # (measured, C3 / C5 over 2-4 handles, admission at 100 % ... 6 % of the batch: the earliest admission always won — the dense steps of a
        # batch of 4096 do not fill the chip either, so two dense solves side by side already gain; profiles/r06_ab/)
        self.admit_below = B if admit_below is None else int(admit_below)
        self.keep_stats = keep_stats
        self.on_done = on_done
        self._job = [None] * self.depth
        self._last = None
        self.jobs = 0
        self.results = []

    def _collect(self, slot):
        if self._job[slot] is None:
            return
        s = self.solvers[slot]
        s.wait()
        if self.on_done is not None:
            self.on_done(self._job[slot], s)
        self.results.append((self._job[slot], int(s.total_iterations), int(s.batch_steps),
                             {k: v.copy() for k, v in s.stats.items()} if self.keep_stats else None))
        if self._last is s:
            self._last = None
        self._job[slot] = None

    def submit(self, prepare=None):
        slot = self.jobs % self.depth
        self._collect(slot)
        s = self.solvers[slot]
        if prepare is not None:
            prepare(s.prob)
        if self._last is not None:
            self._last.wait_below(self.admit_below)
        s.solve_async()
        self._job[slot] = self.jobs
        self._last = s
        self.jobs += 1
        return slot

    def drain(self):
        """Collect everything in flight (oldest first)."""
        order = sorted((j, i) for i, j in enumerate(self._job) if j is not None)
        for _, slot in order:
            self._collect(slot)
        return self

    @property
    def total_iterations(self):
        return sum(r[1] for r in self.results)


def iterations(solver):
    return solver.stats["iterations"]


def status(solver):
    return solver.stats["status"]


def dynamics_defect(obj):
    """max |x_1 (-) x0|, |x_{k+1} (-) f(x_k, u_k)| per trajectory -> [B]: exactly 0 for a rollout, the dynamics infeasibility
    a projected-Newton polish leaves (it treats the dynamics as constraints) otherwise."""
    prob = obj.prob if isinstance(obj, _Solver) else obj
    d = np.empty(prob.B)
    prob._call("dynamics_defect", prob._pd(d))
    return d


def max_violation(obj):
    """max_violation(solver) or max_violation(prob) -> [B]."""
    prob = obj.prob if isinstance(obj, _Solver) else obj
    c = np.empty(prob.B)
    prob._call("max_violation", prob._pd(c))
    return c
