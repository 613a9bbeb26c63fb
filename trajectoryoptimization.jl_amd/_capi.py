"""ctypes binding of the C-ABI declared in ``include/trajopt_hip.h``.

The structures below mirror the header field by field.  ``Library`` binds every entry point of a
shared object that exports that ABI under a symbol prefix (``to_`` for the product library
``csrc/libtrajopt_hip.so``).  The product never loads anything else: ``load_hip_library`` raises
``HipLibraryMissing`` when the HIP extension has not been built — there is no CPU fallback.
(The test-suite binds the same class to the CPU oracle with prefix ``oracle_``; that happens in
``tests/`` only.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

TO_ABI_VERSION = 6
TO_MAX_N, TO_MAX_M, TO_MAX_P = 16, 8, 40
TO_MAX_CON_PARAMS, TO_MAX_CON_INDS = 400, 48

# return codes (to_status_code)
TO_OK = 0
TO_ERR_DIMENSION_MISMATCH, TO_ERR_ARGUMENT, TO_ERR_ASSERTION = -1, -2, -3
TO_ERR_HIP, TO_ERR_UNSUPPORTED, TO_ERR_NULL, TO_ERR_CONE = -4, -5, -6, -7

# solver status (to_solver_status, Altro.jl TerminationStatus order)
(UNSOLVED, LINESEARCH_FAIL, SOLVE_SUCCEEDED, MAX_ITERATIONS, MAX_ITERATIONS_OUTER, MAXIMUM_COST,
 STATE_LIMIT, CONTROL_LIMIT, NO_PROGRESS, COST_INCREASE, REGULARIZATION_MAX, PROJECTION_FAIL) = range(12)

MODEL_DOUBLE_INTEGRATOR, MODEL_CARTPOLE, MODEL_QUADROTOR, MODEL_HYBRID_DOUBLE_INTEGRATOR, MODEL_VECTOR, MODEL_INFEASIBLE = 0, 1, 2, 3, 4, 5
STEP_DOUBLE_INTEGRATOR, STEP_CARTPOLE, STEP_LINEAR_MAP = 0, 1, 2
TO_VECTOR_N, TO_VECTOR_M = 6, 3
RK4, RK3, EULER = 0, 1, 2
COST_DIAGONAL, COST_QUADRATIC, COST_DIAGONAL_QUAT, COST_ERROR_QUADRATIC = 0, 1, 2, 3
CONE_ZERO, CONE_NEGATIVE_ORTHANT, CONE_SECOND_ORDER, CONE_POSITIVE_ORTHANT, CONE_IDENTITY = range(5)
CON_GOAL, CON_BOUND, CON_NORM, CON_CIRCLE, CON_SPHERE, CON_LINEAR, CON_COLLISION, CON_QUATVEC = range(8)


class DimensionMismatch(ValueError):
    """Julia ``DimensionMismatch`` (src/problem.jl:64-68, src/constraint_list.jl:109)."""


class ArgumentError(ValueError):
    """Julia ``ArgumentError`` (src/problem.jl:88, src/constraints.jl:712)."""


class HipError(RuntimeError):
    """HIP runtime failure or no usable device."""


class UnsupportedError(NotImplementedError):
    pass


class ConeError(RuntimeError):
    """``ErrorException("Invalid second-order cone projection")`` (src/cones.jl:124)."""


class HipLibraryMissing(ImportError):
    pass


_ERRORS = {
    TO_ERR_DIMENSION_MISMATCH: DimensionMismatch,
    TO_ERR_ARGUMENT: ArgumentError,
    TO_ERR_ASSERTION: AssertionError,
    TO_ERR_HIP: HipError,
    TO_ERR_UNSUPPORTED: UnsupportedError,
    TO_ERR_NULL: ValueError,
    TO_ERR_CONE: ConeError,
}


class CostDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("terminal", C.c_int32),
        ("Q", C.c_double * (TO_MAX_N * TO_MAX_N)),
        ("R", C.c_double * (TO_MAX_M * TO_MAX_M)),
        ("H", C.c_double * (TO_MAX_M * TO_MAX_N)),
        ("q", C.c_double * TO_MAX_N), ("r", C.c_double * TO_MAX_M),
        ("c", C.c_double), ("w", C.c_double),
        ("q_ref", C.c_double * 4), ("q_ind", C.c_int32 * 4),
    ]


class ConstraintDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("sense", C.c_int32), ("k_first", C.c_int32), ("k_last", C.c_int32),
        ("p", C.c_int32), ("n_inds", C.c_int32), ("inds", C.c_int32 * TO_MAX_CON_INDS),
        ("n_params", C.c_int32), ("params", C.c_double * TO_MAX_CON_PARAMS),
    ]


class StepModel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("n_out", C.c_int32), ("params", C.c_double * 60)]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("integrator", C.c_int32),
        ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("B", C.c_int32),
        ("model_params", C.c_double * 16), ("t0", C.c_double), ("tf", C.c_double),
        ("dt", C.POINTER(C.c_double)),
        ("n_costs", C.c_int32), ("costs", C.POINTER(CostDesc)), ("cost_index", C.POINTER(C.c_int32)),
        ("n_constraints", C.c_int32), ("constraints", C.POINTER(ConstraintDesc)),
        ("step_models", C.POINTER(StepModel)),
    ]


class SolverOpts(C.Structure):
    _fields_ = [
        ("cost_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("iterations", C.c_int32), ("dJ_counter_limit", C.c_int32),
        ("iterations_linesearch", C.c_int32), ("reserved0", C.c_int32),
        ("line_search_lower_bound", C.c_double), ("line_search_upper_bound", C.c_double),
        ("line_search_decrease_factor", C.c_double),
        ("bp_reg_initial", C.c_double), ("bp_reg_increase_factor", C.c_double),
        ("bp_reg_min", C.c_double), ("bp_reg_max", C.c_double), ("bp_reg_fp", C.c_double),
        ("max_cost_value", C.c_double), ("max_state_value", C.c_double), ("max_control_value", C.c_double),
        ("constraint_tolerance", C.c_double), ("cost_tolerance_intermediate", C.c_double),
        ("penalty_initial", C.c_double), ("penalty_scaling", C.c_double), ("penalty_max", C.c_double),
        ("dual_max", C.c_double),
        ("iterations_outer", C.c_int32), ("cost_dt_scaling", C.c_int32),
        ("iterations_total", C.c_int32), ("al_full_newton", C.c_int32),
        # projected-Newton polish (Altro ProjectedNewtonSolver options)
        ("projected_newton_tolerance", C.c_double), ("active_set_tolerance_pn", C.c_double),
        ("rho_chol", C.c_double), ("rho_primal", C.c_double), ("r_threshold", C.c_double),
        ("n_steps", C.c_int32), ("projected_newton", C.c_int32),
    ]


class SolveStats(C.Structure):
    _fields_ = [
        ("iterations", C.POINTER(C.c_int32)), ("iterations_outer", C.POINTER(C.c_int32)),
        ("status", C.POINTER(C.c_int32)),
        ("cost", C.POINTER(C.c_double)), ("dJ", C.POINTER(C.c_double)), ("gradient", C.POINTER(C.c_double)),
        ("c_max", C.POINTER(C.c_double)), ("penalty_max", C.POINTER(C.c_double)),
        ("iterations_pn", C.POINTER(C.c_int32)),
        ("total_iterations", C.c_int64), ("batch_steps", C.c_int32), ("reserved", C.c_int32),
        ("solve_ms", C.c_double),
    ]


_H = C.c_void_p
_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)

# name -> argtypes (restype is int unless listed in _SPECIAL)
SIGNATURES = {
    "abi_version": [],
    "device_count": [C.POINTER(C.c_int)],
    "default_options": [C.POINTER(SolverOpts)],
    "create": [C.POINTER(ProblemDesc), C.POINTER(SolverOpts), C.c_int, C.POINTER(_H)],
    "destroy": [_H],
    "set_options": [_H, C.POINTER(SolverOpts)],
    "get_options": [_H, C.POINTER(SolverOpts)],
    "sync": [_H],
    "dims": [_H, _PI, _PI, _PI, _PI, _PI],
    "num_constraints": [_H, _PI],
    "set_initial_state": [_H, _PD],
    "set_controls": [_H, _PD],
    "set_states": [_H, _PD],
    "set_controls_uniform": [_H, _PD],
    "get_states": [_H, _PD],
    "get_controls": [_H, _PD],
    "get_initial_state": [_H, _PD],
    "set_cost": [_H, C.c_int32, C.POINTER(CostDesc)],
    "set_constraint": [_H, C.c_int32, C.POINTER(ConstraintDesc)],
    "set_cost_linear_batch": [_H, C.c_int32, _PD, _PD],
    "clear_cost_linear_batch": [_H],
    "set_constraint_params_batch": [_H, C.c_int32, _PD],
    "clear_constraint_params_batch": [_H],
    "rollout": [_H],
    "cost": [_H, _PD],
    "stage_costs": [_H, _PD],
    "expand": [_H],
    "backward": [_H],
    "forward": [_H, _PI, _PD],
    "ilqr_solve": [_H, C.POINTER(SolveStats)],
    "al_solve": [_H, C.POINTER(SolveStats)],
    "pn_solve": [_H, C.POINTER(SolveStats)],
    "altro_solve": [_H, C.POINTER(SolveStats)],
    "dynamics_defect": [_H, _PD],
    "get_dynamics_jacobians": [_H, _PD, _PD],
    "get_cost_expansion": [_H, _PD, _PD, _PD, _PD, _PD],
    "get_gains": [_H, _PD, _PD, _PD, _PD],
    "get_cost_to_go": [_H, _PD, _PD],
    "infeasible_controls": [_H],
    "cost_expansion": [_H, _PD, _PD],
    "discrete_jacobian": [_H, _PD],
    "evaluate_constraints": [_H, C.c_int32, _PD],
    "constraint_jacobians": [_H, C.c_int32, _PD],
    "constraint_hessians": [_H, C.c_int32, _PD, _PD],
    "constraint_info": [_H, C.c_int32, _PI, _PI, _PI, _PI],
    "knot_dims": [_H, _PI, _PI],
    "max_violation": [_H, _PD],
    "get_duals": [_H, C.c_int32, _PD, _PD],
    "set_duals": [_H, C.c_int32, _PD, _PD],
    "reset_duals": [_H],
    "dual_update": [_H],
    "al_cost": [_H, _PD],
    "cone_projection": [C.c_int, C.c_int32, C.c_int32, C.c_int64, _PD, _PD, _PI],
    "cone_projection_jacobian": [C.c_int, C.c_int32, C.c_int32, C.c_int64, _PD, _PD],
    "cone_projection_hessian": [C.c_int, C.c_int32, C.c_int32, C.c_int64, _PD, _PD, _PD],
}
# entry points only the HIP product library has (device pointers / streams)
HIP_ONLY = {
    "get_states_device": [_H, C.c_void_p],
    "get_controls_device": [_H, C.c_void_p],
    "comm_unique_id": [C.c_void_p],
    "comm_init_rank": [_H, C.c_int32, C.c_int32, C.c_void_p],
    "allgather": [_H, C.c_void_p, C.c_void_p],
    "comm_shards": [_H, _PI, _PI, C.POINTER(C.c_int64), _PI],
    "allgather_stats": [_H, _PI, _PI, _PD],
    "comm_destroy": [_H],
    "solver_path": [_H, _PI],
    "set_profiling": [_H, C.c_int],
    "get_profile": [_H, _PD, C.POINTER(C.c_int64)],
    "reset_profile": [_H],
    "ilqr_solve_async": [_H, C.POINTER(SolveStats)],
    "al_solve_async": [_H, C.POINTER(SolveStats)],
    "altro_solve_async": [_H, C.POINTER(SolveStats)],
    "solve_wait": [_H],
    "solve_progress": [_H, _PI, _PI, _PI],
    "solve_wait_below": [_H, C.c_int32],
}


class Library:
    """A shared object exporting the trajopt C-ABI under ``prefix``."""

    def __init__(self, path, prefix="to_", hip=True):
        self.path = str(path)
        self.prefix = prefix
        self.dll = C.CDLL(self.path)
        self._fn = {}
        sigs = dict(SIGNATURES)
        if hip:
            sigs.update(HIP_ONLY)
        for name, argtypes in sigs.items():
            f = getattr(self.dll, prefix + name)  # AttributeError if the symbol is missing: loud
            f.argtypes = argtypes
            f.restype = C.c_int
            self._fn[name] = f
        le = getattr(self.dll, prefix + "last_error")
        le.argtypes = []
        le.restype = C.c_char_p
        self._last_error = le
        bid = getattr(self.dll, prefix + "build_id", None)  # the CPU oracle (tests only) has none
        if bid is not None:
            bid.argtypes = []
            bid.restype = C.c_char_p
        self._build_id = bid
        if hip:
            st = getattr(self.dll, prefix + "stream")
            st.argtypes = [_H]
            st.restype = C.c_void_p
            self._fn["stream"] = st

    def last_error(self):
        s = self._last_error()
        return s.decode("utf-8", "replace") if s else ""

    def raw(self, name):
        return self._fn[name]

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc < 0:
            raise _ERRORS.get(rc, RuntimeError)(f"{self.prefix}{name}: {self.last_error()} (code {rc})")
        return rc

    def build_id(self):
        return self._build_id().decode() if self._build_id is not None else None

    def abi_version(self):
        return self._fn["abi_version"]()

    def device_count(self):
        n = C.c_int(0)
        self.call("device_count", C.byref(n))
        return n.value

    def default_options(self):
        o = SolverOpts()
        self.call("default_options", C.byref(o))
        return o


_PKG_DIR = Path(__file__).resolve().parent
HIP_LIBRARY_PATH = _PKG_DIR / "csrc" / "libtrajopt_hip.so"
_hip_library = None


def load_hip_library():
    """Load ``csrc/libtrajopt_hip.so`` (built by ``__graft_entry__.build()``).  Fails loudly if absent."""
    global _hip_library
    if _hip_library is None:
        path = Path(os.environ.get("TRAJOPT_HIP_LIBRARY", HIP_LIBRARY_PATH))
        if not path.exists():
            raise HipLibraryMissing(
                f"{path} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        _hip_library = Library(path, prefix="to_", hip=True)
        if _hip_library.abi_version() != TO_ABI_VERSION:
            raise HipLibraryMissing("libtrajopt_hip.so ABI version mismatch; rebuild")
    return _hip_library
