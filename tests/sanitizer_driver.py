"""Driver of tests/test_sanitizers.py: runs inside a python started with LD_PRELOAD=libasan.so.  Loads the ASan + UBSan build of the
oracle (argv[1]) and of the host build of the projected-Newton kernel source (argv[2]) and puts a bounded set of solves of every model
family and constraint kind through them.  Any sanitizer report aborts the process (non-zero exit); the parent also scans stderr."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

import trajopt_amd as T  # noqa: E402
from trajectoryoptimization_jl_amd import configs  # noqa: E402
import oracle_binding  # noqa: E402

oracle = oracle_binding._bind(sys.argv[1])
done = []


def ran(name, solver):
    st = solver.stats["status"]
    assert solver.total_iterations > 0 or (st != 0).all(), name
    done.append(name)


# every model family / solver / constraint kind, small
ran("cartpole ilqr", T.iLQRSolver(configs.cartpole_problem(batch=5, N=41, tf=2.0, lib=oracle), iterations=30).solve())
ran("cartpole altro", T.ALTROSolver(configs.cartpole_problem(batch=3, constrained=True, lib=oracle)).solve())
ran("quickstart al", T.ALSolver(configs.quickstart_problem(batch=2, lib=oracle)).solve())
ran("quadrotor ilqr", T.iLQRSolver(configs.quadrotor_problem(batch=3, N=31, tf=1.0, lib=oracle), iterations=15).solve())
ran("quadrotor altro", T.ALTROSolver(configs.quadrotor_problem(batch=3, N=31, tf=1.5, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=oracle),
                                     iterations_total=60).solve())
for rot in ("mrp", "rp"):
    model = T.Quadrotor(rotation=rot)
    n, m = model.dims()
    x0 = np.zeros(n); x0[:3] = [0.5, -0.3, 1.0]
    xf = np.zeros(n); xf[:3] = [0.0, 0.0, 1.5]
    obj = T.LQRObjective(np.full(n, 0.1), np.full(m, 0.01), np.full(n, 10.0), xf, 21)
    p = T.Problem(model, obj, x0, 1.0, xf=xf, lib=oracle, batch=2)
    T.initial_controls(p, model.hover_control())
    ran("quadrotor " + rot, T.iLQRSolver(p, iterations=10).solve())
# phase API + operators on a constrained problem
p = configs.quadrotor_problem(batch=2, N=21, tf=1.0, constrained=True, u_norm_max=2.6, lib=oracle)
T.rollout(p)
from trajopt_amd import internal as I  # noqa: E402
I.dual_update(p); I.expand(p); I.backwardpass(p); I.forwardpass(p)
for i in range(len(p.constraints)):
    T.evaluate_constraints(p, i); T.constraint_jacobians(p, i)
T.max_violation(p); T.cost(p); T.stage_costs(p)
done.append("phase api")
# cones
for cone in (T.SecondOrderCone(), T.NegativeOrthant(), T.ZeroCone()):
    x = np.random.default_rng(0).normal(size=(7, 4))
    T.projection(cone, x, lib=oracle); T.grad_projection(cone, x, lib=oracle); T.hess_projection(cone, x, x, lib=oracle)
done.append("cones")
# the kernel source of the polish, host build
if len(sys.argv) > 2:
    pn = C.CDLL(sys.argv[2])
    pn.pn_host_solve.restype = C.c_int
    prob = configs.cartpole_problem(batch=3, constrained=True, lib=oracle)
    T.ALSolver(prob, constraint_tolerance=1e-3).solve()
    X, U = np.ascontiguousarray(T.states(prob)), np.ascontiguousarray(T.controls(prob))
    x0 = np.zeros((prob.B, prob.n)); prob._call("get_initial_state", prob._pd(x0))
    o = T.SolverOptions(lib=prob._lib)
    st, ip, cm = np.zeros(prob.B, np.int32), np.zeros(prob.B, np.int32), np.zeros(prob.B)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert pn.pn_host_solve(C.byref(prob._desc), C.byref(o._o), pd(x0), pd(X), pd(U), pi(st), pi(ip), pd(cm)) == 0
    assert (cm < 1e-6).all()
    done.append("pn kernel source on the host")
print("SANITIZER_DRIVER_OK", ", ".join(done))
