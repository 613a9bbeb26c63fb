"""The projected-Newton KERNEL (csrc/k_pn.h) compiled for the host — its PN_FOR phases become plain loops (tests/host_shim/
pn_harness.cpp) — against the oracle's polish (oracle/oracle_pn.h) on the same trajectories.  The oracle factorises one banded
matrix row by row; the kernel factorises knot blocks with the blocks in (here: emulated) LDS and walks the horizon in sweeps.
Agreement here means the GPU kernel's logic is right before a GPU is involved; the -m gpu suite repeats the comparison on the
device.  No GPU needed."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def pn_host(tmp_path_factory):
    so = tmp_path_factory.mktemp("pnhost") / "libpn_host.so"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-I", str(ROOT / "tests" / "host_shim"),
                    "-I", str(ROOT / "trajectoryoptimization.jl_amd" / "csrc"), "-o", str(so),
                    str(ROOT / "tests" / "host_shim" / "pn_harness.cpp")], check=True)
    lib = C.CDLL(str(so))
    lib.pn_host_solve.restype = C.c_int
    lib.pn_host_last_error.restype = C.c_char_p
    return lib


def host_polish(lib, prob, opts=None):
    """Run the host build of the kernel on the problem's current trajectories; returns X, U, status, it_pn, cmax."""
    X, U = np.ascontiguousarray(T.states(prob)), np.ascontiguousarray(T.controls(prob))
    x0 = np.zeros((prob.B, prob.n)); prob._call("get_initial_state", prob._pd(x0))
    o = T.SolverOptions(lib=prob._lib)
    prob._call("get_options", C.byref(o._o))
    for k, v in (opts or {}).items():
        setattr(o, k, v)
    st, ip, cm = np.zeros(prob.B, np.int32), np.zeros(prob.B, np.int32), np.zeros(prob.B)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = lib.pn_host_solve(C.byref(prob._desc), C.byref(o._o), pd(x0), pd(X), pd(U), pi(st), pi(ip), pd(cm))
    assert rc == 0, lib.pn_host_last_error()
    return X, U, st, ip, cm


def al_then_perturb(build, scale, seed=9):
    prob = build()
    T.ALSolver(prob, constraint_tolerance=1e-3).solve()
    o = T.SolverOptions(lib=prob._lib)
    prob._call("set_options", C.byref(o._o))
    if scale:
        rng = np.random.default_rng(seed)
        X, U = T.states(prob), T.controls(prob)
        T.initial_states(prob, X + scale * rng.normal(size=X.shape))
        T.initial_controls(prob, U + scale * rng.normal(size=U.shape))
    return prob


def _quickstart(oracle, batch):
    p = configs.quickstart_problem(batch=batch, lib=oracle)
    T.initial_controls(p, np.array([0.1, 0.0]))
    return p


def _vector_mix(o):
    from test_model_vector import build, cartpole_mix
    return build(cartpole_mix(), o, batch=2, constrained=True)[0]


CASES = {
    # the general model vector (TO_MODEL_VECTOR): every step through the per-step table, double and dual-number paths
    "model_vector_cartpole_mix": (_vector_mix, 5e-4),
    "cartpole_bounds_goal": (lambda o: configs.cartpole_problem(batch=3, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=o), 1e-3),
    "quickstart_circle_soc_bound_goal": (lambda o: _quickstart(o, 2), 1e-3),
    "quadrotor_goal_soc": (lambda o: configs.quadrotor_problem(batch=3, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=o), 0.0),
    "quadrotor_goal_soc_perturbed": (lambda o: configs.quadrotor_problem(batch=2, N=41, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=o), 3e-4),
}


@pytest.mark.parametrize("name", list(CASES))
def test_kernel_source_on_host_matches_oracle(name, pn_host, oracle):
    build, scale = CASES[name]
    prob = al_then_perturb(lambda: build(oracle), scale)
    Xh, Uh, sth, iph, cmh = host_polish(pn_host, prob)
    s = T.ProjectedNewtonSolver(prob).solve()
    Xo, Uo = T.states(prob), T.controls(prob)
    assert np.array_equal(sth, s.stats["status"]) and np.all(sth == T.capi.SOLVE_SUCCEEDED)
    assert np.array_equal(iph, s.stats["iterations_pn"]) and np.all(iph >= 1)
    np.testing.assert_allclose(Xh, Xo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Uh, Uo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(cmh, s.stats["c_max"], rtol=0, atol=1e-9)
    assert cmh.max() <= 1e-6


def test_kernel_on_host_unconstrained_and_minimal_horizon(pn_host, oracle):
    """no constraint list (defects only) and N = 3: the kernel source against the oracle"""
    for N, tf in ((31, 1.5), (3, 0.1)):
        prob = configs.cartpole_problem(batch=2, N=N, tf=tf, lib=oracle)
        T.rollout(prob)
        X0 = T.states(prob).copy()
        rng = np.random.default_rng(N)
        T.initial_states(prob, X0 + 1e-3 * rng.normal(size=X0.shape))
        Xh, Uh, sth, iph, cmh = host_polish(pn_host, prob)
        s = T.ProjectedNewtonSolver(prob).solve()
        assert np.array_equal(sth, s.stats["status"]) and np.array_equal(iph, s.stats["iterations_pn"]) and np.all(iph >= 1)
        np.testing.assert_allclose(Xh, T.states(prob), rtol=0, atol=1e-10)
        np.testing.assert_allclose(Uh, T.controls(prob), rtol=0, atol=1e-10)


def test_index_helpers_of_the_factorisation(pn_host):
    """pn_div (e / d by one multiplication) and pn_tri_row (row of a packed lower triangle from a float square root) are exact over
    the domains the factorisation uses them on: e < 4096, d = 2 … 64."""
    pn_host.pn_host_index_selftest.restype = C.c_int
    assert pn_host.pn_host_index_selftest() == 0

