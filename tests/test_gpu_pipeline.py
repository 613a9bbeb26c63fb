"""Pipelined solves (to_solve_progress / to_solve_wait_below, api.SolvePipeline): several handles of the same shape, the next solve
admitted while the one in flight drains.  Trajectories are independent (one Z per problem, src/problem.jl:330-340), and every per-batch-step
kernel choice of the solve loop is between bit-identical kernels — so a solve that shared the device with another one must EQUAL the solve that
had the device to itself, bit for bit, whatever the admission threshold."""
import ctypes as C
import time

import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs

pytestmark = pytest.mark.gpu


def _snapshot(solver):
    return ({k: v.copy() for k, v in solver.stats.items()}, T.states(solver.prob), T.controls(solver.prob))


def _assert_same(a, b):
    for k in a[0]:
        np.testing.assert_array_equal(a[0][k], b[0][k], err_msg=k)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])


@pytest.mark.parametrize("kind", ["quadrotor_ilqr", "quadrotor_altro", "cartpole_ilqr"])
@pytest.mark.parametrize("depth,admit", [(2, 0.5), (2, 1.0), (3, 0.1)])
def test_pipelined_solves_are_bit_identical(kind, depth, admit, hip):
    def mk():
        if kind == "quadrotor_ilqr":
            p = configs.quadrotor_problem(batch=1500, N=101, tf=5.0, lib=hip)
            return T.iLQRSolver(p)
        if kind == "quadrotor_altro":
            p = configs.quadrotor_problem(batch=700, N=101, tf=5.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=hip)
            return T.ALTROSolver(p, n_steps=configs.C5_PN_STEPS)
        return T.iLQRSolver(configs.cartpole_problem(batch=20000, N=41, tf=2.0, lib=hip), iterations=60)
    ref = mk()
    u0 = T.controls(ref.prob)[0, 0].copy()
    ref.solve()
    want = _snapshot(ref)
    assert len(set(want[0]["iterations"])) > 5      # the batch drains unevenly: there is a tail to overlap
    solvers = [mk() for _ in range(depth)]
    got = {}
    pipe = T.SolvePipeline(solvers, admit_below=int(admit * ref.prob.B), on_done=lambda job, s: got.__setitem__(job, _snapshot(s)))
    jobs = 2 * depth + 1
    for _ in range(jobs):
        pipe.submit(lambda p: T.initial_controls(p, u0))
    pipe.drain()
    assert sorted(got) == list(range(jobs)) and pipe.total_iterations == jobs * int(want[0]["iterations"].sum())
    for job in range(jobs):
        _assert_same(want, got[job])


def test_progress_and_wait_below(hip):
    """to_solve_progress reports B right after to_*_solve_async, a non-increasing count while the solve runs and 0 once its stage has
    ended; to_solve_wait_below returns no earlier than that and at once when nothing is in flight; both are legal while a solve is in
    flight (every other call on the handle is refused)."""
    p = configs.quadrotor_problem(batch=512, N=101, tf=5.0, lib=hip)
    s = T.iLQRSolver(p)
    assert s.progress() == (0, 0, False)
    s.wait_below(0)                                  # nothing in flight: returns at once
    with pytest.raises(T.ArgumentError):
        p._call("solve_wait_below", -1)
    s.solve_async()
    a0, _, fl = s.progress()
    assert a0 == p.B and fl
    seen = [a0]
    s.wait_below(p.B // 2)
    a1, steps1, fl = s.progress()
    assert a1 <= p.B // 2 and steps1 >= 1 and fl
    while True:
        a, st, fl = s.progress()
        assert a <= seen[-1]
        seen.append(a)
        if a == 0:
            break
        time.sleep(1e-3)
    with pytest.raises(T.ArgumentError, match="in flight"):
        T.states(p)                                  # still refused until to_solve_wait, even with the stage over
    s.wait()
    assert s.progress()[0] == 0 and not s.progress()[2] and s.progress()[1] == s.batch_steps
    assert s.stats["iterations"].max() >= 20


def test_repacked_working_set_carries_duals(hip, monkeypatch):
    """the repacked working set of an iLQR solve must carry the per-trajectory duals and penalties of a constrained
    problem (expansion, forward pass and cost read them through the tile of the working position).  A hand-built AL loop — iLQR solve,
    dual update, iLQR solve — on 40 000 constrained Cartpoles with moves allowed down to 2 048 must equal the loop that never moves."""
    out = []
    for rp in ("0", "2048"):
        monkeypatch.setenv("TRAJOPT_REPACK", rp)
        p = configs.cartpole_problem(batch=40000, N=41, tf=2.0, constrained=True, u_bnd=10.0, lib=hip)
        info = (C.c_int32 * 8)()
        p._call("solver_path", info)
        assert info[0] == 2 and info[1] == 1, "the constrained batch must take the fused lane path"
        s = T.iLQRSolver(p, iterations=25)
        s.solve()
        from trajopt_amd import internal as I
        I.dual_update(p)
        lam = I.get_duals(p, 0)[0]
        assert np.abs(lam).max() > 0 and np.ptp(np.abs(lam).reshape(p.B, -1).max(axis=1)) > 0   # per-trajectory duals differ
        s.solve()
        out.append(({k: v.copy() for k, v in s.stats.items()}, T.states(p), T.controls(p), T.cost(p)))
    _assert_same(out[0][:3], out[1][:3])
    np.testing.assert_array_equal(out[0][3], out[1][3])
    assert len(set(out[0][0]["iterations"])) > 3


@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_pipelined_full_size_baseline_shapes(workload, hip):
    """bench.py's pipelined lines at BASELINE's own sizes — C2 (Cartpole N=101, B=1024) and C3 (Quadrotor N=201, B=4096), four handles, earliest
    admission: every one of the jobs equals the unpipelined solve bit for bit (iterations, status, states, controls of all trajectories)."""
    def mk():
        p = configs.cartpole_problem(batch=1024, lib=hip) if workload == "C2" else configs.quadrotor_problem(batch=4096, lib=hip)
        return T.iLQRSolver(p)
    ref = mk()
    u0 = T.controls(ref.prob)[0, 0].copy()
    ref.solve()
    want = _snapshot(ref)
    solvers = [ref] + [mk() for _ in range(3)]
    bad = []
    def check(job, s):
        got = _snapshot(s)
        ok = all(np.array_equal(want[0][k], got[0][k]) for k in want[0]) and np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
        if not ok:
            bad.append(job)
    pipe = T.SolvePipeline(solvers, on_done=check)
    for _ in range(9):
        pipe.submit(lambda p: T.initial_controls(p, u0))
    pipe.drain()
    assert not bad, f"pipelined jobs {bad} differ from the unpipelined solve"
    assert pipe.total_iterations == 9 * int(want[0]["iterations"].sum())
