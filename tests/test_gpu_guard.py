"""TRAJOPT_GUARD=1 (csrc/handle.h): every handle-owned device array between two red zones, checked after every batch step and phase call.
GPU AddressSanitizer is not available on this pool; this is the device-side counterpart of tests/test_sanitizers.py.
(a) solves of every kernel family run clean under the guard and give the unguarded results bit for bit; (b) a deliberate one-double
overrun behind the nominal states (TRAJOPT_GUARD_SELFTEST) is reported, by array name and batch step, as a TO_ERR_HIP."""
import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs

pytestmark = pytest.mark.gpu

CASES = {
    "cartpole_scan": lambda hip: T.iLQRSolver(configs.cartpole_problem(batch=200, lib=hip), iterations=60),
    "cartpole_lane_repack": lambda hip: T.iLQRSolver(configs.cartpole_problem(batch=40000, N=41, tf=2.0, lib=hip), iterations=40),
    "cartpole_altro": lambda hip: T.ALTROSolver(configs.cartpole_problem(batch=70, constrained=True, lib=hip)),
    "quadrotor_ilqr": lambda hip: T.iLQRSolver(configs.quadrotor_problem(batch=300, N=61, tf=3.0, lib=hip)),
    "quadrotor_altro": lambda hip: T.ALTROSolver(configs.quadrotor_problem(batch=300, N=61, tf=3.0, constrained=True, goal_inds=configs.C5_GOAL_INDS, lib=hip),
                                                 n_steps=configs.C5_PN_STEPS),
    "quickstart_al": lambda hip: T.ALSolver(configs.quickstart_problem(batch=5, lib=hip)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_solves_run_clean_under_the_guard(case, hip, monkeypatch):
    out = []
    for guard in ("0", "1"):
        monkeypatch.setenv("TRAJOPT_GUARD", guard)
        monkeypatch.setenv("TRAJOPT_REPACK", "2048")
        s = CASES[case](hip)
        s.solve()
        s.solve()          # a second solve on the same handle (working sets, polish workspace, event pools re-used)
        out.append(({k: v.copy() for k, v in s.stats.items()}, T.states(s.prob), T.controls(s.prob)))
    for k in out[0][0]:
        np.testing.assert_array_equal(out[0][0][k], out[1][0][k], err_msg=k)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][2], out[1][2])


def test_phase_api_under_the_guard(hip, monkeypatch):
    from trajopt_amd import internal as I
    monkeypatch.setenv("TRAJOPT_GUARD", "1")
    for p in (configs.cartpole_problem(batch=70, constrained=True, lib=hip),
              configs.quadrotor_problem(batch=40, N=41, tf=1.0, constrained=True, u_norm_max=2.6, lib=hip)):
        T.rollout(p); I.dual_update(p); I.expand(p); I.backwardpass(p); I.forwardpass(p)
        T.states(p); T.controls(p); T.cost(p); T.stage_costs(p); I.gains(p); I.dynamics_jacobians(p); I.cost_expansion(p)


def test_the_guard_finds_an_overrun(hip, monkeypatch):
    monkeypatch.setenv("TRAJOPT_GUARD", "1")
    monkeypatch.setenv("TRAJOPT_GUARD_SELFTEST", "1")
    p = configs.cartpole_problem(batch=100, N=41, tf=2.0, lib=hip)
    with pytest.raises(T.HipError, match=r"guard: a red zone of device array '&a\.Xs'.*batch step 0"):
        T.iLQRSolver(p, iterations=10).solve()
    monkeypatch.delenv("TRAJOPT_GUARD_SELFTEST")
    # without the guard the same stray store goes unnoticed (it lands in allocator slack): the mode is what finds it
