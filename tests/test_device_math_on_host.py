"""The device-side model math of csrc/models.h (dynamics templated on double / Dual / MDual, RK steps, error-state maps)
compiled for the HOST with g++ and checked without a GPU: values against the oracle's dynamics, dual-number RK Jacobians
against central differences, chunk-mode dual numbers against single-direction ones (the lane expansion relies on their
agreeing), and the algebra of the error-state maps (E(x) G(x) = I, symmetric second-order terms).  The -m gpu suite then only
has to establish that the GPU runs this same code."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp("devmath") / "harness"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", str(ROOT / "tests" / "host_shim"),
                    "-I", str(ROOT / "trajectoryoptimization.jl_amd" / "csrc"), "-o", str(exe),
                    str(ROOT / "tests" / "host_shim" / "device_math_harness.cpp")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    vals, summary = {}, {}
    for line in out.splitlines():
        tok = line.split()
        if tok[1] in ("x", "u", "f", "step0", "step1", "step2"):
            vals.setdefault(tok[0], {})[tok[1]] = np.array([float(t) for t in tok[2:]])
        else:
            summary[tok[0]] = {tok[i]: float(tok[i + 1]) for i in range(1, len(tok), 2)}
    return vals, summary


MODELS = {"quat": lambda: T.Quadrotor(), "mrp": lambda: T.Quadrotor(rotation="mrp"), "rp": lambda: T.Quadrotor(rotation="rp"),
          "cartpole": lambda: T.Cartpole(), "di2": lambda: T.DoubleIntegrator(1.3, 2)}


@pytest.mark.parametrize("name", list(MODELS))
def test_device_dynamics_match_the_oracle(name, harness, oracle):
    vals, _ = harness
    model = MODELS[name]()
    params = (C.c_double * 16)(*(model.params() + [0.0] * (16 - len(model.params()))))
    x, u = np.ascontiguousarray(vals[name]["x"]), np.ascontiguousarray(vals[name]["u"])
    xd = np.zeros(model.n)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    oracle.call("dynamics", model.model_id, params, pd(x), pd(u), pd(xd))
    # the device code multiplies by reciprocals and merges a few products where the oracle divides: a few ulp
    np.testing.assert_allclose(vals[name]["f"], xd, rtol=1e-13, atol=1e-13)
    # one discrete step with every integrator (the Cartpole's later RK stages take sin / cos by angle addition from stage 1's:
    # models.h trig_stage) against the oracle's step, which evaluates sin / cos of every stage point directly
    for integ in (0, 1, 2):
        xn = np.zeros(model.n)
        oracle.call("discrete_dynamics", model.model_id, params, integ, pd(x), pd(u), 0.05, pd(xn))
        np.testing.assert_allclose(vals[name][f"step{integ}"], xn, rtol=2e-15, atol=2e-16)


@pytest.mark.parametrize("name", list(MODELS))
def test_device_dual_numbers(name, harness):
    _, summary = harness
    s = summary[name]
    assert s["dual_vs_fd"] < 5e-8          # forward-mode RK Jacobians (RK4, RK3, Euler) against central differences
    assert s["mdual_vs_dual"] == 0.0       # chunk mode: every derivative component goes through the same operations
    assert s["left_inverse"] < 1e-14       # E(x) G(x) = I for quaternions (G'G = I) and for three-parameter attitudes (D^-1 D = I)
    assert s["hess_sym"] < 1e-15


def test_cartpole_stage_jacobian_by_hand(harness):
    """cartpole_rk4_jac — the RK4 Jacobian of the Cartpole by the chain rule over hand-derived stage partials, what the lane expansion of
    large batches runs instead of chunk-mode dual numbers — is the same derivative of the same RK4 map: equal to the dual-number Jacobian
    to rounding (200 states: small and large angles, fast rotation, two step sizes) and to central differences."""
    _, summary = harness
    s = summary["cartpole_stage_jac"]
    assert s["vs_dual"] < 2e-14 * max(1.0, s["scale"])
    assert s["vs_fd"] < 2e-7
