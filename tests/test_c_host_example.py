"""examples/cartpole_altro.c: a plain-C host of the C-ABI (the drop-in boundary without the Python mirror).  CPU: it compiles as C,
links against libtrajopt_hip.so and fails loudly without a device; GPU: it reproduces the reference's published ALTRO result
(examples/Cartpole.ipynb cell 19: J = 1.552558743680986) for the trajectory that starts at the notebook's x0."""
import os
import re
import subprocess
from pathlib import Path

import pytest

import trajopt_amd as T

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "trajectoryoptimization.jl_amd" / "csrc"


def _build(tmp_path):
    exe = tmp_path / "cartpole_altro"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "examples" / "cartpole_altro.c"),
                    "-L", str(CSRC), "-ltrajopt_hip", "-lm", "-o", str(exe)], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = f"{CSRC}:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_c_host_compiles_links_and_refuses_without_a_device(tmp_path, hip):   # `hip`: the library, loaded in the order conftest.py prescribes
    lib = hip
    try:
        ndev = lib.device_count()
    except T.HipError:
        ndev = 0
    exe, env = _build(tmp_path)
    if ndev > 0:
        pytest.skip("a GPU is visible: the run is covered by the gpu test")
    r = subprocess.run([str(exe), "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_host_reproduces_the_notebook_result(tmp_path, hip):
    exe, env = _build(tmp_path)
    r = subprocess.run([str(exe), "16"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"trajectory 0: iLQR (\d+) outer (\d+) projections (\d+) status (\d+) J=([0-9.]+) c_max=([0-9.e+-]+)", r.stdout)
    assert m, r.stdout
    assert abs(float(m.group(5)) - 1.552558743680986) <= 1e-6 * 1.5525587 and float(m.group(6)) <= 1e-6
    assert "converged=16" in r.stdout
