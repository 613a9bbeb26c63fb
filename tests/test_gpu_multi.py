"""The RCCL path of the C-ABI with MORE THAN ONE rank: runs whenever the box shows >= 2 GPUs, skipped otherwise
(the 1-GPU gpurun boxes).  One process per GPU, the 128-byte communicator id travels through a file (what a Julia host
without MPI would do), every rank solves its shard and gathers; every rank's gathered result must equal a single-device solve
of the whole batch.  Equal shards take one in-place ncclAllGather, unequal shards (13 = 7 + 6) the grouped ncclBroadcast."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("total", [12, 13])
def test_two_rank_rccl_allgather(total, hip, tmp_path):
    if hip.device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    id_file, out = tmp_path / "nccl_id.bin", tmp_path / "gather"
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_gpu_worker.py"), str(r), str(world), str(total),
                               str(id_file), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("multi-GPU worker timed out")
        assert p.returncode == 0, log.decode()[-3000:]
    ref = configs.cartpole_problem(batch=total, N=41, tf=2.0, lib=hip)
    sv = T.iLQRSolver(ref, iterations=25).solve()
    X, U = T.states(ref), T.controls(ref)
    for r in range(world):
        g = np.load(str(out) + f".rank{r}.npz")
        assert int(g["total"]) == total and list(g["counts"]) == [total - total // 2, total // 2]
        np.testing.assert_array_equal(g["X"], X)      # shards are solved independently: bit-identical to the one-device solve
        np.testing.assert_array_equal(g["U"], U)
        np.testing.assert_array_equal(g["its"], sv.stats["iterations"])
        np.testing.assert_array_equal(g["st"], sv.stats["status"])
        np.testing.assert_allclose(g["J"], sv.stats["cost"], rtol=0, atol=0)


def test_two_rank_bootstrap_on_one_gpu(hip, tmp_path):
    """What a 1-GPU box CAN show of the N > 1 path: two processes, both on device 0, go through the whole hand-shake — rank 0 creates
    the id (to_comm_unique_id), ships it through a file, both enter to_comm_init_rank(nranks = 2) and RCCL's bootstrap brings the two
    ranks together.  RCCL 2.26 then refuses the communicator ("Duplicate GPU detected": one device cannot hold two ranks), and that
    must surface on BOTH ranks as a clean TO_ERR_HIP carrying RCCL's message — no hang, no half-initialised handle.  Should a future
    RCCL allow it, the workers finish and the gather is checked like the two-GPU test."""
    import os
    world, total = 2, 12
    id_file, out = tmp_path / "nccl_id.bin", tmp_path / "gather"
    env = dict(os.environ, TRAJOPT_WORKER_SAME_DEVICE="1", NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_gpu_worker.py"), str(r), str(world), str(total),
                               str(id_file), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung in the communicator hand-shake")
        logs.append((p.returncode, log.decode()))
    if all(rc == 0 for rc, _ in logs):   # an RCCL that accepts two ranks on one device: the full check
        ref = configs.cartpole_problem(batch=total, N=41, tf=2.0, lib=hip)
        T.iLQRSolver(ref, iterations=25).solve()
        for r in range(world):
            np.testing.assert_array_equal(np.load(str(out) + f".rank{r}.npz")["X"], T.states(ref))
        return
    for r, (rc, log) in enumerate(logs):
        assert rc != 0 and f"[rank {r}] to_comm_init_rank(nranks=2)" in log, log[-2000:]     # both reached the hand-shake ...
        assert "to_comm_init_rank" in log and "HipError" in log, log[-2000:]                  # ... and failed through the C-ABI's error path
        assert "Duplicate GPU detected" in log or "invalid usage" in log, log[-2000:]
        assert "communicator up" not in log


def build_rccl_stub(dst):
    """tests/rccl_stub/rccl_stub.cpp -> dst (g++; links the HIP runtime the process already has)."""
    src = ROOT / "tests" / "rccl_stub" / "rccl_stub.cpp"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o", str(dst),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread", "-Wno-format-truncation"], check=True, capture_output=True)
    return dst


@pytest.mark.parametrize("world,total", [(2, 12), (2, 13), (3, 14), (4, 64)])
def test_multi_rank_collectives_on_one_gpu_through_the_stub(world, total, hip, tmp_path):
    """The N > 1 code of the C-ABI on a ONE-GPU box: RCCL refuses two ranks on one device, so the ranks' collective
    library is the shared-memory stand-in of tests/rccl_stub (TRAJOPT_RCCL_LIB) — same eight entry points, same signatures.  `world`
    processes, all on device 0, each with its contiguous shard: to_comm_init_rank(nranks = world) exchanges the shard sizes with
    ncclAllGather, to_allgather gathers (X, U) — in place with equal shards (12 = 6 + 6, 64 = 4 x 16), by grouped ncclBroadcast with
    unequal ones (13 = 7 + 6, 14 = 5 + 5 + 4) — and to_allgather_stats the per-trajectory integers and costs.  Every rank's result must
    equal the one-handle solve of the whole batch bit for bit (shards are solved independently; inputs come from the GLOBAL index)."""
    import os
    stub = build_rccl_stub(tmp_path / "librccl_stub.so")
    id_file, out = tmp_path / "nccl_id.bin", tmp_path / "gather"
    env = dict(os.environ, TRAJOPT_WORKER_SAME_DEVICE="1", TRAJOPT_RCCL_LIB=str(stub))
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_gpu_worker.py"), str(r), str(world), str(total),
                               str(id_file), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        assert p.returncode == 0, log.decode()[-3000:]
        assert "communicator up" in log.decode()
    ref = configs.cartpole_problem(batch=total, N=41, tf=2.0, lib=hip)
    sv = T.iLQRSolver(ref, iterations=25).solve()
    X, U = T.states(ref), T.controls(ref)
    base, extra = divmod(total, world)
    for r in range(world):
        g = np.load(str(out) + f".rank{r}.npz")
        assert int(g["total"]) == total and list(g["counts"]) == [base + (1 if q < extra else 0) for q in range(world)]
        np.testing.assert_array_equal(g["X"], X)
        np.testing.assert_array_equal(g["U"], U)
        np.testing.assert_array_equal(g["its"], sv.stats["iterations"])
        np.testing.assert_array_equal(g["st"], sv.stats["status"])
        np.testing.assert_array_equal(g["J"], sv.stats["cost"])
