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
