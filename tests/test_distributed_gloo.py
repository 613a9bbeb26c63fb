"""N>1 path on CPU: two gloo ranks shard the batch by GLOBAL trajectory index, solve their shards (the oracle
stands in for the GPU library — tests only) and all-gather the converged trajectories with the same
``TrajectoryGather`` bench.py uses over RCCL.  The gathered result must equal a single-process solve of the
whole batch: sharding is independent of world size and there is no data-path collective."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import trajopt_amd as T
    from oracle_binding import load_oracle
    from trajectoryoptimization_jl_amd import configs
    from trajectoryoptimization_jl_amd.distributed import TrajectoryGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b0, cnt = shard_range(rank, world, total)     # 13 trajectories over 2 ranks: shards of 7 and 6
    prob = configs.cartpole_problem(batch=cnt, N=41, tf=2.0, b_offset=b0, lib=load_oracle())
    s = T.iLQRSolver(prob, iterations=25).solve()
    g = TrajectoryGather(prob, dist)
    X, U = g()
    its, st, J = g.stats(s)
    if rank == 0:
        np.savez(Path(out_dir) / "gathered.npz", X=X.numpy(), U=U.numpy(), its=its, st=st, J=J, counts=np.array(g.counts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allgather(tmp_path, oracle):
    """Unequal shards (13 = 7 + 6): trajectories and the per-trajectory stats gathered in global order equal a
    single-process solve of the whole batch."""
    import torch.multiprocessing as mp
    import trajopt_amd as T
    from trajectoryoptimization_jl_amd import configs
    total, world = 13, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "gathered.npz")
    assert list(g["counts"]) == [7, 6]
    prob = configs.cartpole_problem(batch=total, N=41, tf=2.0, lib=oracle)
    s = T.iLQRSolver(prob, iterations=25).solve()
    np.testing.assert_array_equal(g["its"], s.stats["iterations"])
    np.testing.assert_array_equal(g["st"], s.stats["status"])
    np.testing.assert_array_equal(g["J"], s.stats["cost"])
    np.testing.assert_array_equal(g["X"], T.states(prob))
    np.testing.assert_array_equal(g["U"], T.controls(prob))


def test_shard_range_partitions_the_batch():
    from trajectoryoptimization_jl_amd.distributed import shard_range
    for total, world in ((13, 2), (32768, 8), (10, 4), (3, 3)):
        parts = [shard_range(r, world, total) for r in range(world)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == total
        assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_global_index_sharding():
    from trajectoryoptimization_jl_amd import configs
    full = configs.quadrotor_x0(12)
    parts = np.concatenate([configs.quadrotor_x0(4, b_offset=4 * r) for r in range(3)])
    np.testing.assert_array_equal(full, parts)
    assert np.all(full[0, :3] == 0) and np.all(np.abs(full[1:, :3]) <= 1)
    c = configs.cartpole_x0(8, b_offset=0)
    assert np.all(c[0] == 0) and np.all(np.abs(c[1:, 0]) <= 0.5) and np.all(np.abs(c[1:, 1]) <= 0.3)
