"""Worker of tests/test_gpu_multi.py: one process per GPU.  argv: rank world total_batch id_file out_file equal(0/1)
Solves its contiguous shard of a Cartpole batch on device `rank`, runs the C-ABI's RCCL gathers (to_comm_init_rank /
to_allgather / to_allgather_stats) and, on rank 0, saves what it gathered."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np
import torch

import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
from trajectoryoptimization_jl_amd.distributed import shard_range


def main():
    rank, world, total = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    id_file, out_file = Path(sys.argv[4]), Path(sys.argv[5])
    lib = T.load_hip_library()
    b0, cnt = shard_range(rank, world, total)
    # TRAJOPT_WORKER_SAME_DEVICE=1 (the "dry" communicator test of a 1-GPU box): every rank on device 0
    dev = 0 if os.environ.get("TRAJOPT_WORKER_SAME_DEVICE") == "1" else rank
    torch.cuda.set_device(dev)
    prob = configs.cartpole_problem(batch=cnt, b_offset=b0, N=41, tf=2.0, device=dev, lib=lib)
    sv = T.iLQRSolver(prob, iterations=25).solve()
    if rank == 0:
        buf = (C.c_char * 128)()
        lib.call("comm_unique_id", buf)
        tmp = id_file.with_suffix(".tmp")
        tmp.write_bytes(buf.raw)
        os.replace(tmp, id_file)
    for _ in range(600):
        if id_file.exists():
            break
        time.sleep(0.1)
    uid = (C.c_char * 128).from_buffer_copy(id_file.read_bytes())
    print(f"[rank {rank}] to_comm_init_rank(nranks={world}) on device {dev}", flush=True)
    prob._call("comm_init_rank", world, rank, uid)
    print(f"[rank {rank}] communicator up", flush=True)
    nr, rk, tot = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    counts = (C.c_int32 * world)()
    prob._call("comm_shards", C.byref(nr), C.byref(rk), C.byref(tot), counts)
    n, m, N = prob.dims()
    xg = torch.zeros((total, N, n), dtype=torch.float64, device=f"cuda:{dev}")
    ug = torch.zeros((total, N - 1, m), dtype=torch.float64, device=f"cuda:{dev}")
    prob._call("allgather", C.c_void_p(xg.data_ptr()), C.c_void_p(ug.data_ptr()))
    its, st, J = np.zeros(total, np.int32), np.zeros(total, np.int32), np.zeros(total)
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    prob._call("allgather_stats", pi(its), pi(st), J.ctypes.data_as(C.POINTER(C.c_double)))
    torch.cuda.synchronize()
    np.savez(str(out_file) + f".rank{rank}.npz", X=xg.cpu().numpy(), U=ug.cpu().numpy(), its=its, st=st, J=J,
             counts=np.array(list(counts)), total=tot.value, mine_its=sv.stats["iterations"])
    prob._call("comm_destroy")


if __name__ == "__main__":
    main()
