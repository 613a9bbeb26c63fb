"""Multi-GPU plumbing: the batch shards across ranks as independent units — no collective
inside the iLQR/AL loops — and the converged trajectories are all-gathered once per solve, plus one small gather of
the per-trajectory stats (iterations, status, cost).

On GPUs both gathers are the library's own RCCL collectives over xGMI (``to_comm_init_rank`` / ``to_allgather`` /
``to_allgather_stats`` of the C-ABI: what a Julia host calls as well); ``torch.distributed`` only ships the 128-byte
communicator id and provides the barrier.  Shards may differ in size (``to_comm_shards`` reports what the communicator
saw).  The CPU tests (gloo, oracle as the per-rank library) gather host arrays with ``torch.distributed``."""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_offset(rank, batch_per_rank):
    """Global index of the first trajectory owned by ``rank`` (contiguous block partition of equal shards)."""
    return int(rank) * int(batch_per_rank)


def shard_range(rank, world, total):
    """Contiguous block partition of ``total`` trajectories over ``world`` ranks when it does not divide: the first
    ``total % world`` ranks own one more.  -> (first global index, count)."""
    base, extra = divmod(int(total), int(world))
    cnt = base + (1 if rank < extra else 0)
    return rank * base + min(rank, extra), cnt


class TrajectoryGather:
    """Pre-allocated buffers + one all-gather per array.  Rank-major output: X[B_total, N, n], U[B_total, N-1, m]
    (= the C-ABI's (n, N, B_total) column-major layout in global trajectory order)."""

    def __init__(self, prob, dist, device=None):
        import torch
        self.torch, self.dist, self.prob = torch, dist, prob
        live = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if live else 1
        self.rank = dist.get_rank() if live else 0
        n, m, N = prob.dims()
        B = prob.B
        self.on_device = device is not None
        dev = device if self.on_device else "cpu"
        if self.on_device:  # native RCCL communicator behind the handle; world == 1 runs the same code
            uid = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_char * 128)()
                prob._lib.call("comm_unique_id", buf)
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if self.world > 1:
                uid = uid.to(device)
                dist.broadcast(uid, 0)
                uid = uid.cpu()
            self._uid = (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes()))
            prob._call("comm_init_rank", self.world, self.rank, self._uid)
            nr, rk, tot = C.c_int32(0), C.c_int32(0), C.c_int64(0)
            cnt = (C.c_int32 * self.world)()
            prob._call("comm_shards", C.byref(nr), C.byref(rk), C.byref(tot), cnt)
            assert (nr.value, rk.value) == (self.world, self.rank), "RCCL communicator disagrees with torch.distributed"
            self.counts = [int(c) for c in cnt]   # what RCCL itself saw
        else:
            self.counts = [B]
            if self.world > 1:
                lst = [None] * self.world
                dist.all_gather_object(lst, B)
                self.counts = [int(c) for c in lst]
            self.xs = torch.empty((B, N, n), dtype=torch.float64)
            self.us = torch.empty((B, N - 1, m), dtype=torch.float64)
        self.total = sum(self.counts)
        self.xg = torch.empty((self.total, N, n), dtype=torch.float64, device=dev)
        self.ug = torch.empty((self.total, N - 1, m), dtype=torch.float64, device=dev)

    def _cpu_gather(self, out, mine):
        """torch.distributed gather of per-rank blocks of possibly different length (padded to the longest)."""
        torch = self.torch
        live = self.dist is not None and self.dist.is_initialized()
        if not live:
            out.copy_(mine)
            return
        # the collective needs device tensors under the nccl (= RCCL) backend, host tensors under gloo
        dev = torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend() == "nccl" else torch.device("cpu")
        mx = max(self.counts)
        pad = torch.zeros((mx,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=dev)
        pad[: mine.shape[0]] = mine.to(dev)
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(parts, pad)
        off = 0
        for r, c in enumerate(self.counts):
            out[off:off + c] = parts[r][:c].to(out.device)
            off += c

    def __call__(self):
        torch, prob = self.torch, self.prob
        if self.on_device:
            prob._call("allgather", C.c_void_p(self.xg.data_ptr()), C.c_void_p(self.ug.data_ptr()))
            return self.xg, self.ug
        from . import api
        self.xs.copy_(torch.from_numpy(api.states(prob)))
        self.us.copy_(torch.from_numpy(api.controls(prob)))
        self._cpu_gather(self.xg, self.xs)
        self._cpu_gather(self.ug, self.us)
        return self.xg, self.ug

    def stats(self, solver=None):
        """The small gather of SURVEY.md §8e: (iterations[B_total], status[B_total], cost[B_total]) in global order.  On
        the GPU: ``to_allgather_stats`` (RCCL, same communicator); on CPU ``solver.stats`` through torch.distributed."""
        if self.on_device:
            its, st, J = np.zeros(self.total, np.int32), np.zeros(self.total, np.int32), np.zeros(self.total)
            pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
            self.prob._call("allgather_stats", pi(its), pi(st), J.ctypes.data_as(C.POINTER(C.c_double)))
            return its, st, J
        if solver is None:
            from ._capi import ArgumentError
            raise ArgumentError("TrajectoryGather.stats(solver): the CPU / gloo path gathers solver.stats — pass the solver "
                                "(the device path gathers from the handle and recomputes the objective cost of the current trajectories)")
        torch, out = self.torch, []
        for key in ("iterations", "status", "cost"):
            a = torch.from_numpy(np.ascontiguousarray(solver.stats[key]))
            g = torch.empty((self.total,), dtype=a.dtype)
            self._cpu_gather(g, a)
            out.append(g.numpy())
        return tuple(out)

    def close(self):
        if self.on_device:
            self.prob._call("comm_destroy")


def gather_stats(dist, arr):
    """all_gather of a per-trajectory numpy stats array (equal shards) -> rank-major numpy array."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return arr.copy()
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty((dist.get_world_size() * t.numel(),), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.reshape(-1))
    return out.cpu().numpy()
