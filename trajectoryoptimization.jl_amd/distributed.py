"""Multi-GPU plumbing: the batch shards across ranks as independent units — no collective
inside the iLQR/AL loops — and the converged trajectories are all-gathered once per solve.

On GPUs the gather is the library's own RCCL all-gather over xGMI (``to_comm_init_rank`` / ``to_allgather`` of the
C-ABI: what a Julia host calls as well); ``torch.distributed`` only ships the 128-byte communicator id and provides
the barrier.  The CPU tests (gloo, oracle as the per-rank library) gather host arrays with ``torch.distributed``."""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_offset(rank, batch_per_rank):
    """Global index of the first trajectory owned by ``rank`` (contiguous block partition)."""
    return int(rank) * int(batch_per_rank)


class TrajectoryGather:
    """Pre-allocated buffers + one all-gather per array.  Rank-major output: X[world*B, N, n], U[world*B, N-1, m]
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
        self.xg = torch.empty((self.world * B, N, n), dtype=torch.float64, device=dev)
        self.ug = torch.empty((self.world * B, N - 1, m), dtype=torch.float64, device=dev)
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
        else:
            self.xs = torch.empty((B, N, n), dtype=torch.float64)
            self.us = torch.empty((B, N - 1, m), dtype=torch.float64)

    def __call__(self):
        torch, prob = self.torch, self.prob
        if self.on_device:
            prob._call("allgather", C.c_void_p(self.xg.data_ptr()), C.c_void_p(self.ug.data_ptr()))
            return self.xg, self.ug
        from . import api
        self.xs.copy_(torch.from_numpy(api.states(prob)))
        self.us.copy_(torch.from_numpy(api.controls(prob)))
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.xg, self.xs)
            self.dist.all_gather_into_tensor(self.ug, self.us)
        else:
            self.xg.copy_(self.xs)
            self.ug.copy_(self.us)
        return self.xg, self.ug

    def close(self):
        if self.on_device:
            self.prob._call("comm_destroy")


def gather_stats(dist, arr):
    """all_gather of a per-trajectory numpy stats array (iterations, status, cost) -> rank-major numpy array."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return arr.copy()
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = torch.empty((dist.get_world_size() * t.numel(),), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.reshape(-1))
    return out.cpu().numpy()
