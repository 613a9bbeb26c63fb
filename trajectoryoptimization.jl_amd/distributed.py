"""Multi-GPU plumbing: the batch shards across ranks as independent units — no collective
inside the iLQR/AL loops — and the converged trajectories are all-gathered once per solve (RCCL over xGMI
when the process group is NCCL/RCCL; gloo in the CPU tests).  torch.distributed is plumbing only."""
from __future__ import annotations

import ctypes as C

import numpy as np


def shard_offset(rank, batch_per_rank):
    """Global index of the first trajectory owned by ``rank`` (contiguous block partition)."""
    return int(rank) * int(batch_per_rank)


class TrajectoryGather:
    """Pre-allocated buffers + one all_gather per array.  Rank-major output: X[world*B, N, n], U[world*B, N-1, m]."""

    def __init__(self, prob, dist, device=None):
        import torch
        self.torch, self.dist, self.prob = torch, dist, prob
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        n, m, N = prob.dims()
        B = prob.B
        self.on_device = device is not None
        dev = device if self.on_device else "cpu"
        self.xs = torch.empty((B, N, n), dtype=torch.float64, device=dev)
        self.us = torch.empty((B, N - 1, m), dtype=torch.float64, device=dev)
        self.xg = torch.empty((self.world * B, N, n), dtype=torch.float64, device=dev)
        self.ug = torch.empty((self.world * B, N - 1, m), dtype=torch.float64, device=dev)

    def __call__(self):
        torch, prob = self.torch, self.prob
        if self.on_device:  # device-to-device into torch memory, then RCCL
            prob._call("get_states_device", C.c_void_p(self.xs.data_ptr()))
            prob._call("get_controls_device", C.c_void_p(self.us.data_ptr()))
        else:
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
