"""Chains sharded over the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI) used only for the single all-gather of the product samples at the end.

Output sample s depends only on the read-only densities and on its own RNG slice, so
rank g owns the contiguous range [floor(g*Np/G), floor((g+1)*Np/G)); the Philox counters use the
GLOBAL sample index, hence the gathered result is bit-identical for any number of ranks.
PyTorch is plumbing here (device buffers, stream, process group) -- the sampling itself is the HIP
kernel behind `plan.sample_philox_device`.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(Np: int, rank: int, world: int):
    """Contiguous sample range of `rank`."""
    return (Np * rank) // world, (Np * (rank + 1)) // world


class ShardedProduct:
    def __init__(self, plan, device, group=None):
        self.plan = plan
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._bufs = {}

    def _buffers(self, Np):
        key = int(Np)
        if key not in self._bufs:
            D, M, G = self.plan.ndims, self.plan.Ndens, self.world
            chunk = max(shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0] for r in range(G))
            self._bufs[key] = dict(
                chunk=chunk,
                pts=torch.zeros(chunk * D, dtype=torch.float64, device=self.device),
                ind=torch.zeros(chunk * M, dtype=torch.int64, device=self.device),
                all_pts=torch.zeros(G * chunk * D, dtype=torch.float64, device=self.device),
                all_ind=torch.zeros(G * chunk * M, dtype=torch.int64, device=self.device))
        return self._bufs[key]

    def sample(self, Np, Niter=3, seed=0, addEntropy=True, sample_base=0):
        """All ranks return the same (points[D, Np], indices[M, Np]) device tensors."""
        D, M, G = self.plan.ndims, self.plan.Ndens, self.world
        b = self._buffers(Np)
        lo, hi = shard_range(Np, self.rank, G)
        stream = None
        if self.device.type == "cuda":
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if hi > lo:
            self.plan.sample_philox_device(hi - lo, Niter, seed, sample_base + lo, addEntropy, b["pts"], b["ind"],
                                           None, stream)
        if G == 1:
            return b["pts"][: Np * D].view(Np, D).t(), b["ind"][: Np * M].view(Np, M).t()
        # the one collective of the path: all-gather of pGM (+ labels); equal-sized padded chunks
        dist.all_gather_into_tensor(b["all_pts"], b["pts"], group=self.group)
        dist.all_gather_into_tensor(b["all_ind"], b["ind"], group=self.group)
        c = b["chunk"]
        pts = torch.cat([b["all_pts"][r * c * D: r * c * D + (shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0]) * D]
                         for r in range(G)])
        ind = torch.cat([b["all_ind"][r * c * M: r * c * M + (shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0]) * M]
                         for r in range(G)])
        return pts.view(Np, D).t(), ind.view(Np, M).t()
