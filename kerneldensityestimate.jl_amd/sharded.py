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
        """Per-rank send buffer [points | indices] and the gathered receive buffer, both viewed as 8-byte
        words so that ONE all-gather moves the product samples and their labels together."""
        key = int(Np)
        if key not in self._bufs:
            D, M, G = self.plan.ndims, self.plan.Ndens, self.world
            chunk = max(shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0] for r in range(G))
            send = torch.zeros(chunk * (D + M), dtype=torch.float64, device=self.device)
            recv = torch.zeros(G * chunk * (D + M), dtype=torch.float64, device=self.device)
            self._bufs[key] = dict(chunk=chunk, send=send, recv=recv,
                                   pts=send[: chunk * D], ind=send[chunk * D:].view(torch.int64))
        return self._bufs[key]

    def gather(self, b):
        """The one collective of the path (RCCL all-gather over xGMI when the backend is "nccl")."""
        dist.all_gather_into_tensor(b["recv"], b["send"], group=self.group)

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
        self.gather(b)
        c = b["chunk"]
        w = c * (D + M)
        pts, ind = [], []
        for r in range(G):
            n = shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0]
            pts.append(b["recv"][r * w: r * w + n * D])
            ind.append(b["recv"][r * w + c * D: r * w + c * D + n * M].view(torch.int64))
        return torch.cat(pts).view(Np, D).t(), torch.cat(ind).view(Np, M).t()
