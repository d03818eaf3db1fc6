"""Chains sharded over the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI) used only for the single all-gather of the product samples at the end.

Output sample s depends only on the read-only densities and on its own RNG slice, so
rank g owns the contiguous range [floor(g*Np/G), floor((g+1)*Np/G)); the Philox counters use the
GLOBAL sample index, hence the gathered result is bit-identical for any number of ranks.
PyTorch is plumbing here (device buffers, stream, process group) -- the sampling itself is the HIP
kernel behind `plan.sample_philox_device`.

Back-to-back products are pipelined: `sample_async` issues the all-gather of product t as an asynchronous
collective (RCCL runs it on its own stream once the kernel of product t is done) and returns at once, so
the kernel of product t+1 -- written into the other buffer slot -- overlaps it; `PendingProduct.result()`
makes the consumer's stream wait for the gather.
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

    def _buffers(self, Np, slot=0):
        """Per-rank send buffer [points | indices] and the gathered receive buffer, both viewed as 8-byte
        words so that ONE all-gather moves the product samples and their labels together.  Two slots
        (double buffer) serve the pipelined `sample_async`."""
        key = (int(Np), int(slot))
        if key not in self._bufs:
            D, M, G = self.plan.ndims, self.plan.Ndens, self.world
            chunk = max(shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0] for r in range(G))
            send = torch.zeros(chunk * (D + M), dtype=torch.float64, device=self.device)
            recv = torch.zeros(G * chunk * (D + M), dtype=torch.float64, device=self.device)
            self._bufs[key] = dict(chunk=chunk, send=send, recv=recv, pending=None,
                                   pts=send[: chunk * D], ind=send[chunk * D:].view(torch.int64))
        return self._bufs[key]

    def gather(self, b, async_op=False):
        """The one collective of the path (RCCL all-gather over xGMI when the backend is "nccl")."""
        return dist.all_gather_into_tensor(b["recv"], b["send"], group=self.group, async_op=async_op)

    def _run_shard(self, b, Np, Niter, seed, addEntropy, sample_base):
        lo, hi = shard_range(Np, self.rank, self.world)
        stream = None
        if self.device.type == "cuda":
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if hi > lo:
            self.plan.sample_philox_device(hi - lo, Niter, seed, sample_base + lo, addEntropy, b["pts"], b["ind"],
                                           None, stream)

    def _assemble(self, b, Np):
        D, M, G = self.plan.ndims, self.plan.Ndens, self.world
        if G == 1:
            return b["pts"][: Np * D].view(Np, D).t(), b["ind"][: Np * M].view(Np, M).t()
        c = b["chunk"]
        w = c * (D + M)
        pts, ind = [], []
        for r in range(G):
            n = shard_range(Np, r, G)[1] - shard_range(Np, r, G)[0]
            pts.append(b["recv"][r * w: r * w + n * D])
            ind.append(b["recv"][r * w + c * D: r * w + c * D + n * M].view(torch.int64))
        return torch.cat(pts).view(Np, D).t(), torch.cat(ind).view(Np, M).t()

    def sample(self, Np, Niter=3, seed=0, addEntropy=True, sample_base=0):
        """All ranks return the same (points[D, Np], indices[M, Np]) device tensors."""
        return self.sample_async(Np, Niter, seed, addEntropy, sample_base, slot=0).result()

    def sample_async(self, Np, Niter=3, seed=0, addEntropy=True, sample_base=0, slot=0):
        """Starts one product in buffer slot `slot` (0 or 1) and returns a PendingProduct without waiting for
        the all-gather.  A slot is reused only after its previous gather has been waited for (done here)."""
        b = self._buffers(Np, slot)
        if b["pending"] is not None:   # the slot's previous gather still reads `send` / writes `recv`
            b["pending"].wait()
            b["pending"] = None
        self._run_shard(b, Np, Niter, seed, addEntropy, sample_base)
        if self.world > 1:
            b["pending"] = self.gather(b, async_op=True)
        return PendingProduct(self, b, Np)


class PendingProduct:
    """Handle of a product whose all-gather may still be in flight."""

    def __init__(self, owner, bufs, Np):
        self._owner, self._b, self._Np = owner, bufs, Np

    def wait(self):
        """Orders the current stream (CPU backends: the host) after the all-gather."""
        w = self._b["pending"]
        if w is not None:
            w.wait()
            self._b["pending"] = None

    def result(self):
        self.wait()
        return self._owner._assemble(self._b, self._Np)
