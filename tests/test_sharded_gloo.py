"""world_size-2 (and 3) `gloo` tests of the multi-GPU path's host logic on CPU: contiguous sample
ranges, global-sample-index RNG offsets and the single all-gather.  The HIP kernel is replaced by a
stand-in whose output is a pure function of the GLOBAL sample index -- exactly the property the
device Philox stream gives the real kernel -- so the gathered result must equal the 1-rank result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kdehip.sharded import ShardedProduct, shard_range


class FakePlan:
    ndims, Ndens = 3, 2

    def sample_philox_device(self, Np, Niter, seed, sample_offset, addEntropy, d_points, d_indices, d_labels, stream):
        s = torch.arange(sample_offset, sample_offset + Np, dtype=torch.float64)
        pts = (s[:, None] * 10.0 + torch.arange(self.ndims, dtype=torch.float64)[None, :] + seed * 1e-3)
        ind = (s[:, None].to(torch.int64) * 7 + torch.arange(self.Ndens)[None, :] + Niter)
        d_points[: Np * self.ndims] = pts.reshape(-1)
        d_indices[: Np * self.Ndens] = ind.reshape(-1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Np, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = ShardedProduct(FakePlan(), "cpu")
        pts, ind = sp.sample(Np, Niter=4, seed=9, sample_base=100)
        np.save(os.path.join(out_dir, f"pts{rank}.npy"), pts.numpy())
        np.save(os.path.join(out_dir, f"ind{rank}.npy"), ind.numpy())
        # pipelined products: four in flight over two buffer slots, results collected afterwards in order
        pend, got = [], []
        for t in range(4):
            if t >= 2:
                got.append([x.clone() for x in pend[t - 2].result()])
            pend.append(sp.sample_async(Np, Niter=4, seed=9, sample_base=100 + t * Np, slot=t & 1))
        got += [[x.clone() for x in p.result()] for p in pend[2:]]
        for t, (p, i) in enumerate(got):
            np.save(os.path.join(out_dir, f"pipe_pts{rank}_{t}.npy"), p.numpy())
            np.save(os.path.join(out_dir, f"pipe_ind{rank}_{t}.npy"), i.numpy())
    finally:
        dist.destroy_process_group()


def test_shard_ranges_partition_the_samples():
    for Np in (1, 7, 2048, 16385):
        for G in (1, 2, 3, 8):
            r = [shard_range(Np, g, G) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == Np
            assert all(r[i][1] == r[i + 1][0] for i in range(G - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.parametrize("world,Np", [(2, 64), (2, 33), (3, 10)])
def test_all_gather_equals_single_rank(tmp_path, world, Np):
    ref = ShardedProduct(FakePlan(), "cpu")
    pts1, ind1 = [x.clone() for x in ref.sample(Np, Niter=4, seed=9, sample_base=100)]  # (results are views of the slot's buffer)
    pipe = [[x.clone() for x in ref.sample(Np, Niter=4, seed=9, sample_base=100 + t * Np)] for t in range(4)]
    mp.spawn(_worker, args=(world, _free_port(), Np, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"pts{r}.npy"), pts1.numpy())
        assert np.array_equal(np.load(tmp_path / f"ind{r}.npy"), ind1.numpy())
        for t, (p, i) in enumerate(pipe):
            assert np.array_equal(np.load(tmp_path / f"pipe_pts{r}_{t}.npy"), p.numpy())
            assert np.array_equal(np.load(tmp_path / f"pipe_ind{r}_{t}.npy"), i.numpy())
