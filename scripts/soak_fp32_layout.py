#!/usr/bin/env python3
"""fp32 consistency soak for the row-pair tile layout (kdehip_internal.hpp TileAddr): the same fp32 product through every
kernel and width -- register-resident sampler at 4 / 8 / 16 chains per workgroup, general kernel at 4 / 8 / 16, host-packed
plan and GPU-packed resident densities -- must give identical labels and points (one arithmetic, one association), on
shapes with odd and even rows per lane, resident / streamed / chunked tiles, shared and per-node bandwidths, masks.
    python scripts/soak_fp32_layout.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import kdehip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(777)
t0 = time.time()
bad = 0
dev = torch.device("cuda", 0)
for c in range(cases):
    D = int(rng.integers(1, 9))
    M = int(rng.integers(2, 7))
    big = rng.random() < 0.4
    Ns = [int(rng.integers(1500, 9000)) if big else int(rng.integers(1, 700)) for _ in range(M)]
    Np, Niter = int(rng.choice([5, 16, 33])), int(rng.integers(1, 4))
    trees = []
    for n in Ns:
        pts = rng.standard_normal((D, n)) * rng.uniform(0.5, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))
        ks = rng.uniform(0.05, 0.5, size=D)   # (internal levels: per-node bandwidths; leaf level: the shared-bandwidth tile)
        w = rng.uniform(0.1, 1.0, size=n) if rng.random() < 0.3 else None
        trees.append(kdehip.kde(pts, ks, w))
    mask = None
    if rng.random() < 0.15 and D >= 2:
        mask = [[bool(rng.random() < 0.7) for _ in range(D)] for _ in range(M)]
        for d in range(D):
            if not any(m[d] for m in mask):
                mask[int(rng.integers(0, M))][d] = True
    ref = None
    for variant in (0, 1, 2, 4, 8, 16, 32, 38, 46):
        # (variant 1 = every tile read from global memory.  Beyond 64 rows per lane it also swaps the chunked tiles' segment
        # form of the second pass for the general narrowing: the same sums in another association -- equal to the last
        # bit of fp64 in practice, two label sets in 3000 apart in fp32 -- so it is compared up to 4096 points only)
        if variant == 1 and max(Ns) > 4096:
            continue
        with kdehip.ProductPlan(trees, precision=32, partialDimMask=mask) as plan:
            plan.set_variant(variant)
            gp, gi = plan.sample(Np, Niter=Niter, seed=c)
        if ref is None:
            ref = (gp, gi)
        elif not (np.array_equal(gi, ref[1]) and np.array_equal(gp, ref[0])):
            bad += 1
            print(f"MISMATCH case {c} variant {variant}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} mask={mask is not None} "
                  f"labels differ {int((gi != ref[1]).sum())} max|dx| {np.abs(gp - ref[0]).max():.3g}")
    # GPU-packed resident densities
    dd = [kdehip.DeviceDensity(t, device=0) for t in trees]
    P = torch.empty(D * Np, dtype=torch.float64, device=dev)
    I = torch.empty(M * Np, dtype=torch.int64, device=dev)
    kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=Niter, seed=c, precision=32, partialDimMask=mask)
    torch.cuda.synchronize()
    gp = P.cpu().numpy().reshape(Np, D).T
    gi = I.cpu().numpy().reshape(Np, M).T
    if not (np.array_equal(gi, ref[1]) and np.array_equal(gp, ref[0])):
        bad += 1
        print(f"MISMATCH case {c} GPU-packed: D={D} M={M} Ns={Ns} labels differ {int((gi != ref[1]).sum())}")
print(f"{cases} cases x 9 kernels/widths/staging variants + GPU packer: {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
