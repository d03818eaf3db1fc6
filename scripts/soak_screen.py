#!/usr/bin/env python3
"""Soak of the fp32 screen (resident, streamed and chunked screen tiles; csrc/screen_device.hpp): random products inside
its domain -- 2, 3, 4 or 8 densities, 1..8 dimensions, 600..8192 points, ragged, weighted or not, shared or per-point
bandwidth scales, data far from the origin or wide -- sampled with the screen (plan variant 0) and without (variant 5) at 8
and 16 chains per workgroup: labels and points must be identical bit for bit, and the screen must have run.
    python scripts/soak_screen.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(777)
t0 = time.time()
bad = screened = steps = repeats = 0
modes = {}
for c in range(cases):
    D = int(rng.integers(1, 9))
    M = int(rng.choice([2, 3, 4, 8]))
    top = int(rng.choice([1200, 2500, 5000, 8192]))
    Ns = [int(rng.integers(600, top + 1)) for _ in range(M)]
    shift = float(rng.choice([0.0, 0.0, 50.0, 1.0e5]))
    scale = float(rng.choice([1.0, 1.0, 0.05, 20.0]))
    g = []
    for n in Ns:
        pts = (rng.standard_normal((D, n)) * rng.uniform(0.5, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))) * scale + shift
        ks = rng.uniform(0.05, 0.5, size=D) * scale
        w = rng.uniform(0.1, 1.0, size=n) if rng.random() < 0.3 else None
        g.append(kdehip.kde(pts, ks, w))
    width = int(rng.choice([8, 16]))
    Np, Niter = int(rng.choice([width, 3 * width + 5, 10 * width])), int(rng.integers(1, 4))
    with kdehip.ProductPlan(g) as plan:
        res = {}
        for v in (0, 5):
            plan.set_variant(v + (8000 if width == 8 else 6000))
            res[v] = plan.sample(Np, Niter=Niter, seed=c, want_labels=True)
            if v == 0:
                st = plan.screen_stats()
    ok = all(np.array_equal(a, b) for a, b in zip(res[0], res[5]))
    screened += st["steps"] > 0
    steps += st["steps"]
    repeats += st["repeats"]
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} width={width} shift={shift} scale={scale} {st}")
print(f"{cases} cases ({screened} with screened levels; {steps} screened draws, {repeats} repeated in fp64 = "
      f"{100.0 * repeats / max(steps, 1):.2f} %): {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
