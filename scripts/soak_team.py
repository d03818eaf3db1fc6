#!/usr/bin/env python3
"""Soak of the wavefront-team launches (plan variants 52 / 54: a chain on 2 / 4 wavefronts) and of the device-packed
products against one wavefront per chain and the host-packed one-shot call: random dimension, density count 2..4 or 8,
sizes that give resident, streamed and chunked deep levels, random chain counts.  Everything must be bit-identical.
    python scripts/soak_team.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(777)
t0 = time.time()
bad = 0
for c in range(cases):
    D = int(rng.integers(1, 9))
    M = int(rng.choice([2, 3, 4, 8]))
    Ns = [int(rng.choice([60, 300, 600, 1100, 2500, 5000, 9000])) for _ in range(M)]
    Np, Niter = int(rng.choice([1, 9, 33, 70, 257])), int(rng.integers(0, 3))
    prec = 64
    trees = []
    for n in Ns:
        pts = rng.standard_normal((D, n)) * rng.uniform(0.5, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))
        ks = rng.uniform(0.05, 0.5, size=D)
        w = rng.uniform(0.1, 1.0, size=n) if rng.random() < 0.3 else None
        trees.append(kdehip.kde(pts, ks, w))
    res = {}
    with kdehip.ProductPlan(trees, precision=prec) as plan:
        for v in (8, 16, 52, 54):
            plan.set_variant(v)
            res[v] = plan.sample(Np, Niter=Niter, seed=c, want_labels=True)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    dev = kdehip.prodAppxMSGibbsS_resident(dd, Np=Np, Niter=Niter, seed=c, precision=prec)
    for d in dd:
        d.close()
    ok = all(np.array_equal(a, b) for v in (16, 52, 54) for a, b in zip(res[8], res[v]))
    ok = ok and np.array_equal(dev[0], res[8][0]) and np.array_equal(dev[1], res[8][1])
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter}")
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
