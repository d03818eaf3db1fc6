#!/usr/bin/env python3
"""Soak of the chunked-tile paths (frontiers beyond half the LDS pool: rows streamed through LDS, segment second pass
from global memory) against the oracle: random dimension, density count 2..8, sizes 1500..9000, every workgroup width.
    python scripts/soak_chunked.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402
from oracle import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(4242)
t0 = time.time()
bad = 0
for c in range(cases):
    D = int(rng.integers(1, 7))
    M = int(rng.integers(2, 9))
    Ns = [int(rng.integers(1500, 9000)) for _ in range(M)]
    Np, Niter = int(rng.choice([8, 17, 40])), int(rng.integers(1, 3))
    variant = int(rng.choice([0, 2, 8, 16]))
    g, o = [], []
    for n in Ns:
        pts = rng.standard_normal((D, n)) * rng.uniform(0.5, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))
        ks = rng.uniform(0.05, 0.5, size=D)
        w = rng.uniform(0.1, 1.0, size=n) if rng.random() < 0.3 else None
        g.append(kdehip.kde(pts, ks, w))
        o.append(oracle.OracleDensity(pts, ks, w))
    with kdehip.ProductPlan(g) as plan:
        plan.set_variant(variant)
        gp, gi, gl = plan.sample(Np, Niter=Niter, seed=c, want_labels=True)
        u, n = kdehip.philox_streams(c, 0, Np, plan.randu_per_sample(Niter), plan.randn_per_sample())
        modes = plan.stage_modes() if hasattr(plan, "stage_modes") else None
    op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
    ok = np.array_equal(gi, oi) and np.array_equal(gl, ol) and np.allclose(gp, op, rtol=1e-11, atol=1e-11)
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} variant={variant} "
              f"labels differ {int((gi != oi).sum())} max|dx| {np.abs(gp - op).max():.3g}")
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
