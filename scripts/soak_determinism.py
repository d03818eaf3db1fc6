#!/usr/bin/env python3
"""Run-to-run determinism soak on shapes with chunked tiles: the same resident plan sampled repeatedly with the same
seed must return identical arrays (any difference is a race).   python scripts/soak_determinism.py [cases] [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(2718)
t0 = time.time()
bad = 0
for c in range(cases):
    D = int(rng.choice([5, 6]))
    M = int(rng.integers(2, 8))
    Ns = [int(rng.choice([20, 100, 300, 1000, 2500])) for _ in range(M - 1)] + [2500]
    Np = int(rng.choice([7, 64, 257, 1000, 2048]))
    Niter = int(rng.integers(0, 4))
    prec = int(rng.choice([64, 32]))
    variant = int(rng.choice([0, 0, 2, 8, 16, 30]))
    trees = [kdehip.kde(rng.standard_normal((D, n)) + rng.uniform(-1, 1, size=(D, 1)), rng.uniform(0.1, 0.6, size=D)) for n in Ns]
    with kdehip.ProductPlan(trees, precision=prec) as plan:
        plan.set_variant(variant)
        ref = plan.sample(Np, Niter=Niter, seed=c, want_labels=True)
        for r in range(reps):
            got = plan.sample(Np, Niter=Niter, seed=c, want_labels=True)
            if not all(np.array_equal(a, b) for a, b in zip(ref, got)):
                bad += 1
                ch = np.unique(np.nonzero(ref[1] != got[1])[1])
                print(f"NONDETERMINISTIC case {c} rep {r}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} fp{prec} variant={variant}: "
                      f"{int((ref[1] != got[1]).sum())} labels differ, chains {ch[:12]}")
print(f"{cases} cases x {reps} repeats: {bad} nondeterministic runs, {time.time()-t0:.0f} s")
sys.exit(1 if bad else 0)
