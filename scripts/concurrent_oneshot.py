#!/usr/bin/env python3
"""Throughput of concurrent one-shot products from several host threads (the pattern of a multi-threaded belief-propagation
host: many small products at once): calls per second with 1, 4, 16 threads.
    python scripts/concurrent_oneshot.py [calls per thread]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(3)
shapes = [(3, 3, 150, 150, 3), (2, 2, 100, 100, 3), (6, 4, 200, 200, 5)]
for D, M, N, Np, Niter in shapes:
    trees = [kdehip.kde(rng.standard_normal((D, N)) + rng.uniform(-1, 1, size=(D, 1)), rng.uniform(0.2, 0.5, size=D)) for _ in range(M)]
    ref = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=1)

    def worker(t):
        bad = 0
        for _ in range(ncalls):
            got = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=1)
            bad += not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]))
        return bad
    line = f"{D}-D, {M} x {N} points, {Np} chains, Niter {Niter}:"
    for T in (1, 4, 16):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=T) as ex:
            bad = sum(ex.map(worker, range(T)))
        dt = time.perf_counter() - t0
        line += f"  {T:2d} threads {T * ncalls / dt:8.0f} products/s ({dt / (T * ncalls) * 1e3 * T:6.3f} ms per call{' WRONG' if bad else ''})"
    print(line)
