#!/usr/bin/env python3
"""Chunked / streamed fp32 screening against the unscreened run (variant 5) and the all-global run (variant 1), bit for bit.
    KDEHIP_LIB=<development library> python scripts/check_screen_chunk.py D M N [N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip
from tests.helpers import silverman_bw, synth_mixture

D, M = int(sys.argv[1]), int(sys.argv[2])
for N in map(int, sys.argv[3:]):
    rng = np.random.default_rng(N + D)
    g = []
    for j in range(M):
        pts = synth_mixture(rng, D, N - 37 * (j % 3))
        g.append(kdehip.kde(pts, silverman_bw(pts), rng.uniform(0.2, 1.0, size=pts.shape[1]) if j == 1 else None))
    with kdehip.ProductPlan(g) as plan:
        res = {}
        for v in (0, 5, 1):
            plan.set_variant(v)
            res[v] = plan.sample(int(os.environ.get("NP", "1100")), Niter=3, seed=17, want_labels=True)
            if v == 0:
                st = plan.screen_stats()
        same = all(np.array_equal(a, b) for v in (5, 1) for a, b in zip(res[0], res[v]))
        print(f"D={D} M={M} N={N}: identical={same} screen={st}", flush=True)
        assert same
