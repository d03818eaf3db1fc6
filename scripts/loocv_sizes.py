#!/usr/bin/env python3
"""kde!(points) bandwidth search at several sizes: time per call and the evaluations it took.
    [KDEHIP_LOOCV_SPEC=0] python scripts/loocv_sizes.py [D]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip
from tests.helpers import synth_mixture
D = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for N in (128, 300, 500, 1000, 1300, 1500, 1800, 2048, 3000):
    pts = synth_mixture(np.random.default_rng(N), D, N)
    kdehip.auto_bandwidth(pts)
    n = 30
    t = time.perf_counter()
    for _ in range(n):
        bw, ne = kdehip.auto_bandwidth(pts, return_evals=True)
    print("D=%d N=%5d: %.3f ms  evals %d  bw[0] %.12g" % (D, N, (time.perf_counter() - t) / n * 1e3, ne, bw[0]), flush=True)
