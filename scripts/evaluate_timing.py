"""Throughput of the direct evaluation kernel (kdehip_evaluate): wall time per call for a few shapes.
Run under `rocprofv3 --kernel-trace --stats` for the kernel-only time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
import kdehip

rng = np.random.default_rng(0)
for D, N, Nq in [(6, 10000, 65536), (3, 5000, 16384), (1, 2048, 2048), (6, 1000, 2048)]:
    p = kdehip.kde(rng.standard_normal((D, N)), [0.3])
    pos = rng.standard_normal((D, Nq))
    kdehip.evaluateDualTree(p, pos)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        kdehip.evaluateDualTree(p, pos)
    dt = (time.perf_counter() - t0) / reps
    print(f"D={D} N={N} Nq={Nq}: {dt*1e3:.3f} ms per call, {N*Nq/dt/1e9:.1f} G kernel evaluations/s (host buffers in and out)")
