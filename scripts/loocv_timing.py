"""LOOCV bandwidth of a 6 x 2048 sample (the kde!(pGM) step of `*`): wall time per call; run under
rocprofv3 --kernel-trace --stats for the per-round kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import kdehip
rng = np.random.default_rng(0)
shapes = [(6, 2048), (1, 100), (3, 500), (6, 4096), (3, 16384), (6, 65536)] if len(sys.argv) < 2 else \
    [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for D, N in shapes:
    x = rng.standard_normal((D, N))
    kdehip.auto_bandwidth(x)
    reps = 10 if N <= 4096 else 3
    t = time.perf_counter()
    for _ in range(reps):
        bw, nev = kdehip.auto_bandwidth(x, return_evals=True)
    dt = (time.perf_counter() - t) / reps
    print(f"auto_bandwidth {D}x{N}: {dt*1e3:.3f} ms per call, {nev} evaluations, "
          f"{float(np.sum(nev)) * N * N / dt / 1e9:.0f} G pair evaluations/s")
