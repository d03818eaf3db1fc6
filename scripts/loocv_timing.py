"""LOOCV bandwidth of a 6 x 2048 sample (the kde!(pGM) step of `*`): wall time per call; run under
rocprofv3 --kernel-trace --stats for the per-round kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import kdehip
rng = np.random.default_rng(0)
x = rng.standard_normal((6, 2048))
kdehip.auto_bandwidth(x)
t = time.perf_counter()
for _ in range(10):
    bw, nev = kdehip.auto_bandwidth(x, return_evals=True)
print(f"auto_bandwidth 6x2048: {(time.perf_counter()-t)/10*1e3:.3f} ms per call, {nev} evaluations")
