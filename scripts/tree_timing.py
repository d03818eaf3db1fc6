"""Host vs GPU ball-tree builder at a few shapes (run under rocprofv3 --kernel-trace --stats for the kernel part)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip
rng = np.random.default_rng(0)
def T(f, n=20):
    f(); f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3
for D, N in ((6, 2048), (6, 1000), (3, 2000)):
    p = rng.standard_normal((D, N)); k = np.full(D, 0.3)
    print(f"{D}x{N}: host {T(lambda: kdehip.kde(p, k)):.3f} ms | device {T(lambda: kdehip.kde(p, k, device=0)):.3f} ms | "
          f"device batch of 4 {T(lambda: kdehip.kde_batch([(p, k)] * 4, device=0)):.3f} ms | host x4 {T(lambda: [kdehip.kde(p, k) for _ in range(4)]):.3f} ms")
