"""Latency of ONE small product through the host-buffer entry points (the way a Julia caller uses the
library: every call packs, uploads, runs and copies back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip
from oracle import oracle
rng = np.random.default_rng(0)
for (D, M, N, Np, Niter) in [(3, 3, 150, 150, 3), (2, 2, 100, 100, 5), (6, 4, 200, 200, 5), (6, 4, 1000, 2048, 10)]:
    pts = [rng.standard_normal((D, N)) + rng.uniform(-1, 1, size=(D, 1)) for _ in range(M)]
    trees = [kdehip.kde(p, [0.3]) for p in pts]
    otrees = [oracle.OracleDensity(p, [0.3]) for p in pts]
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    def t(f, n=20):
        f(); a = time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter() - a) / n * 1e3
    g1 = t(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN))
    g2 = t(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=1))
    c = t(lambda: oracle.gibbs1(otrees, Np, Niter, randU, randN), 3)
    print(f"D={D} M={M} N={N} Np={Np} Niter={Niter}: gibbs1(streams) {g1:.3f} ms | philox {g2:.3f} ms | CPU oracle 1 core {c:.1f} ms")
