"""Where a one-shot small product spends its time: Python mirror vs raw C call vs GPU kernels (run under
rocprofv3 --kernel-trace --stats for the kernel part)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip
from kdehip import _lib
from kdehip._lib import f64p, i64p, ptr
from oracle import oracle
rng = np.random.default_rng(0)
D, M, N, Np, Niter = 2, 2, 100, 100, 5
pts = [rng.standard_normal((D, N)) for _ in range(M)]
trees = [kdehip.kde(p, [0.3]) for p in pts]
K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
randU, randN = rng.random(nU), rng.standard_normal(nN)
def t(f, n=200):
    f(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e6
py = t(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN))
arr = (_lib.CDensity * M)(*[x._cstruct() for x in trees])
P = np.zeros(D * Np); I = np.zeros(M * Np, dtype=np.int64)
raw = t(lambda: _lib.lib.kdehip_gibbs1(M, arr, Np, Niter, ptr(P, f64p), ptr(I, i64p), ptr(randU, f64p), randU.size,
                                       ptr(randN, f64p), randN.size, 1, D, None, 0))
plan = kdehip.ProductPlan(trees)
res = t(lambda: plan.sample(Np, Niter=Niter, seed=1))
print(f"python mirror {py:.0f} us | raw kdehip_gibbs1 {raw:.0f} us | resident plan + philox + D2H {res:.0f} us")
