"""Device / host memory over thousands of calls of every host-buffer entry point with random shapes: the library's
allocation cache must stay bounded and hand everything back on kdehip_clear_cache()."""
import sys, os, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, kdehip
from kdehip import _lib
rng = np.random.default_rng(0)
free0 = torch.cuda.mem_get_info()[0]
for it in range(1500):
    D = int(rng.integers(1, 5)); M = int(rng.integers(2, 4)); N = int(rng.integers(20, 3000)); Np = int(rng.integers(8, 600))
    trees = [kdehip.kde(rng.standard_normal((D, N)), [0.3]) for _ in range(M)]
    kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=2, Np=Np, seed=it)
    if it % 5 == 0:   # the callers either side of the product, the multi-device entry on one device, the GPU builder
        q = trees[0](rng.standard_normal((D, int(rng.integers(1, 2000)))))
        kdehip.evaluateDualTree(trees[1], lvFlag=True)
        kdehip.auto_bandwidth(rng.standard_normal((D, int(rng.integers(2, 1500)))))
        kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=1, Np=Np, seed=it, ngpus=1, precision=32)
        kdehip.kde_batch([(rng.standard_normal((D, 300)), np.full(D, 0.3))] * 2, device=0)
    if it % 500 == 499:
        print(it, "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6, "| host max RSS MB",
              resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3)
_lib.lib.kdehip_clear_cache()
print("after clear_cache: free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6)
