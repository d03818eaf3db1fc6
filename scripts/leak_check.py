import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, kdehip
from kdehip import _lib
rng = np.random.default_rng(0)
free0 = torch.cuda.mem_get_info()[0]
for it in range(1500):
    D = int(rng.integers(1, 5)); M = int(rng.integers(2, 4)); N = int(rng.integers(20, 3000)); Np = int(rng.integers(8, 600))
    trees = [kdehip.kde(rng.standard_normal((D, N)), [0.3]) for _ in range(M)]
    kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=2, Np=Np, seed=it)
    if it % 500 == 499:
        print(it, "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6)
_lib.lib.kdehip_clear_cache()
print("after clear_cache: free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6)
