import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (the README snippet, runnable from a checkout)
import numpy as np, kdehip
p = kdehip.kde(np.random.randn(3, 500))          # kde!(pts): LOOCV bandwidth on the GPU
q = kdehip.kde(np.random.randn(3, 500) + 1, [0.3])  # kde!(pts, bw)
pGM, labels = kdehip.prodAppxMSGibbsS(p, [p, q], None, None, Niter=5, seed=1)
pq = p * q                                       # product density (Niter=5, kde!(pGM))
vals = pq(np.zeros((3, 1)))                      # evaluate
print(pGM.shape, labels.shape, vals, kdehip.getBW(pq)[:, 0])
