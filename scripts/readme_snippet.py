import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (the README snippet, runnable from a checkout)
import numpy as np, kdehip
p = kdehip.kde(np.random.randn(3, 500))          # kde!(pts): LOOCV bandwidth on the GPU
q = kdehip.kde(np.random.randn(3, 500) + 1, [0.3])  # kde!(pts, bw)
pGM, labels = kdehip.prodAppxMSGibbsS(p, [p, q], None, None, Niter=5, seed=1)
pq = p * q                                       # product density (Niter=5, kde!(pGM))
vals = pq(np.zeros((3, 1)))                      # evaluate
print(pGM.shape, labels.shape, vals, kdehip.getBW(pq)[:, 0])
# many `*` in ONE call (a belief-propagation sweep): batched sampler, shared bandwidth-search launches, pooled trees
d = [kdehip.DeviceDensity(kdehip.kde(np.random.randn(2, 200), [0.3])) for _ in range(6)]
msgs = kdehip.mul_device_batch([[d[0], d[1], d[2]], [d[3], d[4]], [d[5], d[0]]], seeds=[1, 2, 3])   # 3 resident densities
print([m.num_points for m in msgs], msgs[0].bw)
# a product on R x S1: the reference's addop / diffop / getMu / getLambda tuples as a per-dimension enum
a = kdehip.kde(np.vstack([np.random.randn(300), np.random.uniform(-3.1, 3.1, 300)]), [0.3])
pts, idx = kdehip.prodAppxMSGibbsS(None, [a, a], None, None, Niter=3, Np=100, seed=1, manifold=["euclid", "circular"])
print(pts.shape, float(np.abs(pts[1]).max()) < np.pi)
