import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
pGM, _ = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kdehip.auto_bandwidth(pGM)
t = time.perf_counter()
for _ in range(n): bw, ne = kdehip.auto_bandwidth(pGM, return_evals=True)
print("auto_bandwidth %.3f ms, evals %d, bw %s" % ((time.perf_counter() - t) / n * 1e3, ne, bw))
