"""Host-side phases of the one-shot entry points at the headline shape (config 3): run with KDEHIP_TIMING=1 to get the
library's own phase timers on stderr."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cfg]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
def T(f, n=20):
    f(); f(); t = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t) / n * 1e3, r
with kdehip.ProductPlan(trees) as plan:
    K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    t_res, _ = T(lambda: plan.sample(Nout, Niter=Niter, seed=1))
randU, randN = kdehip.philox_streams(1, 0, Nout, K, R)
t_plan, _ = T(lambda: kdehip.ProductPlan(trees).close())
t_g1, _ = T(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, randU=randU, randN=randN))
t_ph, _ = T(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=3))
print(f"{cfg}: resident plan sample+D2H {t_res:.3f} ms | plan create+destroy {t_plan:.3f} ms | gibbs1 (streams, {randU.nbytes/1e6:.1f} MB randU) "
      f"{t_g1:.3f} ms | prodAppxMSGibbsS (philox) {t_ph:.3f} ms")
