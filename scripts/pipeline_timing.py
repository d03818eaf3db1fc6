"""End-to-end timing of the `*` pipeline pieces (host tree build, plan creation = pack + upload, Gibbs
kernel, LOOCV bandwidth of the product, final tree) at the headline shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
def T(f, n=5):
    f(); t = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t) / n * 1e3, r
t_tree, trees = T(lambda: [kdehip.kde(p, b) for p, b in zip(pts, bws)], 20)
t_plan, plan = T(lambda: kdehip.ProductPlan(trees))
t_samp, (pGM, ind) = T(lambda: plan.sample(Nout, Niter=Niter, seed=1))
t_bw, bw = T(lambda: kdehip.auto_bandwidth(pGM), 20)
t_final, _ = T(lambda: kdehip.kde(pGM, bw), 20)
t_eval, _ = T(lambda: kdehip.evaluateDualTree(trees[0], pGM))
print(f"{sys.argv[1] if len(sys.argv)>1 else 'c3'}: host trees of {M} inputs {t_tree:.2f} ms | plan (pack+upload) {t_plan:.2f} ms | "
      f"sample {Nout} chains incl. alloc+D2H {t_samp:.2f} ms | LOOCV bandwidth of pGM ({D}x{Nout}) {t_bw:.2f} ms | "
      f"final tree {t_final:.2f} ms | evaluate {N}x{Nout} {t_eval:.2f} ms")
# host-buffer drop-in (kdehip_gibbs1): pack + H2D of streams + kernel + D2H, the PCIe-inclusive figure
K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
randU, randN = kdehip.philox_streams(1, 0, Nout, K, R)
t_g1, _ = T(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, randU=randU, randN=randN), 5)
print(f"kdehip_gibbs1 host-to-host ({randU.nbytes/1e6:.1f} MB of randU over PCIe): {t_g1:.2f} ms = {Nout/t_g1*1e3:.0f} samples/s")
t_ph, _ = T(lambda: kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=3), 5)
print(f"prodAppxMSGibbsS host-to-host with device Philox (plan create + run + D2H): {t_ph:.2f} ms = {Nout/t_ph*1e3:.0f} samples/s")
# the product operator itself (src/MSGibbs01.jl:707-726): Np = mean Npts (here N, not Nout), Niter = 5, then kde!(pGM)
big = [kdehip.kde(p, b) for p, b in zip(*bench.synth_inputs(kdehip, D, M, Nout, cid))]   # Nout-point inputs: a Nout-chain product
t_mul, _ = T(lambda: kdehip.mul(big, seed=5), 20)
t_auto, _ = T(lambda: kdehip.kde_auto(pGM), 20)
t_auto_ov, _ = T(lambda: kdehip.kde_auto(pGM, overlap=True), 20)
print(f"`*` of {M} densities x {Nout} points ({Nout} chains, Niter 5, then kde!(pGM)): {t_mul:.2f} ms; kde!(pGM) alone {t_auto:.2f} ms "
      f"(search {t_bw:.2f} + tree {t_final:.2f}; with the tree built on a host thread under the search: {t_auto_ov:.2f} ms)")
