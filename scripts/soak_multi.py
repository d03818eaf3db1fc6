#!/usr/bin/env python3
"""Soak of the multi-device entry points on ONE GPU (KDEHIP_ALIAS_DEVICES=1: logical devices wrap around the visible
ones): random products through kdehip_prod_philox / kdehip_gibbs1_multi with 2..8 logical devices must reproduce the
one-device result bit for bit (contiguous chain ranges, global Philox index), in both precisions.
    python scripts/soak_multi.py [cases] [--resident]
--resident also drives resident multi-device plans (kdehip_product_multi_*; needs torch for the device arrays)."""
import os
import sys
import time

os.environ["KDEHIP_ALIAS_DEVICES"] = "1"   # (read once, when the library first needs it)
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402
from oracle import oracle  # noqa: E402

import faulthandler  # noqa: E402

resident = "--resident" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cases = int(argv[0]) if argv else 100
faulthandler.dump_traceback_later(int(os.environ.get("KDEHIP_SOAK_WATCHDOG", "240")), exit=True)  # a hang names its call
rng = np.random.default_rng(31337)
t0 = time.time()
bad = 0
for c in range(cases):
    D = int(rng.integers(1, 7))
    M = int(rng.integers(2, 8))
    Ns = [int(rng.choice([20, 100, 300, 1000, 2500])) for _ in range(M)]
    Np = int(rng.choice([1, 7, 64, 257, 1000]))
    Niter = int(rng.integers(0, 4))
    prec = int(rng.choice([64, 32]))
    trees = [kdehip.kde(rng.standard_normal((D, n)) + rng.uniform(-1, 1, size=(D, 1)), rng.uniform(0.1, 0.6, size=D)) for n in Ns]
    ref = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=c, precision=prec)
    if c == 0:   # the device has been acquired and a product has run: from here on a stall is the library's
        print("SOAK first GPU call done", file=sys.stderr, flush=True)
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    last = f"case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} fp{prec}"
    ref_s = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    for g in (2, 3, 8):
        got = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=c, precision=prec, ngpus=g)
        got_s = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN, ngpus=g)
        ok = all(np.array_equal(a, b) for a, b in zip(ref, got)) and all(np.array_equal(a, b) for a, b in zip(ref_s, got_s))
        if not ok:
            bad += 1
            dp = [int((a != b).sum()) for a, b in zip(ref, got)]
            ds = [int((a != b).sum()) for a, b in zip(ref_s, got_s)]
            where = np.unique(np.nonzero(ref[1] != got[1])[1]) if ref[1].shape == got[1].shape else []
            ref2 = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=c, precision=prec)
            got2 = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=c, precision=prec, ngpus=g)
            refs2 = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
            gots2 = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN, ngpus=g)
            eq = lambda x, y: all(np.array_equal(a, b) for a, b in zip(x, y))
            print(f"MISMATCH case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} fp{prec} ngpus={g}: philox diffs {dp} "
                  f"streams diffs {ds} chains {where[:10]} | repeat: ref==ref2 {eq(ref, ref2)} got==got2 {eq(got, got2)} "
                  f"ref2==got2 {eq(ref2, got2)} | streams: ref==ref2 {eq(ref_s, refs2)} got==got2 {eq(got_s, gots2)} ref2==got2 {eq(refs2, gots2)}")
    # resident plans on several (aliased) devices: after the peer-write all-gather every device holds the whole result
    if resident and c % 4 == 0 and Np > 0:
        import torch
        dev = torch.device("cuda", 0)
        for g in (2, 5):
            with kdehip.MultiProductPlan(trees, precision=prec, ngpus=g) as mp:
                P = [torch.zeros(D * Np, dtype=torch.float64, device=dev) for _ in range(mp.ngpus)]
                I = [torch.zeros(M * Np, dtype=torch.int64, device=dev) for _ in range(mp.ngpus)]
                mp.sample_philox_device(Np, Niter, c, 0, True, P, I)
                torch.cuda.synchronize()
                for k in range(mp.ngpus):
                    gp = P[k].cpu().numpy().reshape(Np, D).T
                    gi = I[k].cpu().numpy().reshape(Np, M).T
                    if not (np.array_equal(gp, ref[0]) and np.array_equal(gi, ref[1])):
                        bad += 1
                        print(f"MISMATCH (resident multi) case {c}: D={D} M={M} Ns={Ns} Np={Np} Niter={Niter} fp{prec} ngpus={g} copy {k}")
print(f"{cases} cases x 3 device counts x 2 random sources: {bad} mismatches, {time.time()-t0:.0f} s")
sys.exit(1 if bad else 0)
