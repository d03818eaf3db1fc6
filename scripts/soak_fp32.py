#!/usr/bin/env python3
"""Soak of the fp32 path against the fp64 path on random products: SURVEY 8(d) gates per dimension (KS, mean,
variance), label agreement, and equal uniform-fallback counts (the reference's pT < 1e-99 rule, reproduced in fp32 by
the raised-exponent repeats).  python scripts/soak_fp32.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402


def ks(a, b):
    both = np.concatenate([a, b])
    order = np.argsort(both, kind="stable")
    return float(np.abs(np.cumsum(np.where(order < a.size, 1.0, -1.0))).max() / a.size)


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(99)
t0 = time.time()
viol, worst_ks, fb_diff, with_fb, worst_lab = 0, 0.0, 0, 0, 0.0
for c in range(cases):
    D = int(rng.integers(1, 7))
    M = int(rng.integers(2, 7))
    Ns = [int(rng.choice([50, 200, 700, 1500, 4000])) for _ in range(M)]
    Nout, Niter = 1024, int(rng.integers(1, 6))
    sep = rng.choice([0.0, 0.0, 0.5, 1.5, 3.0, 8.0, 20.0])  # some products of well separated densities (underflow territory)
    trees = []
    for j, n in enumerate(Ns):
        pts = rng.standard_normal((D, n)) * rng.uniform(0.3, 1.5, size=(D, 1)) + sep * j * rng.standard_normal((D, 1))
        trees.append(kdehip.kde(pts, rng.uniform(0.05, 0.6, size=D)))
    with kdehip.ProductPlan(trees, precision=32) as p32, kdehip.ProductPlan(trees, precision=64) as p64:
        v = int(rng.choice([0, 8, 16]))
        p32.set_variant(v)
        p64.set_variant(v)
        b, ib = p32.sample(Nout, Niter=Niter, seed=c)
        a, ia = p64.sample(Nout, Niter=Niter, seed=c)
        f32, f64 = p32.fallback_count(), p64.fallback_count()
    sd = a.std(axis=1) + 1e-300
    k = max(ks(a[d], b[d]) for d in range(D))
    worst_ks = max(worst_ks, k * np.sqrt(Nout / 2.0))
    bad = k >= 1.36 / np.sqrt(Nout / 2.0) or np.any(np.abs(a.mean(axis=1) - b.mean(axis=1)) >= 5.0 / np.sqrt(Nout) * sd) \
        or np.any(np.abs(a.var(axis=1) - b.var(axis=1)) >= 5.0 / np.sqrt(Nout) * sd ** 2)
    fbad = abs(f32 - f64) > 0.02 * max(f64, 50)
    with_fb += f64 > 0
    worst_lab = max(worst_lab, float((ia != ib).mean()))
    if bad or fbad:
        viol += bad
        fb_diff += fbad
        print(f"case {c}: D={D} M={M} Ns={Ns} Niter={Niter} sep={sep} KS*sqrt(n/2)={k*np.sqrt(Nout/2):.2f} "
              f"labels differ {(ia != ib).mean():.3f} fallbacks fp32 {f32} fp64 {f64}")
print(f"{cases} cases ({with_fb} with uniform fallbacks in fp64): {viol} gate violations, {fb_diff} fallback-count differences, "
      f"worst KS*sqrt(n/2) {worst_ks:.2f}, worst label disagreement {worst_lab:.3f}, {time.time()-t0:.0f} s")
