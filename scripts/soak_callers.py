#!/usr/bin/env python3
"""Soak of the callers either side of the product on random inputs: the GPU ball-tree builder against the host builder
(every array bit for bit, incl. heavy ties), the on-device LOOCV bandwidth against the oracle (1e-9, equal evaluation
counts) and direct evaluation against the oracle (1e-12).   python scripts/soak_callers.py [cases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402
from oracle import oracle  # noqa: E402

FIELDS_BT = ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation")
FIELDS_BD = ("means", "bandwidth", "bandwidthMin", "bandwidthMax")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(777)
t0 = time.time()
bad = {"tree": 0, "loocv": 0, "eval": 0}
for c in range(cases):
    D = int(rng.integers(1, 9))
    N = int(rng.choice([2, 3, 5, 17, 64, 65, 100, 333, 1000, 2048, 3000]))
    kind = rng.choice(["normal", "ties", "mixture"])
    if kind == "ties":
        pts = rng.integers(0, 5, size=(D, N)).astype(float)
    elif kind == "mixture":
        pts = rng.uniform(-2, 2, size=(3, D))[rng.integers(0, 3, N)].T + 0.4 * rng.standard_normal((D, N))
    else:
        pts = rng.standard_normal((D, N)) * rng.uniform(0.1, 10.0, size=(D, 1))
    ks = rng.uniform(0.05, 0.8, size=D)
    w = rng.uniform(0.1, 1.0, size=N) if rng.random() < 0.4 else None
    h = kdehip.kde(pts, ks, w)
    if kdehip._clib.kdehip_make_density_device_supported(D, N):
        g = kdehip.kde(pts, ks, w, device=0)
        same = all(np.array_equal(getattr(g.bt, f), getattr(h.bt, f)) for f in FIELDS_BT) and \
            all(np.array_equal(getattr(g, f), getattr(h, f)) for f in FIELDS_BD)
        if not same:
            bad["tree"] += 1
            print(f"TREE mismatch case {c}: D={D} N={N} kind={kind}")
    if kind != "ties" and D <= 6 and N >= 2:
        gb, gn = kdehip.auto_bandwidth(pts, return_evals=True)
        ob, on = oracle.auto_bandwidth(pts)
        if not (np.allclose(gb, ob, rtol=1e-9, atol=0) and int(np.sum(gn)) == int(on)):
            bad["loocv"] += 1
            print(f"LOOCV mismatch case {c}: D={D} N={N} {gb} vs {ob}, evals {gn} vs {on}")
    o = oracle.OracleDensity(pts, ks, w)
    pos = rng.standard_normal((D, int(rng.integers(1, 400)))) * 2.0
    if not (np.allclose(h(pos), oracle.eval_direct(o, pos), rtol=1e-12, atol=1e-300) and
            np.allclose(kdehip.evaluateDualTree(h, lvFlag=True), oracle.eval_direct(o, loo=True), rtol=1e-12, atol=1e-300)):
        bad["eval"] += 1
        print(f"EVAL mismatch case {c}: D={D} N={N}")
print(f"{cases} cases: mismatches {bad}, {time.time()-t0:.0f} s")
sys.exit(1 if any(bad.values()) else 0)
