#!/usr/bin/env python3
"""Go / no-go measurement for CERTIFIED SKIPPING of far sub-trees inside the fp32 screen.

The idea (evaluateDualTree's, reference src/DualTree01.jl:248-299, applied to the sampler's deep levels): lay a screen
tile out in sub-tree order -- a GROUP = 64 * b consecutive frontier entries, the b rows a wavefront walks before it moves
on -- with the group's bounding box beside it; a group whose largest possible mass

    sum_i w_i * max_{y in box} prod_d (2 pi c_d)^-1/2 exp(-1/2 (x_d - y_d)^2 / c_d)

is below 2^-30 of the draw's total is skipped by a wave-uniform branch, its bound added to the error sum E (the
certification absorbs it: labels stay the fp64 path's).  Worth building only if MOST groups can be skipped: a skipped
group saves its rows, but the layout needs one wavefront scan per group where the lane-contiguous layout needs one per
tile.

CPU only (numpy).  Runs the sampler of a BASELINE configuration in fp64 (the reference's arithmetic, src/MSGibbs01.jl:250-351)
on the bench inputs and, on every label draw of the levels >= --from-level, asks which groups of 64 * b consecutive
frontier entries could be skipped -- with every choice made in the idea's FAVOUR: the tight box of the group's own
means, the exact total instead of a running lower bound, per-node variances bounded by the group's own extremes.

  python scripts/skip_rate.py --config c4 --chains 8 --from-level 12
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench  # noqa: E402
import kdehip  # noqa: E402
from screen_rate import Dens, eval64  # noqa: E402

THRESH = 2.0 ** -30


def group_bounds(d, fr, b):
    """per group of 64 * b consecutive frontier entries: box of the means, extremes of the variances, weight sum"""
    G = 64 * b
    ng = (fr.size + G - 1) // G
    lo, hi, vmin, vmax, wsum = [], [], [], [], []
    for g in range(ng):
        idx = fr[g * G:(g + 1) * G]
        m, v = d.means[idx], d.bw[idx]
        lo.append(m.min(axis=0)); hi.append(m.max(axis=0))
        vmin.append(v.min(axis=0)); vmax.append(v.max(axis=0))
        wsum.append(d.w[idx].sum())
    return np.array(lo), np.array(hi), np.array(vmin), np.array(vmax), np.array(wsum)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--from-level", type=int, default=12)
    ap.add_argument("--seed", type=int, default=20260101)
    args = ap.parse_args()
    D, M, N, _, Niter, _, cid = bench.CONFIGS[args.config]
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    dens = [Dens(kdehip.kde(p, b)) for p, b in zip(pts, bws)]
    L = max(d.L for d in dens)
    K = M * (1 + L * (Niter + 1))
    R = D * (L + 1)
    S = args.chains
    randU, randN = kdehip.philox_streams(args.seed, 0, S, K, R)
    bs = (1, 2, 4, 8)
    gb = {}  # (density, level, b) -> bounds
    for j, d in enumerate(dens):
        for l in range(args.from_level, L + 1):
            for b in bs:
                gb[(j, l, b)] = group_bounds(d, d.fr[min(l, d.L)], b)
    stat = {}  # (level, b) -> [groups, skippable, entries, entries in skippable groups, individually negligible entries]
    for s in range(S):
        ru = randU[s * K:(s + 1) * K]
        rn = randN[s * R:(s + 1) * R]
        c = M
        ind = [0] * M
        for l in range(1, L + 1):
            lam = np.array([1.0 / d.bw[ind[j]] for j, d in enumerate(dens)])
            lmu = np.array([d.means[ind[j]] / d.bw[ind[j]] for j, d in enumerate(dens)])
            cov = 1.0 / lam.sum(axis=0)
            x = cov * lmu.sum(axis=0) + np.sqrt(cov) * rn[(l - 1) * D:(l - 1) * D + D]
            new = list(ind)
            for p in range(Niter + 1):
                for j, d in enumerate(dens):
                    fr = d.fr[min(l, d.L)]
                    if p == 0:
                        center, cv = x, np.zeros(D)
                    else:
                        lam = np.array([1.0 / dens[k].bw[new[k]] for k in range(M) if k != j])
                        lmu = np.array([dens[k].means[new[k]] / dens[k].bw[new[k]] for k in range(M) if k != j])
                        cv = 1.0 / lam.sum(axis=0)
                        center = cv * lmu.sum(axis=0)
                    u = ru[c - 1]
                    c += 1
                    p64 = eval64(d, fr, center, cv)
                    tot = p64.sum()
                    cs = np.cumsum(p64)
                    hit = np.nonzero(u * tot <= cs)[0]
                    z64 = int(hit[0]) if hit.size else fr.size - 1
                    if l >= args.from_level and tot > 0:
                        for b in bs:
                            lo, hi, vmin, vmax, wsum = gb[(j, l, b)]
                            dist = np.maximum(0.0, np.maximum(lo - center, center - hi))   # per group and dimension
                            ub = wsum / np.sqrt(np.prod(vmin + cv, axis=1)) * np.exp(-0.5 * (dist ** 2 / (vmax + cv)).sum(axis=1))
                            skip = ub < THRESH * tot
                            G = 64 * b
                            ng = skip.size
                            sizes = np.minimum(G, fr.size - np.arange(ng) * G)
                            st = stat.setdefault((l, b), [0, 0, 0, 0, 0])
                            st[0] += ng
                            st[1] += int(skip.sum())
                            st[2] += fr.size
                            st[3] += int(sizes[skip].sum())
                            st[4] += int((p64 < THRESH * tot / fr.size).sum())
                    new[j] = int(fr[z64])
            ind = new
    print(f"config {args.config}: {S} chains, levels {args.from_level}..{L}; a group = 64 * b consecutive frontier entries; "
          f"skippable = its largest possible mass < 2^-30 of the draw's total")
    for (l, b) in sorted(stat):
        g, sk, n, nsk, neg = stat[(l, b)]
        print(f"  level {l:2d}  b = {b}: groups skippable {100.0 * sk / g:5.1f} %  (entries in them {100.0 * nsk / n:5.1f} %; "
              f"entries individually below 2^-30 / n of the total: {100.0 * neg / n:5.1f} %)")


if __name__ == "__main__":
    main()
