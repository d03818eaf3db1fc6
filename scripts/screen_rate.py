#!/usr/bin/env python3
"""Go / no-go measurement for fp32 SCREENING with fp64 certification of the deep levels.

CPU only (numpy).  Runs the multiscale Gibbs sampler of BASELINE config 3 (or c4 with --config c4) in fp64 -- the
reference's arithmetic, reference src/MSGibbs01.jl:250-351 -- and, on every label draw of the levels >= --from-level,
repeats the kernel evaluations in IEEE fp32 the way the packed-fp32 evaluators of csrc/gibbs_device.hpp form them
(subtract, square, multiply-add; product/rsqrt form for per-node bandwidths), sums them in the tile order (lane = z / B,
row = z % B; lane sums, then the scan over lanes) and asks

  (a) how far the fp32 cumulative sums are from the fp64 ones (the measured relative error, against the bound delta of
      DESIGN.md "fp32 screening");
  (b) how often the certification `b[z-1] * kappa < u * total` and `u * total * kappa <= b[z]` (kappa = 1 + 3 delta) fails,
      i.e. how often the step would be repeated in fp64 (per step, and per 8-chain workgroup-step: the unit that would
      matter if the chains of a workgroup had to wait for each other);
  (c) that a certified fp32 decision is ALWAYS the fp64 decision (0 wrong certified decisions is the soundness check).

  python scripts/screen_rate.py [--config c3] [--chains 64] [--from-level 9] [--delta-scale 1.0]
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kdehip  # noqa: E402

f32 = np.float32
U32 = 2.0 ** -24


def frontiers(bd):
    """levelDown! (src/MSGibbs01.jl:500-523): the frontier of every level, node ids 1-based."""
    N = bd.bt.num_points
    L = int(math.floor(math.log(N) / math.log(2) + 1))
    left, right = bd.bt.left_child, bd.bt.right_child
    cur = [1]
    out = [cur]
    for _ in range(L):
        nxt = []
        for z in cur:
            lc, rc = int(left[z - 1]), int(right[z - 1])
            if 0 < lc <= 2 * N:
                nxt.append(lc)
            if 0 < rc <= 2 * N:
                nxt.append(rc)
        cur = nxt
        out.append(cur)
    return out


class Dens:
    def __init__(self, bd):
        D, N = bd.bt.dims, bd.bt.num_points
        self.D, self.N = D, N
        self.means = np.asarray(bd.means).reshape(2 * N, D)
        self.bw = np.asarray(bd.bandwidth).reshape(2 * N, D)
        self.w = np.asarray(bd.bt.weights)
        self.fr = [np.array(f, dtype=np.int64) - 1 for f in frontiers(bd)]
        self.L = len(self.fr) - 1


def tile_order_cumsum(p, B):
    """Cumulative sums over the frontier in the order the kernel forms them: entry z sits in lane z // B, row z % B; the
    boundaries b[z] = (sum of the lanes before) + (sum of the rows 0..r of its lane).  Sequential sums in p's dtype."""
    n = p.size
    pad = (-n) % B
    q = np.concatenate([p, np.zeros(pad, dtype=p.dtype)]).reshape(-1, B)
    inl = np.cumsum(q, axis=1, dtype=p.dtype)
    lane_tot = inl[:, -1]
    lane_excl = np.concatenate([[p.dtype.type(0)], np.cumsum(lane_tot, dtype=p.dtype)[:-1]]).astype(p.dtype)
    b = (lane_excl[:, None] + inl).reshape(-1)[:n]
    return b


def eval64(d, fr, center, cov):
    m, v, w = d.means[fr], d.bw[fr], d.w[fr]
    c = v + cov
    q = ((m - center) ** 2 / c).sum(axis=1)
    return w / np.sqrt(np.prod(c, axis=1)) * np.exp(-0.5 * q)


def eval32(d, fr, center, cov, mu0):
    """The packed-fp32 evaluators on tiles centred at mu0 (fp64 subtraction at pack time, then one rounding)."""
    m = (d.means[fr] - mu0).astype(f32)
    v = d.bw[fr].astype(f32)
    w = d.w[fr].astype(f32)
    cen = (center - mu0).astype(f32)
    c = (v + cov.astype(f32)).astype(f32)
    dl = (m - cen).astype(f32)
    d2 = (dl * dl).astype(f32)
    uniform = bool(np.all(d.bw[fr] == d.bw[fr][0]))
    LOG2E = f32(1.4426950408889634)
    if uniform:
        ninv = (f32(-0.5) * LOG2E / c[0]).astype(f32)
        acc = np.zeros(m.shape[0], dtype=f32)
        for k in range(d.D):
            acc = (d2[:, k] * ninv[k] + acc).astype(f32)
        scale = f32(1.0) / np.sqrt(np.prod(c[0].astype(f32), dtype=f32), dtype=f32)
        front = (w * scale).astype(f32)
        x = acc
    else:
        # fraction tree: num / prod by pairwise addition of fractions (gibbs_device.hpp fraction_sum)
        def frac(lo, hi):
            if hi - lo == 1:
                return d2[:, lo], c[:, lo]
            mid = lo + (hi - lo + 1) // 2
            na, ea = frac(lo, mid)
            nb, eb = frac(mid, hi)
            return (na * eb + (nb * ea).astype(f32)).astype(f32), (ea * eb).astype(f32)
        num, prod = frac(0, d.D)
        r = (f32(1.0) / np.sqrt(prod, dtype=f32)).astype(f32)
        q = ((num * r).astype(f32) * r).astype(f32)
        front = (w * r).astype(f32)
        x = (f32(-0.5) * LOG2E * q).astype(f32)
    return (front * np.exp2(x, dtype=f32)).astype(f32), x


def delta_bound(d, fr, center, mu0, cov):
    """The rigorous relative bound on every fp32 cumulative sum (DESIGN.md "fp32 screening"): exponent error
    |dx| <= 2 sqrt(X / c0) |a| + kx u X with a_d = (max|m'_d| + |center'_d|) / sqrt(min c_d) * u * c0, X = 160 (largest
    exponent magnitude of a term that is not flushed), kx = 24, c0 = 0.7213 (= log2(e) / 2); plus the rounding of front,
    exp2 and the sums."""
    c0 = 0.7213475204444817
    mp = np.abs(d.means[fr] - mu0).max(axis=0) * (1 + U32)
    cmin = (d.bw[fr] + cov).min(axis=0)
    a = (mp + np.abs(center - mu0) * (1 + U32)) / np.sqrt(cmin) * (2 * U32) * c0
    X = 160.0
    dx = 2.0 * math.sqrt(X / c0) * float(np.sqrt((a * a).sum())) + 24 * U32 * X
    B = (fr.size + 63) // 64
    return dx * math.log(2.0) * 1.01 + (B + 16) * U32


def weighted_terms(d, fr, center, mu0, cov, old=False):
    """The per-term bound the kernel accumulates beside the sums (csrc/screen_device.hpp, gibbs_lean.hip step_screen): term i
    carries the relative error A + Bc |x_i| (x_i its base-2 exponent).  old: the constants rounds 5a-5o ran with."""
    c0 = 0.72134752
    D = d.D
    uniform = bool(np.all(d.bw[fr] == d.bw[fr][0]))
    mmax = np.abs((d.means[fr] - mu0).astype(f32)).max(axis=0).astype(np.float64)
    acen = np.abs((center - mu0).astype(f32)).astype(np.float64)
    cf = (d.bw[fr].astype(f32).min(axis=0).astype(np.float64) + cov).astype(f32).astype(np.float64)
    g = np.minimum(mmax + acen, 2.0 * acen)
    na = math.sqrt(float((g * g / cf).sum())) * (U32 * math.sqrt(c0) * 1.01)
    if os.environ.get("SCREEN_RATE_NA_SCALE"):  # (attribution experiments only)
        na *= float(os.environ["SCREEN_RATE_NA_SCALE"])
    B = (fr.size + 63) // 64
    depth = 0 if D <= 1 else 1 if D <= 2 else 2 if D <= 4 else 3
    if old:
        kx, const = 32.0, B + 40.0
    else:
        kx = (12.0 + D) if uniform else (16.0 + 2 * depth + D)
        const = (B + 1) // 2 + 9 + ((D + 9.0) if uniform else float((3 * D + 18) // 2))
    A = math.log(2.0) * 1.01 * na + const * U32
    Bc = math.log(2.0) * 1.01 * (na + kx * U32)
    return A, Bc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weighted", action="store_true", help="per-term error weights (what the kernel does)")
    ap.add_argument("--config", default="c3")
    ap.add_argument("--chains", type=int, default=64)
    ap.add_argument("--from-level", type=int, default=9)
    ap.add_argument("--delta-scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--old-constants", action="store_true", help="the bound's constants before they were tightened (kx = 32, B + 40)")
    args = ap.parse_args()
    D, M, N, _, Niter, _, cid = bench.CONFIGS[args.config]
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    dens = [Dens(kdehip.kde(p, b)) for p, b in zip(pts, bws)]
    L = max(d.L for d in dens)
    K = M * (1 + L * (Niter + 1))
    R = D * (L + 1)
    S = args.chains
    randU, randN = kdehip.philox_streams(args.seed, 0, S, K, R)
    mu0 = [d.means[0].copy() for d in dens]  # the root mean of every density: the centring point of its fp32 tiles

    steps = amb = wrong = fallback32 = 0
    diag = {}
    amb_by_level = {}
    steps_by_level = {}
    max_rel = 0.0
    max_ratio = 0.0
    wg_amb = {}
    max_ratio_w = [0.0]
    for s in range(S):
        ru = randU[s * K:(s + 1) * K]
        rn = randN[s * R:(s + 1) * R]
        c = M  # call counter; call c reads ru[c - 1]
        ind = [0] * M  # node index (0-based) of the selected kernel
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
                    B = (fr.size + 63) // 64
                    b64 = tile_order_cumsum(p64, B)
                    t64 = u * b64[-1]
                    hit = np.nonzero(t64 <= b64)[0]
                    z64 = int(hit[0]) if hit.size else fr.size - 1
                    if l >= args.from_level:
                        p32, x32 = eval32(d, fr, center, cv, mu0[j])
                        b32 = tile_order_cumsum(p32, B).astype(np.float64)
                        delta = delta_bound(d, fr, center, mu0[j], cv) * args.delta_scale
                        kappa = 1.0 + 3.0 * delta
                        tot = b32[-1]
                        ok = (tot >= 2.0 ** -40) and (tot < 2.0 ** 100)
                        steps += 1
                        steps_by_level[l] = steps_by_level.get(l, 0) + 1
                        if ok:
                            rel = np.abs(b32 - b64) / b64
                            sig = b64 > 1e-30 * b64[-1]
                            if sig.any():
                                mr = float(rel[sig].max())
                                max_rel = max(max_rel, mr)
                                max_ratio = max(max_ratio, mr / delta)
                            t32 = u * tot
                            h32 = np.nonzero(t32 <= b32)[0]
                            z32 = int(h32[0]) if h32.size else fr.size - 1
                            below = b32[z32 - 1] if z32 > 0 else 0.0
                            certified = (below * kappa < t32) and (t32 * kappa <= b32[z32])
                            if args.weighted:
                                A, Bc = weighted_terms(d, fr, center, mu0[j], cv, args.old_constants)
                                e = p32.astype(np.float64) * (A + Bc * np.abs(x32.astype(np.float64)))
                                Etot = float(e.sum()) * args.delta_scale
                                if os.environ.get("SCREEN_RATE_DIAG"):
                                    xa = float((p32.astype(np.float64) * np.abs(x32.astype(np.float64))).sum() / tot)
                                    diag.setdefault(l, []).append((Etot / tot / U32, A / U32, Bc / U32, xa))
                                mrg = 2.1 * Etot
                                certified = (below + mrg < t32) and (t32 + mrg <= b32[z32])
                                mr = float(np.abs(b32 - b64).max())
                                max_ratio_w[0] = max(max_ratio_w[0], mr / Etot)
                        else:
                            fallback32 += 1
                            certified = False
                        if not certified:
                            amb += 1
                            amb_by_level[l] = amb_by_level.get(l, 0) + 1
                            wg_amb[(s // 8, l, p, j)] = 1
                        elif z32 != z64:
                            wrong += 1
                    new[j] = int(fr[z64])  # (the first pass never reads `new`: adopting at once equals :376-384)
            ind = new

    wg_steps = (S // 8) * sum(M * (Niter + 1) for l in range(args.from_level, L + 1)) if S >= 8 else 0
    print(f"config {args.config}: {S} chains, levels {args.from_level}..{L}, delta scale {args.delta_scale}")
    print(f"  screened steps {steps}; repeated in fp64 {amb} = {100.0 * amb / max(steps, 1):.3f} % "
          f"(of which fp32 total out of range: {fallback32})")
    for l in sorted(steps_by_level):
        print(f"    level {l}: {amb_by_level.get(l, 0)} / {steps_by_level[l]} = "
              f"{100.0 * amb_by_level.get(l, 0) / steps_by_level[l]:.3f} %")
    for l in sorted(diag):
        a = np.array(diag[l])
        print(f"    level {l}: mean E/total {a[:, 0].mean():.0f} u (median {np.median(a[:, 0]):.0f}), A {a[:, 1].mean():.0f} u, "
              f"Bc {a[:, 2].mean():.0f} u (median {np.median(a[:, 2]):.0f}), value-weighted |x| {a[:, 3].mean():.1f}")
    if wg_steps:
        print(f"  8-chain workgroup-steps with at least one repeat: {len(wg_amb)} / {wg_steps} = "
              f"{100.0 * len(wg_amb) / wg_steps:.3f} %")
    print(f"  largest relative error of an fp32 cumulative sum: {max_rel:.3e}; largest error / bound: {max_ratio:.3f}")
    if args.weighted:
        print(f"  weighted bound: largest |fp32 - fp64| of a cumulative sum / E_total: {max_ratio_w[0]:.3f}")
    print(f"  certified decisions that differ from fp64: {wrong}  (must be 0)")


if __name__ == "__main__":
    main()
