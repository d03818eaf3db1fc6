"""fp32 path (BASELINE config 5's precision) on the GPU: the SURVEY.md 8(d) distributional gates at the configuration's
per-GPU batch, and the reference's underflow rule `pT < 1e-99 -> uniform draw` (src/MSGibbs01.jl:299-315), which an
fp32 sum cannot express directly (exp underflows at 2^-126) and which the kernel reproduces with raised exponents
(csrc/gibbs_kernel.hip, Num<float>::tiny_total)."""
import numpy as np
import pytest

import kdehip
from oracle import oracle

pytestmark = pytest.mark.gpu


def _ks(a, b):
    """two-sample Kolmogorov-Smirnov statistic sup |F_a - F_b| of equal-size samples"""
    both = np.concatenate([a, b])
    order = np.argsort(both, kind="stable")
    steps = np.where(order < a.size, 1.0, -1.0)
    return float(np.abs(np.cumsum(steps)).max() / a.size)


def _gates(a, b, Nout):
    """KS < 1.36/sqrt(Nout/2) per dimension, mean and variance within 5/sqrt(Nout)*sigma."""
    sd = a.std(axis=1)
    for d in range(a.shape[0]):
        assert _ks(a[d], b[d]) < 1.36 / np.sqrt(Nout / 2.0), (d, _ks(a[d], b[d]))
    assert np.all(np.abs(a.mean(axis=1) - b.mean(axis=1)) < 5.0 / np.sqrt(Nout) * sd)
    assert np.all(np.abs(a.var(axis=1) - b.var(axis=1)) < 5.0 / np.sqrt(Nout) * sd ** 2)


def test_config5_per_gpu_batch_gates_against_fp64():
    """BASELINE config 5 as one GPU of eight runs it: 6-D, 4 densities x 10000 points (bench.py's synthetic
    mixture), 8192 chains, Niter = 20, fp32 -- against the fp64 plan on the same Philox stream."""
    import bench
    D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c5"]
    assert (D, M, N, Nout, Niter, prec) == (6, 4, 10000, 8192, 20, 32)
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    seed = 20260101
    with kdehip.ProductPlan(trees, precision=32) as p32, kdehip.ProductPlan(trees, precision=64) as p64:
        assert p32.fast_math_path and p32.bytes_per_eval == 52 and p32.evals_per_sample(Niter) == 2216088
        b, ib = p32.sample(Nout, Niter=Niter, seed=seed)
        a, ia = p64.sample(Nout, Niter=Niter, seed=seed)
        again = p32.sample(Nout, Niter=Niter, seed=seed)
        f32, f64 = p32.fallback_count(), p64.fallback_count()
    assert np.array_equal(again[0], b) and np.array_equal(again[1], ib)      # deterministic
    assert np.isfinite(b).all() and ib.min() >= 2 and ib.max() <= N + 1
    _gates(a, b, Nout)
    # same uniforms: fp32 rounding flips a label only where u falls within ~1e-6 of a CDF step, but a flipped
    # coarse-level label redirects the rest of that chain, so agreement is high, not total
    assert (ia != ib).mean() < 0.15
    assert f64 == 0 and f32 == 0       # overlapping mixtures: no draw underflows in either precision


def test_headline_shape_fp32_gates():
    """config 3's shape in fp32 (6-D, 4 x 1000, 2048 chains, Niter = 10): same gates."""
    import bench
    D, M, N, Nout, Niter, _, cid = bench.CONFIGS["c3"]
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    with kdehip.ProductPlan(trees, precision=32) as p32, kdehip.ProductPlan(trees, precision=64) as p64:
        b, ib = p32.sample(Nout, Niter=Niter, seed=3)
        a, ia = p64.sample(Nout, Niter=Niter, seed=3)
    _gates(a, b, Nout)
    assert (ia != ib).mean() < 0.05


@pytest.mark.parametrize("width", [0, 8, 16])
def test_fp32_products_of_more_than_four_densities(width):
    """fp32 with 5..8 densities has no lean instantiation (build time): the general kernel runs it at every width --
    same gates against the fp64 run (which takes the lean kernel at 8 and 16 chains per workgroup)."""
    import bench
    rng = np.random.default_rng(17)
    D, M, N, Nout, Niter = 3, 6, 400, 1024, 4
    pts, bws = bench.synth_inputs(kdehip, D, M, N, 7)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    with kdehip.ProductPlan(trees, precision=32) as p32, kdehip.ProductPlan(trees, precision=64) as p64:
        p32.set_variant(width)
        p64.set_variant(width)
        b, ib = p32.sample(Nout, Niter=Niter, seed=11)
        a, ia = p64.sample(Nout, Niter=Niter, seed=11)
    _gates(a, b, Nout)
    assert (ia != ib).mean() < 0.05


def _two_clusters(sep, seed=5):
    rng = np.random.default_rng(seed)
    pa = rng.standard_normal((2, 200)) * 0.05
    pb = rng.standard_normal((2, 200)) * 0.05 + np.array([[sep], [0.0]])
    Np, Niter = 64, 3
    K, R, nU, nN = oracle.rng_sizes(2, 2, Np, Niter, [200, 200])
    return pa, pb, Np, Niter, rng.random(nU), rng.standard_normal(nN)


@pytest.mark.parametrize("sep,expect", [(2.5, "none"), (3.0, "none"), (3.3, "some"), (3.35, "some"), (4.0, "most"),
                                        (500.0, "most")])
def test_fp32_takes_the_uniform_fallback_exactly_where_fp64_does(sep, expect):
    """Two clusters `sep` apart (kernel sigma 0.1): from sep ~ 2 on the best exponent is below fp32's exp range
    (e^-100), from sep ~ 3.3 on the fp64 sum drops below 1e-99 for some draws and the reference switches to a
    uniform draw.  The fp32 plan must switch for the same draws as the fp64 plan (counted by the kernels), which
    in turn must switch exactly where the oracle does."""
    pa, pb, Np, Niter, randU, randN = _two_clusters(sep)
    oa, ob = oracle.OracleDensity(pa, [0.1]), oracle.OracleDensity(pb, [0.1])
    oracle.fallback_count(reset=True)
    o_pts, o_ind = oracle.gibbs1([oa, ob], Np, Niter, randU, randN)
    n_oracle = oracle.fallback_count(reset=True)
    ga, gb = kdehip.kde(pa, [0.1]), kdehip.kde(pb, [0.1])
    import torch
    dev = torch.device("cuda", 0)
    dU, dN = torch.from_numpy(randU).to(dev), torch.from_numpy(randN).to(dev)
    res = {}
    for prec in (64, 32):
        with kdehip.ProductPlan([ga, gb], precision=prec) as plan:
            for use_tables in (True, False):
                plan.set_variant(0 if use_tables else 4)
                before = plan.fallback_count()
                d_pts = torch.zeros(2 * Np, dtype=torch.float64, device=dev)
                d_ind = torch.zeros(2 * Np, dtype=torch.int64, device=dev)
                plan.sample_streams_device(Np, Niter, dU, randU.size, dN, randN.size, True, d_pts, d_ind)
                torch.cuda.synchronize()
                res[prec, use_tables] = (d_pts.cpu().numpy().reshape(Np, 2).T, d_ind.cpu().numpy().reshape(Np, 2).T,
                                         plan.fallback_count() - before)
    total = Np * 2 * 8 * (Niter + 1)
    for use_tables in (True, False):
        p64, i64, n64 = res[64, use_tables]
        p32, i32, n32 = res[32, use_tables]
        assert np.array_equal(i64, o_ind) and np.allclose(p64, o_pts, rtol=1e-11, atol=1e-11)
        assert n64 == n_oracle
        if expect == "none":
            assert n64 == 0 and n32 == 0
        else:
            assert n64 > 0 and abs(n32 - n64) <= max(2, 0.02 * n64), (n32, n64, total)
            assert (expect == "some") == (n64 < 0.7 * total)
        if n64 == 0:   # no uniform draws: fp32 follows the fp64 chains up to the usual rounding flips
            assert (i32 != i64).mean() < 0.1


def test_config5_total_chain_count_on_one_gpu():
    """BASELINE config 5 as stated -- fp32, 6-D, 4 densities x 10000 points, Nout = 65536, Niter = 20 -- all chains on
    one GPU (what `bench.py --strong --config c5 --gpus 1` times): the same chains drawn as eight shards of 8192 (the
    8-GPU split of the config: global Philox index) are bit-identical to the one call, and an 8192-chain slice from
    the MIDDLE of the batch passes the SURVEY.md 8(d) gates against fp64 on the same stream."""
    import bench
    D, M, N, _, Niter, prec, cid = bench.CONFIGS["c5"]
    Np = bench.TOTAL_NOUT["c5"]
    assert (Np, Niter, prec) == (65536, 20, 32)
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    seed = 20260101
    with kdehip.ProductPlan(trees, precision=32) as p32:
        assert p32.kernel_name(Np) == "gibbs_lean_kernel"
        b, ib = p32.sample(Np, Niter=Niter, seed=seed)
        shards = [p32.sample(Np // 8, Niter=Niter, seed=seed, sample_offset=g * (Np // 8)) for g in range(8)]
    assert np.isfinite(b).all() and ib.min() >= 2 and ib.max() <= N + 1
    assert np.array_equal(np.concatenate([s[0] for s in shards], axis=1), b)
    assert np.array_equal(np.concatenate([s[1] for s in shards], axis=1), ib)
    lo = 3 * (Np // 8)
    with kdehip.ProductPlan(trees, precision=64) as p64:
        a, ia = p64.sample(Np // 8, Niter=Niter, seed=seed, sample_offset=lo)
    _gates(a, b[:, lo:lo + Np // 8], Np // 8)
    assert (ia != ib[:, lo:lo + Np // 8]).mean() < 0.15


def test_fp32_row_pair_tiles_give_one_result_through_every_kernel():
    """fp32 tiles keep the two rows of a pair adjacent (kdehip_internal.hpp TileAddr): the same fp32 product through
    the register-resident sampler at 4 / 8 / 16 chains per workgroup (also with every tile read from global memory, and without
    the conditional tables), the general kernel at 4 / 8 / 16, a host-packed plan
    and GPU-packed resident densities must be bit-identical -- odd and even rows per lane, resident / streamed / chunked
    tiles, masks (scripts/soak_fp32_layout.py, 40 random shapes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_fp32_layout.py"), "40"], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 mismatches" in out.stdout
