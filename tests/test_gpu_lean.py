"""The register-resident kernel for products of 2..4 (fp64: 2..8) densities (csrc/gibbs_lean.hip) against the general kernel
(csrc/gibbs_kernel.hip, forced with plan variants 30+): same labels, bit-identical points, every staging mode,
both precisions, with and without conditional tables; plus oracle parity through the lean path."""
import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


def _trees(seed, D, Ns, weighted=False):
    rng = np.random.default_rng(seed)
    g, o = [], []
    for N in Ns:
        pts = synth_mixture(rng, D, N)
        ks = silverman_bw(pts) if N > 1 else np.full(D, 0.5)
        ks = np.where(ks > 0, ks, 0.5)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        g.append(kdehip.kde(pts, ks, w))
        o.append(oracle.OracleDensity(pts, ks, w))
    return g, o


@pytest.mark.parametrize("D,Ns,Np,Niter,weighted", [
    (1, [100, 100], 100, 5, False),            # BASELINE config 1
    (2, [200, 200, 200], 256, 5, False),       # config 2
    (6, [1000] * 4, 300, 4, False),            # config 3 shape: resident + streamed levels, tables
    (3, [37, 128, 129, 300], 70, 2, True),     # ragged sizes
    (2, [5000, 4000], 40, 2, False),           # frontiers beyond 4096 nodes: recursive narrowing, streamed tiles
    (6, [10000] * 4, 24, 2, False),            # config 5 shape: chunked staging
    (8, [130, 90], 50, 1, True),
    (4, [1, 60, 7], 33, 3, False),             # a single-point density
])
@pytest.mark.parametrize("prec", [64, 32])
def test_lean_kernel_equals_general_kernel(D, Ns, Np, Niter, weighted, prec):
    g, o = _trees(100 * D + len(Ns), D, Ns, weighted)
    seed = 77
    with kdehip.ProductPlan(g, precision=prec) as plan:
        assert plan.fast_math_path
        res = {}
        for variant in (0, 4, 1, 30, 34, 31):   # lean: default / no tables / global; general: the same three
            plan.set_variant(variant)
            res[variant] = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for variant in (4, 1, 30, 34, 31):
        for a, b in zip(res[0], res[variant]):
            assert np.array_equal(a, b), variant
    if prec == 64 and max(Ns) <= 5000:
        u, n = kdehip.philox_streams(seed, 0, Np, K, R)
        op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
        assert np.array_equal(res[0][1], oi) and np.array_equal(res[0][2], ol)
        assert np.allclose(res[0][0], op, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("D,Ns,Np,Niter,weighted", [
    (3, [5000] * 8, 48, 2, False),                     # BASELINE config 4's shape: chunked staging, 8 densities
    (2, [120, 80, 200, 64, 33], 64, 3, True),          # 5 ragged, weighted densities
    (6, [300] * 6, 50, 2, False),
    (1, [50, 64, 65, 7, 100, 128, 90], 80, 3, False),  # 7 densities
])
def test_lean_kernel_5_to_8_densities(D, Ns, Np, Niter, weighted):
    """fp64 products of 5..8 densities take the lean kernel at 8 and 16 chains per workgroup (variants 8, 16) and the
    general kernel otherwise (default width for these chain counts; variants 38, 46 force it at the same widths):
    identical labels and points, equal to the oracle's."""
    g, o = _trees(900 + D + len(Ns), D, Ns, weighted)
    seed = 78
    with kdehip.ProductPlan(g) as plan:
        assert plan.fast_math_path
        res = {}
        for variant in (8, 16, 0, 38, 46):
            plan.set_variant(variant)
            res[variant] = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for variant in (16, 0, 38, 46):
        for a, b in zip(res[8], res[variant]):
            assert np.array_equal(a, b), variant
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
    assert np.array_equal(res[8][1], oi) and np.array_equal(res[8][2], ol)
    assert np.allclose(res[8][0], op, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("width", [2, 8, 16])
def test_lean_kernel_widths_and_caller_streams(width):
    """every workgroup width, caller-supplied randU/randN (the reference's consumption order) and addEntropy=false"""
    D, Ns, Np, Niter = 3, [300, 200, 257], 100, 3
    g, o = _trees(5, D, Ns)
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    rng = np.random.default_rng(1)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    import torch
    dev = torch.device("cuda", 0)
    dU, dN = torch.from_numpy(randU).to(dev), torch.from_numpy(randN).to(dev)
    for addEntropy in (True, False):
        op, oi = oracle.gibbs1(o, Np, Niter, randU, randN, addEntropy=addEntropy)
        with kdehip.ProductPlan(g) as plan:
            for variant in (width, 30 + width):
                plan.set_variant(variant)
                d_pts = torch.zeros(D * Np, dtype=torch.float64, device=dev)
                d_ind = torch.zeros(len(Ns) * Np, dtype=torch.int64, device=dev)
                plan.sample_streams_device(Np, Niter, dU, nU, dN, nN, addEntropy, d_pts, d_ind)
                torch.cuda.synchronize()
                gp = d_pts.cpu().numpy().reshape(Np, D).T
                gi = d_ind.cpu().numpy().reshape(Np, len(Ns)).T
                assert np.array_equal(gi, oi), (variant, addEntropy)
                assert np.allclose(gp, op, rtol=1e-12, atol=1e-12)


def test_which_kernel_runs_is_pinned():
    """The register-resident kernel exists for products of 2..4 densities (both precisions) and for fp64 products of 8
    (BASELINE config 4) at 8 and 16 chains per workgroup; everything else -- 5..7 densities, fp32 products of more than
    4, masked products, one-density "products" -- runs the general kernel (their instantiations were dropped in round 3:
    a third of the build).  Pinned here so that a change of the dispatch shows up as a change of this test."""
    rng = np.random.default_rng(0)
    def plan_of(M, prec=64, mask=None, D=2):
        trees = [kdehip.kde(rng.standard_normal((D, 40)), [0.3]) for _ in range(M)]
        return kdehip.ProductPlan(trees, precision=prec, partialDimMask=mask)
    for M in (2, 3, 4):
        for prec in (64, 32):
            with plan_of(M, prec) as p:
                assert p.kernel_name(2048) == "gibbs_lean_kernel" and p.kernel_name(100) == "gibbs_lean_kernel"
    for M in (5, 6, 7, 9):
        with plan_of(M) as p:
            assert p.kernel_name(2048) == "gibbs_product_kernel"
    with plan_of(8) as p:
        assert p.kernel_name(2048) == "gibbs_lean_kernel" and p.kernel_name(16384) == "gibbs_lean_kernel"
        assert p.kernel_name(512) == "gibbs_product_kernel"     # 4 chains per workgroup: not instantiated for 8 densities
        p.set_variant(38)
        assert p.kernel_name(2048) == "gibbs_product_kernel"    # plan variants 30+ force the general kernel
    for M in (5, 8):
        with plan_of(M, prec=32) as p:
            assert p.kernel_name(2048) == "gibbs_product_kernel"
    with plan_of(1) as p:
        assert p.kernel_name(64) == "gibbs_product_kernel"
    with plan_of(3, mask=[[True, False], [True, True], [False, True]]) as p:
        assert p.kernel_name(64) == "gibbs_product_kernel"
