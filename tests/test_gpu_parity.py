"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on identical random streams.

Bars: labels (integer work) identical; points within 1e-12 relative (fp64; the HIP path orders its
floating-point sums differently from the reference); fp32 judged distributionally.
"""
import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu

PT_TOL = 1e-12


def _pair(pts, ks, w=None):
    return kdehip.kde(pts, ks, w), oracle.OracleDensity(pts, ks, w)


def _make_inputs(seed, D, M, Ns, weighted=False):
    rng = np.random.default_rng(seed)
    gp, op = [], []
    for j in range(M):
        N = Ns[j] if isinstance(Ns, (list, tuple)) else Ns
        pts = synth_mixture(rng, D, N)
        ks = silverman_bw(pts) if N > 1 else np.full(D, 0.5)
        ks = np.where(ks > 0, ks, 0.5)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        a, b = _pair(pts, ks, w)
        gp.append(a)
        op.append(b)
    return gp, op


def _compare(g, o, tol=PT_TOL):
    g_pts, g_ind = g
    o_pts, o_ind = o
    assert g_ind.shape == o_ind.shape and g_pts.shape == o_pts.shape
    mism = int((g_ind != o_ind).sum())
    assert mism == 0, f"{mism} label mismatches of {g_ind.size}"
    scale = np.maximum(1.0, np.abs(o_pts))
    err = float((np.abs(g_pts - o_pts) / scale).max())
    assert err <= tol, f"max relative point error {err}"


CASES = [
    # (D, M, N or [N_j], Np, Niter, weighted)
    (1, 2, 100, 100, 5, False),       # BASELINE config 1
    (2, 3, 200, 256, 5, False),       # BASELINE config 2
    (1, 2, 3, 16, 1, False),
    (1, 2, [1, 5], 32, 2, False),     # single-point density (BallTree01.jl:351-362)
    (2, 2, [64, 65], 64, 3, True),    # frontier exactly / just beyond one wavefront
    (3, 4, [37, 128, 129, 300], 50, 2, True),
    (4, 2, 500, 40, 1, False),
    (5, 3, 90, 33, 2, True),
    (6, 4, 1000, 64, 2, False),       # BASELINE config 3 shape, few samples
    (7, 2, 50, 20, 1, False),
    (8, 2, 130, 20, 1, True),
    (3, 8, 257, 24, 1, False),
    (2, 2, 5000, 8, 1, False),        # frontier > 4096 nodes: recursive narrowing
    (1, 3, 300, 64, 0, False),        # Niter = 0: only the sampleIndices! pass
]


@pytest.mark.parametrize("D,M,N,Np,Niter,weighted", CASES)
def test_streams_parity(D, M, N, Np, Niter, weighted):
    """Caller-supplied randU/randN (the reference's own consumption order) through kdehip_gibbs1."""
    gp, op = _make_inputs(1000 + 17 * D + M, D, M, N, weighted)
    npts = [t.num_points for t in op]
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, npts)
    rng = np.random.default_rng(42)
    randU, randN = rng.random(nU), rng.standard_normal(nN)  # the reference's allocation sizes
    for addEntropy in (True, False):
        g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                    addEntropy=addEntropy)
        o = oracle.gibbs1(op, Np, Niter, randU, randN, addEntropy=addEntropy)
        _compare(g, o)


@pytest.mark.parametrize("D,M,N,Np,Niter", [(2, 3, 200, 256, 5), (6, 4, 1000, 96, 3), (3, 2, 77, 130, 4)])
def test_philox_parity_and_labels_trace(D, M, N, Np, Niter):
    """On-device Philox run == oracle fed with the host twin's arrays; per-level label trace too."""
    gp, op = _make_inputs(7 + D, D, M, N)
    seed = 20260101
    with kdehip.ProductPlan(gp) as plan:
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        assert (K, R) == oracle.rng_sizes(M, D, Np, Niter, [N] * M)[:2]
        g_pts, g_ind, g_lab = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        randU, randN = kdehip.philox_streams(seed, 0, Np, K, R)
        o_pts, o_ind, o_lab = oracle.gibbs1(op, Np, Niter, randU, randN, want_labels=True)
        _compare((g_pts, g_ind), (o_pts, o_ind), tol=1e-11)  # device/host libm differ by ulps in randn
        assert np.array_equal(g_lab, o_lab)
        # bitwise repeatable, and independent of how the samples are split over calls (or GPUs)
        again = plan.sample(Np, Niter=Niter, seed=seed)
        assert np.array_equal(again[0], g_pts) and np.array_equal(again[1], g_ind)
        h = Np // 3
        a = plan.sample(h, Niter=Niter, seed=seed, sample_offset=0)
        b = plan.sample(Np - h, Niter=Niter, seed=seed, sample_offset=h)
        assert np.array_equal(np.concatenate([a[0], b[0]], axis=1), g_pts)
        assert np.array_equal(np.concatenate([a[1], b[1]], axis=1), g_ind)


def test_partial_dim_mask_parity():
    """partialDimMask semantics (reference test/testPartialProd.jl:8-58) incl. poisoned dimensions."""
    rng = np.random.default_rng(3)
    pts1 = rng.random((2, 100)) + 10.0
    pts2 = rng.random((2, 100))
    pts3 = rng.random((2, 100)) - 10.0
    bw1, bw2, bw3 = silverman_bw(pts1), silverman_bw(pts2), silverman_bw(pts3)
    pts1[1, :] = 9999999.0
    pts3[0, :] = 9999999.0
    mask = [[True, False], [True, True], [False, True]]
    gp, op = zip(_pair(pts1, bw1), _pair(pts2, bw2), _pair(pts3, bw3))
    Np, Niter = 100, 3
    K, R, nU, nN = oracle.rng_sizes(3, 2, Np, Niter, [100] * 3)
    rng = np.random.default_rng(11)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, list(gp), None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                partialDimMask=mask)
    o = oracle.gibbs1(list(op), Np, Niter, randU, randN, partialDimMask=mask)
    _compare(g, o)
    pGM = g[0]
    assert 80 < int(((0 < pGM[0]) & (pGM[0] < 10)).sum())    # testPartialProd.jl:51
    assert 80 < int(((-10 < pGM[1]) & (pGM[1] < 0)).sum())   # testPartialProd.jl:53
    # Philox path with a mask
    with kdehip.ProductPlan(list(gp), partialDimMask=mask) as plan:
        assert plan.fast_math_path  # masked products run the masked product/rsqrt form
        Kp, Rp = plan.randu_per_sample(Niter), plan.randn_per_sample()
        gg = plan.sample(Np, Niter=Niter, seed=5)
        u, n = kdehip.philox_streams(5, 0, Np, Kp, Rp)
        _compare(gg, oracle.gibbs1(list(op), Np, Niter, u, n, partialDimMask=mask), tol=1e-11)


def test_generic_arithmetic_path_on_extreme_bandwidths():
    """Variance products outside the comfortable fp64 range select the reference-arithmetic path."""
    rng = np.random.default_rng(9)
    D, N = 6, 120
    gp, op = [], []
    for j in range(2):
        pts = rng.standard_normal((D, N)) * 1e-24
        a, b = _pair(pts, np.full(D, 1e-24))
        gp.append(a)
        op.append(b)
    with kdehip.ProductPlan(gp) as plan:
        assert not plan.fast_math_path
    Np, Niter = 40, 2
    K, R, nU, nN = oracle.rng_sizes(2, D, Np, Niter, [N] * 2)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    o = oracle.gibbs1(op, Np, Niter, randU, randN)
    assert np.array_equal(g[1], o[1])
    assert np.allclose(g[0], o[0], rtol=1e-10, atol=0)


def test_far_apart_densities_take_the_uniform_fallback():
    """pT < 1e-99 -> uniform draw over the frontier (src/MSGibbs01.jl:311-315)."""
    rng = np.random.default_rng(2)
    a, ao = _pair(rng.standard_normal((2, 60)) * 0.01, [0.01])
    b, bo = _pair(rng.standard_normal((2, 60)) * 0.01 + 500.0, [0.01])
    Np, Niter = 64, 2
    K, R, nU, nN = oracle.rng_sizes(2, 2, Np, Niter, [60, 60])
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    o = oracle.gibbs1([ao, bo], Np, Niter, randU, randN)
    _compare(g, o)


def test_short_random_streams_raise_like_bounds_error():
    gp, op = _make_inputs(1, 1, 2, 3)
    K, R, nU, nN = oracle.rng_sizes(2, 1, 2, 1, [3, 3])
    u, n = np.full(nU, 0.5), np.zeros(nN)
    kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=1, Np=2, randU=u[: 2 * K - 1], randN=n[: 2 * R])
    with pytest.raises(IndexError):
        kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=1, Np=2, randU=u[: 2 * K - 2], randN=n)
    with pytest.raises(IndexError):
        kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=1, Np=2, randU=u, randN=n[: 2 * R - 1])


def test_hand_trace_appendix_a():
    """SURVEY.md Appendix A through the HIP path."""
    A = kdehip.kde([0.0, 1.0, 3.0], [0.5])
    B = kdehip.kde([0.2, 2.0, 2.5], [0.4])
    randU = np.array([((37 * i + 11) % 100) / 100.0 + 0.005 for i in range(24)])
    randN = np.array([-1.25, 0.5, -0.5, 1.25, 0.25, -0.75])
    dummy = kdehip.kde(np.zeros(2), [1.0])
    pts, ind = kdehip.prodAppxMSGibbsS(dummy, [A, B], None, None, Niter=1, randU=randU, randN=randN,
                                       addEntropy=False)
    assert ind.tolist() == [[3, 3], [2, 3]]
    assert np.allclose(pts, [[(4 * 1.0 + 6.25 * 0.2) / 10.25, (4 * 1.0 + 6.25 * 2.0) / 10.25]], rtol=0, atol=1e-14)
    pts2, ind2 = kdehip.prodAppxMSGibbsS(dummy, [A, B], None, None, Niter=1, randU=randU, randN=randN)
    assert ind2.tolist() == ind.tolist()
    assert np.allclose(pts2, [[0.3560213600626134, 1.3754954547280667]], rtol=0, atol=1e-13)


def _closed_form_points(trees, ind):
    """addEntropy=false: pGM[d,s] = sum_j mu_j/var_j / sum_j 1/var_j over the selected leaves
    (reference examples/ExtractingLabels.jl:12-37; labels are original index + 1)."""
    D, Np = kdehip.Ndim(trees[0]), ind.shape[1]
    num, den = np.zeros((D, Np)), np.zeros((D, Np))
    for j, t in enumerate(trees):
        pts, bw = kdehip.getPoints(t), kdehip.getBW(t) ** 2
        sel = ind[j] - 2  # value = permutation + 1 with 1-based permutation
        num += pts[:, sel] / bw[:, sel]
        den += 1.0 / bw[:, sel]
    return num / den


def test_full_size_headline_config_properties():
    """BASELINE config 3 at full size (6-D, 4 x 1000 points, Nout = 2048, Niter = 10): properties that
    need no oracle run -- closed-form point invariant, label range, addEntropy-independence of the
    labels, determinism -- plus oracle parity on the first 48 samples."""
    D, M, N, Np, Niter, seed = 6, 4, 1000, 2048, 10, 20260101
    gp, op = _make_inputs(33, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        assert plan.fast_math_path and plan.nlevels == 10
        assert plan.evals_per_sample(Niter) == 88968 and plan.bytes_per_eval == 104
        pe, ie = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=True)
        pn, i_n = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=False)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    assert (K, R) == (444, 66)
    assert np.array_equal(ie, i_n)                      # SURVEY 4, known-answer fact 2
    assert ie.min() >= 2 and ie.max() <= N + 1
    assert np.allclose(pn, _closed_form_points(gp, i_n), rtol=1e-11, atol=1e-12)
    assert np.isfinite(pe).all() and np.abs(pe).max() < 10.0
    ns = 48
    randU, randN = kdehip.philox_streams(seed, 0, ns, K, R)
    o = oracle.gibbs1(op, ns, Niter, randU, randN)
    _compare((pe[:, :ns], ie[:, :ns]), o, tol=1e-11)


def test_fp32_path_is_distributionally_equivalent():
    D, M, N, Np, Niter = 3, 3, 400, 4096, 5
    gp, _ = _make_inputs(77, D, M, N)
    with kdehip.ProductPlan(gp, precision=64) as p64, kdehip.ProductPlan(gp, precision=32) as p32:
        a, ia = p64.sample(Np, Niter=Niter, seed=1)
        b, ib = p32.sample(Np, Niter=Niter, seed=1)
    # same random stream: almost every chain picks the same labels; moments agree to sampling noise
    assert (ia != ib).mean() < 0.02
    sd = a.std(axis=1)
    assert np.all(np.abs(a.mean(axis=1) - b.mean(axis=1)) < 5.0 / np.sqrt(Np) * sd)
    assert np.all(np.abs(a.std(axis=1) - b.std(axis=1)) < 5.0 / np.sqrt(Np) * sd)
    from scipy.stats import ks_2samp
    for d in range(D):
        assert ks_2samp(a[d], b[d]).statistic < 1.36 / np.sqrt(Np / 2)


def _test_prods(rng, D=3, M=6, N=100, n=100, dev=1.0, MCMC=5):
    """testProds (reference test/runtests.jl:167-182) with the Silverman bandwidth in place of the
    LOOCV one (kde!(pts) auto-bandwidth is exercised in test_gpu_bandwidth.py)."""
    P = []
    for _ in range(M):
        pts = dev * rng.standard_normal((D, N))
        P.append(kdehip.kde(pts, silverman_bw(pts)))
    dummy = kdehip.kde(rng.standard_normal((D, n)), [1.0])
    pGM, _ = kdehip.prodAppxMSGibbsS(dummy, P, None, None, Niter=MCMC, seed=int(rng.integers(1 << 60)))
    assert np.abs(pGM).sum() > 1e-14
    prodDev = np.sqrt(dev ** (2 * M) / (M * dev ** 2))
    t1 = np.linalg.norm(pGM.mean(axis=1)) < prodDev
    t2 = all(0.66 * prodDev < pGM[i].std(ddof=1) < 1.33 * prodDev for i in range(D))
    return t1 and t2


@pytest.mark.parametrize("kw", [dict(D=2, M=2), dict(D=2, M=4), dict(D=2, M=6), dict(D=3, M=6, MCMC=10),
                                dict(D=4, M=6, n=200, MCMC=10), dict(D=3, M=5, N=300), dict(D=2, M=7, n=300),
                                dict(D=3, M=2, MCMC=100)])
def test_reference_statistical_acceptance_matrix(kw):
    """rangeUnitTests (reference test/runtests.jl:184-201): 10 repetitions, at least 5 must pass."""
    rng = np.random.default_rng(2026)
    assert sum(bool(_test_prods(rng, **kw)) for _ in range(10)) >= 5


@pytest.mark.parametrize("name", ["c1", "c2", "d6"])
def test_committed_known_answer_fixtures(golden_dir, name):
    """HIP path against the committed vectors tests/golden/gibbs_kat_*.npz (made by
    tests/golden/make_gibbs_kat.py from the oracle)."""
    import os
    from tests.helpers import kat_streams
    z = np.load(os.path.join(golden_dir, f"gibbs_kat_{name}.npz"))
    D, M, N, Np, Niter = (int(z[x]) for x in ("D", "M", "N", "Np", "Niter"))
    trees = [kdehip.kde(z["points"][j], z["bw"][j]) for j in range(M)]
    _, _, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = kat_streams(nU, nN)
    for addEntropy, key in ((True, "pGM_entropy"), (False, "pGM_mean")):
        p, i = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                       addEntropy=addEntropy)
        assert np.array_equal(i, z["indices"])
        assert np.allclose(p, z[key], rtol=1e-12, atol=1e-12)


def test_config4_shape_full_density_size():
    """BASELINE config 4 shape (3-D, 8 densities x 5000 points, Niter = 10; chains reduced to 24 so the
    oracle finishes in seconds): frontiers beyond 4096 nodes, tiles too large for LDS (global mode)."""
    D, M, N, Np, Niter, seed = 3, 8, 5000, 24, 10, 11
    gp, op = _make_inputs(44, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        assert plan.nlevels == 13 and plan.evals_per_sample(Niter) == 1160720
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        assert (K, R) == (1152, 42)
        g = plan.sample(Np, Niter=Niter, seed=seed)
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    _compare(g, oracle.gibbs1(op, Np, Niter, u, n), tol=1e-11)


def test_config5_shape_fp64_parity_and_fp32_agreement():
    """BASELINE config 5 shape (6-D, 4 x 10000 points, Niter = 20).  The fp64 plan must match the oracle
    exactly on a few chains; the fp32 plan (the configuration's precision) must pick the same labels for
    the overwhelming majority of draws and agree in distribution."""
    D, M, N, Niter, seed = 6, 4, 10000, 20, 5
    gp, op = _make_inputs(55, D, M, N)
    with kdehip.ProductPlan(gp, precision=64) as p64, kdehip.ProductPlan(gp, precision=32) as p32:
        assert p64.nlevels == 14 and p64.evals_per_sample(Niter) == 2216088 and p32.bytes_per_eval == 52
        K, R = p64.randu_per_sample(Niter), p64.randn_per_sample()
        assert (K, R) == (1180, 90)
        ns = 8
        g = p64.sample(ns, Niter=Niter, seed=seed)
        u, n = kdehip.philox_streams(seed, 0, ns, K, R)
        _compare(g, oracle.gibbs1(op, ns, Niter, u, n), tol=1e-11)
        Np = 1024
        a, ia = p64.sample(Np, Niter=Niter, seed=seed)
        b, ib = p32.sample(Np, Niter=Niter, seed=seed)
    assert (ia != ib).mean() < 0.15
    sd = a.std(axis=1)
    assert np.all(np.abs(a.mean(axis=1) - b.mean(axis=1)) < 6.0 / np.sqrt(Np) * sd)


@pytest.mark.parametrize("variant", [2, 8, 16, 1])
def test_workgroup_width_variants_give_identical_results(variant):
    """8 or 16 chains per workgroup, and the all-global-memory staging mode, are scheduling choices only."""
    D, M, N, Np, Niter, seed = 6, 4, 1000, 100, 3, 9
    gp, op = _make_inputs(66, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        ref = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        plan.set_variant(variant)
        got = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    _compare(got[:2], oracle.gibbs1(op, Np, Niter, u, n), tol=1e-11)


@pytest.mark.parametrize("D,M,N,weighted,mask", [(2, 2, 300, False, None), (3, 3, [60, 200, 33], True, None),
                                                 (6, 4, 1000, False, None), (1, 5, 40, True, None),
                                                 (2, 3, 100, False, [[1, 0], [1, 1], [0, 1]])])
def test_conditional_tables_are_exact(D, M, N, weighted, mask):
    """Levels whose frontiers fit one wavefront row (power-of-two sizes) are sampled from conditional tables
    built at the first run; the result must be bit-identical to the table-free path (variant 4) and match
    the oracle."""
    Np, Niter, seed = 600, 4, 21
    gp, op = _make_inputs(300 + D + M, D, M, N, weighted)
    for prec in (64, 32):
        with kdehip.ProductPlan(gp, partialDimMask=mask, precision=prec) as plan:
            plan.set_variant(4)
            ref = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
            plan.set_variant(0)
            got = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)   # builds + uses the tables
            small = plan.sample(100, Niter=Niter, seed=seed)                  # tables stay in use once built
            K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
        assert np.array_equal(small[0], ref[0][:, :100]) and np.array_equal(small[1], ref[1][:, :100])
        if prec == 64:
            u, n = kdehip.philox_streams(seed, 0, Np, K, R)
            _compare(got[:2], oracle.gibbs1(op, Np, Niter, u, n, partialDimMask=mask), tol=1e-11)


def test_record_choosen_label_trace_like_the_reference_example():
    """examples/ExtractingLabels.jl: `glbs = makeEmptyGbGlb(); glbs.recordChoosen = true; *( [X1;X2;X3], glbs=glbs,
    addEntropy=false)` then `glbs.labelsChoosen[sample][density][level]`.  Through the drop-in (`gibbs1` with the
    caller's streams) the trace must equal the oracle's at every level, its last level must name the returned
    index, and with addEntropy=false the product points are the precision-weighted means of the traced leaves."""
    X = [kdehip.kde(np.array(v), [1.0]) for v in ([1.0, 2.0, 3.0], [0.5, 1.5, 2.5], [4.0, 5.0, 6.0])]
    O = [oracle.OracleDensity(np.array(v), [1.0]) for v in ([1.0, 2.0, 3.0], [0.5, 1.5, 2.5], [4.0, 5.0, 6.0])]
    Np, Niter = 3, 5
    K, R, nU, nN = oracle.rng_sizes(3, 1, Np, Niter, [3, 3, 3])
    rng = np.random.default_rng(8)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    glbs = kdehip.makeEmptyGbGlb()
    glbs.recordChoosen = True
    pts, ind = kdehip.prodAppxMSGibbsS(None, X, None, None, Niter=Niter, Np=Np, addEntropy=False,
                                       randU=randU, randN=randN, glbs=glbs)
    o_pts, o_ind, o_lab = oracle.gibbs1(O, Np, Niter, randU, randN, addEntropy=False, want_labels=True)
    lc = glbs.labelsChoosen
    L = kdehip.nlevels(3)
    assert sorted(lc) == [1, 2, 3] and sorted(lc[1]) == [1, 2, 3] and sorted(lc[1][1]) == list(range(1, L + 1))
    for s in range(Np):
        mu = 0.0
        for j in range(3):
            assert [lc[s + 1][j + 1][l + 1] for l in range(L)] == list(o_lab[s, j])
            assert lc[s + 1][j + 1][L] + 1 == ind[j, s]          # newIndices = permutation + 1 (:615)
            mu += kdehip.getPoints(X[j])[0, lc[s + 1][j + 1][L] - 1]
        assert abs(pts[0, s] - mu / 3.0) < 1e-12                  # equal bandwidths: plain mean of the leaves
    # the reference records inside sampleIndex only: no sweeps, no entries
    g0 = kdehip.makeEmptyGbGlb(recordChoosen=True)
    kdehip.prodAppxMSGibbsS(None, X, None, None, Niter=0, Np=Np, randU=randU, randN=randN, glbs=g0)
    assert g0.labelsChoosen[1][1] == {}
    # device-Philox front end and `*`
    g1 = kdehip.makeEmptyGbGlb(recordChoosen=True)
    p123 = kdehip.mul(X, glbs=g1, addEntropy=False, seed=4)
    assert kdehip.Npts(p123) == 3 and sorted(g1.labelsChoosen[3][2]) == list(range(1, L + 1))
    got = np.sort(kdehip.getPoints(p123)[0])
    want = np.sort([np.mean([kdehip.getPoints(X[j])[0, g1.labelsChoosen[s][j + 1][L] - 1] for j in range(3)])
                    for s in (1, 2, 3)])
    assert np.allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("D,Ns,Np,Niter,mask", [
    (1, [3, 3], 4, 1, None), (1, [1, 6], 5, 2, None), (2, [9, 14, 5], 6, 2, None), (3, [20, 33], 5, 3, None),
    (2, [12, 12, 12], 8, 2, [[1, 0], [1, 1], [0, 1]]), (2, [10], 4, 1, None), (2, [70, 130], 6, 2, None),
])
def test_gpu_against_the_pure_python_restatement(D, Ns, Np, Niter, mask):
    """The HIP path directly against tests/pymodel.py -- the second, independently written reading of the
    reference (1-based indexing, the reference's own loop structure), without the C oracle in between."""
    from tests import pymodel
    rng = np.random.default_rng(13 * D + len(Ns) + Np)
    raw = [rng.standard_normal((D, n)) for n in Ns]
    kss = [list(rng.uniform(0.2, 0.6, D)) for _ in Ns]
    ws = [list(rng.uniform(0.3, 1.0, n)) for n in Ns]
    gt = [kdehip.kde(p, k, np.array(w)) for p, k, w in zip(raw, kss, ws)]
    mt = [pymodel.kde([list(p[:, i]) for i in range(p.shape[1])], k, w) for p, k, w in zip(raw, kss, ws)]
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    for addEntropy in (True, False):
        gp, gi = kdehip.prodAppxMSGibbsS(None, gt, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                         addEntropy=addEntropy, partialDimMask=mask)
        mp, mi = pymodel.prodAppxMSGibbsS(mt, Np, Niter, list(randU), list(randN), addEntropy, mask)
        assert np.array_equal(gi, np.array(mi, dtype=np.int64))
        assert np.allclose(gp, np.array(mp, dtype=float), rtol=1e-12, atol=1e-13)
