"""GPU parity of the callers either side of the product: direct evaluation (`evaluateDualTree`, `bd(pos)`),
the LOOCV bandwidth of `kde!(points)`, and `*` -- against the oracle, the reference's LOOCV golden and
the reference's own statistical tests (which build their inputs with `kde!(randn(...))`)."""
import os

import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import parse_mat_print_kde, synth_mixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,N,Nq,weighted", [(1, 100, 33, False), (2, 257, 300, True), (3, 1000, 129, False),
                                             (6, 2048, 2048, False), (8, 50, 7, True), (1, 2, 1, False)])
def test_evaluate_direct_parity(D, N, Nq, weighted):
    rng = np.random.default_rng(10 * D + N)
    pts = rng.standard_normal((D, N))
    w = rng.uniform(0.2, 1.0, N) if weighted else None
    bw = rng.uniform(0.2, 0.6, D)
    g, o = kdehip.kde(pts, bw, w), oracle.OracleDensity(pts, bw, w)
    pos = rng.standard_normal((D, Nq)) * 1.5
    assert np.allclose(kdehip.evaluateDualTree(g, pos), oracle.eval_direct(o, pos), rtol=1e-12, atol=1e-300)
    assert np.allclose(g(pos), oracle.eval_direct(o, pos), rtol=1e-12, atol=1e-300)          # functor form
    assert np.allclose(kdehip.evaluateDualTree(g, lvFlag=True), oracle.eval_direct(o, loo=True), rtol=1e-12)
    assert np.allclose(kdehip.evaluateDualTree(g, g), oracle.eval_direct(o, loo=True), rtol=1e-12)  # bd == pos
    with pytest.raises(ValueError):
        kdehip.evaluateDualTree(g, np.zeros((D + 1, 3)))


def test_evaluate_large_batch_against_oracle_sample_and_split_invariance():
    """10000 sources x 65536 queries (the grouped, multi-block path with one pinned image each way): a sample of the
    queries against the oracle, and the same queries evaluated in two halves (another grouping of the source
    chunks) to rounding."""
    D, N, Nq = 6, 10000, 65536
    rng = np.random.default_rng(99)
    pts = rng.standard_normal((D, N))
    bw = rng.uniform(0.3, 0.6, D)
    g, o = kdehip.kde(pts, bw), oracle.OracleDensity(pts, bw)
    pos = rng.standard_normal((D, Nq)) * 1.2
    full = kdehip.evaluateDualTree(g, pos)
    assert full.shape == (Nq,) and np.all(np.isfinite(full)) and np.all(full >= 0)
    pick = rng.choice(Nq, size=96, replace=False)
    assert np.allclose(full[pick], oracle.eval_direct(o, pos[:, pick]), rtol=1e-12, atol=1e-300)
    halves = np.concatenate([kdehip.evaluateDualTree(g, pos[:, :Nq // 2]), kdehip.evaluateDualTree(g, pos[:, Nq // 2:])])
    assert np.allclose(full, halves, rtol=1e-13, atol=1e-300)
    assert np.array_equal(full, kdehip.evaluateDualTree(g, pos))   # deterministic


def test_loocv_bandwidth_reproduces_reference_golden(golden_dir):
    """UnitTest1Dlcv01 (reference test/runtests.jl:104-116) through the GPU path."""
    from tests.test_host_cpu import _Flat
    from tests.helpers import check_density_against_golden
    gold = parse_mat_print_kde(os.path.join(golden_dir, "test1Dlcv100Result.txt"))
    x = np.loadtxt(os.path.join(golden_dir, "test1Dlcv100.txt")).ravel()
    d = kdehip.kde(x)  # kde!(x): automatic bandwidth
    check_density_against_golden(_Flat(d), gold, 1e-4)


@pytest.mark.parametrize("D,N", [(1, 100), (2, 300), (3, 64), (6, 2048), (4, 1000), (2, 2), (3, 7), (2, 3000), (1, 5000),
                                 # the tile-pair kernel's circle of offsets: 2, 4 and 64 tiles (even: the half-way
                                 # offset is held by the lower tiles only), 3 and 63 tiles (odd), ragged last tiles
                                 (2, 128), (2, 193), (3, 130), (1, 4096), (1, 4000), (8, 65)])
def test_loocv_bandwidth_parity_with_oracle(D, N):
    rng = np.random.default_rng(100 + D)
    pts = rng.standard_normal((D, N)) * rng.uniform(0.3, 3.0, size=(D, 1)) + rng.uniform(-2, 2, size=(D, 1))
    g, nev = kdehip.auto_bandwidth(pts, return_evals=True)
    o, onev = oracle.auto_bandwidth(pts)
    assert np.allclose(g, o, rtol=1e-9, atol=0), (g, o)
    assert nev == onev


def _test_prods(rng, D=3, M=6, N=100, n=100, dev=1.0, MCMC=5):
    """testProds exactly as the reference builds it (test/runtests.jl:167-182): inputs via kde!(randn)."""
    P = [kdehip.kde(dev * rng.standard_normal((D, N))) for _ in range(M)]
    dummy = kdehip.kde(rng.standard_normal((D, n)), [1.0])
    pGM, _ = kdehip.prodAppxMSGibbsS(dummy, P, None, None, Niter=MCMC, seed=int(rng.integers(1 << 60)))
    assert np.abs(pGM).sum() > 1e-14
    prodDev = np.sqrt(dev ** (2 * M) / (M * dev ** 2))
    t1 = np.linalg.norm(pGM.mean(axis=1)) < prodDev
    return t1 and all(0.66 * prodDev < pGM[i].std(ddof=1) < 1.33 * prodDev for i in range(D))


@pytest.mark.parametrize("kw", [dict(D=2, M=2), dict(D=2, M=4), dict(D=2, M=6), dict(D=3, M=6, MCMC=10),
                                dict(D=4, M=6, n=200, MCMC=10), dict(D=3, M=5, N=300), dict(D=2, M=7, n=300),
                                dict(D=3, M=2, MCMC=100)])
def test_reference_range_unit_tests_with_loocv_inputs(kw):
    """rangeUnitTests (reference test/runtests.jl:184-201): >= 5 of 10 repetitions pass."""
    rng = np.random.default_rng(77)
    assert sum(bool(_test_prods(rng, **kw)) for _ in range(10)) >= 5


def test_partial_product_reference_test_with_loocv_inputs():
    """reference test/testPartialProd.jl:8-58, inputs built exactly as there (kde!(pts) bandwidths)."""
    rng = np.random.default_rng(5)
    pts1, pts2, pts3 = rng.random((2, 100)) + 10.0, rng.random((2, 100)), rng.random((2, 100)) - 10.0
    P1, P2, P3 = kdehip.kde(pts1), kdehip.kde(pts2), kdehip.kde(pts3)
    bw1, bw3 = kdehip.getBW(P1)[:, 0], kdehip.getBW(P3)[:, 0]
    pts1[1, :] = 9999999.0
    pts3[0, :] = 9999999.0
    P1, P3 = kdehip.kde(pts1, bw1), kdehip.kde(pts3, bw3)
    dummy = kdehip.kde(rng.random((2, 100)))
    pGM, _ = kdehip.prodAppxMSGibbsS(dummy, [P1, P2, P3], None, None, seed=3,
                                     partialDimMask=[[True, False], [True, True], [False, True]])
    assert 80 < int(((0 < pGM[0]) & (pGM[0] < 10)).sum())
    assert 80 < int(((-10 < pGM[1]) & (pGM[1] < 0)).sum())


def test_star_product():
    """`p * q` and `*([..])` (reference src/MSGibbs01.jl:707-736): Np = round(mean Npts), Niter = 5, kde!(pGM)."""
    rng = np.random.default_rng(8)
    p = kdehip.kde(rng.standard_normal((2, 120)) + 1.0)
    q = kdehip.kde(rng.standard_normal((2, 80)) - 1.0)
    pq = p * q  # (operator form: seeded from the OS, so only its shape is asserted)
    assert (kdehip.Ndim(pq), kdehip.Npts(pq)) == (2, 100) and np.isfinite(kdehip.getPoints(pq)).all()
    m = kdehip.getPoints(kdehip.mul([p, q], seed=11)).mean(axis=1)
    assert np.all(np.abs(m) < 0.8)  # product of N(+1, ~1) and N(-1, ~1) sits near 0 (100 samples: +-0.1 noise)
    r = kdehip.mul([p, q, p], seed=4)
    assert kdehip.Npts(r) == round((120 + 80 + 120) / 3)
    with pytest.raises(ValueError, match="same dimension"):
        kdehip.mul([p, kdehip.kde(rng.standard_normal(50), [0.3])])
    # hack fix for #70 (:713-716): one density, no entropy -> kde!(its own points)
    one = kdehip.mul([p], addEntropy=False)
    assert np.allclose(kdehip.getPoints(one), kdehip.getPoints(p))


def test_kde_auto_builds_its_tree_under_the_search_and_equals_the_sequential_form():
    """`kde!(points)` = LOOCV bandwidth, then `kde!(points, bw)` (src/KDE01.jl:3-27).  kdehip_make_density_auto builds the
    tree on the library's host threads WHILE the GPU searches (topology and means do not depend on the bandwidth) and
    fills the variances in afterwards: every array must be bit-identical to the two steps run one after the other."""
    rng = np.random.default_rng(21)
    for D, N in [(1, 2), (1, 100), (3, 257), (6, 2048), (2, 9000), (3, 1000), (8, 513)]:
        pts = rng.standard_normal((D, N)) * rng.uniform(0.5, 2.0, size=(D, 1))
        a = kdehip.kde_auto(pts)
        b = kdehip.kde_auto(pts, overlap=False)
        c = kdehip.kde(pts, kdehip.auto_bandwidth(pts))
        for other in (b, c):
            for f in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
                assert np.array_equal(getattr(a, f), getattr(other, f)), (D, N, f)
            for f in ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation"):
                assert np.array_equal(getattr(a.bt, f), getattr(other.bt, f)), (D, N, f)
    with pytest.raises(kdehip.KdeHipError):   # N >= 2 (the mirror routes a single point to the sequential form instead)
        import ctypes as C
        from kdehip import _lib
        one = np.zeros(1)
        z = np.zeros(4)
        zi = np.zeros(4, dtype=np.int64)
        P, Q = _lib.f64p, _lib.i64p
        _lib.check(_lib.lib.kdehip_make_density_auto(1, 1, _lib.ptr(one, P), _lib.ptr(z, P), None, 0, _lib.ptr(z, P), _lib.ptr(z, P),
                                                     _lib.ptr(z, P), _lib.ptr(zi, Q), _lib.ptr(zi, Q), _lib.ptr(zi, Q), _lib.ptr(zi, Q),
                                                     _lib.ptr(zi, Q), _lib.ptr(z, P), _lib.ptr(z, P), _lib.ptr(z, P), _lib.ptr(z, P)))


_TWO_LAUNCH_SCRIPT = r'''
import json, sys, numpy as np, kdehip
from tests.helpers import synth_mixture
out = {}
for D, N in [(1, 65), (1, 128), (2, 129), (1, 191), (3, 192), (1, 257), (2, 320), (6, 500), (1, 1000), (6, 1000), (2, 1999),
             (6, 2048), (1, 2049), (3, 3000), (1, 4032), (2, 4096)]:
    rng = np.random.default_rng(1000 * D + N)
    bw, ne = kdehip.auto_bandwidth(synth_mixture(rng, D, N), return_evals=True)
    out[f"{D}x{N}"] = ([float(x) for x in bw], int(ne))
print("RESULT " + json.dumps(out))
'''


def test_fused_loocv_rounds_equal_the_two_launch_rounds():
    """The one-launch LOOCV round hands its slots between workgroups and XCDs inside the launch with device-scope relaxed
    atomics (csrc/evaluate.hip slot_store / pairs_arrive: leaning on gfx950's write-through behaviour, see there).  The
    two-launch rounds have no such hand-over (kernel boundaries order everything): both forms must select the same bandwidth
    with the same number of likelihood evaluations -- tile counts 2 .. 64, odd and even circles, ragged last tiles."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    # the library's choice (speculative rounds -- three evaluations per launch, csrc/evaluate.hip loo_round_spec_kernel -- for
    # the smaller marginals, plain one-launch rounds for the larger), plain one-launch rounds everywhere, two-launch rounds
    for extra in ({}, {"KDEHIP_LOOCV_SPEC": "0"}, {"KDEHIP_LOOCV_TWO_LAUNCH": "1"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        out = subprocess.run([sys.executable, "-c", _TWO_LAUNCH_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[7:]))
    for other in res[:2]:
        assert other.keys() == res[2].keys()
        for k in other:
            (b0, n0), (b1, n1) = other[k], res[2][k]
            assert n0 == n1, (k, n0, n1)
            assert np.allclose(b0, b1, rtol=1e-9, atol=0.0), (k, b0, b1)


def test_concurrent_searches_share_the_device():
    """Three host threads searching at once on their own streams (the slots, counters and search states of a call live in
    its own device block): both must return what a lone search returns."""
    import threading
    rng = np.random.default_rng(77)
    # (3 x 1500: speculative rounds; 6 x 2048: plain one-launch rounds; 2 x 300: speculative, a few tiles)
    data = [synth_mixture(rng, 3, 1500), synth_mixture(rng, 6, 2048), synth_mixture(rng, 2, 300)]
    alone = [kdehip.auto_bandwidth(x) for x in data]
    got = [None] * len(data)

    def work(i):
        for _ in range(6):
            got[i] = kdehip.auto_bandwidth(data[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(data))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(len(data)):
        assert np.array_equal(got[i], alone[i]), i
