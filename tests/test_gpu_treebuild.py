"""The GPU ball-tree builder (csrc/treebuild.hip) against the host builder, the oracle and the reference's golden
files: every array bit for bit (node numbering, leaf order, statistics)."""
import os

import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import check_density_against_golden, parse_mat_print_kde

pytestmark = pytest.mark.gpu

FIELDS_BT = ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation")
FIELDS_BD = ("means", "bandwidth", "bandwidthMin", "bandwidthMax")


def _same(a, b):
    for f in FIELDS_BT:
        assert np.array_equal(getattr(a.bt, f), getattr(b.bt, f)), f
    for f in FIELDS_BD:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


class _Flat:
    def __init__(self, d):
        self.dims, self.num_points = d.bt.dims, d.bt.num_points
        for f in FIELDS_BT:
            setattr(self, f, getattr(d.bt, f))
        for f in FIELDS_BD:
            setattr(self, f, getattr(d, f))


@pytest.mark.parametrize("D,N,weighted,kind", [
    (1, 2, False, "normal"), (1, 3, False, "normal"), (2, 5, True, "normal"), (1, 100, False, "normal"),
    (2, 64, False, "normal"), (2, 65, True, "normal"), (3, 257, False, "normal"), (6, 1000, False, "mixture"),
    (6, 2048, False, "mixture"), (3, 3000, True, "normal"), (2, 4096, False, "normal"), (8, 300, False, "normal"),
    (2, 500, False, "ties"), (3, 200, False, "constant"), (1, 1000, False, "ties"), (4, 33, True, "normal"),
])
def test_device_builder_equals_host_builder_and_oracle(D, N, weighted, kind):
    rng = np.random.default_rng(D * 1000 + N)
    if kind == "mixture":
        pts = rng.uniform(-2, 2, size=(3, D))[rng.integers(0, 3, N)].T + 0.5 * rng.standard_normal((D, N))
    elif kind == "ties":        # many equal keys: the scan's treatment of "not less" elements matters
        pts = rng.integers(0, 4, size=(D, N)).astype(float)
    elif kind == "constant":
        pts = np.ones((D, N)) * 0.5
    else:
        pts = rng.standard_normal((D, N))
    ks = rng.uniform(0.1, 0.5, size=D)
    w = rng.uniform(0.2, 1.0, size=N) if weighted else None
    assert kdehip._clib.kdehip_make_density_device_supported(D, N)
    g = kdehip.kde(pts, ks, w, device=0)
    h = kdehip.kde(pts, ks, w)
    _same(g, h)
    o = oracle.OracleDensity(pts, ks, w)
    for f in ("centers", "ranges", "weights", "left_child", "right_child", "permutation"):
        assert np.array_equal(getattr(g.bt, f), getattr(o, f)), f
    assert np.array_equal(g.means, o.means) and np.array_equal(g.bandwidth, o.bandwidth)


def test_device_builder_reproduces_the_reference_goldens(golden_dir):
    d = kdehip.kde([0.1, 0.45, 0.55, 3.8], [0.08], device=0)
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test1DResult.txt")), 1e-5)
    d = kdehip.kde(np.array([[0.5172, 0.7169, 0.4049], [0.0312, 1.0094, 2.0204]]), [0.1], device=0)
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test2DResult.txt")), 1e-5)
    d = kdehip.kde(np.array([[0.5172, 7.169, 4.049], [0.0312, 10.0094, -2.0204]]), [0.1, 1.0], device=0)
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test2DvarResult.txt")), 1e-4)
    gold = parse_mat_print_kde(os.path.join(golden_dir, "test1Dlcv100Result.txt"))
    x = np.loadtxt(os.path.join(golden_dir, "test1Dlcv100.txt")).ravel()
    d = kdehip.kde(x, [np.sqrt(gold["bandwidth"][100])], device=0)
    check_density_against_golden(_Flat(d), gold, 1e-4)
    for name, tol in (("test2Dlcv100", 1e-4), ("test2Dvarlcv100", 2e-3)):
        gold = parse_mat_print_kde(os.path.join(golden_dir, name + "Result.txt"))
        pts = np.ascontiguousarray(np.loadtxt(os.path.join(golden_dir, name + ".txt")).T)
        d = kdehip.kde(pts, np.sqrt(gold["bandwidth"][200:202]), device=0)
        check_density_against_golden(_Flat(d), gold, tol)


def test_batched_build_and_fallback_for_large_densities():
    rng = np.random.default_rng(3)
    items = [(rng.standard_normal((3, n)), rng.uniform(0.2, 0.4, 3)) for n in (10, 500, 1000, 2047, 20000, 1)]
    got = kdehip.kde_batch(items, device=0)       # 20000 x 3 and the single point go to the host builder
    for (p, k), g in zip(items, got):
        _same(g, kdehip.kde(p, k))


def test_products_on_device_built_trees_and_star():
    """`*` end to end: product, LOOCV bandwidth and the final tree all on the GPU."""
    rng = np.random.default_rng(5)
    a = kdehip.kde(rng.standard_normal((2, 300)), [0.3, 0.3], device=0)
    b = kdehip.kde(rng.standard_normal((2, 300)) + 0.5, [0.3, 0.3], device=0)
    ah, bh = kdehip.kde(kdehip.getPoints(a), [0.3, 0.3]), kdehip.kde(kdehip.getPoints(b), [0.3, 0.3])
    p1 = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Np=200, seed=4)
    p2 = kdehip.prodAppxMSGibbsS(None, [ah, bh], None, None, Np=200, seed=4)
    assert np.array_equal(p1[0], p2[0]) and np.array_equal(p1[1], p2[1])
    ab = kdehip.mul([a, b], seed=9)
    assert kdehip.Npts(ab) == 300 and np.isfinite(kdehip.getBW(ab)).all()
