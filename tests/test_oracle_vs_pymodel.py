"""The C oracle against an independently written pure-Python restatement of the reference (tests/pymodel.py)
on tiny cases: two separate readings of the Julia source must agree -- trees exactly, Gibbs labels
exactly, points to 1e-13."""
import numpy as np
import pytest

from oracle import oracle
from tests import pymodel


def _tree_equal(o, m):
    N, D = o.num_points, o.dims
    for name in ("centers", "ranges", "means", "bandwidth"):
        assert np.array_equal(getattr(o, name), np.array(getattr(m, name)[1:], dtype=float)), name
    assert np.array_equal(o.weights, np.array(m.weights[1:], dtype=float))
    for name in ("left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation"):
        assert np.array_equal(getattr(o, name), np.array(getattr(m, name)[1:], dtype=np.int64)), name


@pytest.mark.parametrize("D,N,weighted", [(1, 1, False), (1, 2, False), (1, 9, True), (2, 17, False), (3, 40, True)])
def test_tree_builders_agree(D, N, weighted):
    rng = np.random.default_rng(7 * D + N)
    pts = rng.standard_normal((D, N))
    if N > 4:
        pts[:, 3] = pts[:, 1]  # a duplicate
    ks = list(rng.uniform(0.1, 0.7, D))
    w = list(rng.uniform(0.2, 1.0, N)) if weighted else None
    o = oracle.OracleDensity(pts, ks, w)
    m = pymodel.kde([list(pts[:, i]) for i in range(N)], ks, w)
    _tree_equal(o, m)


@pytest.mark.parametrize("D,Ns,Np,Niter,mask", [
    (1, [3, 3], 4, 1, None), (1, [1, 6], 5, 2, None), (2, [9, 14, 5], 6, 2, None), (3, [20, 33], 5, 3, None),
    (2, [12, 12, 12], 8, 2, [[1, 0], [1, 1], [0, 1]]), (2, [10], 4, 1, None),
])
def test_gibbs_engines_agree(D, Ns, Np, Niter, mask):
    rng = np.random.default_rng(11 * D + len(Ns) + Np)
    raw = [rng.standard_normal((D, n)) for n in Ns]
    kss = [list(rng.uniform(0.2, 0.6, D)) for _ in Ns]
    ws = [list(rng.uniform(0.3, 1.0, n)) for n in Ns]
    ot = [oracle.OracleDensity(p, k, w) for p, k, w in zip(raw, kss, ws)]
    mt = [pymodel.kde([list(p[:, i]) for i in range(p.shape[1])], k, w) for p, k, w in zip(raw, kss, ws)]
    M = len(Ns)
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    for addEntropy in (True, False):
        op, oi = oracle.gibbs1(ot, Np, Niter, randU, randN, addEntropy=addEntropy, partialDimMask=mask)
        mp, mi = pymodel.prodAppxMSGibbsS(mt, Np, Niter, list(randU), list(randN), addEntropy, mask)
        assert np.array_equal(oi, np.array(mi, dtype=np.int64))
        assert np.allclose(op, np.array(mp, dtype=float), rtol=0, atol=1e-13)
