"""on-manifold operators as an ENUMERATED per-dimension manifold {0: Euclidean, 1: circular (2 pi)}.

The reference takes its operators as per-dimension function tuples `addop / diffop / getMu / getLambda`
(src/MSGibbs01.jl:650-653) and applies them at three hook points -- the difference inside a kernel evaluation (:290), the
product's mean and information (:183-184, 210-213) and the sample's composition with its noise (:456) -- but defines only the
Euclidean set; the circular functions live in its callers' packages.  The semantic of the enum's circular member is
therefore THIS repo's (include/kdehip.h "manifolds"): wrap to [-pi, pi), information-weighted mean in the tangent space at
the first contributing kernel.  What can be checked without the reference: (1) the all-Euclidean enum is the plain path,
(2) the C oracle's enum equals tests/pymodel.py -- the restatement that keeps the reference's OWN structure, operator tuples
of callables at the reference's hook points -- run with circular callables, hook by hook, (3) the semantic does what a
circular manifold must (a product of kernels either side of the cut lands at the cut, not at 0)."""
import numpy as np
import pytest

from oracle import oracle
from tests import pymodel


def _case(seed, D, Ns, circ, spread=2.5):
    """densities of angles (circular dimensions: a cloud centred at the cut +-pi, so its kernels straddle it)"""
    rng = np.random.default_rng(seed)
    raw = []
    for n in Ns:
        p = rng.standard_normal((D, n)) * 0.6
        for d in range(D):
            if circ[d]:
                p[d] = pymodel_wrap(np.pi + spread * rng.standard_normal(n) * 0.4)   # (centred AT the cut)
        raw.append(p)
    kss = [list(rng.uniform(0.2, 0.5, D)) for _ in Ns]
    ws = [list(rng.uniform(0.3, 1.0, n)) for n in Ns]
    return rng, raw, kss, ws


def pymodel_wrap(a):
    return np.array([pymodel.wrapRad(float(t)) for t in np.atleast_1d(a)])


def _both(raw, kss, ws):
    ot = [oracle.OracleDensity(p, k, w) for p, k, w in zip(raw, kss, ws)]
    mt = [pymodel.kde([list(p[:, i]) for i in range(p.shape[1])], k, w) for p, k, w in zip(raw, kss, ws)]
    return ot, mt


@pytest.mark.parametrize("D,Ns,Np,Niter", [(1, [5, 7], 6, 2), (3, [20, 9, 14], 5, 2)])
def test_all_euclidean_enum_is_the_plain_path(D, Ns, Np, Niter):
    rng, raw, kss, ws = _case(3 + D, D, Ns, [False] * D)
    ot, _ = _both(raw, kss, ws)
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    a = oracle.gibbs1(ot, Np, Niter, randU, randN, want_labels=True)
    b = oracle.gibbs1(ot, Np, Niter, randU, randN, want_labels=True, manifold=[0] * D)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# one case per hook, then all of them together with a mask
@pytest.mark.parametrize("D,Ns,Np,Niter,circ,mask,why", [
    (1, [6, 6], 8, 0, [1], None, "diffop (:290) and addop (:456) alone: Niter = 0 runs no leave-one-out product"),
    (1, [9, 5, 7], 8, 3, [1], None, "getMu / getLambda (:183-184): the leave-one-out product of two circular kernels"),
    (2, [12, 15], 6, 2, [0, 1], None, "a Euclidean and a circular dimension side by side"),
    (3, [20, 33, 8], 5, 2, [1, 0, 1], None, "SE(2)-like with a second angle: 3-D, three densities"),
    (2, [12, 12, 12], 8, 2, [1, 1], [[1, 0], [1, 1], [0, 1]], "partialDimMask: the first CONTRIBUTING kernel is the reference angle"),
])
def test_enum_equals_the_operator_tuples_at_the_references_hooks(D, Ns, Np, Niter, circ, mask, why):
    rng, raw, kss, ws = _case(17 * D + len(Ns) + Np, D, Ns, circ)
    ot, mt = _both(raw, kss, ws)
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    ops = [pymodel.CIRCULAR_OPS if c else pymodel.EUCLID_OPS for c in circ]
    for addEntropy in (True, False):
        op, oi = oracle.gibbs1(ot, Np, Niter, randU, randN, addEntropy=addEntropy, partialDimMask=mask, manifold=circ)
        mp, mi = pymodel.prodAppxMSGibbsS(mt, Np, Niter, list(randU), list(randN), addEntropy, mask, ops=ops)
        assert np.array_equal(oi, np.array(mi, dtype=np.int64)), why
        assert np.allclose(op, np.array(mp, dtype=float), rtol=0, atol=1e-13), why
        for d in range(D):
            if circ[d] and addEntropy:
                assert np.all(op[d] >= -np.pi) and np.all(op[d] < np.pi), "addop wraps the drawn sample"
    # and the circular run is NOT the Euclidean one on these data (the kernels straddle the cut)
    ep, ei = oracle.gibbs1(ot, Np, Niter, randU, randN, partialDimMask=mask)
    cp, ci = oracle.gibbs1(ot, Np, Niter, randU, randN, partialDimMask=mask, manifold=circ)
    assert not (np.array_equal(ei, ci) and np.allclose(ep, cp))


def test_product_across_the_cut_lands_at_the_cut():
    """Two one-kernel densities at +3.1 and -3.1 rad (0.083 rad apart on the circle): the circular product's mean is at the
    cut (|x| > 3.1), the Euclidean one at 0; with equal bandwidths it is exactly the circular midpoint."""
    a = oracle.OracleDensity(np.array([[3.1]]), [0.2])
    b = oracle.OracleDensity(np.array([[-3.1]]), [0.2])
    K, R, nU, nN = oracle.rng_sizes(2, 1, 4, 1, [1, 1])
    rng = np.random.default_rng(0)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    pe, _ = oracle.gibbs1([a, b], 4, 1, randU, randN, addEntropy=False)
    pc, _ = oracle.gibbs1([a, b], 4, 1, randU, randN, addEntropy=False, manifold=[1])
    assert np.allclose(pe, 0.0, atol=1e-12)
    assert np.all(np.abs(pc) > 3.1)
    assert np.allclose(np.abs(pc), np.pi, atol=1e-12)


def test_bad_manifold_value_is_an_argument_error():
    a = oracle.OracleDensity(np.array([[0.0, 1.0]]), [0.2])
    with pytest.raises(IndexError):
        oracle.gibbs1([a, a], 2, 1, np.full(64, 0.5), np.zeros(64), manifold=[2])
