"""SURVEY.md 8(f) row 4 on the GPU: the enumerated per-dimension manifold {Euclidean, circular (2 pi)} of
`kdehip_gibbs1_manifold` (include/kdehip.h "manifolds") against the CPU oracle's same enum (oracle/kde_oracle.c
okde_gibbs1_manifold), one case per hook point of the reference (src/MSGibbs01.jl:290 diffop; :183-184 / 210-213 getMu,
getLambda; :456 addop), then mixed dimensions, masks, label traces and larger frontiers.  Labels must be identical, points
equal to 1e-12 (same expressions in the same order; fp64 divide and floor are exact on both sides).  tests/
test_oracle_manifold.py pins the oracle's enum against the reference's own operator-tuple structure (tests/pymodel.py)."""
import numpy as np
import pytest

import kdehip
from oracle import oracle

pytestmark = pytest.mark.gpu


def _wrap(t):
    return t - 2.0 * np.pi * np.floor((t + np.pi) / (2.0 * np.pi))


def _trees(seed, D, Ns, circ):
    rng = np.random.default_rng(seed)
    g, o = [], []
    for n in Ns:
        p = rng.standard_normal((D, n)) * 0.7
        for d in range(D):
            if circ[d]:
                p[d] = _wrap(np.pi + 1.0 * rng.standard_normal(n))   # a cloud centred AT the cut
        ks = rng.uniform(0.15, 0.5, D)
        w = rng.uniform(0.3, 1.0, n)
        g.append(kdehip.kde(p, ks, w))
        o.append(oracle.OracleDensity(p, ks, w))
    return rng, g, o


@pytest.mark.parametrize("D,Ns,Np,Niter,circ,mask,why", [
    (1, [6, 6], 64, 0, [1], None, "diffop (:290) and addop (:456) alone: Niter = 0 runs no leave-one-out product"),
    (1, [9, 5, 7], 64, 3, [1], None, "getMu / getLambda (:183-184) of two circular kernels"),
    (2, [40, 55], 100, 2, [0, 1], None, "a Euclidean and a circular dimension side by side"),
    (3, [200, 333, 80], 150, 2, [1, 0, 1], None, "SE(2)-like with a second angle"),
    (2, [120, 120, 120], 96, 2, [1, 1], [[1, 0], [1, 1], [0, 1]], "partialDimMask: the first CONTRIBUTING kernel is the reference angle"),
    (6, [1000, 700, 1000, 513], 130, 3, [0, 0, 0, 1, 1, 1], None, "SE(3)-like: three positions, three angles, streamed levels"),
    (3, [5000, 3000], 70, 1, [0, 0, 1], None, "chunked tiles"),
])
def test_gpu_enum_equals_the_oracles(D, Ns, Np, Niter, circ, mask, why):
    rng, g, o = _trees(23 * D + len(Ns) + Np, D, Ns, circ)
    M = len(Ns)
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    for addEntropy in (True, False):
        op, oi, ol = oracle.gibbs1(o, Np, Niter, randU, randN, addEntropy=addEntropy, partialDimMask=mask, manifold=circ,
                                   want_labels=True)
        glbs = kdehip.makeEmptyGbGlb()
        glbs.recordChoosen = True
        gp, gi = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                         addEntropy=addEntropy, partialDimMask=mask, manifold=circ, glbs=glbs)
        assert np.array_equal(gi, oi), why
        assert np.allclose(gp, op, rtol=0, atol=1e-12), (why, float(np.abs(gp - op).max()))
        for d in range(D):
            if circ[d] and addEntropy:
                assert np.all(gp[d] >= -np.pi) and np.all(gp[d] < np.pi)
    # the circular run is not the Euclidean one on these data, and the all-Euclidean enum is the plain call
    ep, ei = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN, partialDimMask=mask)
    zp, zi = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN, partialDimMask=mask,
                                     manifold=[0] * D)
    assert np.array_equal(zi, ei) and np.allclose(zp, ep, rtol=0, atol=1e-12)   # (generic against fast arithmetic: labels equal)
    cp, ci = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN, partialDimMask=mask,
                                     manifold=circ)
    assert not (np.array_equal(ci, ei) and np.allclose(cp, ep))


def test_product_across_the_cut_lands_at_the_cut_on_the_gpu():
    a = kdehip.kde(np.array([[3.1]]), [0.2])
    b = kdehip.kde(np.array([[-3.1]]), [0.2])
    pe, _ = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Niter=1, Np=8, seed=1, addEntropy=False)
    pc, _ = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Niter=1, Np=8, seed=1, addEntropy=False, manifold=["circular"])
    assert np.allclose(pe, 0.0, atol=1e-12)
    assert np.allclose(np.abs(pc), np.pi, atol=1e-12)


def test_manifold_with_the_device_stream_seed_equals_explicit_streams():
    rng, g, _ = _trees(5, 2, [50, 60], [0, 1])
    p1, i1 = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=2, Np=40, seed=77, manifold=["euclid", "circular"])
    L = kdehip.nlevels(60)
    u, n = kdehip.philox_streams(77, 0, 40, 2 * (1 + L * 3), 2 * (L + 1))
    p2, i2 = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=2, Np=40, randU=u, randN=n, manifold=[0, 1])
    assert np.array_equal(i1, i2) and np.array_equal(p1, p2)


def test_manifold_argument_errors():
    g = [kdehip.kde(np.random.default_rng(0).standard_normal((2, 20)), [0.3]) for _ in range(2)]
    with pytest.raises(ValueError):
        kdehip.prodAppxMSGibbsS(None, g, None, None, Np=4, seed=1, manifold=[1])          # one entry per dimension
    with pytest.raises(kdehip.KdeHipError):
        kdehip.prodAppxMSGibbsS(None, g, None, None, Np=4, seed=1, manifold=[0, 2])       # not a member of the enum
