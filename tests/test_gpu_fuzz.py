"""Randomised parity sweep: many small random products (dimension, density count, sizes, weights, masks,
Niter, chain counts around the workgroup/table thresholds) through the HIP path vs the oracle."""
import os

import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.test_gpu_parity import _compare, _pair

pytestmark = pytest.mark.gpu


def _random_case(rng):
    D = int(rng.integers(1, 9))
    M = int(rng.integers(1, 7))
    Ns = [int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 33, 64, 65, 100, 128, 200, 257, 300, 448, 512, 513, 600]))
          for _ in range(M)]  # up to 10 rows per lane: both second-pass forms and their boundaries
    Np = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 40, 64, 130]))
    Niter = int(rng.integers(0, 4))
    weighted = bool(rng.integers(0, 2))
    mask = None
    if D > 1 and M > 1 and rng.random() < 0.3:
        mask = rng.random((M, D)) < 0.7
        mask[rng.integers(0, M), :] = True          # at least one fully informed density
        for d in range(D):
            if not mask[:, d].any():
                mask[rng.integers(0, M), d] = True
    return D, M, Ns, Np, Niter, weighted, mask


@pytest.mark.parametrize("seed", range(int(os.environ.get("KDEHIP_FUZZ_N", 60))))  # soak: KDEHIP_FUZZ_N=3000
def test_random_products_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    D, M, Ns, Np, Niter, weighted, mask = _random_case(rng)
    gp, op = [], []
    for n in Ns:
        pts = rng.standard_normal((D, n)) * rng.uniform(0.3, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))
        ks = rng.uniform(0.1, 0.8, size=D)
        w = rng.uniform(0.1, 1.0, size=n) if weighted else None
        a, b = _pair(pts, ks, w)
        gp.append(a)
        op.append(b)
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    for addEntropy in (True, False):
        g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN,
                                    addEntropy=addEntropy, partialDimMask=mask)
        o = oracle.gibbs1(op, Np, Niter, randU, randN, addEntropy=addEntropy, partialDimMask=mask)
        _compare(g, o, tol=1e-11)
    # the Philox path and its per-level label trace
    with kdehip.ProductPlan(gp, partialDimMask=mask) as plan:
        gp_, gi_, gl_ = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        u, n = kdehip.philox_streams(seed, 0, Np, plan.randu_per_sample(Niter), plan.randn_per_sample())
    op_, oi_, ol_ = oracle.gibbs1(op, Np, Niter, u, n, partialDimMask=mask, want_labels=True)
    _compare((gp_, gi_), (op_, oi_), tol=1e-11)
    if Niter > 0:
        assert np.array_equal(gl_, ol_)


@pytest.mark.parametrize("seed", range(int(os.environ.get("KDEHIP_FUZZ_N", 40))))
def test_random_products_every_width_up_to_nine_densities(seed):
    """the same sweep with 2..9 densities at a random workgroup width (4, 8, 16 chains): products of 5..8 densities
    take the lean kernel at 8 and 16, the general kernel at 4 and 12; 9 densities always the general kernel"""
    rng = np.random.default_rng(5000 + seed)
    D, _, _, Np, Niter, weighted, _ = _random_case(rng)
    M = int(rng.integers(2, 10))
    Ns = [int(rng.choice([1, 3, 16, 33, 64, 65, 100, 128, 200, 257, 300, 513])) for _ in range(M)]
    variant = int(rng.choice([2, 8, 16]))
    gp, op = [], []
    for n in Ns:
        pts = rng.standard_normal((D, n)) * rng.uniform(0.3, 2.0, size=(D, 1)) + rng.uniform(-1, 1, size=(D, 1))
        ks = rng.uniform(0.15, 0.8, size=D)
        w = rng.uniform(0.1, 1.0, size=n) if weighted else None
        a, b = _pair(pts, ks, w)
        gp.append(a)
        op.append(b)
    with kdehip.ProductPlan(gp) as plan:
        plan.set_variant(variant)
        gp_, gi_, gl_ = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        u, n = kdehip.philox_streams(seed, 0, Np, plan.randu_per_sample(Niter), plan.randn_per_sample())
    op_, oi_, ol_ = oracle.gibbs1(op, Np, Niter, u, n, want_labels=True)
    _compare((gp_, gi_), (op_, oi_), tol=1e-11)
    if Niter > 0:
        assert np.array_equal(gl_, ol_)
