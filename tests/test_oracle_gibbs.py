"""CPU checks of the oracle's Gibbs engine: regression against the committed known-answer fixtures,
the closed-form invariants of SURVEY.md section 4, and the reference's own statistical acceptance tests."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import kat_streams, silverman_bw


def load_kat(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"gibbs_kat_{name}.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["c1", "c2", "d6"])
def test_oracle_reproduces_committed_kat(golden_dir, name):
    k = load_kat(golden_dir, name)
    D, M, N, Np, Niter = (int(k[x]) for x in ("D", "M", "N", "Np", "Niter"))
    trees = [oracle.OracleDensity(k["points"][j], k["bw"][j]) for j in range(M)]
    _, _, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = kat_streams(nU, nN)
    p, i = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=True)
    assert np.array_equal(i, k["indices"])
    assert np.allclose(p, k["pGM_entropy"], rtol=1e-13, atol=1e-13)
    # multi-threaded split of the same work gives the same numbers (samples are independent)
    p2, i2 = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=True, nthreads=3)
    assert np.array_equal(i2, i) and np.array_equal(p2, p)


def test_closed_form_invariant_and_entropy_independence():
    """addEntropy=false => each point is the precision-weighted mean of the selected leaves; labels do
    not depend on addEntropy (reference examples/ExtractingLabels.jl:12-37; src/MSGibbs01.jl:455-459)."""
    rng = np.random.default_rng(0)
    D, M, N, Np, Niter = 3, 3, 50, 40, 2
    raw = [rng.standard_normal((D, N)) for _ in range(M)]
    bws = [rng.uniform(0.2, 0.5, size=D) for _ in range(M)]
    trees = [oracle.OracleDensity(p, b) for p, b in zip(raw, bws)]
    _, _, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    pe, ie = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=True)
    pn, i_n = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=False)
    assert np.array_equal(ie, i_n)
    num, den = np.zeros((D, Np)), np.zeros((D, Np))
    for j in range(M):
        sel = i_n[j] - 2
        num += raw[j][:, sel] / (bws[j] ** 2)[:, None]
        den += 1.0 / (bws[j] ** 2)[:, None]
    assert np.allclose(pn, num / den, rtol=0, atol=1e-14)


def _test_prods(rng, D=3, M=6, N=100, n=100, dev=1.0, MCMC=5):
    """testProds, reference test/runtests.jl:167-182 (Silverman bandwidth instead of LOOCV)."""
    P = []
    for _ in range(M):
        pts = dev * rng.standard_normal((D, N))
        P.append(oracle.OracleDensity(pts, silverman_bw(pts)))
    _, _, nU, nN = oracle.rng_sizes(M, D, n, MCMC, [N] * M)
    pGM, _ = oracle.gibbs1(P, n, MCMC, rng.random(nU), rng.standard_normal(nN))
    assert np.abs(pGM).sum() > 1e-14
    prodDev = np.sqrt(dev ** (2 * M) / (M * dev ** 2))
    t1 = np.linalg.norm(pGM.mean(axis=1)) < prodDev
    return t1 and all(0.66 * prodDev < pGM[i].std(ddof=1) < 1.33 * prodDev for i in range(D))


@pytest.mark.parametrize("kw", [dict(D=2, M=2), dict(D=2, M=4), dict(D=2, M=6), dict(D=3, M=6, MCMC=10),
                                dict(D=4, M=6, n=200, MCMC=10), dict(D=3, M=5, N=300), dict(D=2, M=7, n=300),
                                dict(D=3, M=2, MCMC=100)])
def test_reference_statistical_acceptance_matrix(kw):
    """rangeUnitTests, reference test/runtests.jl:184-201: >= 5 of 10 repetitions must pass."""
    rng = np.random.default_rng(2026)
    assert sum(bool(_test_prods(rng, **kw)) for _ in range(10)) >= 5


def test_partial_product_acceptance():
    """reference test/testPartialProd.jl:8-58 on the oracle."""
    rng = np.random.default_rng(4)
    pts1, pts2, pts3 = rng.random((2, 100)) + 10.0, rng.random((2, 100)), rng.random((2, 100)) - 10.0
    bw1, bw2, bw3 = silverman_bw(pts1), silverman_bw(pts2), silverman_bw(pts3)
    pts1[1, :] = 9999999.0
    pts3[0, :] = 9999999.0
    trees = [oracle.OracleDensity(pts1, bw1), oracle.OracleDensity(pts2, bw2), oracle.OracleDensity(pts3, bw3)]
    mask = [[1, 0], [1, 1], [0, 1]]
    _, _, nU, nN = oracle.rng_sizes(3, 2, 100, 3, [100] * 3)
    pGM, _ = oracle.gibbs1(trees, 100, 3, rng.random(nU), rng.standard_normal(nN), partialDimMask=mask)
    assert 80 < int(((0 < pGM[0]) & (pGM[0] < 10)).sum())
    assert 80 < int(((-10 < pGM[1]) & (pGM[1] < 0)).sum())


def test_oracle_label_tuples_follow_the_exact_mixture_weights():
    """The CPU oracle against mathematics (see tests/test_gpu_exact_mixture.py): final label tuples of a tiny product
    distributed like the analytic component weights of the product mixture."""
    from tests.test_gpu_exact_mixture import exact_component_weights, make_case
    D, Ns, weighted = 2, [4, 3, 4], True
    pts, sds, ws = make_case(100 + D + len(Ns), D, Ns, weighted)
    exact, mean, var = exact_component_weights(pts, sds, ws)
    trees = [oracle.OracleDensity(p, s, w) for p, s, w in zip(pts, sds, ws)]
    Np, Niter = 40_000, 25
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    rng = np.random.default_rng(7)
    x, ind = oracle.gibbs1(trees, Np, Niter, rng.random(nU), rng.standard_normal(nN), nthreads=8)
    labels, counts = np.unique(ind.T, axis=0, return_counts=True)
    freq = {tuple(int(v) for v in lab): c / Np for lab, c in zip(labels, counts)}
    for combo, w in exact.items():
        se = np.sqrt(max(w * (1 - w), 1e-12) / Np)
        assert abs(freq.get(combo, 0.0) - w) < 5.0 * se + 2e-4, (combo, w, freq.get(combo, 0.0))
    assert np.all(np.abs(x.mean(axis=1) - mean) < 5.0 * np.sqrt(var / Np) + 1e-3)
