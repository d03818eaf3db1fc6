"""Ground truth that does not pass through any restatement of the reference: the product of M Gaussian mixtures is itself a
mixture whose component (i_1, .., i_M) has the analytic weight  w ~ prod_k w_{i_k} * Z(i_1..i_M),  Z = integral of the
product of the M Gaussians (closed form for diagonal covariances).  The sampler's sweep step draws label j from its exact
conditional given the other labels (makeFasterSampleIndex!, src/MSGibbs01.jl:250-328: leave-one-out product, w_z *
N(mu_z; m, C + sigma_z^2)), so with enough sweeps at the leaf level the final label tuples must be distributed like the
exact component weights, and the product samples like the exact mixture.  This pins the ARITHMETIC of the kernel
evaluation (weights, variances, the normalisation of each kernel) against mathematics -- the part of the Gibbs path the
reference holds no golden vector for.  The same check runs on the CPU oracle in
tests/test_oracle_gibbs.py."""
import itertools

import numpy as np
import pytest

import kdehip

pytestmark = pytest.mark.gpu


def exact_component_weights(points, sds, weights):
    """points[k]: (D, N_k); sds[k]: (D,) bandwidth standard deviations; weights[k]: (N_k,) normalised.
    Returns {label tuple as the reference returns it -- bt.permutation + 1, i.e. the 1-based point index plus one
    (src/MSGibbs01.jl:615) --: weight}, the mixture mean and the mixture variance per dimension."""
    M = len(points)
    D = points[0].shape[0]
    out = {}
    for combo in itertools.product(*[range(p.shape[1]) for p in points]):
        logz = 0.0
        for d in range(D):
            mu = np.array([points[k][d, combo[k]] for k in range(M)])
            var = np.array([sds[k][d] ** 2 for k in range(M)])
            prec = 1.0 / var
            vstar = 1.0 / prec.sum()
            mstar = vstar * (mu * prec).sum()
            logz += (-0.5 * (M - 1) * np.log(2 * np.pi) + 0.5 * (np.log(vstar) - np.log(var).sum())
                     - 0.5 * ((mu * mu * prec).sum() - mstar * mstar / vstar))
        out[tuple(c + 2 for c in combo)] = np.exp(logz) * np.prod([weights[k][combo[k]] for k in range(M)])
    tot = sum(out.values())
    for k in out:
        out[k] /= tot
    mean = np.zeros(D)
    second = np.zeros(D)
    for combo, w in out.items():
        for d in range(D):
            mu = np.array([points[k][d, combo[k] - 2] for k in range(M)])
            prec = 1.0 / np.array([sds[k][d] ** 2 for k in range(M)])
            vstar = 1.0 / prec.sum()
            mstar = vstar * (mu * prec).sum()
            mean[d] += w * mstar
            second[d] += w * (vstar + mstar * mstar)
    return out, mean, second - mean * mean


CASES = [
    # D, points per density, weighted
    (1, [4, 4], False),
    (2, [4, 3, 4], True),
    (3, [5, 4], True),
    (1, [3, 3, 3, 3], False),
]


def make_case(seed, D, Ns, weighted):
    rng = np.random.default_rng(seed)
    pts = [rng.uniform(-1.0, 1.0, size=(D, n)) for n in Ns]
    sds = [rng.uniform(0.5, 0.9, size=D) for _ in Ns]
    ws = [rng.uniform(0.3, 1.0, size=n) if weighted else np.ones(n) for n in Ns]
    ws = [w / w.sum() for w in ws]
    return pts, sds, ws


@pytest.mark.parametrize("D,Ns,weighted", CASES)
@pytest.mark.parametrize("prec", [64, 32])
def test_label_tuples_follow_the_exact_mixture_weights(D, Ns, weighted, prec):
    pts, sds, ws = make_case(100 + D + len(Ns), D, Ns, weighted)
    exact, mean, var = exact_component_weights(pts, sds, ws)
    trees = [kdehip.kde(p, s, w) for p, s, w in zip(pts, sds, ws)]
    Np, Niter = 200_000, 25
    with kdehip.ProductPlan(trees, precision=prec) as plan:
        x, ind = plan.sample(Np, Niter=Niter, seed=2026)
    labels, counts = np.unique(ind.T, axis=0, return_counts=True)
    freq = {tuple(int(v) for v in lab): c / Np for lab, c in zip(labels, counts)}
    # every component within 5 standard errors of its exact weight (a few hundred components at most)
    for combo, w in exact.items():
        se = np.sqrt(max(w * (1 - w), 1e-12) / Np)
        assert abs(freq.get(combo, 0.0) - w) < 5.0 * se + 1e-4, (combo, w, freq.get(combo, 0.0))
    # Pearson chi-square over the components that carry mass
    keys = [k for k, w in exact.items() if w * Np >= 20]
    chi2 = sum((freq.get(k, 0.0) * Np - exact[k] * Np) ** 2 / (exact[k] * Np) for k in keys)
    dof = len(keys) - 1
    assert chi2 < dof + 6.0 * np.sqrt(2.0 * dof) + 10.0, (chi2, dof)
    # the product samples themselves: mean and variance of the exact mixture
    assert np.all(np.abs(x.mean(axis=1) - mean) < 5.0 * np.sqrt(var / Np) + 1e-4)
    assert np.all(np.abs(x.var(axis=1) - var) < 0.02 * var + 1e-3)
