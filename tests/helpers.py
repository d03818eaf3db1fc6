"""Shared helpers for the test-suite (parsing of the reference's MATLAB golden files, synthetic inputs)."""
import numpy as np


def parse_mat_print_kde(path):
    """Mirror of parseMatPrintKDE (reference test/runtests.jl:8-19): `name=[a, b, ...]` lines."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if "=" not in line:
                continue
            name, rest = line.split("=", 1)
            body = rest.split("[", 1)[1].split("]", 1)[0]
            out[name] = np.array([float(x) for x in body.split(",") if x.strip()])
    return out


def inds_match(ref0, ours1):
    """Index rule of testSubtract/testInds (reference test/runtests.jl:55-68):
    golden is 0-based; ref<0 => ours == -1, else ours == ref+1."""
    ref0 = np.asarray(ref0).astype(np.int64)
    ours1 = np.asarray(ours1).astype(np.int64)
    ok = np.where(ref0 < 0, ours1 + 1 == 0, ref0 + 1 - ours1 == 0)
    return bool(ok.all())


def check_density_against_golden(d, gold, tol):
    """The full testSubtract comparison (reference test/runtests.jl:42-83)."""
    N = d.num_points
    assert int(gold["dims"][0]) == d.dims and int(gold["num_points"][0]) == N
    for name, ours in [("centers", d.centers), ("ranges", d.ranges), ("weights", d.weights),
                       ("means", d.means), ("bandwidth", d.bandwidth), ("bwMin", d.bandwidthMin),
                       ("bwMax", d.bandwidthMax)]:
        n = np.linalg.norm(gold[name] - np.asarray(ours))
        assert n <= tol, (name, n)
    assert inds_match(gold["left_child"], d.left_child)
    assert inds_match(gold["right_child"], d.right_child)
    assert inds_match(gold["lowest_leaf"], d.lowest_leaf)
    assert inds_match(gold["highest_leaf"], d.highest_leaf)
    assert inds_match(gold["permutation"][N:], d.permutation[N:])


def synth_mixture(rng, D, N, ncomp=3, spread=2.0, std=0.5):
    """3-component Gaussian mixture around the origin (SURVEY.md 8(d) recipe, numpy RNG variant)."""
    centres = rng.uniform(-spread, spread, size=(ncomp, D))
    comp = rng.integers(0, ncomp, size=N)
    pts = centres[comp] + std * rng.standard_normal((N, D))
    return np.ascontiguousarray(pts.T)  # D x N


def silverman_bw(pts):
    D, N = pts.shape
    return pts.std(axis=1, ddof=1) * (4.0 / ((D + 2.0) * N)) ** (1.0 / (D + 4.0))


# ---- closed-form streams of the committed Gibbs known-answer fixtures (tests/golden/gibbs_kat_*.npz)
def weyl(n, alpha, offset=0.0):
    i = np.arange(1, n + 1, dtype=np.float64)
    return np.mod(offset + i * alpha, 1.0)


def weyl_normal(n, a1, a2):
    u1 = np.clip(weyl(n, a1, 0.11), 1e-12, 1.0)
    u2 = weyl(n, a2, 0.37)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)



def kat_streams(nU, nN):
    """randU / randN of the KAT fixtures (reference allocation sizes, src/MSGibbs01.jl:661-662)."""
    return weyl(nU, (np.sqrt(5.0) - 1.0) / 2.0, 0.0123), weyl_normal(nN, np.sqrt(7.0) % 1.0, np.sqrt(11.0) % 1.0)
