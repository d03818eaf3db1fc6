"""Wavefront teams (csrc/gibbs_device.hpp "wavefront teams", csrc/gibbs_lean.hip): one chain on 2 or 4 wavefronts of a
16-wavefront workgroup on the deep levels (plan variants 52 / 54; chosen by default for few chains on deep trees).
Results must not depend on the team size: every variant adds the same lane sums in the same association (LaneAcc),
so labels, label traces and points are compared bit for bit between one wavefront per chain at every workgroup
width, teams of 2 and teams of 4 -- and with the oracle (labels exactly, points to 1e-11)."""
import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _needs_a_team_build():
    """The team instantiations are an optional part of the library since round 4 (`make TEAMS=1 -C
    kerneldensityestimate.jl_amd/csrc`: measured slower than one wavefront per chain at every BASELINE shape, never selected
    by default, a fifth of the build time).  A default build runs the team variants as 16 one-wavefront chains and says so
    through kdehip_product_launch_geometry: these tests then have nothing to test."""
    rng = np.random.default_rng(0)
    t = [kdehip.kde(synth_mixture(rng, 3, 3000), [0.3]) for _ in range(2)]
    with kdehip.ProductPlan(t) as plan:
        plan.set_variant(52)
        built = plan.launch_geometry(256)["team"] == 2
    if not built:
        pytest.skip("libkdehip.so was built without the wavefront-team kernels (make TEAMS=1)")


WIDTHS = (2, 8, 16)   # one wavefront per chain: 4, 8, 16 chains per workgroup
TEAMS = (52, 54)          # 16 wavefronts per workgroup as 8 chains x 2 / 4 chains x 4


def _trees(seed, D, Ns, weighted=False, sep=0.0):
    rng = np.random.default_rng(seed)
    g, o = [], []
    for k, N in enumerate(Ns):
        pts = synth_mixture(rng, D, N)
        pts[0] += sep * k
        ks = silverman_bw(pts) if N > 1 else np.full(D, 0.5)
        ks = np.where(ks > 0, ks, 0.5)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        g.append(kdehip.kde(pts, ks, w))
        o.append(oracle.OracleDensity(pts, ks, w))
    return g, o


@pytest.mark.parametrize("D,Ns,Np,Niter,weighted", [
    (6, [1000] * 4, 301, 3, False),           # config 3's shape: streamed levels of 8 and 16 rows per lane shared
    (3, [5000] * 3, 45, 2, False),            # chunked tiles, 79 rows per lane: segment notes exchanged as well
    (3, [5000] * 8, 21, 1, False),            # config 4's shape (8 densities: the second set of instantiations)
    (2, [3000, 100, 700], 77, 2, True),       # ragged: shared levels hold tiles below the sharing threshold
    (1, [2048, 2047, 513], 130, 2, False),    # resident deep tiles (small D): two barriers per shared step
    (2, [100, 120], 50, 3, False),            # no deep level at all: a forced team leaves the chain to member 0
    (6, [10000] * 4, 19, 1, False),           # config 5's shape in fp64: 157 rows per lane, several segments
    (8, [1100, 900], 37, 2, True),
])
def test_team_sizes_give_identical_results(D, Ns, Np, Niter, weighted):
    g, o = _trees(4000 + D + len(Ns), D, Ns, weighted)
    seed = 4242
    with kdehip.ProductPlan(g) as plan:
        assert plan.fast_math_path
        res = {}
        for variant in (0,) + WIDTHS + TEAMS:
            if len(Ns) > 4 and variant == 2:   # (5..8 densities: lean kernel at 8 and 16 chains per workgroup only)
                continue
            plan.set_variant(variant)
            res[variant] = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for variant, r in res.items():
        for a, b in zip(res[0], r):
            assert np.array_equal(a, b), variant
    if max(Ns) <= 5000:
        u, n = kdehip.philox_streams(seed, 0, Np, K, R)
        op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
        assert np.array_equal(res[0][1], oi) and np.array_equal(res[0][2], ol)
        assert np.allclose(res[0][0], op, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("variant", TEAMS)
def test_team_with_caller_streams(variant):
    """caller-supplied randU/randN in the reference's consumption order, addEntropy on and off"""
    D, Ns, Np, Niter = 3, [1500, 1200, 1025], 61, 2
    g, o = _trees(6, D, Ns)
    K, R, nU, nN = oracle.rng_sizes(len(Ns), D, Np, Niter, Ns)
    rng = np.random.default_rng(3)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    import torch
    dev = torch.device("cuda", 0)
    dU, dN = torch.from_numpy(randU).to(dev), torch.from_numpy(randN).to(dev)
    for addEntropy in (True, False):
        op, oi = oracle.gibbs1(o, Np, Niter, randU, randN, addEntropy=addEntropy)
        with kdehip.ProductPlan(g) as plan:
            plan.set_variant(variant)
            d_pts = torch.zeros(D * Np, dtype=torch.float64, device=dev)
            d_ind = torch.zeros(len(Ns) * Np, dtype=torch.int64, device=dev)
            plan.sample_streams_device(Np, Niter, dU, nU, dN, nN, addEntropy, d_pts, d_ind)
            torch.cuda.synchronize()
            gp = d_pts.cpu().numpy().reshape(Np, D).T
            gi = d_ind.cpu().numpy().reshape(Np, len(Ns)).T
            assert np.array_equal(gi, oi), (variant, addEntropy)
            assert np.allclose(gp, op, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("Np", [61, 130])   # not a multiple of any workgroup's chain count: the last workgroup replays
def test_fallback_count_does_not_depend_on_the_geometry(Np):
    """Two clusters far apart: many label draws take the reference's underflow branch (src/MSGibbs01.jl:311-315).  The
    plan's counter must equal the oracle's whatever the workgroup width or team size -- surplus wavefronts replaying
    the last chain and all but the first member of a team do not count."""
    D, Ns, Niter = 2, [1200, 1100], 2
    g, o = _trees(11, D, Ns, sep=60.0)
    seed = 5
    counts = {}
    ref = None
    for variant in (0,) + WIDTHS + TEAMS + (38,):
        with kdehip.ProductPlan(g) as plan:
            plan.set_variant(variant)
            r = plan.sample(Np, Niter=Niter, seed=seed)
            counts[variant] = plan.fallback_count()
            K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        if ref is None:
            ref = r
        else:
            assert np.array_equal(ref[1], r[1]) and np.array_equal(ref[0], r[0]), variant
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    oracle.fallback_count(reset=True)
    _, oi = oracle.gibbs1(o, Np, Niter, u, n)
    n_oracle = oracle.fallback_count(reset=True)
    assert np.array_equal(ref[1], oi)
    assert n_oracle > 0
    assert all(c == n_oracle for c in counts.values()), (counts, n_oracle)


def test_launch_geometry_reports_teams():
    """config 4's shape at its per-GPU batch: the forced team variants report their geometry; the DEFAULT launch is one
    wavefront per chain (teams measured slower on MI355X, csrc/kdehip_internal.hpp kTeamMinShare) -- pinned here so that
    switching the default on is a deliberate change."""
    D, Ns, Np, Niter = 3, [5000] * 8, 2048, 1
    g, _ = _trees(12, D, Ns)
    with kdehip.ProductPlan(g) as plan:
        assert plan.launch_geometry(Np) == {"waves": 8, "team": 1}
        assert plan.launch_geometry(16384) == {"waves": 16, "team": 1}
        a = plan.sample(Np, Niter=Niter, seed=9)
        plan.set_variant(52)
        assert plan.launch_geometry(Np) == {"waves": 16, "team": 2}
        b = plan.sample(Np, Niter=Niter, seed=9)
        plan.set_variant(54)
        assert plan.launch_geometry(Np) == {"waves": 16, "team": 4}
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
