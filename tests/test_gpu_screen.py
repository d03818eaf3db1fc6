"""fp32 screening of the deep levels with fp64 certification (csrc/screen_device.hpp; kdehip_product_screen_stats).

The screened run (default) must give the labels and points of the run without screening (plan variant 5) and of the
all-global run (variant 1) BIT FOR BIT, and the oracle's labels; the statistics must show that the screen really ran
(levels > 0, steps > 0) and that only a small share of its steps had to be repeated in fp64.  Inputs outside the ranges the
error bound assumes must fall back to fp64 silently (still identical)."""
import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


def _trees(seed, D, Ns, shift=0.0, scale=1.0, weighted=False, want_oracle=True):
    rng = np.random.default_rng(seed)
    g, o = [], []
    for N in Ns:
        pts = synth_mixture(rng, D, N) * scale + shift
        ks = silverman_bw(pts)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        g.append(kdehip.kde(pts, ks, w))
        if want_oracle:
            o.append(oracle.OracleDensity(pts, ks, w))
    return g, o


def _run_variants(plan, Np, Niter, seed, variants=(0, 5, 1)):
    res = {}
    for v in variants:
        plan.set_variant(v)
        res[v] = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
    plan.set_variant(0)
    return res


@pytest.mark.parametrize("D,Ns,Np,Niter,weighted,levels", [
    (6, [1000] * 4, 2048, 10, False, 2),     # BASELINE config 3: levels 9 (per-node bandwidths) and 10 (leaves) screened
    (6, [1000] * 4, 77, 3, True, 2),         # weighted, a ragged last workgroup
    (6, [700, 1000, 513], 300, 4, False, None),  # ragged sizes: level 10 has 700 / 1000 / 513 leaves
    (2, [3000, 2500], 200, 3, False, None),
    (4, [2000] * 4, 128, 2, False, None),
    (3, [1500] * 8, 96, 2, False, None),     # 8 densities: the register-resident kernel's other translation unit
    (6, [2048] * 4, 136, 3, False, None),    # levels 10 and 11: screen tiles streamed one per step
    (3, [3000, 2048, 2500], 100, 3, True, None),  # streamed screen tiles of ragged sizes, weighted
    (6, [4096] * 4, 1100, 2, False, None),   # level 12 (4096 leaves, 64 rows per lane): screen tiles in CHUNKS, second pass from global memory
    (6, [8000, 5000, 4096], 1100, 2, True, None),  # chunked, up to 125 rows per lane (two second-pass rounds), per-node bandwidths at level 12
    (6, [4096] * 4, 60, 2, False, None),     # 4 chains per workgroup: that build has no chunked screens (fp64 there), the rest is screened
    (3, [5000] * 8, 160, 2, False, None),    # BASELINE config 4's shape: levels 9-11 resident / streamed, 12-13 chunked
])
def test_screened_run_is_the_fp64_run(D, Ns, Np, Niter, weighted, levels):
    g, o = _trees(4200 + D + len(Ns), D, Ns, weighted=weighted, want_oracle=(Np <= 300))
    seed = 911
    with kdehip.ProductPlan(g) as plan:
        assert plan.fast_math_path
        if len(Ns) == 8:  # (8 densities: the register-resident kernel at 8 / 16 chains per workgroup)
            res = {}
            for v in (8, 16):
                plan.set_variant(v)
                res[v] = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
            st = plan.screen_stats()
            plan.set_variant(38)  # the general kernel (never screened)
            ref = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
            for v in (8, 16):
                for a, b in zip(res[v], ref):
                    assert np.array_equal(a, b), v
            res0 = res[8]
        else:
            before = plan.screen_stats()
            assert before["steps"] == 0
            res = _run_variants(plan, Np, Niter, seed)
            st = plan.screen_stats()
            for v in (5, 1):
                for a, b in zip(res[0], res[v]):
                    assert np.array_equal(a, b), v
            res0 = res[0]
        if levels is not None:
            assert st["levels"] == levels, st
        assert st["levels"] >= 1 and st["steps"] > 0
        assert st["steps"] % (len(Ns) * (Niter + 1)) == 0
        assert st["repeats"] <= 0.08 * st["steps"], st
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    if o:
        u, n = kdehip.philox_streams(seed, 0, Np, K, R)
        op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
        assert np.array_equal(res0[1], oi) and np.array_equal(res0[2], ol)
        assert np.allclose(res0[0], op, rtol=1e-11, atol=1e-11)


def test_screen_statistics_of_config_3():
    """Every draw on the two screened levels goes through the screen; about 1 % of them are repeated in fp64."""
    g, _ = _trees(3, 6, [1000] * 4, want_oracle=False)
    Np, Niter = 2048, 10
    with kdehip.ProductPlan(g) as plan:
        plan.sample(Np, Niter=Niter, seed=5)
        st = plan.screen_stats()
        assert st["levels"] == 2
        assert st["steps"] == Np * 2 * 4 * (Niter + 1)
        assert 0 < st["repeats"] < 0.04 * st["steps"], st
        plan.set_variant(5)
        plan.sample(Np, Niter=Niter, seed=5)
        assert plan.screen_stats()["steps"] == st["steps"]  # (variant 5 draws nothing through the screen)


@pytest.mark.parametrize("shift,scale,why", [
    (1.0e6, 1.0, "far from the origin: the tiles are centred, the screen still applies"),
    (0.0, 3.0e4, "coordinates beyond 2^16 after centring: outside the bound's range, fp64 throughout"),
    (0.0, 1.0e-3, "variances below 2^-7: outside the bound's range, fp64 throughout"),
    (5.0e3, 5.0, "wide data, large bandwidths"),
])
def test_screen_range_rules(shift, scale, why):
    D, Ns, Np, Niter = 6, [1000, 900, 1000], 256, 3
    g, o = _trees(77, D, Ns, shift=shift, scale=scale)
    with kdehip.ProductPlan(g) as plan:
        res = _run_variants(plan, Np, Niter, 13)
        st = plan.screen_stats()
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for v in (5, 1):
        for a, b in zip(res[0], res[v]):
            assert np.array_equal(a, b), (v, why)
    assert st["levels"] >= 1 and st["steps"] > 0, st
    if scale in (3.0e4, 1.0e-3):
        assert st["repeats"] == st["steps"], (st, why)   # never certified: every step ran in fp64
    else:
        assert st["repeats"] < 0.2 * st["steps"], (st, why)
    u, n = kdehip.philox_streams(13, 0, Np, K, R)
    op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
    assert np.array_equal(res[0][1], oi) and np.array_equal(res[0][2], ol)


@pytest.mark.parametrize("scale,why", [
    (1.0, "ordinary data: ~1 % of the draws post a request, mostly one per workgroup-step"),
    (3.0e4, "tiles outside the bound's range: EVERY wavefront posts at EVERY step -- four rounds of two requests, all 8 slots"),
])
def test_cooperative_repeat_on_chunked_levels(scale, why):
    """The fp64 repeat of a chunked screen level in the 8-chain builds is cooperative (gibbs_lean.hip kCoop: four wavefronts
    take the four LaneAcc row classes of a request through an LDS exchange area, two requests at a time).  Each class is one
    sequential sum, so labels and points must be the unscreened run's (variant 5), bit for bit, and the oracle's."""
    D, Ns, Np, Niter = 6, [4096, 5000, 4096, 4500], 72, 2   # 9 workgroups of 8 chains; level 12-13 chunked, up to 79 rows per lane
    g, o = _trees(808, D, Ns, scale=scale)
    with kdehip.ProductPlan(g) as plan:
        plan.set_variant(8)   # eight chains per workgroup whatever the chain count: the builds that have the exchange area
        assert plan.kernel_name(Np) == "gibbs_lean_kernel" and plan.launch_geometry(Np)["waves"] == 8
        res = _run_variants(plan, Np, Niter, 31, variants=(8, 5))
        res[0] = res[8]
        plan.set_variant(8)
        plan.sample(Np, Niter=Niter, seed=31)
        st = plan.screen_stats()
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for a, b in zip(res[0], res[5]):
        assert np.array_equal(a, b), why
    assert st["steps"] > 0 and st["repeats"] > 0, st
    if scale != 1.0:
        assert st["repeats"] == st["steps"], (st, why)
    u, n = kdehip.philox_streams(31, 0, Np, K, R)
    op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
    assert np.array_equal(res[0][1], oi) and np.array_equal(res[0][2], ol), why


def test_screen_linearisation_guard():
    """Clusters thousands of bandwidths apart (coordinates ~ +-4000, sigma 0.2): the tiles pass the range checks (|m'| <=
    2^16, variances >= 2^-7) but the centring term na = sqrt(c0) u |g / sigma| is ~4e-3, where the bound's linearised
    exp(d) - 1 ~ d no longer holds.  The step must run in fp64 there (kScreenMaxNa): every screened step
    of such a level is a repeat, and the results are the unscreened run's and the oracle's."""
    D, Ns, Np, Niter = 6, [1000, 900, 1000], 192, 3
    rng = np.random.default_rng(4242)
    g, o = [], []
    for N in Ns:
        pts = synth_mixture(rng, D, N) + rng.choice([-4000.0, 4000.0], size=(D, 1)) * (rng.uniform(size=(D, N)) < 0.5)
        g.append(kdehip.kde(pts, [0.2]))
        o.append(oracle.OracleDensity(pts, [0.2]))
    with kdehip.ProductPlan(g) as plan:
        res = _run_variants(plan, Np, Niter, 29)
        st = plan.screen_stats()
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    for v in (5, 1):
        for a, b in zip(res[0], res[v]):
            assert np.array_equal(a, b), v
    assert st["levels"] >= 1 and st["steps"] > 0, st
    assert st["repeats"] == st["steps"], st
    u, n = kdehip.philox_streams(29, 0, Np, K, R)
    op, oi, ol = oracle.gibbs1(o, Np, Niter, u, n, want_labels=True)
    assert np.array_equal(res[0][1], oi) and np.array_equal(res[0][2], ol)


def test_screen_through_every_entry_point():
    """One-shot calls, caller streams, resident densities and the batched entry give the screened plan's result."""
    D, Ns, Np, Niter, seed = 6, [1000] * 4, 160, 4, 321
    g, _ = _trees(55, D, Ns, want_oracle=False)
    with kdehip.ProductPlan(g) as plan:
        plan.set_variant(5)
        ref = plan.sample(Np, Niter=Niter, seed=seed)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    pts, ind = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed)
    assert np.array_equal(pts, ref[0]) and np.array_equal(ind, ref[1])
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    pts, ind = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, randU=u, randN=n)
    # (caller streams: the host twin's normals may differ from the device's in the last bit -- libm against the device's
    # logarithm and sine / cosine -- so the points agree to an ulp, the labels exactly)
    assert np.array_equal(ind, ref[1]) and np.allclose(pts, ref[0], rtol=0.0, atol=1e-12)
    dd = [kdehip.DeviceDensity(t) for t in g]
    try:
        pts, ind = kdehip.prodAppxMSGibbsS_resident(dd, Np=Np, Niter=Niter, seed=seed)
        assert np.array_equal(pts, ref[0]) and np.array_equal(ind, ref[1])
    finally:
        for d in dd:
            d.close()
