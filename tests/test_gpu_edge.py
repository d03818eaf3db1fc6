"""Edge cases and size-independent properties of the HIP product path at and beyond the BASELINE sizes."""
import numpy as np
import pytest
import torch  # before the first HIP call of libkdehip: torch brings its own HIP runtime build

import kdehip
from oracle import oracle
from tests.test_gpu_parity import _closed_form_points, _compare, _make_inputs, _pair

pytestmark = pytest.mark.gpu


def test_maximum_density_count_and_dimension():
    """16 densities (KDEHIP_MAX_DENS) in 8 dimensions (KDEHIP_MAX_DIMS)."""
    D, M, N, Np, Niter = 8, 16, 40, 24, 1
    gp, op = _make_inputs(5, D, M, N)
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    rng = np.random.default_rng(1)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    _compare(g, oracle.gibbs1(op, Np, Niter, randU, randN))


def test_all_single_point_densities_and_single_density_product():
    a, ao = _pair(np.array([[0.3], [1.0]]), [0.5])
    b, bo = _pair(np.array([[-0.2], [2.0]]), [0.25])
    rng = np.random.default_rng(2)
    Np, Niter = 16, 2
    K, R, nU, nN = oracle.rng_sizes(2, 2, Np, Niter, [1, 1])
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    _compare(g, oracle.gibbs1([ao, bo], Np, Niter, randU, randN))
    assert np.all(g[1] == 2)  # the only point of each density
    # a "product" of one density is resampling it
    c, co = _pair(rng.standard_normal((2, 50)), [0.3])
    K, R, nU, nN = oracle.rng_sizes(1, 2, Np, Niter, [50])
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, [c], None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    _compare(g, oracle.gibbs1([co], Np, Niter, randU, randN))


def test_duplicate_points_and_extreme_weights():
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((2, 64))
    pts[:, 10:30] = pts[:, [5]]                     # 20 identical points (ties in the tree build)
    w = np.ones(64)
    w[3], w[40] = 1e-12, 1e3                         # nearly-zero and dominating weights
    a, ao = _pair(pts, [0.2, 0.4], w)
    b, bo = _pair(rng.standard_normal((2, 64)) * 0.5, [0.3])
    Np, Niter = 200, 3
    K, R, nU, nN = oracle.rng_sizes(2, 2, Np, Niter, [64, 64])
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, [a, b], None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    _compare(g, oracle.gibbs1([ao, bo], Np, Niter, randU, randN))


def test_many_chains_properties_and_batch_invariance():
    """100k chains in one call: label range, closed-form invariant, and equality with the same chains
    drawn in several smaller calls (Philox keyed by the global sample index)."""
    D, M, N, Np, Niter, seed = 3, 3, 300, 100_000, 2, 77
    gp, _ = _make_inputs(9, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        p, i = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=False)
        parts = [plan.sample(n, Niter=Niter, seed=seed, sample_offset=o, addEntropy=False)
                 for o, n in ((0, 1), (1, 4095), (4096, 50_000), (54_096, Np - 54_096))]
    assert i.min() >= 2 and i.max() <= N + 1
    assert np.allclose(p, _closed_form_points(gp, i), rtol=1e-11, atol=1e-12)
    assert np.array_equal(np.concatenate([q[0] for q in parts], axis=1), p)
    assert np.array_equal(np.concatenate([q[1] for q in parts], axis=1), i)
    # every leaf of every density is reachable
    assert all(len(np.unique(i[j])) > 0.9 * N for j in range(M))


def test_config4_full_batch_properties():
    """BASELINE config 4 at its full per-GPU batch (2048 chains of the 16384, 8 x 5000 points)."""
    D, M, N, Np, Niter, seed = 3, 8, 5000, 2048, 10, 4
    gp, _ = _make_inputs(44, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        pe, ie = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=True)
        pn, i_n = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=False)
    assert np.array_equal(ie, i_n) and ie.min() >= 2 and ie.max() <= N + 1
    assert np.allclose(pn, _closed_form_points(gp, i_n), rtol=1e-11, atol=1e-12)
    assert np.isfinite(pe).all()


def test_config4_total_chain_count_on_one_gpu():
    """BASELINE config 4 as stated -- 3-D, 8 densities x 5000 points, Nout = 16384, Niter = 10 -- all chains on one GPU
    (what `bench.py --strong --config c4 --gpus 1` times): properties over the whole batch, oracle parity on the
    first chains, and the same chains drawn as eight shards (the 8-GPU split of the config) bit for bit."""
    D, M, N, Np, Niter, seed = 3, 8, 5000, 16384, 10, 4
    gp, op = _make_inputs(44, D, M, N)
    with kdehip.ProductPlan(gp) as plan:
        pn, i_n = plan.sample(Np, Niter=Niter, seed=seed, addEntropy=False)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        shards = [plan.sample(Np // 8, Niter=Niter, seed=seed, sample_offset=g * (Np // 8), addEntropy=False)
                  for g in range(8)]
    assert i_n.min() >= 2 and i_n.max() <= N + 1
    assert np.allclose(pn, _closed_form_points(gp, i_n), rtol=1e-11, atol=1e-12)
    assert np.array_equal(np.concatenate([s[0] for s in shards], axis=1), pn)
    assert np.array_equal(np.concatenate([s[1] for s in shards], axis=1), i_n)
    ns = 12
    u, n = kdehip.philox_streams(seed, 0, ns, K, R)
    o_pts, o_ind = oracle.gibbs1(op, ns, Niter, u, n, addEntropy=False)
    assert np.array_equal(i_n[:, :ns], o_ind) and np.allclose(pn[:, :ns], o_pts, rtol=1e-11, atol=1e-11)


def test_zero_chains_and_argument_errors():
    gp, _ = _make_inputs(1, 2, 2, 20)
    with kdehip.ProductPlan(gp) as plan:
        p, i = plan.sample(0)
        assert p.shape == (2, 0) and i.shape == (2, 0)
        with pytest.raises(kdehip.KdeHipError):
            plan.sample(4, Niter=-1)
    with pytest.raises(kdehip.KdeHipError):
        kdehip.ProductPlan(gp, precision=16)
    with pytest.raises(kdehip.KdeHipError):
        kdehip.ProductPlan(gp, device=99)


def test_runs_can_be_captured_in_a_hip_graph():
    """After the first run (which builds the conditional tables) a run only enqueues one kernel: it can be
    captured into a HIP graph and replayed, e.g. to batch many small products without launch overhead."""
    D, M, N, Np, Niter, seed = 2, 3, 200, 256, 5, 3
    gp, _ = _make_inputs(12, D, M, N)
    dev = torch.device("cuda", 0)
    with kdehip.ProductPlan(gp) as plan:
        ref_p, ref_i = plan.sample(Np, Niter=Niter, seed=seed)          # first run: tables built here
        P = torch.zeros(Np * D, dtype=torch.float64, device=dev)
        I = torch.zeros(Np * M, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            plan.sample_philox_device(Np, Niter, seed, 0, True, P, I, None, side.cuda_stream)  # warm-up on this stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            plan.sample_philox_device(Np, Niter, seed, 0, True, P, I, None, torch.cuda.current_stream(dev).cuda_stream)
        P.zero_()
        I.zero_()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, ref_p)
        assert np.array_equal(I.cpu().numpy().reshape(Np, M).T, ref_i)


def test_concurrent_calls_from_host_threads():
    """SURVEY 8b "Threading": the blocking entry points are safe to call concurrently.  Eight host threads
    run different one-shot products through kdehip_gibbs1 while four more share ONE resident plan;
    every result must equal the oracle / the single-threaded result."""
    from concurrent.futures import ThreadPoolExecutor

    cases = []
    for t in range(8):
        D, M, N = 1 + t % 4, 2 + t % 3, 30 + 37 * t
        gp, op = _make_inputs(100 + t, D, M, N)
        Np, Niter = 64 + 16 * t, 2 + t % 3
        K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
        rng = np.random.default_rng(t)
        cases.append((gp, op, Np, Niter, rng.random(nU), rng.standard_normal(nN)))

    def one_shot(c):
        gp, op, Np, Niter, randU, randN = c
        out = None
        for _ in range(5):
            out = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
        return out

    gp, _ = _make_inputs(9, 3, 4, 500)
    plan = kdehip.ProductPlan(gp)
    want = plan.sample(4096, 3, seed=5)

    def shared_plan(i):
        return [plan.sample(4096, 3, seed=5) for _ in range(5)][-1]

    with ThreadPoolExecutor(max_workers=12) as ex:
        f1 = [ex.submit(one_shot, c) for c in cases]
        f2 = [ex.submit(shared_plan, i) for i in range(4)]
        r1 = [f.result() for f in f1]
        r2 = [f.result() for f in f2]
    for c, g in zip(cases, r1):
        _compare(g, oracle.gibbs1(c[1], c[2], c[3], c[4], c[5]))
    for g in r2:
        assert np.array_equal(g[0], want[0]) and np.array_equal(g[1], want[1])


def test_very_large_ragged_densities():
    """150 000- and 70 001-point densities (frontiers of up to 2344 rows per lane: chunked LDS streaming, two
    rounds of second-pass narrowing) against a 3-point one; labels must still match the oracle exactly."""
    rng = np.random.default_rng(11)
    sizes = [150_000, 70_001, 3]
    gp, op = [], []
    for k, N in enumerate(sizes):
        pts = rng.standard_normal((2, N)) * (1.0 + 0.3 * k) + 0.2 * k
        g, o = _pair(pts, [0.05 + 0.02 * k, 0.08])
        gp.append(g)
        op.append(o)
    Np, Niter = 24, 1
    K, R, nU, nN = oracle.rng_sizes(3, 2, Np, Niter, sizes)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN)
    _compare(g, oracle.gibbs1(op, Np, Niter, randU, randN, nthreads=8))


def test_allocation_cache_reuse_and_clear():
    """One-shot calls recycle device blocks through the library's cache; results are unaffected, interleaved
    sizes do not alias, and kdehip_clear_cache() is harmless between calls."""
    from kdehip import _lib
    rng = np.random.default_rng(21)
    cases = []
    for D, M, N, Np in [(2, 2, 50, 40), (3, 3, 300, 500), (2, 2, 50, 40), (4, 2, 1000, 100)]:
        gp, op = _make_inputs(200 + N, D, M, N)
        K, R, nU, nN = oracle.rng_sizes(M, D, Np, 2, [N] * M)
        cases.append((gp, op, Np, rng.random(nU), rng.standard_normal(nN)))
    for rep in range(3):
        for gp, op, Np, randU, randN in cases:
            g = kdehip.prodAppxMSGibbsS(None, gp, None, None, Niter=2, Np=Np, randU=randU, randN=randN)
            _compare(g, oracle.gibbs1(op, Np, 2, randU, randN))
            p = kdehip.evaluateDualTree(gp[0], kdehip.getPoints(gp[1])[:, :17])
            assert np.allclose(p, oracle.eval_direct(op[0], kdehip.getPoints(gp[1])[:, :17]), rtol=1e-11)
        if rep == 1:
            _lib.lib.kdehip_clear_cache()


@pytest.mark.parametrize("N", [300, 6000])   # packed by the calling thread alone / on the host pool's threads
def test_malformed_tree_is_refused_with_its_message(N):
    """A child array that makes a frontier outgrow its density (here: the root is its own two children) must be refused
    by the packer -- with the message in the CALLING thread's error slot, also when the frontiers of a large product are
    expanded on the library's worker threads (csrc/pack_levels.cpp, csrc/host_pool.hpp)."""
    rng = np.random.default_rng(3)
    good = [kdehip.kde(rng.standard_normal((2, N)), np.full(2, 0.3)) for _ in range(3)]
    bad = kdehip.kde(rng.standard_normal((2, N)), np.full(2, 0.3))
    bad.bt.left_child[0] = 1
    bad.bt.right_child[0] = 1
    for order in ([good[0], bad, good[1]], [bad, good[0], good[1]], [good[0], good[1], good[2], bad]):
        with pytest.raises(kdehip.KdeHipError, match="malformed tree"):
            kdehip.ProductPlan(order)
    with kdehip.ProductPlan(good) as plan:   # (and the pool is none the worse for it)
        p, i = plan.sample(64, Niter=1, seed=2)
        assert np.isfinite(p).all()


@pytest.mark.parametrize("offset,expect_compact", [(0.0, True), (3.0e3, True), (4.0e7, False)])
def test_shared_bandwidth_evaluator_far_from_the_origin(offset, expect_compact):
    """The fp64 shared-bandwidth evaluator forms (m - mu) s as fma(m, s, -mu s): two instructions per dimension, and a
    rounding error that grows with |m| s (csrc/gibbs_device.hpp EvalUniform).  The packer therefore gives a leaf frontier
    the compact shared-bandwidth tile only while |m| / sqrt(2 bandwidth) <= 1e5 (csrc/pack_levels.cpp) and keeps the
    per-node form, which subtracts first, beyond: data thousands of bandwidths away from the origin still give the
    oracle's labels either way."""
    rng = np.random.default_rng(17)
    D, M, N, Np, Niter = 3, 3, 700, 200, 3
    g, o = [], []
    for j in range(M):
        pts = rng.standard_normal((D, N)) * 0.8 + rng.uniform(-1, 1, size=(D, 1)) + offset
        ks = np.full(D, 0.25)
        a, b = _pair(pts, ks)
        g.append(a)
        o.append(b)
    with kdehip.ProductPlan(g) as plan:
        # (compact leaf tiles carry D + 1 fields per node instead of 2 D + 1: the plan's size tells which it got)
        size = plan.packed_bytes
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
        randU, randN = kdehip.philox_streams(5, 0, Np, K, R)
        got = plan.sample(Np, Niter=Niter, seed=5)
    g0 = [kdehip.kde(t_pts, [0.25]) for t_pts in [kdehip.getPoints(t) - offset for t in g]]
    with kdehip.ProductPlan(g0) as plan0:
        size0 = plan0.packed_bytes
    assert (size == size0) == expect_compact, (size, size0)
    ref = oracle.gibbs1(o, Np, Niter, randU, randN)
    assert np.array_equal(got[1], ref[1])
    assert np.allclose(got[0], ref[0], rtol=1e-9, atol=1e-6)
