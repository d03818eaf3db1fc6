"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the
header declares, the product's own tree builder (csrc/balltree.cpp) reproduces the reference's
golden files and agrees bit-for-bit with the oracle, the Philox host twin is Philox4x32-10, and
argument errors surface like the reference's.  No compute entry point is called (no GPU here)."""
import os
import re

import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import check_density_against_golden, parse_mat_print_kde

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Flat:
    """Adapter: product BallTreeDensity -> the flat attribute names the golden checker uses."""

    def __init__(self, bd):
        bt = bd.bt
        self.dims, self.num_points = bt.dims, bt.num_points
        for k in ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf",
                  "highest_leaf", "permutation"):
            setattr(self, k, getattr(bt, k))
        for k in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
            setattr(self, k, getattr(bd, k))


def test_library_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "kdehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kdehip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 16
    lib = ctypes.CDLL(kdehip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libkdehip.so does not export {name}"
    from importlib import import_module
    bound = set(import_module("kdehip._lib").SIGNATURES)
    assert declared == bound, declared ^ bound
    assert kdehip.version() == 600


def test_ctypes_signatures_match_the_header():
    """The Python mirror's ctypes prototypes against include/kdehip.h: same arity, integer widths and pointer-ness
    position by position (a 32-bit integer bound where the header says int64_t is the classic silent FFI fault)."""
    import ctypes as C
    from importlib import import_module
    from tests.test_julia_shim_syntax import header_params
    sigs = import_module("kdehip._lib").SIGNATURES
    ints = {"int": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "int32_t": C.c_int32, "uint32_t": C.c_uint32}
    for name, params in header_params().items():
        res, args = sigs[name]
        assert len(args) == len(params), f"{name}: ctypes binds {len(args)} arguments, the header declares {len(params)}"
        for pos, (a, ct) in enumerate(zip(args, params)):
            if ct.endswith("*"):
                is_ptr = a is C.c_void_p or a is C.c_char_p or hasattr(a, "_type_") and hasattr(a, "contents")
                assert is_ptr, f"{name}: argument {pos} is {ct} in the header but bound as {a}"
            else:
                assert ct in ints and a is ints[ct], f"{name}: argument {pos} is {ct} in the header but bound as {a}"


def test_product_tree_builder_matches_reference_goldens(golden_dir):
    d = kdehip.kde([0.1, 0.45, 0.55, 3.8], [0.08])
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test1DResult.txt")), 1e-5)
    d = kdehip.kde(np.array([[0.5172, 0.7169, 0.4049], [0.0312, 1.0094, 2.0204]]), [0.1])
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test2DResult.txt")), 1e-5)
    d = kdehip.kde(np.array([[0.5172, 7.169, 4.049], [0.0312, 10.0094, -2.0204]]), [0.1, 1.0])
    check_density_against_golden(_Flat(d), parse_mat_print_kde(os.path.join(golden_dir, "test2DvarResult.txt")), 1e-4)
    gold = parse_mat_print_kde(os.path.join(golden_dir, "test1Dlcv100Result.txt"))
    x = np.loadtxt(os.path.join(golden_dir, "test1Dlcv100.txt")).ravel()
    d = kdehip.kde(x, [np.sqrt(gold["bandwidth"][100])])
    check_density_against_golden(_Flat(d), gold, 1e-4)
    # 2-D, 100 points (goldens of the reference's disabled UnitTest2Dlcv01 / UnitTest2Dvarlcv01, runtests.jl:131-141,
    # :155-165): bandwidth taken from the golden, tree + statistics reproduced at the reference's own tolerances
    for name, tol in (("test2Dlcv100", 1e-4), ("test2Dvarlcv100", 2e-3)):
        gold = parse_mat_print_kde(os.path.join(golden_dir, name + "Result.txt"))
        pts = np.ascontiguousarray(np.loadtxt(os.path.join(golden_dir, name + ".txt")).T)
        d = kdehip.kde(pts, np.sqrt(gold["bandwidth"][200:202]))
        check_density_against_golden(_Flat(d), gold, tol)


@pytest.mark.parametrize("D,N,weighted", [(1, 1, False), (1, 2, False), (1, 7, True), (2, 33, False),
                                          (3, 200, True), (6, 1000, False), (4, 513, True),
                                          # 512 leaves and up: the top levels build their left side on pool threads
                                          (2, 1024, False), (3, 2049, True), (2, 9001, False)])
def test_product_tree_builder_equals_oracle_bitwise(D, N, weighted):
    rng = np.random.default_rng(100 * D + N)
    pts = rng.standard_normal((D, N))
    pts[:, N // 3] = pts[:, 0]  # duplicate point: exercises ties in the quick-select
    ks = rng.uniform(0.05, 0.6, size=D)
    w = rng.uniform(0.1, 1.0, size=N) if weighted else None
    a = kdehip.kde(pts, ks, w)
    b = oracle.OracleDensity(pts, ks, w)
    for k in ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf",
              "permutation"):
        assert np.array_equal(getattr(a.bt, k), getattr(b, k)), k
    for k in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert np.allclose(kdehip.getPoints(a), pts)
    assert np.allclose(kdehip.getBW(a), np.repeat(ks[:, None], N, axis=1))
    ww = np.ones(N) if w is None else w
    assert np.allclose(kdehip.getWeights(a), ww / ww.sum())


def test_tree_builder_from_concurrent_host_threads():
    """several callers share the builder's worker pool (csrc/host_pool.hpp); an owner runs what no worker has started"""
    import threading
    rng = np.random.default_rng(77)
    cases = [(rng.standard_normal((3, N)), rng.uniform(0.1, 0.5, size=3)) for N in (700, 2048, 3000, 5000, 1500, 4097)]
    want = [kdehip.kde(p, k) for p, k in cases]
    got = [None] * (4 * len(cases))

    def run(i):
        p, k = cases[i % len(cases)]
        got[i] = kdehip.kde(p, k)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(got))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i, g in enumerate(got):
        w = want[i % len(cases)]
        for k in ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation"):
            assert np.array_equal(getattr(g.bt, k), getattr(w.bt, k)), (i, k)
        assert np.array_equal(g.means, w.means) and np.array_equal(g.bandwidth, w.bandwidth)


def test_tree_builder_in_a_forked_child():
    """a fork()ed child has none of the pool's workers: it must build (serially) the same tree, not wait for them"""
    import os
    rng = np.random.default_rng(5)
    pts, ks = rng.standard_normal((3, 4000)), np.full(3, 0.2)
    a = kdehip.kde(pts, ks)   # (starts the workers in this process)
    pid = os.fork()
    if pid == 0:
        try:
            b = kdehip.kde(pts, ks)
            ok = np.array_equal(a.means, b.means) and np.array_equal(a.bt.permutation, b.bt.permutation)
        except BaseException:
            ok = False
        os._exit(0 if ok else 1)
    import time
    deadline = time.time() + 60
    while time.time() < deadline:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
            return
        time.sleep(0.02)
    os.kill(pid, 9)
    os.waitpid(pid, 0)
    raise AssertionError("the forked child did not finish its tree build")


def test_kde_argument_forms():
    # vector input -> 1-D density (src/KDE01.jl:78-84); scalar ks repeated over dims (:41-43)
    a = kdehip.kde([1.0, 2.0, 3.0], [0.5])
    assert (kdehip.Ndim(a), kdehip.Npts(a)) == (1, 3)
    b = kdehip.kde(np.arange(8.0).reshape(2, 4), [0.3])
    assert np.allclose(b.bandwidth[2 * 4:], 0.09)
    with pytest.raises(ValueError):
        kdehip.kde(np.zeros((2, 4)), [0.1, 0.2, 0.3])
    with pytest.raises(ValueError):
        kdehip.kde(np.zeros((2, 4)), [0.1], weights=[1.0, 2.0])


# ---- Philox ---------------------------------------------------------------------------------------

def _philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = list(ctr)
    k = list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def test_philox_known_answers_and_host_twin():
    # Random123 known-answer vectors for philox4x32-10
    assert _philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]

    seed, s0, ns, K, R = 0x0123456789ABCDEF, 5, 3, 9, 7
    u, n = kdehip.philox_streams(seed, s0, ns, K, R)
    key = [seed & 0xFFFFFFFF, seed >> 32]

    def unit(lo, hi):
        return ((((hi << 32) | lo) >> 11) + 0.5) / 2.0 ** 53

    for s in range(ns):
        for i in range(K):
            c = i + 1  # slot i is consumed by select call c = i+1
            v = _philox4x32_10([s0 + s, 0, c >> 1, 0], key)
            exp = unit(v[2], v[3]) if c & 1 else unit(v[0], v[1])
            assert u[s * K + i] == exp
        for r in range(R):
            v = _philox4x32_10([s0 + s, 0, r >> 1, 1], key)
            u1, u2 = unit(v[0], v[1]), unit(v[2], v[3])
            rad, ang = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
            exp = rad * np.sin(ang) if r & 1 else rad * np.cos(ang)
            assert abs(n[s * R + r] - exp) <= 1e-14 * max(1.0, abs(exp))
    big_u, big_n = kdehip.philox_streams(7, 0, 2000, 50, 50)
    assert 0.0 < big_u.min() and big_u.max() < 1.0
    assert abs(big_u.mean() - 0.5) < 0.005 and abs(big_n.mean()) < 0.01 and abs(big_n.std() - 1.0) < 0.01


# ---- argument / error behaviour (validated on the host before any device work) -------------------

def test_dimension_mismatch_is_an_error():
    a = kdehip.kde(np.zeros((2, 5)) + np.arange(5.0), [0.1])
    b = kdehip.kde(np.arange(5.0), [0.1])
    with pytest.raises(ValueError, match="same dimension"):  # src/MSGibbs01.jl:720-722
        kdehip.ProductPlan([a, b], ndims=2)


def test_limits_are_reported():
    a = kdehip.kde(np.arange(5.0), [0.1])
    with pytest.raises(kdehip.KdeHipError) as ei:
        kdehip.ProductPlan([a] * 17)
    assert ei.value.code == -7
    with pytest.raises(NotImplementedError):
        kdehip.prodAppxMSGibbsS(a, [a, a], None, None, addop=(lambda x, y: x + y,))


def test_no_device_fails_loudly_instead_of_falling_back():
    if kdehip.device_count() > 0:
        pytest.skip("a GPU is present")
    a = kdehip.kde(np.arange(5.0), [0.1])
    with pytest.raises(kdehip.KdeHipError) as ei:
        kdehip.prodAppxMSGibbsS(a, [a, a], None, None, seed=1)
    assert ei.value.code == -4


def test_nlevels_formula():
    for n, L in [(1, 1), (2, 2), (3, 2), (4, 3), (100, 7), (200, 8), (1000, 10), (1024, 11), (5000, 13), (10000, 14)]:
        assert kdehip.nlevels(n) == L == oracle.nlevels(n)


def test_prod_front_end_keyword_surface():
    """The reference's keyword list (src/MSGibbs01.jl:645-664) is accepted; argument errors are raised before any
    device work (so this runs without a GPU)."""
    import inspect
    names = set(inspect.signature(kdehip.prodAppxMSGibbsS).parameters)
    for kw in ("Niter", "addop", "diffop", "getMu", "getLambda", "glbs", "addEntropy", "ndims", "Ndens", "Np", "maxNp",
               "Nlevels", "randU", "randN", "partialDimMask"):
        assert kw in names, kw
    p = kdehip.kde(np.array([[0.0, 1.0, 2.0]]), [0.5])
    with pytest.raises(NotImplementedError):
        kdehip.prodAppxMSGibbsS(p, [p, p], None, None, addop=(lambda a, b: a + b,))
    with pytest.raises(ValueError):
        kdehip.prodAppxMSGibbsS(p, [p, p], None, None, randU=np.zeros(10))
    with pytest.raises(TypeError):
        kdehip.prodAppxMSGibbsS(p, [p, p], None, None, 3, 4)


def test_set_bandwidth_equals_building_with_that_bandwidth():
    """kdehip_density_set_bandwidth recomputes the bandwidth-dependent statistics of an existing tree (leaf variances,
    moment-matched variances of the internal nodes, src/BallTreeDensity01.jl:141-187): bit-identical to
    kdehip_make_density with that bandwidth -- what lets kde!(points) build its tree while the GPU searches the bandwidth."""
    from kdehip import _lib
    from kdehip._lib import f64p, i64p, ptr
    rng = np.random.default_rng(0)
    for D, N, weighted in [(1, 1, False), (2, 2, False), (1, 3, False), (3, 37, True), (2, 100, True), (6, 1500, False)]:
        pts = rng.standard_normal((D, N))
        pts[:, : N // 3] = pts[:, N // 3: 2 * (N // 3)]          # ties
        w = rng.uniform(0.1, 1.0, N) if weighted else None
        bw = rng.uniform(0.1, 0.9, D)
        ref = kdehip.kde(pts, bw, w)
        bd = kdehip.kde(pts, np.ones(D), w)
        bt = bd.bt
        rc = _lib.lib.kdehip_density_set_bandwidth(D, N, ptr(bw, f64p), bw.size, ptr(bt.weights, f64p), ptr(bt.left_child, i64p),
                                                   ptr(bt.right_child, i64p), ptr(bd.means, f64p), ptr(bd.bandwidth, f64p),
                                                   ptr(bd.bandwidthMin, f64p), ptr(bd.bandwidthMax, f64p))
        assert rc == 0
        for f in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
            assert np.array_equal(getattr(bd, f), getattr(ref, f)), (D, N, f)
        for f in ("centers", "ranges", "weights", "left_child", "right_child", "permutation", "lowest_leaf", "highest_leaf"):
            assert np.array_equal(getattr(bd.bt, f), getattr(ref.bt, f)), (D, N, f)
