"""Multi-GPU entry points of the C ABI (one process, one plan per device; include/kdehip.h sections 1 and 2b).
The GPU boxes of the test pool have ONE device: the one-device degenerate path must reproduce the single-GPU entry
points bit for bit, a device range beyond the visible devices must be refused, and where a second device exists the
two-device results must equal the one-device results (chains are independent; Philox counters use the global index)."""
import numpy as np
import pytest

import kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


def _trees(seed, D, M, N):
    rng = np.random.default_rng(seed)
    g, o = [], []
    for _ in range(M):
        pts = synth_mixture(rng, D, N)
        ks = silverman_bw(pts)
        g.append(kdehip.kde(pts, ks))
        o.append(oracle.OracleDensity(pts, ks))
    return g, o


def test_one_shot_philox_equals_resident_plan_and_oracle():
    D, M, N, Np, Niter, seed = 3, 3, 300, 257, 4, 99
    g, o = _trees(1, D, M, N)
    with kdehip.ProductPlan(g) as plan:
        ref = plan.sample(Np, Niter=Niter, seed=seed, want_labels=True)
        K, R = plan.randu_per_sample(Niter), plan.randn_per_sample()
    glbs = kdehip.makeEmptyGbGlb(recordChoosen=True)
    pts, ind = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, glbs=glbs, ngpus=1)
    assert np.array_equal(pts, ref[0]) and np.array_equal(ind, ref[1])
    L = ref[2].shape[2]
    assert [glbs.labelsChoosen[5][2][l + 1] for l in range(L)] == list(ref[2][4, 1])
    u, n = kdehip.philox_streams(seed, 0, Np, K, R)
    op, oi = oracle.gibbs1(o, Np, Niter, u, n)
    assert np.array_equal(ind, oi) and np.allclose(pts, op, rtol=1e-11, atol=1e-11)
    # the drop-in with the same numbers as caller streams, through the multi entry with one device
    p2 = np.zeros(D * Np)
    i2 = np.ones((M, Np), dtype=np.int64)
    kdehip.gibbs1(M, g, Np, Niter, p2, i2, u, n, ngpus=1)
    assert np.array_equal(i2, oi) and np.allclose(p2.reshape(Np, D).T, op, rtol=1e-11, atol=1e-11)
    # fp32 one-shot = fp32 resident plan
    with kdehip.ProductPlan(g, precision=32) as p32:
        r32 = p32.sample(Np, Niter=Niter, seed=seed)
    q32 = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, precision=32)
    assert np.array_equal(q32[0], r32[0]) and np.array_equal(q32[1], r32[1])


def test_resident_multi_plan_with_one_device_equals_single_plan():
    import torch
    D, M, N, Np, Niter, seed = 6, 4, 500, 300, 3, 5
    g, _ = _trees(2, D, M, N)
    dev = torch.device("cuda", 0)
    with kdehip.ProductPlan(g) as plan:
        ref = plan.sample(Np, Niter=Niter, seed=seed, sample_offset=40)
    with kdehip.MultiProductPlan(g, first_device=0, ngpus=1) as mp:
        assert mp.ngpus == 1
        P = torch.zeros(D * Np, dtype=torch.float64, device=dev)
        I = torch.zeros(M * Np, dtype=torch.int64, device=dev)
        st = torch.cuda.Stream(device=dev)
        mp.sample_philox_device(Np, Niter, seed, 40, True, [P], [I], [st.cuda_stream])
        st.synchronize()
        assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, ref[0])
        assert np.array_equal(I.cpu().numpy().reshape(Np, M).T, ref[1])
        mp.sample_philox_device(Np, Niter, seed, 40, True, [P], [I])   # null streams
        torch.cuda.synchronize()
        assert np.array_equal(I.cpu().numpy().reshape(Np, M).T, ref[1])


def test_device_range_beyond_the_visible_devices_is_refused():
    g, _ = _trees(3, 2, 2, 50)
    n = kdehip.device_count()
    with pytest.raises(kdehip.KdeHipError) as ei:
        kdehip.prodAppxMSGibbsS(None, g, None, None, Np=10, seed=1, ngpus=n + 1)
    assert ei.value.code == -1
    with pytest.raises(kdehip.KdeHipError):
        kdehip.MultiProductPlan(g, first_device=0, ngpus=n + 1)


@pytest.mark.skipif(kdehip.device_count() < 2, reason="needs two visible GPUs")
def test_two_devices_equal_one_device():
    import torch
    D, M, N, Np, Niter, seed = 3, 3, 400, 501, 3, 8
    g, _ = _trees(4, D, M, N)
    one = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, ngpus=1)
    two = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, ngpus=2)
    assert np.array_equal(one[0], two[0]) and np.array_equal(one[1], two[1])
    with kdehip.MultiProductPlan(g, first_device=0, ngpus=2) as mp:
        Ps = [torch.zeros(D * Np, dtype=torch.float64, device=torch.device("cuda", d)) for d in range(2)]
        Is = [torch.zeros(M * Np, dtype=torch.int64, device=torch.device("cuda", d)) for d in range(2)]
        mp.sample_philox_device(Np, Niter, seed, 0, True, Ps, Is)
        for d in range(2):
            torch.cuda.synchronize(d)
            assert np.array_equal(Ps[d].cpu().numpy().reshape(Np, D).T, one[0])   # every device holds everything
            assert np.array_equal(Is[d].cpu().numpy().reshape(Np, M).T, one[1])


_ALIAS_SCRIPT = r'''
import numpy as np, torch, kdehip
from oracle import oracle
from tests.helpers import silverman_bw, synth_mixture
rng = np.random.default_rng(11)
D, M, N, Np, Niter, seed = 3, 3, 400, 1001, 3, 21
pts = [synth_mixture(rng, D, N) for _ in range(M)]
g = [kdehip.kde(p, silverman_bw(p)) for p in pts]
one = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, ngpus=1)
for G in (2, 3, 8):
    many = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed, ngpus=G)
    assert np.array_equal(one[0], many[0]) and np.array_equal(one[1], many[1]), G
# caller streams + label trace over 3 logical devices
K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
randU, randN = rng.random(nU), rng.standard_normal(nN)
res = []
for G in (1, 3):
    glbs = kdehip.makeEmptyGbGlb(recordChoosen=True)
    p = np.zeros(D * Np); i = np.ones((M, Np), dtype=np.int64)
    kdehip.gibbs1(M, g, Np, Niter, p, i, randU, randN, glbs=glbs, ngpus=G)
    res.append((p.copy(), i.copy(), glbs.labelsChoosen))
assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
o = oracle.gibbs1([oracle.OracleDensity(p, silverman_bw(p)) for p in pts], Np, Niter, randU, randN)
assert np.array_equal(res[1][1], o[1])
# resident plans on 2, 3, 4 and 8 logical devices, a stream each: afterwards EVERY device array holds the complete
# result; the all-gather is fused into the kernel (no copy-engine transfers)
dev = torch.device("cuda", 0)
other = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed + 1, ngpus=1)
for G in (2, 3, 4, 8):
    with kdehip.MultiProductPlan(g, first_device=0, ngpus=G) as mp:
        assert mp.ngpus == G and mp.transfers_per_product == 0
        Ps = [torch.zeros(D * Np, dtype=torch.float64, device=dev) for _ in range(G)]
        Is = [torch.zeros(M * Np, dtype=torch.int64, device=dev) for _ in range(G)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(G)]
        for rep in range(2):
            mp.sample_philox_device(Np, Niter, seed, 0, True, Ps, Is, [s.cuda_stream for s in sts])
        # Write after read: a consumer of product 1 sits on every stream (held back by a long sleep) when product 2
        # -- another seed, other numbers -- is enqueued into the SAME arrays; the consumers must still see product 1.
        kept = []
        for k in range(G):
            with torch.cuda.stream(sts[k]):
                torch.cuda._sleep(20_000_000)   # ~10 ms
                kept.append((Ps[k].clone(), Is[k].clone()))
        mp.sample_philox_device(Np, Niter, seed + 1, 0, True, Ps, Is, [s.cuda_stream for s in sts])
        for k in range(G):
            sts[k].synchronize()
            assert np.array_equal(kept[k][0].cpu().numpy().reshape(Np, D).T, one[0]), (G, k, "consumer of product 1")
            assert np.array_equal(kept[k][1].cpu().numpy().reshape(Np, M).T, one[1]), (G, k, "consumer of product 1")
            assert np.array_equal(Ps[k].cpu().numpy().reshape(Np, D).T, other[0]), (G, k)
            assert np.array_equal(Is[k].cpu().numpy().reshape(Np, M).T, other[1]), (G, k)
# diagnostics of a multi-device product: per-device launch duration and arrival skew (kdehip_product_multi_timing)
with kdehip.MultiProductPlan(g, first_device=0, ngpus=4) as mp:
    Ps = [torch.zeros(D * Np, dtype=torch.float64, device=dev) for _ in range(4)]
    Is = [torch.zeros(M * Np, dtype=torch.int64, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    try:
        mp.timing()
        raise SystemExit("timing of an untimed product must be refused")
    except kdehip.KdeHipError:
        pass
    kdehip._clib.kdehip_profile_sampler(1)
    mp.sample_philox_device(Np, Niter, seed, 0, True, Ps, Is, None)
    kms, dms = mp.timing()
    kdehip._clib.kdehip_profile_sampler(0)
    assert kms.shape == (4,) and (kms > 0).all() and (kms < 100).all(), kms
    assert dms.min() == 0.0 and (dms >= 0).all() and (dms < 1000).all(), dms
    torch.cuda.synchronize()
    assert np.array_equal(Ps[3].cpu().numpy().reshape(Np, D).T, one[0])
print("alias ok")
'''

_PEER_CHECK_SCRIPT = r'''
import ctypes, numpy as np, torch, kdehip
from tests.helpers import silverman_bw, synth_mixture
rng = np.random.default_rng(12)
D, M, N, Np, Niter, seed = 2, 3, 200, 300, 2, 5
g = [kdehip.kde(p, silverman_bw(p)) for p in [synth_mixture(rng, D, N) for _ in range(M)]]
one = kdehip.prodAppxMSGibbsS(None, g, None, None, Niter=Niter, Np=Np, seed=seed)
dev = torch.device("cuda", 0)
hip = ctypes.CDLL(None)   # the HIP runtime libkdehip.so is bound to
with kdehip.MultiProductPlan(g, first_device=0, ngpus=2) as mp:
    # (1) plain device memory (torch's allocator): the look-up passes, the gather stays fused
    Ps = [torch.zeros(D * Np, dtype=torch.float64, device=dev) for _ in range(2)]
    Is = [torch.zeros(M * Np, dtype=torch.int64, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    mp.sample_philox_device(Np, Niter, seed, 0, True, Ps, Is, None)
    torch.cuda.synchronize()
    assert mp.transfers_per_product == 0
    assert np.array_equal(Ps[1].cpu().numpy().reshape(Np, D).T, one[0])
    # (2) arrays from the stream-ordered pool (hipMallocAsync): looked at, reachable from the device itself -> still correct
    h = hip
    if hasattr(h, "hipMallocAsync"):
        ptrs = []
        for nbytes in (8 * D * Np, 8 * M * Np, 8 * D * Np, 8 * M * Np):
            p = ctypes.c_void_p()
            assert h.hipMallocAsync(ctypes.byref(p), ctypes.c_size_t(nbytes), None) == 0
            ptrs.append(p.value)
        h.hipDeviceSynchronize()
        mp.sample_philox_device(Np, Niter, seed, 0, True, [ptrs[0], ptrs[2]], [ptrs[1], ptrs[3]], None)
        h.hipDeviceSynchronize()
        out = np.zeros(D * Np)
        assert h.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptrs[2]), ctypes.c_size_t(8 * D * Np), 2) == 0
        assert np.array_equal(out.reshape(Np, D).T, one[0]), "pool-allocated destination"
        print("pool transfers:", mp.transfers_per_product)
        for p in ptrs:
            h.hipFreeAsync(ctypes.c_void_p(p), None)
        h.hipDeviceSynchronize()
print("peer check ok")
'''


def test_multi_device_code_paths_with_aliased_devices():
    """KDEHIP_ALIAS_DEVICES=1 lets logical devices wrap around the visible ones: the complete N > 1 paths (slicing,
    one plan per device, per-device streams, the all-gather fused into the kernel epilogue, event waits in both
    directions (write-after-read included), per-device result slices) run on one
    GPU and must reproduce the one-device result bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KDEHIP_ALIAS_DEVICES="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", _ALIAS_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "alias ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_peer_destination_lookup_on_aliased_devices():
    """The fused gather stores into the caller's arrays on the other devices; which arrays a peer can reach is looked up
    per product (plain device memory: yes; stream-ordered pools / virtual-memory mappings: what their access flags say;
    csrc/product.hip peer_can_store).  KDEHIP_PEER_CHECK_ALIASED=1 runs that look-up on one GPU: results stay correct for
    plain and for pool-allocated destinations, whichever path the product takes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KDEHIP_ALIAS_DEVICES="1", KDEHIP_PEER_CHECK_ALIASED="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", _PEER_CHECK_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "peer check ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
