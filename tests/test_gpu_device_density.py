"""Densities that live in HBM (include/kdehip.h section 2c, csrc/pack_device.hip): `kdehip_density_upload` keeps a
BallTreeDensity and its per-level frontiers on the device, `kdehip_prod_philox_device` lays a product out on the GPU
(gather kernel) and samples it with outputs left in HBM.  Same layout, same kernels, same Philox stream as the
host-packed one-shot call: results must be bit-identical to `prodAppxMSGibbsS(..., seed=...)`, whatever the shapes."""
import numpy as np
import pytest

import kdehip
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


def _trees(seed, D, Ns, weighted=False, per_dim_bw=True):
    rng = np.random.default_rng(seed)
    out = []
    for N in Ns:
        pts = synth_mixture(rng, D, N)
        ks = silverman_bw(pts) if (N > 1 and per_dim_bw) else np.full(D, 0.4)
        ks = np.where(ks > 0, ks, 0.4)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        out.append(kdehip.kde(pts, ks, w))
    return out


@pytest.mark.parametrize("D,Ns,Np,Niter,weighted,prec", [
    (6, [1000] * 4, 300, 3, False, 64),          # BASELINE config 3's shape
    (2, [200, 200, 200], 256, 5, False, 64),     # config 2
    (3, [5000] * 8, 40, 1, False, 64),           # config 4's shape: chunked tiles
    (6, [10000] * 4, 24, 2, False, 32),          # config 5's shape, fp32 tiles
    (3, [37, 128, 129, 300], 70, 2, True, 64),   # ragged sizes, weights: frontiers repeat beyond a density's own depth
    (4, [1, 60, 7], 33, 3, False, 64),           # a single-point density
    (1, [100, 100], 100, 5, False, 32),
    (8, [130, 90, 64, 65, 33], 50, 1, True, 64), # five densities: the general kernel
])
def test_device_resident_product_equals_host_packed_product(D, Ns, Np, Niter, weighted, prec):
    import torch
    dev = torch.device("cuda", 0)
    trees = _trees(700 + D + len(Ns), D, Ns, weighted)
    M = len(Ns)
    seed = 31
    ref_p, ref_i = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=seed, precision=prec)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    assert [d.num_points for d in dd] == list(Ns) and all(d.dims == D for d in dd)
    P = torch.zeros(D * Np, dtype=torch.float64, device=dev)
    I = torch.zeros(M * Np, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()   # (the zero fills ran on torch's stream, the products run on `st`)
    for rep in range(3):   # (plans of earlier calls are released by later ones)
        kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=Niter, seed=seed, precision=prec, stream=st.cuda_stream)
    st.synchronize()
    assert np.array_equal(I.cpu().numpy().reshape(Np, M).T, ref_i)
    assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, ref_p)
    # a sample offset continues the same stream: chains [Np/2, Np) of the call above
    half = Np // 2
    P2 = torch.zeros(D * (Np - half), dtype=torch.float64, device=dev)
    I2 = torch.zeros(M * (Np - half), dtype=torch.int64, device=dev)
    kdehip.prodAppxMSGibbsS_device(dd, P2, I2, Np=Np - half, Niter=Niter, seed=seed, sample_offset=half, precision=prec)
    torch.cuda.synchronize()
    assert np.array_equal(I2.cpu().numpy().reshape(Np - half, M).T, ref_i[:, half:])
    assert np.array_equal(P2.cpu().numpy().reshape(Np - half, D).T, ref_p[:, half:])
    # host outputs (the route of a host without device arrays of its own)
    rp, ri = kdehip.prodAppxMSGibbsS_resident(dd, Np=Np, Niter=Niter, seed=seed, precision=prec)
    assert np.array_equal(ri, ref_i) and np.array_equal(rp, ref_p)
    for d in dd:
        d.close()
    kdehip._clib.kdehip_clear_cache()


def test_device_resident_product_with_mask_and_labels():
    import torch
    dev = torch.device("cuda", 0)
    D, Ns, Np, Niter = 3, [150, 180, 90], 64, 2
    trees = _trees(5, D, Ns)
    mask = [[True, True, False], [True, False, True], [False, True, True]]
    glbs = kdehip.makeEmptyGbGlb(recordChoosen=True)
    ref_p, ref_i = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=8, partialDimMask=mask, glbs=glbs)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    L = kdehip.nlevels(max(Ns))
    P = torch.zeros(D * Np, dtype=torch.float64, device=dev)
    I = torch.zeros(len(Ns) * Np, dtype=torch.int64, device=dev)
    Lab = torch.zeros(Np * len(Ns) * L, dtype=torch.int32, device=dev)
    kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=Niter, seed=8, partialDimMask=mask, d_labels=Lab)
    torch.cuda.synchronize()
    assert np.array_equal(I.cpu().numpy().reshape(Np, len(Ns)).T, ref_i)
    assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, ref_p)
    lab = Lab.cpu().numpy().reshape(Np, len(Ns), L)
    assert [glbs.labelsChoosen[7][2][l + 1] for l in range(L)] == list(lab[6, 1])


def test_device_density_errors():
    trees = _trees(9, 2, [50, 60])
    other = _trees(10, 3, [40])
    dd = [kdehip.DeviceDensity(t) for t in trees] + [kdehip.DeviceDensity(other[0])]
    with pytest.raises(ValueError):   # "kdes must have same dimension" (src/MSGibbs01.jl:720-722)
        kdehip.prodAppxMSGibbsS_device(dd, 0, 0, Np=0)
    with pytest.raises(kdehip.KdeHipError):
        kdehip.prodAppxMSGibbsS_device(dd[:2], None, None, Np=10)   # null outputs


def test_back_to_back_products_prepared_under_each_other():
    """An asynchronous caller enqueues product after product: the library prepares product k+1 (descriptor upload, tile
    gather, conditional tables) on a stream of its own while product k samples (csrc/product.hip prep_stream).  Forty
    products of four different shapes, interleaved on two caller streams and the legacy default stream, every one into
    its own output arrays: each must equal the blocking call with the same seed -- no plan may see another's tiles,
    tables or recycled blocks."""
    import torch
    dev = torch.device("cuda", 0)
    shapes = [(6, [1000] * 4, 2048, 2), (2, [200, 150, 300], 700, 3), (3, [2500] * 3, 96, 1), (4, [64, 640], 1024, 4)]
    probs = []
    for k, (D, Ns, Np, Niter) in enumerate(shapes):
        trees = _trees(40 + k, D, Ns)
        probs.append((D, len(Ns), Np, Niter, [kdehip.DeviceDensity(t) for t in trees], trees))
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), None]
    runs = []
    outs = []
    for c in range(40):
        D, M, Np, Niter, dd, trees = probs[c % len(probs)]
        outs.append((torch.zeros(D * Np, dtype=torch.float64, device=dev), torch.zeros(M * Np, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()   # (the fills ran on torch's stream; the products below run on others)
    for c in range(40):
        D, M, Np, Niter, dd, trees = probs[c % len(probs)]
        P, I = outs[c]
        st = streams[c % len(streams)]
        kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=Niter, seed=1000 + c,
                                       stream=None if st is None else st.cuda_stream)
        runs.append((c, P, I))
    torch.cuda.synchronize()
    for c, P, I in runs:
        D, M, Np, Niter, dd, trees = probs[c % len(probs)]
        rp, ri = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=1000 + c)
        assert np.array_equal(I.cpu().numpy().reshape(Np, M).T, ri), c
        assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, rp), c
    for pr in probs:
        for d in pr[4]:
            d.close()
    kdehip._clib.kdehip_clear_cache()


def test_sampler_profile_counts_the_launches_it_brackets():
    """kdehip_profile_sampler: while enabled, every asynchronous device product has its sampling launch bracketed by timing
    events on the caller's stream (what bench.py reports as the kernel's duration in its loop)."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    D, Ns, Np = 3, [400, 300], 512
    trees = _trees(77, D, Ns)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    P = torch.zeros(D * Np, dtype=torch.float64, device=dev)
    I = torch.zeros(len(Ns) * Np, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib = kdehip._clib
    ms, n = C.c_double(-1.0), C.c_int64(-1)
    lib.kdehip_profile_sampler(1)
    for c in range(3):
        kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=2, seed=c)
    assert lib.kdehip_profile_sampler_read(0, None, C.byref(ms), C.byref(n)) == 0
    assert n.value == 3 and 0.0 < ms.value < 100.0
    lib.kdehip_profile_sampler(0)   # (off, and the statistics start over)
    kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=2, seed=9)
    assert lib.kdehip_profile_sampler_read(0, None, C.byref(ms), C.byref(n)) == 0
    assert n.value == 0 and ms.value == 0.0
    ref_p, ref_i = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=2, Np=Np, seed=9)
    assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, ref_p)
    for d in dd:
        d.close()
