"""Many products in one call (include/kdehip.h section 2e, `kdehip_prod_philox_batch`): one device block, one gather launch,
one sampling launch per (dimension count, density count) group -- and every product's numbers are those of its own
`kdehip_prod_philox_device` call, bit for bit."""
import numpy as np
import pytest

import kdehip
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu


def _trees(seed, D, Ns, weighted=False):
    rng = np.random.default_rng(seed)
    out = []
    for N in Ns:
        pts = synth_mixture(rng, D, N)
        ks = silverman_bw(pts) if N > 1 else np.full(D, 0.4)
        ks = np.where(ks > 0, ks, 0.4)
        w = rng.uniform(0.2, 1.0, size=N) if weighted else None
        out.append(kdehip.kde(pts, ks, w))
    return out


def _run_batch(specs, precision=64, stream=None):
    """specs: (trees, Np, Niter, seed, kwargs) -> list of (points, indices[, labels]) from ONE batched call"""
    import torch
    dev = torch.device("cuda", 0)
    dds, prods, outs = [], [], []
    for trees, Np, Niter, seed, kw in specs:
        dd = [kdehip.DeviceDensity(t) for t in trees]
        dds.append(dd)
        D, M = trees[0].bt.dims, len(trees)
        P = torch.full((max(1, D * Np),), -7.0, dtype=torch.float64, device=dev)
        I = torch.full((max(1, M * Np),), -7, dtype=torch.int64, device=dev)
        pr = dict(trees=dd, d_points=P, d_indices=I, Np=Np, Niter=Niter, seed=seed, **kw)
        L = kdehip.nlevels(max(t.bt.num_points for t in trees))
        if kw.get("want_labels"):
            pr.pop("want_labels")
            pr["d_labels"] = torch.zeros(max(1, Np * M * L), dtype=torch.int32, device=dev)
        prods.append(pr)
    torch.cuda.synchronize()
    kdehip.prodAppxMSGibbsS_batch(prods, precision=precision, stream=stream)
    torch.cuda.synchronize()
    for (trees, Np, Niter, seed, kw), pr in zip(specs, prods):
        D, M = trees[0].bt.dims, len(trees)
        o = [pr["d_points"].cpu().numpy()[:D * Np].reshape(Np, D).T, pr["d_indices"].cpu().numpy()[:M * Np].reshape(Np, M).T]
        if "d_labels" in pr:
            L = kdehip.nlevels(max(t.bt.num_points for t in trees))
            o.append(pr["d_labels"].cpu().numpy()[:Np * M * L].reshape(Np, M, L))
        outs.append(o)
    for dd in dds:
        for d in dd:
            d.close()
    return outs


def _reference(trees, Np, Niter, seed, kw, precision=64):
    glbs = kdehip.makeEmptyGbGlb(recordChoosen=True) if kw.get("want_labels") else None
    p, i = kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=seed, precision=precision,
                                   addEntropy=kw.get("addEntropy", True), partialDimMask=kw.get("partialDimMask"), glbs=glbs)
    return p, i, glbs


def test_sixty_four_config2_shaped_products_in_one_call():
    """BASELINE config 2's shape (2-D, 3 x 200 points, 256 chains, Niter 5) sixty-four times, different densities and
    seeds: one sampling launch, every product equal to its single call."""
    specs = [(_trees(500 + k, 2, [200, 200, 200]), 256, 5, 9000 + k, {}) for k in range(64)]
    outs = _run_batch(specs)
    for k, (spec, o) in enumerate(zip(specs, outs)):
        rp, ri, _ = _reference(*spec)
        assert np.array_equal(o[1], ri), k
        assert np.array_equal(o[0], rp), k


def test_mixed_batch_groups_and_single_products():
    """Groups of different (D, M), chain counts that are no multiple of the workgroup width, an empty product, label traces,
    addEntropy off -- and products outside the batched kernel's domain (five densities, one density, a mask) that the call
    enqueues one by one: all equal to their single calls."""
    mask = np.array([[1, 1, 0], [1, 0, 1]], dtype=bool)
    specs = [
        (_trees(1, 3, [150, 151]), 100, 3, 11, {}),
        (_trees(2, 3, [64, 300]), 17, 2, 12, {"want_labels": True}),
        (_trees(3, 3, [500, 20]), 1, 4, 13, {}),
        (_trees(4, 6, [200] * 4), 200, 3, 14, {}),
        (_trees(5, 6, [1000, 300, 65, 200]), 33, 1, 15, {"addEntropy": False}),
        (_trees(6, 2, [90, 91, 92]), 0, 3, 16, {}),
        (_trees(7, 2, [90, 91, 92]), 48, 0, 17, {}),
        (_trees(8, 2, [128, 128, 64]), 129, 5, 18, {"want_labels": True}),
        (_trees(9, 4, [70, 80, 90, 100, 110], weighted=True), 40, 2, 19, {}),   # five densities: the general kernel
        (_trees(10, 2, [300]), 25, 2, 20, {}),                                  # one density
        (_trees(11, 3, [120, 130]), 50, 2, 21, {"partialDimMask": mask}),       # masked
        (_trees(12, 3, [2000, 3000]), 20, 1, 22, {}),                           # streamed tiles
        (_trees(13, 1, [5000, 4000]), 16, 1, 23, {}),
    ]
    outs = _run_batch(specs)
    for k, (spec, o) in enumerate(zip(specs, outs)):
        Np = spec[1]
        if Np == 0:
            continue
        rp, ri, glbs = _reference(*spec)
        assert np.array_equal(o[1], ri), k
        assert np.array_equal(o[0], rp), k
        if spec[4].get("want_labels"):
            M = len(spec[0])
            for s in (0, Np - 1):
                for j in range(M):
                    got = o[2][s, j, :]
                    exp = [glbs.labelsChoosen[s + 1][j + 1][l + 1] for l in range(len(got))]
                    assert list(got) == exp, (k, s, j)


def test_batch_in_fp32_and_on_a_caller_stream():
    import torch
    specs = [(_trees(30 + k, 3, [100 + k, 120]), 64 + k, 2, 50 + k, {}) for k in range(6)]
    st = torch.cuda.Stream()
    for prec in (32, 64):
        outs = _run_batch(specs, precision=prec, stream=st.cuda_stream)
        for k, (spec, o) in enumerate(zip(specs, outs)):
            rp, ri, _ = _reference(*spec, precision=prec)
            assert np.array_equal(o[1], ri), (prec, k)
            assert np.array_equal(o[0], rp), (prec, k)


def test_batches_back_to_back_reuse_released_blocks():
    """Several batches in flight on one stream (each call's block is released by a later call once its work has run)."""
    import torch
    dev = torch.device("cuda", 0)
    trees = [_trees(60 + k, 2, [150, 160, 170]) for k in range(8)]
    dd = [[kdehip.DeviceDensity(t) for t in ts] for ts in trees]
    Np = 96
    P = [[torch.zeros(2 * Np, dtype=torch.float64, device=dev) for _ in range(8)] for _ in range(12)]
    I = [[torch.zeros(3 * Np, dtype=torch.int64, device=dev) for _ in range(8)] for _ in range(12)]
    torch.cuda.synchronize()
    for r in range(12):
        kdehip.prodAppxMSGibbsS_batch([dict(trees=dd[k], d_points=P[r][k], d_indices=I[r][k], Np=Np, Niter=3, seed=100 * r + k)
                                       for k in range(8)])
    torch.cuda.synchronize()
    for r in (0, 5, 11):
        for k in range(8):
            rp, ri = kdehip.prodAppxMSGibbsS(None, trees[k], None, None, Niter=3, Np=Np, seed=100 * r + k)
            assert np.array_equal(I[r][k].cpu().numpy().reshape(Np, 3).T, ri)
            assert np.array_equal(P[r][k].cpu().numpy().reshape(Np, 2).T, rp)
    for ds in dd:
        for d in ds:
            d.close()
    kdehip._clib.kdehip_clear_cache()


def test_batch_argument_errors():
    import torch
    t2 = _trees(1, 2, [50, 60])
    t3 = _trees(2, 3, [50])
    dd = [kdehip.DeviceDensity(t) for t in t2] + [kdehip.DeviceDensity(t3[0])]
    P = torch.zeros(2 * 10, dtype=torch.float64, device="cuda:0")
    I = torch.zeros(2 * 10, dtype=torch.int64, device="cuda:0")
    with pytest.raises(ValueError, match="same dimension"):
        kdehip.prodAppxMSGibbsS_batch([dict(trees=[dd[0], dd[2]], d_points=P, d_indices=I, Np=10)])
    with pytest.raises(kdehip.KdeHipError):
        kdehip.prodAppxMSGibbsS_batch([dict(trees=dd[:2], d_points=None, d_indices=I, Np=10)])
    kdehip.prodAppxMSGibbsS_batch([])   # nothing to do
    for d in dd:
        d.close()
