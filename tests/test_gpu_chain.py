"""The resident chain (include/kdehip.h section 2d, csrc/pack_device.hip): `kde!(pGM)` of a product that never left HBM
(`kdehip_density_from_device_points`) and the `*` operator on handles (`kdehip_mul_device`, reference
src/MSGibbs01.jl:707-726).  The device route must give, bit for bit, the density the host route builds from the same
numbers: same LOOCV kernels on the same values, the same pooled host tree builder, the same moment matching."""
import numpy as np
import pytest

import kdehip
from tests.helpers import silverman_bw, synth_mixture

pytestmark = pytest.mark.gpu

ARRAYS = ("centers", "ranges", "weights", "left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation")
DARRAYS = ("means", "bandwidth", "bandwidthMin", "bandwidthMax")


def assert_same_density(a, b, what=""):
    assert a.bt.dims == b.bt.dims and a.bt.num_points == b.bt.num_points, what
    for name in ARRAYS:
        assert np.array_equal(getattr(a.bt, name), getattr(b.bt, name)), (what, name)
    for name in DARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), (what, name)


def _trees(seed, D, Ns):
    rng = np.random.default_rng(seed)
    out = []
    for N in Ns:
        pts = synth_mixture(rng, D, N)
        out.append(kdehip.kde(pts, silverman_bw(pts)))
    return out


@pytest.mark.parametrize("D,N", [(1, 2), (1, 3), (2, 100), (6, 1000), (6, 2048), (3, 2049), (2, 3000), (8, 257), (1, 4097)])
def test_kde_of_device_points_equals_kde_of_host_points(D, N):
    """`kde!(points)` from a device matrix == `kde!(points)` from the host's copy (marginals up to 2048 points are prepared
    by the GPU straight from the device matrix, larger ones on the host; N > 4096 takes the two-launch rounds)."""
    import torch
    rng = np.random.default_rng(100 * D + N)
    pts = synth_mixture(rng, D, N)
    ref = kdehip.kde_auto(pts)
    flat = torch.from_numpy(np.ascontiguousarray(pts.T).ravel()).to("cuda:0")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        flat2 = flat * 1.0   # (produced on a stream of its own: the entry waits for it)
    with kdehip.DeviceDensity.from_device_points(flat2, D, N, stream=st.cuda_stream) as dd:
        assert dd.num_points == N and dd.dims == D
        got = dd.download()
        assert_same_density(got, ref, f"D={D} N={N}")
        assert np.array_equal(dd.bw, kdehip.getBW(ref)[:, 0])
        assert dd.nevals > 0
        # ... and it is a product input like any uploaded density
        other = kdehip.DeviceDensity(ref)
        Np = 64
        P = torch.zeros(D * Np, dtype=torch.float64, device="cuda:0")
        I = torch.zeros(2 * Np, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        kdehip.prodAppxMSGibbsS_device([dd, other], P, I, Np=Np, Niter=2, seed=5)
        torch.cuda.synchronize()
        rp, ri = kdehip.prodAppxMSGibbsS(None, [ref, ref], None, None, Niter=2, Np=Np, seed=5)
        assert np.array_equal(I.cpu().numpy().reshape(Np, 2).T, ri)
        assert np.array_equal(P.cpu().numpy().reshape(Np, D).T, rp)
        other.close()


@pytest.mark.parametrize("D,Ns", [(6, [1000] * 4), (2, [200, 200, 200]), (3, [150, 151]), (1, [100, 100]), (4, [37, 300, 129, 64, 90])])
def test_mul_device_equals_mul(D, Ns):
    trees = _trees(40 + D, D, Ns)
    ref = kdehip.mul(trees, seed=77)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    with kdehip.mul_device(dd, seed=77) as out:
        assert out.num_points == int(round(float(np.mean(Ns))))
        assert_same_density(out.download(), ref, f"D={D} Ns={Ns}")
    ref2 = kdehip.mul(trees, seed=78, addEntropy=False)
    with kdehip.mul_device(dd, seed=78, addEntropy=False) as out:
        assert_same_density(out.download(), ref2)
    # "hack fix for #70" (src/MSGibbs01.jl:713-716): one density, no entropy -> kde! of its own points
    ref3 = kdehip.mul(trees[:1], addEntropy=False)
    with kdehip.mul_device(dd[:1], addEntropy=False) as out:
        assert_same_density(out.download(), ref3)
    for d in dd:
        d.close()


def test_mul_device_rejects_mixed_dimensions():
    a = kdehip.DeviceDensity(_trees(1, 2, [50])[0])
    b = kdehip.DeviceDensity(_trees(2, 3, [50])[0])
    with pytest.raises(ValueError, match="same dimension"):
        kdehip.mul_device([a, b], seed=1)
    with pytest.raises(kdehip.KdeHipError):
        kdehip.DeviceDensity(_trees(1, 2, [50])[0]).download()   # an uploaded density has no mirror: its arrays are the caller's
    a.close(); b.close()


def test_ten_deep_chain_on_resident_densities():
    """A chain of ten `*`: the device route (every link's product, bandwidth search and density stay in HBM) against the
    host route link by link with the same seeds -- the final densities are identical bit for bit."""
    D, N = 6, 500
    trees = _trees(9, D, [N] * 4)
    dd = [kdehip.DeviceDensity(t) for t in trees]
    h, d = trees[0], dd[0]
    made = []
    for k in range(10):
        h = kdehip.mul([h, trees[1], trees[2], trees[3]], seed=1000 + k)
        d = kdehip.mul_device([d, dd[1], dd[2], dd[3]], seed=1000 + k)
        made.append(d)
    assert_same_density(d.download(), h, "link 10")
    for x in made + dd:
        x.close()
    kdehip._clib.kdehip_clear_cache()


def _same_as_singles(products, flags, seeds):
    """mul_device_batch(products) against one mul_device per product: every array of every density, the LOOCV bandwidths
    and the evaluation counts, bit for bit; then the batch's densities must WORK as inputs (a product of two of them)."""
    outs = kdehip.mul_device_batch(products, addEntropy=flags, seeds=seeds)
    assert len(outs) == len(products)
    for k, (trees, out) in enumerate(zip(products, outs)):
        with kdehip.mul_device(trees, addEntropy=flags[k], seed=seeds[k]) as ref:
            assert out.num_points == ref.num_points and out.dims == ref.dims, k
            assert np.array_equal(out.bw, ref.bw), (k, out.bw, ref.bw)
            assert out.nevals == ref.nevals, k
            assert_same_density(out.download(), ref.download(), f"product {k}")
    return outs


def test_mul_device_batch_of_config2_shapes_equals_singles():
    """64 products of the reference's own serving shape (BASELINE config 2: 2-D, 3 x 200 points, Np = 200, Niter = 5;
    src/MSGibbs01.jl:707-726 with test/runtests.jl:189-201's sizes) in ONE call."""
    D, B = 2, 64
    pool = [kdehip.DeviceDensity(t) for t in _trees(321, D, [200] * 12)]
    rng = np.random.default_rng(5)
    products = [[pool[j] for j in rng.choice(len(pool), size=3, replace=False)] for _ in range(B)]
    seeds = [9000 + k for k in range(B)]
    outs = _same_as_singles(products, [True] * B, seeds)
    # the results are ordinary resident densities: a product of two of them == the product of the same two built singly
    with kdehip.mul_device(products[0], seed=seeds[0]) as a, kdehip.mul_device(products[1], seed=seeds[1]) as b:
        with kdehip.mul_device([a, b], seed=1) as ref, kdehip.mul_device([outs[0], outs[1]], seed=1) as got:
            assert_same_density(got.download(), ref.download(), "product of two batch results")
    # freeing in any order: the batch's shared block goes with the last handle
    for o in outs[::2] + outs[1::2]:
        o.close()
    for d in pool:
        d.close()
    kdehip._clib.kdehip_clear_cache()


def test_mul_device_batch_ragged_sizes_shortcut_and_loose_items():
    """Mixed in one call: results of different sizes and dimension counts (one bandwidth search per (D, N) group), 1 to 5
    densities per product, addEntropy on and off, the one-density shortcut (src/MSGibbs01.jl:713-716), and a result
    beyond 2048 points (built by a call of its own inside the batch)."""
    t2 = [kdehip.DeviceDensity(t) for t in _trees(11, 2, [200, 200, 150, 151, 64])]
    t6 = [kdehip.DeviceDensity(t) for t in _trees(12, 6, [300, 300, 1000, 1000])]
    t3 = [kdehip.DeviceDensity(t) for t in _trees(13, 3, [2500, 2500])]
    t1 = [kdehip.DeviceDensity(t) for t in _trees(14, 1, [100, 100, 37])]
    products = [
        [t2[0], t2[1]],                 # 2-D, Np 200
        [t2[0], t2[1], t2[2]],          # mean 183.33 -> 183
        [t2[2], t2[3]],                 # mean 150.5 -> 150 (halves to even)
        [t2[1], t2[0]],                 # Np 200 again: same group as product 0
        [t2[4]],                        # one density, entropy on: a real product of one
        [t2[4]],                        # one density, no entropy: the shortcut
        [t6[0], t6[1]],                 # 6-D, Np 300
        [t6[2], t6[3], t6[0]],          # 6-D, mean 766.67 -> 767
        [t3[0], t3[1]],                 # Np 2500 > 2048: outside the batched path
        [t1[0], t1[1]],                 # 1-D
        [t1[0], t1[1], t1[2], t1[0], t1[1]],  # five densities: the general sampler inside the batch
        [t6[0], t6[1]],                 # 6-D, Np 300 again, other seed
    ]
    flags = [True, False, True, True, True, False, True, False, True, True, True, False]
    seeds = [500 + 7 * k for k in range(len(products))]
    outs = _same_as_singles(products, flags, seeds)
    for o in outs:
        o.close()
    for d in t2 + t6 + t3 + t1:
        d.close()
    kdehip._clib.kdehip_clear_cache()


def test_mul_device_batch_argument_errors():
    a = kdehip.DeviceDensity(_trees(1, 2, [50])[0])
    b = kdehip.DeviceDensity(_trees(2, 3, [50])[0])
    assert kdehip.mul_device_batch([]) == []
    with pytest.raises(ValueError, match="same dimension"):
        kdehip.mul_device_batch([[a, a], [a, b]], seeds=[1, 2])
    with pytest.raises(kdehip.KdeHipError):
        kdehip.mul_device_batch([[a, a], []], seeds=[1, 2])
    a.close(); b.close()


def test_mul_device_batch_beyond_one_search():
    """More marginals than one bandwidth search indexes (21,000: 3,600 products x 6 dimensions): the batch splits the results of
    one size over several searches; spot-checked against single calls at both ends and across the split."""
    D, B = 6, 3600
    pool = [kdehip.DeviceDensity(t) for t in _trees(77, D, [5, 5, 5, 5])]
    rng = np.random.default_rng(3)
    products = [[pool[j] for j in rng.choice(4, size=2, replace=False)] for _ in range(B)]
    seeds = [40000 + k for k in range(B)]
    outs = kdehip.mul_device_batch(products, seeds=seeds)
    assert len(outs) == B and all(o.num_points == 5 for o in outs)
    for k in (0, 1, 3498, 3499, 3500, 3501, B - 2, B - 1):
        with kdehip.mul_device(products[k], seed=seeds[k]) as ref:
            assert np.array_equal(outs[k].bw, ref.bw) and outs[k].nevals == ref.nevals, k
            assert_same_density(outs[k].download(), ref.download(), f"product {k}")
    for o in outs:
        o.close()
    for d in pool:
        d.close()
    kdehip._clib.kdehip_clear_cache()
