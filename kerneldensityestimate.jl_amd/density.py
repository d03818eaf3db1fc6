"""Host-side mirror of the reference's density container and constructors.

`BallTreeDensity` carries exactly the flat arrays of the reference struct (src/BallTreeDensity01.jl:11-24
with the embedded BallTree, src/BallTree01.jl:10-28) under the same field names; `kde` mirrors
`kde!(points, ks[, weights])` (src/KDE01.jl:34-84).  Construction runs in libkdehip.so
(csrc/balltree.cpp) -- never in the CPU oracle.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import f64p, i64p, ptr


class BallTree:
    """Fields of the reference `BallTree` (1-based node ids, NO_CHILD = -1)."""
    __slots__ = ("dims", "num_points", "centers", "ranges", "weights", "left_child", "right_child",
                 "lowest_leaf", "highest_leaf", "permutation")


class BallTreeDensity:
    """Fields of the reference `BallTreeDensity`; `bandwidth` holds VARIANCES (src/KDE01.jl:45)."""
    __slots__ = ("bt", "multibandwidth", "means", "bandwidth", "bandwidthMin", "bandwidthMax", "_cstruct_cache")

    def __repr__(self):
        return f"BallTreeDensity(dims={Ndim(self)}, Npts={Npts(self)}, bws={np.round(getBW(self)[:, 0], 6)})"

    def _cstruct(self):
        """The C view of the six arrays a product reads (kdehip_density); kept while the arrays are the same objects."""
        bt = self.bt
        # the array OBJECTS are part of the key (compared with `is`): that also pins their lifetime, so the id of a freed
        # array can never be taken over by a new one while the cached struct still points at the old memory
        key = (self.means, self.bandwidth, bt.weights, bt.left_child, bt.right_child, bt.permutation, bt.num_points, bt.dims)
        cached = getattr(self, "_cstruct_cache", None)
        if cached is not None and len(cached[0]) == len(key) and all(a is b for a, b in zip(cached[0][:6], key[:6])) \
                and cached[0][6:] == key[6:]:
            return cached[1]
        cs = _lib.CDensity(bt.num_points, bt.dims, ptr(self.means, f64p), ptr(self.bandwidth, f64p),
                           ptr(bt.weights, f64p), ptr(bt.left_child, i64p), ptr(bt.right_child, i64p),
                           ptr(bt.permutation, i64p))
        self._cstruct_cache = (key, cs)
        return cs

    # `bd(pos)`: evaluate the density at points (reference functor, src/DualTree01.jl:431-446)
    def __call__(self, pos, lvFlag=False, errTol=1e-3):
        from .bandwidth import evaluateDualTree
        return evaluateDualTree(self, pos, lvFlag, errTol)

    # `p1 * p2` / `*([p1, p2, ...])`, reference src/MSGibbs01.jl:707-736
    def __mul__(self, other):
        from .product import mul
        return mul([self, other])


def _prepare(points, ks, weights):
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 1:
        pts = pts.reshape(1, -1)
    if pts.ndim != 2:
        raise ValueError("points must be a (D, N) matrix or a vector")
    D, N = pts.shape
    ks = np.ascontiguousarray(np.atleast_1d(np.asarray(ks, dtype=np.float64)).ravel())
    if ks.size not in (1, D):
        raise ValueError("ks must have 1 or D entries")
    w = None
    if weights is not None:
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).ravel())
        if w.size != N:
            raise ValueError("weights must have one entry per point")
    flat = np.ascontiguousarray(pts.T).ravel()  # column-major D x N
    return D, N, flat, ks, w


def _empty_density(D, N) -> BallTreeDensity:
    """The reference's twelve arrays as views of ONE block: a fresh allocation of this size is an mmap of its own, and a
    dozen mmap/munmap pairs per density (with the TLB shoot-downs a munmap costs a process that runs worker threads) were
    a third of `kde()`'s time."""
    f8 = [("centers", 2 * N * D), ("ranges", 2 * N * D), ("means", 2 * N * D), ("bandwidth", 2 * N * D),
          ("bandwidthMin", N * D), ("bandwidthMax", N * D), ("weights", 2 * N)]
    i8 = ["left_child", "right_child", "lowest_leaf", "highest_leaf", "permutation"]
    words = sum(n for _, n in f8) + len(i8) * 2 * N
    block = np.empty(words, dtype=np.float64)
    views, at = {}, 0
    for name, n in f8:
        views[name] = block[at:at + n]
        at += n
    for name in i8:
        views[name] = block[at:at + 2 * N].view(np.int64)
        at += 2 * N
    bt = BallTree()
    bt.dims, bt.num_points = D, N
    for name in ("centers", "ranges", "weights", *i8):
        setattr(bt, name, views[name])
    bd = BallTreeDensity()
    bd.bt = bt
    bd.multibandwidth = 0
    for name in ("means", "bandwidth", "bandwidthMin", "bandwidthMax"):
        setattr(bd, name, views[name])
    return bd


def kde(points, ks=None, weights=None, device=None) -> BallTreeDensity:
    """`kde!(points, ks)` / `kde!(points, ks, weights)` (reference src/KDE01.jl:34-84); with ks=None the
    automatic LOOCV bandwidth `kde!(points)` (src/KDE01.jl:3-27, GPU).

    points: (D, N) array, or a length-N vector for 1-D data (:78-84).  ks: bandwidth as STANDARD
    DEVIATION, one entry (repeated over dimensions, :41-43) or D entries.  weights: N values,
    normalised to sum 1 (:46); default ones (:67).  device: None = the host builder (csrc/balltree.cpp);
    a HIP ordinal = the GPU builder (csrc/treebuild.hip, bit-identical arrays) where the density fits it.
    """
    if ks is None:
        if weights is not None:
            raise ValueError("kde!(points) with automatic bandwidth takes no weights")
        from .bandwidth import kde_auto
        return kde_auto(points, device=0 if device is None else device)
    if device is not None:
        return kde_batch([(points, ks, weights)], device=device)[0]
    D, N, flat, ks, w = _prepare(points, ks, weights)
    bd = _empty_density(D, N)
    bt = bd.bt
    _lib.check(_lib.lib.kdehip_make_density(
        D, N, ptr(flat, f64p), ptr(ks, f64p), ks.size, None if w is None else ptr(w, f64p),
        ptr(bt.centers, f64p), ptr(bt.ranges, f64p), ptr(bt.weights, f64p), ptr(bt.left_child, i64p),
        ptr(bt.right_child, i64p), ptr(bt.lowest_leaf, i64p), ptr(bt.highest_leaf, i64p),
        ptr(bt.permutation, i64p), ptr(bd.means, f64p), ptr(bd.bandwidth, f64p),
        ptr(bd.bandwidthMin, f64p), ptr(bd.bandwidthMax, f64p)))
    return bd


def kde_batch(items, device=0):
    """Several `kde!(points, ks[, weights])` at once on the GPU (kdehip_make_densities_device: one workgroup per
    density).  items: (points, ks) or (points, ks, weights) tuples of ONE dimension count and ks length.  Densities
    the device builder cannot hold (kdehip_make_density_device_supported) are built by the host builder instead."""
    import ctypes as C
    prepared = []
    for it in items:
        points, ks = it[0], it[1]
        weights = it[2] if len(it) > 2 else None
        prepared.append(_prepare(points, ks, weights))
    out = [None] * len(prepared)
    groups = {}
    for idx, (D, N, flat, ks, w) in enumerate(prepared):
        if N >= 2 and _lib.lib.kdehip_make_density_device_supported(D, N):
            groups.setdefault((D, ks.size), []).append(idx)
        else:
            out[idx] = kde(np.asarray(items[idx][0]), items[idx][1], items[idx][2] if len(items[idx]) > 2 else None)
    for (D, nks), idxs in groups.items():
        for c0 in range(0, len(idxs), _lib.MAX_DENS):
            chunk = idxs[c0:c0 + _lib.MAX_DENS]
            nb = len(chunk)
            dens = [_empty_density(D, prepared[i][1]) for i in chunk]

            def arr(get):
                return (C.c_void_p * nb)(*[None if get(k) is None else get(k).ctypes.data for k in range(nb)])
            Ns = np.array([prepared[i][1] for i in chunk], dtype=np.int64)
            any_w = any(prepared[i][4] is not None for i in chunk)
            _lib.check(_lib.lib.kdehip_make_densities_device(
                nb, D, ptr(Ns, i64p), arr(lambda k: prepared[chunk[k]][2]), arr(lambda k: prepared[chunk[k]][3]), nks,
                arr(lambda k: prepared[chunk[k]][4]) if any_w else None,
                arr(lambda k: dens[k].bt.centers), arr(lambda k: dens[k].bt.ranges), arr(lambda k: dens[k].bt.weights),
                arr(lambda k: dens[k].bt.left_child), arr(lambda k: dens[k].bt.right_child),
                arr(lambda k: dens[k].bt.lowest_leaf), arr(lambda k: dens[k].bt.highest_leaf),
                arr(lambda k: dens[k].bt.permutation), arr(lambda k: dens[k].means), arr(lambda k: dens[k].bandwidth),
                arr(lambda k: dens[k].bandwidthMin), arr(lambda k: dens[k].bandwidthMax), int(device)))
            for k, i in enumerate(chunk):
                out[i] = dens[k]
    return out


kde_b = kde  # spelling of `kde!` for callers that want the bang visible


def density_from_arrays(dims, num_points, means, bandwidth, weights, left_child, right_child,
                        permutation) -> BallTreeDensity:
    """Wrap flat arrays produced elsewhere (e.g. exported from a Julia BallTreeDensity)."""
    bt = BallTree()
    bt.dims, bt.num_points = int(dims), int(num_points)
    bt.weights = np.ascontiguousarray(weights, dtype=np.float64)
    bt.left_child = np.ascontiguousarray(left_child, dtype=np.int64)
    bt.right_child = np.ascontiguousarray(right_child, dtype=np.int64)
    bt.permutation = np.ascontiguousarray(permutation, dtype=np.int64)
    bt.centers = bt.ranges = bt.lowest_leaf = bt.highest_leaf = None
    bd = BallTreeDensity()
    bd.bt = bt
    bd.multibandwidth = 0
    bd.means = np.ascontiguousarray(means, dtype=np.float64)
    bd.bandwidth = np.ascontiguousarray(bandwidth, dtype=np.float64)
    bd.bandwidthMin = bd.bandwidthMax = None
    return bd


def Ndim(bd: BallTreeDensity) -> int:
    return bd.bt.dims


def Npts(bd: BallTreeDensity) -> int:
    return bd.bt.num_points


def getPoints(bd: BallTreeDensity, idx=None):
    """Points in the caller's original order (reference src/KDE01.jl:91-101); `idx` (0-based here) selects
    columns like the reference's second argument."""
    N, D = bd.bt.num_points, bd.bt.dims
    perm = bd.bt.permutation[N:] - 1
    out = np.zeros((D, N))
    out[:, perm] = bd.bt.centers[N * D:].reshape(N, D).T
    return out if idx is None else out[:, idx]


def getBW(bd: BallTreeDensity, ind=None):
    """Per-point bandwidth as standard deviation, (D, N) (reference src/KDE01.jl:109-120); `ind` 0-based."""
    N, D = bd.bt.num_points, bd.bt.dims
    perm = bd.bt.permutation[N:] - 1
    out = np.zeros((D, N))
    out[:, perm] = bd.bandwidth[N * D:].reshape(N, D).T
    out = np.sqrt(out)
    return out if ind is None else out[:, ind]


def getWeights(bd: BallTreeDensity, ind=None):
    """Normalised point weights in original order (reference src/KDE01.jl:127-136); `ind` 0-based."""
    N = bd.bt.num_points
    out = np.zeros(N)
    out[bd.bt.permutation[N:] - 1] = bd.bt.weights[N:]
    return out if ind is None else out[ind]
