"""Host-side mirror of the callers either side of the product: automatic-bandwidth `kde!(points)`
(reference src/KDE01.jl:3-27) and direct evaluation `evaluateDualTree` / `bd(pos)`
(src/DualTree01.jl:370-446), both running on the GPU through libkdehip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import f64p, ptr
from .density import BallTreeDensity, kde


def auto_bandwidth(points, device=0, return_evals=False):
    """Per-dimension LOOCV bandwidth (standard deviations) that `kde!(points)` selects."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 1:
        pts = pts.reshape(1, -1)
    D, N = pts.shape
    flat = np.ascontiguousarray(pts.T).ravel()
    bw = np.zeros(D)
    ne = C.c_int32(0)
    _lib.check(_lib.lib.kdehip_auto_bandwidth(D, N, ptr(flat, f64p), ptr(bw, f64p), C.byref(ne), int(device)))
    return (bw, ne.value) if return_evals else bw


def kde_auto(points, device=0, overlap=None) -> BallTreeDensity:
    """`kde!(points)`: LOOCV bandwidth per dimension, then `kde!(points, bwds)` (src/KDE01.jl:24).

    The tree's topology, bounding boxes, weights and means do not depend on the bandwidth: the host builder runs on the
    library's worker threads WHILE the GPU searches the bandwidth, and the variances are filled in afterwards
    (kdehip_make_density_auto) -- bit-identical to building with the final bandwidth.  `overlap=False`: the two steps one
    after the other (what the tests compare with)."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 1:
        pts = pts.reshape(1, -1)
    D, N = pts.shape
    if overlap is None:
        overlap = True
    if N < 2 or not overlap:
        return kde(pts, auto_bandwidth(pts, device=device))
    from .density import _empty_density
    flat = np.ascontiguousarray(pts.T).ravel()
    bd = _empty_density(D, N)
    bt = bd.bt
    bw = np.empty(D)
    i64p = _lib.i64p
    _lib.check(_lib.lib.kdehip_make_density_auto(
        D, N, ptr(flat, f64p), ptr(bw, f64p), None, int(device), ptr(bt.centers, f64p), ptr(bt.ranges, f64p),
        ptr(bt.weights, f64p), ptr(bt.left_child, i64p), ptr(bt.right_child, i64p), ptr(bt.lowest_leaf, i64p),
        ptr(bt.highest_leaf, i64p), ptr(bt.permutation, i64p), ptr(bd.means, f64p), ptr(bd.bandwidth, f64p),
        ptr(bd.bandwidthMin, f64p), ptr(bd.bandwidthMax, f64p)))
    return bd


def evaluateDualTree(bd: BallTreeDensity, pos=None, lvFlag=False, errTol=1e-3, device=0):
    """`evaluateDualTree(bd, pos, lvFlag)` (src/DualTree01.jl:370-421) with FORCE_EVAL_DIRECT = true
    (errTol is then unused, as in the reference).  pos: (D, Nq) matrix, a vector of 1-D positions, or a
    BallTreeDensity whose points are used; lvFlag=True (or pos is bd) evaluates leave-one-out at bd's
    own points and returns the values in the original point order."""
    cd = bd._cstruct()
    if lvFlag or pos is bd:
        out = np.zeros(bd.bt.num_points)
        _lib.check(_lib.lib.kdehip_evaluate(C.byref(cd), None, 0, 1, ptr(out, f64p), int(device)))
        return out
    if isinstance(pos, BallTreeDensity):
        from .density import getPoints
        pos = getPoints(pos)
    pos = np.asarray(pos, dtype=np.float64)
    if pos.ndim == 1:
        pos = pos.reshape(1, -1)
    if pos.shape[0] != bd.bt.dims:
        raise ValueError("bd and pos must have the same dimension")
    flat = np.ascontiguousarray(pos.T).ravel()
    out = np.zeros(pos.shape[1])
    _lib.check(_lib.lib.kdehip_evaluate(C.byref(cd), ptr(flat, f64p), pos.shape[1], 0, ptr(out, f64p), int(device)))
    return out
