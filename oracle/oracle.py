"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps oracle/libkdeoracle.so (oracle/kde_oracle.c), the plain-C restatement of the reference's
`kde!(pts, bw[, w])` (src/KDE01.jl:34-84) and `prodAppxMSGibbsS`/`gibbs1` (src/MSGibbs01.jl:527-703).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


class _Tree(C.Structure):
    _fields_ = [
        ("npts", C.c_int64),
        ("ndim", C.c_int64),
        ("means", _f64p),
        ("bandwidth", _f64p),
        ("weights", _f64p),
        ("left_child", _i64p),
        ("right_child", _i64p),
        ("permutation", _i64p),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    so = os.path.join(_HERE, "libkdeoracle.so")
    src = os.path.join(_HERE, "kde_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkdeoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.okde_make_density.restype = C.c_int
        L.okde_make_density.argtypes = [C.c_int64, C.c_int64, _f64p, _f64p, C.c_int64, _f64p] + \
            [_f64p, _f64p, _f64p, _i64p, _i64p, _i64p, _i64p, _i64p, _f64p, _f64p, _f64p, _f64p]
        L.okde_nlevels.restype = C.c_int
        L.okde_nlevels.argtypes = [C.c_int64]
        L.okde_randu_per_sample.restype = C.c_int64
        L.okde_randu_per_sample.argtypes = [C.c_int, C.c_int, C.c_int]
        L.okde_randn_per_sample.restype = C.c_int64
        L.okde_randn_per_sample.argtypes = [C.c_int, C.c_int]
        L.okde_gibbs1.restype = C.c_int
        L.okde_gibbs1.argtypes = [C.c_int, C.POINTER(_Tree), C.c_int64, C.c_int, _f64p, _i64p, _f64p,
                                  C.c_int64, _f64p, C.c_int64, C.c_int, C.c_int, _u8p, _i32p]
        L.okde_gibbs1_manifold.restype = C.c_int
        L.okde_gibbs1_manifold.argtypes = [C.c_int, C.POINTER(_Tree), C.c_int64, C.c_int, _f64p, _i64p, _f64p,
                                           C.c_int64, _f64p, C.c_int64, C.c_int, C.c_int, _u8p, _u8p, _i32p]
        L.okde_gibbs1_omp.restype = C.c_int
        L.okde_gibbs1_omp.argtypes = [C.c_int, C.POINTER(_Tree), C.c_int64, C.c_int, _f64p, _i64p, _f64p,
                                      C.c_int64, _f64p, C.c_int64, C.c_int, C.c_int, _u8p, C.c_int]
        L.okde_fallback_count.restype = C.c_int64
        L.okde_fallback_count.argtypes = [C.c_int]
        L.okde_eval_direct.restype = C.c_int
        L.okde_eval_direct.argtypes = [C.POINTER(_Tree), _f64p, C.c_int64, C.c_int, _f64p]
        L.okde_auto_bandwidth.restype = C.c_int
        L.okde_auto_bandwidth.argtypes = [C.c_int64, C.c_int64, _f64p, _f64p, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class OracleDensity:
    """Flat BallTreeDensity arrays exactly as the reference lays them out (1-based node ids)."""

    def __init__(self, points, ks, weights=None):
        pts = np.asarray(points, dtype=np.float64)
        if pts.ndim == 1:
            pts = pts.reshape(1, -1)
        D, N = pts.shape
        ks = np.atleast_1d(np.asarray(ks, dtype=np.float64)).ravel()
        self.dims, self.num_points = D, N
        pts_f = np.ascontiguousarray(pts.T).ravel()  # column-major D x N
        self.centers = np.zeros(2 * N * D)
        self.ranges = np.zeros(2 * N * D)
        self.weights = np.zeros(2 * N)
        self.left_child = np.zeros(2 * N, dtype=np.int64)
        self.right_child = np.zeros(2 * N, dtype=np.int64)
        self.lowest_leaf = np.zeros(2 * N, dtype=np.int64)
        self.highest_leaf = np.zeros(2 * N, dtype=np.int64)
        self.permutation = np.zeros(2 * N, dtype=np.int64)
        self.means = np.zeros(2 * N * D)
        self.bandwidth = np.zeros(2 * N * D)
        self.bandwidthMin = np.zeros(N * D)
        self.bandwidthMax = np.zeros(N * D)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        rc = lib().okde_make_density(
            D, N, _p(pts_f, _f64p), _p(ks, _f64p), ks.size, None if w is None else _p(w, _f64p),
            _p(self.centers, _f64p), _p(self.ranges, _f64p), _p(self.weights, _f64p),
            _p(self.left_child, _i64p), _p(self.right_child, _i64p), _p(self.lowest_leaf, _i64p),
            _p(self.highest_leaf, _i64p), _p(self.permutation, _i64p), _p(self.means, _f64p),
            _p(self.bandwidth, _f64p), _p(self.bandwidthMin, _f64p), _p(self.bandwidthMax, _f64p))
        if rc != 0:
            raise ValueError(f"okde_make_density failed rc={rc}")

    @classmethod
    def from_arrays(cls, D, N, means, bandwidth, weights, left_child, right_child, permutation):
        self = cls.__new__(cls)
        self.dims, self.num_points = int(D), int(N)
        self.means = np.ascontiguousarray(means, dtype=np.float64)
        self.bandwidth = np.ascontiguousarray(bandwidth, dtype=np.float64)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.left_child = np.ascontiguousarray(left_child, dtype=np.int64)
        self.right_child = np.ascontiguousarray(right_child, dtype=np.int64)
        self.permutation = np.ascontiguousarray(permutation, dtype=np.int64)
        return self

    def _ctree(self):
        return _Tree(self.num_points, self.dims, _p(self.means, _f64p), _p(self.bandwidth, _f64p),
                     _p(self.weights, _f64p), _p(self.left_child, _i64p), _p(self.right_child, _i64p),
                     _p(self.permutation, _i64p))

    # getPoints src/KDE01.jl:91-101
    def get_points(self):
        N, D = self.num_points, self.dims
        perm = self.permutation[N:] - 1
        res = self.centers[N * D:].reshape(N, D).T
        pts = np.zeros((D, N))
        pts[:, perm] = res
        return pts


def nlevels(max_npts: int) -> int:
    return int(lib().okde_nlevels(int(max_npts)))


def rng_sizes(Ndens, ndims, Np, Niter, trees_npts):
    """(K, R, nU_alloc, nN_alloc): per-sample consumption and the reference's allocation sizes
    (src/MSGibbs01.jl:659-662)."""
    L = nlevels(max(trees_npts))
    K = int(lib().okde_randu_per_sample(Ndens, L, Niter))
    R = int(lib().okde_randn_per_sample(ndims, L))
    La = nlevels(max([Np] + list(trees_npts)))
    return K, R, Np * Ndens * (Niter + 2) * La, ndims * Np * (La + 1)


def gibbs1(trees, Np, Niter, randU, randN, addEntropy=True, partialDimMask=None, want_labels=False,
           nthreads=0, manifold=None):
    """prodAppxMSGibbsS body (src/MSGibbs01.jl:680-702): returns (points[D,Np], indices[Ndens,Np][, labels])."""
    M = len(trees)
    D = max(t.dims for t in trees)
    arr = (_Tree * M)(*[t._ctree() for t in trees])
    pts = np.zeros(D * Np)
    ind = np.ones(M * Np, dtype=np.int64)
    randU = np.ascontiguousarray(randU, dtype=np.float64)
    randN = np.ascontiguousarray(randN, dtype=np.float64)
    mask = None
    if partialDimMask is not None:
        mask = np.ascontiguousarray(np.asarray(partialDimMask, dtype=np.uint8).reshape(M, D))
    labels = None
    if want_labels:
        L = nlevels(max(t.num_points for t in trees))
        labels = np.zeros((Np, M, L), dtype=np.int32)
    if manifold is not None:   # per-dimension 0 = Euclidean / 1 = circular (kde_oracle.h okde_gibbs1_manifold)
        man = np.ascontiguousarray(np.asarray(manifold, dtype=np.uint8).reshape(D))
        rc = lib().okde_gibbs1_manifold(M, arr, Np, Niter, _p(pts, _f64p), _p(ind, _i64p), _p(randU, _f64p),
                                        randU.size, _p(randN, _f64p), randN.size, int(addEntropy), D,
                                        None if mask is None else _p(mask, _u8p), _p(man, _u8p),
                                        None if labels is None else _p(labels, _i32p))
    elif nthreads and nthreads > 1:
        rc = lib().okde_gibbs1_omp(M, arr, Np, Niter, _p(pts, _f64p), _p(ind, _i64p), _p(randU, _f64p),
                                   randU.size, _p(randN, _f64p), randN.size, int(addEntropy), D,
                                   None if mask is None else _p(mask, _u8p), int(nthreads))
    else:
        rc = lib().okde_gibbs1(M, arr, Np, Niter, _p(pts, _f64p), _p(ind, _i64p), _p(randU, _f64p),
                               randU.size, _p(randN, _f64p), randN.size, int(addEntropy), D,
                               None if mask is None else _p(mask, _u8p),
                               None if labels is None else _p(labels, _i32p))
    if rc != 0:
        raise IndexError(f"okde_gibbs1 failed rc={rc} (randU/randN too short = Julia BoundsError)")
    out = (pts.reshape(Np, D).T.copy(), ind.reshape(Np, M).T.copy())
    return out + (labels,) if want_labels else out


def fallback_count(reset=False):
    """How often makeFasterSampleIndex! took its `pT < 1e-99` branch (src/MSGibbs01.jl:311-315) since the last reset."""
    return int(lib().okde_fallback_count(int(bool(reset))))


def eval_direct(tree, pos=None, loo=False):
    """evaluateDualTree with FORCE_EVAL_DIRECT (reference src/DualTree01.jl:130-162,303-346): density of
    `tree` at pos (D x Nq); loo=True: at its own points (original order), leave-one-out."""
    t = tree._ctree()
    if loo:
        p = np.zeros(tree.num_points)
        rc = lib().okde_eval_direct(C.byref(t), None, 0, 1, _p(p, _f64p))
    else:
        pos = np.asarray(pos, dtype=np.float64)
        if pos.ndim == 1:
            pos = pos.reshape(1, -1)
        flat = np.ascontiguousarray(pos.T).ravel()
        p = np.zeros(pos.shape[1])
        rc = lib().okde_eval_direct(C.byref(t), _p(flat, _f64p), pos.shape[1], 0, _p(p, _f64p))
    if rc != 0:
        raise ValueError(f"okde_eval_direct rc={rc}")
    return p


def auto_bandwidth(points):
    """The per-dimension LOOCV bandwidth of kde!(points) (reference src/KDE01.jl:3-27); returns (bw[D], n_evals)."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 1:
        pts = pts.reshape(1, -1)
    D, N = pts.shape
    flat = np.ascontiguousarray(pts.T).ravel()
    bw = np.zeros(D)
    ne = C.c_int(0)
    rc = lib().okde_auto_bandwidth(D, N, _p(flat, _f64p), _p(bw, _f64p), C.byref(ne))
    if rc != 0:
        raise ValueError(f"okde_auto_bandwidth rc={rc}")
    return bw, ne.value


def kde_auto(points):
    """kde!(points): LOOCV bandwidth, then the explicit-bandwidth constructor (src/KDE01.jl:24)."""
    bw, _ = auto_bandwidth(points)
    return OracleDensity(points, bw)
