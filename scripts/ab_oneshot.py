#!/usr/bin/env python3
"""Interleaved A/B of the one-shot entry point (kdehip_prod_philox: pack + upload + kernels + copy back, host buffers in
and out) between builds of the library, in ONE process on ONE device.
    python scripts/ab_oneshot.py --libs kerneldensityestimate.jl_amd/libkdehip.so /tmp/other.so --configs c3 c2"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--configs", nargs="+", default=["c3"])
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--calls", type=int, default=12)
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime)
    import bench
    import kdehip
    from kdehip import _lib
    libs = []
    for path in args.libs:
        lib = C.CDLL(os.path.abspath(path))
        for name, (res, at) in _lib.SIGNATURES.items():
            if hasattr(lib, name):
                getattr(lib, name).restype = res
                getattr(lib, name).argtypes = at
        libs.append(lib)
    for cname in args.configs:
        D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cname]
        pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
        trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
        arr = (_lib.CDensity * M)(*[t._cstruct() for t in trees])
        P = np.zeros(D * Nout)
        I = np.zeros(M * Nout, dtype=np.int64)

        def call(lib, seed):
            rc = lib.kdehip_prod_philox(M, arr, Nout, Niter, _lib.ptr(P, _lib.f64p), _lib.ptr(I, _lib.i64p),
                                        C.c_uint64(seed), 1, D, None, prec, 0, 1, None)
            assert rc == 0, rc
        ref = None
        for lib in libs:
            call(lib, 5)
            call(lib, 5)
            got = (P.copy(), I.copy())
            if ref is None:
                ref = got
            else:
                print(f"  {cname}: identical results: {np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])}")
        times = [[] for _ in libs]
        for r in range(args.rounds):
            for i, lib in enumerate(libs):
                t0 = time.perf_counter()
                for k in range(args.calls):
                    call(lib, 100 + k)
                times[i].append((time.perf_counter() - t0) / args.calls * 1e3)
        for path, t in zip(args.libs, times):
            print(f"{cname} one-shot {os.path.basename(path):28s} median {np.median(t):8.4f} ms  min {np.min(t):8.4f} ms")


if __name__ == "__main__":
    main()
