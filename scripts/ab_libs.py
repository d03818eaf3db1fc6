#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libkdehip.so in ONE process on ONE device (methodology: perf deltas
come from interleaved rounds, never from separate invocations or boxes).

    python scripts/ab_libs.py --libs kerneldensityestimate.jl_amd/libkdehip.so kerneldensityestimate.jl_amd/libkdehip_r01.so \
        --configs c3 c5 --rounds 7 --steps 10

Each library gets its own resident plan of the same densities; a round runs `steps` launches per library, timed
with HIP events on the launch stream; the table shows median and min ms per launch."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--configs", nargs="+", default=["c3"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--nout", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--prec", type=int, default=0, help="override the configuration's precision (32 / 64)")
    args = ap.parse_args()
    import torch
    import bench
    import kdehip
    from kdehip import _lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
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
        if args.nout:
            Nout = args.nout
        if args.prec:
            prec = args.prec
        pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
        trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
        arr = (_lib.CDensity * M)(*[t._cstruct() for t in trees])
        plans = []
        for lib in libs:
            h = C.c_void_p()
            rc = lib.kdehip_product_create(C.byref(h), M, arr, D, None, prec, 0)
            assert rc == 0, (rc, lib.kdehip_last_error())
            if args.variant:
                lib.kdehip_product_set_variant(h, args.variant)
            plans.append(h)
        d_pts = torch.zeros(D * Nout, dtype=torch.float64, device=dev)
        d_ind = torch.zeros(M * Nout, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev)

        def run(i, step):
            rc = libs[i].kdehip_product_sample_philox(plans[i], Nout, Niter, C.c_uint64(20260101), step * Nout, 1,
                                                      C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_ind.data_ptr()), None,
                                                      C.c_void_p(stream.cuda_stream))
            assert rc == 0, rc
        ref = None
        for i in range(len(libs)):   # warm-up (table build) + results must agree between builds
            run(i, 0)
            run(i, 0)
            torch.cuda.synchronize()
            got = (d_pts.cpu().numpy().copy(), d_ind.cpu().numpy().copy())
            if ref is None:
                ref = got
            else:
                same = np.array_equal(ref[1], got[1])
                print(f"  {cname}: lib {i} labels identical to lib 0: {same}; max |dx| = {np.abs(ref[0] - got[0]).max():.3g}")
        times = [[] for _ in libs]
        for r in range(args.rounds):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for s in range(args.steps):
                    run(i, 1 + r * args.steps + s)
                e1.record(stream)
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / args.steps)
        for i, path in enumerate(args.libs):
            t = np.array(times[i])
            print(f"{cname} Nout={Nout} {os.path.basename(path):28s} median {np.median(t):8.4f} ms  min {t.min():8.4f} ms  "
                  f"({Nout / np.median(t) / 1e3:.3f} M samples/s)")
        for i, lib in enumerate(libs):
            lib.kdehip_product_destroy(plans[i])


if __name__ == "__main__":
    main()
