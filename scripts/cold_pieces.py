"""Where a fresh process spends its time before the first product is back: dlopen of libkdehip.so, HIP initialisation
(device count, context), the first product (code objects of ITS dimension count are unpacked and loaded at the first
launch of one of their kernels), the first product of ANOTHER dimension count (code loading alone), steady state."""
import ctypes, json, os, sys, time
t0 = time.perf_counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "kerneldensityestimate.jl_amd", "libkdehip.so"))
t1 = time.perf_counter()
lib.kdehip_device_count.restype = ctypes.c_int
n = lib.kdehip_device_count()          # hipGetDeviceCount: runtime initialisation
t2 = time.perf_counter()
hip = ctypes.CDLL("libamdhip64.so")   # (already mapped: libkdehip.so links it)
hip.hipSetDevice(0)
hip.hipFree(None)                      # context creation
t3 = time.perf_counter()
sys.path.insert(0, root)
import numpy as np
import kdehip
import bench
t4 = time.perf_counter()
out = {"dlopen_ms": (t1 - t0) * 1e3, "hip_runtime_init_ms": (t2 - t1) * 1e3, "context_ms": (t3 - t2) * 1e3,
       "python_imports_ms": (t4 - t3) * 1e3}


def product(cfg, seed):
    D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cfg]
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    t = time.perf_counter()
    kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Nout, seed=seed, precision=prec)
    return (time.perf_counter() - t) * 1e3


order = sys.argv[1:] or ["c3", "c3", "c2", "c2", "c4", "c4"]
for k, c in enumerate(order):
    out[f"{k}_{c}_ms"] = product(c, k)
print(json.dumps(out))
