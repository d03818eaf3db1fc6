"""Reads the in-kernel phase stamps of the diagnostic build (one wavefront, one level at a time)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
from kdehip import _lib
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
lib = C.CDLL(_lib.LIB_PATH)
names = ["LOO product+uniform", "draw total", "  pass-1 rows", "  scan+select lane", "  pass 2 (eval+scan)", "set_particle", "barrier wait", "-"]
steps = M * (Niter + 1)
for lvl in (1, 4, 6, 7, 8, 9, 10):
    plan.set_variant((lvl << 8) * 1000)
    for _ in range(2):
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
    out = (C.c_ulonglong * 16)()
    assert lib.kdehip_debug_read_stamps(out) == 0
    v = [out[k] / steps for k in range(7)]
    print(f"level {lvl:2d} (cycles per step, one wavefront): " + " | ".join(f"{n.strip()} {x:7.0f}" for n, x in zip(names, v)))
