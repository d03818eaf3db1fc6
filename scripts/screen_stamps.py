"""Phase stamps of a SCREENED step (diagnostic build: scripts/dev_lean.sh stamps -DKDEHIP_SCREEN_STAMPS -DKDEHIP_X_SCREEN_NOKEPT),
one wavefront (workgroup 3, wavefront 5), cycles per step:  KDEHIP_LIB=.../libkdehip_stamps.so python scripts/screen_stamps.py"""
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
names = ["product+uniform", "screen draw (all)", "fp64 repeat", "adopt (L2)", "  rows", "  scans+lane", "  second pass", "-"]
steps = M * (Niter + 1)
for lvl in (9, 10):
    plan.set_variant(200 + lvl)
    for _ in range(3):
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    assert lib.kdehip_debug_read_screen_stamps(out) == 0
    v = [out[k] / steps for k in range(7)]
    print(f"level {lvl:2d} (cycles per step, total {sum(v[:4]):.0f}): " + " | ".join(f"{n.strip()} {x:6.0f}" for n, x in zip(names, v)))

rnames = ["product+uniform", "broadcasts", "rows+select", "adopt", "between steps"]
for lvl in (5, 6, 7, 8):
    plan.set_variant(300 + lvl)
    for _ in range(3):
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    assert lib.kdehip_debug_read_screen_stamps(out) == 0
    v = [out[8 + k] / steps for k in range(5)]
    print(f"level {lvl:2d} resident fp64 step (cycles per step, total {sum(v):.0f}): " + " | ".join(f"{n} {x:6.0f}" for n, x in zip(rnames, v)))
