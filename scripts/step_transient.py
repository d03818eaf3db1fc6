#!/usr/bin/env python3
"""Per-call timeline of back-to-back kdehip_prod_philox_device calls after an idle period (config 3): host time of each
enqueue, and the stream-side interval between the calls' completion events.  Shows what a SHORT timed region (the
driver's --steps 20 --warmup 5) pays that a long one does not."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, kdehip

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    D, M, N, Nout, Niter, prec = 6, 4, 1000, 2048, 10, 64
    pts, bw = bench.synth_inputs(kdehip, D, M, N, 3)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bw)]
    dd = [kdehip.DeviceDensity(t, device=0) for t in trees]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    outp = [torch.empty(D * Nout, dtype=torch.float64, device=dev) for _ in range(2)]
    outi = [torch.empty(M * Nout, dtype=torch.int64, device=dev) for _ in range(2)]
    spin = len(sys.argv) > 3
    if spin:
        plan = kdehip.ProductPlan(trees, precision=prec, device=0)
    for rep in range(2):
        torch.cuda.synchronize(); time.sleep(idle)
        if spin:   # bench.py's order: 40 ms of resident-plan launches, 5 warm-up calls, a synchronize, then the timed calls
            ts = time.perf_counter()
            while time.perf_counter() - ts < 0.040:
                for _ in range(4):
                    plan.sample_philox_device(Nout, Niter, 1, 0, True, outp[0], outi[0], None, st.cuda_stream)
                torch.cuda.synchronize()
            for i in range(5):
                kdehip.prodAppxMSGibbsS_device(dd, outp[i & 1], outi[i & 1], Np=Nout, Niter=Niter, seed=1, sample_offset=i * Nout,
                                               precision=prec, stream=st.cuda_stream)
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        host = []
        ev[0].record(st)
        for i in range(n):
            t0 = time.perf_counter()
            kdehip.prodAppxMSGibbsS_device(dd, outp[i & 1], outi[i & 1], Np=Nout, Niter=Niter, seed=1, sample_offset=i * Nout,
                                           precision=prec, stream=st.cuda_stream)
            host.append((time.perf_counter() - t0) * 1e6)
            ev[i + 1].record(st)
        torch.cuda.synchronize()
        gaps = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
        print(f"pass {rep}: host us per call:", " ".join(f"{h:.0f}" for h in host))
        print(f"pass {rep}: stream us per call:", " ".join(f"{g:.0f}" for g in gaps))

main()
