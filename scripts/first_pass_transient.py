#!/usr/bin/env python3
"""What makes the FIRST pass of a process through kdehip_prod_philox_device slower (config 3: ~525 us per call decaying to
~485 over the first ~40 calls, also behind a 40 ms spin-up; scripts/step_transient.py)?  Variants, each in a fresh process:
  plain            spin-up (resident launches), then 90 calls
  warm-small N     the same, preceded by N calls of a SMALL product (config 2's shape): warms the HIP runtime's launch /
                   signal / event pools without touching this product's memory blocks
  warm-alloc       preceded by allocating and freeing ten 4 MB blocks through the library's cache path (10 one-shot products)
Prints the stream-side microseconds per call in groups of 10."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, kdehip


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
    nwarm = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    D, M, N, Nout, Niter, prec = 6, 4, 1000, 2048, 10, 64
    pts, bw = bench.synth_inputs(kdehip, D, M, N, 3)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bw)]
    dd = [kdehip.DeviceDensity(t, device=0) for t in trees]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    outp = [torch.empty(D * Nout, dtype=torch.float64, device=dev) for _ in range(2)]
    outi = [torch.empty(M * Nout, dtype=torch.int64, device=dev) for _ in range(2)]
    plan = kdehip.ProductPlan(trees, precision=prec, device=0)
    if mode == "warm-small":
        p2, b2 = bench.synth_inputs(kdehip, 2, 3, 200, 2)
        d2 = [kdehip.DeviceDensity(kdehip.kde(p, b), device=0) for p, b in zip(p2, b2)]
        o2p = torch.empty(2 * 256, dtype=torch.float64, device=dev)
        o2i = torch.empty(3 * 256, dtype=torch.int64, device=dev)
        for i in range(nwarm):
            kdehip.prodAppxMSGibbsS_device(d2, o2p, o2i, Np=256, Niter=5, seed=1, sample_offset=i * 256, stream=st.cuda_stream)
        torch.cuda.synchronize()
    if mode == "warm-alloc":
        for i in range(10):
            kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=1, Np=64, seed=i)
    torch.cuda.synchronize(); time.sleep(1.0)
    ts = time.perf_counter()
    while time.perf_counter() - ts < 0.040:
        for _ in range(4):
            plan.sample_philox_device(Nout, Niter, 1, 0, True, outp[0], outi[0], None, st.cuda_stream)
        torch.cuda.synchronize()
    n = 90
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(st)
    for i in range(n):
        kdehip.prodAppxMSGibbsS_device(dd, outp[i & 1], outi[i & 1], Np=Nout, Niter=Niter, seed=1, sample_offset=i * Nout,
                                       precision=prec, stream=st.cuda_stream)
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    gaps = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print(mode, nwarm, "us per call, means of 10:", " ".join(f"{np.mean(gaps[k:k + 10]):.0f}" for k in range(0, n, 10)))


main()
