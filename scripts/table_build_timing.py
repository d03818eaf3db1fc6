"""One-time cost of the conditional tables (built lazily by the first run with >= 256 chains)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
for cfg in ("c2", "c3", "c4"):
    D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cfg]
    pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
    trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
    dev = torch.device("cuda", 0)
    P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    ts = []
    for rep in range(3):
        plan = kdehip.ProductPlan(trees, precision=prec)
        torch.cuda.synchronize()
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(st)
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)   # builds tables + samples
        b.record(st)
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)   # samples only
        c.record(st); torch.cuda.synchronize()
        ts.append((a.elapsed_time(b), b.elapsed_time(c)))
        pb = plan.packed_bytes
        plan.close()
    first, steady = min(t[0] for t in ts), min(t[1] for t in ts)
    print(f"{cfg}: first run {first*1e3:.0f} us, steady run {steady*1e3:.0f} us -> table build ~{(first-steady)*1e3:.0f} us; plan bytes {pb}")
