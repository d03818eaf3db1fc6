"""(needs the diagnostic build, scripts/stamps.sh all) Timing experiment: cumulative kernel time when the anneal stops after level k (variant 100+k)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev)
def run(v, reps=10):
    plan.set_variant(v)
    for _ in range(3): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
    b.record(st); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
prev = 0.0
for k in range(1, plan.nlevels + 1):
    t = run(100 + k)
    print(f"levels<= {k:2d}: {t*1e3:8.1f} us   (+{(t-prev)*1e3:7.1f})")
    prev = t
print("full:", run(0) * 1e3, "us; all-global variant 1:", run(1) * 1e3, "us")
