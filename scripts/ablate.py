"""(needs the diagnostic build, scripts/stamps.sh all) Ablation timing: kernel time for levels <= k with pieces of a step disabled (variant flags)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c3"]
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
    return a.elapsed_time(b) / reps * 1e3
for k in (6, 10):
    print(f"--- levels <= {k}")
    for name, fl in [("full", 0), ("no philox", 1), ("no set_particle", 2), ("no product_dim", 4), ("no draw", 8),
                     ("no philox+setp+prod", 7), ("nothing (loop only)", 15)]:
        print(f"{name:24s} {run(fl * 1000 + 100 + k):8.1f} us")
