"""(needs a -DKDEHIP_EXPERIMENTS development build of gibbs_lean.hip, scripts/dev_lean.sh) Cumulative kernel time when
the anneal stops after level k (variant 100 + k), for a launch geometry given by the thousands digit of the variant
(0 default, 6 / 8 sixteen / eight one-wavefront chains per workgroup):
    KDEHIP_LIB=.../libkdehip_x.so python scripts/level_timing2.py c3 [nout] geometry [geometry ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
cfg = sys.argv[1]
rest = [int(a) for a in sys.argv[2:]]
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cfg]
if rest and rest[0] > 9:
    Nout = rest.pop(0)
geos = rest or [0]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev)
def run(v, reps=8):
    plan.set_variant(v)
    for _ in range(2): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
    b.record(st); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print(cfg, "Nout", Nout, "geometries", geos, [plan.launch_geometry(Nout)])
prev = {g: 0.0 for g in geos}
for k in range(1, plan.nlevels + 1):
    row = f"levels<= {k:2d}:"
    for g in geos:
        t = run(g * 1000 + 100 + k)
        row += f"   g{g}: {t:8.1f} (+{t - prev[g]:7.1f})"
        prev[g] = t
    print(row)
