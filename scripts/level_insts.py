"""(diagnostic build, KDEHIP_EXPERIMENTS) One launch per level cut-off k = 1..L (variant 100+k); run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace` and difference the
per-launch counters to get instructions per level (scripts/level_insts.sh prints the table)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
CFG = os.environ.get("KDEHIP_LEVEL_CONFIG", "c3")
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[CFG]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)   # builds the tables (first launches)
torch.cuda.synchronize()
for k in range(1, plan.nlevels + 1):
    plan.set_variant(100 + k)
    plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
    torch.cuda.synchronize()
