"""(diagnostic build) Launches with level cut-offs 5 and 6 under each ablation flag; differenced PMC counters
give the instruction cost of the parts of one single-row (B = 1) step.  See scripts/step_insts.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kdehip, bench
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS["c3"]
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
plan = kdehip.ProductPlan([kdehip.kde(p, b) for p, b in zip(pts, bws)], precision=prec)
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
torch.cuda.synchronize()
for flags in (0, 1, 2, 4, 8, 14):
    for k in (5, 6):
        plan.set_variant(flags * 1000 + 100 + k)
        plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, None)
        torch.cuda.synchronize()
