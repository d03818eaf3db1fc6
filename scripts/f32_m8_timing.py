import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench, kdehip
D, M, N, Nout, Niter = 3, 8, 5000, 2048, 10
pts, bws = bench.synth_inputs(kdehip, D, M, N, 4)
trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
dev = torch.device("cuda", 0)
P = torch.zeros(Nout * D, dtype=torch.float64, device=dev); I = torch.zeros(Nout * M, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev)
for prec in (32, 64):
    with kdehip.ProductPlan(trees, precision=prec) as plan:
        for v in (0, 38):
            plan.set_variant(v)
            for _ in range(3): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(10): plan.sample_philox_device(Nout, Niter, 1, 0, True, P, I, None, st.cuda_stream)
            b.record(st); torch.cuda.synchronize()
            print("c4 shape, fp%d, variant %d (%s): %.3f ms" % (prec, v, plan.kernel_name(Nout), a.elapsed_time(b) / 10))
