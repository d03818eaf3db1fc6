"""Many small products: one after another on one stream vs spread over several HIP streams.  A small product
(a few hundred chains) occupies only some of the 256 CUs and is latency bound, so independent products overlap
almost perfectly -- the serving pattern of belief propagation, where one inference step multiplies many
small densities."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kdehip
rng = np.random.default_rng(0)
D, M, N, Np, Niter, nprod = 3, 3, 200, 128, 5, 64
dev = torch.device("cuda", 0)
plans, outs = [], []
for p in range(nprod):
    trees = []
    for j in range(M):
        pts = rng.standard_normal((D, N)) + rng.uniform(-1, 1, size=(D, 1))
        trees.append(kdehip.kde(pts, [0.3]))
    plans.append(kdehip.ProductPlan(trees))
    outs.append((torch.zeros(Np * D, dtype=torch.float64, device=dev), torch.zeros(Np * M, dtype=torch.int64, device=dev)))
for pl, (P, I) in zip(plans, outs):
    pl.sample_philox_device(Np, Niter, 1, 0, True, P, I, None, None)  # builds the tables
torch.cuda.synchronize()
def run(nstreams, reps=5):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for i, (pl, (P, I)) in enumerate(zip(plans, outs)):
            pl.sample_philox_device(Np, Niter, 2, 0, True, P, I, None, streams[i % nstreams].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for ns in (1, 2, 4, 8, 16, 32):
    ms = run(ns)
    print(f"{nprod} products of {Np} chains ({D}-D, {M} x {N} pts, Niter={Niter}) on {ns:2d} stream(s): {ms:7.2f} ms "
          f"= {nprod*Np/ms*1e3:9.0f} samples/s, {ms/nprod*1e3:6.1f} us per product")
