#!/usr/bin/env python3
"""Thread-safety soak: several host threads call the blocking entry points concurrently on one device (one-shot
products with device Philox and with caller streams, products of densities kept in HBM -- shared handles --, evaluation,
LOOCV bandwidth, `kde!(points)` with its tree built on the pooled host builder under the search); every result
must equal the single-threaded one.   python scripts/soak_threads.py [threads] [calls per thread]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdehip  # noqa: E402
from oracle import oracle  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(5)
jobs = []
for c in range(24):   # a pool of prepared problems with their single-threaded answers
    D = int(rng.integers(1, 7))
    M = int(rng.integers(2, 6))
    Ns = [int(rng.choice([20, 300, 1000, 2500])) for _ in range(M)]
    Np, Niter = int(rng.choice([7, 64, 500])), int(rng.integers(0, 3))
    trees = [kdehip.kde(rng.standard_normal((D, n)), rng.uniform(0.1, 0.6, size=D)) for n in Ns]
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, Ns)
    randU, randN = rng.random(nU), rng.standard_normal(nN)
    pos = rng.standard_normal((D, 300))
    x = rng.standard_normal((D, int(rng.choice([400, 1500, 2500]))))
    jobs.append(dict(trees=trees, dd=[kdehip.DeviceDensity(t) for t in trees], Np=Np, Niter=Niter, randU=randU, randN=randN,
                     pos=pos, x=x, seed=c,
                     a=kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, seed=c),
                     b=kdehip.prodAppxMSGibbsS(None, trees, None, None, Niter=Niter, Np=Np, randU=randU, randN=randN),
                     e=trees[0](pos), bw=kdehip.auto_bandwidth(x), kd=kdehip.kde_auto(x, overlap=False)))
for j in jobs[:6]:   # (a few of them: the resident `*` and its host twin)
    j["mul"] = kdehip.mul(j["trees"], seed=j["seed"]) if max(t.bt.num_points for t in j["trees"]) <= 1000 else None


def worker(t):
    r = np.random.default_rng(1000 + t)
    bad = 0
    for _ in range(ncalls):
        j = jobs[int(r.integers(0, len(jobs)))]
        kind = int(r.integers(0, 10))
        if kind == 0:
            got = kdehip.prodAppxMSGibbsS(None, j["trees"], None, None, Niter=j["Niter"], Np=j["Np"], seed=j["seed"])
            ok = np.array_equal(got[0], j["a"][0]) and np.array_equal(got[1], j["a"][1])
        elif kind == 1:
            got = kdehip.prodAppxMSGibbsS(None, j["trees"], None, None, Niter=j["Niter"], Np=j["Np"], randU=j["randU"], randN=j["randN"])
            ok = np.array_equal(got[0], j["b"][0]) and np.array_equal(got[1], j["b"][1])
        elif kind == 2:
            ok = np.array_equal(j["trees"][0](j["pos"]), j["e"])
        elif kind == 4:   # densities resident in HBM, handles shared by all threads; the plan queue is shared state
            got = kdehip.prodAppxMSGibbsS_resident(j["dd"], Np=j["Np"], Niter=j["Niter"], seed=j["seed"])
            ok = np.array_equal(got[0], j["a"][0]) and np.array_equal(got[1], j["a"][1])
        elif kind == 6:   # asynchronous products of resident densities: three in a row, prepared under each other
            import torch
            dev = torch.device("cuda", 0)
            D, M, Np = j["trees"][0].bt.dims, len(j["trees"]), j["Np"]
            outs = [(torch.zeros(D * Np, dtype=torch.float64, device=dev), torch.zeros(M * Np, dtype=torch.int64, device=dev))
                    for _ in range(3)]
            torch.cuda.synchronize()
            st = torch.cuda.Stream(device=dev)
            for P, I in outs:
                kdehip.prodAppxMSGibbsS_device(j["dd"], P, I, Np=Np, Niter=j["Niter"], seed=j["seed"], stream=st.cuda_stream)
            st.synchronize()
            ok = all(np.array_equal(P.cpu().numpy().reshape(Np, D).T, j["a"][0]) and
                     np.array_equal(I.cpu().numpy().reshape(Np, M).T, j["a"][1]) for P, I in outs)
        elif kind == 7:   # `*` on resident densities (kdehip_mul_device): product, bandwidth search, tree, upload -- per thread
            if j.get("mul") is None:
                continue
            with kdehip.mul_device(j["dd"], seed=j["seed"]) as out:
                got = out.download()
            ok = all(np.array_equal(getattr(got, f), getattr(j["mul"], f)) for f in ("means", "bandwidth")) and \
                all(np.array_equal(getattr(got.bt, f), getattr(j["mul"].bt, f)) for f in ("weights", "left_child", "permutation"))
        elif kind == 8:   # several products in one batched call, on a stream of the thread's own
            import torch
            dev = torch.device("cuda", 0)
            picks = [jobs[int(r.integers(0, len(jobs)))] for _ in range(int(r.integers(2, 6)))]
            prods, want = [], []
            for q in picks:
                D, M, Np = q["trees"][0].bt.dims, len(q["trees"]), q["Np"]
                P = torch.zeros(D * Np, dtype=torch.float64, device=dev)
                I = torch.zeros(M * Np, dtype=torch.int64, device=dev)
                prods.append(dict(trees=q["dd"], d_points=P, d_indices=I, Np=Np, Niter=q["Niter"], seed=q["seed"]))
                want.append((q["a"], Np, D, M))
            torch.cuda.synchronize()
            st = torch.cuda.Stream(device=dev)
            kdehip.prodAppxMSGibbsS_batch(prods, stream=st.cuda_stream)
            st.synchronize()
            ok = all(np.array_equal(pr["d_points"].cpu().numpy().reshape(Np, D).T, a[0]) and
                     np.array_equal(pr["d_indices"].cpu().numpy().reshape(Np, M).T, a[1]) for pr, (a, Np, D, M) in zip(prods, want))
        elif kind == 9:   # several `*` in ONE call (kdehip_mul_device_batch): shared LOOCV launches, pooled trees, shared block
            picks = [q for q in jobs[:6] if q.get("mul") is not None]
            if len(picks) < 2:
                continue
            picks = [picks[int(r.integers(0, len(picks)))] for _ in range(int(r.integers(2, 5)))]
            outs = kdehip.mul_device_batch([q["dd"] for q in picks], seeds=[q["seed"] for q in picks])
            ok = True
            for q, out in zip(picks, outs):
                got = out.download()
                ok = ok and all(np.array_equal(getattr(got, f), getattr(q["mul"], f)) for f in ("means", "bandwidth")) and \
                    all(np.array_equal(getattr(got.bt, f), getattr(q["mul"].bt, f)) for f in ("weights", "left_child", "permutation"))
                out.close()
        elif kind == 5:   # kde!(points): the worker pool of the host tree builder is shared by every caller
            got = kdehip.kde_auto(j["x"], overlap=bool(r.integers(0, 2)))
            ok = all(np.array_equal(getattr(got, f), getattr(j["kd"], f)) for f in ("means", "bandwidth")) and \
                all(np.array_equal(getattr(got.bt, f), getattr(j["kd"].bt, f)) for f in ("centers", "ranges", "weights", "left_child", "permutation"))
        else:
            ok = np.array_equal(kdehip.auto_bandwidth(j["x"]), j["bw"])
        bad += not ok
    return bad


t0 = time.time()
with ThreadPoolExecutor(max_workers=nthreads) as ex:
    bad = sum(ex.map(worker, range(nthreads)))
print(f"{nthreads} threads x {ncalls} calls: {bad} wrong results, {time.time()-t0:.0f} s")
sys.exit(1 if bad else 0)
