"""Per-link time of a chain of `*` (reference src/MSGibbs01.jl:707-726) at a configuration's shape: the host route
(`mul`: one-shot product with host buffers, then `kde!(pGM)` from the host's copy) against the resident route
(`mul_device`: product, LOOCV search and new density stay in HBM).  scripts/chain_timing.py [config] [links] [npts]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import kdehip

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
links = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D, M, N, Nout, Niter, prec, cid = bench.CONFIGS[cfg]
if len(sys.argv) > 3:
    N = int(sys.argv[3])
pts, bws = bench.synth_inputs(kdehip, D, M, N, cid)
trees = [kdehip.kde(p, b) for p, b in zip(pts, bws)]
dd = [kdehip.DeviceDensity(t) for t in trees]


def host_chain():
    h = trees[0]
    for k in range(links):
        h = kdehip.mul([h] + trees[1:], seed=1000 + k)
    return h


def device_chain():
    d, made = dd[0], []
    for k in range(links):
        d = kdehip.mul_device([d] + dd[1:], seed=1000 + k)
        made.append(d)
    return d, made


def T(f, n=3):
    f()
    t = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t) / n * 1e3, r


t_host, h = T(host_chain)
t_dev, (d, made) = T(device_chain)
got = d.download()
same = all(np.array_equal(getattr(got.bt, n), getattr(h.bt, n)) for n in ("centers", "ranges", "weights", "left_child", "right_child", "permutation")) \
    and all(np.array_equal(getattr(got, n), getattr(h, n)) for n in ("means", "bandwidth"))
print(f"{cfg}: `*` of {M} densities x {N} points in {D}-D, chain of {links}: host route {t_host / links:.3f} ms per link | "
      f"resident route {t_dev / links:.3f} ms per link | final densities identical: {same}")

# the pieces of one resident link
import ctypes as C
import torch
Np = int(round(float(np.mean([t.bt.num_points for t in trees]))))
P = torch.zeros(D * Np, dtype=torch.float64, device="cuda:0")
I = torch.zeros(M * Np, dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()


def prod():
    kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=Np, Niter=5, seed=7)
    torch.cuda.synchronize()


def kde_dev():
    x = kdehip.DeviceDensity.from_device_points(P, D, Np)
    x.close()


t_prod, _ = T(prod, 20)
t_kde, _ = T(kde_dev, 20)
pg = P.cpu().numpy().reshape(Np, D).T
t_bw, _ = T(lambda: kdehip.auto_bandwidth(pg), 20)
t_tree, _ = T(lambda: kdehip.kde(pg, np.full(D, 0.3)), 20)
t_auto, _ = T(lambda: kdehip.kde_auto(pg), 20)
print(f"pieces at Np = {Np}: product on resident densities (enqueue + wait) {t_prod:.3f} ms | kde!(pGM) from device points {t_kde:.3f} ms "
      f"(LOOCV search alone from host points {t_bw:.3f}, host tree alone {t_tree:.3f}, kde_auto from host points {t_auto:.3f})")
os.environ["KDEHIP_TIMING"] = "1"
