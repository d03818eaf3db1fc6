#!/usr/bin/env python3
""""host tree alone 2-3.5 ms" in scripts/chain_timing.py against 0.21-0.37 ms in a fresh process.
One controlled run: the SAME build (kdehip.kde of 6 x 2048 points, the library's pooled host builder) timed

  (a) in this process before torch has done anything,
  (b) after torch has run a CPU op that uses its intra-op pool (P.cpu() of a device tensor + a reduction) and 20 GPU calls,
  (c) the same after torch.set_num_threads(1),

each as the median / max of 50 builds.  Run it three times: plain, with OMP_WAIT_POLICY=PASSIVE, and with
KDEHIP_HOST_THREADS=0 (the library's builder serial: no contention for cores with torch's workers)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kdehip  # noqa: E402
from tests.helpers import silverman_bw, synth_mixture  # noqa: E402


def builds(pts, ks, n=50):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        kdehip.kde(pts, ks)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    return f"median {np.median(ts):.3f} ms, max {ts.max():.3f} ms, min {ts.min():.3f} ms"


def main():
    D, N = 6, 2048
    pts = synth_mixture(np.random.default_rng(1), D, N)
    ks = silverman_bw(pts)
    print(f"OMP_WAIT_POLICY={os.environ.get('OMP_WAIT_POLICY')} KDEHIP_HOST_THREADS={os.environ.get('KDEHIP_HOST_THREADS')}")
    print("(a) fresh process:                        ", builds(pts, ks))
    import torch
    print("    torch threads:", torch.get_num_threads())
    dev = torch.device("cuda", 0)
    trees = [kdehip.kde(synth_mixture(np.random.default_rng(10 + j), D, N), ks) for j in range(4)]
    dd = [kdehip.DeviceDensity(t) for t in trees]
    P = torch.zeros(D * N, dtype=torch.float64, device=dev)
    I = torch.zeros(4 * N, dtype=torch.int64, device=dev)
    for k in range(20):
        kdehip.prodAppxMSGibbsS_device(dd, P, I, Np=N, Niter=5, seed=k)
        x = P.cpu()
        (x.reshape(D, N) @ x.reshape(D, N).T).sum()   # an intra-op parallel region on the host
    print("(b) after torch CPU ops + 20 GPU calls:   ", builds(pts, ks))
    for k in range(5):   # alternate: a torch CPU op right before every build (what chain_timing.py's loop does)
        x = P.cpu()
        (x.reshape(D, N) @ x.reshape(D, N).T).sum()
        t0 = time.perf_counter()
        kdehip.kde(pts, ks)
        print(f"    build right behind a torch CPU op: {(time.perf_counter() - t0) * 1e3:.3f} ms")
    torch.set_num_threads(1)
    for k in range(3):
        x = P.cpu()
        (x.reshape(D, N) @ x.reshape(D, N).T).sum()
    print("(c) after torch.set_num_threads(1):       ", builds(pts, ks))
    for k in range(5):
        x = P.cpu()
        (x.reshape(D, N) @ x.reshape(D, N).T).sum()
        t0 = time.perf_counter()
        kdehip.kde(pts, ks)
        print(f"    build right behind a torch CPU op: {(time.perf_counter() - t0) * 1e3:.3f} ms")
    for d in dd:
        d.close()


if __name__ == "__main__":
    main()
