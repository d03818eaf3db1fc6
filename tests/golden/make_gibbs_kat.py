"""Generates the deterministic known-answer fixtures tests/golden/gibbs_kat_*.npz.

The reference holds no golden vector for the Gibbs arithmetic and Julia is not available to run it,
so these vectors come from the build's own CPU oracle (oracle/kde_oracle.c) -- they pin the oracle
against regressions and let the GPU box check the HIP path against committed numbers.  Inputs are
closed-form (Weyl sequences), so nothing depends on a library RNG.  Run from the repo root:
    python tests/golden/make_gibbs_kat.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests.helpers import kat_streams, weyl_normal  # noqa: E402


def inputs(D, M, N):
    pts, bws = [], []
    for j in range(M):
        p = np.empty((D, N))
        for d in range(D):
            centre = 0.4 * ((j * 7 + d * 3) % 5 - 2)
            p[d] = centre + 0.6 * weyl_normal(N, np.sqrt(2.0 + j + 0.1 * d) % 1.0, np.sqrt(3.0 + d + 0.1 * j) % 1.0)
        pts.append(p)
        bws.append(p.std(axis=1, ddof=1) * (4.0 / ((D + 2.0) * N)) ** (1.0 / (D + 4.0)))
    return pts, bws


def make(name, D, M, N, Np, Niter):
    pts, bws = inputs(D, M, N)
    trees = [oracle.OracleDensity(p, b) for p, b in zip(pts, bws)]
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = kat_streams(nU, nN)
    p_e, i_e = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=True)
    p_n, i_n = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=False)
    assert np.array_equal(i_e, i_n)
    out = os.path.join(ROOT, "tests", "golden", f"gibbs_kat_{name}.npz")
    np.savez_compressed(out, D=D, M=M, N=N, Np=Np, Niter=Niter, points=np.stack(pts), bw=np.stack(bws),
                        indices=i_e.astype(np.int32), pGM_entropy=p_e, pGM_mean=p_n)
    print(out, os.path.getsize(out), "bytes")


def dump_text(name, D, M, N, Np, Niter, out_dir):
    """Plain-text inputs + expected outputs of one case for oracle/julia_crosscheck.jl (the definitive parity check
    against the real reference; needs a Julia installation, which this image does not have)."""
    os.makedirs(out_dir, exist_ok=True)
    pts, bws = inputs(D, M, N)
    trees = [oracle.OracleDensity(p, b) for p, b in zip(pts, bws)]
    K, R, nU, nN = oracle.rng_sizes(M, D, Np, Niter, [N] * M)
    randU, randN = kat_streams(nU, nN)
    p_e, i_e = oracle.gibbs1(trees, Np, Niter, randU, randN, addEntropy=True)
    np.savetxt(os.path.join(out_dir, "meta.txt"), np.array([[D, M, N, Np, Niter]]), fmt="%d")
    for j in range(M):
        np.savetxt(os.path.join(out_dir, f"points_{j + 1}.txt"), pts[j], fmt="%.17g")
        np.savetxt(os.path.join(out_dir, f"bw_{j + 1}.txt"), bws[j], fmt="%.17g")
    np.savetxt(os.path.join(out_dir, "randU.txt"), randU, fmt="%.17g")
    np.savetxt(os.path.join(out_dir, "randN.txt"), randN, fmt="%.17g")
    np.savetxt(os.path.join(out_dir, "indices.txt"), i_e, fmt="%d")
    np.savetxt(os.path.join(out_dir, "pGM.txt"), p_e, fmt="%.17g")
    print("wrote", out_dir, f"({name})")


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--dump-text":
        dump_text("c1", 1, 2, 100, 100, 5, os.path.join(sys.argv[2], "c1"))
        dump_text("c2", 2, 3, 200, 256, 5, os.path.join(sys.argv[2], "c2"))
        dump_text("d6", 6, 4, 300, 64, 3, os.path.join(sys.argv[2], "d6"))
        # BASELINE config 3's full shape (6-D, 4 x 1000 points, Niter 10), 64 chains: the headline configuration is the
        # first thing a maintainer with Julia checks (1.2 MB of text)
        dump_text("c3", 6, 4, 1000, 64, 10, os.path.join(sys.argv[2], "c3"))
        sys.exit(0)
    make("c1", 1, 2, 100, 100, 5)   # BASELINE config 1
    make("c2", 2, 3, 200, 256, 5)   # BASELINE config 2
    make("d6", 6, 4, 300, 64, 3)    # config-3 shape, reduced
