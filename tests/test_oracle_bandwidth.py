"""CPU checks of the oracle's LOOCV-bandwidth and direct-evaluation restatement.  This row IS pinned by
a reference golden: `kde!(x)` on test1Dlcv100.txt must reproduce test1Dlcv100Result.txt
(UnitTest1Dlcv01, reference test/runtests.jl:104-116, tolerance 1e-4)."""
import os

import numpy as np

from oracle import oracle
from tests.helpers import check_density_against_golden, parse_mat_print_kde


def test_golden_1d_lcv100_full_loocv(golden_dir):
    gold = parse_mat_print_kde(os.path.join(golden_dir, "test1Dlcv100Result.txt"))
    x = np.loadtxt(os.path.join(golden_dir, "test1Dlcv100.txt")).ravel()
    d = oracle.kde_auto(x)
    check_density_against_golden(d, gold, 1e-4)
    bw, nev = oracle.auto_bandwidth(x)
    assert abs(bw[0] - np.sqrt(gold["bandwidth"][100])) < 1e-6 and 10 <= nev <= 40


def test_direct_evaluation_matches_closed_form():
    rng = np.random.default_rng(1)
    D, N, Nq = 3, 200, 50
    pts, w = rng.standard_normal((D, N)), rng.uniform(0.5, 1.5, N)
    bw = np.array([0.3, 0.5, 0.2])
    d = oracle.OracleDensity(pts, bw, w)
    pos = rng.standard_normal((D, Nq))
    wn = w / w.sum()
    diff = pos[:, :, None] - pts[:, None, :]
    K = np.exp(-0.5 * ((diff / bw[:, None, None]) ** 2).sum(axis=0)) / ((2 * np.pi) ** (D / 2) * bw.prod())
    assert np.allclose(oracle.eval_direct(d, pos), K @ wn, rtol=1e-12)
    # leave-one-out at the density's own points, original order (src/DualTree01.jl:141,335)
    diff = pts[:, :, None] - pts[:, None, :]
    K = np.exp(-0.5 * ((diff / bw[:, None, None]) ** 2).sum(axis=0)) / ((2 * np.pi) ** (D / 2) * bw.prod())
    np.fill_diagonal(K, 0.0)
    assert np.allclose(oracle.eval_direct(d, loo=True), (K @ wn) / (1 - wn), rtol=1e-12)


def test_auto_bandwidth_is_sane_in_higher_dims():
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((3, 150)) * np.array([[1.0], [0.2], [5.0]])
    bw, _ = oracle.auto_bandwidth(pts)
    sil = pts.std(axis=1, ddof=1) * (4.0 / (3 * 150)) ** 0.2  # 1-D Silverman per marginal
    assert np.all(bw > 0.1 * sil) and np.all(bw < 5.0 * sil)  # LOOCV may undersmooth; only the scale must be right
