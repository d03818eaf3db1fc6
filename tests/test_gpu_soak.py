"""Soak of the one-shot entry points on products with streamed and chunked tiles.  Every call packs, uploads and samples
FRESH plans (cold TLB / HBM: the kernel's direct-to-LDS copies are at their slowest), split over 2, 3 and 8 logical
devices that alias the one GPU (KDEHIP_ALIAS_DEVICES=1, hence the subprocess): results must be bit-identical to the
one-device call.  This is where a missing `vmcnt` wait before the staging barriers showed -- about five transient wrong
workgroups per 18,000 comparisons, with every other test green (csrc/gibbs_device.hpp `staging_barrier`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_shot_calls_on_fresh_plans_are_reproducible():
    env = dict(os.environ, KDEHIP_SOAK_WATCHDOG="150")  # (a stuck child dumps its Python stack and exits)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_multi.py"), "1500"], capture_output=True,
                           text=True, timeout=300, env=env)
    except subprocess.TimeoutExpired:
        pytest.skip("the soak subprocess did not finish in 300 s on this box (25 s normally): no verdict")
    if "Timeout (0:02:30)!" in r.stderr:
        pytest.skip("the soak subprocess stalled (watchdog): no verdict\n" + r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 mismatches" in r.stdout
