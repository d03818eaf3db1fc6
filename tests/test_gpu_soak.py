"""Soak of the one-shot entry points on products with streamed and chunked tiles.  Every call packs, uploads and samples
FRESH plans (cold TLB / HBM: the kernel's direct-to-LDS copies are at their slowest), split over 2, 3 and 8 logical
devices that alias the one GPU (KDEHIP_ALIAS_DEVICES=1, hence the subprocess): results must be bit-identical to the
one-device call.  This is where a missing `vmcnt` wait before the staging barriers showed -- about five transient wrong
workgroups per 18,000 comparisons, with every other test green (csrc/gibbs_device.hpp `staging_barrier`).

A broken barrier / wait-count protocol has a second typical symptom besides wrong numbers: a hang (a wavefront that
never reaches the barrier the others wait at).  The child therefore runs under a watchdog (faulthandler dumps its
Python stack and exits) and reports when its first product has come back; a stall AFTER that point -- inside a
libkdehip call, with the device already acquired and working -- FAILS the test with the dumped stack.  Only a child
that never got as far as its first product (a fresh box can take minutes to page in the HIP runtime and to hand out
the device; that says nothing about the kernels) is started once more, and skipped if the second child does not get
there either."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIRST_CALL = "SOAK first GPU call done"


def _run_child():
    """(verdict, text): 'ok' | 'mismatch' | 'stalled-in-library' | 'never-started'"""
    env = dict(os.environ, KDEHIP_SOAK_WATCHDOG="150")  # (a stuck child dumps its Python stack and exits)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_multi.py"), "1500"], capture_output=True,
                           text=True, timeout=300, env=env)
        out, err, rc = r.stdout, r.stderr, r.returncode
    except subprocess.TimeoutExpired as e:  # (the watchdog should have fired first)
        dec = lambda b: b.decode("utf-8", "replace") if isinstance(b, bytes) else (b or "")
        out, err, rc = dec(e.stdout), dec(e.stderr) + "\n[no exit within 300 s]", None
    text = out[-3000:] + "\n" + err[-3000:]
    stalled = rc is None or "Timeout (0:02:30)!" in err
    if stalled:
        return ("stalled-in-library" if FIRST_CALL in err else "never-started"), text
    if rc != 0 or " 0 mismatches" not in out:
        return "mismatch", text
    return "ok", text


def test_one_shot_calls_on_fresh_plans_are_reproducible():
    verdict, text = _run_child()
    if verdict == "never-started":   # device acquisition / first import on a cold box: one fresh child
        verdict, text = _run_child()
        if verdict == "never-started":
            pytest.skip("two soak children never completed their first product (device acquisition): no verdict\n" + text)
    assert verdict != "stalled-in-library", "the soak child HUNG inside a libkdehip call (watchdog dump below)\n" + text
    assert verdict == "ok", text
