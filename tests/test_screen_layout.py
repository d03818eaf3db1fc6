"""The fp32 screen layout of fp64 plans (csrc/pack_levels.cpp phase 3b) checked on the host: which levels of the BASELINE
shapes are screened and how they are staged (resident / streamed / chunked), fits, alignments, no overlaps -- over 170
products.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kerneldensityestimate.jl_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_screen_layout(tmp_path):
    exe = str(tmp_path / "screen_layout_check")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           os.path.join(ROOT, "tests", "cpp", "screen_layout_check.cpp"), os.path.join(CSRC, "balltree.cpp"),
           os.path.join(CSRC, "pack_levels.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ)
    for k in ("KDEHIP_SCREEN", "KDEHIP_SCREEN_STREAM", "KDEHIP_SCREEN_CHUNK"):
        env.pop(k, None)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "screen layout ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
