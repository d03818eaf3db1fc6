"""The one place that knows the tile layouts (csrc/kdehip_internal.hpp TileAddr: fp64 rows, fp32 row pairs) checked on the
host: a bijection into the tile body, free pad elements, aligned and adjacent pairs, contiguous chunks."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_tile_addressing(tmp_path):
    exe = str(tmp_path / "tile_addr_check")
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "kerneldensityestimate.jl_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "tile_addr_check.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "tile addressing ok" in out.stdout, out.stdout[-2000:]
