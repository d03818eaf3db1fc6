"""The C ABI seen from plain C11: tests/abi_harness.c is compiled with gcc (not g++ / hipcc), links libkdehip.so and
runs SURVEY.md Appendix A through it.  Without a GPU the harness checks the layout assertions, the host tree
builder and the loud no-device failure; on the GPU box (-m gpu) it runs kdehip_gibbs1 / kdehip_gibbs1_trace."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kerneldensityestimate.jl_amd")


def _build_and_run(tmp_path):
    exe = str(tmp_path / "abi_harness")
    cmd = ["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_harness.c"), "-L" + PKG, "-lkdehip", "-lm", "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    return out.stdout.strip().splitlines()[-1]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_compiles_as_c11_and_fails_loudly_without_a_device(tmp_path):
    import kdehip
    last = _build_and_run(tmp_path)
    assert last == ("ok: device" if kdehip.device_count() > 0 else "ok: no-device")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_appendix_a_through_the_c_abi_from_c(tmp_path):
    assert _build_and_run(tmp_path) == "ok: device"
