"""AddressSanitizer + UBSan, and ThreadSanitizer, over the host code of libkdehip (tree builder, level packer, the worker
pool they share) -- CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kerneldensityestimate.jl_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("flags", [["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], ["-fsanitize=thread"]],
                         ids=["asan+ubsan", "tsan"])
def test_tree_builder_and_level_packer_are_sanitizer_clean(tmp_path, flags):
    """tsan: the top levels of a tree and the densities of a product are built / packed on the library's worker threads,
    from one caller and from four concurrent ones."""
    exe = str(tmp_path / "asan_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", *flags, "-pthread",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           os.path.join(ROOT, "tests", "asan_harness.cpp"), os.path.join(CSRC, "balltree.cpp"),
           os.path.join(CSRC, "pack_levels.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ok:" in out.stdout
