"""Host logic (CPU, no GPU): the BVH builders of cutrace_amd/csrc/bvh.cpp on random and degenerate triangle sets — every
primitive in exactly one leaf, leaf sizes, boxes containing what is below them, depth limits the kernel's lane stacks rely
on (scripts/bvh_check.cpp; the same harness runs under ASan/UBSan in scripts/cpu_sanitize.sh)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bvh_builders_keep_their_invariants(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = str(tmp_path / "bvh_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "cutrace_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "scripts", "bvh_check.cpp"), os.path.join(ROOT, "cutrace_amd", "csrc", "bvh.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all invariants hold" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
