"""Host logic (CPU, no GPU): the per-light occluder-distance maps of cutrace_amd/csrc/occl.cpp are CONSERVATIVE — for random lights and
triangle sets (tiny, huge, needle-shaped, degenerate, on cube-face edges and corners, ending within 1e-9 of a cell boundary) every point of
every triangle finds a bound no larger than its own distance in the cell the kernel's lookup rule picks, also for directions a few float ulps
off (scripts/occl_check.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_occluder_maps_are_conservative(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++ here")
    exe = str(tmp_path / "occl_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "cutrace_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "scripts", "occl_check.cpp"), os.path.join(ROOT, "cutrace_amd", "csrc", "occl.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "every map conservative" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
