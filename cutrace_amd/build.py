"""Build the native libraries in-tree (no network, no cmake needed).

  libcutrace_host.so   g++    host side: JSON loader, STL reader, image writers
  libcutrace_amd.so    hipcc  gfx950 render kernel + the C-ABI of include/cutrace_amd.h
  cutrace              hipcc  the drop-in CLI (`cutrace <scene.json>`, reference main.cu)

Everything float-sensitive is compiled with -ffp-contract=off so that each float
operation is rounded once, like the oracle.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cutrace_amd")
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INC = os.path.join(ROOT, "include")

HOST_LIB = os.path.join(PKG, "libcutrace_host.so")
HIP_LIB = os.path.join(PKG, "libcutrace_amd.so")
CLI = os.path.join(PKG, "cutrace")

HOST_SRCS = [os.path.join(HOST, "scene_host.cpp"), os.path.join(HOST, "images.cpp")]
HIP_SRCS = [os.path.join(CSRC, "render_kernel.hip"), os.path.join(CSRC, "ctr_api.cpp"), os.path.join(CSRC, "bvh.cpp"),
            os.path.join(CSRC, "ctr_multi.hip")]
CLI_SRCS = [os.path.join(HOST, "main.cpp")]

HOST_FLAGS = ["-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-pthread", "-Wall", "-I" + INC]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
    "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics",
    # no SLP vectorisation: left to itself the compiler pairs scalar f32 multiplies/adds into v_pk_* and pays for it with
    # v_mov / v_pk_mov shuffles to assemble the register pairs — a packed op issues at half rate (scripts/valu_issue.hip),
    # so that is a loss; the packed instructions that DO pay (SGPR-pair operands) are written out by hand.
    # Measured on one box: bunny 1.38 -> 1.24 ms, 64k-triangle bunny 3.18 -> 2.76 ms.
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-I" + INC, "-I" + CSRC,
]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    deps = list(sources)
    for d in (INC, CSRC, HOST):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                deps.append(os.path.join(d, f))
    deps.append(os.path.abspath(__file__))
    return all(os.path.getmtime(s) <= t for s in deps if os.path.exists(s))


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def build_host(force=False):
    if force or not _newer(HOST_LIB, HOST_SRCS):
        _run(["g++", *HOST_FLAGS, "-shared", "-o", HOST_LIB, *HOST_SRCS])
    return HOST_LIB


def build_hip(force=False):
    if force or not _newer(HIP_LIB, HIP_SRCS):
        _run([hipcc(), *HIP_FLAGS, "-shared", "-o", HIP_LIB, *HIP_SRCS, "-ldl"])
    return HIP_LIB


def build_cli(force=False):
    build_host(force)
    build_hip(force)
    if force or not _newer(CLI, CLI_SRCS + [HOST_LIB, HIP_LIB]):
        _run([hipcc(), "-std=c++17", "-O2", "-ffp-contract=off", "-pthread", "-I" + INC, "-I" + HOST, "-o", CLI, *CLI_SRCS,
              "-L" + PKG, "-lcutrace_amd", "-lcutrace_host", "-Wl,-rpath,$ORIGIN"])
    return CLI


def build_oracle(force=False):
    """Checker libraries (tests/bench only). _ref is rebuilt only where the reference exists."""
    args = ["make", "-C", os.path.join(ROOT, "oracle"), "all"]
    if force:
        args.insert(1, "-B")
    _run(args)


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_cli(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
