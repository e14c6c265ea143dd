"""Build libcutrace_amd variants for same-box A/B runs (scripts/gpu_ab.py name=build_variants/<name>.so ...).
usage: build_variant.py name[:-DFLAG ...] ...      e.g.  build_variant.py cur nooct:-DCTR_NO_OCT
       build_variant.py --rev HEAD name            builds the committed sources of a revision (git worktree)"""
import os, subprocess, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutrace_amd import build
OUT = os.path.join(ROOT, "build_variants")
os.makedirs(OUT, exist_ok=True)
args = sys.argv[1:]
if args and args[0] == "--rev":
    rev, name = args[1], args[2]
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "--detach", tmp, rev], stdout=subprocess.DEVNULL)
    try:
        srcs = [os.path.join(tmp, os.path.relpath(p, ROOT)) for p in build.HIP_SRCS]
        flags = [f.replace(ROOT, tmp) if f.startswith("-I") else f for f in build.HIP_FLAGS]
        subprocess.check_call([build.hipcc(), *flags, "-shared", "-o", os.path.join(OUT, name + ".so"), *srcs, "-ldl"])
    finally:
        subprocess.call(["git", "-C", ROOT, "worktree", "remove", "--force", tmp])
    print("built", name, "from", rev)
    sys.exit(0)
procs = []
for spec in args:
    name, _, flags = spec.partition(":")
    lib = os.path.join(OUT, name + ".so")
    cmd = [build.hipcc(), *build.HIP_FLAGS, *flags.split(), "-shared", "-o", lib, *build.HIP_SRCS, "-ldl"]
    procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
for name, p in procs:
    out, _ = p.communicate()
    print(name, "OK" if p.returncode == 0 else "FAILED\n" + out[-2000:])
