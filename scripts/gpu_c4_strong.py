"""C4 (4x4 bunny grid @4096x4096), strong scaling rehearsal on ONE GPU: the whole frame, then what each
rank of 2/4/8 would render of it (interleaved 8-row blocks, no collective).  Prints kernel ms per part and
the load-balance bound on the efficiency, t(1) / (n * max_r t_r)."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
path = scenes.make_bunny_grid(gen)

def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", path, "--width", "4096", "--height", "4096",
           "--scaling", "strong", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--skip-probe", "--no-extras"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(r.stderr[-800:]); raise SystemExit(1)
    return json.loads(line[-1])

for label, extra in (("one launch after the other", []), ("two frames in flight (--in-flight 2)", ["--in-flight", "2"])):
    print("#", label, flush=True)
    full = run(extra)
    t1 = full["ms_per_step"]
    print(f"1 GPU: {t1:.3f} ms/frame, {full['value']:.0f} Mrays/s", flush=True)
    for n in (2, 4, 8):
        ts = [run(extra + ["--of", str(n), "--as-rank", str(r)])["ms_per_step"] for r in range(n)]
        print(f"{n} ranks: per-rank ms {[round(t, 3) for t in ts]}  -> balance-limited efficiency {t1 / (n * max(ts)):.3f}", flush=True)
