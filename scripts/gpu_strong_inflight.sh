set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, json, subprocess, tempfile
ROOT=os.getcwd(); sys.path.insert(0, ROOT)
from cutrace_amd import scenes
path = scenes.make_bunny_grid(tempfile.mkdtemp())
def run(extra):
    cmd=[sys.executable,"bench.py","--scene",path,"--width","4096","--height","4096","--scaling","strong","--steps","24","--warmup","4","--no-cpu-baseline","--skip-probe","--no-extras"]+extra
    r=subprocess.run(cmd,capture_output=True,text=True,timeout=600)
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"]
for nf in (1,2,3,4):
    t1=run(["--in-flight",str(nf)])
    t8=[run(["--in-flight",str(nf),"--of","8","--as-rank",str(r)]) for r in (0,5)]
    print("in flight",nf,"whole %.3f"%t1,"1/8 parts",[round(x,3) for x in t8],"eff %.3f"%(t1/(8*max(t8))),flush=True)
PY
