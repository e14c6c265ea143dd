import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
s = ca.HostScene.load(scenes.make_dense_bunny(tempfile.mkdtemp(), int(sys.argv[1]) if len(sys.argv) > 1 else 3))
ds = ca.DeviceScene(s)
for _ in range(3):
    r = ds.render()
print("kernel_ms", r["kernel_ms"])
