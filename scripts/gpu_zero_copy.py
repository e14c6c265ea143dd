"""ctr_render into a page-locked frame block: the kernel storing the frame into it directly (default) against device
buffers + one DMA (CTR_VAR_NO_DIRECT).  First frame of a fresh scene handle, steady state, and the bytes compared.
(A third option, row bands on streams of their own with their copies queued behind them, was measured in round 2 and
dropped: profiles/r02/host_call_direct_store.txt.)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca

s = ca.HostScene.load(sys.argv[1] if len(sys.argv) > 1 else "scene/bunny.json")
bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warm = ca.DeviceScene(s)
warm.render(bounces=bounces, rows=(0, 8), pinned=True)  # code object, clocks
res = {}
for name, var in (("device buffers + one DMA", ca.VAR_NO_DIRECT), ("kernel stores to host", 0)):
    firsts = []
    for rep in range(3):
        ds = ca.DeviceScene(s)
        ds.set_variant(var)
        ds._pinned_frame(ds.w * ds.h)
        ds.render(bounces=bounces, rows=(0, 8), pinned=True)    # another shape
        r = ds.render(bounces=bounces, pinned=True)
        firsts.append((r["total_ms"], r["kernel_ms"]))
        if rep < 2:
            ds.close()
    rr = [ds.render(bounces=bounces, pinned=True) for _ in range(12)][4:]
    res[name] = {k: rr[-1][k].copy() for k in ("depth", "color", "normal")}
    print(f"{name:26s}: first frame total {statistics.median(f[0] for f in firsts):.3f} ms (kernel {statistics.median(f[1] for f in firsts):.3f}); "
          f"steady total {statistics.median(x['total_ms'] for x in rr):.3f} ms (kernel {statistics.median(x['kernel_ms'] for x in rr):.3f})", flush=True)
    ds.close()
a, b = res.values()
print("bytes equal:", all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in a))
# ordinary (pageable) destinations: three hipMemcpy after the kernel, into reused and into fresh buffers
ds = ca.DeviceScene(s)
bufs = ds.render(bounces=bounces)
rr = [ds.render(bounces=bounces, into=bufs) for _ in range(12)][4:]
fr = [ds.render(bounces=bounces) for _ in range(6)][2:]
print(f"pageable, hipMemcpy       : buffers reused: steady total {statistics.median(x['total_ms'] for x in rr):.3f} ms; fresh numpy buffers per call: "
      f"{statistics.median(x['total_ms'] for x in fr):.3f} ms (kernel {statistics.median(x['kernel_ms'] for x in rr):.3f})", flush=True)
ds.close()
