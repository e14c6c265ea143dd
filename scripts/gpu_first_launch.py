"""First launch of a shape (what the drop-in CLI gets, main.cu:30) vs steady state; and ctr_render's total_ms with page-locked vs pageable destination buffers."""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5),
        ("dense64k", scenes.make_dense_bunny(d, 3), 5), ("c3deep", scenes.make_mirror_deep(d), 8)]
for name, path, b in todo:
    s = ca.HostScene.load(path)
    h = s.size[1]
    res, warm = {}, {}
    for label, var in (("image", 0),):
        firsts = []
        for rep in range(3):
            ds = ca.DeviceScene(s)
            ds.set_variant(var)
            ds.render(bounces=b, rows=(0, 8))          # another shape: loads the code object, warms the clocks
            firsts.append(ds.render(bounces=b)["kernel_ms"])
            # the same again on warm buffers: a slightly smaller image is a new shape (first launch of it)
            ds.set_size(s.size[0], h - 8 * (rep + 1))
            warm.setdefault(label, []).append(ds.render(bounces=b)["kernel_ms"])
            ds.close()
        res[label] = min(firsts)
    ds = ca.DeviceScene(s)
    for _ in range(4):
        ds.render(bounces=b)
    steady = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(7))
    # whole host-buffer calls: pageable buffers (reused from call to call / fresh ones), page-locked with the frame by one
    # DMA after the kernel (CTR_VAR_NO_DIRECT), page-locked with the frame delivered by the kernel
    bufs = ds.render(bounces=b)
    for _ in range(3):
        ds.render(bounces=b, into=bufs)
    tot_pageable = statistics.median(ds.render(bounces=b, into=bufs)["total_ms"] for _ in range(5))
    tot_fresh = statistics.median(ds.render(bounces=b)["total_ms"] for _ in range(3))
    ds.set_variant(ca.VAR_NO_DIRECT)
    for _ in range(3):
        ds.render(bounces=b, pinned=True)
    tot_dma = statistics.median(ds.render(bounces=b, pinned=True)["total_ms"] for _ in range(5))
    ds.set_variant(0)
    for _ in range(4):
        ds.render(bounces=b, pinned=True)
    tot_pinned = statistics.median(ds.render(bounces=b, pinned=True)["total_ms"] for _ in range(5))
    print(f"{name:13s} first launch (no costs yet: centre-out or image order) {res['image']:.3f} ms, new shape on warm buffers {min(warm['image']):.3f} ms; steady {steady:.3f} ms "
          f"(first/steady = {res['image'] / steady:.3f}); ctr_render total_ms: pageable {tot_pageable:.2f} (fresh buffers per call {tot_fresh:.2f}), "
          f"page-locked by DMA {tot_dma:.2f}, page-locked delivered by the kernel {tot_pinned:.2f}", flush=True)
