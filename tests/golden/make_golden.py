#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE build (oracle/_ref).

Run in the build container only (needs /root/reference to have built
oracle/_ref/libcutrace_ref.so):   python tests/golden/make_golden.py

What is stored (data only — inputs are the scene files under scene/, outputs are float buffers):
  small_<scene>_<w>x<h>_b<bounces>.npz   full depth/color/normal/hit_id buffers + ray count
  full_<scene>_<w>x<h>_b<bounces>.npz    for 1920x1080: double-precision sums of the buffers,
                                         finite-depth pixel count, ray count, and a fixed
                                         pseudo-random sample (seed 1234) of 4096 pixels
The reference build is g++ -O2 -ffp-contract=off of the reference's own headers (oracle/Makefile).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cutrace_amd as ca  # noqa: E402
import oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
THREADS = os.cpu_count() or 8

SMALL = [
    ("triangle", 20, 20, 5), ("triangle", 128, 128, 5),
    ("sphere_plane", 96, 54, 5), ("sphere_plane", 96, 54, 2),
    ("bunny", 96, 54, 5), ("bunny", 64, 36, 0),
    ("mirror", 96, 54, 5), ("mirror", 96, 54, 8),
]
FULL = [("sphere_plane", 1920, 1080, 5), ("mirror", 1920, 1080, 5), ("bunny", 1920, 1080, 5), ("mirror", 1920, 1080, 8)]
# C4 (4x4 bunny grid @4096x4096, cutrace_amd/scenes.py:make_bunny_grid): the reference needs minutes per frame on the
# CPU, so whole rows are stored instead: these rows of the 4096-row frame, every pixel of each
C4_ROWS = [40, 1111, 2050, 2600, 3000, 3333, 3700, 4000]


def sums(r):
    d = r["depth"]
    fin = np.isfinite(d)
    return dict(
        sum_color=r["color"].astype(np.float64).reshape(-1, 3).sum(0),
        sum_normal=r["normal"].astype(np.float64).reshape(-1, 3).sum(0),
        sum_depth=np.float64(d[fin].astype(np.float64).sum()),
        n_finite=np.int64(fin.sum()),
        ray_count=np.int64(r["ray_count"]),
    )


UV = [("triangle", 20, 20), ("sphere_plane", 96, 54), ("bunny", 96, 54)]


def main():
    # texture coordinates of the primary hit (ray_cast's tex_coords), straight from the reference build
    for name, w, h in UV:
        s = ca.HostScene.load(f"scene/{name}.json")
        s.set_size(w, h)
        r = oracle.ref_render(s, bounces=0, threads=THREADS, uv=True)
        path = os.path.join(OUT, f"uv_{name}_{w}x{h}.npz")
        np.savez_compressed(path, uv=r["uv"], hit_id=r["hit_id"].astype(np.int32), depth=r["depth"])
        print("wrote", path)
    # the kernel.hpp:52 cast with ray_cast's ignore_transparent = true (ray_cast.hpp:30,39-40; no caller of the reference
    # passes true, the harness can): sphere_plane.json has a transparent sphere (scene/sphere_plane.json:26)
    s = ca.HostScene.load("scene/sphere_plane.json")
    s.set_size(96, 54)
    r = oracle.ref_render(s, bounces=5, threads=THREADS, uv=True, ignore_transparent_primary=True)
    path = os.path.join(OUT, "ignore_transparent_sphere_plane_96x54_b5.npz")
    np.savez_compressed(path, depth=r["depth"], color=r["color"], normal=r["normal"], uv=r["uv"], hit_id=r["hit_id"].astype(np.int32),
                        ray_count=np.int64(r["ray_count"]))
    print("wrote", path)
    # ... and at the full 1920x1080 (samples + checksums, like the full_* fixtures)
    if "--no-full-ign" not in sys.argv:
        s = ca.HostScene.load("scene/sphere_plane.json")
        r = oracle.ref_render(s, bounces=5, threads=THREADS, uv=True, ignore_transparent_primary=True)
        w, h = s.size
        idx = np.sort(np.random.RandomState(4321).choice(w * h, 4096, replace=False)).astype(np.int64)
        path = os.path.join(OUT, f"full_ignore_transparent_sphere_plane_{w}x{h}_b5.npz")
        np.savez_compressed(path, sample_idx=idx, depth=r["depth"].reshape(-1)[idx], color=r["color"].reshape(-1, 3)[idx],
                            normal=r["normal"].reshape(-1, 3)[idx], uv=r["uv"].reshape(-1, 2)[idx], hit_id=r["hit_id"].reshape(-1)[idx].astype(np.int32),
                            **sums(r))
        print("wrote", path, sums(r))
    if "--uv-only" in sys.argv:
        return
    only_small = "--small" in sys.argv
    for name, w, h, b in SMALL:
        s = ca.HostScene.load(f"scene/{name}.json")
        assert s.ok
        s.set_size(w, h)
        r = oracle.ref_render(s, bounces=b, threads=THREADS)
        path = os.path.join(OUT, f"small_{name}_{w}x{h}_b{b}.npz")
        np.savez_compressed(path, depth=r["depth"], color=r["color"], normal=r["normal"], hit_id=r["hit_id"].astype(np.int32),
                            **sums(r))
        print("wrote", path, "casts", r["ray_count"])
    if only_small:
        return
    if "--no-c4" not in sys.argv:
        import tempfile
        from cutrace_amd import scenes
        d = tempfile.mkdtemp()
        s = ca.HostScene.load(scenes.make_bunny_grid(d))
        assert s.ok and s.size == (4096, 4096)
        out = {}
        for y in C4_ROWS:
            r = oracle.ref_render(s, bounces=5, rows=(y, y + 1), threads=THREADS)
            out[f"depth_{y}"], out[f"color_{y}"], out[f"normal_{y}"] = r["depth"][0], r["color"][0], r["normal"][0]
            out[f"rays_{y}"] = np.int64(r["ray_count"])
            print("C4 row", y, "casts", r["ray_count"], flush=True)
        path = os.path.join(OUT, "rows_bunny_grid4x4_4096x4096_b5.npz")
        np.savez_compressed(path, rows=np.asarray(C4_ROWS, np.int64), **out)
        print("wrote", path)
    if "--only-new" in sys.argv:
        todo = [t for t in FULL if not os.path.exists(os.path.join(OUT, f"full_{t[0]}_{t[1]}x{t[2]}_b{t[3]}.npz"))]
    else:
        todo = FULL
    rng = np.random.RandomState(1234)
    for name, w, h, b in FULL:
        if (name, w, h, b) not in todo:
            rng.choice(w * h, 4096, replace=False)  # keep the sample sequence of the fixtures already committed
            continue
        s = ca.HostScene.load(f"scene/{name}.json")
        assert s.ok and s.size == (w, h)
        r = oracle.ref_render(s, bounces=b, threads=THREADS)
        idx = np.sort(rng.choice(w * h, 4096, replace=False)).astype(np.int64)
        path = os.path.join(OUT, f"full_{name}_{w}x{h}_b{b}.npz")
        np.savez_compressed(path, sample_idx=idx, depth=r["depth"].reshape(-1)[idx], color=r["color"].reshape(-1, 3)[idx],
                            normal=r["normal"].reshape(-1, 3)[idx], hit_id=r["hit_id"].reshape(-1)[idx].astype(np.int32),
                            **sums(r))
        print("wrote", path, "casts", r["ray_count"], sums(r))


if __name__ == "__main__":
    main()
