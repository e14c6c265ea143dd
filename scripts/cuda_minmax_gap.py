"""How far is "the reference compiled for the host" from a CUDA build's min/max?  (CPU only.)

The parity target is the reference's headers compiled for the host, where the unqualified min()/max() of the device code
(default_schema.hpp:109-110,241; shading.hpp:87,91) bind to std::min/std::max (a < b selects).  nvcc binds them to
float overloads that behave like fminf/fmaxf (a NaN operand is dropped).  The two differ only when a NaN reaches them:
an axis-parallel ray whose origin lies exactly on a mesh's box plane (0 x inf in mesh::bound_intersects).  This script
renders every config with both flavours of the reference build (oracle/_ref) and counts the pixels that differ.
It does not change what "the reference's output" means here; it tells a maintainer how far that definition is from
their CUDA binary.  (nvcc's default FMA contraction is a second, separate difference: SURVEY.md §7 'Hard parts'.)
usage: python scripts/cuda_minmax_gap.py > profiles/r02/cuda_minmax_gap.txt"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
import oracle
from cutrace_amd import scenes

NT = os.cpu_count() or 4
d = tempfile.mkdtemp()
axis = {  # the regression scene of tests/test_gpu_configs.py::test_axis_parallel_rays_through_meshes, larger
    "camera": {"eye": [0, 0, 3], "up": [0, 1, 0], "look": [0, 0, 0], "near_plane": 0.1, "far_plane": 10,
               "width": 640, "height": 360, "ambient": 0.05},
    "lights": [{"type": "point", "point": [0, 0, 2.5]}, {"type": "sun", "direction": [0, -1, 0]},
               {"type": "point", "point": [0.7431471, 0.5, 0.0]}],
    "materials": [{"type": "solid", "color": [0.8, 0.8, 0.8], "reflect": 0.3},
                  {"type": "solid", "color": [0.2, 0.5, 0.9], "reflect": 0.5}],
    "objects": [{"type": "mesh", "file": "scene/bunny.stl", "material": 0},
                {"type": "mesh", "file": "scene/frame.stl", "material": 1},
                {"type": "plane", "point": [0, -0.7391002, 0], "normal": [0, 1, 0], "material": 1},
                {"type": "plane", "point": [-1, 0, 0], "normal": [1, 0, 0], "material": 1}],
}
p_axis = os.path.join(d, "axis.json")
json.dump(axis, open(p_axis, "w"))
todo = [("C0 triangle.json@128x128", "scene/triangle.json", (128, 128), 5),
        ("C1 sphere_plane.json@1920x1080", "scene/sphere_plane.json", None, 5),
        ("C2 bunny.json@960x540", "scene/bunny.json", (960, 540), 5),
        ("C3 mirror.json@1920x1080 b8", "scene/mirror.json", None, 8),
        ("C3-deep @480x270 b8", scenes.make_mirror_deep(d), (480, 270), 8),
        ("C4 4x4 bunny grid @256x256", scenes.make_bunny_grid(d), (256, 256), 5),
        ("axis-aligned camera through bunny + frame @640x360 b4", p_axis, None, 4)]
print("pixels whose depth / normal / colour bits differ between the reference build with std::min/max (the parity target)")
print("and the same build with fminf/fmaxf semantics for the unqualified min/max (what nvcc's device overloads do):")
for name, path, size, b in todo:
    s = ca.HostScene.load(path)
    if size:
        s.set_size(*size)
    a = oracle.ref_render(s, bounces=b, threads=NT)
    c = oracle.ref_cudaminmax_render(s, bounces=b, threads=NT)
    w, h = s.size
    diff = (a["depth"].view(np.uint32) != c["depth"].view(np.uint32)) | \
           (a["normal"].view(np.uint32) != c["normal"].view(np.uint32)).any(-1) | \
           (a["color"].view(np.uint32) != c["color"].view(np.uint32)).any(-1)
    over = (np.abs(a["color"].astype(np.float64) - c["color"].astype(np.float64)) > 1e-4).any(-1)
    print(f"  {name:56s} {int(diff.sum()):7d} of {w * h:8d} pixels differ ({int(over.sum())} by more than 1e-4 in colour), "
          f"rays {a['ray_count']} vs {c['ray_count']}", flush=True)
