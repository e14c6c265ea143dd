"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI (libcutrace_amd.so),
against the CPU oracle on the same inputs, against the committed golden fixtures (generated
from the reference build), and through size-independent properties at full resolution.

Bar: depth / normal / which-object are BIT-EXACT (only +,-,*,/,sqrt arithmetic, -ffp-contract=off
on both sides); colour within 1e-4 per channel (the specular pow() may differ by 1 ulp between
glibc powf and the device's f64 pow)."""
import glob
import os

import numpy as np
import pytest

import oracle

from tests.conftest import load_scene
from tests.test_oracle_golden import GOLD, parse
from tests.util import assert_parity, same_bits, mesh_scene as _mesh_scene, corner_meshes

pytestmark = pytest.mark.gpu

SMALL = sorted(glob.glob(os.path.join(GOLD, "small_*.npz")))


@pytest.fixture(scope="module")
def gpu(ca):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from cutrace_amd import _lib
    assert _lib.hip_lib().ctr_device_count() >= 1
    return ca


@pytest.mark.parametrize("path", SMALL, ids=[os.path.basename(p) for p in SMALL])
def test_gpu_matches_golden(gpu, path):
    name, w, h, b = parse(path)
    g = np.load(path)
    s = load_scene(gpu, name, w, h)
    ds = gpu.DeviceScene(s)
    r = ds.render(bounces=b)
    assert_parity(r, g, what=os.path.basename(path))
    assert r["ray_count"] == int(g["ray_count"])
    fin = np.isfinite(g["depth"])
    want_max = float(g["depth"][fin].max()) if fin.any() else 0.0
    assert r["max_depth"] == np.float32(want_max)


VARIANTS = [0, 32, 2, 4, 8, 4 | 8, 2 | 4 | 8 | 32]  # AUTO, EXACT_POW, NO_PREFILTER, NO_ANYHIT, NO_CLUSTER(BVH off), combos


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name,w,h,b", [("bunny", 160, 90, 5), ("mirror", 160, 90, 8), ("sphere_plane", 160, 90, 5)])
def test_gpu_matches_oracle_all_variants(gpu, name, w, h, b, variant):
    s = load_scene(gpu, name, w, h)
    o = oracle.oracle_render(s, bounces=b, threads=os.cpu_count() or 4)
    ds = gpu.DeviceScene(s)
    ds.set_variant(variant)
    r = ds.render(bounces=b)
    assert_parity(r, o, what=f"{name} variant {variant}")
    assert r["ray_count"] == o["ray_count"]


def test_variants_agree_bitwise_at_full_resolution(gpu):
    """prefilter / any-hit / BVH are pure accelerations: at 1920x1080 every buffer must be
    bit-identical with them on and off (size-independent property, no oracle needed)."""
    s = load_scene(gpu, "bunny")
    ds = gpu.DeviceScene(s)
    ds.set_variant(gpu.VAR_NO_PREFILTER | gpu.VAR_NO_ANYHIT | gpu.VAR_NO_CLUSTER | gpu.VAR_EXACT_POW)
    a = ds.render()
    ds.set_variant(gpu.VAR_EXACT_POW)
    b = ds.render()
    for k in ("depth", "normal", "color"):
        assert same_bits(a[k], b[k]), k
    assert a["ray_count"] == b["ray_count"] == 64278888  # SURVEY §8(c) oracle count


@pytest.mark.parametrize("name,casts", [("sphere_plane", 13973091), ("mirror", 8712144), ("bunny", 64278888)])
def test_full_resolution_against_golden_samples(gpu, name, casts):
    path = os.path.join(GOLD, f"full_{name}_1920x1080_b5.npz")
    if not os.path.exists(path):
        pytest.skip("full-resolution golden not generated")
    g = np.load(path)
    s = load_scene(gpu, name)
    r = gpu.DeviceScene(s).render(bounces=5)
    idx = g["sample_idx"]
    assert same_bits(r["depth"].reshape(-1)[idx], g["depth"])
    assert same_bits(r["normal"].reshape(-1, 3)[idx], g["normal"])
    d = np.abs(r["color"].reshape(-1, 3)[idx].astype(np.float64) - g["color"].astype(np.float64))
    assert (d > 1e-4).sum() == 0, (int((d > 1e-4).sum()), float(d.max()))
    assert r["ray_count"] == int(g["ray_count"]) == casts
    assert int(np.isfinite(r["depth"]).sum()) == int(g["n_finite"])
    fin = np.isfinite(r["depth"])
    assert abs(float(r["depth"][fin].astype(np.float64).sum()) - float(g["sum_depth"])) < 1e-6 * abs(float(g["sum_depth"]))
    assert np.allclose(r["color"].astype(np.float64).reshape(-1, 3).sum(0), g["sum_color"], rtol=1e-6)


def test_row_tiling_reassembles_the_frame(gpu):
    """ctr_rows interleaved blocks (multi-GPU tiling): parts reassemble to the full frame bit-exactly."""
    s = load_scene(gpu, "bunny", 200, 120)
    ds = gpu.DeviceScene(s)
    full = ds.render()
    n_parts, br = 4, 8
    total = 0
    for part in range(n_parts):
        rows = (0, 120, br, part, n_parts)
        r = ds.render(rows=rows)
        ys = [y for y in range(120) if (y // br) % n_parts == part]
        assert r["rows"] == len(ys)
        for k in ("depth", "normal", "color"):
            assert same_bits(r[k], full[k][ys]), (k, part)
        total += r["ray_count"]
    assert total == full["ray_count"]
    band = ds.render(rows=(40, 77))
    assert same_bits(band["color"], full["color"][40:77])


def test_edge_cases(gpu):
    # empty scene: every ray misses -> +inf depth, zero normal/colour, 2 casts/pixel
    import json
    base = {"camera": {"eye": [0, 0, -5], "up": [0, 1, 0], "look": [0, 0, 0], "near_plane": 0.1, "far_plane": 10,
                       "width": 37, "height": 19, "ambient": 0.2}, "lights": [], "materials": [], "objects": []}
    s = gpu.HostScene.parse(json.dumps(base))
    assert s.ok
    r = gpu.DeviceScene(s).render()
    assert np.isinf(r["depth"]).all() and not r["color"].any() and not r["normal"].any()
    assert r["ray_count"] == 2 * 37 * 19 and r["max_depth"] == 0.0
    # no lights: ambient only; ragged size (not a multiple of the 8x8 wave tile); bounces 0
    base["materials"] = [{"type": "solid", "color": [0.5, 0.25, 1.0], "reflect": 0.5, "transparency": 0.5}]
    base["objects"] = [{"type": "sphere", "center": [0, 0, 0], "radius": 1.5, "material": 0},
                       {"type": "triangle", "p1": [-3, -3, 1], "p2": [3, -3, 1], "p3": [0, 3, 1], "material": 0}]
    s = gpu.HostScene.parse(json.dumps(base))
    for b in (0, 1, 3, 15):
        o = oracle.oracle_render(s, bounces=b)
        r = gpu.DeviceScene(s).render(bounces=b)
        assert_parity(r, o, what=f"no-lights bounces={b}")
        assert r["ray_count"] == o["ray_count"]


def test_deep_recursion_scene(gpu):
    """C3-deep: mirror.json with the wall material made reflective so that
    every path really reaches depth 8 (divergent secondary rays)."""
    s = load_scene(gpu, "mirror", 128, 72)
    s.set_material(1, reflexivity=0.5)
    o = oracle.oracle_render(s, bounces=8, threads=os.cpu_count() or 4)
    r = gpu.DeviceScene(s).render(bounces=8)
    assert_parity(r, o, what="mirror deep b8")
    assert r["ray_count"] == o["ray_count"]


def test_device_buffer_form_with_torch(gpu):
    """ctr_render_device: torch owns the device memory and the stream (plumbing only)."""
    import torch
    s = load_scene(gpu, "sphere_plane", 96, 54)
    ds = gpu.DeviceScene(s)
    ref = ds.render()
    dev = torch.device("cuda:0")
    depth = torch.empty(54 * 96, dtype=torch.float32, device=dev)
    color = torch.empty(54 * 96 * 3, dtype=torch.float32, device=dev)
    normal = torch.empty(54 * 96 * 3, dtype=torch.float32, device=dev)
    counters = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ds.render_device(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), counters.data_ptr(),
                         stream.cuda_stream)
    stream.synchronize()
    assert same_bits(depth.cpu().numpy().reshape(54, 96), ref["depth"])
    assert same_bits(color.cpu().numpy().reshape(54, 96, 3), ref["color"])
    assert int(counters[0].item()) == ref["ray_count"]


def test_batch_launch_equals_single_frames(gpu):
    """ctr_render_device_batch: a camera path in one launch == the frames rendered one by one."""
    import ctypes as C
    import torch
    from cutrace_amd import _lib
    s = load_scene(gpu, "bunny", 96, 56)
    cam0 = s.desc.contents.cam
    cams = []
    for i in range(3):
        c = gpu.Camera()
        C.memmove(C.byref(c), C.byref(cam0), C.sizeof(gpu.Camera))
        _lib.host_lib().ctr_camera_look_at(C.byref(c), _lib.Vec3(1.0 + 0.2 * i, 0.1 * i, 2.0), _lib.Vec3(0, 1, 0),
                                           _lib.Vec3(-0.92388, 0, -0.38268))
        cams.append(c)
    singles = []
    for c in cams:
        ds1 = gpu.DeviceScene(s)
        ds1.set_cameras([c])
        singles.append(ds1.render())
    ds = gpu.DeviceScene(s)
    ds.set_cameras(cams)
    dev = torch.device("cuda:0")
    rows = (0, 56, 8, 1, 2)  # this "rank" owns every second 8-row block
    ys = [y for y in range(56) if (y // 8) % 2 == 1]
    cap = 32 + 3             # padded frame stride (largest part = 32 rows), like the multi-GPU tiler
    depth = torch.zeros(3 * cap * 96, dtype=torch.float32, device=dev)
    color = torch.zeros(3 * cap * 96 * 3, dtype=torch.float32, device=dev)
    normal = torch.zeros(3 * cap * 96 * 3, dtype=torch.float32, device=dev)
    counters = torch.zeros(16, dtype=torch.int64, device=dev)
    ds.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=3,
                           frame_stride_px=cap * 96, d_counters=counters.data_ptr(), rows=rows)
    torch.cuda.synchronize()
    d = depth.cpu().numpy().reshape(3, cap, 96)
    c = color.cpu().numpy().reshape(3, cap, 96, 3)
    for f in range(3):
        assert same_bits(d[f, :len(ys)], singles[f]["depth"][ys]), f
        assert same_bits(c[f, :len(ys)], singles[f]["color"][ys]), f
        assert not d[f, len(ys):].any()  # padding rows untouched
    # rotating parts: frame f renders part (1 + f) % 2
    depth.zero_(); color.zero_()
    ds.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=3,
                           frame_stride_px=cap * 96, rows=rows, part_stride=1)
    torch.cuda.synchronize()
    d = depth.cpu().numpy().reshape(3, cap, 96)
    for f in range(3):
        yf = [y for y in range(56) if (y // 8) % 2 == (1 + f) % 2]
        assert same_bits(d[f, :len(yf)], singles[f]["depth"][yf]), f
        assert not d[f, len(yf):].any()
    with pytest.raises(RuntimeError):
        ds.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=4, frame_stride_px=cap * 96)


def test_bad_arguments_fail_loudly(gpu):
    s = load_scene(gpu, "triangle")
    ds = gpu.DeviceScene(s)
    with pytest.raises(RuntimeError):
        ds.render(bounces=99)
    with pytest.raises(RuntimeError):
        ds.render(rows=(3, 20, 8, 0, 2))  # row_begin not block-aligned with n_parts > 1


def _random_scene(seed, w=72, h=48, opaque_mesh=False, extra_planes=False):
    """Seeded random scene mixing every primitive, light and material feature (incl. reflect +
    transparency on the same material, coincident planes, a mesh and stand-alone triangles)."""
    import json
    rng = np.random.RandomState(seed)

    def v(lo, hi):
        return [float(x) for x in rng.uniform(lo, hi, 3)]

    mats = []
    for _ in range(int(rng.randint(2, 6))):
        mats.append({"type": "solid", "color": v(0.05, 1.0), "specular": float(rng.uniform(0, 1)),
                     "reflect": float(rng.choice([0.0, 0.0, 0.3, 0.9])), "phong": float(rng.choice([0.0, 1.0, 20.0, 300.0])),
                     "transparency": 0.0 if opaque_mesh else float(rng.choice([0.0, 0.0, 0.0, 0.5]))})
    nm = len(mats)
    objs = []
    for _ in range(int(rng.randint(1, 5))):
        objs.append({"type": "sphere", "center": v(-1.5, 1.5), "radius": float(rng.uniform(0.2, 0.8)), "material": int(rng.randint(nm))})
    for _ in range(int(rng.randint(1, 4))):
        objs.append({"type": "triangle", "p1": v(-2, 2), "p2": v(-2, 2), "p3": v(-2, 2), "material": int(rng.randint(nm))})
    floor = {"type": "plane", "point": [0, -1.2, 0], "normal": [0, 1, 0], "material": int(rng.randint(nm))}
    objs.append(floor)
    if rng.rand() < 0.5:
        objs.append(dict(floor, material=int(rng.randint(nm))))  # coincident plane: exact tie on t
    objs.append({"type": "plane", "point": [0, 0, -3], "normal": [0, 0, 1], "material": int(rng.randint(nm))})
    if extra_planes:
        # more walls, axis-aligned (zeros of either sign, any length of normal) and not, anywhere in the object list
        rng2 = np.random.RandomState(seed + 7919)
        for _ in range(int(rng2.randint(1, 7))):
            if rng2.rand() < 0.7:
                a = int(rng2.randint(3))
                n = [float(rng2.choice([0.0, -0.0])) for _ in range(3)]
                n[a] = float(rng2.choice([-1.0, 1.0]) * rng2.choice([1.0, 0.25, 3.0, 1e-3]))
                pt = [float(x) for x in rng2.uniform(-1, 1, 3)]
                pt[a] = float(-np.sign(n[a]) * rng2.uniform(2.0, 6.0))
            else:
                n = [float(x) for x in rng2.uniform(-1, 1, 3)]
                pt = [float(-4.0 * x) for x in n]
            objs.insert(int(rng2.randint(len(objs) + 1)), {"type": "plane", "point": pt, "normal": n, "material": int(rng2.randint(nm))})
    if opaque_mesh or rng.rand() < 0.7:
        objs.insert(int(rng.randint(len(objs) + 1)), {"type": "mesh", "file": "scene/skull.stl", "material": int(rng.randint(nm))})
    lights = [{"type": "sun", "direction": v(-1, 1), "color": v(0.2, 1)}]
    for _ in range(int(rng.randint(0, 3))):
        lights.append({"type": "point", "point": v(-3, 3), "color": v(0.2, 1)})
    cam = {"eye": [float(rng.uniform(-1, 3)), float(rng.uniform(-0.5, 2)), 4.0], "up": [0, 1, 0], "look": v(-0.5, 0.5),
           "near_plane": 0.1, "far_plane": 100.0, "width": w, "height": h, "ambient": float(rng.uniform(0, 0.3))}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


@pytest.mark.parametrize("seed", list(range(8)))
def test_seeded_random_scenes(gpu, seed):
    s = gpu.HostScene.parse(_random_scene(seed))
    assert s.ok
    b = [0, 2, 3, 5][seed % 4]
    o = oracle.oracle_render(s, bounces=b, threads=os.cpu_count() or 4)
    ds = gpu.DeviceScene(s)
    for variant in (gpu.VAR_AUTO, gpu.VAR_EXACT_POW | gpu.VAR_NO_CLUSTER | gpu.VAR_NO_PREFILTER | gpu.VAR_NO_ANYHIT):
        ds.set_variant(variant)
        r = ds.render(bounces=b)
        assert_parity(r, o, what=f"random scene seed {seed} variant {variant}")
        assert r["ray_count"] == o["ray_count"]


@pytest.mark.parametrize("seed", list(range(100, 108)))
def test_seeded_random_opaque_mesh_scenes(gpu, seed):
    """All-opaque scenes with a mesh: the any-hit shadow path incl. the prefilter's decisive accept and
    ray-parameter rejects (render_kernel.hip, prefilter stage 2) against the oracle and, bitwise, against
    the kernel with every shortcut switched off."""
    s = gpu.HostScene.parse(_random_scene(seed, w=96, h=64, opaque_mesh=True))
    assert s.ok
    b = [1, 2, 3, 5][seed % 4]
    o = oracle.oracle_render(s, bounces=b, threads=os.cpu_count() or 4)
    ds = gpu.DeviceScene(s)
    r = ds.render(bounces=b)
    assert_parity(r, o, what=f"opaque mesh scene seed {seed}")
    assert r["ray_count"] == o["ray_count"]
    ds.set_variant(gpu.VAR_NO_CLUSTER | gpu.VAR_NO_PREFILTER | gpu.VAR_NO_ANYHIT)
    plain = ds.render(bounces=b)
    for k in ("depth", "normal", "color"):
        assert same_bits(r[k], plain[k]), (k, seed)


def test_tile_scheduling_feedback_never_changes_results(gpu):
    """Cost-ordered dispatch (include/cutrace_amd.h "Tile scheduling"): the order comes from the previous
    launch of the same shape; any order must give the same bits, every pixel written exactly once, also
    when the shape changes between launches (stale order dropped) and for batches with rotating parts."""
    import torch
    s = load_scene(gpu, "bunny", 200, 120)
    ref_ds = gpu.DeviceScene(s)
    ref_ds.set_variant(gpu.VAR_NO_REORDER)
    ref = ref_ds.render()
    ds = gpu.DeviceScene(s)
    for i in range(4):                       # launch 0 image order, 1.. cost order
        r = ds.render()
        for k in ("depth", "normal", "color"):
            assert same_bits(r[k], ref[k]), (k, i)
        assert r["ray_count"] == ref["ray_count"]
    part = ds.render(rows=(0, 120, 8, 1, 3))  # other shape: fewer waves than the stored order
    ys = [y for y in range(120) if (y // 8) % 3 == 1]
    assert same_bits(part["color"], ref["color"][ys])
    part = ds.render(rows=(0, 120, 8, 1, 3))  # same shape again: its own order now
    assert same_bits(part["color"], ref["color"][ys])
    r = ds.render()                           # back to the full frame (larger than the stored order)
    assert same_bits(r["color"], ref["color"]) and same_bits(r["depth"], ref["depth"])
    s2 = load_scene(gpu, "bunny", 64, 40)     # set_size on the same handle
    small_ref = gpu.DeviceScene(s2)
    small_ref.set_variant(gpu.VAR_NO_REORDER)
    small = small_ref.render()
    ds.set_size(64, 40)
    for i in range(2):
        r = ds.render()
        assert same_bits(r["color"], small["color"]) and same_bits(r["normal"], small["normal"])
    # device-buffer batch with rotating parts, launched three times on the same buffers
    ds = gpu.DeviceScene(s)
    dev = torch.device("cuda:0")
    cams = [s.desc.contents.cam] * 2
    ds.set_cameras(cams)
    cap = 64
    rows = (0, 120, 8, 0, 2)
    outs = []
    for i in range(3):
        depth = torch.full((2 * cap * 200,), -1.0, dtype=torch.float32, device=dev)
        color = torch.full((2 * cap * 200 * 3,), -1.0, dtype=torch.float32, device=dev)
        normal = torch.zeros(2 * cap * 200 * 3, dtype=torch.float32, device=dev)
        ds.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=2,
                               frame_stride_px=cap * 200, rows=rows, part_stride=1)
        torch.cuda.synchronize()
        outs.append((depth.cpu().numpy().reshape(2, cap, 200), color.cpu().numpy().reshape(2, cap, 200, 3)))
    for f in range(2):
        yf = [y for y in range(120) if (y // 8) % 2 == f % 2]
        for d, c in outs:
            assert same_bits(d[f, :len(yf)], ref["depth"][yf]), f
            assert same_bits(c[f, :len(yf)], ref["color"][yf]), f
            assert (d[f, len(yf):] == -1.0).all()   # padding rows untouched


def _check_all_ways(gpu, s, what, bounces=3, fudge=1e-3):
    o = oracle.oracle_render(s, bounces=bounces, fudge=fudge, threads=os.cpu_count() or 4)
    ds = gpu.DeviceScene(s)
    r = ds.render(bounces=bounces, fudge=fudge)
    assert_parity(r, o, what=what)
    assert r["ray_count"] == o["ray_count"]
    ds.set_variant(gpu.VAR_NO_CLUSTER | gpu.VAR_NO_PREFILTER | gpu.VAR_NO_ANYHIT)
    plain = ds.render(bounces=bounces, fudge=fudge)
    for k in ("depth", "normal", "color"):
        assert same_bits(r[k], plain[k]), (what, k)
    return r


def test_mesh_corner_cases(gpu, tmp_path):
    """Meshes the shortcuts could trip over: coincident duplicate triangles (exact ties on t -> file
    order), zero-area triangles (alpha == 0: the reference divides by zero), a single-triangle mesh (BVH
    root is a leaf), an empty mesh, triangles meeting edge-on under the camera, far-away coordinates;
    image sizes 1x1, 8x8 and 9x9; fudge 0 and negative (prefilter stage 2 must switch itself off)."""
    quad, dup, degenerate, fan, far = corner_meshes()
    cases = [("duplicates", dup, 64, 40), ("degenerate", degenerate, 64, 40), ("single", quad[:1], 40, 24),
             ("fan", fan, 72, 48), ("1x1", fan, 1, 1), ("8x8", fan, 8, 8), ("9x9", dup, 9, 9)]
    for name, tris, w, h in cases:
        s = gpu.HostScene.parse(_mesh_scene(str(tmp_path / f"{name}.stl"), w, h, tris))
        assert s.ok, name
        _check_all_ways(gpu, s, name)
    # far-away mesh next to a near one: margins scale with the coordinates
    s = gpu.HostScene.parse(_mesh_scene(str(tmp_path / "far.stl"), 64, 40, far,
                                        extra_objects=[{"type": "mesh", "file": str(tmp_path / "fan.stl"), "material": 1}]))
    assert s.ok
    _check_all_ways(gpu, s, "far + near mesh")
    # fudge 0 / negative: self-intersections at t ~ 0 become valid hits in the reference
    s = gpu.HostScene.parse(_mesh_scene(str(tmp_path / "fan.stl"), 48, 32, fan))
    for fudge in (0.0, -0.5, 1e-6):
        _check_all_ways(gpu, s, f"fudge {fudge}", bounces=2, fudge=fudge)


@pytest.mark.parametrize("w,h", [(1001, 777), (2048, 1536), (8, 8), (1, 4000)])
def test_tile_order_is_a_permutation_at_odd_and_large_sizes(gpu, w, h):
    """The counting sort behind the tile order (after_render) at wave counts that are not multiples of its
    block size, at one wave, and at ~49k waves: every launch must still write every pixel exactly once."""
    s = load_scene(gpu, "sphere_plane", w, h)
    ref_ds = gpu.DeviceScene(s)
    ref_ds.set_variant(gpu.VAR_NO_REORDER)
    ref = ref_ds.render(bounces=3)
    ds = gpu.DeviceScene(s)
    for i in range(3):
        r = ds.render(bounces=3)
        for k in ("depth", "normal", "color"):
            assert same_bits(r[k], ref[k]), (k, i, w, h)
        assert r["ray_count"] == ref["ray_count"]


def _coplanar_scene(ca, tmp_path, w, h, row, n_tris, seed):
    """A mesh whose triangles all lie (to float rounding) in the plane that contains EVERY primary ray of image row
    `row`: for those rays alpha = det[a b c] of default_schema.hpp:59 is pure rounding noise, and so are beta, gamma
    and t — the regime in which the reference's float test can report a hit for a ray that passes far from the
    triangle (BVH caveat)."""
    import ctypes as C
    import json
    from cutrace_amd import _lib, scenes
    rng = np.random.default_rng(seed)
    eye, up, look = (0.3, 0.8, 4.0), (0.0, 1.0, 0.0), (-0.05, -0.15, -1.0)
    cam = _lib.Camera()
    _lib.host_lib().ctr_camera_look_at(C.byref(cam), _lib.Vec3(*eye), _lib.Vec3(*up), _lib.Vec3(*look))
    f32 = np.float32
    E, R, U, F = (np.array(v.tup(), f32) for v in (cam.pos, cam.right, cam.up, cam.forward))
    v = (f32(0.5) - f32(row) / f32(h)) * U + F           # the row's rays: E + s*right*k + t*v
    tris = []
    for _ in range(n_tris):
        s0, t0 = f32(rng.uniform(-1.2, 1.2)), f32(rng.uniform(2.0, 5.0))
        pts = []
        for _ in range(3):
            s, t = s0 + f32(rng.uniform(-0.25, 0.25)), t0 + f32(rng.uniform(-0.4, 0.4))
            pts.append((E + s * R + t * v).astype(f32))
        tris.append(pts)
    stl = str(tmp_path / f"coplanar_{seed}.stl")
    scenes.write_stl(stl, np.asarray(tris, f32))
    sc = {"camera": {"eye": list(eye), "up": list(up), "look": list(look), "near_plane": 0.1, "far_plane": 100.0,
                     "width": w, "height": h, "ambient": 0.1},
          "lights": [{"type": "point", "point": [1.5, 2.5, 2.0], "color": [0.8, 0.8, 0.8]},
                     {"type": "point", "point": [float(E[0] + 0.5 * R[0] + 1.0 * v[0]), float(E[1] + 0.5 * R[1] + 1.0 * v[1]),
                                                  float(E[2] + 0.5 * R[2] + 1.0 * v[2])]}],   # a light IN the plane too
          "materials": [{"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.3, "phong": 40},
                        {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.2, "phong": 10}],
          "objects": [{"type": "mesh", "file": stl, "material": 0},
                      {"type": "plane", "point": [0, -1.0, 0], "normal": [0, 1, 0], "material": 1},
                      {"type": "plane", "point": [0, 0, -6.0], "normal": [0, 0, 1], "material": 1}]}
    s = ca.HostScene.parse(json.dumps(sc))
    assert s.ok
    return s


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rays_coplanar_with_triangles(ca, tmp_path, seed):
    """Adversarial case for the BVH culling: 48 triangles in the plane of image row 12, so
    that 512 primary rays (and the shadow rays towards the in-plane light) are coplanar with every one of them up to
    rounding.  The accelerated kernel must still equal the plain linear walk bit for bit and the oracle within the
    parity bar; the number of pixels of that row whose hit is one of these noise-level triangles is reported."""
    w, h, row = 512, 24, 12
    s = _coplanar_scene(ca, tmp_path, w, h, row, 48, seed)
    o = oracle.oracle_render(s, bounces=2, threads=os.cpu_count() or 4)
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_EXACT_POW)
    fast = ds.render(bounces=2)
    ds.set_variant(ca.VAR_EXACT_POW | ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT)
    plain = ds.render(bounces=2)
    assert_parity(plain, o, what=f"coplanar seed {seed}: plain walk vs oracle")
    on_mesh = int((o["hit_id"][row] == 0).sum())
    diff = int((fast["depth"].view(np.uint32) != plain["depth"].view(np.uint32)).sum())
    print(f"coplanar seed {seed}: {on_mesh} of {w} pixels of row {row} hit the in-plane mesh; accelerated vs plain: {diff} depth values differ")
    for k in ("depth", "normal", "color"):
        assert same_bits(fast[k], plain[k]), f"seed {seed}: {k}"
    assert fast["ray_count"] == plain["ray_count"] == o["ray_count"]


def _mirror_coplanar_scene(ca, tmp_path, w, h, row, n_tris, seed, transparent, two_mirrors=False):
    """The same regime for SECONDARY rays, which no upload-time guard can see (ctr_api.cpp refresh_linear_meshes checks eyes
    and lights): a tilted mirror reflects every primary ray of image row `row` into ONE plane — the mirror image of the
    row's plane, through the mirror image of the eye — and the mesh's triangles lie in that plane to float rounding, some
    on the reflected rays' way, some far to the side of it.  For those reflected rays (and, with `transparent`, for the
    pass-through rays that continue from an in-plane hit in the same plane) alpha and all three numerators of
    default_schema.hpp:57-78 are rounding noise.  Neither the eye nor any light lies in that plane."""
    import ctypes as C
    import json
    from cutrace_amd import _lib, scenes
    rng = np.random.default_rng(seed)
    eye, up, look = (0.2, 0.9, 3.5), (0.0, 1.0, 0.0), (0.02, -0.1, -1.0)
    cam = _lib.Camera()
    _lib.host_lib().ctr_camera_look_at(C.byref(cam), _lib.Vec3(*eye), _lib.Vec3(*up), _lib.Vec3(*look))
    f32, f64 = np.float32, np.float64
    E, R, U, F = (np.array(v.tup(), f64) for v in (cam.pos, cam.right, cam.up, cam.forward))
    v = (0.5 - row / h) * U + F                           # the row's rays: E + s*R + t*v
    pm = np.array([0.0, 0.0, -2.0])                       # the mirror: a plane through pm, tilted towards the ceiling
    nm = np.array([0.0, 0.35, 1.0])
    mirrors = [(pm, nm)]
    if two_mirrors:                                       # ... and a second one above that sends the rays down again
        mirrors.append((np.array([0.0, 3.0, 0.0]), np.array([0.0, -1.0, 0.25])))
    E2, R2, v2, t_mirror = E, R, v, 0.0
    for (p_, n_) in mirrors:                              # images of the eye and of the row's plane, mirror after mirror
        nh = n_ / np.linalg.norm(n_)
        t_mirror = np.dot(p_ - E2, nh) / np.dot(v2, nh)   # where the row's central ray meets this mirror
        assert t_mirror > 0
        E2 = E2 - 2.0 * np.dot(E2 - p_, nh) * nh
        R2, v2 = R2 - 2.0 * np.dot(R2, nh) * nh, v2 - 2.0 * np.dot(v2, nh) * nh
    tris = []
    for _ in range(n_tris):
        s0, t0 = rng.uniform(-1.5, 1.5), t_mirror + rng.uniform(0.4, 3.5)   # beyond the mirror point = on the reflected side
        pts = []
        for _ in range(3):
            s_, t_ = s0 + rng.uniform(-0.3, 0.3), t0 + rng.uniform(-0.35, 0.35)
            pts.append((E2 + s_ * R2 + t_ * v2).astype(f32))
        tris.append(pts)
    stl = str(tmp_path / f"mirror_coplanar_{seed}_{int(transparent)}.stl")
    scenes.write_stl(stl, np.asarray(tris, f32))
    mesh_mat = {"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.3, "phong": 40}
    if transparent:
        mesh_mat["transparency"] = 0.4
    sc = {"camera": {"eye": list(eye), "up": list(up), "look": list(look), "near_plane": 0.1, "far_plane": 100.0,
                     "width": w, "height": h, "ambient": 0.1},
          "lights": [{"type": "point", "point": [1.5, 2.5, 2.0], "color": [0.8, 0.8, 0.8]},
                     {"type": "sun", "direction": [0.3, -1.0, -0.2], "color": [0.4, 0.4, 0.4]}],
          "materials": [mesh_mat,
                        {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.0, "phong": 10},
                        {"type": "solid", "color": [0.9, 0.9, 0.9], "specular": 0.1, "reflect": 0.9, "phong": 20}],
          "objects": [{"type": "mesh", "file": stl, "material": 0},
                      {"type": "plane", "point": [0, -1.5, 0], "normal": [0, 1, 0], "material": 1}] +
                     [{"type": "plane", "point": [float(x) for x in p_], "normal": [float(x) for x in n_], "material": 2}
                      for (p_, n_) in mirrors]}
    s = ca.HostScene.parse(json.dumps(sc))
    assert s.ok
    return s


@pytest.mark.parametrize("seed,transparent,two_mirrors", [(1, False, False), (2, False, False), (3, False, False), (4, True, False),
                                                          (5, True, False), (6, False, True), (7, False, True), (8, True, True)])
def test_secondary_rays_coplanar_with_triangles(ca, tmp_path, seed, transparent, two_mirrors):
    """reflected (and passed-through) rays that fall into the plane of mesh triangles by construction —
    after one flat mirror, or (two_mirrors) after a mirror and a second mirror.  Without the virtual-eye guard
    (ctr_api.cpp refresh_linear_meshes) seed 2 differed from the linear walk in one pixel.  The accelerated kernel (BVH culling, prefilter, any-hit where all materials are opaque) must equal the plain linear
    walk — the reference's own traversal — bit for bit, and the oracle within the parity bar."""
    w, h, row = 384, 32, 9
    s = _mirror_coplanar_scene(ca, tmp_path, w, h, row, 40, seed, transparent, two_mirrors)
    o = oracle.oracle_render(s, bounces=3, threads=os.cpu_count() or 4)
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_EXACT_POW)
    fast = ds.render(bounces=3)
    ds.set_variant(ca.VAR_EXACT_POW | ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT)
    plain = ds.render(bounces=3)
    assert_parity(plain, o, what=f"mirror-coplanar seed {seed}: plain walk vs oracle")
    on_mirror = int((o["hit_id"][row] == 2).sum())
    diff = int((fast["color"].view(np.uint32) != plain["color"].view(np.uint32)).any(axis=-1).sum())
    print(f"mirror-coplanar seed {seed} transparent={transparent}: {on_mirror} of {w} pixels of row {row} see the mirror; "
          f"accelerated vs plain: {diff} pixels differ in colour")
    assert on_mirror > w // 2
    for k in ("depth", "normal", "color"):
        assert same_bits(fast[k], plain[k]), f"seed {seed}: {k}"
    assert fast["ray_count"] == plain["ray_count"] == o["ray_count"]


def test_exact_sqrt_and_reciprocal_shortcut_is_exhaustively_correct(ca):
    """The kernel normalises vectors with a shorter instruction sequence than hipcc's correctly rounded sqrtf and
    division (render_kernel.hip norm_and_inverse); it must give the same bits for EVERY mantissa."""
    import ctypes as C
    from cutrace_amd import _lib
    n = C.c_uint64(1)
    assert _lib.hip_lib().ctr_selftest_exact_math(C.byref(n)) == 0
    assert n.value == 0, f"{n.value} of {6 * 2**23} (sqrt, 1/sqrt) pairs differ from sqrtf / IEEE division"


def test_six_wave_build_and_pinned_frames_change_nothing(ca):
    """The build compiled for 6 waves per SIMD (picked for >= 1000 mesh triangles; five cold dwords parked in LDS) against
    the 5-wave build, and delivery into a page-locked frame block (one DMA) against pageable buffers: same bytes."""
    s = load_scene(ca, "bunny")            # 1000 triangles: the 6-wave build by default
    ds = ca.DeviceScene(s)
    a = ds.render(bounces=5)
    ds.set_variant(ca.VAR_NO_OCC6)
    b = ds.render(bounces=5)
    ds.set_variant(0)
    c = ds.render(bounces=5, pinned=True)              # the kernel stores into the page-locked block itself
    for k in ("depth", "normal", "color"):
        assert same_bits(a[k], b[k]), k
        assert same_bits(a[k], c[k]), k
    assert a["ray_count"] == b["ray_count"] == c["ray_count"] == 64278888
    assert a["max_depth"] == b["max_depth"] == c["max_depth"]
    c["depth"][:] = 0.0
    ds.set_variant(ca.VAR_NO_DIRECT)                   # device buffers + one DMA into the same block
    c2 = ds.render(bounces=5, pinned=True)
    ds.set_variant(0)
    for k in ("depth", "normal", "color"):
        assert same_bits(a[k], c2[k]), k
    c2["depth"][:] = 0.0
    c3 = ds.render(bounces=5, pinned=True)             # and directly again, every pixel rewritten
    assert same_bits(a["depth"], c3["depth"]) and same_bits(a["color"], c3["color"])
    # a row subset into the same (larger) pinned block, then the diagnostics
    d = ds.render(bounces=5, rows=(0, 1080, 8, 3, 8), pinned=True)
    ys = [y for y in range(1080) if (y // 8) % 8 == 3]
    assert same_bits(d["depth"], a["depth"][ys]) and same_bits(d["color"], a["color"][ys])
    costs = ds.tile_costs()
    assert costs.size == 240 * ((len(ys) + 7) // 8) and costs.min() > 0
    ds.set_variant(ca.VAR_STATS)
    ds.render(bounces=5)
    cnt = ds.last_counters()
    assert int(cnt[0]) == 64278888 and int(cnt[4]) == 972000     # 30 trips for each of the 32 400 waves
    assert 0 < int(cnt[10]) <= 64 * int(cnt[5]) and 0 < int(cnt[11]) <= 64 * int(cnt[6]) and 0 < int(cnt[12]) <= 64 * int(cnt[7])


def test_host_delivery_with_changing_frames(ca):
    """Delivery into page-locked memory by the kernel stages every tile in device memory and copies it to the host from
    another wave, usually on another XCD (render_kernel.hip "Host delivery").  Frames of DIFFERENT content through the
    same scene handle — a moving camera, changing bounce counts, full size — so that a pixel left over from the previous
    frame (a stale cache line, a group copied before its last tile arrived) cannot go unnoticed: every frame bitwise
    equal to the same frame through device buffers + DMA on a second handle."""
    import ctypes as C
    from cutrace_amd import _lib
    s = load_scene(ca, "bunny")
    cam0 = s.desc.contents.cam
    direct, plain = ca.DeviceScene(s), ca.DeviceScene(s)
    plain.set_variant(ca.VAR_NO_DIRECT)
    for i in range(6):
        c = ca.Camera()
        C.memmove(C.byref(c), C.byref(cam0), C.sizeof(ca.Camera))
        _lib.host_lib().ctr_camera_look_at(C.byref(c), _lib.Vec3(1.0 - 0.1 * i, 0.05 * i, 2.0 - 0.03 * i), _lib.Vec3(0, 1, 0),
                                           _lib.Vec3(-0.92388 + 0.02 * i, 0.0, -0.38268))
        direct.set_cameras([c])
        plain.set_cameras([c])
        b = (5, 1, 3, 0, 5, 2)[i]
        got = direct.render(bounces=b, pinned=True)
        want = plain.render(bounces=b, pinned=True)
        for k in ("depth", "normal", "color"):
            assert same_bits(got[k], want[k]), (i, k)
        assert got["ray_count"] == want["ray_count"] and got["max_depth"] == want["max_depth"]


@pytest.mark.parametrize("w,h", [(1, 1), (9, 9), (65, 7), (64, 8), (63, 129), (200, 3), (513, 70)])
def test_host_delivery_odd_sizes_and_separate_buffers(ca, w, h):
    """Delivery by the kernel at sizes with ragged tiles, ragged tile groups and fewer rows than a tile, into ONE
    page-locked block and into three separately page-locked buffers (torch pinned tensors, in a different order in memory),
    with whole-frame and interleaved-part row selections: bitwise the frame through pageable buffers."""
    import ctypes as C
    import torch
    from cutrace_amd import _lib
    s = load_scene(ca, "bunny", w, h)
    ds = ca.DeviceScene(s)
    want = ds.render(bounces=3)
    got = ds.render(bounces=3, pinned=True)
    for k in ("depth", "normal", "color"):
        assert same_bits(got[k], want[k]), k
    assert got["ray_count"] == want["ray_count"] and got["max_depth"] == want["max_depth"]
    L = _lib.hip_lib()
    for rows in (None, (0, h, 8, 1, 2)):
        ref = ds.render(bounces=3, rows=rows)
        n = ref["depth"].shape[0]
        if n == 0:
            continue
        normal = torch.zeros(n * w * 3, dtype=torch.float32).pin_memory()   # allocated first: not in [depth|color|normal] order
        depth = torch.zeros(n * w, dtype=torch.float32).pin_memory()
        color = torch.zeros(n * w * 3, dtype=torch.float32).pin_memory()
        r = ca.make_rows(h, rows)
        stats = ca.RenderStats()
        assert L.ctr_render(ds._h, C.c_float(1e-3), 3, C.byref(r), depth.data_ptr(), color.data_ptr(), normal.data_ptr(),
                            C.byref(stats)) == 0
        assert same_bits(depth.numpy().reshape(n, w), ref["depth"])
        assert same_bits(color.numpy().reshape(n, w, 3), ref["color"])
        assert same_bits(normal.numpy().reshape(n, w, 3), ref["normal"])
        assert int(stats.ray_count) == ref["ray_count"]


@pytest.mark.parametrize("w,h,rows", [(1000, 700, None), (1920, 1080, (0, 1080, 8, 1, 2)), (1027, 1033, (16, 1001))])
def test_first_launch_centre_out_order_is_a_permutation(ca, w, h, rows):
    """The first launch of a shape dispatches its tiles centre-out (render_kernel.hip first_order; scenes with >= 1000 mesh
    triangles, >= 8192 tiles).  A tile left out or visited twice would show as a wrong pixel: the first frame of a fresh
    handle — ragged blocks of tiles, interleaved row parts, a row sub-range — against image order, bitwise, plus the ray
    count; then the second frame (measured order)."""
    s = load_scene(ca, "bunny", w, h)
    ref = ca.DeviceScene(s)
    ref.set_variant(ca.VAR_NO_REORDER)
    want = ref.render(bounces=2, rows=rows)
    ds = ca.DeviceScene(s)
    for call in range(2):
        got = ds.render(bounces=2, rows=rows)
        for k in ("depth", "normal", "color"):
            assert same_bits(got[k], want[k]), (call, k)
        assert got["ray_count"] == want["ray_count"] and got["max_depth"] == want["max_depth"]
    costs = ds.tile_costs()
    assert costs.size == ((w + 7) // 8) * ((want["depth"].shape[0] + 7) // 8) and costs.min() > 0   # every tile ran


def test_aborted_direct_launch_is_reported_and_the_handle_recovers(ca):
    """Host delivery keeps one completion counter per group of tiles.  A launch that is cut short (here: a dispatch order
    whose second half names no tile, ctr_debug_poison_next_order) must not pass for a frame — ctr_render returns
    CTR_E_DELIVERY — and must not poison the handle: the counters are cleared at the head of every launch, so the next
    render on the SAME handle is complete and bit-identical to the device-buffer path."""
    from cutrace_amd import _lib
    L = _lib.hip_lib()
    s = load_scene(ca, "bunny", 640, 360)
    ds = ca.DeviceScene(s)
    want = ds.render(bounces=3)                      # pageable destination: device buffers + copies
    for _ in range(2):
        ds.render(bounces=3, pinned=True)
    assert L.ctr_debug_poison_next_order(ds._h) == 0
    with pytest.raises(RuntimeError, match="not delivered"):
        ds.render(bounces=3, pinned=True)
    for _ in range(3):                               # first launch after the abort, then the re-sorted order
        got = ds.render(bounces=3, pinned=True)
        for k in ("depth", "normal", "color"):
            assert same_bits(got[k], want[k]), k
        assert got["ray_count"] == want["ray_count"]
    # the same abort on the device-buffer path renders half a frame without an error code (nothing is delivered by
    # the kernel there), and the handle is just as usable afterwards
    assert L.ctr_debug_poison_next_order(ds._h) == 0
    ds.render(bounces=3)
    got = ds.render(bounces=3)
    for k in ("depth", "normal", "color"):
        assert same_bits(got[k], want[k]), k
    with pytest.raises(RuntimeError):
        ds.set_variant(1)                            # a variant bit round 2 removed: now rejected, not ignored


def test_delivery_self_check(ca, monkeypatch):
    """CUTRACE_VERIFY_DELIVERY=1 makes ctr_render compare what the kernel delivered into page-locked memory with the staged
    frame it still holds on the device, pixel by pixel (a debug aid): it must pass, at full size and at a ragged one."""
    monkeypatch.setenv("CUTRACE_VERIFY_DELIVERY", "1")
    for (w, h) in ((1920, 1080), (333, 131)):
        s = load_scene(ca, "bunny", w, h)
        ds = ca.DeviceScene(s)
        for _ in range(2):
            r = ds.render(bounces=3, pinned=True)     # raises if the self-check fails
        assert r["ray_count"] > 0


def _uv_close(got, want, tol=1e-4):
    """uv within the parity bar; NaN where the reference has NaN (plane normal without x and y, default_schema.hpp:170)"""
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), f"{int((nan_g != nan_w).sum())} uv values are NaN on one side only"
    d = np.abs(np.where(nan_w, 0, got) - np.where(nan_w, 0, want))
    lim = tol * np.maximum(1.0, np.abs(np.where(nan_w, 0, want)))
    assert (d <= lim).all(), f"uv differs by up to {float(d.max()):.3e}"
    return float(d.max())


@pytest.mark.parametrize("name,w,h", [("triangle", 20, 20), ("sphere_plane", 96, 54), ("bunny", 96, 54)])
def test_texture_coordinates_against_reference_fixture(ca, name, w, h):
    """ctr_render_uv: the fourth output (ray_cast's tex_coords of the primary hit) against the reference build's buffers
    (tests/golden/uv_*.npz): triangle, plane, sphere and mesh formulas.  Triangle / plane / mesh coordinates are plain
    float arithmetic on the hit point and come out bit-identical; the sphere's go through atan2f / asinf, whose device
    versions differ from glibc's in the last bits (bar: 1e-4, the north star's per-channel tolerance).  The other three
    buffers are those of ctr_render, bit for bit."""
    g = np.load(os.path.join(GOLD, f"uv_{name}_{w}x{h}.npz"))
    s = load_scene(ca, name, w, h)
    ds = ca.DeviceScene(s)
    r = ds.render_uv(bounces=0)
    worst = _uv_close(r["uv"], g["uv"])
    hit = g["hit_id"]
    kinds = {0: "triangle", 1: "mesh", 2: "plane", 3: "sphere"}
    types = [int(o.type) for o in s.desc.contents.objects[:s.desc.contents.n_objects]]
    exact = np.ones(hit.shape, bool)
    for i, t in enumerate(types):
        if kinds[t] == "sphere":
            exact &= hit != i
    a, b = r["uv"][exact], g["uv"][exact]
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), \
        "uv of non-sphere hits must be bit-identical"
    plain = ds.render(bounces=0)
    for k in ("depth", "normal", "color"):
        assert same_bits(r[k], plain[k]), k
    print(f"uv {name}: max |diff| {worst:.2e} ({int((~exact).sum())} sphere pixels through atan2f/asinf)")


def test_texture_coordinates_random_scenes_vs_oracle(ca):
    """the same against the oracle on seeded scenes that mix every primitive (stand-alone triangles included)"""
    for seed in range(6):
        s = ca.HostScene.parse(_random_scene(seed, w=96, h=64))
        o = oracle.oracle_render(s, bounces=1, threads=os.cpu_count() or 4, uv=True)
        r = ca.DeviceScene(s).render_uv(bounces=1)
        _uv_close(r["uv"], o["uv"])
        assert same_bits(r["depth"], o["depth"]) and same_bits(r["normal"], o["normal"])


def _axis_plane_scene(case, w=64, h=40):
    """Scenes for the axis-aligned plane path (render_kernel.hip "axis-aligned planes"; scene_device.h DPlanePair)."""
    import json
    mats = [{"type": "solid", "color": [0.9, 0.3, 0.2], "specular": 0.4, "reflect": 0.5, "phong": 20.0, "transparency": 0.0},
            {"type": "solid", "color": [0.2, 0.8, 0.3], "specular": 0.1, "reflect": 0.0, "phong": 1.0, "transparency": 0.0},
            {"type": "solid", "color": [0.3, 0.3, 0.9], "specular": 0.7, "reflect": 0.3, "phong": 300.0,
             "transparency": 0.5 if case in ("many_per_axis", "underflow") else 0.0}]
    lights = [{"type": "point", "point": [0.5, 1.5, -1.0], "color": [1, 1, 1]}, {"type": "sun", "direction": [0.3, -1, 0.2], "color": [0.5, 0.5, 0.4]}]
    eye = [0.0, 0.0, -3.5]
    objs = [{"type": "sphere", "center": [0.3, -0.2, 0.5], "radius": 0.8, "material": 0}]
    if case in ("box", "eye_on_plane", "fudge_zero", "fudge_negative"):
        # a closed room; zeros of both signs in the normals; the middle column of an even-width image has dir.x == 0
        objs += [{"type": "plane", "point": [-3, 0, 0], "normal": [1, -0.0, 0], "material": 1},
                 {"type": "plane", "point": [3, 0, 0], "normal": [-1, 0, -0.0], "material": 2},
                 {"type": "plane", "point": [0, -2, 0], "normal": [-0.0, 1, 0], "material": 1},
                 {"type": "plane", "point": [0, 2.5, 0], "normal": [0, -2, -0.0], "material": 0},
                 {"type": "plane", "point": [0, 0, 4], "normal": [0, 0, -1], "material": 2},
                 {"type": "plane", "point": [0, 0, -4], "normal": [-0.0, -0.0, 0.5], "material": 1}]
        if case == "eye_on_plane":
            eye = [0.0, -2.0, -3.5]  # on the floor plane: a zero numerator for every primary ray
    elif case == "eye_near_flt_max":
        # the eye so high that (point - origin).y overflows to -inf for walls normal to x and z — the reference then
        # multiplies inf by the normal's zero into a NaN and MISSES those walls; a one-product numerator would report hits
        eye = [0.0, 3.39e38, -3.5]
        objs += [{"type": "plane", "point": [-3, -1e37, 0], "normal": [1, -0.0, 0], "material": 1},
                 {"type": "plane", "point": [3, -1e37, 0], "normal": [-1, 0, -0.0], "material": 2},
                 {"type": "plane", "point": [0, -1e37, 4], "normal": [0, 0, -1], "material": 2},
                 {"type": "plane", "point": [0, 3.0e38, 0], "normal": [0, 1, 0], "material": 1}]
    elif case == "one_axis_only":
        objs += [{"type": "plane", "point": [0, -1.5, 0], "normal": [0, 1, 0], "material": 1},
                 {"type": "plane", "point": [1, 0, 5], "normal": [0.2, 0.1, -1], "material": 2}]
    elif case == "many_per_axis":
        # five planes normal to y (three triples), one to x, two general ones, some transparent (the ordered shadow loop)
        for k, y in enumerate((-1.0, -1.5, -2.5, 3.0, 4.0)):
            objs.append({"type": "plane", "point": [k, y, -k], "normal": [0, 1.0 if y < 0 else -3.0, 0], "material": k % 3})
        objs += [{"type": "plane", "point": [4, 0, 0], "normal": [-1, 0, 0], "material": 2},
                 {"type": "plane", "point": [0, 0, 6], "normal": [0.1, 0, -1], "material": 1},
                 {"type": "plane", "point": [-5, 0, 0], "normal": [1, 0.05, 0.02], "material": 0}]
    elif case == "underflow":
        # normals so short that products underflow to zeros and denormals
        objs += [{"type": "plane", "point": [0, -1.5, 0], "normal": [0, 1e-38, 0], "material": 1},
                 {"type": "plane", "point": [0, 0, 5], "normal": [0, 0, -1e-30], "material": 2},
                 {"type": "plane", "point": [-3, 1e-9, 0], "normal": [3e-39, 0, 0], "material": 0}]
    cam = {"eye": eye, "up": [0, 1, 0], "look": [0.1, -0.2, 1.0] if case == "eye_near_flt_max" else [eye[0], eye[1], 0.0], "near_plane": 0.1, "far_plane": 100.0, "width": w, "height": h,
           "ambient": 0.15}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


@pytest.mark.parametrize("case", ["box", "eye_on_plane", "one_axis_only", "many_per_axis", "underflow", "fudge_zero", "fudge_negative",
                                  "eye_near_flt_max"])
def test_axis_aligned_planes(gpu, case):
    """Planes with exactly one non-zero normal component take a three-instruction path for numerator and denominator
    (the products with zero left out); with fudge <= 0 the kernel must fall back to the reference's full expression
    (a zero's sign could then decide).  Everything against the oracle, and bitwise against the kernel without shortcuts."""
    s = gpu.HostScene.parse(_axis_plane_scene(case))
    assert s.ok, s.error
    fudge = {"fudge_zero": 0.0, "fudge_negative": -0.25}.get(case, 1e-3)
    _check_all_ways(gpu, s, f"axis-aligned planes: {case}", bounces=3, fudge=fudge)


def test_ignore_transparent_primary_cast(ca):
    """CTR_VAR_IGNORE_TRANSPARENT: the cast of kernel.hpp:52 made with ray_cast's ignore_transparent = true
    (inc/ray_cast.hpp:30,39-40).  Depth, normal and texture coordinates then come from a scene without its transparent
    objects; the colour is that of the plain render (ray_color's own casts pass false).  Against the reference-build fixture,
    the oracle on random scenes (every object type with transparent materials), and the ray count (unchanged)."""
    g = np.load(os.path.join(GOLD, "ignore_transparent_sphere_plane_96x54_b5.npz"))
    s = load_scene(ca, "sphere_plane", 96, 54)
    ds = ca.DeviceScene(s)
    plain = ds.render(bounces=5)
    ds.set_variant(ca.VAR_IGNORE_TRANSPARENT)
    r = ds.render_uv(bounces=5)
    assert same_bits(r["depth"], g["depth"]) and same_bits(r["normal"], g["normal"])
    assert np.abs(r["color"].astype(np.float64) - g["color"].astype(np.float64)).max() <= 1e-4
    assert same_bits(r["color"], plain["color"]) and r["ray_count"] == int(g["ray_count"]) == plain["ray_count"]
    assert _uv_close(r["uv"], g["uv"])
    assert int((r["depth"].view(np.uint32) != plain["depth"].view(np.uint32)).sum()) > 300
    r2 = ds.render(bounces=5)           # the three-buffer call honours the bit too
    assert same_bits(r2["depth"], r["depth"]) and same_bits(r2["normal"], r["normal"]) and same_bits(r2["color"], r["color"])
    with pytest.raises(RuntimeError):   # not for the device-buffer calls
        ds.render_device(1, 1, 1)
    ds.close()
    for seed in range(6):
        sc = ca.HostScene.parse(_random_scene(seed + 40, w=80, h=48))
        assert sc.ok
        o = oracle.oracle_render(sc, bounces=3, threads=os.cpu_count() or 4, uv=True, ignore_transparent_primary=True)
        x = ca.DeviceScene(sc)
        x.set_variant(ca.VAR_IGNORE_TRANSPARENT | ca.VAR_EXACT_POW)
        got = x.render_uv(bounces=3)
        assert_parity(got, o, what=f"ignore_transparent seed {seed}")
        assert got["ray_count"] == o["ray_count"]
        x.close()


def test_ignore_transparent_full_resolution_against_golden_samples(ca):
    """CTR_VAR_IGNORE_TRANSPARENT at BASELINE's full size (sphere_plane.json @1920x1080) against 4096 sampled pixels and the checksums of the
    frame the reference build rendered with its ray_cast called with `true`: depth / normal bit-exact, uv within 1e-4, colour within 1e-4 (and
    bit-equal to the plain render's: ray_color's own casts pass false)."""
    g = np.load(os.path.join(GOLD, "full_ignore_transparent_sphere_plane_1920x1080_b5.npz"))
    s = load_scene(ca, "sphere_plane")
    ds = ca.DeviceScene(s)
    plain = ds.render(bounces=5)
    ds.set_variant(ca.VAR_IGNORE_TRANSPARENT)
    r = ds.render_uv(bounces=5)
    idx = g["sample_idx"]
    assert same_bits(r["depth"].reshape(-1)[idx], g["depth"]) and same_bits(r["normal"].reshape(-1, 3)[idx], g["normal"])
    assert np.abs(r["color"].reshape(-1, 3)[idx].astype(np.float64) - g["color"].astype(np.float64)).max() <= 1e-4
    assert _uv_close(r["uv"].reshape(-1, 2)[idx], g["uv"])
    assert same_bits(r["color"], plain["color"]) and r["ray_count"] == int(g["ray_count"]) == 13973091
    fin = np.isfinite(r["depth"])
    assert int(fin.sum()) == int(g["n_finite"])
    assert abs(float(r["depth"][fin].astype(np.float64).sum()) - float(g["sum_depth"])) < 1e-6 * abs(float(g["sum_depth"]))
    assert int((r["depth"].view(np.uint32) != plain["depth"].view(np.uint32)).sum()) > 100000   # the glass sphere is gone from the depth map
    ds.close()
