"""The merged walk (round 4, CTR_VAR_MERGE — opt-in: measured slower than the two-level walk on every shipped config,
profiles/r04/exp_merged_tree_ab.txt): scenes with several meshes are walked through ONE four-wide tree over the triangles of all
meshes (render_kernel.hip "merged walk", ctr_api.cpp ctr_scene::Merged) instead of a tree over the meshes' boxes and
then each mesh's own tree.  What the reference does per mesh — its AABB test before any triangle
(inc/default_schema.hpp:99-114,126), "first mesh in scene order wins ties" and "a mesh whose nearest valid t equals
min_t is rejected whole" (inc/ray_cast.hpp:43) — must survive that: every case against the oracle, and bit for bit
against the two-level walk (the default) and the plain linear walk."""
import json
import os

import numpy as np
import pytest

import oracle
from tests.util import assert_parity, same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats_render(ca, ds, **kw):
    """One CTR_VAR_STATS render: how many wave casts went through the merged walk, how many it handed back."""
    ds.set_variant(ca.VAR_STATS | ca.VAR_MERGE)
    ca.DeviceScene.lane_stats(reset=True)
    ds.render(**kw)
    st = ca.DeviceScene.lane_stats(reset=True)
    ds.set_variant(ca.VAR_AUTO)
    return st["merged_walks"], st["merged_walks_redone"]


def _all_ways(ca, s, what, bounces=3, fudge=1e-3, expect_merged=True):
    o = oracle.oracle_render(s, bounces=bounces, fudge=fudge, threads=os.cpu_count() or 4)
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_MERGE)
    r = ds.render(bounces=bounces, fudge=fudge)
    assert_parity(r, o, what=what)
    assert r["ray_count"] == o["ray_count"], what
    for variant in (ca.VAR_AUTO, ca.VAR_MERGE | ca.VAR_EXACT_POW, ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT):
        ds.set_variant(variant)
        other = ds.render(bounces=bounces, fudge=fudge)
        for k in ("depth", "normal") + (() if variant & ca.VAR_EXACT_POW else ("color",)):
            assert same_bits(r[k], other[k]), (what, variant, k)
        assert other["ray_count"] == r["ray_count"]
    walks, redone = _stats_render(ca, ds, bounces=bounces, fudge=fudge)
    assert (walks > 0) == expect_merged, (what, walks)
    ds.close()
    return o, r, walks, redone


def _multi_mesh_scene(tmp_path, seed, w=96, h=64, opaque=False, n_mesh=4):
    """n_mesh meshes cut out of scene/bunny.stl and scene/skull.stl, translated so that their boxes overlap, at random
    places of the object list, mixed with planes, a sphere and a stand-alone triangle."""
    from cutrace_amd import scenes
    rng = np.random.RandomState(seed)
    src = [scenes.read_stl(os.path.join(ROOT, "scene", f)) for f in ("bunny.stl", "skull.stl")]
    mats = [{"type": "solid", "color": [float(x) for x in rng.uniform(0.1, 1, 3)], "specular": float(rng.uniform(0, 1)),
             "reflect": float(rng.choice([0.0, 0.3, 0.8])), "phong": float(rng.choice([1.0, 20.0, 200.0])),
             "transparency": 0.0 if opaque else float(rng.choice([0.0, 0.0, 0.4]))} for _ in range(4)]
    objs = [{"type": "plane", "point": [0, -1.3, 0], "normal": [0, 1, 0], "material": 0},
            {"type": "plane", "point": [0, 0, -4], "normal": [0, 0, 1], "material": 1},
            {"type": "sphere", "center": [1.2, 0.4, -0.5], "radius": 0.5, "material": 2},
            {"type": "triangle", "p1": [-2, -1, -1], "p2": [-1, 1.5, -1.5], "p3": [-2.5, 1, 0], "material": 3}]
    for m in range(n_mesh):
        t = src[int(rng.randint(2))]
        t = t[rng.rand(len(t)) < rng.uniform(0.2, 0.7)]           # a random part of the mesh (open surface)
        c = t.reshape(-1, 3).mean(0)
        scale = np.float32(1.2 / np.abs(t.reshape(-1, 3) - c).max())
        t = ((t - c) * scale + np.float32(rng.uniform(-0.7, 0.7, 3))).astype(np.float32)
        path = str(tmp_path / f"mm_{seed}_{m}.stl")
        scenes.write_stl(path, t)
        objs.insert(int(rng.randint(len(objs) + 1)), {"type": "mesh", "file": path, "material": int(rng.randint(4))})
    if rng.rand() < 0.5:   # the same mesh twice: exact ties on t between two MESHES (the first in scene order wins)
        first = next(o for o in objs if o["type"] == "mesh")
        objs.append(dict(first, material=int(rng.randint(4))))
    lights = [{"type": "sun", "direction": [float(x) for x in rng.uniform(-1, 1, 3)], "color": [0.7, 0.7, 0.7]},
              {"type": "point", "point": [float(x) for x in rng.uniform(-3, 3, 3)], "color": [0.6, 0.5, 0.4]}]
    cam = {"eye": [float(rng.uniform(-1, 1)), float(rng.uniform(-0.3, 1.0)), 4.0], "up": [0, 1, 0],
           "look": [float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.1)), -1.0], "near_plane": 0.1, "far_plane": 100.0,
           "width": w, "height": h, "ambient": 0.15}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_scenes_with_several_meshes(ca, tmp_path, seed):
    s = ca.HostScene.parse(_multi_mesh_scene(tmp_path, seed, opaque=(seed % 2 == 0), n_mesh=2 + seed % 4))
    assert s.ok
    _all_ways(ca, s, f"multi-mesh seed {seed}", bounces=[1, 3, 5][seed % 3])


def _axis_scene(tmp_path, name, eye, look, meshes, w, h):
    from cutrace_amd import scenes
    objs = []
    for k, tris in enumerate(meshes):
        path = str(tmp_path / f"{name}_{k}.stl")
        scenes.write_stl(path, np.asarray(tris, np.float32))
        objs.append({"type": "mesh", "file": path, "material": k % 2})
    mats = [{"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.0, "phong": 40},
            {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.0, "phong": 10}]
    lights = [{"type": "sun", "direction": [-0.3, -1.0, -0.2], "color": [0.6, 0.6, 0.6]}]
    cam = {"eye": eye, "up": [0, 1, 0], "look": look, "near_plane": 0.1, "far_plane": 100.0, "width": w, "height": h, "ambient": 0.2}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


def test_mesh_whose_nearest_valid_t_equals_min_t_is_rejected_whole(ca, tmp_path):
    """inc/ray_cast.hpp:43 accepts an object only if dist > min_dist (strict), while triangle::intersect accepts
    min_t <= t0 (inc/default_schema.hpp:68): a mesh whose NEAREST valid triangle sits at t0 == min_t exactly is rejected
    whole — also its farther triangles.  The centre pixel's ray is exactly (0, 0, -1); with fudge = 1 mesh 0 (triangles at
    z = -1 and z = -2) vanishes for it and mesh 1 (z = -3) is what it sees.  The merged walk cannot know that while it
    walks (the farther triangle of mesh 0 looks like a hit): it must notice and hand the cast to the two-level walk."""
    w, h = 64, 48
    big = lambda z: [[-4, -4, z], [4, -4, z], [0, 5, z]]
    s = ca.HostScene.parse(_axis_scene(tmp_path, "mint", [0, 0, 0], [0, 0, -1], [[big(-1.0), big(-2.0)], [big(-3.0)]], w, h))
    assert s.ok
    o, r, walks, redone = _all_ways(ca, s, "t0 == min_t", bounces=1, fudge=1.0)
    assert o["depth"][h // 2, w // 2] == 3.0            # the reference's rule in action (mesh 0 rejected whole there)
    assert o["depth"][h // 2, w // 2 + 1] < 1.01        # ... and only there
    assert redone > 0


def test_mesh_whose_box_test_fails_is_missed_whatever_its_triangles_say(ca, tmp_path):
    """mesh::intersect tests the mesh's box first (inc/default_schema.hpp:126).  For the rays of the image's centre
    column dir.z is exactly 0 and the eye lies exactly in the box's z-min plane: (bmin.z - start.z) * (1 / 0) = NaN on the
    LAST axis leaves tmin = tmax = NaN and `tmin <= tmax` false — the mesh is missed although those rays meet the
    triangle's edge in that plane (gamma = -0, accepted).  The merged walk finds that triangle; the mesh's own box test
    afterwards must take it away again (and the cast is redone through the two-level walk)."""
    w, h = 64, 48
    edge_on = [[2, -1, 0], [2, 1, 0], [2, 0, 1]]         # edge (2,-1,0)-(2,1,0) in the plane z = 0; box z in [0, 1]
    behind = [[5, -9, -9], [5, 9, -9], [5, 0, 12]]
    s = ca.HostScene.parse(_axis_scene(tmp_path, "nanbox", [0, 0, 0], [1, 0, 0], [[edge_on], [behind]], w, h))
    assert s.ok
    o, r, walks, redone = _all_ways(ca, s, "NaN box test", bounces=1)
    col = o["depth"][:, w // 2]
    assert np.all(col[np.isfinite(col)] > 4.0) and np.isfinite(col).any()   # the centre column sees only the mesh behind
    assert (o["depth"][:, w // 2 + 1] < 3.0).any() or (o["depth"][:, w // 2 - 1] < 3.0).any()   # its neighbours see the near triangle
    assert redone > 0


def test_more_meshes_than_the_key_has_room_for(ca, tmp_path):
    """The merged tree's tie-break key keeps the mesh's rank in 8 bits: 255 meshes are merged, 256 fall back to the
    two-level walk.  One-triangle meshes, many of them coincident (ties between meshes: the first in scene order wins)."""
    rng = np.random.RandomState(5)
    base = rng.uniform(-1.5, 1.5, (40, 3, 3)).astype(np.float32)
    base[:, :, 2] -= 1.0
    for n, merged in ((255, True), (256, False)):
        meshes = [[base[k % 40]] for k in range(n)]
        s = ca.HostScene.parse(_axis_scene(tmp_path, f"many{n}", [0.1, 0.2, 4.0], [0, 0, -1], meshes, 48, 32))
        assert s.ok
        _all_ways(ca, s, f"{n} meshes", bounces=1, expect_merged=merged)


@pytest.mark.parametrize("seed", [1, 2])
def test_rays_coplanar_with_triangles_of_two_meshes(ca, tmp_path, seed):
    """The guard records of the merged tree: triangles lying in the plane that contains every primary ray of one image
    row (tests/test_gpu_parity.py::_coplanar_scene), split over two meshes."""
    from tests.test_gpu_parity import _coplanar_scene
    from cutrace_amd import scenes
    w, h, row = 384, 32, 9
    one = _coplanar_scene(ca, tmp_path, w, h, row, 48, seed)
    d = one.desc.contents
    tris = np.array([[[getattr(getattr(d.triangles[k], p), c) for c in "xyz"] for p in ("p1", "p2", "p3")]
                     for k in range(d.n_triangles)], np.float32)
    a, b = str(tmp_path / f"cop_a{seed}.stl"), str(tmp_path / f"cop_b{seed}.stl")
    scenes.write_stl(a, tris[0::2])
    scenes.write_stl(b, tris[1::2])
    lights = [{"type": "point", "point": [float(d.lights[k].v.x), float(d.lights[k].v.y), float(d.lights[k].v.z)],
               "color": [0.8, 0.8, 0.8]} for k in range(d.n_lights)]   # (the second one lies IN the plane, like the eye)
    sc = {"camera": {"eye": [0.3, 0.8, 4.0], "up": [0, 1, 0], "look": [-0.05, -0.15, -1.0], "near_plane": 0.1, "far_plane": 100.0,
                     "width": w, "height": h, "ambient": 0.1},
          "lights": lights,
          "materials": [{"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.3, "phong": 40},
                        {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.2, "phong": 10}],
          "objects": [{"type": "mesh", "file": a, "material": 0}, {"type": "plane", "point": [0, -1.0, 0], "normal": [0, 1, 0], "material": 1},
                      {"type": "mesh", "file": b, "material": 1}, {"type": "plane", "point": [0, 0, -6.0], "normal": [0, 0, 1], "material": 1}]}
    s = ca.HostScene.parse(json.dumps(sc))
    assert s.ok
    _all_ways(ca, s, f"coplanar, two meshes, seed {seed}", bounces=2)
