"""The occluder maps (round 4; cutrace_amd/csrc/occl.h, render_kernel.hip "occluder map"): a shadow ray whose receiver is nearer to its
point light than every mesh triangle seen in its direction skips the meshes.  The reference's result must not move: every case against the
oracle, and bit for bit against the kernel without the maps (CTR_VAR_NO_OCCLUDER_MAP) and the plain linear walk."""
import json
import os

import numpy as np
import pytest

import oracle
from tests.util import assert_parity, same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_ways(ca, s, what, bounces=3, fudge=1e-3):
    o = oracle.oracle_render(s, bounces=bounces, fudge=fudge, threads=os.cpu_count() or 4)
    ds = ca.DeviceScene(s)
    r = ds.render(bounces=bounces, fudge=fudge)
    assert_parity(r, o, what=what)
    assert r["ray_count"] == o["ray_count"], what
    for variant in (ca.VAR_NO_OCCLUDER_MAP, ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT, ca.VAR_MERGE):
        ds.set_variant(variant)
        other = ds.render(bounces=bounces, fudge=fudge)
        for k in ("depth", "normal", "color"):
            assert same_bits(r[k], other[k]), (what, variant, k)
    ds.set_variant(ca.VAR_STATS)
    ca.DeviceScene.lane_stats(reset=True)
    ds.render(bounces=bounces, fudge=fudge)
    st = ca.DeviceScene.lane_stats(reset=True)["shadow_casts_at_meshes"]
    ds.close()
    return o, r, st


def _scene(tmp_path, name, meshes, lights, eye=(0.3, 0.8, 4.0), look=(-0.05, -0.15, -1.0), w=96, h=64, transparent=False, extra=()):
    from cutrace_amd import scenes
    objs = [{"type": "plane", "point": [0, -1.5, 0], "normal": [0, 1, 0], "material": 1},
            {"type": "plane", "point": [0, 0, -5], "normal": [0, 0, 1], "material": 1}] + list(extra)
    for k, tris in enumerate(meshes):
        path = str(tmp_path / f"{name}_{k}.stl")
        scenes.write_stl(path, np.asarray(tris, np.float32).reshape(-1, 3, 3))
        objs.insert(k % 2, {"type": "mesh", "file": path, "material": 0 if k % 2 == 0 else 2})
    mats = [{"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.3, "phong": 40, "transparency": 0.4 if transparent else 0.0},
            {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.2, "phong": 10},
            {"type": "solid", "color": [0.5, 0.9, 0.4], "specular": 0.3, "reflect": 0.0, "phong": 5}]
    cam = {"eye": list(eye), "up": [0, 1, 0], "look": list(look), "near_plane": 0.1, "far_plane": 100.0, "width": w, "height": h, "ambient": 0.1}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


def _bunny(scale=1.0, shift=(0, 0, 0)):
    from cutrace_amd import scenes
    t = scenes.read_stl(os.path.join(ROOT, "scene", "bunny.stl"))
    c = t.reshape(-1, 3).mean(0)
    return ((t - c) * np.float32(scale) + c + np.float32(shift)).astype(np.float32)


def test_the_maps_are_consulted_and_change_nothing(ca, tmp_path):
    """bunny.json itself: most shadow casts that reach the mesh are taken out by the maps; the frame is bit for bit the one without them."""
    s = ca.HostScene.load("scene/bunny.json")
    s.set_size(320, 180)
    o, r, st = _all_ways(ca, s, "bunny.json", bounces=5)
    assert st["skipped_whole_by_occluder_map"] > 0.3 * st["wave_casts"], st


@pytest.mark.parametrize("case", ["light_inside_box", "light_on_triangle", "light_in_a_triangles_plane", "cage_around_light", "sun_and_point",
                                  "transparent_meshes", "many_lights", "receiver_between"])
def test_adversarial_lights_and_meshes(ca, tmp_path, case):
    point = lambda p: {"type": "point", "point": list(p), "color": [0.8, 0.8, 0.8]}
    bun = _bunny(1.6)
    extra, transparent, lights, meshes = (), False, [point((1.5, 2.5, 2.0))], [bun]
    if case == "light_inside_box":        # the light between the bunny's ears: inside its AABB, triangles in every direction
        c = bun.reshape(-1, 3).mean(0)
        lights = [point((float(c[0]), float(bun[..., 1].max()) - 0.05, float(c[2]))), point((1.5, 2.5, 2.0))]
    elif case == "light_on_triangle":     # distance 0: the builder must give the map up, not divide by zero
        p = bun[17].mean(0)
        lights = [point([float(x) for x in p])]
    elif case == "light_in_a_triangles_plane":   # the in-plane regime of ctr_api.cpp refresh_linear_meshes: this light's map is switched off
        a, b, c = bun[40]
        q = a + 3.0 * (b - a) + 2.5 * (c - a)
        lights = [point([float(x) for x in q]), point((1.5, 2.5, 2.0))]
    elif case == "cage_around_light":     # twelve large triangles around the light, each spanning several cube faces, with holes
        L = np.array([0.5, 1.0, 1.0], np.float32)
        cage = []
        rng = np.random.RandomState(3)
        for k in range(12):
            v = rng.normal(size=(3, 3)).astype(np.float32)
            v = v / np.linalg.norm(v, axis=1, keepdims=True) * np.float32(0.8 + 0.1 * k)
            cage.append(L + v)
        meshes = [bun, np.asarray(cage, np.float32)]
        lights = [point([float(x) for x in L])]
    elif case == "sun_and_point":
        lights = [{"type": "sun", "direction": [-0.3, -1.0, -0.2], "color": [0.5, 0.5, 0.5]}, point((1.5, 2.5, 2.0)), point((-2.0, 0.5, 3.0))]
    elif case == "transparent_meshes":    # no any-hit build: the ordered shadow loop walks through the glass bunny, cast after cast
        transparent = True
        meshes = [bun, _bunny(0.8, (1.2, 0.3, 0.8))]
        lights = [point((1.5, 2.5, 2.0)), point((-1.5, 0.2, 3.0))]
    elif case == "many_lights":
        lights = [point((np.cos(a) * 2.5, 1.0 + 0.5 * np.sin(3 * a), np.sin(a) * 2.5 + 0.5)) for a in np.linspace(0, 6.0, 7)]
    elif case == "receiver_between":      # receivers nearer to the light than the mesh AND behind it, planes in front of and behind the bunny
        extra = [{"type": "plane", "point": [0, 0, 2.2], "normal": [0.3, 0.2, -1], "material": 2},
                 {"type": "sphere", "center": [0.8, 0.9, 1.4], "radius": 0.3, "material": 2}]
        lights = [point((0.9, 1.8, 1.9))]
    s = ca.HostScene.parse(_scene(tmp_path, case, meshes, lights, transparent=transparent, extra=extra))
    assert s.ok
    _all_ways(ca, s, case, bounces=2)
