"""Scene-JSON loader + STL reader + image writers (host side, CPU only).
Loader semantics follow inc/loader.hpp:679-780 and inc/default_schema.hpp:463-898."""
import ctypes as C
import json
import os
import struct

import numpy as np

import oracle

from cutrace_amd import _lib
from tests.conftest import load_scene


def test_shipped_scenes_load(ca):
    counts = {"triangle": (1, 1, 1, 0), "sphere_plane": (4, 2, 3, 0), "bunny": (6, 4, 6, 1000), "mirror": (8, 2, 5, 924)}
    for name, (no, nl, nm, nt) in counts.items():
        s = load_scene(ca, name)
        d = s.desc.contents
        assert (d.n_objects, d.n_lights, d.n_materials, d.n_triangles) == (no, nl, nm, nt), name
    s = load_scene(ca, "triangle")
    assert s.size == (20, 20)


def test_stale_schema_scene_fails_like_the_reference(ca, capfd):
    """scene/bunny_small.json uses the stale schema.md keys and must fail."""
    s = ca.HostScene.load("scene/bunny_small.json")
    assert not s.ok
    err = capfd.readouterr().err
    assert "Error while loading object #0: Type 'model' is invalid." in err
    assert "Error while loading light #0: Cannot find key 'point' in object." in err
    assert "Error while loading material #0: Cannot find key 'type' in object." in err
    assert "Could not find 'camera' object or it's invalid: Cannot find key 'ambient' in object.." in err


BASE = {
    "camera": {"eye": [0, 0, -5], "up": [0, 1, 0], "look": [0, 0, 0], "near_plane": 0.1, "far_plane": 10,
               "width": 8, "height": 4, "ambient": 0.2},
    "lights": [{"type": "sun", "direction": [1, 1, 1]}],
    "materials": [{"type": "solid", "color": [0.5, 0.25, 1]}],
    "objects": [{"type": "sphere", "center": [0, 0, 0], "radius": 1, "material": 0}],
}


def test_defaults_and_casts(ca):
    s = ca.HostScene.parse(json.dumps(BASE))
    assert s.ok
    d = s.desc.contents
    m, l = d.materials[0], d.lights[0]
    assert (m.specular, m.reflexivity, m.phong_exp, m.transparency) == (np.float32(0.3), 0.0, 32.0, 0.0)
    assert l.color.tup() == (1.0, 1.0, 1.0)
    assert (d.cam.w, d.cam.h) == (8, 4)
    assert d.cam.forward.tup() == (0.0, 0.0, 1.0)
    assert d.cam.right.tup() == (-1.0, 0.0, 0.0)  # forward x up


def test_error_messages(ca, capfd):
    bad = json.loads(json.dumps(BASE))
    bad["objects"] = [
        {"type": "sphere", "center": [0, 0], "radius": 1, "material": 0},
        {"type": "sphere", "center": [0, 0, "x"], "radius": 1, "material": 0},
        {"type": "plane", "point": [0, 0, 0], "material": 0},
        {"type": "sphere", "center": [0, 0, 0], "radius": "big", "material": 0},
        17,
        {"center": [0, 0, 0]},
    ]
    del bad["lights"]
    s = ca.HostScene.parse(json.dumps(bad))
    assert not s.ok
    err = capfd.readouterr().err
    assert "object #0: Expected a 3-value array, got 2 instead." in err
    assert "object #1: Expected a value of type float." in err
    assert "object #2: Cannot find key 'normal' in object." in err
    assert "object #3: Expected a value of type float." in err
    assert "object #4: Value is not a JSON object." in err
    assert "object #5: Cannot find key 'type' in object." in err
    assert "Could not find 'lights' array: Cannot find key 'lights' in object.." in err
    assert s.desc.contents.n_objects == 0


def test_syntax_error_is_a_failure(ca, capfd):
    s = ca.HostScene.parse('{"camera": ')
    assert not s.ok
    assert "Error while loading file" in capfd.readouterr().err


def test_stl_roundtrip_and_bounds(ca, tmp_path):
    L = _lib.host_lib()
    n = L.ctr_stl_read(b"scene/bunny.stl", None, 0)
    assert n == 1000
    tris = (_lib.Triangle * n)()
    assert L.ctr_stl_read(b"scene/bunny.stl", tris, n) == n
    raw = open(os.path.join(_lib.ROOT, "scene/bunny.stl"), "rb").read()
    v = struct.unpack_from("<9f", raw, 84 + 12)
    assert (tris[0].p1.tup(), tris[0].p2.tup(), tris[0].p3.tup()) == (v[0:3], v[3:6], v[6:9])
    out = str(tmp_path / "x.stl").encode()
    assert L.ctr_stl_write(out, tris, n) == 0
    tris2 = (_lib.Triangle * n)()
    assert L.ctr_stl_read(out, tris2, n) == n
    assert bytes(tris) == bytes(tris2)
    mn, mx = _lib.Vec3(), _lib.Vec3()
    L.ctr_mesh_bounds(tris, n, C.byref(mn), C.byref(mx))
    arr = np.frombuffer(bytes(tris), np.float32).reshape(-1, 3)
    assert np.array_equal(np.array(mn.tup(), np.float32), arr.min(0))
    assert np.array_equal(np.array(mx.tup(), np.float32), arr.max(0))
    # the loader stores the same AABB in the mesh object
    s = load_scene(ca, "bunny")
    o = s.desc.contents.objects[0]
    assert o.v0.tup() == mn.tup() and o.v1.tup() == mx.tup()


def test_quantisation_matches_oracle(ca):
    rng = np.random.RandomState(7)
    n = 5000
    depth = rng.uniform(0.1, 9, n).astype(np.float32)
    depth[::17] = np.inf
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    normal[::13] = 0
    color = rng.uniform(-0.2, 1.3, (n, 3)).astype(np.float32)
    H, O = _lib.host_lib(), oracle.oracle_lib()
    for fn_h, fn_o, src, extra in (
        (H.ctr_quantise_depth, O.orc_quantise_depth, depth, (C.c_float(9.0),)),
        (H.ctr_quantise_normal, O.orc_quantise_normal, normal, ()),
        (H.ctr_quantise_color, O.orc_quantise_color, color, ()),
    ):
        a, b = np.zeros((n, 3), np.uint8), np.zeros((n, 3), np.uint8)
        fn_h(src.ctypes.data, n, *extra, a.ctypes.data)
        fn_o(src.ctypes.data, n, *extra, b.ctypes.data)
        assert np.array_equal(a, b)
    # spot checks of the reference formulas (images.hpp:27-29,73-76)
    c = np.array([[0.5, 1.0, -1.0]], np.float32)
    out = np.zeros((1, 3), np.uint8)
    H.ctr_quantise_color(c.ctypes.data, 1, out.ctypes.data)
    assert out.tolist() == [[127, 255, 0]]


def test_quantisers_known_answers_from_the_reference_text(ca):
    """Known answers worked out BY HAND from the reference's formulas (not from any restatement of ours):
      depth   inc/images.hpp:27-29   isfinite(v) ? (byte)(255 * (m - v) / m) : 0
      normal  inc/images.hpp:48-54   norm <= 1e-6 ? (0,0,0) : (byte)(255 * (0.5 + 0.5 * normalized))
      colour  inc/images.hpp:73-76   (byte)(255 * min(1, max(0, c)))        (byte casts truncate)
    and the same vectors through the oracle's restatement, which pins it to the reference text as well."""
    H, O = _lib.host_lib(), oracle.oracle_lib()
    m = np.float32(8.0)
    depth = np.array([8.0, 0.0, 4.0, 2.0, 6.0, np.inf, -np.inf, np.nan, 7.0], np.float32)
    #  v = m -> 255*0/8 = 0;  v = 0 -> 255;  4 -> 127.5 -> 127;  2 -> 191.25 -> 191;  6 -> 63.75 -> 63;
    #  non-finite -> 0;  7 -> 31.875 -> 31
    want_d = [0, 255, 127, 191, 63, 0, 0, 0, 31]
    normal = np.array([[0, 0, 1], [0, 0, -1], [0, 2, 0], [-1, 0, 0], [1e-7, 0, 0], [0, 0, 0], [1e-5, 0, 0], [0, -0.25, 0]],
                      np.float32)
    #  (0,0,1) -> 0.5+0.5*(0,0,1) = (.5,.5,1) -> (127,127,255);  (0,0,-1) -> (127,127,0);  (0,2,0) -> (127,255,127);
    #  (-1,0,0) -> (0,127,127);  |n| = 1e-7 <= 1e-6 -> 0;  zero -> 0;  |n| = 1e-5 > 1e-6 -> (1,0,0) -> (255,127,127);
    #  (0,-.25,0) -> (0,-1,0) -> (127,0,127)
    want_n = [[127, 127, 255], [127, 127, 0], [127, 255, 127], [0, 127, 127], [0, 0, 0], [0, 0, 0], [255, 127, 127],
              [127, 0, 127]]
    color = np.array([[0.999, 1.3, -0.2], [0.5, 1.0, 0.0], [0.25, 0.75, 2.0], [np.nan, 0.004, 0.00392]], np.float32)
    #  .999 -> 254.745 -> 254;  1.3 -> 1 -> 255;  -.2 -> 0;  .5 -> 127.5 -> 127;  .25 -> 63.75 -> 63;  .75 -> 191.25 -> 191;
    #  NaN: max(0, NaN) = 0 (a < b is false) -> 0;  .004 -> 1.02 -> 1;  .00392 -> 0.9996 -> 0
    want_c = [[254, 255, 0], [127, 255, 0], [63, 191, 255], [0, 1, 0]]
    for L, pre in ((H, "ctr"), (O, "orc")):
        out = np.zeros((len(depth), 3), np.uint8)
        getattr(L, pre + "_quantise_depth")(depth.ctypes.data, len(depth), C.c_float(float(m)), out.ctypes.data)
        assert out[:, 0].tolist() == want_d and np.array_equal(out[:, 0], out[:, 1]) and np.array_equal(out[:, 0], out[:, 2]), pre
        out = np.zeros((len(normal), 3), np.uint8)
        getattr(L, pre + "_quantise_normal")(normal.ctypes.data, len(normal), out.ctypes.data)
        assert out.tolist() == want_n, pre
        out = np.zeros((len(color), 3), np.uint8)
        getattr(L, pre + "_quantise_color")(color.ctypes.data, len(color), out.ctypes.data)
        assert out.tolist() == want_c, pre


def test_jpeg_writer_produces_a_decodable_image(ca, tmp_path):
    from PIL import Image
    w, h = 67, 45  # not multiples of 8
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 255 // (w + h))], -1).astype(np.uint8)
    path = str(tmp_path / "t.jpg")
    assert _lib.host_lib().ctr_write_jpg(path.encode(), w, h, np.ascontiguousarray(rgb).ctypes.data, 90) == 0
    im = np.asarray(Image.open(path).convert("RGB")).astype(np.int32)
    assert im.shape == (h, w, 3)
    assert np.abs(im - rgb.astype(np.int32)).mean() < 3.0


def test_dump_scene_text_is_the_reference_format(ca, capfd):
    """gpu::dump_scene (inc/kernel.hpp:150-166): ' -> Have %-4llu objects:' / '  -> Object   #%-4llu has type #%-2llu'
    with the reference's variant indices (triangle 0, mesh 1, plane 2, sphere 3; sun 0, point 1; solid 0)."""
    from cutrace_amd import _lib
    from tests.conftest import load_scene
    s = load_scene(ca, "sphere_plane")
    capfd.readouterr()
    _lib.host_lib().ctr_dump_scene(s.desc)
    out = capfd.readouterr().out.splitlines()
    d = s.desc.contents
    assert out[0] == " -> Have %-4d objects:" % d.n_objects
    want_types = {"sphere": 3, "plane": 2}
    import json
    js = json.load(open("scene/sphere_plane.json"))
    for i, o in enumerate(js["objects"]):
        assert out[1 + i] == "  -> Object   #%-4d has type #%-2d" % (i, want_types[o["type"]])
    k = 1 + len(js["objects"])
    assert out[k] == " -> Have %-4d lights:" % len(js["lights"])
    for i, l in enumerate(js["lights"]):
        assert out[k + 1 + i] == "  -> Light    #%-4d has type #%-2d" % (i, {"sun": 0, "point": 1}[l["type"]])
    k += 1 + len(js["lights"])
    assert out[k] == " -> Have %-4d materials:" % len(js["materials"])
    for i in range(len(js["materials"])):
        assert out[k + 1 + i] == "  -> Material #%-4d has type #%-2d" % (i, 0)
    assert len(out) == k + 1 + len(js["materials"])
