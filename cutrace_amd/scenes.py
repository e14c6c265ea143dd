"""Generators for the derived benchmark/parity scenes.

The reference ships five scene JSONs and four binary STLs (copied as data under scene/).  Three
configs of BASELINE.json need inputs the reference does not ship:

  C2-dense  bunny.json with every mesh triangle split by `rounds` rounds of midpoint
            subdivision (1000 * 4^3 = 64 000 triangles, same surface) — BASELINE.json says
            "~70k tris" while scene/bunny.stl holds 1000.
  C3-deep   mirror.json with the wall material (index 1) made reflective (0.5) so that paths
            really reach the requested recursion depth.
  C4        16 copies of bunny.stl translated on a 4x4 grid (the reference has no instancing:
            a mesh takes only `file` and `material`, inc/default_schema.hpp:603-606), the five
            walls moved out to enclose them, camera 4096 x 4096.

Everything is written under a caller-supplied directory as plain scene JSON + binary STL, i.e.
exactly what `cutrace <scene.json>` (reference or this build) loads.  Vertex arithmetic is
float32 throughout so that every consumer sees identical triangles.
"""
import json
import os
import struct

import numpy as np

from . import _lib

ROOT = _lib.ROOT


def read_stl(path):
    raw = open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 80)[0]
    assert len(raw) == 84 + 50 * n, "not a binary STL"
    rec = np.frombuffer(raw, dtype=np.uint8, count=50 * n, offset=84).reshape(n, 50)
    return rec[:, 12:48].copy().view(np.float32).reshape(n, 3, 3)  # (tri, vertex, xyz)


def write_stl(path, tris):
    tris = np.ascontiguousarray(tris, dtype=np.float32)
    n = tris.shape[0]
    rec = np.zeros((n, 50), np.uint8)
    rec[:, 12:48] = tris.reshape(n, 9).view(np.uint8).reshape(n, 36)
    with open(path, "wb") as f:
        f.write(b"cutrace_amd generated mesh".ljust(80, b"\0"))
        f.write(struct.pack("<I", n))
        f.write(rec.tobytes())


def subdivide(tris, rounds):
    """Each round: triangle (a,b,c) -> (a,ab,ca) (ab,b,bc) (ca,bc,c) (ab,bc,ca); midpoints (p+q)*0.5f."""
    t = np.asarray(tris, np.float32)
    half = np.float32(0.5)
    for _ in range(rounds):
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab, bc, ca = (a + b) * half, (b + c) * half, (c + a) * half
        t = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                      np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
    return t.astype(np.float32)


def _load_json(name):
    return json.load(open(os.path.join(ROOT, "scene", name)))


def make_dense_bunny(out_dir, rounds=3, width=None, height=None):
    os.makedirs(out_dir, exist_ok=True)
    sc = _load_json("bunny.json")
    stl = os.path.join(out_dir, f"bunny_sub{rounds}.stl")
    write_stl(stl, subdivide(read_stl(os.path.join(ROOT, "scene", "bunny.stl")), rounds))
    for o in sc["objects"]:
        if o["type"] == "mesh":
            o["file"] = stl
    if width:
        sc["camera"]["width"], sc["camera"]["height"] = width, height
    path = os.path.join(out_dir, f"bunny_dense{rounds}.json")
    json.dump(sc, open(path, "w"), indent=1)
    return path


def make_mirror_deep(out_dir, reflect=0.5, width=None, height=None):
    os.makedirs(out_dir, exist_ok=True)
    sc = _load_json("mirror.json")
    sc["materials"][1]["reflect"] = reflect
    for o in sc["objects"]:
        if o["type"] == "mesh":
            o["file"] = os.path.join(ROOT, o["file"])
    if width:
        sc["camera"]["width"], sc["camera"]["height"] = width, height
    path = os.path.join(out_dir, "mirror_deep.json")
    json.dump(sc, open(path, "w"), indent=1)
    return path


def make_bunny_grid(out_dir, n=4, spacing=1.6, width=4096, height=4096):
    """C4: n x n translated bunnies in an enlarged box (walls moved out to enclose them),
    camera and lights moved back proportionally."""
    os.makedirs(out_dir, exist_ok=True)
    sc = _load_json("bunny.json")
    base = read_stl(os.path.join(ROOT, "scene", "bunny.stl"))
    mesh = next(o for o in sc["objects"] if o["type"] == "mesh")
    objs = []
    span = np.float32(spacing)
    for i in range(n):
        for j in range(n):
            off = np.array([(i - (n - 1) / 2) * span, 0.0, (j - (n - 1) / 2) * span], np.float32)
            stl = os.path.join(out_dir, f"bunny_{i}_{j}.stl")
            write_stl(stl, (base + off).astype(np.float32))
            objs.append({"type": "mesh", "material": mesh["material"], "file": stl})
    half = float((n - 1) / 2 * spacing + 1.0)  # wall distance in x and z
    k = half  # the original box has half-width 1
    for o in sc["objects"]:
        if o["type"] != "plane":
            continue
        p = o["point"]
        o["point"] = [p[0] * k, p[1], p[2] * k]  # x/z walls move out, floor/ceiling stay at y = -1/+1
        objs.append(o)
    sc["objects"] = objs
    cam = sc["camera"]
    cam["eye"] = [cam["eye"][0] * k, 0.6, cam["eye"][2] * k]
    cam["look"] = [cam["look"][0] * k, -0.2, cam["look"][2] * k]
    cam["width"], cam["height"] = width, height
    for l in sc["lights"]:
        l["point"] = [l["point"][0] * k, l["point"][1], l["point"][2] * k]
    path = os.path.join(out_dir, f"bunny_grid{n}x{n}.json")
    json.dump(sc, open(path, "w"), indent=1)
    return path
