"""Shared comparison helpers for the parity tests."""
import numpy as np

TOL = 1e-4  # per-channel float tolerance stated by BASELINE.json:north_star


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def depth_report(got, want):
    """depth: +inf must match +inf; finite compared. Returns (n_bad_bitexact, n_bad_tol, max_abs)."""
    fin_g, fin_w = np.isfinite(got), np.isfinite(want)
    mism_inf = int((fin_g != fin_w).sum())
    both = fin_g & fin_w
    diff = np.zeros(got.shape, np.float64)
    diff[both] = np.abs(got[both].astype(np.float64) - want[both].astype(np.float64))
    nbits = int((bits(got) != bits(want)).sum())
    return dict(inf_mismatch=mism_inf, not_bitexact=nbits, over_tol=int((diff > TOL).sum()) + mism_inf,
                max_abs=float(diff.max()) if diff.size else 0.0)


def float_report(got, want, tol=TOL):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    d = np.where(np.isnan(d), np.inf, d)
    px_bad = (d.reshape(d.shape[0], d.shape[1], -1) > tol).any(-1) if d.ndim == 3 else (d > tol)
    return dict(not_bitexact=int((bits(got) != bits(want)).sum()), over_tol=int(px_bad.sum()),
                max_abs=float(d.max()) if d.size else 0.0)


def assert_parity(got, want, what="", color_tol=TOL):
    """The bar: depth and normal bit-exact (pure +,-,*,/,sqrt arithmetic), colour within 1e-4
    per channel (pow() differs by ≤1 ulp between glibc and the device)."""
    dr = depth_report(got["depth"], want["depth"])
    nr = float_report(got["normal"], want["normal"])
    cr = float_report(got["color"], want["color"], color_tol)
    msg = f"{what}: depth {dr} normal {nr} color {cr}"
    assert dr["not_bitexact"] == 0, msg
    assert nr["not_bitexact"] == 0, msg
    assert cr["over_tol"] == 0, msg
    return dict(depth=dr, normal=nr, color=cr)


def mesh_scene(stl_path, w, h, tris, extra_objects=(), fudge_note=""):
    import json
    from cutrace_amd import scenes
    scenes.write_stl(stl_path, np.asarray(tris, np.float32).reshape(-1, 3, 3))
    mats = [{"type": "solid", "color": [0.8, 0.6, 0.3], "specular": 0.4, "reflect": 0.3, "phong": 40},
            {"type": "solid", "color": [0.3, 0.5, 0.9], "specular": 0.2, "reflect": 0.0, "phong": 10}]
    objs = [{"type": "mesh", "file": stl_path, "material": 0},
            {"type": "plane", "point": [0, -1.0, 0], "normal": [0, 1, 0], "material": 1}] + list(extra_objects)
    lights = [{"type": "point", "point": [1.5, 2.5, 2.0], "color": [0.8, 0.8, 0.8]},
              {"type": "sun", "direction": [-0.3, -1.0, -0.2], "color": [0.4, 0.4, 0.4]}]
    cam = {"eye": [0.3, 0.8, 4.0], "up": [0, 1, 0], "look": [-0.05, -0.15, -1.0], "near_plane": 0.1, "far_plane": 100.0,
           "width": w, "height": h, "ambient": 0.1}
    return json.dumps({"camera": cam, "lights": lights, "materials": mats, "objects": objs})


def corner_meshes():
    """Triangle lists for the mesh corner-case tests: quad, duplicates, degenerate, fan, far."""
    quad = [[[-1, -1, 0], [1, -1, 0], [1, 1, 0]], [[-1, -1, 0], [1, 1, 0], [-1, 1, 0]]]
    dup = quad + quad + [[[-1, -1, 0.5], [1, -1, 0.5], [0, 1, 0.5]]] * 3          # duplicates and triplicates
    degenerate = quad + [[[0, 0, 1], [0, 0, 1], [0, 0, 1]], [[0, 0, 1], [1, 1, 1], [2, 2, 1]]]  # point, collinear
    fan = [[[0, 0, 0.3], [float(np.cos(a)), float(np.sin(a)), 0.0], [float(np.cos(a + 0.7)), float(np.sin(a + 0.7)), 0.0]]
           for a in np.arange(0, 6.28, 0.7)]
    far = (np.asarray(quad, np.float32) * 500.0 + np.float32([3000, 0, -9000])).tolist()
    return quad, dup, degenerate, fan, far
