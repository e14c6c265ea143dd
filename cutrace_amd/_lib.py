"""ctypes view of the C-ABI (include/cutrace_amd.h, include/cutrace_host.h).

Python is plumbing here (tests, bench.py); the product is the C-ABI library and the
C++ host code.  There is NO Python/CPU fallback for rendering: if libcutrace_amd.so
is missing or no HIP device is present, rendering raises.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Triangle(C.Structure):
    _fields_ = [("p1", Vec3), ("p2", Vec3), ("p3", Vec3)]


class Object(C.Structure):
    _fields_ = [("type", C.c_uint32), ("reserved", C.c_uint32), ("mat_idx", C.c_uint64),
                ("v0", Vec3), ("v1", Vec3), ("v2", Vec3), ("f0", C.c_float),
                ("tri_begin", C.c_uint64), ("tri_count", C.c_uint64)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("v", Vec3), ("color", Vec3)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", Vec3), ("specular", C.c_float), ("reflexivity", C.c_float),
                ("phong_exp", C.c_float), ("transparency", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", Vec3), ("up", Vec3), ("forward", Vec3), ("right", Vec3),
                ("near_plane", C.c_float), ("far_plane", C.c_float), ("ambient", C.c_float),
                ("reserved", C.c_uint32), ("w", C.c_uint64), ("h", C.c_uint64)]


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.POINTER(Object)), ("n_objects", C.c_uint64),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_uint64),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint64),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint64),
                ("cam", Camera)]


class Rows(C.Structure):
    _fields_ = [("row_begin", C.c_uint64), ("row_end", C.c_uint64), ("block_rows", C.c_uint64),
                ("part", C.c_uint32), ("n_parts", C.c_uint32)]


class ReintPart(C.Structure):
    _fields_ = [("d_depth", C.c_void_p), ("d_color3", C.c_void_p), ("d_normal3", C.c_void_p)]


class RenderStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("ray_count", C.c_uint64),
                ("rows", C.c_uint64), ("max_depth", C.c_float), ("reserved", C.c_uint32)]


HOST_SYMBOLS = [
    "ctr_host_scene_load", "ctr_host_scene_parse", "ctr_host_scene_free", "ctr_host_scene_desc",
    "ctr_host_scene_set_size", "ctr_host_scene_set_material", "ctr_stl_read", "ctr_stl_write",
    "ctr_dump_scene", "ctr_dump_schema", "ctr_quantise_depth", "ctr_quantise_normal", "ctr_quantise_color",
    "ctr_write_jpg", "ctr_write_depth_map", "ctr_write_normal_map", "ctr_write_colorized",
    "ctr_camera_look_at", "ctr_mesh_bounds", "ctr_rows_count",
]
HIP_SYMBOLS = [
    "ctr_abi_version", "ctr_last_error", "ctr_device_count", "ctr_scene_create", "ctr_scene_destroy",
    "ctr_scene_size", "ctr_scene_set_size", "ctr_render", "ctr_render_device", "ctr_set_variant",
    "ctr_scene_set_cameras", "ctr_render_device_batch",
    "ctr_algorithmic_bytes", "ctr_frame_alloc", "ctr_frame_free", "ctr_tile_costs", "ctr_last_counters", "ctr_selftest_exact_math",
    "ctr_debug_poison_next_order", "ctr_render_uv", "ctr_debug_lane_stats",
    "ctr_multi_create", "ctr_multi_destroy", "ctr_multi_devices", "ctr_multi_transport", "ctr_multi_size", "ctr_multi_set_size",
    "ctr_multi_set_variant", "ctr_render_multi", "ctr_multi_kernel_ms", "ctr_reinterleave_device",
    "ctr_multi_submit", "ctr_multi_wait",
]

_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        path = os.environ.get("CUTRACE_HOST_LIB") or os.path.join(PKG, "libcutrace_host.so")  # override: sanitizer builds (scripts/cpu_sanitize.sh)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -m cutrace_amd.build` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.ctr_host_scene_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ctr_host_scene_parse.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ctr_host_scene_free.argtypes = [C.c_void_p]
        L.ctr_host_scene_desc.argtypes = [C.c_void_p]
        L.ctr_host_scene_desc.restype = C.POINTER(SceneDesc)
        L.ctr_host_scene_set_size.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ctr_host_scene_set_material.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Material)]
        L.ctr_stl_read.argtypes = [C.c_char_p, C.POINTER(Triangle), C.c_uint64]
        L.ctr_stl_read.restype = C.c_int64
        L.ctr_stl_write.argtypes = [C.c_char_p, C.POINTER(Triangle), C.c_uint64]
        L.ctr_dump_scene.argtypes = [C.POINTER(SceneDesc)]
        L.ctr_camera_look_at.argtypes = [C.POINTER(Camera), Vec3, Vec3, Vec3]
        L.ctr_mesh_bounds.argtypes = [C.POINTER(Triangle), C.c_uint64, C.POINTER(Vec3), C.POINTER(Vec3)]
        L.ctr_rows_count.argtypes = [C.POINTER(Rows), C.c_uint64]
        L.ctr_rows_count.restype = C.c_uint64
        for fn in ("ctr_quantise_normal", "ctr_quantise_color"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ctr_quantise_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        L.ctr_write_jpg.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ctr_write_depth_map.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_float]
        L.ctr_write_normal_map.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.ctr_write_colorized.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
        _host = L
    return _host


def hip_lib():
    """The HIP C-ABI library.  Fails loudly when it is missing: there is no fallback."""
    global _hip
    if _hip is None:
        path = os.environ.get("CUTRACE_AMD_LIB") or os.path.join(PKG, "libcutrace_amd.so")  # override: tuning builds
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is not built; refusing to fall back to CPU")
        # the CLI links both libs; load host first so shared symbols resolve identically
        host_lib()
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.ctr_last_error.restype = C.c_char_p
        L.ctr_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.ctr_scene_destroy.argtypes = [C.c_void_p]
        L.ctr_scene_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ctr_scene_set_size.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ctr_render.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(Rows), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(RenderStats)]
        L.ctr_render_device.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(Rows), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.ctr_set_variant.argtypes = [C.c_void_p, C.c_uint32]
        L.ctr_scene_set_cameras.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32]
        L.ctr_render_device_batch.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(Rows), C.c_uint32, C.c_uint32,
                                              C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ctr_algorithmic_bytes.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(Rows), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]
        # entry points added after round 1 (an older build may be loaded through CUTRACE_AMD_LIB for A/B timing)
        opt = {
            "ctr_frame_alloc": ([C.c_uint64, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                 C.POINTER(C.POINTER(C.c_float))], C.c_int),
            "ctr_frame_free": ([C.POINTER(C.c_float)], None),
            "ctr_multi_create": ([C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)], C.c_int),
            "ctr_multi_destroy": ([C.c_void_p], None),
            "ctr_multi_devices": ([C.c_void_p], C.c_int),
            "ctr_multi_transport": ([C.c_void_p], C.c_char_p),
            "ctr_multi_size": ([C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)], C.c_int),
            "ctr_multi_set_size": ([C.c_void_p, C.c_uint64, C.c_uint64], C.c_int),
            "ctr_multi_set_variant": ([C.c_void_p, C.c_uint32], C.c_int),
            "ctr_render_multi": ([C.c_void_p, C.c_float, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(RenderStats)], C.c_int),
            "ctr_multi_submit": ([C.c_void_p, C.c_float, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
            "ctr_multi_wait": ([C.c_void_p, C.POINTER(RenderStats)], C.c_int),
            "ctr_reinterleave_device": ([C.POINTER(ReintPart), C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
            "ctr_multi_kernel_ms": ([C.c_void_p, C.POINTER(C.c_double), C.c_int], C.c_int),
            "ctr_selftest_exact_math": ([C.POINTER(C.c_uint64)], C.c_int),
            "ctr_last_counters": ([C.c_void_p, C.c_void_p], C.c_int),
            "ctr_debug_poison_next_order": ([C.c_void_p], C.c_int),
            "ctr_debug_lane_stats": ([C.c_void_p, C.c_int], C.c_int),
            "ctr_render_uv": ([C.c_void_p, C.c_float, C.c_int, C.POINTER(Rows), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(RenderStats)], C.c_int),
            "ctr_tile_costs": ([C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)], C.c_int),
        }
        for name, (argt, rest) in opt.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.argtypes = argt
                fn.restype = rest
        _hip = L
    return _hip
