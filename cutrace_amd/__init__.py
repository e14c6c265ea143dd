"""cutrace_amd — MI355X-native ray-cast + shading path of jay-tux/cutrace.

The product is native: `libcutrace_amd.so` (HIP kernels behind the C-ABI in
include/cutrace_amd.h), `libcutrace_host.so` (scene loader / image writers) and the
`cutrace` CLI.  This package is the thin ctypes plumbing tests and bench.py use.
"""
import contextlib
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Camera, Light, Material, Object, RenderStats, Rows, SceneDesc, Triangle, Vec3  # noqa: F401

ROOT = _lib.ROOT

VAR_AUTO = 0
VAR_NO_PREFILTER = 2
VAR_NO_ANYHIT = 4
VAR_NO_CLUSTER = 8
VAR_STATS = 16
VAR_EXACT_POW = 32
VAR_NO_REORDER = 256
VAR_NO_OCC6 = 512
VAR_NO_DIRECT = 1024
VAR_IMAGE_ORDER_FIRST = 2048
VAR_MERGE = 4096
VAR_IGNORE_TRANSPARENT = 8192


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


class HostScene:
    """A scene loaded by the C++ loader (libcutrace_host.so): owns the flat arrays."""

    def __init__(self, handle, status):
        self._h = handle
        self.status = status

    @classmethod
    def load(cls, json_path, cwd=None):
        """Load a scene JSON. Mesh paths inside it are relative to `cwd` (default: repo root,
        like the reference which expects to be run from its repository root)."""
        L = _lib.host_lib()
        h = C.c_void_p()
        with _cwd(cwd or ROOT):
            st = L.ctr_host_scene_load(os.fsencode(json_path), C.byref(h))
        if not h:
            raise IOError(f"cannot read scene file {json_path}")
        return cls(h, st)

    @classmethod
    def parse(cls, text, cwd=None):
        L = _lib.host_lib()
        h = C.c_void_p()
        with _cwd(cwd or ROOT):
            st = L.ctr_host_scene_parse(text.encode(), C.byref(h))
        return cls(h, st)

    @property
    def ok(self):
        return self.status == 0

    @property
    def desc(self):
        return _lib.host_lib().ctr_host_scene_desc(self._h)

    def set_size(self, w, h):
        _lib.host_lib().ctr_host_scene_set_size(self._h, w, h)

    def set_material(self, idx, **kw):
        d = self.desc.contents
        m = Material()
        C.memmove(C.byref(m), C.byref(d.materials[idx]), C.sizeof(Material))
        for k, v in kw.items():
            if k == "color":
                m.color = Vec3(*v)
            else:
                setattr(m, k, v)
        st = _lib.host_lib().ctr_host_scene_set_material(self._h, idx, C.byref(m))
        if st:
            raise ValueError("bad material index")

    @property
    def size(self):
        cam = self.desc.contents.cam
        return int(cam.w), int(cam.h)

    def close(self):
        if self._h:
            _lib.host_lib().ctr_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_rows(h, rows=None):
    """rows: None (whole frame) | (row_begin,row_end) | (row_begin,row_end,block_rows,part,n_parts)"""
    if rows is None:
        return Rows(0, h, max(h, 1), 0, 1)
    if len(rows) == 2:
        return Rows(rows[0], rows[1], max(h, 1), 0, 1)
    return Rows(*rows)


def rows_count(h, rows):
    r = make_rows(h, rows)
    return int(_lib.host_lib().ctr_rows_count(C.byref(r), h))


class DeviceScene:
    """A scene uploaded to one GPU through the C-ABI (ctr_scene_create)."""

    def __init__(self, host_scene, device=0):
        L = _lib.hip_lib()
        h = C.c_void_p()
        st = L.ctr_scene_create(host_scene.desc, device, C.byref(h))
        if st:
            raise RuntimeError(f"ctr_scene_create failed ({st}): {L.ctr_last_error().decode()}")
        self._h = h
        self.device = device
        self.w, self.h = host_scene.size

    def set_variant(self, bits):
        st = _lib.hip_lib().ctr_set_variant(self._h, bits)
        if st:
            raise RuntimeError(f"ctr_set_variant failed: {_lib.hip_lib().ctr_last_error().decode()}")

    def set_size(self, w, h):
        st = _lib.hip_lib().ctr_scene_set_size(self._h, w, h)
        if st:
            raise RuntimeError("ctr_scene_set_size failed")
        self.w, self.h = w, h

    def _pinned_frame(self, px):
        """One page-locked block for a frame's three buffers (ctr_frame_alloc), kept with the scene handle."""
        if getattr(self, "_pin_px", 0) < px:
            self._free_pinned()
            d, c, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
            st = _lib.hip_lib().ctr_frame_alloc(px, C.byref(d), C.byref(c), C.byref(n))
            if st:
                raise RuntimeError("ctr_frame_alloc failed")
            self._pin, self._pin_px = d, px
        return np.ctypeslib.as_array(self._pin, shape=(7 * self._pin_px,))

    def _free_pinned(self):
        if getattr(self, "_pin_px", 0):
            _lib.hip_lib().ctr_frame_free(self._pin)
            self._pin_px = 0

    def render(self, fudge=1e-3, bounces=5, rows=None, pinned=False, into=None):
        """Host-buffer form (ctr_render): returns numpy buffers + stats.  pinned=True: the buffers are views of
        the scene handle's page-locked frame block (valid until the next pinned render / close).  into: a dict
        returned by an earlier call of the same shape, whose buffers are written again."""
        L = _lib.hip_lib()
        r = make_rows(self.h, rows)
        n = rows_count(self.h, rows)
        if into is not None:
            depth, color, normal = into["depth"], into["color"], into["normal"]
            assert depth.shape == (n, self.w) and color.shape == (n, self.w, 3) and normal.shape == (n, self.w, 3)
        elif pinned:
            px = max(n * self.w, 1)
            blk = self._pinned_frame(px)
            depth = blk[:px].reshape(n, self.w) if n else np.empty((0, self.w), np.float32)
            color = blk[px:4 * px].reshape(n, self.w, 3) if n else np.empty((0, self.w, 3), np.float32)
            normal = blk[4 * px:7 * px].reshape(n, self.w, 3) if n else np.empty((0, self.w, 3), np.float32)
        else:
            depth = np.empty((n, self.w), np.float32)
            color = np.empty((n, self.w, 3), np.float32)
            normal = np.empty((n, self.w, 3), np.float32)
        stats = RenderStats()
        st = L.ctr_render(self._h, C.c_float(fudge), bounces, C.byref(r), depth.ctypes.data, color.ctypes.data,
                          normal.ctypes.data, C.byref(stats))
        if st:
            raise RuntimeError(f"ctr_render failed ({st}): {L.ctr_last_error().decode()}")
        return dict(depth=depth, color=color, normal=normal, ray_count=int(stats.ray_count),
                    kernel_ms=stats.kernel_ms, total_ms=stats.total_ms, max_depth=float(stats.max_depth),
                    rows=int(stats.rows))

    def render_uv(self, fudge=1e-3, bounces=5, rows=None):
        """ctr_render_uv: the three buffers plus `uv` (n, w, 2): ray_cast's texture coordinates of the primary hit."""
        L = _lib.hip_lib()
        r = make_rows(self.h, rows)
        n = rows_count(self.h, rows)
        depth = np.empty((n, self.w), np.float32)
        color = np.empty((n, self.w, 3), np.float32)
        normal = np.empty((n, self.w, 3), np.float32)
        uv = np.empty((n, self.w, 2), np.float32)
        stats = RenderStats()
        st = L.ctr_render_uv(self._h, C.c_float(fudge), bounces, C.byref(r), depth.ctypes.data, color.ctypes.data,
                             normal.ctypes.data, uv.ctypes.data, C.byref(stats))
        if st:
            raise RuntimeError(f"ctr_render_uv failed ({st}): {L.ctr_last_error().decode()}")
        return dict(depth=depth, color=color, normal=normal, uv=uv, ray_count=int(stats.ray_count),
                    kernel_ms=stats.kernel_ms, total_ms=stats.total_ms, max_depth=float(stats.max_depth), rows=int(stats.rows))

    def render_device(self, d_depth, d_color, d_normal, d_counters=0, stream=0, fudge=1e-3, bounces=5, rows=None):
        """Device-buffer form (ctr_render_device): raw device pointers, async on `stream`."""
        L = _lib.hip_lib()
        r = make_rows(self.h, rows)
        st = L.ctr_render_device(self._h, C.c_float(fudge), bounces, C.byref(r), d_depth, d_color, d_normal,
                                 d_counters, stream)
        if st:
            raise RuntimeError(f"ctr_render_device failed ({st}): {L.ctr_last_error().decode()}")

    def set_cameras(self, cams):
        """Upload a camera path (list of _lib.Camera, same w/h): frames of ctr_render_device_batch."""
        arr = (Camera * len(cams))(*cams)
        st = _lib.hip_lib().ctr_scene_set_cameras(self._h, arr, len(cams))
        if st:
            raise RuntimeError(f"ctr_scene_set_cameras failed: {_lib.hip_lib().ctr_last_error().decode()}")
        self.n_cams = len(cams)

    def render_device_batch(self, d_depth, d_color, d_normal, n_frames, frame_stride_px, first_frame=0, d_counters=0,
                            stream=0, fudge=1e-3, bounces=5, rows=None, part_stride=0):
        """n_frames frames in ONE launch (ctr_render_device_batch)."""
        L = _lib.hip_lib()
        r = make_rows(self.h, rows)
        st = L.ctr_render_device_batch(self._h, C.c_float(fudge), bounces, C.byref(r), first_frame, n_frames,
                                       frame_stride_px, part_stride, d_depth, d_color, d_normal, d_counters, stream)
        if st:
            raise RuntimeError(f"ctr_render_device_batch failed ({st}): {L.ctr_last_error().decode()}")

    def algorithmic_bytes(self, fudge=1e-3, bounces=5, rows=None):
        L = _lib.hip_lib()
        r = make_rows(self.h, rows)
        b = C.c_uint64()
        n = C.c_uint64()
        st = L.ctr_algorithmic_bytes(self._h, C.c_float(fudge), bounces, C.byref(r), C.byref(b), C.byref(n))
        if st:
            raise RuntimeError(f"ctr_algorithmic_bytes failed ({st}): {L.ctr_last_error().decode()}")
        return int(b.value), int(n.value)

    def last_counters(self):
        """The 16 counter words of the last host-form render (ctr_last_counters)."""
        out = np.zeros(16, np.uint64)
        _lib.hip_lib().ctr_last_counters(self._h, out.ctypes.data)
        return out

    @staticmethod
    def lane_stats(reset=True):
        """Live-lane statistics of the VAR_STATS launches since the last reset (ctr_debug_lane_stats), decoded."""
        raw = np.zeros(96, np.uint64)
        if _lib.hip_lib().ctr_debug_lane_stats(raw.ctypes.data, 1 if reset else 0):
            raise RuntimeError("ctr_debug_lane_stats failed")
        c = [int(x) for x in raw]
        trips, live = c[72], c[73]
        out = {"wave_trips": trips, "live_lanes_per_trip": live / trips if trips else None,
               "live_fraction": live / (64.0 * trips) if trips else None,
               "live_fraction_of_in_image_lanes": (live / trips) / (c[75] / c[74]) if trips and c[74] and c[75] else None,
               "trips_by_live_lanes_1_8_to_57_64": c[64:72], "trips_mixing_kinds": c[76], "waves": c[74],
               "merged_walks": c[78], "merged_walks_redone": c[77],
               "shadow_casts_at_meshes": {"wave_casts": c[80], "receivers_all_off_mesh": c[81], "of_those_unoccluded_by_meshes": c[82],
                                          "unoccluded_by_meshes_any_receiver": c[83]}}
        kinds = {}
        for d in range(16):
            for name, base in (("radiance", 0), ("shadow", 16)):
                if c[32 + base + d]:
                    label = "primary" if (name == "radiance" and d == 0) else f"{name}@depth{d}"
                    kinds[label] = {"lanes": c[base + d], "trips": c[32 + base + d],
                                    "lanes_per_trip": c[base + d] / c[32 + base + d]}
        out["by_kind"] = kinds
        return out

    def tile_costs(self):
        """Per-tile cost of the last launch (ctr_tile_costs), as a uint32 array."""
        L = _lib.hip_lib()
        n = C.c_uint64()
        L.ctr_tile_costs(self._h, None, 0, C.byref(n))
        out = np.zeros(int(n.value), np.uint32)
        if n.value:
            L.ctr_tile_costs(self._h, out.ctypes.data, n.value, C.byref(n))
        return out

    def close(self):
        if self._h:
            self._free_pinned()
            _lib.hip_lib().ctr_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """A scene replicated on several GPUs of this process (ctr_multi_create): one frame per call, row-tiled over the
    devices, gathered to devices[0] (RCCL) and re-interleaved there (ctr_render_multi)."""

    def __init__(self, host_scene, devices):
        L = _lib.hip_lib()
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        st = L.ctr_multi_create(host_scene.desc, devs, len(devices), C.byref(h))
        if st:
            raise RuntimeError(f"ctr_multi_create failed ({st}): {L.ctr_last_error().decode()}")
        self._h = h
        self.devices = list(devices)
        self.w, self.h = host_scene.size

    @property
    def transport(self):
        return _lib.hip_lib().ctr_multi_transport(self._h).decode()

    def set_variant(self, bits):
        if _lib.hip_lib().ctr_multi_set_variant(self._h, bits):
            raise RuntimeError(f"ctr_multi_set_variant failed: {_lib.hip_lib().ctr_last_error().decode()}")

    def set_size(self, w, h):
        if _lib.hip_lib().ctr_multi_set_size(self._h, w, h):
            raise RuntimeError("ctr_multi_set_size failed")
        self.w, self.h = w, h

    def render(self, fudge=1e-3, bounces=5, block_rows=8, pinned=False):
        """pinned=True: the buffers are views of a page-locked frame block of this group (valid until the next pinned
        render / close); device 0 then writes the frame into it directly."""
        L = _lib.hip_lib()
        if pinned:
            px = self.w * self.h
            if getattr(self, "_pin_px", 0) != px:
                self._free_pinned()
                d, c, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
                if L.ctr_frame_alloc(px, C.byref(d), C.byref(c), C.byref(n)):
                    raise RuntimeError("ctr_frame_alloc failed")
                self._pin_ptr, self._pin_px = d, px
                self._pin = np.ctypeslib.as_array(d, shape=(7 * px,))
            blk = self._pin
            depth = blk[:px].reshape(self.h, self.w)
            color = blk[px:4 * px].reshape(self.h, self.w, 3)
            normal = blk[4 * px:7 * px].reshape(self.h, self.w, 3)
        else:
            depth = np.empty((self.h, self.w), np.float32)
            color = np.empty((self.h, self.w, 3), np.float32)
            normal = np.empty((self.h, self.w, 3), np.float32)
        stats = RenderStats()
        st = L.ctr_render_multi(self._h, C.c_float(fudge), bounces, block_rows, depth.ctypes.data, color.ctypes.data,
                                normal.ctypes.data, C.byref(stats))
        if st:
            raise RuntimeError(f"ctr_render_multi failed ({st})")
        ms = (C.c_double * len(self.devices))()
        L.ctr_multi_kernel_ms(self._h, ms, len(self.devices))
        return dict(depth=depth, color=color, normal=normal, ray_count=int(stats.ray_count), kernel_ms=stats.kernel_ms,
                    total_ms=stats.total_ms, max_depth=float(stats.max_depth), rows=int(stats.rows),
                    kernel_ms_per_device=list(ms))

    def alloc_frame(self):
        """A page-locked frame block of the group's size (ctr_frame_alloc): dict of depth / color / normal views + a
        handle to pass to free_frame.  Page-locked destinations keep ctr_multi_submit asynchronous."""
        L = _lib.hip_lib()
        px = self.w * self.h
        d, c, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        if L.ctr_frame_alloc(px, C.byref(d), C.byref(c), C.byref(n)):
            raise RuntimeError("ctr_frame_alloc failed")
        blk = np.ctypeslib.as_array(d, shape=(7 * px,))
        return dict(depth=blk[:px].reshape(self.h, self.w), color=blk[px:4 * px].reshape(self.h, self.w, 3),
                    normal=blk[4 * px:7 * px].reshape(self.h, self.w, 3), _ptr=d)

    @staticmethod
    def free_frame(fr):
        _lib.hip_lib().ctr_frame_free(fr["_ptr"])

    def submit(self, into, fudge=1e-3, bounces=5, block_rows=8):
        """Queue one frame into the buffers of `into` (dict with depth / color / normal arrays of the group's size) and
        return at once (ctr_multi_submit); at most two frames in flight."""
        L = _lib.hip_lib()
        st = L.ctr_multi_submit(self._h, C.c_float(fudge), bounces, block_rows, into["depth"].ctypes.data,
                                into["color"].ctypes.data, into["normal"].ctypes.data)
        if st:
            raise RuntimeError(f"ctr_multi_submit failed ({st}): {L.ctr_last_error().decode()}")

    def wait(self):
        """Block until the oldest queued frame is complete (ctr_multi_wait); returns its statistics."""
        L = _lib.hip_lib()
        stats = RenderStats()
        st = L.ctr_multi_wait(self._h, C.byref(stats))
        if st:
            raise RuntimeError(f"ctr_multi_wait failed ({st}): {L.ctr_last_error().decode()}")
        ms = (C.c_double * len(self.devices))()
        L.ctr_multi_kernel_ms(self._h, ms, len(self.devices))
        return dict(ray_count=int(stats.ray_count), kernel_ms=stats.kernel_ms, total_ms=stats.total_ms,
                    max_depth=float(stats.max_depth), rows=int(stats.rows), kernel_ms_per_device=list(ms))

    def _free_pinned(self):
        if getattr(self, "_pin_px", 0):
            self._pin = None
            _lib.hip_lib().ctr_frame_free(self._pin_ptr)
            self._pin_px = 0

    def close(self):
        if self._h:
            self._free_pinned()
            _lib.hip_lib().ctr_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
