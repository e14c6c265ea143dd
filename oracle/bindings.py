"""ctypes bindings of oracle/libctr_oracle.so (plain-C restatement) and oracle/_ref/libcutrace_ref.so
(the reference's own headers built for the host).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

from cutrace_amd import _lib, make_rows, rows_count  # the ABI structs only; never the other way round

HERE = os.path.dirname(os.path.abspath(__file__))
_oracle = None
_ref = None


def _render_sig(fn, uv=False, ex=False):
    fn.argtypes = [C.POINTER(_lib.SceneDesc), C.c_float, C.c_int, C.POINTER(_lib.Rows), C.c_int, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + ([C.c_void_p] if (uv or ex) else []) + ([C.c_int] if ex else [])
    fn.restype = C.c_int


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.environ.get("CUTRACE_ORACLE_LIB") or os.path.join(HERE, "libctr_oracle.so")  # override: sanitizer builds
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle oracle`")
        L = C.CDLL(path)
        _render_sig(L.orc_render)
        _render_sig(L.orc_render_uv, uv=True)
        _render_sig(L.orc_render_ex, ex=True)
        L.orc_look_at.argtypes = [C.POINTER(_lib.Camera), _lib.Vec3, _lib.Vec3, _lib.Vec3]
        L.orc_quantise_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        L.orc_quantise_normal.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_quantise_color.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_trace_pixel.argtypes = [C.POINTER(_lib.SceneDesc), C.c_uint64, C.c_uint64, C.c_float, C.c_int]
        _oracle = L
    return _oracle


def ref_lib():
    """None when oracle/_ref has not been built (it needs /root/reference at build time)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libcutrace_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        _render_sig(L.ref_render)
        if hasattr(L, "ref_render_uv"):
            _render_sig(L.ref_render_uv, uv=True)
        if hasattr(L, "ref_render_ex"):
            _render_sig(L.ref_render_ex, ex=True)
        L.ref_look_at.argtypes = [C.POINTER(_lib.Camera), _lib.Vec3, _lib.Vec3, _lib.Vec3]
        _ref = L
    return _ref


_ref_cuda = None


def ref_cudaminmax_lib():
    """DIAGNOSTIC flavour of the reference build: unqualified min/max behave like nvcc's device float overloads
    (fminf/fmaxf: a NaN operand is dropped).  None when not built.  Never the parity target."""
    global _ref_cuda
    if _ref_cuda is None:
        path = os.path.join(HERE, "_ref", "libcutrace_ref_cudaminmax.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        _render_sig(L.ref_render)
        _ref_cuda = L
    return _ref_cuda


def ref_cudaminmax_render(scene, fudge=1e-3, bounces=5, rows=None, threads=1, hit_ids=True):
    L = ref_cudaminmax_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libcutrace_ref_cudaminmax.so not built (needs /root/reference)")
    return _cpu_render(L.ref_render, scene, fudge, bounces, rows, threads, hit_ids)


_ref_fmad = None


def ref_fmad_lib():
    """DIAGNOSTIC flavour: fminf/fmaxf semantics AND contraction of a*b+c into fused multiply-adds allowed (g++ -ffp-contract=fast
    -mfma) — the kind of difference nvcc's default --fmad=true makes.  None when not built.  Never the parity target."""
    global _ref_fmad
    if _ref_fmad is None:
        path = os.path.join(HERE, "_ref", "libcutrace_ref_fmad.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        _render_sig(L.ref_render)
        _ref_fmad = L
    return _ref_fmad


def ref_fmad_render(scene, fudge=1e-3, bounces=5, rows=None, threads=1, hit_ids=True):
    L = ref_fmad_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libcutrace_ref_fmad.so not built (needs /root/reference)")
    return _cpu_render(L.ref_render, scene, fudge, bounces, rows, threads, hit_ids)


def _cpu_render(fn, scene, fudge, bounces, rows, threads, want_hit_ids, want_uv=False, ignore_transparent=None):
    w, h = scene.size
    r = make_rows(h, rows)
    n = rows_count(h, rows)
    depth = np.empty((n, w), np.float32)
    color = np.empty((n, w, 3), np.float32)
    normal = np.empty((n, w, 3), np.float32)
    hit = np.empty((n, w), np.int64) if want_hit_ids else None
    counters = (C.c_uint64 * 2)()
    uv = np.empty((n, w, 2), np.float32) if want_uv else None
    args = [scene.desc, C.c_float(fudge), bounces, C.byref(r), threads, depth.ctypes.data, color.ctypes.data,
            normal.ctypes.data, hit.ctypes.data if hit is not None else None, counters]
    if want_uv or ignore_transparent is not None:
        args.append(uv.ctypes.data if uv is not None else None)
    if ignore_transparent is not None:   # the *_render_ex entry points
        args.append(1 if ignore_transparent else 0)
    st = fn(*args)
    if st:
        raise RuntimeError(f"cpu render failed: {st}")
    return dict(depth=depth, color=color, normal=normal, hit_id=hit, ray_count=int(counters[0]),
                alg_bytes=int(counters[1]), uv=uv)


def oracle_render(scene, fudge=1e-3, bounces=5, rows=None, threads=1, hit_ids=True, uv=False, ignore_transparent_primary=False):
    """CPU restatement (oracle/ctr_oracle.c).  uv=True: also the texture coordinates of the primary hit.
    ignore_transparent_primary: the kernel.hpp:52 cast with ray_cast's ignore_transparent = true (ray_cast.hpp:39-40)."""
    L = oracle_lib()
    if ignore_transparent_primary:
        return _cpu_render(L.orc_render_ex, scene, fudge, bounces, rows, threads, hit_ids, uv, True)
    return _cpu_render(L.orc_render_uv if uv else L.orc_render, scene, fudge, bounces, rows, threads, hit_ids, uv)


def ref_render(scene, fudge=1e-3, bounces=5, rows=None, threads=1, hit_ids=True, uv=False, ignore_transparent_primary=False):
    """The reference's own headers compiled for the host (oracle/_ref)."""
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libcutrace_ref.so not built (needs /root/reference)")
    if ignore_transparent_primary:
        return _cpu_render(L.ref_render_ex, scene, fudge, bounces, rows, threads, hit_ids, uv, True)
    return _cpu_render(L.ref_render_uv if uv else L.ref_render, scene, fudge, bounces, rows, threads, hit_ids, uv)
