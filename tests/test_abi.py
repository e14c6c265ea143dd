"""The C-ABI libraries load and export every symbol the headers declare (no GPU needed)."""
import ctypes
import os
import re

from cutrace_amd import _lib

ROOT = _lib.ROOT


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ctr_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("cutrace_amd.h")
    assert set(names) == set(_lib.HIP_SYMBOLS), (names, _lib.HIP_SYMBOLS)
    L = _lib.hip_lib()
    for n in names:
        assert hasattr(L, n), f"libcutrace_amd.so does not export {n}"
    assert L.ctr_abi_version() == 3


def test_host_library_exports_every_declared_symbol():
    names = declared("cutrace_host.h")
    assert set(names) == set(_lib.HOST_SYMBOLS), (names, _lib.HOST_SYMBOLS)
    L = _lib.host_lib()
    for n in names:
        assert hasattr(L, n), f"libcutrace_host.so does not export {n}"


def test_struct_sizes_match_the_header_layout():
    # sizes the C compiler gives the header's structs (x86-64 SysV)
    assert ctypes.sizeof(_lib.Vec3) == 12
    assert ctypes.sizeof(_lib.Triangle) == 36
    assert ctypes.sizeof(_lib.Object) == 72
    assert ctypes.sizeof(_lib.Light) == 28
    assert ctypes.sizeof(_lib.Material) == 32
    assert ctypes.sizeof(_lib.Camera) == 80
    assert ctypes.sizeof(_lib.Rows) == 32
    assert ctypes.sizeof(_lib.RenderStats) == 40


def test_no_gpu_means_loud_failure_not_fallback(ca):
    """Without a HIP device ctr_scene_create must FAIL (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return  # covered by the gpu tests
    from tests.conftest import load_scene
    s = load_scene(ca, "triangle")
    try:
        ca.DeviceScene(s)
    except RuntimeError as e:
        assert "ctr_scene_create failed" in str(e)
    else:
        raise AssertionError("ctr_scene_create succeeded without a GPU?")


def test_scene_create_validates_the_description_before_touching_the_gpu(ca):
    """Malformed descriptions are rejected with CTR_E_INVALID (1) — the kernel trusts these indices."""
    import ctypes as C
    from tests.conftest import load_scene
    L = _lib.hip_lib()
    s = load_scene(ca, "bunny")
    d = s.desc.contents

    def create(desc):
        h = C.c_void_p()
        st = L.ctr_scene_create(C.byref(desc), 0, C.byref(h))
        if st == 0:
            L.ctr_scene_destroy(h)
        return st

    def clone():
        c = _lib.SceneDesc()
        C.memmove(C.byref(c), C.byref(d), C.sizeof(_lib.SceneDesc))
        objs = (_lib.Object * d.n_objects)()
        C.memmove(objs, d.objects, C.sizeof(objs))
        c.objects = C.cast(objs, C.POINTER(_lib.Object))
        return c, objs

    c, objs = clone()
    objs[1].mat_idx = 99
    assert create(c) == 1 and b"material index" in L.ctr_last_error()
    c, objs = clone()
    objs[0].tri_count = 5000
    assert create(c) == 1 and b"triangle range" in L.ctr_last_error()
    c, objs = clone()
    objs[2].type = 7
    assert create(c) == 1 and b"bad type" in L.ctr_last_error()
    h = C.c_void_p()
    assert L.ctr_scene_create(None, 0, C.byref(h)) == 1
    assert L.ctr_render(None, C.c_float(1e-3), 5, None, None, None, None, None) == 1
