"""ctypes binding of oracle/librt_oracle.so (oracle/rt_oracle.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.int32), ("inst", np.int32)])


def build(force=False):
    so = os.path.join(_HERE, "librt_oracle.so")
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _bind(L):
    L.orc_create.restype = C.c_void_p
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.orc_set_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_set_uniforms.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_set_skybox.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.orc_intersect.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_hit_attributes.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_jitter.argtypes = [C.c_float, C.c_float, C.c_float]
    L.orc_jitter.restype = C.c_float
    L.orc_sin.argtypes = [C.c_double]
    L.orc_sin.restype = C.c_double
    L.orc_pow100.argtypes = [C.c_float]
    L.orc_pow100.restype = C.c_float
    L.orc_invert_affine.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_sample_sky.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_tri_test.argtypes = [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p]
    L.orc_bounce_step.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_render_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_set_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.orc_set_instance_types.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_pow_int.argtypes = [C.c_float, C.c_uint32]
    L.orc_pow_int.restype = C.c_float
    L.orc_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


_NATIVE = None


def lib_native():
    """The -O3 -march=native build (bench.py's cpu_baseline): always compiled on the machine that runs it."""
    global _NATIVE
    if _NATIVE is None:
        so = os.path.join(_HERE, "librt_oracle_native.so")
        if os.path.exists(so):
            os.remove(so)                     # a copy built on another CPU may hold instructions this one lacks
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle_native.so"], stdout=subprocess.DEVNULL)
        _NATIVE = _bind(C.CDLL(so))
    return _NATIVE


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, native=False):
        self.L = lib_native() if native else lib()
        self.h = C.c_void_p(self.L.orc_create())
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_geometry(self, verts, idx, ranges):
        verts = np.ascontiguousarray(verts, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        r = np.zeros(len(ranges), dtype=np.dtype([("ff", np.uint64), ("fi", np.uint64), ("pc", np.uint32), ("pad", np.uint32)]))
        for i, (ff, fi, pc) in enumerate(ranges):
            r[i] = (ff, fi, pc, 0)
        rc = self.L.orc_set_geometry(self.h, _ptr(verts), verts.size, _ptr(idx), idx.size, _ptr(r), len(ranges))
        assert rc == 0, "orc_set_geometry failed"

    def set_instances(self, packed_list):
        buf = b"".join(packed_list)
        rc = self.L.orc_set_instances(self.h, C.c_char_p(buf), len(packed_list))
        assert rc == 0, "orc_set_instances failed"

    def set_uniforms(self, packed):
        assert len(packed) == 104
        assert self.L.orc_set_uniforms(self.h, C.c_char_p(packed)) == 0

    def set_skybox(self, faces):
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        h, w = faces[0].shape[:2]
        arr = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        assert self.L.orc_set_skybox(self.h, arr, w, h) == 0

    def set_materials(self, table, prim_material):
        """table: structured array of 48-byte material records (or None to remove), prim_material: uint32 per triangle"""
        if table is None or len(table) == 0:
            assert self.L.orc_set_materials(self.h, None, 0, None, 0) == 0
            return
        table = np.ascontiguousarray(table)
        assert table.dtype.itemsize == 48
        pm = np.ascontiguousarray(prim_material, np.uint32)
        assert self.L.orc_set_materials(self.h, _ptr(table), len(table), _ptr(pm), len(pm)) == 0, "orc_set_materials failed"

    def set_instance_types(self, types):
        t = np.ascontiguousarray(types if types is not None else [], np.uint32)
        assert self.L.orc_set_instance_types(self.h, _ptr(t) if len(t) else None, len(t)) == 0

    def intersect(self, rays8, any_hit=False, use_bvh=True, counts=False):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays8), HIT_DTYPE)
        vc = np.zeros(2, np.uint64)
        self.L.orc_intersect(self.h, len(rays8), _ptr(rays8), int(any_hit), int(use_bvh), _ptr(out), _ptr(vc) if counts else None)
        return (out, vc) if counts else out

    def hit_attributes(self, hits):
        hits = np.ascontiguousarray(hits)
        out = np.zeros((len(hits), 7), np.float32)
        self.L.orc_hit_attributes(self.h, len(hits), _ptr(hits), _ptr(out))
        return out

    def render(self, W, H, y0=0, y1=None, threads=0, use_bvh=True):
        y1 = H if y1 is None else y1
        out = np.zeros((H, W, 4), np.float32)
        rc = np.zeros(3, np.uint64)
        assert self.L.orc_render(self.h, W, H, y0, y1, _ptr(out), threads, int(use_bvh), _ptr(rc)) == 0
        return out, rc

    def sample_sky(self, d):
        d = np.ascontiguousarray(d, np.float32)
        o = np.zeros(3, np.float32)
        self.L.orc_sample_sky(self.h, _ptr(d), _ptr(o))
        return o

    def bounce_step(self, od6, sample_index, hits):
        """replays recorded bounces through the renderer's own per-bounce code; returns (n, 28) floats (see orc_bounce_step)"""
        od6 = np.ascontiguousarray(od6, np.float32).reshape(-1, 6)
        si = np.ascontiguousarray(sample_index, np.uint32)
        hits = np.ascontiguousarray(hits, HIT_DTYPE)
        out = np.zeros((len(od6), 28), np.float32)
        assert self.L.orc_bounce_step(self.h, len(od6), _ptr(od6), _ptr(si), _ptr(hits), _ptr(out)) == 0
        return out

    def render_pixels(self, W, H, xy, use_bvh=True):
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        out = np.zeros((len(xy), 4), np.float32)
        assert self.L.orc_render_pixels(self.h, W, H, len(xy), _ptr(xy), _ptr(out), int(use_bvh)) == 0
        return out

    def primary_ray(self, px, py, W, H, i):
        o = np.zeros(6, np.float32)
        self.L.orc_primary_ray(self.h, px, py, W, H, i, _ptr(o))
        return o
