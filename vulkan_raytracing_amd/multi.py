"""ctypes mirror of include/rt_multi.h (librt_multi.so): several MI355X of one node driven by ONE host process, RCCL called
from C++.  The reference binds one device (src/main.cpp:928) and copies the traced image into the presented one
(src/main.cpp:2683-2686); here every device renders its interleaved 8-row bands and ONE gather per frame brings the compact
shards to the first device."""
import ctypes as C
import os

import numpy as np

from . import _native, api
from .api import INSTANCE_DTYPE, MATERIAL_DTYPE, MESH_RANGE_DTYPE, UNIFORMS_DTYPE, RtError, RtStats

RTM_LOOPBACK = 1
EXPORTS = ["rtm_create", "rtm_destroy", "rtm_upload_geometry", "rtm_build_blas", "rtm_set_skybox", "rtm_set_param", "rtm_set_materials",
           "rtm_set_instance_types", "rtm_set_timing", "rtm_set_instances", "rtm_set_batch", "rtm_set_uniforms", "rtm_trace_async", "rtm_trace_wait",
           "rtm_frame_device", "rtm_device_count", "rtm_last_error"]
_M = None


def mlib():
    global _M
    if _M is None:
        api.lib()   # librt_mi355x.so first (and torch's HIP runtime before it, see _native.load_rt)
        L = C.CDLL(os.path.join(_native.PKG_DIR, "librt_multi.so"))
        vp = C.c_void_p
        L.rtm_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
        L.rtm_destroy.argtypes = [vp]
        L.rtm_destroy.restype = None
        L.rtm_upload_geometry.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int]
        L.rtm_build_blas.argtypes = [vp, C.c_int]
        L.rtm_set_skybox.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int]
        L.rtm_set_param.argtypes = [vp, C.c_char_p, C.c_int]
        L.rtm_set_materials.argtypes = [vp, vp, C.c_int, vp, C.c_size_t]
        L.rtm_set_instance_types.argtypes = [vp, C.c_int, vp, C.c_int]
        L.rtm_set_timing.argtypes = [vp, C.c_int]
        L.rtm_set_instances.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
        L.rtm_set_batch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        L.rtm_set_uniforms.argtypes = [vp, C.c_int, vp]
        L.rtm_trace_async.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.rtm_trace_wait.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(RtStats)]
        L.rtm_frame_device.argtypes = [vp, C.c_int]
        L.rtm_frame_device.restype = vp
        L.rtm_device_count.argtypes = [vp]
        L.rtm_last_error.argtypes = [vp]
        L.rtm_last_error.restype = C.c_char_p
        _M = L
    return _M


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RtMulti:
    """rtm_create: a scene + `frames_in_flight` frame slots on each of `device_ids`; device_ids[0] assembles the frame."""

    def __init__(self, device_ids, frames_in_flight=4, loopback=False):
        self.L = mlib()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.L.rtm_create(C.byref(h), len(device_ids), ids, frames_in_flight, RTM_LOOPBACK if loopback else 0)
        if rc:
            raise RtError(rc, "rtm_create", self.L.rtm_last_error(None).decode())
        self.h, self.P, self.n = h, frames_in_flight, len(device_ids)
        self._rgba8 = False
        self._shape = {}
        self._k = {}

    def _chk(self, rc, fn):
        if rc:
            raise RtError(rc, fn, self.L.rtm_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rtm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # the four setters a workloads.Workload.apply() needs
    def upload_geometry(self, verts6, idx, ranges, build=True):
        verts6 = np.ascontiguousarray(verts6, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        r = np.zeros(len(ranges), MESH_RANGE_DTYPE)
        for i, (ff, fi, pc) in enumerate(ranges):
            r[i] = (ff, fi, pc, 0)
        self._chk(self.L.rtm_upload_geometry(self.h, _p(verts6), verts6.size, _p(idx), idx.size, _p(r), len(ranges)), "rtm_upload_geometry")
        if build:
            for m in range(len(ranges)):
                self._chk(self.L.rtm_build_blas(self.h, m), "rtm_build_blas")

    def set_instances(self, instances, update=False, slot=None):
        inst = np.ascontiguousarray(instances, INSTANCE_DTYPE)
        for j in (range(self.P) if slot is None else [slot]):
            self._chk(self.L.rtm_set_instances(self.h, j, _p(inst), len(inst), int(update)), "rtm_set_instances")
            self._k[j] = 1

    def set_batch(self, slot, instances, uniforms, update=False):
        """rtm_set_batch: the slot's next trace_async renders K consecutive frames (instances (K, n), K uniform blocks) in one pass"""
        u = np.ascontiguousarray(uniforms, UNIFORMS_DTYPE).reshape(-1)
        K = len(u)
        inst = np.ascontiguousarray(instances, INSTANCE_DTYPE).reshape(K, -1)
        self._chk(self.L.rtm_set_batch(self.h, slot, K, _p(inst), inst.shape[1], _p(u), int(update)), "rtm_set_batch")
        self._k[slot] = K

    def set_uniforms(self, uniforms, slot=None):
        u = np.ascontiguousarray(uniforms, UNIFORMS_DTYPE).reshape(1)
        for j in (range(self.P) if slot is None else [slot]):
            self._chk(self.L.rtm_set_uniforms(self.h, j, _p(u)), "rtm_set_uniforms")

    def set_skybox(self, faces):
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        h, w = faces[0].shape[:2]
        arr = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        self._chk(self.L.rtm_set_skybox(self.h, arr, w, h), "rtm_set_skybox")

    def set_materials(self, table, prim_material=None):
        if table is None or len(table) == 0:
            self._chk(self.L.rtm_set_materials(self.h, None, 0, None, 0), "rtm_set_materials")
            return
        t = np.ascontiguousarray(table, MATERIAL_DTYPE)
        pm = np.ascontiguousarray(prim_material, np.uint32)
        self._chk(self.L.rtm_set_materials(self.h, _p(t), len(t), _p(pm), len(pm)), "rtm_set_materials")

    def set_instance_types(self, types, slot=None):
        t = np.ascontiguousarray(types if types is not None else [], np.uint32)
        for j in (range(self.P) if slot is None else [slot]):
            self._chk(self.L.rtm_set_instance_types(self.h, j, _p(t) if len(t) else None, len(t)), "rtm_set_instance_types")

    def set_param(self, name, value):
        self._chk(self.L.rtm_set_param(self.h, name.encode(), int(value)), "rtm_set_param")
        if name in ("output_rgba8", "output_bgra8"):
            self._rgba8 = bool(value)

    def set_timing(self, level):
        self._chk(self.L.rtm_set_timing(self.h, int(level)), "rtm_set_timing")

    def trace_async(self, slot, W, H):
        self._chk(self.L.rtm_trace_async(self.h, slot, W, H), "rtm_trace_async")
        K = self._k.get(slot, 1)
        self._shape[slot] = (H, W, 4) if K == 1 else (K, H, W, 4)

    def trace_wait(self, slot, copy=True):
        """(pixels or None with host_copy 0, stats)"""
        px = C.c_void_p()
        st = RtStats()
        self._chk(self.L.rtm_trace_wait(self.h, slot, C.byref(px), C.byref(st)), "rtm_trace_wait")
        if not px.value:
            return None, st
        ct = C.c_uint8 if self._rgba8 else C.c_float
        img = np.ctypeslib.as_array(C.cast(px, C.POINTER(ct)), shape=self._shape[slot])
        return (img.copy() if copy else img), st

    def frame_device_ptr(self, slot):
        return self.L.rtm_frame_device(self.h, slot)
