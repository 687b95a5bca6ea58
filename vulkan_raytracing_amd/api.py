"""ctypes mirror of include/rt_api.h (the C ABI of librt_mi355x.so)."""
import ctypes as C

import numpy as np

from . import _native

HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.int32), ("inst", np.int32)])
MESH_RANGE_DTYPE = np.dtype([("first_float", np.uint64), ("first_index", np.uint64), ("prim_count", np.uint32), ("reserved", np.uint32)])
INSTANCE_DTYPE = np.dtype([("transform", np.float32, 12), ("custom_index_and_mask", np.uint32), ("sbt_offset_and_flags", np.uint32), ("mesh", np.uint64)])
UNIFORMS_DTYPE = np.dtype([("position", np.float32, 4), ("right", np.float32, 4), ("up", np.float32, 4), ("forward", np.float32, 4),
                           ("light_position", np.float32, 3), ("light_intensity", np.float32),
                           ("max_bounce_count", np.uint32), ("samples_per_pixel", np.uint32),
                           ("center_object_type", np.uint32), ("orbiting_object_type", np.uint32),
                           ("orbiting_object_primitive_offset", np.uint32), ("orbiting_object_vertex_offset", np.uint32)])
MATERIAL_DTYPE = np.dtype([("ka", np.float32, 3), ("ns", np.float32), ("kd", np.float32, 3), ("ni", np.float32), ("ks", np.float32, 3), ("type", np.uint32)])
MATERIAL_TYPE_OF_INSTANCE = 0xFFFFFFFF
assert MATERIAL_DTYPE.itemsize == 48
assert INSTANCE_DTYPE.itemsize == 64 and UNIFORMS_DTYPE.itemsize == 104 and MESH_RANGE_DTYPE.itemsize == 24


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("node_visits_shadow", C.c_uint64), ("tri_tests_shadow", C.c_uint64),
                ("diag", C.c_uint64 * 6), ("closest_rays", C.c_uint64),
                ("ms_frame", C.c_float), ("ms_raygen", C.c_float), ("ms_trace_closest", C.c_float), ("ms_trace_shadow", C.c_float),
                ("ms_shade", C.c_float), ("ms_resolve", C.c_float),
                ("launches_trace_closest", C.c_uint32), ("launches_total", C.c_uint32), ("timed_frames", C.c_uint32), ("ms_tail", C.c_float),
                ("bvh_node_bytes", C.c_uint32), ("bvh_tri_bytes", C.c_uint32), ("tail_faults", C.c_uint32), ("frames_rerendered", C.c_uint32),
                ("blob_tiles", C.c_uint64), ("blob_tiles_large", C.c_uint64), ("blob_tiles_refused", C.c_uint64), ("blob_nodes", C.c_uint64), ("blob_tris", C.c_uint64),
                ("tile_rays", C.c_uint64), ("tile_rays_handed_on", C.c_uint64), ("tile_diag", C.c_uint64 * 6), ("rays_shadow_untraced", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k in ("diag", "tile_diag") else getattr(self, k)) for k, _ in self._fields_}

    @property
    def rays_total(self):
        return self.rays_primary + self.rays_secondary + self.rays_shadow


EXPORTS = ["rt_create", "rt_create_frame_slot", "rt_destroy", "rt_upload_geometry", "rt_build_blas", "rt_set_instances", "rt_set_materials", "rt_set_instance_types", "rt_set_uniforms", "rt_set_skybox",
           "rt_trace", "rt_trace_async", "rt_trace_wait", "rt_trace_shard", "rt_set_batch", "rt_trace_shard_batch", "rt_assemble_shards", "rt_shard_rows", "rt_synchronize", "rt_get_stats", "rt_set_timing", "rt_intersect",
           "rt_trace_counting", "rt_set_param", "rt_debug_check_builders", "rt_debug_sizing", "rt_last_error", "rt_device_info", "rt_abi_version"]

_LIBS = {}


def lib(variant=None):
    """librt_mi355x.so, or librt_mi355x_<variant>.so (variant "alt": the build that also holds the traversal kernels which
    measured slower — k_packet, the quad/BVH4 kernel, 4-ary records; `make alt`).  RT_LIB_VARIANT names the default."""
    if variant not in _LIBS:
        L = _native.load_rt(variant)
        vp = C.c_void_p
        L.rt_create.argtypes = [C.POINTER(vp), C.c_int]
        L.rt_create_frame_slot.argtypes = [vp, C.POINTER(vp)]
        L.rt_destroy.argtypes = [vp]
        L.rt_destroy.restype = None
        L.rt_upload_geometry.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int]
        L.rt_build_blas.argtypes = [vp, C.c_int]
        L.rt_set_instances.argtypes = [vp, vp, C.c_int, C.c_int]
        L.rt_set_materials.argtypes = [vp, vp, C.c_int, vp, C.c_size_t]
        L.rt_set_instance_types.argtypes = [vp, vp, C.c_int]
        L.rt_set_uniforms.argtypes = [vp, vp]
        L.rt_set_skybox.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int]
        L.rt_trace.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(RtStats)]
        L.rt_trace_counting.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(RtStats)]
        L.rt_trace_async.argtypes = [vp, C.c_int, C.c_int]
        L.rt_trace_wait.argtypes = [vp, C.POINTER(vp), C.POINTER(RtStats)]
        L.rt_trace_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
        L.rt_trace_shard_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.c_size_t, vp]
        L.rt_set_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.rt_assemble_shards.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
        L.rt_shard_rows.argtypes = [C.c_int] * 4
        L.rt_synchronize.argtypes = [vp]
        L.rt_get_stats.argtypes = [vp, C.POINTER(RtStats)]
        L.rt_set_timing.argtypes = [vp, C.c_int]
        L.rt_set_param.argtypes = [vp, C.c_char_p, C.c_int]
        L.rt_debug_check_builders.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp]
        L.rt_debug_sizing.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, vp]
        L.rt_intersect.argtypes = [vp, C.c_size_t, vp, C.c_int, vp, C.c_int, C.POINTER(RtStats)]
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = C.c_char_p
        L.rt_device_info.argtypes = [vp]
        L.rt_device_info.restype = C.c_char_p
        _LIBS[variant] = L
    return _LIBS[variant]


class RtError(RuntimeError):
    """Mirror of the reference's std::runtime_error("Vulkan API exception: return code N (fn)")
    (src/main.cpp:138-147)."""

    def __init__(self, code, fn, detail):
        super().__init__("RT API exception: return code %d (%s): %s" % (code, fn, detail))
        self.code = code


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RtContext:
    """One context = one GPU (rt_create).  Methods map 1:1 onto the C ABI."""

    def __init__(self, device=0, _parent=None, variant=None):
        self.L = _parent.L if _parent is not None else lib(variant)
        h = C.c_void_p()
        if _parent is not None:
            rc = self.L.rt_create_frame_slot(_parent.h, C.byref(h))
        else:
            rc = self.L.rt_create(C.byref(h), device)
        if rc:
            raise RtError(rc, "rt_create_frame_slot" if _parent is not None else "rt_create", self.L.rt_last_error(None).decode())
        self.h = h

    def frame_slot(self):
        """rt_create_frame_slot: a context for one more frame in flight that shares this context's scene (geometry, BLAS,
        cube map) and owns its instances/TLAS, uniforms, queues and stream."""
        return RtContext(_parent=self)

    def _chk(self, rc, fn):
        if rc:
            raise RtError(rc, fn, self.L.rt_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_info(self):
        return self.L.rt_device_info(self.h).decode()

    def upload_geometry(self, verts6, idx, ranges, build=True):
        verts6 = np.ascontiguousarray(verts6, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        r = np.zeros(len(ranges), MESH_RANGE_DTYPE)
        for i, (ff, fi, pc) in enumerate(ranges):
            r[i] = (ff, fi, pc, 0)
        self._chk(self.L.rt_upload_geometry(self.h, _p(verts6), verts6.size, _p(idx), idx.size, _p(r), len(ranges)), "rt_upload_geometry")
        if build:
            for m in range(len(ranges)):
                self.build_blas(m)

    def build_blas(self, mesh):
        self._chk(self.L.rt_build_blas(self.h, mesh), "rt_build_blas")

    def set_instances(self, instances, update=False):
        inst = np.ascontiguousarray(instances, INSTANCE_DTYPE)
        self._chk(self.L.rt_set_instances(self.h, _p(inst), len(inst), int(update)), "rt_set_instances")

    def set_materials(self, table, prim_material=None):
        """row n4: MTL material table + material id of every triangle of the index buffer; table None/empty removes it"""
        if table is None or len(table) == 0:
            self._chk(self.L.rt_set_materials(self.h, None, 0, None, 0), "rt_set_materials")
            return
        t = np.ascontiguousarray(table, MATERIAL_DTYPE)
        pm = np.ascontiguousarray(prim_material, np.uint32)
        self._chk(self.L.rt_set_materials(self.h, _p(t), len(t), _p(pm), len(pm)), "rt_set_materials")

    def set_instance_types(self, types):
        t = np.ascontiguousarray(types if types is not None else [], np.uint32)
        self._chk(self.L.rt_set_instance_types(self.h, _p(t) if len(t) else None, len(t)), "rt_set_instance_types")

    def set_uniforms(self, uniforms):
        u = np.ascontiguousarray(uniforms, UNIFORMS_DTYPE).reshape(1)
        self._chk(self.L.rt_set_uniforms(self.h, _p(u)), "rt_set_uniforms")

    def set_skybox(self, faces):
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        assert len(faces) == 6
        h, w = faces[0].shape[:2]
        arr = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        self._chk(self.L.rt_set_skybox(self.h, arr, w, h), "rt_set_skybox")

    def trace(self, W, H, counting=False):
        out = np.zeros((H, W, 4), np.uint8 if getattr(self, "_rgba8", False) else np.float32)
        st = RtStats()
        fn = self.L.rt_trace_counting if counting else self.L.rt_trace
        self._chk(fn(self.h, W, H, _p(out), C.byref(st)), "rt_trace")
        return out, st

    def shard_rows(self, H, band_rows, shard, n_shards):
        return self.L.rt_shard_rows(H, band_rows, shard, n_shards)

    def trace_async(self, W, H):
        """Enqueue a frame and its copy to the context's pinned host buffer; returns at once (one pending frame per context)."""
        self._chk(self.L.rt_trace_async(self.h, W, H), "rt_trace_async")
        self._async_shape = (H, W, 4)

    def trace_wait(self, copy=True):
        """Wait for the frame of trace_async; returns (pixels, stats).  With copy=False the array aliases the pinned
        buffer and is valid until the next trace_async on this context."""
        px = C.c_void_p()
        st = RtStats()
        self._chk(self.L.rt_trace_wait(self.h, C.byref(px), C.byref(st)), "rt_trace_wait")
        H, W, _ = self._async_shape
        ct = C.c_uint8 if getattr(self, "_rgba8", False) else C.c_float
        img = np.ctypeslib.as_array(C.cast(px, C.POINTER(ct)), shape=(H, W, 4))
        return (img.copy() if copy else img), st

    def trace_shard(self, W, H, band_rows, shard, n_shards, d_out_ptr, capacity_bytes, stream_ptr=None):
        self._chk(self.L.rt_trace_shard(self.h, W, H, band_rows, shard, n_shards, C.c_void_p(d_out_ptr), capacity_bytes,
                                        C.c_void_p(stream_ptr) if stream_ptr else None), "rt_trace_shard")

    def set_batch(self, instances, uniforms, update=False):
        """rt_set_batch: instances (K, n) records and K uniform blocks — K consecutive frames for one pass of the pipeline"""
        inst = np.ascontiguousarray(instances, INSTANCE_DTYPE)
        u = np.ascontiguousarray(uniforms, UNIFORMS_DTYPE).reshape(-1)
        K = len(u)
        inst = inst.reshape(K, -1)
        self._chk(self.L.rt_set_batch(self.h, K, _p(inst), inst.shape[1], _p(u), int(update)), "rt_set_batch")

    def trace_shard_batch(self, W, H, band_rows, shard, n_shards, d_out_ptr, capacity_bytes, stream_ptr=None, frame_stride_bytes=0):
        self._chk(self.L.rt_trace_shard_batch(self.h, W, H, band_rows, shard, n_shards, C.c_void_p(d_out_ptr), frame_stride_bytes, capacity_bytes,
                                              C.c_void_p(stream_ptr) if stream_ptr else None), "rt_trace_shard_batch")

    def assemble_shards(self, d_gathered_ptr, n_shards, shard_stride_bytes, W, H, band_rows, d_frame_ptr, capacity_bytes, stream_ptr=None):
        self._chk(self.L.rt_assemble_shards(self.h, C.c_void_p(d_gathered_ptr), n_shards, shard_stride_bytes, W, H, band_rows, C.c_void_p(d_frame_ptr),
                                            capacity_bytes, C.c_void_p(stream_ptr) if stream_ptr else None), "rt_assemble_shards")

    def synchronize(self):
        self._chk(self.L.rt_synchronize(self.h), "rt_synchronize")

    def stats(self):
        st = RtStats()
        self._chk(self.L.rt_get_stats(self.h, C.byref(st)), "rt_get_stats")
        return st

    def set_timing(self, on):
        self._chk(self.L.rt_set_timing(self.h, int(on)), "rt_set_timing")

    def set_param(self, name, value):
        self._chk(self.L.rt_set_param(self.h, name.encode(), int(value)), "rt_set_param")
        if name in ("output_rgba8", "output_bgra8"):
            self._rgba8 = bool(value)   # frames come back as uint8 (H, W, 4)

    def intersect(self, rays8, any_hit=False, counting=False):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays8), HIT_DTYPE)
        st = RtStats()
        self._chk(self.L.rt_intersect(self.h, len(rays8), _p(rays8), int(any_hit), _p(out), int(counting), C.byref(st)), "rt_intersect")
        return out, st


def check_builders(verts6, idx):
    """rt_debug_check_builders: host-only invariants of the BVH builders; returns (status, stats dict)."""
    verts6 = np.ascontiguousarray(verts6, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    out = np.zeros(8, np.uint64)
    rc = lib().rt_debug_check_builders(_p(verts6), verts6.size, _p(idx), idx.size, _p(out))
    keys = ("nodes", "leaves", "depth", "max_leaf", "bvh4_nodes", "bvh4_stack_need", "violations", "reached")
    return rc, dict(zip(keys, (int(x) for x in out)))
