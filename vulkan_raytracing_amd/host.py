"""Python view of the C++ host-side code (include/rt_host.hpp, include/camera.h, include/obj_loader.h)
through librt_host.so.  Mirrors what the reference's main() does around its ray-tracing stage:
OBJ ingest and buffer flattening (src/main.cpp:1606-1729), instance records (:538-551, :1805-1825),
the uniform block (:1847-1873), the animation (:2836-2844), the camera (src/camera.cpp) and the
skybox faces (:2064-2080)."""
import ctypes as C
import os

import numpy as np

from . import _native
from .api import INSTANCE_DTYPE, MATERIAL_DTYPE, MESH_RANGE_DTYPE, UNIFORMS_DTYPE

_H = None
SKYBOX_FACES = ("right", "left", "top", "bottom", "front", "back")  # src/main.cpp:2064-2071
RIGHT, LEFT, UP, DOWN, FORWARD, BACKWARD = range(6)  # include/camera.h CameraMovementDirection


def hlib():
    global _H
    if _H is None:
        L = _native.load_host()
        vp = C.c_void_p
        L.rth_scene_load.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int]
        L.rth_scene_load.restype = vp
        L.rth_scene_free.argtypes = [vp]
        for fn, rt in (("rth_scene_n_floats", C.c_uint64), ("rth_scene_n_idx", C.c_uint64), ("rth_scene_n_meshes", C.c_int),
                       ("rth_scene_verts", vp), ("rth_scene_idx", vp), ("rth_scene_ranges", vp),
                       ("rth_scene_n_materials", C.c_int), ("rth_scene_materials", vp), ("rth_scene_n_prim_material", C.c_uint64), ("rth_scene_prim_material", vp),
                       ("rth_scene_orbit_prim_offset", C.c_uint32), ("rth_scene_orbit_vert_offset", C.c_uint32)):
            getattr(L, fn).argtypes = [vp]
            getattr(L, fn).restype = rt
        L.rth_write_armadillo_standin.argtypes = [C.c_char_p, C.c_int]
        L.rth_write_armadillo_limbs.argtypes = [C.c_char_p, C.c_int]
        L.rth_write_armadillo_limbs.restype = C.c_longlong
        L.rth_default_uniforms.argtypes = [vp]
        L.rth_make_instance.argtypes = [vp, C.c_uint32, C.c_uint64, vp]
        L.rth_anim_init.argtypes = [vp]
        L.rth_anim_step.argtypes = [vp, C.c_float]
        L.rth_anim_transforms.argtypes = [vp, vp]
        L.rth_camera_new.argtypes = [C.c_float] * 3
        L.rth_camera_new.restype = vp
        L.rth_camera_free.argtypes = [vp]
        L.rth_camera_move.argtypes = [vp, C.c_int, C.c_float]
        L.rth_camera_mouse.argtypes = [vp, C.c_float, C.c_float]
        L.rth_camera_look.argtypes = [vp, C.c_int]
        L.rth_camera_get.argtypes = [vp, vp]
        L.rth_camera_to_uniforms.argtypes = [vp, vp]
        if hasattr(L, "rth_decode_jpeg"):
            L.rth_decode_jpeg.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
            L.rth_free.argtypes = [vp]
        _H = L
    return _H


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class SceneGeometry:
    """Result of rthost::loadScene — the two shared buffers and per-object ranges."""

    def __init__(self, paths):
        L = hlib()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        err = C.create_string_buffer(512)
        h = L.rth_scene_load(arr, len(paths), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        try:
            nf, ni, nm = L.rth_scene_n_floats(h), L.rth_scene_n_idx(h), L.rth_scene_n_meshes(h)
            self.verts = np.ctypeslib.as_array(C.cast(L.rth_scene_verts(h), C.POINTER(C.c_float)), (nf,)).copy()
            self.idx = np.ctypeslib.as_array(C.cast(L.rth_scene_idx(h), C.POINTER(C.c_uint32)), (ni,)).copy()
            rbuf = C.string_at(L.rth_scene_ranges(h), nm * MESH_RANGE_DTYPE.itemsize)
            r = np.frombuffer(rbuf, MESH_RANGE_DTYPE)
            self.ranges = [(int(x["first_float"]), int(x["first_index"]), int(x["prim_count"])) for x in r]
            # row n4: the MTL materials the loader parsed (entry 0 = the reference's hard-coded surface) and each triangle's material
            nmat, npm = L.rth_scene_n_materials(h), L.rth_scene_n_prim_material(h)
            self.materials = np.frombuffer(C.string_at(L.rth_scene_materials(h), nmat * MATERIAL_DTYPE.itemsize), MATERIAL_DTYPE).copy()
            self.prim_material = np.ctypeslib.as_array(C.cast(L.rth_scene_prim_material(h), C.POINTER(C.c_uint32)), (npm,)).copy() if npm else np.zeros(0, np.uint32)
            self.orbiting_primitive_offset = L.rth_scene_orbit_prim_offset(h)
            self.orbiting_vertex_offset = L.rth_scene_orbit_vert_offset(h)
        finally:
            L.rth_scene_free(h)


LIMBS_RESOLUTION = 306


def _obj_triangle_count(path):
    """the generators write '# ... N triangles' on the first line"""
    with open(path) as fh:
        head = fh.readline()
    try:
        return int(head.strip().split()[-2])
    except Exception:
        return -1


def armadillo_path(resources_dir, cache_dir=None, frequency=132, kind="standin"):
    """resources/armadillo.obj if the user supplied it, else a generated stand-in, named as such:
    kind "standin" = the geodesic blob of host/standin.cpp (star-shaped: easy for a BVH),
    kind "limbs"   = the implicit figure of host/standin_limbs.cpp (limbs, claws, concavities: hard)."""
    real = os.path.join(resources_dir, "armadillo.obj")
    if os.path.exists(real):
        return real, "armadillo.obj (user supplied)"
    cache_dir = cache_dir or os.path.join(resources_dir, "generated")
    os.makedirs(cache_dir, exist_ok=True)
    if kind == "limbs":
        path = os.path.join(cache_dir, "armadillo_limbs_r%d.obj" % LIMBS_RESOLUTION)
    elif kind == "standin":
        path = os.path.join(cache_dir, "armadillo_standin_f%d.obj" % frequency)
    else:
        raise ValueError("unknown stand-in kind " + kind)
    mtl = os.path.join(resources_dir, "armadillo.mtl")
    if os.path.exists(mtl) and not os.path.exists(os.path.join(cache_dir, "armadillo.mtl")):
        import shutil
        tmp_mtl = os.path.join(cache_dir, "armadillo.mtl.tmp%d" % os.getpid())   # several ranks may start at once
        shutil.copyfile(mtl, tmp_mtl)
        os.replace(tmp_mtl, os.path.join(cache_dir, "armadillo.mtl"))
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        if kind == "limbs":
            if hlib().rth_write_armadillo_limbs(os.fsencode(tmp), LIMBS_RESOLUTION) < 0:
                raise RuntimeError("limbs stand-in generation failed")
        elif hlib().rth_write_armadillo_standin(os.fsencode(tmp), frequency) != 0:
            raise RuntimeError("stand-in generation failed")
        os.replace(tmp, path)
    if kind == "limbs":
        return path, "armadillo LIMBS STAND-IN (implicit figure, surface nets r=%d, %d triangles)" % (LIMBS_RESOLUTION, _obj_triangle_count(path))
    return path, "armadillo STAND-IN (geodesic f=%d, %d triangles)" % (frequency, 20 * frequency * frequency)


def default_uniforms(**over):
    u = np.zeros(1, UNIFORMS_DTYPE)
    hlib().rth_default_uniforms(_p(u))
    for k, v in over.items():
        u[0][k] = v
    return u


def make_instance(transform12, obj_index, mesh):
    out = np.zeros(1, INSTANCE_DTYPE)
    t = np.ascontiguousarray(transform12, np.float32).reshape(12)
    hlib().rth_make_instance(_p(t), obj_index, mesh, _p(out))
    return out[0]


class SceneAnimation:
    """rthost::SceneAnimation: M0 = I, M1 = T(0,0,5) at start; animate(timeParam) per frame."""

    def __init__(self):
        self.state = np.zeros(32, np.float32)
        hlib().rth_anim_init(_p(self.state))

    def animate(self, time_param):
        hlib().rth_anim_step(_p(self.state), float(time_param))

    def transforms(self):
        out = np.zeros(24, np.float32)
        hlib().rth_anim_transforms(_p(self.state), _p(out))
        return out.reshape(2, 12)

    def instances(self, meshes=(0, 1)):
        t = self.transforms()
        inst = np.zeros(2, INSTANCE_DTYPE)
        for i in range(2):
            inst[i] = make_instance(t[i], i, meshes[i])
        return inst


class Camera:
    def __init__(self, position=(0.0, 0.0, 20.0)):
        self.h = C.c_void_p(hlib().rth_camera_new(*[float(x) for x in position]))

    def __del__(self):
        try:
            hlib().rth_camera_free(self.h)
        except Exception:
            pass

    def move(self, direction, distance):
        hlib().rth_camera_move(self.h, direction, float(distance))

    def process_mouse_movement(self, xoff, yoff):
        hlib().rth_camera_mouse(self.h, float(xoff), float(yoff))

    def look(self, direction):
        hlib().rth_camera_look(self.h, direction)

    def vectors(self):
        out = np.zeros(12, np.float32)
        hlib().rth_camera_get(self.h, _p(out))
        o = out.reshape(4, 3)
        return {"position": o[0], "front": o[1], "up": o[2], "right": o[3]}

    def to_uniforms(self, uniforms):
        hlib().rth_camera_to_uniforms(self.h, _p(uniforms))
        return uniforms


def decode_jpeg(path):
    """RGBA8 (h, w, 4) through the product's own decoder (csrc: host/jpeg_decode.cpp)."""
    L = hlib()
    if not hasattr(L, "rth_decode_jpeg"):
        raise RuntimeError("librt_host.so was built without the JPEG decoder")
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    if L.rth_decode_jpeg(os.fsencode(path), C.byref(ptr), C.byref(w), C.byref(h), err, 256) != 0:
        raise RuntimeError("decode_jpeg(%s): %s" % (path, err.value.decode()))
    try:
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    finally:
        L.rth_free(ptr)


def load_skybox(directory):
    """Six faces in the reference's order (src/main.cpp:2064-2071), each (h, w, 4) uint8."""
    return [decode_jpeg(os.path.join(directory, f + ".jpg")) for f in SKYBOX_FACES]
