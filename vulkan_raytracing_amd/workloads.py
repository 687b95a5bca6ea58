"""The BASELINE.json configurations as scene descriptions, shared by bench.py and the tests so that the
workload a number is quoted on and the workload the parity suite checks are the same object.

A workload is host-side data only: OBJ paths, instance records (64-byte mirrors of VkAccelerationStructureInstanceKHR,
src/main.cpp:538-551), the 104-byte uniform block (src/main.cpp:1847-1873) and the skybox directory.  `apply()` pushes it
through the C ABI (RtContext) and/or into a checker object with the same four setters (the tests' oracle scene)."""
import os

import numpy as np

from . import host
from .api import INSTANCE_DTYPE

IDENTITY12 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def ring_instances(n_inst, radius, ring_mesh=1, center_mesh=0, phase=0.0, center_transform=None):
    """cfg5: n instances of the orbiting mesh's BLAS on a ring about the origin (generalises M1 = T(0,0,5) of
    src/main.cpp:1805-1808), all with customIndex 1 (src/shader.rchit:52 selects the orbiting mesh's buffer range with
    it), plus the center mesh as instance 0 / customIndex 0."""
    inst = np.zeros(n_inst + 1, INSTANCE_DTYPE)
    inst[0] = host.make_instance(IDENTITY12 if center_transform is None else center_transform, 0, center_mesh)
    for k in range(n_inst):
        a = 2.0 * np.pi * k / n_inst + phase
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        # R_y(a) * T(0,0,radius): rotation, then the translated offset
        t = np.array([c, 0, s, s * radius, 0, 1, 0, 0, -s, 0, c, c * radius], np.float32)
        inst[k + 1] = host.make_instance(t, 1, ring_mesh)
    return inst


def raised_camera(uniforms, position=(0.0, 20.0, 28.0), pitch=-0.62):
    """cfg5 camera: the reference's fly camera (src/camera.cpp:16-25, 91-106) moved up and pitched down so that it looks
    over the near ring members at the mirror teapot in the middle — with the start-up camera (0,0,20) the instance at
    ring angle 0 hides the teapot and no bounce ray is ever traced."""
    cam = host.Camera(position)
    cam.process_mouse_movement(0.0, pitch)
    return cam.to_uniforms(uniforms)


class Workload:
    def __init__(self, name, paths, instances, uniforms, sky_dir, width, height, mesh_label, note=""):
        self.name, self.paths, self.instances, self.uniforms = name, list(paths), instances, uniforms
        self.sky_dir, self.width, self.height, self.mesh_label, self.note = sky_dir, width, height, mesh_label, note
        self._geom = None
        self._sky = None
        self._anim = None

    @property
    def geometry(self):
        if self._geom is None:
            self._geom = host.SceneGeometry(self.paths)
        return self._geom

    @property
    def sky(self):
        if self._sky is None and self.sky_dir:
            self._sky = host.load_skybox(self.sky_dir)
        return self._sky

    def animate(self, time_param):
        """Instance records of the next animated frame (src/main.cpp:2836-2851): M0 accumulates a tiny spin, the orbiting
        mesh circles the centre at angle pi * timeParam — for cfg5 the whole ring does.  Same instance count as
        `instances`, so rt_set_instances(update=1) applies."""
        if self._anim is None:
            self._anim = host.SceneAnimation()
        self._anim.animate(float(time_param))
        if len(self.instances) == 2:
            return self._anim.instances((0, 1))
        if len(self.instances) == 1:
            t = self._anim.transforms()
            inst = self.instances.copy()
            inst[0] = host.make_instance(t[0], 0, 0)
            return inst
        t = self._anim.transforms()
        return ring_instances(len(self.instances) - 1, 10.0, phase=float(np.pi * np.float32(time_param)), center_transform=t[0])

    def apply(self, target, sky=None):
        """target: RtContext, or any object with upload_geometry/set_instances/set_uniforms/set_skybox."""
        g = self.geometry
        target.upload_geometry(g.verts, g.idx, g.ranges)
        target.set_instances(self.instances)
        target.set_uniforms(self.uniforms)
        faces = sky if sky is not None else self.sky
        if faces is not None:
            target.set_skybox(faces)

    def describe(self):
        u = self.uniforms[0]
        return "BASELINE %s: %s, %s, %dx%d, maxBounceCount %d (depth %d) + shadow rays, spp %d%s" % (
            self.name, self.mesh_label, os.path.basename(self.sky_dir) if self.sky_dir else "no skybox", self.width, self.height,
            int(u["max_bounce_count"]), int(u["max_bounce_count"]) + 1, int(u["samples_per_pixel"]), self.note)


def _two_object_uniforms(paths, max_bounce, spp, center_type, orbit_type):
    g = host.SceneGeometry(paths) if len(paths) > 1 else None
    return host.default_uniforms(max_bounce_count=max_bounce, samples_per_pixel=spp, center_object_type=center_type, orbiting_object_type=orbit_type,
                                 orbiting_object_primitive_offset=g.orbiting_primitive_offset if g else 0,
                                 orbiting_object_vertex_offset=g.orbiting_vertex_offset if g else 0), g


def make(name, resources, mesh="standin"):
    """name: cfg1..cfg5 (BASELINE.json configs in order).  mesh selects the armadillo: "standin" (geodesic blob, the
    round-1 mesh), "limbs" (the non-star-shaped stand-in) — ignored when resources/armadillo.obj exists."""
    res = resources
    teapot, cube, cube_scene = (os.path.join(res, f) for f in ("teapot.obj", "cube.obj", "cube_scene.obj"))
    if name == "cfg1":
        u = host.default_uniforms(max_bounce_count=0, samples_per_pixel=1, center_object_type=0, orbiting_object_type=0)
        inst = np.zeros(1, INSTANCE_DTYPE)
        inst[0] = host.make_instance(IDENTITY12, 0, 0)
        return Workload(name, [cube_scene], inst, u, None, 256, 256, "cube_scene.obj diffuse")
    if name == "cfg2":
        paths = [teapot, cube]
        u, g = _two_object_uniforms(paths, 1, 4, 1, 0)
        w = Workload(name, paths, host.SceneAnimation().instances((0, 1)), u, os.path.join(res, "skybox_texture_test"), 1280, 720,
                     "teapot.obj mirror + cube.obj diffuse")
        w._geom = g
        return w
    arm, arm_label = host.armadillo_path(res, kind=mesh)
    paths = [teapot, arm]
    sea = os.path.join(res, "skybox_texture_sea")
    if name in ("cfg3", "cfg4"):
        W, H, mb = (1920, 1080, 3) if name == "cfg3" else (3840, 2160, 5)
        u, g = _two_object_uniforms(paths, mb, 4, 1, 0)
        w = Workload(name, paths, host.SceneAnimation().instances((0, 1)), u, sea, W, H, "teapot.obj mirror + %s diffuse" % arm_label)
        w._geom = g
        return w
    if name == "cfg5":
        u, g = _two_object_uniforms(paths, 3, 4, 1, 0)
        raised_camera(u)
        w = Workload(name, paths, ring_instances(16, 10.0), u, sea, 1920, 1080, "teapot.obj mirror + %s diffuse x16 instances on a ring (one BLAS, two-level BVH)" % arm_label,
                     note=", camera raised to (0,20,28) pitch -0.62 rad so the mirror teapot is in view")
        w._geom = g
        return w
    raise ValueError("unknown workload " + name)
