"""Scene set-up shared by the tests: builds the SAME inputs for the oracle (checker) and for the
product's C ABI.  Geometry comes from the product's host code (librt_host.so) — its equality with the
oracle's ingest restatement and the reference loader is tested separately in test_host.py."""
import os

import numpy as np

from oracle import ingest, oracle
from vulkan_raytracing_amd import host
from vulkan_raytracing_amd.api import INSTANCE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "resources")


def synthetic_skybox(size=64, seed=7):
    """Small deterministic cube map with per-face gradients + noise (face-distinguishable)."""
    rng = np.random.default_rng(seed)
    faces = []
    yy, xx = np.mgrid[0:size, 0:size]
    for f in range(6):
        img = np.zeros((size, size, 4), np.uint8)
        img[..., 0] = (xx * 255 // (size - 1) + 37 * f) % 256
        img[..., 1] = (yy * 255 // (size - 1) + 91 * f) % 256
        img[..., 2] = rng.integers(0, 256, (size, size))
        img[..., 3] = 255
        faces.append(img)
    return faces


class ScenePair:
    """Oracle scene + (optionally) product context with identical inputs."""

    def __init__(self, obj_paths, instances, uniforms, sky=None, ctx=None):
        self.geom_paths = list(obj_paths)
        self.geom = host.SceneGeometry(obj_paths)
        self.instances = np.ascontiguousarray(instances, INSTANCE_DTYPE)
        self.uniforms = uniforms
        self.sky = sky
        self.orc = oracle.OracleScene()
        self.orc.set_geometry(self.geom.verts, self.geom.idx, self.geom.ranges)
        self.orc.set_instances([self.instances[i].tobytes() for i in range(len(self.instances))])
        self.orc.set_uniforms(uniforms.tobytes())
        if sky is not None:
            self.orc.set_skybox(sky)
        self.ctx = ctx
        if ctx is not None:
            ctx.upload_geometry(self.geom.verts, self.geom.idx, self.geom.ranges)
            ctx.set_instances(self.instances)
            ctx.set_uniforms(uniforms)
            if sky is not None:
                ctx.set_skybox(sky)

    def set_uniforms(self, uniforms):
        self.uniforms = uniforms
        self.orc.set_uniforms(uniforms.tobytes())
        if self.ctx is not None:
            self.ctx.set_uniforms(uniforms)

    def set_materials(self, table, prim_material=None):
        """row n4: the same MTL material table for the oracle and the product (None removes it)"""
        self.orc.set_materials(table, prim_material)
        if self.ctx is not None:
            self.ctx.set_materials(table, prim_material)

    def set_instance_types(self, types):
        self.orc.set_instance_types(types)
        if self.ctx is not None:
            self.ctx.set_instance_types(types)

    def set_instances(self, instances, update=False):
        self.instances = np.ascontiguousarray(instances, INSTANCE_DTYPE)
        self.orc.set_instances([self.instances[i].tobytes() for i in range(len(self.instances))])
        if self.ctx is not None:
            self.ctx.set_instances(self.instances, update=update)


def two_object_scene(center, orbiting, center_type, orbit_type, max_bounce, spp, sky=None, ctx=None, time_param=None):
    """The reference's scene: center mesh at M0, orbiting mesh at M1 (src/main.cpp:1805-1808)."""
    paths = [center, orbiting]
    anim = host.SceneAnimation()
    if time_param is not None:
        anim.animate(time_param)
    inst = anim.instances((0, 1))
    geom_probe = host.SceneGeometry(paths)
    u = host.default_uniforms(max_bounce_count=max_bounce, samples_per_pixel=spp, center_object_type=center_type,
                              orbiting_object_type=orbit_type,
                              orbiting_object_primitive_offset=geom_probe.orbiting_primitive_offset,
                              orbiting_object_vertex_offset=geom_probe.orbiting_vertex_offset)
    return ScenePair(paths, inst, u, sky=sky, ctx=ctx)


def ring_scene(mesh_path, n_inst, radius, max_bounce, spp, sky=None, ctx=None, center_path=None):
    """cfg5: n instances of one BLAS on a ring about the origin (generalises M1), all customIndex 1;
    optional center mesh as instance 0 / mesh 0."""
    paths = ([center_path] if center_path else []) + [mesh_path]
    ring_mesh = len(paths) - 1
    inst = []
    if center_path:
        inst.append(host.make_instance(ingest.glm_to_vulkan(ingest.mat_identity()), 0, 0))
    for k in range(n_inst):
        ang = 2.0 * np.pi * k / n_inst
        m = ingest.mat_translate(ingest.mat_rotate_y(ingest.mat_identity(), np.float32(ang)), (0, 0, radius))
        inst.append(host.make_instance(ingest.glm_to_vulkan(m), 1, ring_mesh))
    geom_probe = host.SceneGeometry(paths)
    u = host.default_uniforms(max_bounce_count=max_bounce, samples_per_pixel=spp, center_object_type=1, orbiting_object_type=0,
                              orbiting_object_primitive_offset=geom_probe.orbiting_primitive_offset,
                              orbiting_object_vertex_offset=geom_probe.orbiting_vertex_offset)
    return ScenePair(paths, np.asarray(inst, INSTANCE_DTYPE), u, sky=sky, ctx=ctx)


def random_rays(n, seed, origin_radius=20.0, target_radius=4.0, tmin=0.001, tmax=10000.0):
    """Rays from a sphere of origins towards a ball around the scene centre (most of them hit)."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True); o *= origin_radius * rng.uniform(0.3, 1.0, (n, 1))
    t = rng.normal(size=(n, 3)); t /= np.linalg.norm(t, axis=1, keepdims=True); t *= target_radius * rng.uniform(0, 1, (n, 1)) ** (1 / 3)
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o; rays[:, 3] = tmin; rays[:, 4:7] = d; rays[:, 7] = tmax
    return rays


def grazing_rays(n, seed):
    """Rays for two_object_scene(teapot, cube) that run almost parallel to the xz plane: origins on a far ring, aimed at points
    ON the top / bottom faces of the orbiting cube (centre (0,0,5), half size 1) and into the teapot's height range, with
    the direction's y component between 1e-2 and 1e-7 of its length."""
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 8), np.float32)
    ang = rng.uniform(0, 2 * np.pi, n)
    o = np.stack([20 * np.cos(ang), np.zeros(n), 20 * np.sin(ang)], 1)
    face = rng.integers(0, 3, n)
    tgt = np.stack([rng.uniform(-1, 1, n), rng.choice([-1.0, 1.0], n), 5 + rng.uniform(-1, 1, n)], 1)
    tgt[face == 1] = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(0.0, 1.6, n), rng.uniform(-1.5, 1.5, n)], 1)[face == 1]
    o[:, 1] = tgt[:, 1] + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -2, n) * 20.0
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 0:3] = o; rays[:, 3] = 0.001; rays[:, 4:7] = d; rays[:, 7] = 10000.0
    return rays


# ---- fixtures produced by interpreting the reference's own SPIR-V shaders (tests/golden/make_spirv_fixtures.py) ---------------
class SpirvFixtureScene:
    def __init__(self, meta, bounces, pixels):
        self.meta, self.bounces, self.pixels = meta, bounces, pixels
        self.name, self.width, self.height = meta["name"], meta["width"], meta["height"]
        self.paths = [os.path.join(ROOT, p) for p in meta["paths"]]
        self.instances = np.frombuffer(bytes.fromhex("".join(meta["instances"])), INSTANCE_DTYPE).copy()
        from vulkan_raytracing_amd.api import UNIFORMS_DTYPE
        self.uniforms = np.frombuffer(bytes.fromhex(meta["uniforms"]), UNIFORMS_DTYPE).copy()
        self.sky_dir = os.path.join(ROOT, meta["sky"]) if meta["sky"] else None
        self._geom = None

    @property
    def geometry(self):
        if self._geom is None:
            self._geom = host.SceneGeometry(self.paths)
        return self._geom

    def apply(self, target):
        """target: RtContext or the tests' oracle adapter (upload_geometry / set_instances / set_uniforms / set_skybox)"""
        g = self.geometry
        target.upload_geometry(g.verts, g.idx, g.ranges)
        target.set_instances(self.instances)
        target.set_uniforms(self.uniforms)
        if self.sky_dir:
            target.set_skybox(host.load_skybox(self.sky_dir))

    def oracle_scene(self):
        S = oracle.OracleScene()
        g = self.geometry
        S.set_geometry(g.verts, g.idx, g.ranges)
        S.set_instances([self.instances[i].tobytes() for i in range(len(self.instances))])
        S.set_uniforms(self.uniforms.tobytes())
        if self.sky_dir:
            S.set_skybox(host.load_skybox(self.sky_dir))
        return S


def load_spirv_fixtures():
    import json
    gold = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gold, "spirv_fixtures.json")))
    data = np.load(os.path.join(gold, "spirv_fixtures.npz"), allow_pickle=False)
    for kind in ("standin", "limbs"):      # the generated meshes the cfg3 / cfg5 scenes name (deterministic generators)
        host.armadillo_path(RES, kind=kind)
    b, p = data["bounces"], data["pixels"]
    return [SpirvFixtureScene(m, b[b["scene"] == m["id"]], p[p["scene"] == m["id"]]) for m in meta["scenes"]]
