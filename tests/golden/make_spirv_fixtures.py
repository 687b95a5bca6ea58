#!/usr/bin/env python3
"""Golden vectors for the shading path, produced by EXECUTING THE REFERENCE'S OWN COMPILED SHADERS.

The reference ships shaders/shader.rgen.spv, shader.rchit.spv, shader.rmiss.spv and shader_shadow.rmiss.spv (glslang output
of src/shader.*).  They are the only executable artefact of its shading code, and they are read here strictly as data:
oracle/spirv_interp.py interprets their instruction streams, one ray-generation invocation (= one pixel) at a time.
What the Vulkan driver does for the reference is bound from outside the modules:

    OpTraceRayKHR              -> the oracle's two-level closest-hit / any-hit traversal (orc_intersect); on a hit the
                                  reference's rchit module is interpreted with gl_PrimitiveID, gl_InstanceCustomIndexEXT,
                                  hitAttributeEXT, gl_ObjectToWorldEXT, gl_WorldToObjectEXT set from the hit record; on a
                                  miss the reference's miss module [missIndex] is interpreted
    OpImageSampleExplicitLod   -> the oracle's cube sampler (orc_sample_sky)
    OpImageWrite               -> recorded

Everything else — the sample loop, the jitter hash, the primary ray, the bounce loop, the material switch, Blinn-Phong,
reflect / refract / TIR, the accumulation — runs from the reference's binaries.  Per traced ray the script records the ray,
the hit record, the payload the rchit module produced, the shadow ray, the state after the bounce, and per pixel the value
written to the image.  tests/test_oracle.py replays the records through the oracle (CPU), tests/test_gpu_parity.py
compares HIP frames with the recorded pixels.

Run in the authoring container (needs /root/reference, does not travel to the GPU box):
    python tests/golden/make_spirv_fixtures.py
writes tests/golden/spirv_fixtures.npz (+ .json with the scene list).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle, spirv_interp as SI  # noqa: E402
from vulkan_raytracing_amd import host, workloads  # noqa: E402

REF = os.environ.get("RT_REFERENCE", "/root/reference")
RES = os.path.join(ROOT, "resources")
F32 = np.float32

BOUNCE_DTYPE = np.dtype([
    ("scene", np.int32), ("px", np.int32), ("py", np.int32), ("sample", np.int32), ("bounce", np.int32),
    ("o", np.float32, 3), ("d", np.float32, 3),                       # closest-hit traceRayEXT arguments (tmin 0.001, tmax 10000)
    ("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.int32), ("inst", np.int32),   # the driver's answer
    ("P", np.float32, 3), ("N", np.float32, 3), ("object_index", np.int32),                            # payload after rchit / rmiss
    ("shadow", np.int32),                                             # 1: a shadow ray was traced at this hit
    ("so", np.float32, 3), ("sl", np.float32, 3), ("stmax", np.float32), ("occluded", np.int32),
    ("last", np.int32),                                               # 1: the bounce loop ended after this trace
    ("no", np.float32, 3), ("nd", np.float32, 3),                     # rayOrigin / rayDirection after the bounce
    ("color", np.float32, 3)])                                        # tmpColor after the bounce
PIXEL_DTYPE = np.dtype([("scene", np.int32), ("px", np.int32), ("py", np.int32), ("rgba", np.float32, 4)])


def load_modules():
    out = {}
    for key, f in (("rgen", "shader.rgen.spv"), ("rchit", "shader.rchit.spv"), ("rmiss", "shader.rmiss.spv"), ("shadow_miss", "shader_shadow.rmiss.spv")):
        out[key] = SI.Module(open(os.path.join(REF, "shaders", f), "rb").read())
    return out


class Env:
    """Everything outside the SPIR-V modules: descriptor set 0 (src/main.cpp:1305-1335), built-ins, the driver."""

    def __init__(self, mods, wl, W, H):
        self.mods, self.W, self.H = mods, W, H
        g = wl.geometry
        self.verts, self.idx = np.ascontiguousarray(g.verts, np.float32), np.ascontiguousarray(g.idx, np.uint32)
        self.inst = wl.instances
        self.orc = oracle.OracleScene()
        self.orc.set_geometry(g.verts, g.idx, g.ranges)
        self.orc.set_instances([self.inst[i].tobytes() for i in range(len(self.inst))])
        self.orc.set_uniforms(wl.uniforms.tobytes())
        if wl.sky is not None:
            self.orc.set_skybox(wl.sky)
        u = wl.uniforms[0]
        self.ubo = [[F32(x) for x in u["position"]], [F32(x) for x in u["right"]], [F32(x) for x in u["up"]], [F32(x) for x in u["forward"]],
                    [F32(x) for x in u["light_position"]], F32(u["light_intensity"]), int(u["max_bounce_count"]), int(u["samples_per_pixel"]),
                    int(u["center_object_type"]), int(u["orbiting_object_type"]), int(u["orbiting_object_primitive_offset"]), int(u["orbiting_object_vertex_offset"])]
        L = oracle.lib()
        self.o2w, self.w2o = [], []
        for i in range(len(self.inst)):
            m = np.ascontiguousarray(self.inst[i]["transform"], np.float32)
            w = np.zeros(12, np.float32)
            L.orc_invert_affine(m.ctypes.data, w.ctypes.data)       # gl_WorldToObjectEXT = inverse(gl_ObjectToWorldEXT): the driver's job
            # mat4x3 = 4 columns of 3 rows; the row-major 3x4 record holds element (r, c) at 4r + c
            self.o2w.append([[F32(m[4 * r + c]) for r in range(3)] for c in range(4)])
            self.w2o.append([[F32(w[4 * r + c]) for r in range(3)] for c in range(4)])
        self.px = self.py = 0
        self.hit = None
        self.events = []
        self.image = None
        self.watch_ids = {}
        m = mods["rgen"]
        for vid, nm in m.names.items():
            if nm in ("tmpColor", "rayOrigin", "rayDirection", "i", "j", "color"):
                self.watch_ids[vid] = nm

    # -- resources ---------------------------------------------------------------------------------------------------------
    def resource(self, module, vid, storage, binding, builtin, location):
        if builtin == "LaunchIdKHR":
            return [self.px, self.py, 0]
        if builtin == "LaunchSizeKHR":
            return [self.W, self.H, 1]
        if builtin == "InstanceCustomIndexKHR":
            return int(self.inst[self.hit["inst"]]["custom_index_and_mask"]) & 0xFFFFFF
        if builtin == "PrimitiveId":
            return int(self.hit["prim"])
        if builtin == "ObjectToWorldKHR":
            return self.o2w[self.hit["inst"]]
        if builtin == "WorldToObjectKHR":
            return self.w2o[self.hit["inst"]]
        if storage == "HitAttributeKHR":
            return [F32(self.hit["u"]), F32(self.hit["v"])]
        if binding == 1:
            return self.ubo
        if binding == 2:
            return [self.idx]
        if binding == 3:
            return [self.verts]
        if binding in (0, 4, 5):
            return ("binding", binding)
        return None

    # -- the driver ----------------------------------------------------------------------------------------------------------
    def trace_ray(self, inv, flags, cull_mask, sbt_offset, sbt_stride, miss_index, origin, tmin, direction, tmax, payload_ptr):
        assert cull_mask == 0xFF and sbt_offset == 0 and sbt_stride == 0
        terminate_first, skip_chit = bool(flags & 0x4), bool(flags & 0x8)
        ray = np.array([[origin[0], origin[1], origin[2], tmin, direction[0], direction[1], direction[2], tmax]], np.float32)
        h = self.orc.intersect(ray, any_hit=terminate_first, use_bvh=True)[0]
        hit = h["inst"] >= 0
        self.events.append(("trace", dict(flags=flags, miss_index=miss_index, ray=ray[0].copy(), hit=h.copy())))
        if hit and not skip_chit:
            self.hit = h
            shared = {vid: payload_ptr.cell for vid, (_, sc) in self.mods["rchit"].globals.items() if SI.STORAGE.get(sc) == "IncomingRayPayloadKHR"}
            SI.Invocation(self.mods["rchit"], self, shared).run()
        elif not hit:
            mod = self.mods["rmiss"] if miss_index == 0 else self.mods["shadow_miss"]
            shared = {vid: payload_ptr.cell for vid, (_, sc) in mod.globals.items() if SI.STORAGE.get(sc) == "IncomingRayPayloadKHR"}
            SI.Invocation(mod, self, shared).run()
        v = payload_ptr.cell.value
        self.events.append(("payload", [list(x) if isinstance(x, list) else x for x in v] if isinstance(v, list) else v))

    def sample(self, inv, image, coord, lod):
        assert image == ("binding", 5) and float(lod) == 0.0
        c = self.orc.sample_sky(np.array([coord[0], coord[1], coord[2]], np.float32))
        return [F32(c[0]), F32(c[1]), F32(c[2]), F32(1.0)]

    def image_write(self, inv, image, coord, texel):
        assert image == ("binding", 4) and (int(coord[0]), int(coord[1])) == (self.px, self.py)
        self.image = [float(x) for x in texel]

    def on_store(self, name, value):
        self.events.append(("store", name, list(value) if isinstance(value, list) else value))

    # -- one pixel -> records --------------------------------------------------------------------------------------------------
    def run_pixel(self, scene_id, px, py, bounces, pixels):
        self.px, self.py, self.events, self.image = px, py, [], None
        inv = SI.Invocation(self.mods["rgen"], self)
        inv.watch = self.watch_ids
        inv.run()
        state = dict(tmpColor=[0, 0, 0], rayOrigin=[0, 0, 0], rayDirection=[0, 0, 1], i=0, j=0)
        cur = None      # record of the bounce in progress
        ev = self.events
        k = 0

        def close(last):
            cur["last"] = last
            cur["no"], cur["nd"], cur["color"] = state["rayOrigin"], state["rayDirection"], state["tmpColor"]
            bounces.append(cur.copy())

        rec = None
        while k < len(ev):
            e = ev[k]
            if e[0] == "store":
                if e[1] == "color" and rec is not None:          # color += vec4(tmpColor, 1): the sample is complete
                    cur = rec
                    close(1)
                    rec = None
                elif e[1] in state:
                    state[e[1]] = e[2]
            elif e[0] == "trace":
                t, pay = e[1], ev[k + 1][1]
                k += 1
                if t["miss_index"] == 0:                         # closest-hit trace of the bounce loop
                    if rec is not None:
                        cur = rec
                        close(0)
                    rec = np.zeros((), BOUNCE_DTYPE)
                    rec["scene"], rec["px"], rec["py"], rec["sample"], rec["bounce"] = scene_id, px, py, state["i"], state["j"]
                    rec["o"], rec["d"] = t["ray"][0:3], t["ray"][4:7]
                    assert t["ray"][3] == F32(0.001) and t["ray"][7] == F32(10000.0) and t["flags"] == 1
                    h = t["hit"]
                    rec["t"], rec["u"], rec["v"], rec["prim"], rec["inst"] = h["t"], h["u"], h["v"], h["prim"], h["inst"]
                    rec["P"], rec["N"], rec["object_index"] = pay[0], pay[1], pay[2]
                else:                                            # shadow ray (flags 13, miss index 1)
                    assert t["flags"] == 13 and t["ray"][3] == F32(0.001)
                    rec["shadow"] = 1
                    rec["so"], rec["sl"], rec["stmax"] = t["ray"][0:3], t["ray"][4:7], t["ray"][7]
                    rec["occluded"] = 1 if pay else 0           # isShadow stays true unless the shadow miss shader ran
            k += 1
        assert rec is None and self.image is not None
        p = np.zeros((), PIXEL_DTYPE)
        p["scene"], p["px"], p["py"], p["rgba"] = scene_id, px, py, self.image
        pixels.append(p)


def scene_list():
    """(name, workload factory kwargs, overrides) — the BASELINE configurations plus material mixes that exercise every branch
    of src/shader.rgen:96-165."""
    def cfg2_variant(ct, ot, mb, tparam):
        wl = workloads.make("cfg2", RES)
        anim = host.SceneAnimation()
        anim.animate(tparam)
        wl.instances = anim.instances((0, 1))
        wl.uniforms[0]["center_object_type"], wl.uniforms[0]["orbiting_object_type"], wl.uniforms[0]["max_bounce_count"] = ct, ot, mb
        return wl
    return [
        ("cfg1 cube_scene 256x256 depth 1 spp 1", lambda: workloads.make("cfg1", RES), 256, 256, 600),
        ("cfg2 teapot mirror + cube diffuse, skybox_texture_test, 1280x720 depth 2 spp 4", lambda: workloads.make("cfg2", RES), 1280, 720, 500),
        ("cfg2 scene, teapot refractive + cube mirror, depth 6, timeParam 0.35", lambda: cfg2_variant(2, 1, 5, 0.35), 1280, 720, 500),
        ("cfg2 scene, teapot diffuse + cube refractive, depth 4, timeParam 0.8", lambda: cfg2_variant(0, 2, 3, 0.8), 1280, 720, 400),
        ("cfg3 teapot mirror + armadillo stand-in (geodesic) diffuse, skybox_texture_sea, 1920x1080 depth 4 spp 4", lambda: workloads.make("cfg3", RES, mesh="standin"), 1920, 1080, 400),
        ("cfg3 with the limbs stand-in", lambda: workloads.make("cfg3", RES, mesh="limbs"), 1920, 1080, 400),
        ("cfg5 16 instances of the limbs stand-in on a ring + mirror teapot, raised camera, 1920x1080 depth 4 spp 4", lambda: workloads.make("cfg5", RES, mesh="limbs"), 1920, 1080, 400),
    ]


def choose_pixels(env, W, H, n, seed):
    """70 % of the pixels where geometry is (found with a coarse oracle pre-pass of primary rays), 30 % anywhere."""
    rng = np.random.default_rng(seed)
    step = max(4, W // 160)
    xs, ys = np.meshgrid(np.arange(step // 2, W, step), np.arange(step // 2, H, step))
    rays = np.zeros((xs.size, 8), np.float32)
    for k, (x, y) in enumerate(zip(xs.ravel(), ys.ravel())):
        od = env.orc.primary_ray(int(x), int(y), W, H, 0)
        rays[k] = (od[0], od[1], od[2], 0.001, od[3], od[4], od[5], 10000.0)
    hit = env.orc.intersect(rays)["inst"] >= 0
    cand = np.stack([xs.ravel()[hit], ys.ravel()[hit]], axis=1)
    n_obj = min(len(cand), int(0.7 * n))
    chosen = set()
    if n_obj:
        for x, y in cand[rng.choice(len(cand), n_obj, replace=False)]:
            chosen.add((int(min(W - 1, max(0, x + rng.integers(-step // 2, step // 2 + 1)))), int(min(H - 1, max(0, y + rng.integers(-step // 2, step // 2 + 1))))))
    while len(chosen) < n:
        chosen.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
    chosen.update([(0, 0), (W - 1, H - 1)])
    return sorted(chosen)


def main():
    mods = load_modules()
    bounces, pixels, meta = [], [], []
    t0 = time.time()
    for sid, (name, make, W, H, n) in enumerate(scene_list()):
        wl = make()
        env = Env(mods, wl, W, H)
        px = choose_pixels(env, W, H, n, seed=100 + sid)
        nb0 = len(bounces)
        for (x, y) in px:
            env.run_pixel(sid, x, y, bounces, pixels)
        meta.append(dict(id=sid, name=name, width=W, height=H, pixels=len(px), bounce_records=len(bounces) - nb0,
                         paths=[os.path.relpath(p, ROOT) for p in wl.paths], sky=os.path.relpath(wl.sky_dir, ROOT) if wl.sky_dir else None,
                         instances=[wl.instances[i].tobytes().hex() for i in range(len(wl.instances))], uniforms=wl.uniforms.tobytes().hex()))
        print("%-100s %4d pixels %6d bounce records  %.0f s" % (name[:100], len(px), len(bounces) - nb0, time.time() - t0), flush=True)
    b = np.array(bounces, BOUNCE_DTYPE)
    p = np.array(pixels, PIXEL_DTYPE)
    np.savez_compressed(os.path.join(HERE, "spirv_fixtures.npz"), bounces=b, pixels=p)
    with open(os.path.join(HERE, "spirv_fixtures.json"), "w") as fh:
        json.dump(dict(generator="tests/golden/make_spirv_fixtures.py",
                       source="interpreted /root/reference/shaders/{shader.rgen,shader.rchit,shader.rmiss,shader_shadow.rmiss}.spv (read as data)",
                       arithmetic="binary32 +,-,*,/ and OpDot literally (one rounding per operation, left to right, no contraction); GLSL.std.450 instructions "
                                  "in binary64 rounded once; traversal / inverse transform / cube sampling bound to the oracle",
                       scenes=meta), fh, indent=1)
    print("wrote %d bounce records, %d pixels" % (len(b), len(p)))


if __name__ == "__main__":
    sys.exit(main())
