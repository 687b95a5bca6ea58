#!/usr/bin/env python3
"""Stress of k_tail's grid barrier: several contexts in flight, every frame forced through k_tail with MANY secondary
rays (mirror armadillo + glass teapot, 9 bounces), images compared with the per-bounce launches."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (imports torch before the library)
from vulkan_raytracing_amd import RtContext, host  # noqa: E402


def main():
    res = os.path.join(ROOT, "resources")
    arm, _ = host.armadillo_path(res)
    geom = host.SceneGeometry([os.path.join(res, "teapot.obj"), arm])
    anim = host.SceneAnimation()
    inst = anim.instances((0, 1))
    u = host.default_uniforms(max_bounce_count=8, samples_per_pixel=2, center_object_type=2, orbiting_object_type=1,
                              orbiting_object_primitive_offset=geom.orbiting_primitive_offset, orbiting_object_vertex_offset=geom.orbiting_vertex_offset)
    sky = host.load_skybox(os.path.join(res, "skybox_texture_sea"))
    W, H = 1920, 1080
    P = int(os.environ.get("P", "6"))
    ctxs = []
    for _ in range(P):
        c = RtContext(0)
        c.upload_geometry(geom.verts, geom.idx, geom.ranges)
        c.set_instances(inst); c.set_uniforms(u); c.set_skybox(sky)
        ctxs.append(c)
    ctxs[0].set_param("tail_kernel", 0)
    ref, st = ctxs[0].trace(W, H)
    print("rays", st.rays_primary, st.rays_secondary, st.rays_shadow, flush=True)
    for mode in (2, 1, 0):
        for c in ctxs:
            c.set_param("tail_kernel", mode)
        t0 = time.perf_counter()
        n = 0
        for rnd in range(6):
            for c in ctxs:
                c.trace_async(W, H)
            for c in ctxs:
                img, s2 = c.trace_wait(copy=False)
                assert np.array_equal(img, ref), "image differs (mode %d)" % mode
                assert (s2.rays_secondary, s2.rays_shadow) == (st.rays_secondary, st.rays_shadow)
                n += 1
        print("tail_kernel %d: %d frames identical, %.3f ms per frame with %d in flight" % (mode, n, (time.perf_counter() - t0) / n * 1e3, P), flush=True)


if __name__ == "__main__":
    main()
