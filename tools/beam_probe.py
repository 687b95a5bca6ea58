"""Pixel beams (csrc/kernels_beam.inc) against one walk per ray on a workload: node visits and triangle tests of the closest-hit kernels
(counting build), identity of the frames, lone-frame time by kernel category.  Usage: python3 tools/beam_probe.py [workload] [mesh]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkan_raytracing_amd import RtContext, workloads

RES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "resources")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh = sys.argv[2] if len(sys.argv) > 2 else "standin"
ctx = RtContext(0)
wl = workloads.make(name, RES, mesh=mesh)
wl.apply(ctx)
imgs = {}
for on in (1, 0, 1, 0):
    ctx.set_param("pixel_beams", on)
    img, st = ctx.trace(wl.width, wl.height, counting=True)
    imgs[on] = img
    print("pixel_beams=%d counting: closest rays %d  node visits %d (%.2f per closest ray)  triangle tests %d (%.2f per closest ray)" % (
        on, st.closest_rays, st.node_visits, st.node_visits / max(1, st.closest_rays), st.tri_tests, st.tri_tests / max(1, st.closest_rays)))
    ctx.set_timing(1)
    for _ in range(3):
        ctx.trace(wl.width, wl.height)
    ms = []
    for _ in range(8):
        _, s2 = ctx.trace(wl.width, wl.height)
        ms.append((s2.ms_frame, s2.ms_raygen, s2.ms_trace_closest, s2.ms_shade, s2.ms_tail, s2.ms_trace_shadow, s2.ms_resolve))
    ctx.set_timing(0)
    print("   lone frame %.3f ms: cover+entry+raygen %.3f  closest %.3f  shade %.3f  tail %.3f  shadow %.3f  resolve %.3f" % tuple(np.median(np.array(ms), axis=0)))
print("identical frames:", bool(np.array_equal(imgs[0], imgs[1])))
