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
for on in (1, 0, 2, 1, 0, 2):      # 2: pixel beams for the primary rays only
    ctx.set_param("pixel_beams", 1 if on else 0)
    ctx.set_param("shadow_beams", 1 if on == 1 else 0)
    img, st = ctx.trace(wl.width, wl.height, counting=True)
    imgs[on] = img
    print("beams=%d counting: closest rays %d  node visits %d (%.2f per closest ray)  triangle tests %d (%.2f)  | shadow rays %d  node visits %.2f per ray  triangle tests %.2f" % (
        on, st.closest_rays, st.node_visits, st.node_visits / max(1, st.closest_rays), st.tri_tests, st.tri_tests / max(1, st.closest_rays),
        st.rays_shadow, st.node_visits_shadow / max(1, st.rays_shadow), st.tri_tests_shadow / max(1, st.rays_shadow)))
    print("   shadow rays settled in k_shade (their outcome cannot change the sample): %d of %d" % (st.rays_shadow_untraced, st.rays_shadow))
    d = list(st.diag)
    print("   interior loop: closest %d wave trips with %.1f lanes busy; shadow %d wave trips with %.1f lanes busy" % (d[0], d[1] / max(1, d[0]), d[3], d[4] / max(1, d[3])))
    td = list(st.tile_diag)
    if on and td[2]:
        print("   k_beam runs: %d; node visits of the longest pixel of a run: mean %.1f, of the frame: %d (mean pixel: %.1f)" % (td[2], td[1] / td[2], td[0], st.node_visits / max(1.0, st.closest_rays / 4.0)))
    ctx.set_timing(1)
    for _ in range(3):
        ctx.trace(wl.width, wl.height)
    ms = []
    for _ in range(8):
        _, s2 = ctx.trace(wl.width, wl.height)
        ms.append((s2.ms_frame, s2.ms_raygen, s2.ms_trace_closest, s2.ms_shade, s2.ms_tail, s2.ms_trace_shadow, s2.ms_resolve))
    ctx.set_timing(0)
    print("   lone frame %.3f ms: cover+entry+raygen %.3f  closest %.3f  shade %.3f  tail %.3f  shadow %.3f  resolve %.3f" % tuple(np.median(np.array(ms), axis=0)))
print("identical frames:", bool(np.array_equal(imgs[0], imgs[1])) and bool(np.array_equal(imgs[0], imgs[2])))
