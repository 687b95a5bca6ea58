"""Tile-blob statistics and lone kernel times of a workload with the tile path on and off (one context, frames one at a time).
Usage: python3 tools/blob_probe.py [cfg3|cfg4|cfg5] [standin|limbs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from vulkan_raytracing_amd import RtContext, workloads

RES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "resources")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh = sys.argv[2] if len(sys.argv) > 2 else "standin"
ctx = RtContext(0)
wl = workloads.make(name, RES, mesh=mesh)
wl.apply(ctx)
W, H = wl.width, wl.height
frames = {}
for on in (1, 0, 1, 0):
    ctx.set_param("tile_blobs", on)
    img, st = ctx.trace(W, H, counting=True)
    frames[on] = img
    print("tile_blobs=%d counting: closest rays %d  tile rays %d (handed on %d)  blobs %d (large %d) refused %d  nodes/blob %.1f tris/blob %.1f  visits/ray %.2f tris/ray %.2f" % (
        on, st.closest_rays, st.tile_rays, st.tile_rays_handed_on, st.blob_tiles, st.blob_tiles_large, st.blob_tiles_refused,
        st.blob_nodes / max(1, st.blob_tiles), st.blob_tris / max(1, st.blob_tiles), st.node_visits / max(1, st.closest_rays), st.tri_tests / max(1, st.closest_rays)))
    if st.blob_tiles:
        w = 4.0 * st.blob_tiles
        print("   k_tile wave cycles (counting build): before the walk %.0f (entry there at %.0f, blob in LDS at %.0f, ray set up at %.0f)  walk %.0f  all %.0f per wave" % (
            st.tile_diag[0] / w, st.tile_diag[3] / w, st.tile_diag[4] / w, st.tile_diag[5] / w, st.tile_diag[1] / w, st.tile_diag[2] / w))
    ctx.set_timing(1)
    for _ in range(3):
        ctx.trace(W, H)
    ms = []
    for _ in range(8):
        _, s2 = ctx.trace(W, H)
        ms.append((s2.ms_frame, s2.ms_raygen, s2.ms_trace_closest, s2.ms_shade, s2.ms_tail, s2.ms_trace_shadow, s2.ms_resolve))
    ctx.set_timing(0)
    m = np.median(np.array(ms), axis=0)
    print("   lone frame %.3f ms: cover+entry+blob+raygen %.3f  closest %.3f  shade %.3f  tail %.3f  shadow %.3f  resolve %.3f" % tuple(m))
print("identical frames:", bool(np.array_equal(frames[0], frames[1])))
