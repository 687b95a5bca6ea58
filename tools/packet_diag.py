#!/usr/bin/env python3
"""Packet kernel: wave-level work of one counting frame (cfg3) and lone kernel times with the packet kernel on and off."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkan_raytracing_amd import RtContext, workloads
wl = workloads.make(os.environ.get("WORKLOAD", "cfg3"), os.path.join(ROOT, "resources"), mesh=os.environ.get("MESH", "standin"))
c = RtContext(0)
wl.apply(c)
for kv in [x for x in os.environ.get("RT_PARAMS", "").split(",") if x]:
    k, v = kv.split("="); c.set_param(k, int(v))
W, H = wl.width, wl.height
for pk in (1, 0):
    c.set_param("packet_trace", pk)
    _, st = c.trace(W, H, counting=True)
    d = list(st.diag)
    c.set_timing(True)
    ms = []
    for _ in range(5):
        _, s2 = c.trace(W, H)
        ms.append((s2.ms_trace_closest, s2.ms_trace_shadow, s2.ms_raygen, s2.ms_frame))
    c.set_timing(False)
    ms.sort()
    print(json.dumps({"packet": pk, "closest_rays": st.closest_rays, "shadow_rays": st.rays_shadow,
                      "per_ray": {"closest_nodes": round(st.node_visits / max(1, st.closest_rays), 2), "closest_tris": round(st.tri_tests / max(1, st.closest_rays), 2),
                                  "shadow_nodes": round(st.node_visits_shadow / max(1, st.rays_shadow), 2), "shadow_tris": round(st.tri_tests_shadow / max(1, st.rays_shadow), 2)},
                      "per_packet_of_64" if pk else "diag": {"closest_nodes": round(d[0] / max(1, st.closest_rays / 64), 1), "closest_tris": round(d[1] / max(1, st.closest_rays / 64), 1), "closest_wave_cycles": d[2],
                                                               "shadow_nodes": round(d[3] / max(1, st.rays_shadow / 64), 1), "shadow_tris": round(d[4] / max(1, st.rays_shadow / 64), 1), "shadow_wave_cycles": d[5]},
                      "lone_ms_median": {"closest": round(ms[2][0], 4), "shadow": round(ms[2][1], 4), "raygen": round(ms[2][2], 4), "frame": round(ms[2][3], 4)}}), flush=True)
c.close()
