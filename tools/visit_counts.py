#!/usr/bin/env python3
"""Mean node visits / triangle tests per ray of the closest-hit and the shadow traversal (instrumented kernels), per workload
and mesh, with the given rt_set_param settings:  [RT_PARAMS=name=value,...] [WORKLOADS=cfg3,cfg5] [MESHES=standin,limbs] python3 tools/visit_counts.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkan_raytracing_amd import RtContext, workloads  # noqa: E402


def main():
    params = [kv.split("=") for kv in os.environ.get("RT_PARAMS", "").split(",") if kv]
    for name in os.environ.get("WORKLOADS", "cfg3").split(","):
        for mesh in os.environ.get("MESHES", "standin,limbs").split(","):
            wl = workloads.make(name, os.path.join(ROOT, "resources"), mesh=mesh)
            c = RtContext(0)
            wl.apply(c)
            for k, v in params:
                c.set_param(k, int(v))
            _, st = c.trace(wl.width, wl.height, counting=True)
            print(json.dumps({"workload": name, "mesh": mesh, "params": params,
                              "rays": [st.rays_primary, st.rays_secondary, st.rays_shadow], "closest_rays": st.closest_rays,
                              "nodes_per_closest_ray": round(st.node_visits / max(1, st.closest_rays), 3), "tris_per_closest_ray": round(st.tri_tests / max(1, st.closest_rays), 3),
                              "nodes_per_shadow_ray": round(st.node_visits_shadow / max(1, st.rays_shadow), 3), "tris_per_shadow_ray": round(st.tri_tests_shadow / max(1, st.rays_shadow), 3)}), flush=True)
            c.close()


if __name__ == "__main__":
    main()
