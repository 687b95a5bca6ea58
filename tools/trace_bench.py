#!/usr/bin/env python3
"""Kernel-level A/B on the cfg3 frame: per-kernel HIP-event times for each traversal variant and
persistent-grid size.  Usage (GPU box): python tools/trace_bench.py [--time-param T] [--frames N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from vulkan_raytracing_amd import RtContext, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--time-param", type=float, default=0.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounce", type=int, default=3)
    ap.add_argument("--center-type", type=int, default=1)
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--blocks", default="4,6,8")
    ap.add_argument("--blas-builder", type=int, default=0)
    args = ap.parse_args()
    res = os.path.join(ROOT, "resources")
    ctx = RtContext(0)
    arm, label = host.armadillo_path(res)
    geom = host.SceneGeometry([os.path.join(res, "teapot.obj"), arm])
    import time
    ctx.set_param("blas_builder", args.blas_builder)
    ctx.upload_geometry(geom.verts, geom.idx, geom.ranges, build=False)
    for m in range(len(geom.ranges)):
        t0 = time.perf_counter(); ctx.build_blas(m); print(json.dumps({"blas_builder": args.blas_builder, "mesh": m, "triangles": geom.ranges[m][2], "build_ms": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    anim = host.SceneAnimation()
    if args.time_param:
        anim.animate(args.time_param)
    ctx.set_instances(anim.instances((0, 1)))
    ctx.set_uniforms(host.default_uniforms(max_bounce_count=args.bounce, samples_per_pixel=4, center_object_type=args.center_type, orbiting_object_type=0))
    ctx.set_skybox(host.load_skybox(os.path.join(res, "skybox_texture_sea")))
    ctx.set_timing(True)
    W, H = args.width, args.height
    for v in [int(x) for x in args.variants.split(",")]:
        ctx.set_param("trace_variant", v)
        _, cst = ctx.trace(W, H, counting=True)
        dg = list(cst.diag)
        if v in (0, 2):
            print(json.dumps({"variant": v, "diag_closest": {"interior_wave_trips": dg[0], "lanes_busy_per_interior_trip": round(dg[1] / max(1, dg[0]), 2), "wave_cycles": dg[2]},
                              "diag_shadow": {"interior_wave_trips": dg[3], "lanes_busy_per_interior_trip": round(dg[4] / max(1, dg[3]), 2), "wave_cycles": dg[5]}}), flush=True)
        if v == 1:
            # wave-level trips of the interior loop / of the leaf-instance-finish phase, quad-level node visits
            print(json.dumps({"variant": v,
                              "diag_closest": {"interior_wave_trips": dg[0], "other_wave_trips": dg[1], "wave_cycles": dg[2],
                                               "quads_busy_per_interior_trip": round(cst.node_visits / max(1, dg[0]), 2)},
                              "diag_shadow": {"interior_wave_trips": dg[3], "other_wave_trips": dg[4], "wave_cycles": dg[5],
                                              "quads_busy_per_interior_trip": round(cst.node_visits_shadow / max(1, dg[3]), 2)}}), flush=True)
        for b in [int(x) for x in args.blocks.split(",")]:
            ctx.set_param("trace_blocks_per_cu", b)
            ctx.trace(W, H)
            acc = None
            for _ in range(args.frames):
                _, st = ctx.trace(W, H)
                d = st.as_dict(); d.pop("diag", None)
                acc = d if acc is None else {k: (acc[k] + d[k]) for k in d}
            m = {k: acc[k] / args.frames for k in acc}
            rays = m["rays_primary"] + m["rays_secondary"] + m["rays_shadow"]
            print(json.dumps({"variant": v, "blocks_per_cu": b, "frame_ms": round(m["ms_frame"], 4), "Mrays_s": round(rays / m["ms_frame"] / 1e3, 1),
                              "trace_closest_ms": round(m["ms_trace_closest"], 4), "trace_shadow_ms": round(m["ms_trace_shadow"], 4),
                              "shade_ms": round(m["ms_shade"], 4), "raygen_ms": round(m["ms_raygen"], 4), "resolve_ms": round(m["ms_resolve"], 4),
                              "rays": [int(m["rays_primary"]), int(m["rays_secondary"]), int(m["rays_shadow"])],
                              "node_visits_per_closest_ray": round(cst.node_visits / max(1, cst.closest_rays), 3),
                              "tri_tests_per_closest_ray": round(cst.tri_tests / max(1, cst.closest_rays), 3),
                              "node_visits_per_shadow_ray": round(cst.node_visits_shadow / max(1, cst.rays_shadow), 3),
                              "tri_tests_per_shadow_ray": round(cst.tri_tests_shadow / max(1, cst.rays_shadow), 3)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
