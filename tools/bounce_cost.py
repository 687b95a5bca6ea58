#!/usr/bin/env python3
"""What do bounces >= 1 cost a frame with P slots in flight?  cfg3 with maxBounceCount 3 (the workload) against the same scene
with maxBounceCount 0 (no k_tail, no secondary rays): the difference bounds what hiding k_tail behind the shadow kernel could gain.
    P=4 python tools/bounce_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402


def run(max_bounce, P, params=()):
    wl = workloads.make("cfg3", os.path.join(ROOT, "resources"), mesh=os.environ.get("MESH", "standin"))
    wl.uniforms[0]["max_bounce_count"] = max_bounce
    W, H = wl.width, wl.height
    root = RtContext(0)
    wl.apply(root)
    ctxs = [root] + [root.frame_slot() for _ in range(P - 1)]
    for c in ctxs[1:]:
        c.set_instances(wl.instances)
        c.set_uniforms(wl.uniforms)
    for c in ctxs:
        for k, v in params:
            c.set_param(k, v)
    streams = [torch.cuda.Stream() for _ in ctxs]
    bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in ctxs]
    K = 120
    for phase in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            j = i % P
            ctxs[j].trace_shard(W, H, tiling.BAND_ROWS, 0, 1, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
    st = ctxs[0].stats()
    for c in reversed(ctxs):
        c.close()
    return dt, (st.rays_primary, st.rays_secondary, st.rays_shadow)


P = int(os.environ.get("P", "4"))
for mb, params in ((3, ()), (0, ()), (3, (("tail_kernel", 0),))):
    dt, rays = run(mb, P, params)
    print("maxBounceCount %d %s: %.4f ms per frame, rays %s" % (mb, dict(params), dt, rays), flush=True)
