#!/usr/bin/env python3
"""Per-rank cost of an N-way sharded frame measured on ONE GPU (rank 0's bands only, no gather):
wall time per step with asynchronous launches vs the HIP-event GPU time.  Shows whether a small
shard is GPU-bound or launch-bound.  Usage: python tools/shard_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from vulkan_raytracing_amd import RtContext, tiling  # noqa: E402


def main():
    ctx = RtContext(0)
    bench.build_scene(ctx, os.path.join(ROOT, "resources"))
    W, H, band = bench.WIDTH, bench.HEIGHT, tiling.BAND_ROWS
    stream = torch.cuda.current_stream()
    rpl = int(os.environ.get("RPL", "4")); mb = int(os.environ.get("MINB", "256"))
    ctx.set_param("trace_rays_per_lane", rpl); ctx.set_param("trace_min_blocks", mb)
    print("rays_per_lane", rpl, "min_blocks", mb)
    for n in (1, 2, 4, 8):
        rows = tiling.max_shard_rows(H, band, n)
        buf = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0")
        for timing in (False, True):
            ctx.set_timing(timing)
            for _ in range(3):
                ctx.trace_shard(W, H, band, 0, n, buf.data_ptr(), buf.numel() * 4, stream.cuda_stream)
            torch.cuda.synchronize()
            K = 50
            t0 = time.perf_counter()
            for _ in range(K):
                ctx.trace_shard(W, H, band, 0, n, buf.data_ptr(), buf.numel() * 4, stream.cuda_stream)
            t_launch = (time.perf_counter() - t0) / K * 1e3
            torch.cuda.synchronize()
            t_wall = (time.perf_counter() - t0) / K * 1e3
            st = ctx.stats()
            print("shards %d timing %d: wall %.3f ms/step, host enqueue %.3f ms/step, gpu frame (events) %.3f ms, rays %d | raygen %.3f closest %.3f (%d launches) shade %.3f shadow %.3f resolve %.3f" %
                  (n, timing, t_wall, t_launch, st.ms_frame, st.rays_primary + st.rays_secondary + st.rays_shadow,
                   st.ms_raygen, st.ms_trace_closest, st.launches_trace_closest, st.ms_shade, st.ms_trace_shadow, st.ms_resolve), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
