#!/usr/bin/env python3
"""Frames in flight: wall time per frame with P contexts on P streams (rank 0's bands of an N-way split)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from vulkan_raytracing_amd import RtContext, tiling  # noqa: E402


def main():
    W, H, band = bench.WIDTH, bench.HEIGHT, tiling.BAND_ROWS
    ctxs = []
    params = [kv.split("=") for kv in os.environ.get("RT_PARAMS", "").split(",") if kv]   # e.g. RT_PARAMS=tail_kernel=0,trace_rays_per_lane=2
    for _ in range(int(os.environ.get("N_CTX", "8"))):
        c = RtContext(0)
        bench.build_scene(c, os.path.join(ROOT, "resources"))
        for k, v in params:
            c.set_param(k, int(v))
        ctxs.append(c)
    print("params", params, flush=True)
    streams = [torch.cuda.Stream() for _ in ctxs]
    for n in [int(x) for x in os.environ.get("N_LIST", "1,8").split(",")]:
        rows = tiling.max_shard_rows(H, band, n)
        bufs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in ctxs]
        for P in [int(x) for x in os.environ.get("P_LIST", "1,2,3,4,6,8").split(",")]:
            K = 60
            for phase in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    j = i % P
                    ctxs[j].trace_shard(W, H, band, 0, n, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / K * 1e3
            print("shards %d frames_in_flight %d: %.3f ms/frame" % (n, P, dt), flush=True)


if __name__ == "__main__":
    main()
