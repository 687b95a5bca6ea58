#!/usr/bin/env python3
"""Frames in flight: wall time per frame with P frame slots on P streams, for rank 0's bands of an N-way split of the
cfg3 frame — the compute side of the N-GPU scaling ceiling (whole frame time / N-way shard time), measurable on one GPU.
    N_LIST=1,2,4,8 P_LIST=1,4,8 [RT_PARAMS=name=value,...] [MESH=standin|limbs] python tools/pipeline_cost.py
Run with GPU_MAX_HW_QUEUES=8 for P > 4 (bench.py sets it for sharded runs)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402


def main():
    wl = workloads.make(os.environ.get("WORKLOAD", "cfg3"), os.path.join(ROOT, "resources"), mesh=os.environ.get("MESH", "standin"))
    W, H, band = wl.width, wl.height, tiling.BAND_ROWS
    params = [kv.split("=") for kv in os.environ.get("RT_PARAMS", "").split(",") if kv]   # e.g. RT_PARAMS=tail_kernel=0,trace_rays_per_lane=2
    root = RtContext(0)
    wl.apply(root)
    ctxs = [root] + [root.frame_slot() for _ in range(int(os.environ.get("N_CTX", "8")) - 1)]
    for c in ctxs[1:]:
        c.set_instances(wl.instances)
        c.set_uniforms(wl.uniforms)
    for c in ctxs:
        for k, v in params:
            c.set_param(k, int(v))
    print("params", params, flush=True)
    streams = [torch.cuda.Stream() for _ in ctxs]
    whole = {}
    for n in [int(x) for x in os.environ.get("N_LIST", "1,8").split(",")]:
        rows = tiling.max_shard_rows(H, band, n)
        bufs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in ctxs]
        for P in [int(x) for x in os.environ.get("P_LIST", "1,4,8").split(",")]:
            K = 96
            for phase in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    j = i % P
                    ctxs[j].trace_shard(W, H, band, 0, n, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
                for c in ctxs[:P]:
                    c.synchronize()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / K * 1e3
            if n == 1:
                whole[P] = dt
            # per-kernel times of one shard frame run alone
            ctxs[0].set_timing(True)
            for _ in range(3):
                ctxs[0].trace_shard(W, H, band, 0, n, bufs[0].data_ptr(), bufs[0].numel() * 4, streams[0].cuda_stream)
                st = ctxs[0].stats()
            ctxs[0].set_timing(False)
            print(json.dumps({"shards": n, "frames_in_flight": P, "ms_per_frame": round(dt, 4),
                              "ceiling_vs_whole_frame_best": round(min(whole.values()) / dt, 2) if whole else None,
                              "alone_ms": {"frame": round(st.ms_frame, 4), "raygen": round(st.ms_raygen, 4), "closest": round(st.ms_trace_closest, 4), "shade": round(st.ms_shade, 4),
                                           "tail": round(st.ms_tail, 4), "shadow": round(st.ms_trace_shadow, 4), "resolve": round(st.ms_resolve, 4)}}), flush=True)
    for c in reversed(ctxs):
        c.close()


if __name__ == "__main__":
    main()
