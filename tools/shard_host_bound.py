"""Is the 1/N-shard loop host-bound?  Enqueue time per shard frame (the Python + C++ + HIP launch path, no waiting) against the
wall time per shard frame, rank 0's shard of an N-way split, P slots in flight.  N=8 P=16 python3 tools/shard_host_bound.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402

wl = workloads.make("cfg3", os.path.join(ROOT, "resources"))
W, H, band = wl.width, wl.height, tiling.BAND_ROWS
n, P = int(os.environ.get("N", "8")), int(os.environ.get("P", "16"))
root = RtContext(0)
wl.apply(root)
ctxs = [root] + [root.frame_slot() for _ in range(P - 1)]
for c in ctxs[1:]:
    c.set_instances(wl.instances); c.set_uniforms(wl.uniforms)
for kv in filter(None, os.environ.get("RT_PARAMS", "").split(",")):
    k, v = kv.split("=")
    for c in ctxs:
        c.set_param(k, int(v))
streams = [torch.cuda.Stream() for _ in ctxs]
rows = tiling.max_shard_rows(H, band, n)
bufs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in ctxs]
K = 192
for phase in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        j = i % P
        ctxs[j].trace_shard(W, H, band, 0, n, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
    t1 = time.perf_counter()
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("N=%d P=%d: enqueue %.4f ms per shard frame, wall %.4f ms per shard frame" % (n, P, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
