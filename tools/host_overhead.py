#!/usr/bin/env python3
"""Host-side cost per step of bench.py's sharded loop (rank 0 of an 8-way split: 136 rows), measured on ONE GPU with a
world-size-1 RCCL group: trace_shard + RGB slice + dist.gather + index_select, P slots in flight.  Prints the time the host
needs to enqueue a step and the wall time per step; a step whose enqueue takes longer than the GPU needs is host-bound.
    GPU_MAX_HW_QUEUES=16 P=16 python tools/host_overhead.py"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402

P = int(os.environ.get("P", "16"))
N = int(os.environ.get("N", "8"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
wl = workloads.make("cfg3", os.path.join(ROOT, "resources"))
W, H, band = wl.width, wl.height, tiling.BAND_ROWS
root = RtContext(0)
wl.apply(root)
ctxs = [root] + [root.frame_slot() for _ in range(P - 1)]
for c in ctxs[1:]:
    c.set_instances(wl.instances); c.set_uniforms(wl.uniforms)
rows = tiling.max_shard_rows(H, band, N)
shards = [torch.zeros((rows, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
gathered = [torch.zeros((1, rows, W, 3), dtype=torch.float32, device=dev) for _ in range(P)]
full = [torch.zeros((rows, W, 3), dtype=torch.float32, device=dev) for _ in range(P)]
perm = torch.arange(rows, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(P)]


def step(i, mode):
    j = i % P
    with torch.cuda.stream(streams[j]):
        ctxs[j].trace_shard(W, H, band, 0, N, shards[j].data_ptr(), shards[j].numel() * 4, streams[j].cuda_stream)
        if mode >= 1:
            rgb = shards[j][..., :3].contiguous()
            if mode >= 2:
                dist.gather(rgb, list(gathered[j].unbind(0)), dst=0)
                torch.index_select(gathered[j].view(rows, W, 3), 0, perm, out=full[j])


for mode, label in ((0, "trace_shard only"), (1, "+ RGB slice"), (2, "+ gather + index_select")):
    for phase in range(2):
        torch.cuda.synchronize()
        K = 480
        t0 = time.perf_counter()
        for i in range(K):
            step(i, mode)
        t1 = time.perf_counter()
        for s in streams:
            s.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("%-26s host enqueue %.1f us per step, wall %.1f us per step" % (label, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
for c in reversed(ctxs):
    c.close()
dist.destroy_process_group()
