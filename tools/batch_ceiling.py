"""Compute-side ceiling of the N-GPU split WITH frame batches: rank 0's shard of an N-way split rendered K frames at a time
(rt_trace_shard_batch) on P frame slots, against the whole frame (N = 1, K = 1, 4 slots).  ms per SHARD FRAME.
    N_LIST=8 K_LIST=1,2,4,8 P_LIST=2,4 python3 tools/batch_ceiling.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402

wl = workloads.make("cfg3", os.path.join(ROOT, "resources"), mesh=os.environ.get("MESH", "standin"))
W, H, band = wl.width, wl.height, tiling.BAND_ROWS
animate = os.environ.get("ANIMATE", "0") == "1"
PMAX = max(int(x) for x in os.environ.get("P_LIST", "2,4").split(","))
root = RtContext(0)
wl.apply(root)
ctxs = [root] + [root.frame_slot() for _ in range(PMAX - 1)]
streams = [torch.cuda.Stream() for _ in ctxs]
t_anim = np.float32(0.0)


def inputs(K):
    global t_anim
    insts, unis = [], []
    for _ in range(K):
        if animate:
            t_anim = np.float32(t_anim + np.float32(1.0 / 60.0) * np.float32(0.1))
            insts.append(np.array(wl.animate(t_anim)))
        else:
            insts.append(np.array(wl.instances))
        unis.append(wl.uniforms)
    return np.stack(insts), np.concatenate(unis)


whole = None
for n in [int(x) for x in os.environ.get("N_LIST", "1,8").split(",")]:
    rows = tiling.max_shard_rows(H, band, n)
    for K in [int(x) for x in os.environ.get("K_LIST", "1,8").split(",")]:
        for P in [int(x) for x in os.environ.get("P_LIST", "2,4").split(",")]:
            bufs = [torch.zeros((K, rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(P)]
            first = [True] * P
            frames = 192
            for phase in range(2):
                t_anim = np.float32(0.0)      # every configuration renders the same frames
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = 0
                i = 0
                while done < frames:
                    j = i % P
                    inst, uni = inputs(K)
                    ctxs[j].set_batch(inst, uni, update=not first[j]); first[j] = False
                    ctxs[j].trace_shard_batch(W, H, band, 0, n, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
                    done += K; i += 1
                for c in ctxs[:P]:
                    c.synchronize()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / done * 1e3
            if n == 1 and (whole is None or dt < whole):
                whole = dt
            print(json.dumps({"shards": n, "frames_per_batch": K, "slots": P, "animated": animate, "ms_per_shard_frame": round(dt, 4),
                              "ceiling_vs_whole_frame": round(whole / dt, 2) if whole else None}), flush=True)
            for c in ctxs:
                c.set_instances(wl.instances)      # back to single frames (a context that holds a batch refuses rt_trace_shard)
for c in reversed(ctxs):
    c.close()
