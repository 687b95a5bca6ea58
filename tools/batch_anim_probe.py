"""Where does a pass of 8 ANIMATED frames spend its time?  The leg of bench.py (config.in_passes_of_8.animated_ms_per_step) replayed with the
host calls timed one by one.  Usage: python3 tools/batch_anim_probe.py [passes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vulkan_raytracing_amd import RtContext, tiling, workloads  # noqa: E402

wl = workloads.make("cfg3", os.path.join(ROOT, "resources"))
W, H, band = wl.width, wl.height, tiling.BAND_ROWS
P, K = 4, 8
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 8
root = RtContext(0)
for kv in filter(None, os.environ.get("RT_PARAMS", "").split(",")):
    k, v = kv.split("=")
    root.set_param(k, int(v))
wl.apply(root)
ctxs = [root] + [root.frame_slot() for _ in range(P - 1)]
streams = [torch.cuda.Stream() for _ in ctxs]
rows = tiling.max_shard_rows(H, band, 1)
bufs = [torch.zeros((K, rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(P)]
if os.environ.get("PRE"):      # as bench.py does before its passes: the animated loop frame by frame on the same slots
    for c in ctxs:
        c.set_uniforms(wl.uniforms)
    tp = np.float32(0.0)
    one = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(P)]
    for i in range(int(os.environ["PRE"])):
        j = i % P
        tp = np.float32(tp + np.float32(1.0 / 60.0) * np.float32(0.1))
        ctxs[j].set_instances(wl.animate(tp), update=i >= P)
        ctxs[j].trace_shard(W, H, band, 0, 1, one[j].data_ptr(), one[j].numel() * 4, streams[j].cuda_stream)
    for c in ctxs:
        c.synchronize()
    print("pre: %d animated frames one by one; slot 0's last: rays %d + %d + %d" % (int(os.environ["PRE"]), ctxs[0].stats().rays_primary, ctxs[0].stats().rays_secondary, ctxs[0].stats().rays_shadow))
    for c in ctxs:
        c.set_instances(wl.instances)
for leg in ("static", "animated"):
    first = [True] * P
    for phase in range(2):
        tp = np.float32(0.0)
        t_anim = t_set = t_trace = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_pass):
            j = i % P
            a0 = time.perf_counter()
            insts = []
            for _ in range(K):
                if leg == "animated":
                    tp = np.float32(tp + np.float32(1.0 / 60.0) * np.float32(0.1))
                    insts.append(np.array(wl.animate(tp)))
                else:
                    insts.append(np.array(wl.instances))
            a1 = time.perf_counter()
            ctxs[j].set_batch(np.stack(insts), np.concatenate([wl.uniforms] * K), update=not first[j]); first[j] = False
            a2 = time.perf_counter()
            ctxs[j].trace_shard_batch(W, H, band, 0, 1, bufs[j].data_ptr(), bufs[j].numel() * 4, streams[j].cuda_stream)
            a3 = time.perf_counter()
            t_anim += a1 - a0; t_set += a2 - a1; t_trace += a3 - a2
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    st = ctxs[0].stats()
    print("%s: %.4f ms per frame (%d passes of %d); host per pass: animate %.3f ms, rt_set_batch %.3f ms, rt_trace_shard_batch %.3f ms; last pass of slot 0: rays %d + %d + %d, tail faults %d, re-rendered %d" % (
        leg, dt / (n_pass * K) * 1e3, n_pass, K, t_anim / n_pass * 1e3, t_set / n_pass * 1e3, t_trace / n_pass * 1e3,
        st.rays_primary, st.rays_secondary, st.rays_shadow, st.tail_faults, st.frames_rerendered))
for c in ctxs:
    c.set_instances(wl.instances)
for c in reversed(ctxs):
    c.close()
