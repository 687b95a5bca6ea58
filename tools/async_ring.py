#!/usr/bin/env python3
"""Frames in flight through rt_trace_async / rt_trace_wait (pixels copied to pinned host memory every frame):
ms per frame for P contexts, with and without a per-frame TLAS update.  Usage: python tools/async_ring.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vulkan_raytracing_amd import RtContext, host  # noqa: E402


def main():
    W, H = bench.WIDTH, bench.HEIGHT
    ctxs = []
    for _ in range(4):
        c = RtContext(0)
        geom, inst, u, sky, label = bench.build_scene(c, os.path.join(ROOT, "resources"))
        ctxs.append(c)
    K = 40
    for rgba8, update in ((0, False), (0, True), (1, True)):
        for c in ctxs:
            c.set_param("output_rgba8", rgba8)
        for P in (1, 2, 4):
            ring = ctxs[:P]
            for phase in range(2):
                pending = [False] * P
                t0 = time.perf_counter()
                for i in range(K):
                    j = i % P
                    if pending[j]:
                        ring[j].trace_wait(copy=False)
                    if update:
                        ring[j].set_instances(inst, update=True)
                    ring[j].trace_async(W, H)
                    pending[j] = True
                for j in range(P):
                    if pending[j]:
                        ring[j].trace_wait(copy=False)
                dt = (time.perf_counter() - t0) / K * 1e3
            print("rgba8 %d tlas_update %d frames_in_flight %d: %.3f ms/frame (pixels in host memory)" % (rgba8, update, P, dt), flush=True)


if __name__ == "__main__":
    main()
