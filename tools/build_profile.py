#!/usr/bin/env python3
"""BLAS build only (cfg3 scene, device builder): run under rocprofv3 --kernel-trace --stats for per-kernel build times.
    RT_GPU_BVH_ALGO=1|2|3 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_build -- python3 tools/build_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkan_raytracing_amd import RtContext, workloads  # noqa: E402

wl = workloads.make("cfg3", os.path.join(ROOT, "resources"), mesh=os.environ.get("MESH", "standin"))
for _ in range(int(os.environ.get("REPS", "3"))):
    c = RtContext(0)
    wl.apply(c)
    c.close()
