"""Frames one at a time on one context, for rocprofv3 --kernel-trace --stats (tools/kstats.sh): RT_PARAMS="a=1,b=2" sets parameters.
Usage: python3 tools/lone_frames.py [workload] [mesh] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkan_raytracing_amd import RtContext, workloads

RES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "resources")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh = sys.argv[2] if len(sys.argv) > 2 else "standin"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = RtContext(0)
for kv in filter(None, os.environ.get("RT_PARAMS", "").split(",")):
    k, v = kv.split("=")
    ctx.set_param(k, int(v))
wl = workloads.make(name, RES, mesh=mesh)
wl.apply(ctx)
for _ in range(n):
    img, st = ctx.trace(wl.width, wl.height)
print("rays", st.rays_primary, st.rays_secondary, st.rays_shadow, "tile rays", st.tile_rays, "blobs", st.blob_tiles, st.blob_tiles_refused)
