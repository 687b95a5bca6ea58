import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vulkan_raytracing_amd import RtContext
RES = os.path.join(ROOT, "resources")
ctx = RtContext(0)
sp = scenes.two_object_scene(os.path.join(RES, "teapot.obj"), os.path.join(RES, "cube.obj"), 1, 0, 1, 1, ctx=ctx, time_param=0.3)
rng = np.random.default_rng(5)
for dist in (20.0, 200.0, 2000.0, 20000.0, 200000.0):
    n = 40000
    o = rng.normal(size=(n, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True); o *= dist
    tgt = rng.uniform(-2.5, 2.5, (n, 3)); tgt[:, 1] = rng.uniform(0, 1.6, n)
    half = rng.random(n) < 0.5
    tgt[half] = np.array([0, 0, 0]) + rng.uniform(-1.2, 1.2, (int(half.sum()), 3)) + np.array([np.sin(0.3*np.pi)*5*0, 0, 0])
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32); rays[:, 0:3] = o; rays[:, 3] = 0.001; rays[:, 4:7] = d; rays[:, 7] = 1e9
    g, _ = ctx.intersect(rays)
    b = sp.orc.intersect(rays, use_bvh=False)
    v = sp.orc.intersect(rays, use_bvh=True)
    same = (g["prim"] == b["prim"]) & (g["inst"] == b["inst"]) & (g["t"].view(np.uint32) == b["t"].view(np.uint32))
    samev = (v["prim"] == b["prim"]) & (v["inst"] == b["inst"]) & (v["t"].view(np.uint32) == b["t"].view(np.uint32))
    for k in np.nonzero(~same)[0][:4]:
        print("   ray", k, "o", rays[k, :3], "d", rays[k, 4:7], "gpu", g[k], "brute", b[k])
    print("dist %8.0f: hits %.2f  gpu != brute: %d   oracle-bvh != brute: %d   (gpu missed hits: %d, gpu extra hits: %d)" % (
        dist, (b["inst"] >= 0).mean(), int((~same).sum()), int((~samev).sum()), int(((g["inst"] < 0) & (b["inst"] >= 0)).sum()), int(((g["inst"] >= 0) & (b["inst"] < 0)).sum())))
ctx.close()
