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
for dist in (20.0, 700.0, 2000.0, 5000.0, 10000.0, 20000.0, 60000.0, 200000.0):
    n = 30000
    o = rng.normal(size=(n, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True); o *= dist
    tgt = rng.uniform(-2.5, 2.5, (n, 3)); tgt[:, 1] = rng.uniform(0, 1.6, n)
    half = rng.random(n) < 0.5
    tgt[half] = rng.uniform(-1.2, 1.2, (int(half.sum()), 3))
    ax = rng.integers(0, 3, n)
    par = rng.random(n) < 0.33
    o[par] = tgt[par]
    o[par, ax[par]] += dist * rng.choice([-1.0, 1.0], int(par.sum()))
    o[par] += rng.normal(scale=1e-3, size=(int(par.sum()), 3))
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32); rays[:, 0:3] = o; rays[:, 3] = 0.001; rays[:, 4:7] = d; rays[:, 7] = 1e9
    g, _ = ctx.intersect(rays)
    b = sp.orc.intersect(rays, use_bvh=False)
    same = (g["prim"] == b["prim"]) & (g["inst"] == b["inst"]) & (g["t"].view(np.uint32) == b["t"].view(np.uint32))
    print("dist", dist, "mismatches", int((~same).sum()))
    for k in np.nonzero(~same)[0][:4]:
        print("   ray", k, "par", bool(par[k]), "o", rays[k, :3].tolist(), "d", rays[k, 4:7].tolist(), "gpu", g[k], "brute", b[k])
        # all hits of this ray by brute force per triangle? ask the oracle with tmax just above/below
        r2 = rays[k:k+1].copy(); r2[0, 7] = b[k]["t"] * (1 + 1e-6)
        print("     brute tmax-limited:", sp.orc.intersect(r2, use_bvh=False)[0], " gpu:", ctx.intersect(r2)[0][0])
        for v in (1, 2):
            ctx.set_param("trace_variant", v); print("     variant", v, ctx.intersect(rays[k:k+1])[0][0]); 
        ctx.set_param("trace_variant", 0)
ctx.close()
