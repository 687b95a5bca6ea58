import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vulkan_raytracing_amd import RtContext
RES = os.path.join(ROOT, "resources")
ctx = RtContext(0)
sp = scenes.two_object_scene(os.path.join(RES, "teapot.obj"), os.path.join(RES, "cube.obj"), 1, 0, 1, 1, ctx=ctx, time_param=0.3)
for cam, fwd in (((0.0, 0.0, 5000.0), (0.0, 0.0, -2000.0)), ((3000.0, 200.0, 4000.0), (-1200.0, -80.0, -1600.0)), ((0.0, 0.0, 20.0), (0.0, 0.0, -1.0))):
    u = sp.uniforms.copy()
    u[0]["position"][:3] = cam; u[0]["forward"][:3] = fwd
    u[0]["center_object_type"] = 0
    sp.set_uniforms(u)
    ref, rc = sp.orc.render(200, 120)
    for ep in (0, 1):
        for se in (0, 1):
            ctx.set_param("entry_points", ep); ctx.set_param("shadow_entry", se)
            img, st = ctx.trace(200, 120)
            bad = int((np.abs(img - ref).max(axis=2) > 1e-3).sum())
            print("cam", cam, "entry", ep, "shadow_entry", se, "rays", (st.rays_primary, st.rays_secondary, st.rays_shadow), "oracle", tuple(int(x) for x in rc), "bad px", bad)
ctx.close()
