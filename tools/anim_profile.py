import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np, torch
from vulkan_raytracing_amd import RtContext, workloads, tiling
ROOT=os.getcwd()
wl = workloads.make("cfg3", os.path.join(ROOT, "resources"))
W,H=wl.width,wl.height
root=RtContext(0); wl.apply(root)
ctxs=[root]+[root.frame_slot() for _ in range(3)]
for c in ctxs[1:]: c.set_instances(wl.instances); c.set_uniforms(wl.uniforms)
bufs=[torch.zeros((H,W,4),dtype=torch.float32,device="cuda:0") for _ in ctxs]
streams=[torch.cuda.Stream() for _ in ctxs]
T=dict(animate=0.0,set_inst=0.0,set_uni=0.0,trace=0.0)
tp=np.float32(0)
def step(k, prof):
    global tp
    j=k%4; c=ctxs[j]
    t0=time.perf_counter(); tp=np.float32(tp+np.float32(1/600)); inst=wl.animate(tp) if not os.environ.get('STATIC_INST') else wl.instances; t1=time.perf_counter()
    if not os.environ.get('NO_SET'): c.set_instances(inst, update=True)
    t2=time.perf_counter()
    c.set_uniforms(wl.uniforms); t3=time.perf_counter()
    c.trace_shard(W,H,8,0,1,bufs[j].data_ptr(),bufs[j].numel()*4,streams[j].cuda_stream); t4=time.perf_counter()
    if prof:
        T["animate"]+=t1-t0; T["set_inst"]+=t2-t1; T["set_uni"]+=t3-t2; T["trace"]+=t4-t3
for k in range(16): step(k, False)
torch.cuda.synchronize()
t0=time.perf_counter()
N=200
for k in range(N): step(k, True)
torch.cuda.synchronize()
dt=time.perf_counter()-t0
print("animated: %.3f ms/step; host ms per step:"%(dt/N*1e3), {k:round(v/N*1e3,4) for k,v in T.items()})
