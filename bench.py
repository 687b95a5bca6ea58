#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + secondary + shadow) of the MI355X ray-tracing stage on BASELINE
config 3: teapot.obj (mirror) + armadillo (diffuse; STAND-IN mesh unless resources/armadillo.obj is
supplied) + skybox_texture_sea, 1920x1080, depth 4 (maxBounceCount 3) + shadow rays, spp 4.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU, --frames-in-flight independent frames in flight per GPU (the reference's swapchain keeps
minImageCount + 1 frames in flight, src/main.cpp:1203, 2967), each on its own HIP stream with its own hardware queue:
4 frames / the 4 default queues on one GPU, 8 frames / GPU_MAX_HW_QUEUES=8 when the frame is sharded.  A step = one frame of the hot path: raygen -> [closest-hit traversal -> shade]
x 4 bounces -> any-hit shadow traversal -> resolve, on this rank's interleaved 8-row bands, followed
(N > 1) by ONE RCCL gather of the compact shards to rank 0 and the row permutation that reassembles
the frame.  The frame is fixed, so scaling is STRONG.  Inputs (scene, BVH, cube map) are resident in
HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# A 1/N frame shard is latency-bound, so a sharded run keeps 8 frames in flight and gives each of their streams its own
# hardware queue (HIP maps streams onto 4 by default; read at HIP start-up).  Measured on one GPU, rank 0's shard of an
# 8-way split: 0.143 ms per frame with 4 frames / 4 queues, 0.124 ms with 8 / 8; a whole frame is 3 % slower with 8 / 8,
# so a single-GPU run keeps the defaults.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vulkan_raytracing_amd import RtContext, host, tiling, workloads  # noqa: E402

WIDTH, HEIGHT, MAX_BOUNCE, SPP = 1920, 1080, 3, 4   # BASELINE config 3 (the headline); --workload cfg4 / cfg5 change them
WORKLOAD = "cfg3"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
RAY_BYTES, HIT_BYTES = 32, 20


_WL = {}


def workload(res, mesh):
    # host-side ingest (OBJ parse, JPEG decode) happens once per process; every context gets its own upload + BLAS/TLAS
    key = (WORKLOAD, mesh)
    if key not in _WL:
        _WL[key] = workloads.make(WORKLOAD, res, mesh=mesh)
    return _WL[key]


def cpu_baseline(wl, budget_s=12.0):
    """The oracle (oracle/rt_oracle.cpp, kind "port") on this box's host cores: whole 1920x1080 frames of
    the same workload, repeated until about `budget_s` seconds of CPU work.  Checker code: used here ONLY
    as the reported CPU baseline, never by the product path.  Threads = the box's CPU share for one GPU
    (16) unless RT_CPU_THREADS says otherwise."""
    from oracle import oracle as orc
    S = orc.OracleScene()
    geom, inst = wl.geometry, wl.instances
    S.set_geometry(geom.verts, geom.idx, geom.ranges)
    S.set_instances([inst[i].tobytes() for i in range(len(inst))])
    S.set_uniforms(wl.uniforms.tobytes())
    S.set_skybox(wl.sky)
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = int(os.environ.get("RT_CPU_THREADS", min(avail, 16)))
    t0 = time.time()
    _, rc = S.render(WIDTH, HEIGHT, threads=cores)
    one = max(time.time() - t0, 1e-3)
    reps = int(max(1, min(200, round(budget_s / one))))
    rays = 0
    t0 = time.time()
    for _ in range(reps):
        _, rc = S.render(WIDTH, HEIGHT, threads=cores)
        rays += int(rc.sum())
    dt = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the same workload (%d rays, %.1f s), oracle/rt_oracle.cpp with its own SAH BVH, %d threads of %d visible"
                      % (reps, WIDTH, HEIGHT, rays, dt, cores, avail)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frames in flight per GPU, each on its own stream and buffers; the reference keeps "
                         "swapchainImageCount = minImageCount + 1 frames in flight (src/main.cpp:1203, 2790, 2905-2967).  0 = auto: "
                         "4 on one GPU, 8 when the frame is split over several — one per HIP hardware queue (isolated kernels are latency-bound; frames in flight fill the gaps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--param", action="append", default=[], help="rt_set_param NAME=VALUE on every context (experiments), repeatable")
    ap.add_argument("--save-image", default=None, help="write the last frame as PFM (rank 0)")
    ap.add_argument("--variant", type=int, default=None, help="traversal kernel: 0 = quantized BVH2, one lane per ray (default); 1 = BVH4, four lanes per ray; 2 = 4-ary records, one lane per ray")
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg4", "cfg5"],
                    help="cfg3 (default, the headline): 1920x1080 depth 4; cfg4: 3840x2160 depth 6; cfg5: 16 instances of the armadillo BLAS, 1920x1080 depth 4")
    ap.add_argument("--mesh", default="standin", choices=["limbs", "standin"],
                    help="which stand-in replaces the missing resources/armadillo.obj: limbs = the non-star-shaped figure (default, "
                         "the harder and more armadillo-like one), standin = the geodesic blob of round 1; ignored when the real file is supplied")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks share cuda:0 and gather through gloo/CPU tensors: exercises rank->band mapping, gather and "
                         "reassembly where only one GPU exists (its throughput is meaningless)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: still create the RCCL process group and run the per-frame gather (to itself) — a smoke test of the N > 1 code path")
    args = ap.parse_args()
    global WIDTH, HEIGHT, MAX_BOUNCE, WORKLOAD
    WORKLOAD = args.workload
    if WORKLOAD == "cfg4":
        WIDTH, HEIGHT, MAX_BOUNCE = 3840, 2160, 5

    # stdout carries exactly one JSON line: library banners (RCCL prints its version to fd 1) go to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-tracing stage has no CPU path")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
        if args.force_collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", local_rank)
    n = world
    collective = n > 1 or args.force_collective
    assert args.gpus == n, "--gpus must equal the number of launched ranks"
    P = args.frames_in_flight if args.frames_in_flight > 0 else (4 if n == 1 else 8)

    res = os.path.join(ROOT, "resources")
    if rank == 0:
        host.armadillo_path(res, kind=args.mesh)  # generate the stand-in once before the other ranks look for it
    if collective:
        dist.barrier()
    wl = workload(res, args.mesh)
    # one context (scene replica, queues, counters) per frame in flight — the analogue of the reference's
    # per-swapchain-image command buffer, fence and semaphores (src/main.cpp:2597, 2740-2749)
    ctxs = []
    for _ in range(P):
        c = RtContext(local_rank)
        wl.apply(c)
        if args.variant is not None:
            c.set_param("trace_variant", args.variant)
        for kv in args.param:
            k, v = kv.split("=")
            c.set_param(k, int(v))
        if args.blocks_per_cu is not None:
            c.set_param("trace_blocks_per_cu", args.blocks_per_cu)
        ctxs.append(c)
    ctx = ctxs[0]

    band = tiling.BAND_ROWS
    rows_max = tiling.max_shard_rows(HEIGHT, band, n)
    shards = [torch.zeros((rows_max, WIDTH, 4), dtype=torch.float32, device=dev) for _ in range(P)]
    # the gather carries RGB only: alpha is exactly 1.0 in every pixel (sum of spp ones divided by spp, src/shader.rgen:180-183),
    # so 25 % of the xGMI traffic into rank 0 would be constants
    gathered = [torch.zeros((n, rows_max, WIDTH, 3), dtype=torch.float32, device=dev) if (rank == 0 and collective) else None for _ in range(P)]
    # the assembled frame on rank 0 stays RGB (alpha is the constant 1.0): the row permutation writes it in one kernel
    full = [torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev) if (rank == 0 and collective) else None for _ in range(P)]
    perm = None
    if rank == 0 and collective:
        src = np.zeros(HEIGHT, np.int64)
        for s in range(n):
            m = tiling.shard_row_map(HEIGHT, band, s, n)
            src[m] = s * rows_max + np.arange(len(m))
        perm = torch.as_tensor(src, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    frames = [None] * P
    counter = [0]

    def step():
        j = counter[0] % P
        counter[0] += 1
        with torch.cuda.stream(streams[j]):
            ctxs[j].trace_shard(WIDTH, HEIGHT, band, rank, n, shards[j].data_ptr(), shards[j].numel() * 4, streams[j].cuda_stream)
            if n > 1 and args.rehearse_on_one_gpu:
                streams[j].synchronize()
                host_shard = shards[j].cpu()
                parts = [torch.zeros_like(host_shard) for _ in range(n)] if rank == 0 else None
                dist.gather(host_shard, parts, dst=0)
                if rank == 0:
                    frames[j] = torch.cat(parts).index_select(0, perm.cpu()).to(dev)
            elif collective:
                rgb = shards[j][..., :3].contiguous()
                dist.gather(rgb, list(gathered[j].unbind(0)) if rank == 0 else None, dst=0)
                if rank == 0:
                    torch.index_select(gathered[j].view(n * rows_max, WIDTH, 3), 0, perm, out=full[j])
                    frames[j] = full[j]
            else:
                frames[j] = shards[j]

    def sync():
        for s_ in streams:
            s_.synchronize()
        torch.cuda.synchronize(dev)
        if collective:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ctx.set_timing(True)   # HIP events around every kernel of context 0's frames (every P-th frame), on their own stream
    # set-up, not a step: one frame per context so that every context has its ray queues allocated before the
    # warm-up/timed steps start (a context allocates them on its first frame)
    for _ in range(P):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    ctx.stats()             # drop the warm-up frames' event times
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()        # counters of context 0's last frame + MEAN event times over all its timed frames (every P-th step)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays = torch.tensor([st.rays_primary, st.rays_secondary, st.rays_shadow], dtype=torch.float64, device=dev)
    if collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    rays_frame = [int(x) for x in rays.tolist()]
    total_rays = sum(rays_frame)

    result = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_rays * args.steps / dt / 1e6
        # the headline workload reports BASELINE.json's own metric string; "secondary" there = every ray after the primary one,
        # i.e. bounce rays + shadow rays (config.rays_per_frame lists the classes)
        metric = "Mrays/sec (primary+secondary+shadow) at %dx%d depth %d" % (WIDTH, HEIGHT, MAX_BOUNCE + 1)
        if WORKLOAD == "cfg3":
            try:
                metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:
                pass
        result = {"metric": metric, "value": value, "unit": "Mrays/s",
                  "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": wl.describe(), "mesh": wl.mesh_label,
                             "rays_per_frame": {"primary": rays_frame[0], "secondary": rays_frame[1], "shadow": rays_frame[2]},
                             "ray_classes": "value counts every traceRayEXT-equivalent: primary + secondary (bounce) + shadow rays; 'secondary' in the metric string means both",
                             "parallelism": "interleaved %d-row bands over %d GPU(s), scene replicated, one RCCL gather per frame, %d frames in flight per GPU" % (band, n, P),
                             "frames_in_flight": P, "device": ctx.device_info}}
    # ---- roofline of the dominant kernel (closest-hit traversal), rank 0's shard -----------------
    if rank == 0:
        # (1) isolated frames: the same shard, one frame at a time on context 0, HIP events around every kernel
        iso = []
        for _ in range(5):
            ctx.trace_shard(WIDTH, HEIGHT, band, rank, n, shards[0].data_ptr(), shards[0].numel() * 4, streams[0].cuda_stream)
            iso.append(ctx.stats())
        iso_ms = sorted(x.ms_trace_closest for x in iso)[len(iso) // 2]
        ctx.set_timing(False)
        # (2) mean node visits / triangle tests per ray from the instrumented build of the same kernel over the
        # full frame (exact for n == 1; for n > 1 rank 0's bands are an interleaved sample of it)
        _, cst = ctx.trace(WIDTH, HEIGHT, counting=True)
        mean_nodes = cst.node_visits / max(1, cst.closest_rays)
        mean_tris = cst.tri_tests / max(1, cst.closest_rays)
        # rays that entered the k_trace<closest> launches: survivors of the TLAS-root test, plus the secondary rays unless
        # k_tail handled bounces >= 1 (its own launch, reported under frame_kernel_ms.tail)
        closest_rays_rank0 = st.closest_rays - (st.rays_secondary if st.ms_tail > 0 else 0)
        alg_bytes = closest_rays_rank0 * (RAY_BYTES + HIT_BYTES + mean_nodes * cst.bvh_node_bytes + mean_tris * cst.bvh_tri_bytes)
        launches = max(1, st.launches_trace_closest)
        live_s = st.ms_trace_closest * 1e-3
        achieved = alg_bytes / live_s / 1e9 if live_s > 0 else 0.0
        achieved_iso = alg_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "traffic": None,
                              "kernel": "closest-hit traversal k_trace<closest> (two-level quantized BVH2, one lane per ray, persistent refill; k_trace4<closest> with --variant 1) + Moller-Trumbore; bounces >= 1 run inside k_tail when few paths survive",
                              "launches_per_frame": launches, "avg_launch_ms": st.ms_trace_closest / launches,
                              "algorithmic_bytes_per_launch": alg_bytes / launches,
                              "timing": "HIP events on the kernel's own stream, live in the timed region: mean over context 0's %d timed frames (every %d-th step); with %d frames "
                                        "in flight the kernel shares the GPU with the kernels of the other frames, so a launch lasts longer than when it runs alone" % (st.timed_frames, P, P),
                              "isolated": {"achieved": achieved_iso, "frac": achieved_iso / HBM_PEAK_GBS, "avg_launch_ms": iso_ms / launches,
                                           "timing": "median of 5 frames run one at a time right after the timed region (same process, same buffers)"},
                              "rays_per_frame_in_kernel": int(closest_rays_rank0), "mean_node_visits_per_ray": mean_nodes, "mean_tri_tests_per_ray": mean_tris,
                              "node_bytes": cst.bvh_node_bytes, "tri_bytes": cst.bvh_tri_bytes,
                              "frame_kernel_ms": {"raygen": st.ms_raygen, "trace_closest": st.ms_trace_closest, "shade": st.ms_shade,
                                                  "trace_shadow": st.ms_trace_shadow, "resolve": st.ms_resolve, "tail": st.ms_tail, "frame": st.ms_frame},
                              "note": "scene (BVH+triangles ~25 MB) and cube map (96 MiB) fit the 256 MiB Infinity Cache: HBM traffic << algorithmic bytes"}
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if os.path.exists(traffic_file):
            try:
                result["roofline"]["traffic"] = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        last = (counter[0] - 1) % P
        if args.save_image and frames[last] is not None:
            img = frames[last][:HEIGHT].cpu().numpy()
            if img.shape[-1] == 3:   # assembled multi-rank frame: RGB + the constant alpha
                img = np.concatenate([img, np.ones(img.shape[:2] + (1,), np.float32)], axis=-1)
            with open(args.save_image, "wb") as fh:
                fh.write(b"PF4\n%d %d\n-1.0\n" % (WIDTH, HEIGHT))
                fh.write(img[::-1].astype("<f4").tobytes())
        if n == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl)
        else:
            result["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
