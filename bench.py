#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + secondary + shadow) of the MI355X ray-tracing stage on BASELINE config 3:
teapot.obj (mirror) + armadillo (diffuse; a STAND-IN mesh unless resources/armadillo.obj is supplied) +
skybox_texture_sea, 1920x1080, depth 4 (maxBounceCount 3) + shadow rays, spp 4.

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N ranks itself, see launcher.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Per GPU ONE scene (geometry, BLAS, cube map) resident in HBM and --frames-in-flight frame slots on
it (rt_create_frame_slot: instances/TLAS, uniforms, ray queues, counters, stream per slot) — the reference's swapchain keeps
minImageCount + 1 frames in flight on shared buffers and acceleration structures (src/main.cpp:1203, 2597, 2967).  A step =
one frame of the hot path (raygen -> [closest-hit traversal -> shade] x 4 bounces -> any-hit shadow traversal -> resolve)
on this rank's interleaved 8-row bands, followed (N > 1) by ONE RCCL gather of the compact shards to rank 0 and the row
permutation that reassembles the frame.  The frame is fixed, so scaling is STRONG.  Rank 0 prints one JSON line.

Besides `value` (the contract: t = 0 frames, timed region of K steps) the line carries
  ms_per_frame_single   one frame at a time (SURVEY.md §8d "wall time of the trace pipeline for one frame")
  animated_ms_per_step  the reference's real loop: per step animate -> rt_set_instances(update=1) (TLAS refit) ->
                        rt_set_uniforms -> frame (src/main.cpp:2836-2861, 2901-2903), frames in flight preserved
  other_mesh            the same workload on the other stand-in mesh (N = 1 only)
"""
import argparse
import hashlib
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--gather-format", choices=["rgb32f", "rgba8"], default="rgb32f",
                    help="N > 1: what a rank sends to rank 0 per pixel: 12 bytes of RGB binary32 (default; the assembled frame is bit-identical to the single-GPU frame) "
                         "or the 4 bytes of the reference's 8-bit storage image (rt_set_param output_rgba8: a third of the xGMI traffic, DESIGN.md §9)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frame slots per GPU, each with its own stream; the reference keeps swapchainImageCount = minImageCount + 1 frames in "
                         "flight (src/main.cpp:1203, 2790, 2905-2967).  0 = auto: 4 up to two GPUs, 16 from three on — the smaller a rank's "
                         "shard the more latency-bound its kernels, and frames in flight fill the gaps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the animated leg, the single-frame latency and the second mesh (profiling runs)")
    ap.add_argument("--param", action="append", default=[], help="rt_set_param NAME=VALUE on every context (experiments), repeatable")
    ap.add_argument("--save-image", default=None, help="write the last frame as PFM (rank 0)")
    ap.add_argument("--variant", type=int, default=None, help="traversal kernel: 0 = quantized BVH2, one lane per ray (default); 1 = BVH4, four lanes per ray; 2 = 4-ary records, one lane per ray")
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg4", "cfg5"],
                    help="cfg3 (default, the headline): 1920x1080 depth 4; cfg4: 3840x2160 depth 6; cfg5: 16 instances of the armadillo BLAS, 1920x1080 depth 4")
    ap.add_argument("--mesh", default="standin", choices=["limbs", "standin"],
                    help="which stand-in replaces the missing resources/armadillo.obj: standin = the geodesic blob (default: the mesh of round 1, and by "
                         "node visits per ray the harder of the two), limbs = the non-star-shaped figure; ignored when the real file is supplied")
    ap.add_argument("--animate", action="store_true", help="make the animated loop THE timed region (value then counts animated frames); without it the animated loop is a second field")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks share cuda:0 and gather through gloo/CPU tensors: exercises rank->band mapping, gather and "
                         "reassembly where only one GPU exists (its throughput is meaningless)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: still create the RCCL process group and run the per-frame gather (to itself) — a smoke test of the N > 1 code path")
    ap.add_argument("--host", default="ranks", choices=["ranks", "multi"],
                    help="N > 1: 'ranks' = one process per GPU over torch.distributed/RCCL (started by this script when no launcher did); "
                         "'multi' = ONE process drives all N GPUs through librt_multi.so (include/rt_multi.h: RCCL called from C++, no per-step Python "
                         "on the data path besides two ctypes calls)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per pass of the pipeline (rt_trace_shard_batch: K consecutive frames, each with its own instances and uniforms, rendered with the "
                         "launches of one; one gather per batch).  0 = automatic: 1 on a single GPU (the reference's frame-by-frame loop on 4 frame slots), up to 8 "
                         "when the frame is sharded over N > 1 GPUs (a 1/N shard of ONE frame is eight launches at their latency floors)")
    ap.add_argument("--loopback", action="store_true", help="--host multi only: N LOGICAL devices on cuda:0, shards moved by device copies instead of RCCL (rehearsal on one GPU)")
    return ap.parse_args(argv)


ARGS = parse_args() if __name__ == "__main__" else None

# `python3 bench.py --gpus N` from a plain shell (no torchrun): this process — which has made no HIP or torch.cuda call, it
# has not even imported torch — starts the N ranks as fresh child processes and relays rank 0's JSON line.
if ARGS is not None and ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ and ARGS.host == "ranks":
    from vulkan_raytracing_amd.launcher import spawn_ranks
    rank_cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("RT_BENCH_RANK_CMD"):   # tests: what to start as a rank (checks the plumbing without a GPU)
        import shlex
        rank_cmd = shlex.split(os.environ["RT_BENCH_RANK_CMD"])
    sys.exit(spawn_ranks(ARGS.gpus, rank_cmd))

# A 1/N frame shard is latency-bound (a lone 1/8 shard of cfg3 takes 0.45 ms, 0.16 ms of it the slowest rays of each traversal
# launch), so a sharded run keeps more frames in flight and gives each of their streams its own hardware queue (HIP maps
# streams onto 4 by default; read at HIP start-up).  Measured on one GPU (tools/pipeline_cost.py, profiles/r02_shard_ceiling.txt),
# rank 0's shard of an 8-way split: 0.128 ms per frame with 4 slots, 0.087-0.088 with 16 slots on 16 queues
# (= 6.2 x the whole frame's 0.541 ms; 4-way: 0.180 / 0.158); a whole frame gains nothing from more than 4, so a single-GPU run keeps the defaults.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or (ARGS is not None and ARGS.gpus > 1):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vulkan_raytracing_amd import RtContext, host, tiling, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
RAY_BYTES, HIT_BYTES = 32, 20
ANIM_DT = 1.0 / 60.0    # fixed time step of the animated leg (the reference uses the wall clock, src/main.cpp:2798-2800)


def mark(msg):
    if os.environ.get("RT_BENCH_TRACE"):
        print("[bench] " + msg, file=sys.stderr, flush=True)


def kernels_sha16():
    """identifies the kernel sources a profile was taken with (profiles/*.json carry the same field)"""
    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels_beam.inc", "kernels_tile.inc", "rt_api.cpp", "bvh_gpu.hip", "bvh_build.cpp", "rt_device.h", "rt_kernels.h"):
        h.update(open(os.path.join(ROOT, "vulkan_raytracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def visible_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:   # cgroup v2 CPU quota, if the box sets one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    return n, quota


def cpu_baseline(wl, budget_s=14.0):
    """The oracle (oracle/rt_oracle.cpp, kind "port") built -O3 -march=native ON THIS BOX and timed on its host cores:
    whole frames of the same workload on every visible core, plus a 1-thread figure on a band of rows.  Checker code:
    used here ONLY as the reported CPU baseline, never by the product path."""
    from oracle import oracle as orc
    S = orc.OracleScene(native=True)
    geom, inst = wl.geometry, wl.instances
    S.set_geometry(geom.verts, geom.idx, geom.ranges)
    S.set_instances([inst[i].tobytes() for i in range(len(inst))])
    S.set_uniforms(wl.uniforms.tobytes())
    S.set_skybox(wl.sky)
    W, H = wl.width, wl.height
    visible, quota = visible_cores()
    # threads = the cores the process can really run on: min(visible cores, cgroup CPU quota rounded up) — 256 threads on a quota of
    # 16 cores oversubscribe the oracle's tile queue and measured SLOWER than 16 (41.6 vs 46.4 Mrays/s)
    usable = visible if not quota else max(1, min(visible, int(-(-quota // 1))))
    cores = int(os.environ.get("RT_CPU_THREADS", usable))
    t0 = time.time()
    _, rc = S.render(W, H, threads=cores)
    one = max(time.time() - t0, 1e-3)
    reps = int(max(1, min(400, round(budget_s * 0.6 / one))))
    rays = 0
    t0 = time.time()
    for _ in range(reps):
        _, rc = S.render(W, H, threads=cores)
        rays += int(rc.sum())
    dt = time.time() - t0
    # one thread: the 64 rows through the middle of the frame (where the meshes are), repeated for ~40 % of the budget
    y0 = (H // 2 - 32) & ~7
    t0 = time.time()
    _, rc1 = S.render(W, H, y0=y0, y1=y0 + 64, threads=1)
    one1 = max(time.time() - t0, 1e-3)
    reps1 = int(max(1, min(200, round(budget_s * 0.4 / one1))))
    rays1 = 0
    t0 = time.time()
    for _ in range(reps1):
        _, rc1 = S.render(W, H, y0=y0, y1=y0 + 64, threads=1)
        rays1 += int(rc1.sum())
    dt1 = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                           "sample": "%d x rows %d..%d of the frame (%d rays, %.1f s)" % (reps1, y0, y0 + 63, rays1, dt1)},
            "sample": "%d full %dx%d frames of the same workload (%d rays, %.1f s): oracle/rt_oracle.cpp built g++ -O3 -march=native -ffp-contract=off on this box, "
                      "its own SAH BVH, %d threads = min(%d cores visible to the process, %s)"
                      % (reps, W, H, rays, dt, cores, visible, "cgroup CPU quota %.1f rounded up" % quota if quota else "no cgroup CPU quota"),
            "visible_cores": visible, "cgroup_cpu_quota": quota}


class Rig:
    """One scene on this rank's GPU + P frame slots, their output buffers and streams; step() enqueues one frame."""

    def __init__(self, wl, P, dev, local_rank, rank, n, args, collective):
        self.wl, self.P, self.dev, self.rank, self.n, self.args, self.collective = wl, P, dev, rank, n, args, collective
        W, H = wl.width, wl.height
        root = RtContext(local_rank)
        if args.variant is not None:
            root.set_param("trace_variant", args.variant)
        wl.apply(root)                       # geometry + BLAS + cube map once per GPU; instances / uniforms of slot 0
        self.ctxs = [root] + [root.frame_slot() for _ in range(P - 1)]
        for c in self.ctxs[1:]:
            c.set_instances(wl.instances)
            c.set_uniforms(wl.uniforms)
        for c in self.ctxs:
            for kv in args.param:
                k, v = kv.split("=")
                c.set_param(k, int(v))
            if args.blocks_per_cu is not None:
                c.set_param("trace_blocks_per_cu", args.blocks_per_cu)
        band = self.band = tiling.BAND_ROWS
        rows_max = self.rows_max = tiling.max_shard_rows(H, band, n)
        # frames per pass (rt_trace_shard_batch); 1 = the frame-by-frame loop
        # (sharded: 8 per pass in a long run — 7.5 x the whole frame on 1/8 shards, profiles/r04_shard_ceiling.txt — 4 when the timed region
        # is too short to fill a pipeline of 8-frame passes)
        K = self.K = args.batch if args.batch > 0 else (1 if n == 1 else (8 if args.steps >= 64 else 4))
        if K > 1 and args.rehearse_on_one_gpu:
            K = self.K = 1
        self.last_k = [1] * P            # frames in the last pass of every slot (its statistics are sums over them)
        self.batched_before = [False] * P
        # pixel format of the shards: RGBA binary32 (default), or — sharded runs with --gather-format rgba8 — the 8-bit RGBA of the reference's storage image
        self.rgba8 = n > 1 and getattr(args, "gather_format", "rgb32f") == "rgba8"
        dt = torch.uint8 if self.rgba8 else torch.float32
        self.esize = 1 if self.rgba8 else 4
        if self.rgba8:
            for c in self.ctxs:
                c.set_param("output_rgba8", 1)
        self.shards = [torch.zeros((rows_max, W, 4) if K == 1 else (K, rows_max, W, 4), dtype=dt, device=dev) for _ in range(P)]
        # the binary32 gather carries RGB only: alpha is exactly 1.0 in every pixel (sum of spp ones divided by spp, src/shader.rgen:180-183); the 8-bit one all four bytes
        root_rank = rank == 0 and collective
        ch = self.ch = 4 if self.rgba8 else 3
        self.gathered = [torch.zeros((n, rows_max, W, ch) if K == 1 else (n, K, rows_max, W, ch), dtype=dt, device=dev) if root_rank else None for _ in range(P)]
        self.full = [torch.zeros((H, W, ch) if K == 1 else (K, H, W, ch), dtype=dt, device=dev) if root_rank else None for _ in range(P)]
        self.perm = None
        if root_rank:
            src = np.zeros(H, np.int64)
            for s in range(n):
                m = tiling.shard_row_map(H, band, s, n)
                src[m] = s * rows_max + np.arange(len(m))
            self.perm = torch.as_tensor(src, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
        self.frames = [None] * P
        self.counter = 0
        # animated leg: fixed-step clock, the reference's transforms (src/main.cpp:2836-2844)
        self.time_param = np.float32(0.0)

    def step(self, animate=False):
        a, n, rank, W, H = self.args, self.n, self.rank, self.wl.width, self.wl.height
        j = self.counter % self.P
        self.counter += 1
        c = self.ctxs[j]
        if animate:
            # src/main.cpp:2798-2800, 2836-2861, 2901-2903: advance the clock, animate both transforms, createTLAS(update = true),
            # copy the uniform block — on the slot about to be submitted (waits for THAT slot's previous frame only)
            self.time_param = np.float32(self.time_param + np.float32(ANIM_DT) * np.float32(0.1))
            c.set_instances(self.wl.animate(self.time_param), update=True)
            c.set_uniforms(self.wl.uniforms)
        with torch.cuda.stream(self.streams[j]):
            c.trace_shard(W, H, self.band, rank, n, self.shards[j].data_ptr(), self.shards[j].numel() * self.esize, self.streams[j].cuda_stream)
            if n > 1 and a.rehearse_on_one_gpu:
                self.streams[j].synchronize()
                host_shard = self.shards[j].cpu()
                parts = [torch.zeros_like(host_shard) for _ in range(n)] if rank == 0 else None
                dist.gather(host_shard, parts, dst=0)
                if rank == 0:
                    self.frames[j] = torch.cat(parts).index_select(0, self.perm.cpu()).to(self.dev)
            elif self.collective:
                self._gather(j, 1)
            else:
                self.frames[j] = self.shards[j]

    def _gather(self, j, b):
        """the ONE data-path collective of a frame (or of a pass of b frames): every rank's compact shard to rank 0, rows put back in
        frame order there.  Enqueued on the current stream, behind the shard's kernels."""
        n, rank, W = self.n, self.rank, self.wl.width
        buf = self.shards[j]
        rgb = buf if self.rgba8 else buf[..., :3].contiguous()
        dist.gather(rgb, list(self.gathered[j].unbind(0)) if rank == 0 else None, dst=0)
        if rank != 0:
            return
        if self.K == 1:
            torch.index_select(self.gathered[j].view(n * self.rows_max, W, self.ch), 0, self.perm, out=self.full[j])
            self.frames[j] = self.full[j]
        else:   # (shard, frame, row) -> (frame, shard * rows_max + row) -> frame rows
            g = self.gathered[j].permute(1, 0, 2, 3, 4).reshape(self.K, n * self.rows_max, W, self.ch)
            torch.index_select(g, 1, self.perm, out=self.full[j])
            self.frames[j] = self.full[j][b - 1]

    def step_batch(self, b, animate=False):
        """b <= K consecutive frames in ONE pass of the pipeline on the next slot (rt_set_batch + rt_trace_shard_batch), one gather for
        all of them.  Every frame has its own instances and uniforms (the animated leg advances the clock b times)."""
        a, n, rank, W, H = self.args, self.n, self.rank, self.wl.width, self.wl.height
        j = self.counter % self.P
        self.counter += 1
        c = self.ctxs[j]
        insts, unis = [], []
        for _ in range(b):
            if animate:
                self.time_param = np.float32(self.time_param + np.float32(ANIM_DT) * np.float32(0.1))
                insts.append(np.array(self.wl.animate(self.time_param)))
            else:
                insts.append(np.array(self.wl.instances))
            unis.append(self.wl.uniforms)
        c.set_batch(np.stack(insts), np.concatenate(unis), update=self.batched_before[j])
        self.batched_before[j] = True
        self.last_k[j] = b
        with torch.cuda.stream(self.streams[j]):
            buf = self.shards[j]
            c.trace_shard_batch(W, H, self.band, rank, n, buf.data_ptr(), buf.numel() * self.esize, self.streams[j].cuda_stream,
                                frame_stride_bytes=self.rows_max * W * 4 * self.esize)
            if self.collective:
                self._gather(j, b)
            else:
                self.frames[j] = buf[b - 1]

    def run_frames(self, count, animate=False):
        """`count` frames: one by one (K = 1) or in passes of up to K"""
        if self.K == 1:
            for _ in range(count):
                self.step(animate)
            return
        while count > 0:
            b = min(self.K, count)
            self.step_batch(b, animate)
            count -= b

    def single_frames(self):
        """back to one frame per pass on every slot (the measurements behind the timed regions use rt_trace_shard)"""
        self.sync()
        for c in self.ctxs:
            c.set_instances(self.wl.instances)
            c.set_uniforms(self.wl.uniforms)
        self.batched_before = [False] * self.P

    def sync(self):
        for s_ in self.streams:
            s_.synchronize()
        torch.cuda.synchronize(self.dev)
        # include/rt_api.h: a frame whose bounce kernel lost a grid barrier is rendered AGAIN by the call that collects it
        # (rt_stats.frames_rerendered) — the gather enqueued behind the first attempt has then taken an incomplete shard.  Collect every
        # slot; if ANY rank re-rendered a slot's frame, every rank repeats that slot's gather, so that the collective stays matched.
        if self.collective and not self.args.rehearse_on_one_gpu:
            again = torch.zeros(self.P, dtype=torch.int32, device=self.dev)
            for j, c in enumerate(self.ctxs):
                if c.stats().frames_rerendered:
                    again[j] = 1
            dist.all_reduce(again, op=dist.ReduceOp.MAX)
            for j in [int(x) for x in torch.nonzero(again).flatten().tolist()]:
                with torch.cuda.stream(self.streams[j]):
                    self._gather(j, self.last_k[j])
                self.streams[j].synchronize()
        if self.collective:
            dist.barrier()
            torch.cuda.synchronize(self.dev)

    def timed(self, steps, warmup, animate=False):
        self.run_frames(warmup, animate)
        self.sync()
        self.ctxs[0].stats()       # drop the warm-up frames' event times
        t0 = time.perf_counter()
        self.run_frames(steps, animate)
        self.sync()
        return time.perf_counter() - t0

    def close(self):
        for c in reversed(self.ctxs):
            c.close()


def main_multi(args):
    """--host multi: ONE process drives all N GPUs through librt_multi.so (include/rt_multi.h: a scene + P frame slots per
    device, rt_trace_shard on every device, ONE RCCL gather per frame to the first device, the de-interleave there) — two
    ctypes calls per step (rtm_trace_async / rtm_trace_wait) are all the Python on the data path.  The assembled frames stay
    in the root GPU's HBM ("host_copy" 0), as in the one-process-per-GPU mode; --loopback: N logical devices on cuda:0."""
    from vulkan_raytracing_amd import multi
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-tracing stage has no CPU path")
    n = args.gpus
    if not args.loopback and torch.cuda.device_count() < n:
        raise SystemExit("bench.py --host multi --gpus %d: only %d device(s) visible (use --loopback to rehearse on one GPU)" % (n, torch.cuda.device_count()))
    # frames per pass (rtm_set_batch: the devices render their bands of K consecutive frames with the launches of one, ONE gather per pass)
    K = args.batch if args.batch > 0 else (1 if n == 1 else (8 if args.steps >= 64 else 4))
    P = args.frames_in_flight if args.frames_in_flight > 0 else (4 if (n <= 2 or K > 1) else 16)
    if args.loopback:
        P = min(P, max(1, 16 // n))     # all logical devices share one GPU: at most 16 frame slots on it take k_tail's full grid
    res = os.path.join(ROOT, "resources")
    host.armadillo_path(res, kind=args.mesh)
    wl = workloads.make(args.workload, res, mesh=args.mesh)
    W, H = wl.width, wl.height
    m = multi.RtMulti([0] * n if args.loopback else list(range(n)), P, loopback=args.loopback)
    if args.variant is not None:
        m.set_param("trace_variant", args.variant)
    wl.apply(m)
    for kv in args.param:
        k, v = kv.split("=")
        m.set_param(k, int(v))
    m.set_param("host_copy", 1 if args.save_image else 0)
    time_param = np.float32(0.0)

    batched_before = [False] * P
    last_b = [1]

    def run(steps, animate):
        """`steps` frames, one by one (K = 1) or in passes of up to K; returns (pixels, stats) of the LAST pass and leaves its size in last_b"""
        nonlocal time_param
        pending = [0] * P
        last = None
        order = []
        done = i = 0
        while done < steps:
            j = i % P
            b = min(K, steps - done)
            if pending[j]:
                last = m.trace_wait(j, copy=False); last_b[0] = pending[j]; order.remove(j)
            if K == 1:
                if animate:
                    time_param = np.float32(time_param + np.float32(ANIM_DT) * np.float32(0.1))
                    m.set_instances(wl.animate(time_param), update=True, slot=j)
                    m.set_uniforms(wl.uniforms, slot=j)
            else:
                insts = []
                for _ in range(b):
                    if animate:
                        time_param = np.float32(time_param + np.float32(ANIM_DT) * np.float32(0.1))
                        insts.append(np.array(wl.animate(time_param)))
                    else:
                        insts.append(np.array(wl.instances))
                m.set_batch(j, np.stack(insts), np.concatenate([wl.uniforms] * b), update=batched_before[j])
                batched_before[j] = True
            m.trace_async(j, W, H)
            pending[j] = b
            order.append(j)
            done += b; i += 1
        for j in list(order):          # collect what is in flight, oldest first
            last = m.trace_wait(j, copy=False); last_b[0] = pending[j]; pending[j] = 0
        return last

    run(P, False)                      # set-up: every slot allocates its queues
    run(args.warmup, args.animate)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    px, st = run(args.steps, args.animate)
    dt = time.perf_counter() - t0      # every frame collected: rtm_trace_wait waits for the devices and for the root's gather + de-interleave
    kb = last_b[0]                     # frames in the last pass: its statistics are sums over them, its pixels kb frames back to back
    anim_ms = anim_rays = None
    if not args.no_extras and not args.animate:
        run(max(P, args.warmup), True)
        t0 = time.perf_counter()
        _, sta = run(args.steps, True)
        anim_ms = (time.perf_counter() - t0) / args.steps * 1e3
        anim_rays = sta.rays_total / last_b[0]
        m.set_instances(wl.instances)
        for j in range(P):
            batched_before[j] = False
    ms_step = dt / args.steps * 1e3
    mb = int(wl.uniforms[0]["max_bounce_count"])
    metric = "Mrays/sec (primary+secondary+shadow) at %dx%d depth %d" % (W, H, mb + 1)
    if args.workload == "cfg3":
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            pass
    result = {"metric": metric, "value": st.rays_total / kb * args.steps / dt / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "animated_ms_per_step": anim_ms, "animated_value": (anim_rays / anim_ms / 1e3) if anim_ms else None,
              "config": {"workload": wl.describe() + (" [animated loop timed]" if args.animate else ""), "mesh": wl.mesh_label,
                         "rays_per_frame": {"primary": st.rays_primary // kb, "secondary": st.rays_secondary // kb, "shadow": st.rays_shadow // kb},
                         "parallelism": "ONE host process, librt_multi.so: interleaved %d-row bands over %d %s, one scene per device, one %s per %s, "
                                        "%d frame slots in flight per device" % (tiling.BAND_ROWS, n, "logical devices on one GPU (loopback)" if args.loopback else "GPUs",
                                                                                 "device-to-device copy" if args.loopback else "RCCL gather (ncclGather in one group)",
                                                                                 "frame" if K == 1 else "pass of %d frames" % K, P),
                         "frames_in_flight": P, "frames_per_pass": K, "host": "multi"},
              "roofline": None, "cpu_baseline": None,
              "note": "roofline / cpu_baseline are reported by the N = 1 run (python3 bench.py); this line is the N-GPU throughput of the one-process host"}
    if args.save_image and px is not None:
        img = np.array(px if K == 1 else px[kb - 1], dtype=np.float32)
        with open(args.save_image, "wb") as fh:
            fh.write(b"PF4\n%d %d\n-1.0\n" % (W, H))
            fh.write(img[::-1].astype("<f4").tobytes())
    m.close()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(result), flush=True)
    os.dup2(2, 1)


def main(args):
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
    if args.gpus != n:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (start it as `python3 bench.py --gpus N`, which launches the ranks itself, "
                         "or under torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, n))
    # frame slots: 4 whole frames in flight; 1/N shards one by one want 16 (each is latency-bound), in passes of several frames 4 again
    batched = args.batch > 1 or (args.batch == 0 and n > 1 and not args.rehearse_on_one_gpu)
    P = args.frames_in_flight if args.frames_in_flight > 0 else (4 if (n <= 2 or batched) else 16)

    res = os.path.join(ROOT, "resources")
    if rank == 0:
        host.armadillo_path(res, kind=args.mesh)  # generate the stand-in once before the other ranks look for it
    if collective:
        dist.barrier()
    wl = workloads.make(args.workload, res, mesh=args.mesh)
    W, H = wl.width, wl.height
    rig = Rig(wl, P, dev, local_rank, rank, n, args, collective)
    ctx = rig.ctxs[0]

    # HIP events around the closest-hit traversal launches of slot 0's frames (every P-th frame), on their own stream.  Only that
    # kernel is bracketed inside the timed region: an event record between two kernels costs ~10 us of idle GPU (their
    # kernels otherwise run back to back), so bracketing all seven would slow every P-th frame by 50 us.
    ctx.set_timing(2)
    # set-up, not a step: one frame per slot so that every slot has its ray queues allocated before the warm-up/timed steps
    rig.run_frames(P * rig.K)
    rig.sync()
    mark("set-up frames done")
    dt = rig.timed(args.steps, args.warmup, animate=args.animate)
    mark("timed region done")
    st = ctx.stats()        # counters of slot 0's last frame + MEAN closest-hit launch time over all its timed frames (every P-th step)
    # the other kernels' live times: a short continuation of the same loop (same frames in flight) with events around every kernel
    k0 = rig.last_k[0]      # frames in slot 0's last pass: its counters are sums over them
    if not args.animate and rig.K == 1:
        ctx.set_timing(1)
        for _ in range(max(3 * P, 12)):
            rig.step()
        rig.sync()
        st_all = ctx.stats()
    else:
        st_all = st
    ctx.set_timing(2)
    last_frame = None
    if args.save_image and rank == 0 and rig.frames[(rig.counter - 1) % P] is not None:
        last_frame = rig.frames[(rig.counter - 1) % P][:H].clone()   # the last frame of THE timed region
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays = torch.tensor([st.rays_primary / k0, st.rays_secondary / k0, st.rays_shadow / k0], dtype=torch.float64, device=dev)
    if collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    rays_frame = [int(x) for x in rays.tolist()]
    total_rays = sum(rays_frame)

    # ---- the animated loop as a second timed region (every rank takes part: it contains the gather) ----------------------
    anim_ms = None
    anim_rays = 0.0
    if not args.no_extras and not args.animate:
        ctx.set_timing(False)
        anim_clock0 = np.float32(rig.time_param)      # the clock the animated leg starts from (its warm-up frames included)
        dta = rig.timed(args.steps, max(P, args.warmup), animate=True)
        ta = torch.tensor([dta], dtype=torch.float64, device=dev)
        if collective:
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        anim_ms = float(ta.item()) / args.steps * 1e3
        # the animated frames are different frames (the orbiting mesh moves through the view): their own ray count, from the last frame of every slot
        anim_rays = torch.tensor([float(sum(c.stats().rays_total / rig.last_k[j] for j, c in enumerate(rig.ctxs))) / P], dtype=torch.float64, device=dev)
        if collective:
            dist.all_reduce(anim_rays, op=dist.ReduceOp.SUM)
        anim_rays = float(anim_rays.item())
        mark("animated region done")
        # back to the t = 0 scene for the measurements below
        for c in rig.ctxs:
            c.set_instances(wl.instances)
        rig.sync()
        ctx.set_timing(1)
    rig.single_frames()

    # ---- the timed loop once more with EVERY shadow ray walked (rt_set_param dead_shadow_rays 0), informational: what the frame costs without
    # the settlement of the shadow rays whose outcome cannot change their sample (config.shadow_rays) — same frames, bit for bit
    walked_ms = None
    if n == 1 and rig.K == 1 and not args.no_extras and not args.animate and not any(p.startswith("dead_shadow_rays=") for p in (args.param or [])):
        for c in rig.ctxs:
            c.set_timing(False)
            c.set_param("dead_shadow_rays", 0)
        walked_ms = rig.timed(args.steps, max(P, args.warmup)) / args.steps * 1e3
        for c in rig.ctxs:
            c.set_param("dead_shadow_rays", 1)
        rig.run_frames(P)           # (every slot's last frame is a default one again)
        rig.sync()
        ctx.set_timing(1)
        mark("all-shadow-rays-walked region done")

    # ---- the same frames in passes of 8 (rt_trace_shard_batch), single GPU, informational: `value` stays the frame-by-frame loop ------
    batched_info = None
    if n == 1 and rig.K == 1 and not args.no_extras and not args.animate:
        KB = 8
        bufs = [torch.zeros((KB, rig.rows_max, W, 4), dtype=torch.float32, device=dev) for _ in range(P)]
        out = {}
        n_pass = max(P, (args.steps + KB - 1) // KB)
        for c in rig.ctxs:
            c.set_timing(False)          # (an event record between two kernels costs ~10 us of idle GPU)
        for leg in ("static", "animated"):
            first = [True] * P
            for phase in range(2):           # the first round allocates and warms up
                # the frames of the frame-by-frame animated leg above: its clock, behind its warm-up frames.  The instance records of all
                # passes are made BEFORE the clock starts: a pass needs its 8 frames' inputs ahead of time by definition, and the
                # host-side animation helper is not what is measured (it steps the reference's stateful animation, slowly when its clock is set back)
                tp = np.float32(anim_clock0)
                for _ in range(max(P, args.warmup)):
                    tp = np.float32(tp + np.float32(ANIM_DT) * np.float32(0.1))
                pass_insts = []
                for i in range(n_pass):
                    insts = []
                    for _ in range(KB):
                        if leg == "animated":
                            tp = np.float32(tp + np.float32(ANIM_DT) * np.float32(0.1))
                            insts.append(np.array(wl.animate(tp)))
                        else:
                            insts.append(np.array(wl.instances))
                    pass_insts.append(np.stack(insts))
                unis = np.concatenate([wl.uniforms] * KB)
                rig.sync()
                t0 = time.perf_counter()
                for i in range(n_pass):
                    j = i % P
                    rig.ctxs[j].set_batch(pass_insts[i], unis, update=not first[j]); first[j] = False
                    rig.ctxs[j].trace_shard_batch(W, H, rig.band, rank, n, bufs[j].data_ptr(), bufs[j].numel() * 4, rig.streams[j].cuda_stream)
                rig.sync()
                out[leg] = (time.perf_counter() - t0) / (n_pass * KB) * 1e3
                if os.environ.get("RT_BENCH_DEBUG"):
                    stx = rig.ctxs[0].stats()
                    sys.stderr.write("[in_passes_of_8] %s phase %d: %.4f ms per frame; slot 0's last pass: rays %d + %d + %d, tail faults %d, re-rendered %d\n" % (
                        leg, phase, out[leg], stx.rays_primary, stx.rays_secondary, stx.rays_shadow, stx.tail_faults, stx.frames_rerendered))
        rig.single_frames()
        ctx.set_timing(1)
        del bufs
        batched_info = {"frames_per_pass": KB, "ms_per_step": out["static"], "animated_ms_per_step": out["animated"], "steps": n_pass * KB,
                        "note": "rt_set_batch + rt_trace_shard_batch: 8 consecutive frames (own instances and uniforms each) per pass of the pipeline, 4 slots in flight; "
                                "the frames are bit-identical to the frame-by-frame ones (tests); not `value`: a pass needs the inputs of 8 frames ahead of time, "
                                "which a recorded animation has and an interactive camera has not; the animated figure renders the frames of animated_ms_per_step's loop "
                                "(same clock), rounded up to whole passes"}

    result = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total_rays * args.steps / dt / 1e6
        mb = int(wl.uniforms[0]["max_bounce_count"])
        # the headline workload reports BASELINE.json's own metric string; "secondary" there = every ray after the primary one,
        # i.e. bounce rays + shadow rays (config.rays_per_frame lists the classes)
        metric = "Mrays/sec (primary+secondary+shadow) at %dx%d depth %d" % (W, H, mb + 1)
        if args.workload == "cfg3":
            try:
                metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:
                pass
        result = {"metric": metric, "value": value, "unit": "Mrays/s",
                  "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "animated_ms_per_step": anim_ms,
                  "ms_per_step_all_shadow_rays_walked": walked_ms,     # the same loop with rt_set_param dead_shadow_rays 0 (config.shadow_rays)
                  "value_all_shadow_rays_walked": (total_rays / walked_ms / 1e3) if walked_ms else None,
                  "animated_value": (anim_rays / anim_ms / 1e3) if anim_ms else None,   # Mrays/s of the animated loop (mean rays of its last frames)
                  "config": {"workload": wl.describe() + (" [animated loop timed]" if args.animate else ""), "mesh": wl.mesh_label,
                             "rays_per_frame": {"primary": rays_frame[0], "secondary": rays_frame[1], "shadow": rays_frame[2]},
                             "ray_classes": "value counts every traceRayEXT-equivalent: primary + secondary (bounce) + shadow rays; 'secondary' in the metric string means both",
                             "parallelism": "interleaved %d-row bands over %d GPU(s), one scene per GPU, one RCCL gather per %s, %d frame slots in flight per GPU" % (
                                 rig.band, n, "frame" if rig.K == 1 else "pass of %d frames" % rig.K, P),
                             "gather_format": ("rgba8 (the reference's 8-bit storage image: 4 bytes per pixel)" if rig.rgba8 else "rgb32f (12 bytes per pixel)") if n > 1 else None,
                             "frames_in_flight": P, "frames_per_pass": rig.K, "in_passes_of_8": batched_info, "device": ctx.device_info,
                             "frame_batches": None if rig.K == 1 else "rt_set_batch + rt_trace_shard_batch: %d consecutive frames (own instances, camera and light each) go through one pass "
                                              "of the pipeline; a rank's 1/N shard of ONE frame is eight launches at their latency floors" % rig.K,
                             "primary_rays": "one walk per pixel (k_beam: the samples of a pixel share the camera as origin; boxes against their beam, triangles per ray: "
                                             "hit records bit-identical to one walk per ray); mean_node_visits_per_ray counts a pixel's walk once",
                             "shadow_rays": "a shadow ray is not walked when the light adds exactly nothing to its sample whether it arrives or not (surface and half vector "
                                            "face away from the light: diffuse = specular = 0, src/shader.rgen:113-128, same bits either way) - k_shade settles it; it counts in "
                                            "`value` like every traceRayEXT of the reference (rays_traversed_per_frame says how many were walked); --param dead_shadow_rays=0 walks them all",
                             "animated_loop": "per step: animate (fixed dt 1/60 s) -> rt_set_instances(update=1) = TLAS refit -> rt_set_uniforms -> frame; src/main.cpp:2836-2861, 2901-2903",
                             "kept_between_frames": "what depends on the light, the instances and the trees only, as in the reference: BLAS, TLAS, and (rt_set_param shadow_entry 2, "
                                                    "the default) the shadow rays' entry records around the light, rebuilt when the light or an instance moves — the timed "
                                                    "frames are static, so `value` has them; the animated loop moves the instances every step and never builds them "
                                                    "(animated_value); nothing that depends on the camera or on pixels is kept"}}
    # ---- roofline of the dominant kernel (closest-hit traversal), rank 0's shard -----------------
    if rank == 0:
        # (1) isolated frames: the same shard, one frame at a time on slot 0, HIP events around every kernel
        iso = []
        for _ in range(5):
            ctx.trace_shard(W, H, rig.band, rank, n, rig.shards[0].data_ptr(), rig.shards[0].numel() * rig.esize, rig.streams[0].cuda_stream)
            iso.append(ctx.stats())
        iso_ms = sorted(x.ms_trace_closest for x in iso)[len(iso) // 2]
        mark("isolated frames done")
        ctx.set_timing(False)
        # (2) mean node visits / triangle tests per ray from the instrumented build of the same kernel over the
        # full frame (exact for n == 1; for n > 1 rank 0's bands are an interleaved sample of it)
        _, cst = ctx.trace(W, H, counting=True)
        mark("counting frame done")
        mean_nodes = cst.node_visits / max(1, cst.closest_rays)
        mean_tris = cst.tri_tests / max(1, cst.closest_rays)
        # rays that entered the k_trace<closest> launches: survivors of the TLAS-root test, plus the secondary rays unless
        # k_tail handled bounces >= 1 (its own launch, reported under frame_kernel_ms.tail)
        closest_rays_rank0 = st.closest_rays - (st.rays_secondary if st_all.ms_tail > 0 else 0)
        alg_bytes = closest_rays_rank0 * (RAY_BYTES + HIT_BYTES + mean_nodes * cst.bvh_node_bytes + mean_tris * cst.bvh_tri_bytes)
        launches = max(1, st.launches_trace_closest)
        live_s = st.ms_trace_closest * 1e-3
        achieved = alg_bytes / live_s / 1e9 if live_s > 0 else 0.0
        achieved_iso = alg_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "definition": "achieved/frac = ALGORITHMIC bytes (SURVEY.md §8d: per ray 32 B ray + 20 B hit + visited nodes x node bytes + tested triangles x packet bytes) / launch "
                              "time / 8 TB/s.  It is a work rate priced as if every visited node came from HBM; the scene is cache resident, so it is NOT HBM utilisation: see "
                              "hbm_measured and limiter",
                "kernel": "closest-hit traversal of the primary rays: k_beam (two-level quantized BVH2, one lane per PIXEL: its samples share one walk, boxes against their beam, Moller-Trumbore per ray; --param pixel_beams=0: k_trace<closest>, one lane per ray, persistent refill); bounces >= 1 run inside k_tail when few paths survive",
                "launches_per_frame": launches, "avg_launch_ms": st.ms_trace_closest / launches,
                "algorithmic_bytes_per_launch": alg_bytes / launches,
                "timing": "HIP events on the kernel's own stream, live in the timed region: mean over slot 0's %d timed frames (every %d-th step); with %d frames "
                          "in flight the kernel shares the GPU with the kernels of the other frames, so a launch lasts longer than when it runs alone" % (st.timed_frames, P, P),
                "isolated": {"achieved": achieved_iso, "frac": achieved_iso / HBM_PEAK_GBS, "avg_launch_ms": iso_ms / launches,
                             "timing": "median of 5 frames run one at a time right after the timed region on slot 0 (same process, same buffers, the SAME persistent "
                                       "grid as in the timed region: the library gives each of >= 3 frame slots 3 workgroups per CU; isolated_lone_slot is the launch "
                                       "with the 5 per CU a lone slot gets)"},
                "rays_per_frame_in_kernel": int(closest_rays_rank0), "mean_node_visits_per_ray": mean_nodes, "mean_tri_tests_per_ray": mean_tris,
                "node_bytes": cst.bvh_node_bytes, "tri_bytes": cst.bvh_tri_bytes,
                "frame_kernel_ms": {"raygen": st_all.ms_raygen, "trace_closest": st_all.ms_trace_closest, "shade": st_all.ms_shade,
                                    "trace_shadow": st_all.ms_trace_shadow, "resolve": st_all.ms_resolve, "tail": st_all.ms_tail, "frame": st_all.ms_frame,
                                    "timing": "a continuation of the timed loop (%d more steps, same frames in flight) with events around EVERY kernel of slot 0's frames; the "
                                              "timed region itself brackets only the closest-hit launches (avg_launch_ms)" % max(3 * P, 12)},
                "rocprof": None, "hbm_measured": None}
        # the two traversal kernels together over WALL time: with P frames in flight their launches overlap, so the per-launch
        # figure above (duration stretched by the other frames' kernels) understates what the chip delivers
        sh_rays = cst.rays_shadow - cst.rays_shadow_untraced     # the shadow rays that are walked (the others: config.shadow_rays)
        sh_bytes = sh_rays * (48 + 16) + cst.node_visits_shadow * cst.bvh_node_bytes + cst.tri_tests_shadow * cst.bvh_tri_bytes
        cl_bytes = cst.closest_rays * (RAY_BYTES + HIT_BYTES) + cst.node_visits * cst.bvh_node_bytes + cst.tri_tests * cst.bvh_tri_bytes
        if n == 1:
            rate = (cl_bytes + sh_bytes) / (result["ms_per_step"] * 1e-3) / 1e9
            roof["both_traversal_kernels_over_wall_time"] = {
                "achieved": rate, "frac": rate / HBM_PEAK_GBS, "unit": "GB/s",
                "algorithmic_bytes_per_frame": {"closest_all_bounces": cl_bytes, "shadow": sh_bytes},
                "definition": "algorithmic bytes of ALL closest-hit and shadow traversal of one frame (same per-ray formula; shadow ray 48 B in, 16 B colour out) / ms_per_step: "
                              "what the %d overlapping frames deliver per unit of wall time, the figure comparable with north_star's '>= 60 %% of the HBM roofline in the "
                              "traversal kernel' in the configuration `value` is quoted on" % P}
        # what really bounds the kernel (profiles/r03_l1_probe.txt): the rate at which a CU's vector-memory path serves divergent
        # 16-byte lane requests — two per node visit, three per triangle test, whatever cache level they hit.  Achieved here against
        # the probe's ceiling of 1.2 (set in L2) .. 1.6 (set in L1) requests per CU and cycle.
        try:
            n_cu = int(re.search(r"CUs=(\d+)", ctx.device_info).group(1))
        except Exception:   # noqa: BLE001
            n_cu = 256
        clk_hz = torch.cuda.get_device_properties(dev).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4e9
        req = closest_rays_rank0 * (2.0 * mean_nodes + 3.0 * mean_tris)
        def _lim(ms):
            per = req / launches / (ms * 1e-3 * clk_hz * n_cu) if ms > 0 else 0.0
            return {"lane_requests_per_cu_cycle": per, "frac_of_probe_ceiling": [per / 1.6, per / 1.2]}
        roof["limiter"] = {"name": "CU vector-memory path: divergent 16-byte lane requests (2 per node visit, 3 per triangle test), served from L1 / L2 / Infinity Cache",
                           "lane_requests_per_launch": req / launches, "probe_ceiling_per_cu_cycle": [1.2, 1.6], "clock_hz": clk_hz, "cus": n_cu,
                           "live": _lim(st.ms_trace_closest / launches), "isolated": _lim(iso_ms / launches),
                           "source": "ceiling: tools/l1_probe.hip (profiles/r03_l1_probe.txt: dependent chains of random 32-byte nodes, 1.6 requests per CU cycle from "
                                     "L1, 1.2 from L2, whatever the occupancy); requests: instrumented kernel's visit counts x rays of this launch"}
        # rocprofv3 figures are attached ONLY when the committed profile was taken on this very configuration and kernel source
        tag = {"workload": args.workload, "mesh": args.mesh, "variant": args.variant or 0, "n_gpus": n, "frames_in_flight": P, "kernels_sha16": kernels_sha16()}
        # one committed profile per workload: profiles/latest_profile.json is the headline's (cfg3), the others carry their name
        prof_file = os.path.join(ROOT, "profiles", "latest_profile.json" if args.workload == "cfg3" else "latest_profile_%s.json" % args.workload)
        if os.path.exists(prof_file):
            try:
                prof = json.load(open(prof_file))
                if all(prof.get("tag", {}).get(k) == v for k, v in tag.items()):
                    k_ms = prof["k_trace_closest_avg_ms"]
                    roof["rocprof"] = {"avg_launch_ms": k_ms, "achieved": alg_bytes / launches / (k_ms * 1e-3) / 1e9,
                                       "frac": alg_bytes / launches / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "profile": prof.get("source")}
                    if prof.get("hbm_bytes_per_launch"):
                        roof["traffic"] = prof["hbm_bytes_per_launch"]
                        roof["hbm_measured"] = {"bytes_per_launch": prof["hbm_bytes_per_launch"], "GB/s": prof["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9,
                                                "frac_of_peak": prof["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "profile": prof.get("traffic_source")}
                    if prof.get("limiter"):
                        roof["limiter"]["profile"] = prof["limiter"]
                    if prof.get("k_trace_shadow_avg_ms"):
                        roof["shadow_rocprof"] = {"avg_launch_ms": prof["k_trace_shadow_avg_ms"], "hbm_bytes_per_launch": prof.get("hbm_bytes_per_launch_shadow")}
                else:
                    roof["profile_note"] = "profiles/%s was taken on another configuration or kernel source (%s): rocprof / traffic figures withheld" % (os.path.basename(prof_file), json.dumps(prof.get("tag")))
            except Exception as e:   # noqa: BLE001
                roof["profile_note"] = "profiles/%s unreadable: %r" % (os.path.basename(prof_file), e)
        # the any-hit (shadow) traversal kernel, the other large one: same formula (48 B ray in, 16 B colour out), its live launch time
        # from the all-kernels continuation of the timed loop
        sh_live_s = st_all.ms_trace_shadow * 1e-3
        sh_rate = sh_bytes / sh_live_s / 1e9 if sh_live_s > 0 else 0.0
        if n == 1:   # both traversal kernels of a frame over wall time, as requests
            req_frame = cst.node_visits * 2.0 + cst.tri_tests * 3.0 + cst.node_visits_shadow * 2.0 + cst.tri_tests_shadow * 3.0
            per = req_frame / (result["ms_per_step"] * 1e-3 * clk_hz * n_cu)
            roof["limiter"]["both_traversal_kernels_over_wall_time"] = {"lane_requests_per_frame": req_frame, "lane_requests_per_cu_cycle": per,
                                                                        "frac_of_probe_ceiling": [per / 1.6, per / 1.2]}
        roof["shadow_kernel"] = {"bound": "hbm", "achieved": sh_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sh_rate / HBM_PEAK_GBS,
                                 "kernel": "any-hit traversal k_trace<shadow> (flags 13, src/shader.rgen:66-67,111-112) + the rgen:114-129 epilogue",
                                 "avg_launch_ms": st_all.ms_trace_shadow, "algorithmic_bytes_per_launch": sh_bytes, "rays_per_frame_in_kernel": int(sh_rays),
                                 "mean_node_visits_per_ray": cst.node_visits_shadow / max(1, sh_rays), "mean_tri_tests_per_ray": cst.tri_tests_shadow / max(1, sh_rays),
                                 "timing": "HIP events, live, mean over the continuation loop's frames of slot 0 (events around every kernel)"}
        if roof.get("shadow_rocprof"):
            k_ms = roof.pop("shadow_rocprof")
            roof["shadow_kernel"]["rocprof"] = {"avg_launch_ms": k_ms["avg_launch_ms"], "achieved": sh_bytes / (k_ms["avg_launch_ms"] * 1e-3) / 1e9,
                                                "frac": sh_bytes / (k_ms["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            roof["shadow_kernel"]["traffic"] = k_ms["hbm_bytes_per_launch"]
        result["mean_node_visits_per_closest_ray"] = mean_nodes
        result["mean_node_visits_per_shadow_ray"] = cst.node_visits_shadow / max(1, sh_rays)
        # what `value` is made of: most primary rays are shaded as misses inside k_raygen (coverage mask, empty entry records, TLAS
        # boxes) and never enter a traversal kernel.  value_traversed counts only the rays that did.
        if n == 1:
            traversed = cst.closest_rays + cst.rays_shadow - cst.rays_shadow_untraced
            result["value_traversed"] = traversed / (result["ms_per_step"] * 1e-3) / 1e6
            result["rays_traversed_per_frame"] = {"closest_hit_kernels": int(cst.closest_rays), "shadow_kernel": int(cst.rays_shadow - cst.rays_shadow_untraced),
                                                  "shadow_rays_settled_in_k_shade": int(cst.rays_shadow_untraced),
                                                  "primary_rays_ended_in_raygen": int(rays_frame[0] - (cst.closest_rays - cst.rays_secondary)),
                                                  "share_of_primary_rays_ended_in_raygen": (rays_frame[0] - (cst.closest_rays - cst.rays_secondary)) / max(1, rays_frame[0]),
                                                  "definition": "value_traversed = (rays through k_trace<closest>/k_tail + rays through k_trace<shadow>) / ms_per_step, Mrays/s; "
                                                                "value counts one ray per traceRayEXT-equivalent, including the primary rays whose miss k_raygen settles and the shadow rays "
                                                                "whose outcome cannot change their sample (settled in k_shade)"}
        result["roofline"] = roof
        if last_frame is not None:
            img = last_frame.cpu().numpy()
            if img.dtype == np.uint8:   # --gather-format rgba8
                img = img.astype(np.float32) / np.float32(255.0)
            if img.shape[-1] == 3:   # assembled multi-rank frame: RGB + the constant alpha
                img = np.concatenate([img, np.ones(img.shape[:2] + (1,), np.float32)], axis=-1)
            with open(args.save_image, "wb") as fh:
                fh.write(b"PF4\n%d %d\n-1.0\n" % (W, H))
                fh.write(img[::-1].astype("<f4").tobytes())
    rig.close()
    mark("rig closed")
    if rank == 0 and not args.no_extras:
        # one frame at a time on a context of its own (no frame slots: the library then sizes its persistent grids for a lone
        # frame), wall clock of enqueue + wait, rank 0's shard
        lone = RtContext(local_rank)
        if args.variant is not None:
            lone.set_param("trace_variant", args.variant)
        wl.apply(lone)
        buf = torch.zeros((tiling.max_shard_rows(H, tiling.BAND_ROWS, n), W, 4), dtype=torch.float32, device=dev)
        s1 = torch.cuda.Stream(device=dev)
        for _ in range(4):
            lone.trace_shard(W, H, tiling.BAND_ROWS, rank, n, buf.data_ptr(), buf.numel() * 4, s1.cuda_stream)
            lone.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            lone.trace_shard(W, H, tiling.BAND_ROWS, rank, n, buf.data_ptr(), buf.numel() * 4, s1.cuda_stream)
            lone.synchronize()
        result["ms_per_frame_single"] = (time.perf_counter() - t0) / 20 * 1e3
        lone.set_timing(True)
        lone_ms = []
        for _ in range(5):
            lone.trace_shard(W, H, tiling.BAND_ROWS, rank, n, buf.data_ptr(), buf.numel() * 4, s1.cuda_stream)
            lone_ms.append(lone.stats().ms_trace_closest)
        lone_ms = sorted(lone_ms)[2] / launches
        if lone_ms > 0:
            result["roofline"]["isolated_lone_slot"] = {"achieved": alg_bytes / launches / (lone_ms * 1e-3) / 1e9, "frac": alg_bytes / launches / (lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                        "avg_launch_ms": lone_ms, "timing": "median of 5 frames one at a time on a context without frame slots (5 workgroups per CU)"}
        lone.close()
        mark("lone context done")
    if rank == 0:
        # ---- the same workload on the other stand-in mesh (one GPU only) -------------------------------------------------
        if n == 1 and not args.no_extras and not collective and args.workload in ("cfg3", "cfg5") and not os.path.exists(os.path.join(res, "armadillo.obj")):
            other = "limbs" if args.mesh == "standin" else "standin"
            wl2 = workloads.make(args.workload, res, mesh=other)
            rig2 = Rig(wl2, P, dev, local_rank, rank, n, args, False)
            for _ in range(P):
                rig2.step()
            rig2.sync()
            dt2 = rig2.timed(args.steps, args.warmup)
            st2 = rig2.ctxs[0].stats()
            _, c2 = rig2.ctxs[0].trace(W, H, counting=True)
            result["other_mesh"] = {"mesh": wl2.mesh_label, "value": st2.rays_total * args.steps / dt2 / 1e6, "unit": "Mrays/s", "ms_per_step": dt2 / args.steps * 1e3,
                                    "rays_per_frame": {"primary": st2.rays_primary, "secondary": st2.rays_secondary, "shadow": st2.rays_shadow},
                                    "mean_node_visits_per_ray": c2.node_visits / max(1, c2.closest_rays), "mean_tri_tests_per_ray": c2.tri_tests / max(1, c2.closest_rays),
                                    "mean_node_visits_per_shadow_ray": c2.node_visits_shadow / max(1, c2.rays_shadow - c2.rays_shadow_untraced),
                                    "shadow_rays_settled_in_k_shade": int(c2.rays_shadow_untraced)}
            rig2.close()
            mark("other mesh done")
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


if __name__ == "__main__":
    if ARGS.host == "multi" and ARGS.gpus > 1:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("bench.py --host multi drives every GPU from ONE process: start it without a launcher (python3 bench.py --gpus N --host multi)")
        main_multi(ARGS)
    else:
        main(ARGS)
