"""CPU tests of the host side above the C ABI and of the ABI surface itself (no GPU, no compute)."""
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import ingest
from tests import scenes
from vulkan_raytracing_amd import api, host, tiling

ROOT = scenes.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")


def test_c_abi_exports_every_declared_symbol():
    """librt_mi355x.so loads here (hipcc cross-compiled) and exports exactly what include/rt_api.h declares."""
    hdr = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(rt_[a-z_0-9]+)\s*\(", hdr, re.M))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    L = api.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.rt_abi_version() == 7
    assert L.rt_shard_rows(1080, 8, 0, 8) == tiling.shard_rows(1080, 8, 0, 8)


def test_multi_gpu_library_exports_every_declared_symbol():
    """librt_multi.so (host/rt_multi.cpp: one process, several GPUs, RCCL called directly) loads here and exports exactly
    what include/rt_multi.h declares; without a GPU rtm_create fails loudly."""
    hdr = open(os.path.join(ROOT, "include", "rt_multi.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(rtm_[a-z_0-9]+)\s*\(", hdr, re.M))
    declared |= set(re.findall(r"^\s*const void\*\s+(rtm_[a-z_0-9]+)\s*\(", hdr, re.M))
    from vulkan_raytracing_amd import multi
    assert declared == set(multi.EXPORTS), declared ^ set(multi.EXPORTS)
    so = os.path.join(ROOT, "vulkan_raytracing_amd", "librt_multi.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ROOT, "vulkan_raytracing_amd/librt_multi.so"], stdout=subprocess.DEVNULL)
    api.lib()                                   # librt_mi355x.so first (same search order as a host program's DT_NEEDED)
    L = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(L, name), name
    import torch
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        ids = (ctypes.c_int * 1)(0)
        L.rtm_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
        L.rtm_last_error.restype = ctypes.c_char_p
        L.rtm_last_error.argtypes = [ctypes.c_void_p]
        assert L.rtm_create(ctypes.byref(h), 1, ids, 2, 0) in (3, 5) and not h.value
        assert b"no HIP device" in L.rtm_last_error(None) or b"HIP" in L.rtm_last_error(None)


def test_shipped_traversal_kernels_keep_their_register_budget():
    """the traversal kernels sit exactly on the register budget of 5 waves per SIMD (what the LDS admits), so a
    neutral-looking edit can push them into scratch or down to 4 waves.  `make resource-usage` (hipcc -Rpass-analysis=kernel-
    resource-usage, a cross-compile: no GPU) reports every kernel of the product TU; the instantiations the BASELINE frames launch
    — k_trace<.., FAR = false> for the closest-hit and the shadow rays, with and without entry records — must keep 5 waves per SIMD
    with no scratch and no spills; the pixel-beam kernel of the primary rays (k_beam: four rays per lane) 4 waves; the tile kernels (k_tile: their occupancy is set by the LDS of their size class) no scratch; the far-ray and record-level instantiations 4 waves at least, k_tail
    (few rays, shading fused in) 3 — rt::tail_grid sizes its grid from the occupancy query anyway.
    The product TU must also not contain the alternatives that measured slower (k_packet, k_trace4, 4-ary records)."""
    out = subprocess.run(["make", "-C", ROOT, "resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    kernels = {}
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    names = "\n".join(kernels)
    assert "k_packet" not in names and "k_trace4" not in names, "alternative kernels in the product TU"
    # k_trace<MODE, ANY, WIDE, ENTRY, FAR, CONT>: _ZN2rt7k_traceILi<MODE>ELb<ANY>ELb<WIDE>ELb<ENTRY>ELb<FAR>ELb<CONT>EEEvNS_9TraceArgsE
    hot, other = [], []
    seen_beam = False
    for name, r in kernels.items():
        m = re.match(r"_ZN2rt7k_traceILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEE", name)
        if m:
            mode, _, wide, _, far, _ = (int(x) for x in m.groups())
            assert wide == 0, name
            (hot if (far == 0 and mode != 2) else other).append((name, r))
        elif re.match(r"_ZN2rt6k_tailILb0E", name):      # (the shipped, non-counting instantiation)
            other.append((name, r))
        elif re.match(r"_ZN2rt6k_tileILi\dELb0E", name):
            assert int(r["ScratchSize"]) == 0 and int(r["VGPRs Spill"]) == 0 and int(r["Occupancy"]) >= 2, (name, r)
        elif name == "_ZN2rt6k_beamENS_9TraceArgsEj":
            # the pixel-beam kernel carries four rays per lane: 4 waves per SIMD (its 32 KB of LDS admit no more than 5 workgroups per CU
            # anyway); a loop-invariant spilled before the run loop and reloaded once per run is tolerated, nothing inside the walk
            seen_beam = True
            assert int(r["Occupancy"]) >= 4 and int(r["ScratchSize"]) <= 16, (name, r)
    assert len(hot) >= 5 and seen_beam, names
    for name, r in hot:
        assert int(r["Occupancy"]) >= 5 and int(r["ScratchSize"]) == 0 and int(r["VGPRs Spill"]) == 0 and int(r["SGPRs Spill"]) == 0, (name, r)
    for name, r in other:
        if "k_tail" in name:
            assert int(r["Occupancy"]) >= 3, (name, r)
        else:
            assert int(r["Occupancy"]) >= 4 and int(r["ScratchSize"]) <= 32, (name, r)


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtError) as e:
        api.RtContext(0)
    assert e.value.code in (3, 5)


def test_product_never_imports_oracle():
    """The product path may mention the oracle in comments, but never imports, includes, links or loads it."""
    pat = re.compile(r"(^\s*(import|from)\s+oracle\b)|(#include\s*[<\"][^>\"]*oracle)|librt_oracle\.so|\borc_[a-z_]+\s*\(", re.M)
    files = []
    for dirpath, _, fs in os.walk(os.path.join(ROOT, "vulkan_raytracing_amd")):
        files += [os.path.join(dirpath, f) for f in fs if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp"))]
    for d in ("include", "host"):
        files += [os.path.join(ROOT, d, f) for f in os.listdir(os.path.join(ROOT, d))]
    files.append(os.path.join(ROOT, "Makefile"))
    for f in files:
        txt = open(f).read()
        if f.endswith("Makefile"):
            assert "librt_oracle" not in txt.split("oracle:")[0]
            continue
        assert not pat.search(txt), f


def test_obj_loader_matches_reference_loader_golden(resources):
    """include/obj_loader.h (through rthost::loadScene) == the reference's vendored tiny_obj_loader.h."""
    g = json.load(open(os.path.join(GOLD, "ingest_golden.json")))["obj"]
    for name in ("cube", "teapot", "cube_scene"):
        s = host.SceneGeometry([os.path.join(resources, name + ".obj")])
        assert hashlib.sha256(s.idx.tobytes()).hexdigest() == g[name]["vidx.u32"]["sha256"]
        v = s.verts.reshape(-1, 6)
        assert hashlib.sha256(np.ascontiguousarray(v[:, 0:3]).tobytes()).hexdigest() == g[name]["vertices.f32"]["sha256"]
        if g[name]["normals.f32"]["count"] == g[name]["vertices.f32"]["count"]:  # reference behaviour defined
            assert hashlib.sha256(np.ascontiguousarray(v[:, 3:6]).tobytes()).hexdigest() == g[name]["normals.f32"]["sha256"]
        assert s.ranges[0][2] == g[name]["vidx.u32"]["count"] // 3
        o = ingest.SceneArrays([os.path.join(resources, name + ".obj")])
        assert np.array_equal(o.verts, s.verts) and np.array_equal(o.idx, s.idx)


def test_two_object_offsets(resources):
    s = host.SceneGeometry([os.path.join(resources, "teapot.obj"), os.path.join(resources, "cube.obj")])
    assert s.ranges == [(0, 0, 2256), (7212, 6768, 12)]
    assert (s.orbiting_primitive_offset, s.orbiting_vertex_offset) == (2256, 7212)  # src/main.cpp:1872-1873


def test_obj_loader_polygons_and_indices(tmp_path):
    p = tmp_path / "q.obj"
    p.write_text("v 0 0 0\nv 2 0 0\nv 2 1 0\nv 0 1 0\nv 1 2 0\nvn 0 0 1\n"
                 "f 1//1 2//1 3//1 4//1\n"          # quad, equal diagonals -> (0,1,3),(1,2,3)
                 "f -5 -4 -3\n"                     # negative indices
                 "f 1 2 3 5 4\n")                   # pentagon -> 3 triangles
    s = host.SceneGeometry([str(p)])
    assert s.ranges[0][2] == 2 + 1 + 3
    assert list(s.idx[:9]) == [0, 1, 3, 1, 2, 3, 0, 1, 2]
    assert sorted(set(s.idx[9:].tolist())) == [0, 1, 2, 3, 4]
    q = tmp_path / "bad.obj"
    q.write_text("v 0 0 0\nf 0 1 2\n")
    with pytest.raises(RuntimeError):
        host.SceneGeometry([str(q)])
    with pytest.raises(RuntimeError):
        host.SceneGeometry([str(tmp_path / "missing.obj")])


def test_more_normals_than_vertices_reads_normals_at_the_vertex_index(tmp_path):
    """src/main.cpp:1679-1681 reads normals[3v..3v+2] whatever the faces say.  With MORE `vn` than `v` that read is in bounds
    and must be reproduced as is (the 'last face corner' rule is only for the undefined FEWER-vn case)."""
    p = tmp_path / "m.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 1 0 0\nvn 0 1 0\nvn 0 0 1\nvn 0 0 -1\nvn 0.6 0.8 0\nf 1//5 2//4 3//4\n")
    s = host.SceneGeometry([str(p)])
    v = s.verts.reshape(-1, 6)
    assert np.array_equal(v[:, 3:6], np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32))
    obj = ingest.parse_obj(str(p))
    assert np.array_equal(ingest.interleave(obj), s.verts)


def test_jpeg_decoder_matches_reference_decoder_golden(resources):
    """host/jpeg_decode.cpp == stbi_load(..., STBI_rgb_alpha) byte for byte on all shipped faces
    (baseline 4:2:0 `sea`, progressive 4:4:4 `test`)."""
    g = json.load(open(os.path.join(GOLD, "ingest_golden.json")))["jpg"]
    for sky in ("skybox_texture_test", "skybox_texture_sea"):
        for f in host.SKYBOX_FACES:
            a = host.decode_jpeg(os.path.join(resources, sky, f + ".jpg"))
            k = g[sky + "/" + f]
            assert a.shape == (k["h"], k["w"], 4)
            assert hashlib.sha256(a.tobytes()).hexdigest() == k["sha256"], (sky, f)
    with pytest.raises(RuntimeError):
        host.decode_jpeg(os.path.join(resources, "teapot.obj"))


def test_camera_matches_reference_formulas():
    """src/camera.cpp:8-25, 66-106."""
    c = host.Camera()
    v = c.vectors()
    assert np.allclose(v["position"], [0, 0, 20]) and np.allclose(v["front"], [0, 0, -1], atol=1e-6)
    assert np.allclose(v["right"], [1, 0, 0], atol=1e-6) and np.allclose(v["up"], [0, 1, 0], atol=1e-6)
    assert abs(v["front"][0]) > 0  # cos(-pi/2) in float is ~ -4.4e-8, not 0 (SURVEY.md §8c trap 7)
    c.move(host.FORWARD, 2.0); c.move(host.RIGHT, 1.0); c.move(host.UP, 0.5)
    assert np.allclose(c.vectors()["position"], [1, 0.5, 18], atol=1e-5)
    c.process_mouse_movement(0.3, 5.0)
    v = c.vectors()
    yaw, pitch = -np.pi / 2 + 0.3, 1.57  # pitch clamp
    f = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), np.sin(yaw) * np.cos(pitch)])
    assert np.allclose(v["front"], f, atol=1e-5)
    r = np.array([-f[2], 0, f[0]]); r /= np.linalg.norm(r)
    assert np.allclose(v["right"], r, atol=1e-5) and np.allclose(v["up"], np.cross(r, f), atol=1e-5)
    c.look(host.LEFT)
    assert np.allclose(c.vectors()["front"], [-1, 0, 0])
    u = host.default_uniforms()
    host.Camera((1, 2, 3)).to_uniforms(u)
    assert np.allclose(u[0]["position"], [1, 2, 3, 1]) and u[0]["max_bounce_count"] == 63 and u[0]["samples_per_pixel"] == 4


def test_animation_matches_restatement():
    """src/main.cpp:1805-1808 and :2836-2844 vs the numpy restatement in oracle/ingest.py."""
    a = host.SceneAnimation()
    t = a.transforms()
    assert np.array_equal(t[0], ingest.glm_to_vulkan(ingest.mat_identity()))
    assert np.array_equal(t[1], ingest.glm_to_vulkan(ingest.mat_translate(ingest.mat_identity(), (0, 0, 5))))
    m0 = ingest.mat_identity()
    for k in range(1, 4):
        a.animate(0.1 * k)
        m0, m1 = ingest.animated_transforms(m0, 0.1 * k)
        t = a.transforms()
        assert np.allclose(t[0], ingest.glm_to_vulkan(m0), atol=1e-6) and np.allclose(t[1], ingest.glm_to_vulkan(m1), atol=1e-5)
    inst = a.instances()
    assert inst.dtype.itemsize == 64 and inst[1]["custom_index_and_mask"] == (1 | (0xFF << 24)) and inst[1]["mesh"] == 1


def test_standin_mesh_is_deterministic_and_closed(tmp_path):
    p = str(tmp_path / "s.obj")
    assert host.hlib().rth_write_armadillo_standin(p.encode(), 6) == 0
    s = host.SceneGeometry([p])
    nv, nt = len(s.verts) // 6, s.ranges[0][2]
    assert (nv, nt) == (10 * 36 + 2, 20 * 36)
    # closed manifold: every edge shared by exactly two triangles
    tri = s.idx.reshape(-1, 3)
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert np.all(counts == 2)
    nrm = s.verts.reshape(-1, 6)[:, 3:6]
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-3)
    assert (np.sum(nrm * s.verts.reshape(-1, 6)[:, 0:3], axis=1) > 0).all()  # outward
    p2 = str(tmp_path / "s2.obj")
    host.hlib().rth_write_armadillo_standin(p2.encode(), 6)
    assert open(p).read() == open(p2).read()


def test_limbs_standin_is_closed_deterministic_and_not_star_shaped(tmp_path):
    """host/standin_limbs.cpp: the second armadillo stand-in (implicit figure + surface nets).  Closed manifold, unit normals
    that point out of the surface, deterministic, the size of the Stanford armadillo at the default resolution — and, unlike
    the geodesic blob, NOT star-shaped: rays cross many surfaces."""
    from oracle import oracle
    p = str(tmp_path / "l.obj")
    n = host.hlib().rth_write_armadillo_limbs(p.encode(), 96)
    s = host.SceneGeometry([p])
    assert n == s.ranges[0][2] and 25000 < n < 45000
    tri = s.idx.reshape(-1, 3)
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    # closed: every edge is shared by an even number of triangles — two, except for the handful of edges where two
    # sheets of the surface pinch inside one grid cell (surface nets keeps those as 4-fold edges)
    assert np.all(counts % 2 == 0) and (counts == 2).mean() > 0.995
    v = s.verts.reshape(-1, 6)
    assert np.allclose(np.linalg.norm(v[:, 3:6], axis=1), 1.0, atol=1e-3)
    # winding agrees with the vertex normals (both point out of the figure)
    fn = np.cross(v[tri[:, 1], :3] - v[tri[:, 0], :3], v[tri[:, 2], :3] - v[tri[:, 0], :3])
    assert (np.sum(fn * v[tri[:, 0], 3:6], axis=1) > 0).mean() > 0.995
    p2 = str(tmp_path / "l2.obj")
    host.hlib().rth_write_armadillo_limbs(p2.encode(), 96)
    assert open(p).read() == open(p2).read()
    # surface crossings along rays through the figure (closest hit, advance, repeat): a star-shaped blob gives 2 on
    # rays through its centre; the figure gives 4 and more where limbs, claws and tail overlap
    S = oracle.OracleScene()
    S.set_geometry(s.verts, s.idx, s.ranges)
    inst = np.zeros(1, scenes.INSTANCE_DTYPE)
    inst[0] = host.make_instance(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), 0, 0)
    S.set_instances([inst[0].tobytes()])
    rays = scenes.random_rays(3000, seed=3, origin_radius=12.0, target_radius=2.5)
    crossings = np.zeros(len(rays), int)
    live = np.ones(len(rays), bool)
    for _ in range(12):
        h = S.intersect(rays, use_bvh=True)
        hit = live & (h["inst"] >= 0)
        crossings += hit
        live = hit
        rays[:, 3] = np.where(hit, h["t"] + 1e-3, rays[:, 3])
    assert crossings.max() >= 6 and (crossings >= 4).mean() > 0.10
    assert (crossings % 2 == 0).mean() > 0.99                   # closed surface: a ray that enters leaves (two crossings closer than the 1e-3 step are counted once)
    # the default resolution gives a mesh of the Stanford armadillo's size (345 944 triangles)
    arm, label = host.armadillo_path(os.path.join(ROOT, "resources"), kind="limbs")
    assert "345168 triangles" in label


def test_sizing_rules_of_the_tail_kernel_and_spill_stacks():
    """rt_debug_sizing: the k_tail grid never exceeds 1/16 of the workgroups a device can hold (16 frame slots in flight
    stay co-resident), is a multiple of the 8 queue shards, is 0 (= not used) on a device too small for that; the spill-stack
    allocation covers the LARGER of the traversal grid and the tail grid."""
    L = api.lib()

    def sizing(n_cu, per_cu, trace_blocks, stride):
        out = np.zeros(2, np.uint64)
        assert L.rt_debug_sizing(n_cu, per_cu, trace_blocks, stride, out.ctypes.data_as(ctypes.c_void_p)) == 0
        return int(out[0]), int(out[1])

    assert sizing(256, 5, 1024, 40) == (64, 1024 * 256 * 40)     # MI355X: 64 workgroups, stacks for the 1024-block traversal grid
    assert sizing(256, 2, 1024, 40)[0] == 32
    assert sizing(32, 4, 128, 40) == (8, 128 * 256 * 40)         # a 32-CU partition
    assert sizing(8, 4, 8, 48) == (0, 8 * 256 * 48)              # too small: k_tail off
    g, elems = sizing(64, 8, 16, 40)                             # traversal grid smaller than the tail grid
    assert g == 32 and elems == 32 * 256 * 40
    for n_cu in (1, 7, 20, 64, 80, 256, 304):
        for per_cu in (0, 1, 3, 5, 8):
            g, _ = sizing(n_cu, per_cu, 4 * n_cu, 40)
            assert g % 8 == 0 and g <= 64 and g * 16 <= n_cu * per_cu


def test_band_sharding_covers_frame_once():
    for H, band, n in ((1080, 8, 1), (1080, 8, 8), (1080, 8, 3), (203, 8, 2), (7, 8, 4), (2160, 8, 8)):
        seen = np.zeros(H, int)
        for s in range(n):
            m = tiling.shard_row_map(H, band, s, n)
            assert len(m) == tiling.shard_rows(H, band, s, n)
            seen[m] += 1
        assert np.all(seen == 1)
        shards = []
        full = np.random.default_rng(0).random((H, 5, 4)).astype(np.float32)
        rows_max = tiling.max_shard_rows(H, band, n)
        for s in range(n):
            buf = np.zeros((rows_max, 5, 4), np.float32)
            m = tiling.shard_row_map(H, band, s, n)
            buf[: len(m)] = full[m]
            shards.append(buf)
        assert np.array_equal(tiling.assemble(shards, H, 5, band), full)


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from vulkan_raytracing_amd import tiling
from oracle import oracle, ingest
from tests import scenes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
W, H, band = 96, 54, 8
sp = scenes.two_object_scene(os.path.join(scenes.RES, "teapot.obj"), os.path.join(scenes.RES, "cube.obj"), 1, 0, 1, 1, sky=scenes.synthetic_skybox(16))
# each rank renders ONLY its bands (the oracle stands in for the GPU renderer in this CPU rehearsal)
rows = tiling.shard_row_map(H, band, rank, world)
rows_max = tiling.max_shard_rows(H, band, world)
shard = torch.zeros((rows_max, W, 4))
for y in rows.reshape(-1, 1) if len(rows) else []:
    img, _ = sp.orc.render(W, H, y0=int(y[0]), y1=int(y[0]) + 1, threads=1)
    shard[int(np.where(rows == y[0])[0][0])] = torch.from_numpy(img[int(y[0])])
gl = [torch.zeros_like(shard) for _ in range(world)] if rank == 0 else None
dist.gather(shard, gl, dst=0)
if rank == 0:
    out = tiling.assemble([g.numpy() for g in gl], H, W, band)
    full, _ = sp.orc.render(W, H, threads=2)
    assert np.array_equal(out, full), "gathered frame differs from the single-process frame"
    print("GLOO_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gather_reassembles_frame_gloo(tmp_path):
    """World-size-2 rehearsal of bench.py's N > 1 path on CPU: band sharding -> gather -> assemble."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT, "port": port})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen(["python", str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]


def test_host_bvh_builders_invariants(resources, tmp_path):
    """Binned-SAH BVH2, its BVH4 collapse and the 32-byte quantized nodes (csrc/bvh_build.cpp) — host code, checked here
    without a GPU through rt_debug_check_builders: every triangle in exactly one leaf, children inside parents,
    quantized boxes containing the float boxes, leaves <= 4, depth <= 40, traversal stack within the LDS budget."""
    from vulkan_raytracing_amd import api
    meshes = {n: host.SceneGeometry([os.path.join(resources, n + ".obj")]) for n in ("cube", "cube_scene", "teapot")}
    p = str(tmp_path / "s.obj")
    assert host.hlib().rth_write_armadillo_standin(p.encode(), 24) == 0     # 11 520 triangles
    meshes["standin24"] = host.SceneGeometry([p])
    # degenerate input: many coincident and zero-area triangles
    rng = np.random.default_rng(3)
    v = np.zeros((300, 6), np.float32); v[:, :3] = np.repeat(rng.normal(size=(30, 3)).astype(np.float32), 10, axis=0)
    tri = rng.integers(0, 300, size=(500, 3)).astype(np.uint32)
    for name, (verts, idx) in {**{k: (g.verts, g.idx) for k, g in meshes.items()}, "degenerate": (v.reshape(-1), tri.reshape(-1))}.items():
        rc, st = api.check_builders(verts, idx)
        assert rc == 0 and st["violations"] == 0, (name, st)
        assert st["reached"] == len(idx) // 3 and st["max_leaf"] <= 4 and st["depth"] <= 41, (name, st)
        assert 2 + st["bvh4_stack_need"] + 3 <= 64, (name, st)
    rc, st = api.check_builders(meshes["teapot"].verts, meshes["teapot"].idx)
    assert st["leaves"] >= 2256 // 4 and st["nodes"] < 2256


def test_threaded_host_builder_makes_the_same_tree(tmp_path, monkeypatch):
    """csrc/bvh_build.cpp splits nodes of >= 16 384 references on all threads (chunked binning, stable two-pass partition) and hands the
    subtrees below to the same threads as tasks.  Split decisions depend only on the SETS on either side of a split, so the tree must
    not depend on the thread count: same node / leaf counts, depth and BVH4 collapse, no invariant violated, for 1, 3 and 8 threads."""
    from vulkan_raytracing_amd import api
    p = str(tmp_path / "s.obj")
    assert host.hlib().rth_write_armadillo_standin(p.encode(), 64) == 0     # 81 920 triangles: the root and its first levels take the threaded path
    g = host.SceneGeometry([p])
    monkeypatch.setenv("RT_BVH_MAX_LEAF", "1")
    seen = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RT_BUILD_THREADS", threads)
        rc, st = api.check_builders(g.verts, g.idx)
        assert rc == 0 and st["violations"] == 0 and st["reached"] == 81920 and st["max_leaf"] == 1, (threads, st)
        seen.append(st)
    assert seen[0] == seen[1] == seen[2], seen


def test_mtl_files_are_parsed(resources):
    """The loader reads the MTL files the reference ships (the renderer, like the reference's, then ignores them)."""
    import ctypes as C
    # exercised through the C++ loader in librt_host.so: loading an OBJ with mtllib must not fail or warn about a missing file
    g = host.SceneGeometry([os.path.join(resources, "teapot.obj")])
    assert g.ranges[0][2] == 2256
    txt = open(os.path.join(resources, "teapot.mtl")).read()
    assert "newmtl teapot" in txt and "Ni 1.450000" in txt


_FAKE_RANK = r"""
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "ok":
    print(json.dumps({"rank": rank, "n_gpus": world, "argv": sys.argv[2:]}), flush=True)   # only rank 0's line may reach the caller
elif mode == "fail1":
    if rank == 1:
        sys.exit(7)
    time.sleep(60)          # the survivors must be ended by the launcher, not waited for
elif mode == "hang":
    time.sleep(60)
"""


def test_launcher_starts_ranks_relays_rank0_and_propagates_failures(tmp_path):
    """`python3 bench.py --gpus N` from a plain shell: vulkan_raytracing_amd/launcher.py starts N fresh rank processes with the
    torch.distributed environment, hands through the command line, relays ONLY rank 0's stdout, returns the first non-zero
    child status and ends the other ranks (exact PIDs).  The launcher itself never imports torch or a HIP library."""
    import io
    from vulkan_raytracing_amd import launcher
    src = open(launcher.__file__).read()
    assert "import torch" not in src and "ctypes" not in src and "hip" not in src.replace("HIP", "").replace("shipped", "")
    fake = tmp_path / "rank.py"
    fake.write_text(_FAKE_RANK)

    class Sink(io.StringIO):
        def fileno(self):
            raise OSError("no fd")
    out, err = Sink(), Sink()
    rc = launcher.spawn_ranks(3, [sys.executable, str(fake), "ok", "--gpus", "3", "--steps", "5"], out=out, err=err)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"rank": 0, "n_gpus": 3, "argv": ["--gpus", "3", "--steps", "5"]}
    t0 = time.time()
    out2, err2 = Sink(), Sink()
    assert launcher.spawn_ranks(3, [sys.executable, str(fake), "fail1"], out=out2, err=err2) == 7
    # a failed job relays NO result line (rank 0 may have printed half of one) and names the failing rank on err
    assert out2.getvalue() == "" and "rank 1 exited with status 7" in err2.getvalue()
    assert time.time() - t0 < 30, "the launcher waited for ranks it should have ended"
    assert launcher.spawn_ranks(2, [sys.executable, str(fake), "hang"], out=Sink(), err=Sink(), timeout_s=1.0) == 124
    e0, e1 = launcher.rank_env(0, 2, 1234, base={}), launcher.rank_env(1, 2, 1234, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert (e0["RANK"], e0["WORLD_SIZE"], e0["MASTER_PORT"], e0["HSA_ENABLE_IPC_MODE_LEGACY"]) == ("0", "2", "1234", "0")
    assert e1["LOCAL_RANK"] == "1" and e1["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"     # an explicit setting is kept


def test_bench_parent_launches_ranks_before_any_gpu_use(tmp_path):
    """bench.py --gpus 2 without WORLD_SIZE must hand over to the launcher BEFORE importing torch (the parent may never touch
    the GPU); checked here by running it with a stand-in interpreter environment in which the ranks only echo their environment:
    RT_BENCH_RANK_CMD replaces the rank command line."""
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=dict(os.environ, RT_BENCH_RANK_CMD=sys.executable + " -c \"import os,sys;print(os.environ['RANK']+'/'+os.environ['WORLD_SIZE']);sys.exit(3 if os.environ['RANK']=='1' else 0)\""),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    # rank 1 failed: the job has no result — nothing of rank 0's output is relayed as one, the cause is named on stderr
    assert r.stdout.strip() == "" and "rank 1 exited with status 3" in r.stderr
    assert "| torch" not in r.stderr and "torch.cuda" not in r.stderr, "the launching parent imported torch"
