"""CPU restatement of the reference's geometry ingest.

TEST INFRASTRUCTURE ONLY — imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package.

Follows (paths relative to /root/reference):
  src/main.cpp:51-63, 1606-1626   parseFile + ObjReader (tiny_obj_loader.h v2.0.0 defaults:
                                   triangulate, real_t = float)
  src/main.cpp:1636-1654          index flatten: every index.vertex_index of every shape, in order;
                                   primitiveCount = sum of num_face_vertices.size()
  src/main.cpp:1660-1700          interleave [p.xyz n.xyz] per VERTEX (normal read at the vertex
                                   index, normal_index discarded), objects concatenated
  src/main.cpp:1706-1729          index buffer: object-local uint32 indices back to back
  src/main.cpp:245-249            glmToVulkan: row-major 3x4
  src/main.cpp:1805-1808, 2836-2844  instance transforms at t=0 / animated
  src/main.cpp:1847-1873          UniformStructure defaults and the two orbiting offsets

Pinned by tests/golden/ingest_golden.json, produced by running the reference's own vendored
tiny_obj_loader.h on the same files (oracle/ref_ingest.cpp, tests/golden/make_ingest_golden.py).

Deliberate, documented divergence: src/main.cpp:1679-1681 reads normals[3v..3v+2] out of bounds
when an OBJ has fewer `vn` than `v` records (resources/cube_scene.obj: 18 vn, 44 v) — undefined
behaviour in the reference.  Rule used here and by the product host code: when
len(normals) < len(vertices), the per-vertex normal is the `vn` referenced by the LAST face
corner (file order) that uses that vertex; vertices never referenced keep (0,0,0).  With at least
as many normals as vertices the reference's read is in bounds and is reproduced (normals[3v..3v+2]).
"""
import math
import struct

import numpy as np


class ObjData:
    def __init__(self):
        self.vertices = np.zeros(0, np.float32)  # attrib.vertices (xyz flat)
        self.normals = np.zeros(0, np.float32)  # attrib.normals (xyz flat)
        self.shapes = []  # list of dict(name, vidx[], nidx[], faces)


def _fix(i, n):
    """tinyobj index rule: 1-based positive, negative = relative to the current count."""
    i = int(i)
    if i > 0:
        return i - 1
    if i < 0:
        return n + i
    raise ValueError("zero index in OBJ face")


def parse_obj(path):
    v, vn = [], []
    n_vt = 0
    shapes = []
    cur = None

    def flush(name):
        nonlocal cur
        if cur is not None and cur["vidx"]:
            shapes.append(cur)
        cur = {"name": name, "vidx": [], "nidx": [], "faces": 0}

    flush("")
    with open(path, "r") as fh:
        for line in fh:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            k = tok[0]
            if k == "v":
                v.extend(float(x) for x in tok[1:4])
            elif k == "vn":
                vn.extend(float(x) for x in tok[1:4])
            elif k == "vt":
                n_vt += 1
            elif k == "f":
                corners = []
                for c in tok[1:]:
                    parts = c.split("/")
                    vi = _fix(parts[0], len(v) // 3)
                    ni = -1
                    if len(parts) >= 3 and parts[2] != "":
                        ni = _fix(parts[2], len(vn) // 3)
                    corners.append((vi, ni))
                # fan triangulation (all faces of the shipped resources are triangles)
                for t in range(1, len(corners) - 1):
                    for c in (corners[0], corners[t], corners[t + 1]):
                        cur["vidx"].append(c[0])
                        cur["nidx"].append(c[1])
                    cur["faces"] += 1
            elif k in ("o", "g"):
                flush(" ".join(tok[1:]))
    flush("")
    d = ObjData()
    d.vertices = np.asarray(v, dtype=np.float64).astype(np.float32)
    d.normals = np.asarray(vn, dtype=np.float64).astype(np.float32)
    d.shapes = shapes
    return d


def flatten(obj):
    """src/main.cpp:1641-1654 -> (index list uint32, primitiveCount)."""
    vidx = np.asarray([i for s in obj.shapes for i in s["vidx"]], dtype=np.uint32)
    prims = sum(s["faces"] for s in obj.shapes)
    return vidx, prims


def interleave(obj):
    """src/main.cpp:1673-1682 -> float32 array of 2*len(vertices): [px py pz nx ny nz] per vertex."""
    nv = len(obj.vertices) // 3
    out = np.zeros((nv, 6), np.float32)
    out[:, 0:3] = obj.vertices.reshape(nv, 3)
    if len(obj.normals) >= len(obj.vertices):
        out[:, 3:6] = obj.normals[:3 * nv].reshape(nv, 3)
    else:  # documented divergence (reference behaviour is undefined here)
        nrm = obj.normals.reshape(-1, 3)
        for s in obj.shapes:
            for vi, ni in zip(s["vidx"], s["nidx"]):
                if ni >= 0:
                    out[vi, 3:6] = nrm[ni]
    return out.reshape(-1)


class SceneArrays:
    """What src/main.cpp hands to the device: b3 vertex buffer, b2 index buffer, per-object ranges."""

    def __init__(self, paths):
        verts, idx, ranges = [], [], []
        ff = fi = 0
        self.objs = []
        for p in paths:
            o = parse_obj(p)
            self.objs.append(o)
            vb = interleave(o)
            ib, prims = flatten(o)
            ranges.append((ff, fi, prims))
            verts.append(vb)
            idx.append(ib)
            ff += len(vb)
            fi += len(ib)
        self.verts = np.concatenate(verts).astype(np.float32)
        self.idx = np.concatenate(idx).astype(np.uint32)
        self.ranges = ranges  # (first_float, first_index, prim_count)

    # src/main.cpp:1872-1873
    @property
    def orbiting_primitive_offset(self):
        return self.ranges[1][1] // 3 if len(self.ranges) > 1 else 0

    @property
    def orbiting_vertex_offset(self):
        return self.ranges[1][0] if len(self.ranges) > 1 else 0


# ---- transforms (column-vector convention, then glmToVulkan's row-major 3x4) -------------------

def mat_identity():
    return np.eye(4, dtype=np.float32)


def mat_translate(m, t):
    """glm::translate(m, t) = m * T(t), float32."""
    T = np.eye(4, dtype=np.float32)
    T[0:3, 3] = np.asarray(t, np.float32)
    return (m.astype(np.float32) @ T).astype(np.float32)


def mat_rotate_y(m, angle):
    """glm::rotate(m, angle, (0,1,0)) = m * R_y(angle); angle already float32 (src/main.cpp:2837)."""
    a = np.float32(angle)
    c = np.float32(math.cos(float(a)))
    s = np.float32(math.sin(float(a)))
    R = np.eye(4, dtype=np.float32)
    R[0, 0] = c
    R[0, 2] = s
    R[2, 0] = -s
    R[2, 2] = c
    return (m.astype(np.float32) @ R).astype(np.float32)


def glm_to_vulkan(m):
    """src/main.cpp:245-249: transpose then first 12 floats == rows 0..2 of the column-vector matrix."""
    return np.ascontiguousarray(m[0:3, :], dtype=np.float32).reshape(12)


def initial_transforms():
    """src/main.cpp:1805-1808."""
    return [mat_identity(), mat_translate(mat_identity(), (0, 0, 5))]


def animated_transforms(m0_prev, time_param):
    """src/main.cpp:2836-2844 for one frame at timeParam (fixed-step stand-in for wall clock)."""
    m0 = mat_rotate_y(m0_prev, np.float32(time_param * math.pi * 0.0001))
    m1 = mat_translate(mat_rotate_y(mat_translate(mat_identity(), (0, 0, -5)), np.float32(time_param * math.pi)), (0, 0, 10))
    return m0, m1


def pack_instance(transform12, custom_index, mask=0xFF, flags=0x01, mesh=0):
    """64-byte VkAccelerationStructureInstanceKHR mirror (src/main.cpp:538-551); flags 0x01 =
    VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR."""
    return struct.pack("<12fIIQ", *[float(x) for x in transform12], (custom_index & 0xFFFFFF) | (mask << 24), (0 & 0xFFFFFF) | (flags << 24), mesh)


def pack_uniforms(position=(0, 0, 20), right=(1, 0, 0), up=(0, 1, 0), forward=(0, 0, -1), light=(5, 5, 5),
                  intensity=1.0, max_bounce=63, spp=4, center_type=1, orbit_type=0, prim_offset=0, vert_offset=0):
    """src/main.cpp:1847-1873 (defaults = the reference's initialisers; w components are 1)."""
    return struct.pack("<16f3ff6I", *position, 1.0, *right, 1.0, *up, 1.0, *forward, 1.0, *light, intensity,
                       max_bounce, spp, center_type, orbit_type, prim_offset, vert_offset)
