// rt_device.h — HBM data layout shared by the host library (rt_api.cpp, bvh_build.cpp) and the
// gfx950 kernels (kernels.hip).  Everything here is plain-old-data with explicit sizes.
#pragma once
#include <stdint.h>

namespace rt {

// BVH2 interior node, 64 B = one 64-byte fetch (4 x dwordx4) per visited node.  Holds the boxes
// of BOTH children so a visit decides near/far without touching the children themselves.
//   a = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)   b = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   c = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)   child0, child1, 2 spare words
// child >= 0 : index of an interior node in the same array
// child <  0 : leaf.  BLAS: ~child = (first_tri << 3) | (count - 1), count in 1..8
//                     TLAS: ~child = instance index
// A missing child is the degenerate box lo = hi = (3e38,3e38,3e38), which no ray enters.
struct alignas(16) BvhNode {
  float a[4];
  float b[4];
  float c[4];
  int32_t child0, child1;
  int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// Triangle packet in leaf order, 48 B = 3 x dwordx4.  e1 = v1 - v0 and e2 = v2 - v0 are rounded
// once in binary32 exactly as the oracle computes them at test time.
struct alignas(16) TriPacket {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t prim;   // gl_PrimitiveID (index in the mesh's index buffer / 3)
  uint32_t pad[2];
};
static_assert(sizeof(TriPacket) == 48, "TriPacket must be 48 bytes");

// Per-instance record, 128 B.
struct alignas(16) InstanceDev {
  float w2o[12];        // gl_WorldToObjectEXT, row-major 3x4 (inverse evaluated in binary64, rounded once)
  float o2w[12];        // gl_ObjectToWorldEXT, row-major 3x4 (rt_instance::transform)
  int32_t blas_root;    // global index of the mesh's root node in blas_nodes
  uint32_t mask;        // instance mask (ray mask is 0xFF)
  int32_t custom_index; // gl_InstanceCustomIndexEXT
  uint32_t first_float; // vertexOffset of src/shader.rchit:55 (floats)
  uint32_t first_index; // 3*primitive offset of src/shader.rchit:54 (uint32s)
  uint32_t pad[3];
};
static_assert(sizeof(InstanceDev) == 128, "InstanceDev must be 128 bytes");

// Mirror of rt_uniforms / UniformStructure (104 B), passed to kernels by value.
struct UniformsDev {
  float position[4], right[4], up[4], forward[4];
  float light_position[3];
  float light_intensity;
  uint32_t max_bounce_count, samples_per_pixel, center_object_type, orbiting_object_type;
  uint32_t orbiting_object_primitive_offset, orbiting_object_vertex_offset;
};
static_assert(sizeof(UniformsDev) == 104, "UniformsDev must be 104 bytes");

struct HitRec { float t, u, v; int32_t prim, inst; };  // == rt_hit

// counters[] layout (uint32 unless noted), zeroed at frame start
enum : int {
  CNT_SHADOW = 0,        // shadow-queue tail = number of shadow rays
  CNT_QUEUE0 = 1,        // CNT_QUEUE0 + b = rays in the queue of bounce b (b = 0: primary, incl. dead pads)
  CNT_MAX_BOUNCES = 72,
  CNT_NODE_VISITS = 80,  // uint64 at [80,81]   closest-hit kernel (counting builds only)
  CNT_TRI_TESTS = 82,    // uint64 at [82,83]
  CNT_NODE_VISITS_SH = 84,  // uint64: any-hit (shadow) kernel
  CNT_TRI_TESTS_SH = 86,
  CNT_WORDS = 96
};

constexpr uint32_t SID_DEAD = 0xFFFFFFFFu;   // padding lane of the primary queue
constexpr int STACK_LDS = 24;                // per-lane traversal stack entries kept in LDS
constexpr int STACK_OVF = 40;                // spill entries per lane in HBM (never touched by sane trees)
constexpr int BLAS_MAX_DEPTH = 40;           // builder-enforced; TLAS <= 20; + 1 return marker <= 64

}  // namespace rt
