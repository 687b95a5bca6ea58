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

// Quantized BVH2 node, 32 B = TWO 16-byte requests per lane and visit instead of four.  The one-lane-
// per-ray kernel is bound by the CU's vector-memory address unit (one divergent 16-byte lane request
// per cycle), so halving the requests per visit is worth more than anything arithmetic.
// Planes are 16-bit fixed point inside the bounds of their tree (mesh bounds for a BLAS, instance
// bounds for the TLAS): plane = q_lo + q * q_scale, lower planes rounded down and upper planes
// rounded up by a full quantum, so a quantized box always contains the float box.  The slab test
// never decodes a plane: t = q * (q_scale/d) + (q_lo - o)/d, one cvt + one fma per plane.
//   w0 = c0.lo.x | c0.hi.x << 16   w1 = c0 y   w2 = c0 z   w3 = c1 x   w4 = c1 y   w5 = c1 z   w6, w7 = links
struct alignas(16) BvhNodeQ {
  uint32_t w[6];
  int32_t child0, child1;
};
static_assert(sizeof(BvhNodeQ) == 32, "BvhNodeQ must be 32 bytes");

// Wide (4-ary) view of the same tree, 64 B: record i holds the GRANDCHILDREN of BVH2 node i (a child that
// is a leaf stays as it is), in the same fixed-point space and with the same node numbering, so roots and
// links are shared with the BvhNodeQ array.  One visit tests four boxes: half as many dependent node
// fetches per ray, and the per-visit overhead (address, loop vote, stack) is paid once per four boxes.
// Axis-major so that a lane reads four dwordx4: x[k] = lo.x | hi.x << 16 of entry k, ...; an unused entry
// is the point box at quantum 0 with the link of entry 0 (a ray through that very point repeats entry 0,
// which is harmless: the tie rule makes triangle tests idempotent).
struct alignas(16) WideNodeQ {
  uint32_t x[4], y[4], z[4];
  int32_t ref[4];
};
static_assert(sizeof(WideNodeQ) == 64, "WideNodeQ must be 64 bytes");

// BVH4 node for the quad-cooperative traversal (4 lanes per ray): 128 B = one cache line, child k at
// byte 32k so that the 4 lanes of a quad read 4 consecutive 32-byte records (2 x dwordx4 each).
//   (lo.x, lo.y, lo.z, hi.x) (hi.y, hi.z, ref, 0);  ref as in BvhNode::child*; a missing child is the
//   far-point box with ref 0x7FFFFFFF.
struct alignas(16) Bvh4Child {
  float lo[3];
  float hix;
  float hiy, hiz;
  int32_t ref;
  uint32_t pad;
};
struct alignas(128) Bvh4Node {
  Bvh4Child c[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

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

// Per-instance record, 160 B.
struct alignas(16) InstanceDev {
  float w2o[12];        // gl_WorldToObjectEXT, row-major 3x4 (inverse evaluated in binary64, rounded once)
  float o2w[12];        // gl_ObjectToWorldEXT, row-major 3x4 (rt_instance::transform)
  int32_t blas_root;    // global index of the mesh's root node in blas_nodes
  uint32_t mask;        // instance mask (ray mask is 0xFF)
  int32_t custom_index; // gl_InstanceCustomIndexEXT
  uint32_t first_float; // vertexOffset of src/shader.rchit:55 (floats)
  uint32_t first_index; // 3*primitive offset of src/shader.rchit:54 (uint32s)
  int32_t blas_root4;   // root of the mesh in nodes4 (BVH4)
  float q_lo[3];        // dequantisation of the mesh's BvhNodeQ planes (object space)
  float q_scale[3];
  uint32_t type;        // per-instance object type (rt_set_instance_types) or TYPE_BY_OBJECT_INDEX: the reference's two-way switch
  uint32_t cover_first, cover_count;   // the mesh's frontier boxes in SceneDev::cover_boxes (primary-ray coverage mask, k_cover)
  uint32_t pad;
};
static_assert(sizeof(InstanceDev) == 160, "InstanceDev must be 160 bytes");

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

// Row n4: one MTL material (== rt_material, 48 B).  Without a table the kernels use the constants the reference hard-codes
// (src/shader.rgen:51-55: ka .1 .3 .1, kd .2 1 .2, ks .8, exponent 100, index of refraction 1.52).
struct MaterialDev {
  float ka[3]; float ns;
  float kd[3]; float ni;
  float ks[3]; uint32_t type;   // 0 diffuse, 1 mirror, 2 refractive, TYPE_BY_INSTANCE: the instance decides
};
static_assert(sizeof(MaterialDev) == 48, "MaterialDev must be 48 bytes");
constexpr uint32_t TYPE_BY_OBJECT_INDEX = 0xFFFFFFFFu;   // InstanceDev::type: objectIndex == 0 ? centerObjectType : orbitingObjectType
constexpr uint32_t TYPE_BY_INSTANCE = 0xFFFFFFFFu;       // MaterialDev::type
constexpr uint32_t MATERIAL_NONE = 0xFFFFFFFFu;          // shadow-queue tag: the reference's constant ambient term

// ---- queues and counters ---------------------------------------------------------------------
// Every ray queue is split into N_SHARDS regions of `shard_cap` entries (entry v = shard*shard_cap +
// slot).  A path never leaves the shard its primary ray was generated in, producers append with one
// wave-aggregated atomic on the shard's own tail, consumers pull 64-ray chunks from the shard's own
// cursor and steal from the other shards when theirs runs dry.  Blocks use shard blockIdx % 8, which
// under round-robin dispatch is their XCD: the rays a shard's producers write are read back through
// the same XCD's L2, and no single atomic word sees more than 1/8 of the traffic (one returning
// atomic word saturates near 88 operations/us on MI355X).  Placement is a speed hint, never assumed.
// Primary-ray coverage mask (k_cover): every mesh contributes the boxes of its BLAS frontier at about COVER_TARGET_BOXES (1024 measured best: 256 .. 4096 give the same frame time, 16 384 costs more in k_cover than it culls)
// nodes; a box whose screen rectangle spans more tiles than COVER_MAX_TILES marks the whole frame instead.
constexpr int COVER_TARGET_BOXES = 1024;
constexpr int COVER_MAX_TILES = 4096;
// Entry lists (k_entry): one 32-byte record per tile of a VIEW — view 0 is the camera (8x8-pixel tiles of the frame: all primary
// rays leave the camera), views 1..6 are the faces of a cube around the light (all shadow rays END at the light,
// src/shader.rgen:107-112), each face LIGHT_TILES x LIGHT_TILES tiles.  See kernels.hip.
//   w[0] = stack words (bits 0-3) | words below the instance's BLAS words, marker included (bits 4-7) | instance << 8
//   w[1] = first node to visit, w[2..] = the stack words, bottom first
constexpr int ENTRY_WORDS = 6;                       // stack words a record can hold besides the first node to visit
constexpr uint32_t ENTRY_EMPTY = 0xFFFFFFFFu;        // w[0]: no ray of the tile can hit anything
constexpr uint32_t ENTRY_NO_INST = 0xFFFFFFu;        // w[0] >> 8: the rays of the tile start in world space (TLAS words only)
constexpr uint32_t ENTRY_FROM_ROOT = 0x7FFFFFFFu;    // per-ray record index: none, the walk starts at the TLAS root
constexpr uint32_t ENTRY_REVERSE = 0x80000000u;      // per-ray flag: take the instance's subtrees far (from the view point) first
struct alignas(16) EntryRec { int32_t w[2 + ENTRY_WORDS]; };
static_assert(sizeof(EntryRec) == 32, "EntryRec must be 32 bytes");
constexpr int ENTRY_VIEWS = 7;                       // camera + 6 light faces
// Frame batches (rt_trace_shard_batch): up to BATCH_MAX consecutive frames go through ONE pass of the pipeline — a rank of an N-GPU split
// renders its bands of K frames with the launches of one frame (a 1/8 shard of one frame is eight launches at their latency floors).  The
// frames lie back to back in every buffer (frame k's samples at k * spp * rows * W, its image at k * rows * W), their instance records
// are concatenated (instance ids are global: frame k's TLAS names k * n .. k * n + n - 1) and their TLAS trees follow one another
// `tlas_stride` nodes apart in ONE quantisation; a ray finds its frame from its sample id.  In a batch each frame is one "view" of
// k_cover / k_entry.
constexpr int BATCH_MAX = 8;
constexpr int MAX_VIEWS = 8;                         // >= ENTRY_VIEWS, >= BATCH_MAX
constexpr int LIGHT_TILES_DEFAULT = 256;             // tiles per side of a light face (rt_set_param "light_tiles")
// Tile blobs (k_blob / k_tile, round 4): "BVH nodes and triangle packets staged through LDS".  For every 8x8-pixel tile whose entry
// record names an instance, k_blob continues the beam search of k_entry down to the leaves and writes what the tile's beam can touch of
// that instance's BLAS as ONE compact blob: the nodes in breadth-first order with blob-local links, then their triangle packets.  The
// workgroup that owns the tile in k_tile copies the blob into LDS with coalesced loads, generates the tile's primary rays
// (src/shader.rgen:57-79) and walks them there: a node visit is two ds_read_b128 instead of two divergent 16-byte requests to the CU's
// vector-memory path.  Arena slot of a blob (BLOB_SLOT_BYTES):
//   [0, 28)    the roots as 16-bit blob links, NEAR first (BLOB_MAX_ROOTS words)
//   [64, 112)  3 x uint32 per rest word of the record (other instances / unopened TLAS nodes, near first): its box in the TLAS
//              quantisation — what a ray is tested against before it is handed on to the global walk
//   nodes at byte BLOB_NODES_AT: BvhNodeQ whose two link words hold 16-bit BLOB LINKS — < 0x8000: node index in the blob;
//     0x8000 | (first local packet << 3) | (count - 1): a leaf; a child the beam cannot touch keeps an inverted box (never entered) and
//     the link BLOB_LINK_DONE, which is also the bottom-of-stack sentinel of the walk
//   packets at byte BLOB_TRIS_AT: TriPacket copies
// Blobs come in three size classes, walked by three instantiations of k_tile with different amounts of LDS: half of the tiles are small
// and keep six workgroups per CU, the silhouette tiles (the beam grazes the surface: hundreds of triangles in depth) need up to four
// times the room (measured on cfg3: median 149 nodes / 96 packets, 99th percentile 558 / 367; the figure with limbs 195 / 142 and
// 763 / 578).  k_blob appends (slot, sizes, tile, instance) to the list of the blob's class; a tile whose blob fits none (or finds no
// arena slot) keeps BLOB_NONE in the directory and its rays take the global walk (k_raygen -> k_trace<closest, ENTRY>).
constexpr int BLOB_CLASSES = 3;
constexpr int BLOB_CAP_NODES[BLOB_CLASSES] = {256, 384, 768};      // LDS: 8 / 12 / 24 KB of nodes,
constexpr int BLOB_CAP_TRIS[BLOB_CLASSES] = {170, 256, 512};       //      8 / 12 / 24 KB of packets (+ 11 KB of stacks); whole multiples of 256 16-byte pieces:
                                                                   //      every thread of k_tile copies the same number of pieces, unconditionally
constexpr int BLOB_MAX_NODES = BLOB_CAP_NODES[BLOB_CLASSES - 1], BLOB_MAX_TRIS = BLOB_CAP_TRIS[BLOB_CLASSES - 1];
constexpr int BLOB_REST_AT = 64;                                   // bytes
constexpr int BLOB_NODES_AT = 128;
constexpr int BLOB_TRIS_AT = BLOB_NODES_AT + BLOB_MAX_NODES * 32;
constexpr int BLOB_SLOT_BYTES = BLOB_TRIS_AT + BLOB_MAX_TRIS * 48; // 49 280 bytes per arena slot
constexpr int BLOB_MAX_ROOTS = 7;                                  // ENTRY_WORDS + 1
constexpr int BLOB_MAX_REST = 4;                                   // ENTRY_TLAS_CAP
constexpr uint32_t BLOB_LINK_LEAF = 0x8000u, BLOB_LINK_DONE = 0xFFFFu;
constexpr uint32_t BLOB_NONE = 0xFFFFFFFFu;                        // tile directory: the tile's rays take the global walk
constexpr int TILE_STACK = 22;                                     // per-lane stack rows of k_tile (16-bit entries); k_blob refuses deeper blobs
// list entry of a blob (uint4): x = arena slot, y = nodes | packets << 16, z = local tile, w = instance | roots << 24 | rest words << 28
// a ray that k_tile hands on to the global walk (it may still hit one of its record's REST words) carries in o.w:
constexpr uint32_t CONT_FLAG = 0x80000000u;                        // | surviving rest words (4 bits) << 24 | tile
constexpr int N_SHARDS = 8;
constexpr int CNT_STRIDE = 32;                 // uint32 words between cursors: one 128-byte line each
constexpr int CNT_MAX_BOUNCES = 72;            // bounce queues 0..71 (maxBounceCount <= 69)
constexpr int Q_SHADOW = CNT_MAX_BOUNCES;      // queue id of the shadow-ray queue
// pseudo-queues of the tile path (bounce 0 only).  The rays of tiles with a blob live in the SAME arrays as bounce queue 0, at the
// top end of each shard's region, growing downwards (queue 0 grows upwards; a shard never holds more than shard_cap rays in all):
constexpr int Q_TILE_RAYS = CNT_MAX_BOUNCES + 1;   // tail = rays k_tile walked (statistics; their slots at the top of the shard's region follow from the list sizes);
                                                   // work = rays k_tile handed on (they are in queue 0)
constexpr int HIT_DEAD = -2;                       // hit_inst of a slot of the tile region that holds no ray (k_shade skips it)
constexpr int Q_BLOB = CNT_MAX_BOUNCES + 2;        // tail = slots handed out in sub-arena `shard` (k_blob); work unused
constexpr int Q_BLOB_LIST = CNT_MAX_BOUNCES + 3;   // + class: tail = blobs of that class in list part `shard`
constexpr int Q_DEAD = Q_BLOB_LIST + BLOB_CLASSES;  // pixel runs (kernels_beam.inc): tail = slots of bounce queue 0 that hold no ray (statistics)
constexpr int Q_SHADOW0 = Q_DEAD + 1;               // tail = shadow rays that are not in the compact shadow queue (statistics): those of bounce 0 in their primary rays' slots (shadow runs),
                                                    // and those settled in k_shade (also counted in work of Q_DEAD); work = run cursors of k_beam_shadow
constexpr int N_QUEUES = Q_SHADOW0 + 1;
constexpr int TAIL_BLOCKS = 64;                // largest grid of k_tail (rt_api clamps it to the device: tail_grid())
constexpr int MAX_TAILS_IN_FLIGHT = 16;        // k_tail launches (frame slots) that must be co-resident on one GPU at any time
constexpr uint32_t TAIL_MAX_RAYS = 16384;      // bounces whose queue was larger in the previous frame get their own full-grid launches (k_tail then has at most one ray per lane: a 64-workgroup grid walks 26 k rays in 0.37 ms, the full grid in 0.05)
enum : int {
  CNT_NODE_VISITS = 0,     // uint64: closest-hit kernel (counting builds only)
  CNT_TRI_TESTS = 2,       // uint64
  CNT_NODE_VISITS_SH = 4,  // uint64: any-hit (shadow) kernel
  CNT_TRI_TESTS_SH = 6,    // uint64
  CNT_DIAG = 8,            // 3 x uint64 (diagnostic counting build): loop trips, busy quad-trips, wave cycles
  CNT_DIAG_SH = 16,
  CNT_BARRIER = 24,        // k_tail grid barrier (arrivals)
  CNT_FAULT = 25,          // set by k_tail when a barrier gave up (reported as RT_ERR_DEVICE)
  CNT_TILE_DIAG = 32,      // 6 x uint64 (counting builds): wave cycles of k_tile before the walk, in it, in all; until the list entry is there, the blob is in LDS, the ray is set up
  CNT_BLOB_STATS = 26,     // 5 words (counting builds): tiles with a small blob, with a large one, tiles that fit none, nodes, packets (all blobs of the frame)
  CNT_TAILS = 48
};
// Compact per-frame statistics (uint64 each) that the last kernel of a frame writes to host-mapped memory, so that
// reading a frame's counters needs no copy (a second device-to-host copy behind the pixel copy serialised the frames
// that other contexts had in flight).
enum StatSlot : int {
  STAT_QUEUE0 = 0,        // rays that entered bounce queue 0 (survivors of the TLAS test in k_raygen)
  STAT_SECONDARY = 1,     // sum of bounce queues 1..maxBounceCount
  STAT_SHADOW = 2,
  STAT_QUEUE1 = 3,        // size of bounce queue 1
  STAT_FAULT = 4,
  STAT_NODE_VISITS = 5, STAT_TRI_TESTS = 6, STAT_NODE_VISITS_SH = 7, STAT_TRI_TESTS_SH = 8,
  STAT_DIAG = 9,          // 6 values
  STAT_FAULT_TOTAL = 15,  // frames of this context whose k_tail gave up, EVER (FrameDev::fault_total): unlike STAT_FAULT it survives the
                          // statistics of later frames enqueued behind a faulted one
  STAT_BLOB = 16,         // 5 values: CNT_BLOB_STATS
  STAT_TILE_RAYS = 21,    // primary rays walked in LDS by k_tile
  STAT_CONT_RAYS = 22,    // ... of which handed on to the global walk
  STAT_TILE_DIAG = 23,    // 6 values: CNT_TILE_DIAG
  STAT_SHADOW_UNTRACED = 29,   // shadow rays settled in k_shade: their outcome cannot change the sample (counted in STAT_SHADOW as well)
  STAT_WORDS = 32
};
constexpr int CNT_WORKS = CNT_TAILS + N_QUEUES * N_SHARDS * CNT_STRIDE;
constexpr int CNT_WORDS = CNT_WORKS + N_QUEUES * N_SHARDS * CNT_STRIDE;
// tail (= number of entries) of shard `shard` of queue `queue`; chunk cursor of the launch that consumes it
constexpr inline int cnt_tail(int queue, int shard) { return CNT_TAILS + (queue * N_SHARDS + shard) * CNT_STRIDE; }
constexpr inline int cnt_work(int queue, int shard) { return CNT_WORKS + (queue * N_SHARDS + shard) * CNT_STRIDE; }

constexpr uint32_t SID_DEAD = 0xFFFFFFFFu;   // padding lane of the primary queue
#ifndef RT_LDS_INSTANCES
#define RT_LDS_INSTANCES 32
#endif
constexpr int LDS_INSTANCES = RT_LDS_INSTANCES;   // instance records (80 B each) the one-lane traversal kernels keep in LDS
constexpr int STACK4_LDS = 64;               // quad kernel: stack entries per RAY, all in LDS; the builder
                                             // verifies the worst case of every tree against it
constexpr int STACK_LDS = 24;                // per-lane traversal stack entries kept in LDS
constexpr int STACK_OVF = 40;                // smallest spill area per lane in HBM (rt_api sizes it from the tree depth)
constexpr int BLAS_MAX_DEPTH = 40;           // builder-enforced; TLAS <= 20; + 1 return marker <= 64

}  // namespace rt
