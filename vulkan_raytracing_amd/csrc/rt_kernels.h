// rt_kernels.h — launch interface between the host library (rt_api.cpp) and kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rt {

struct SceneDev {
  const BvhNodeQ* blas_nodes;  // variant 0: quantized BVH2 nodes of all meshes, then the quantized TLAS nodes
  int tlas_root;               // index of the TLAS root in blas_nodes
  int tlas_nodes;              // nodes of this slot's TLAS (they follow the root)
  int tlas_stride;             // frame batches: frame k's TLAS root is tlas_root + k * tlas_stride (0: no batch)
  uint32_t batch_samples;      // frame batches: sample ids per frame (spp * rows * W): frame of a ray = sample id / batch_samples (0: no batch)
  const float4* tris;          // 3 float4 per TriPacket
  float tlas_q_lo[3], tlas_q_scale[3];
  const WideNodeQ* wide_nodes; // variant 2: 4-ary records, same numbering as blas_nodes
  const Bvh4Node* nodes4;      // quad traversal (variant 1): BLAS BVH4 nodes, then the TLAS BVH4 nodes
  int tlas_root4;              // index of the TLAS root in nodes4
  const InstanceDev* inst;
  const float* verts;          // binding 3 (src/main.cpp:1305-1335)
  const uint32_t* idx;         // binding 2
  const uchar4* sky;           // binding 5: 6 layers RGBA8
  int n_inst;
  int sky_w, sky_h;
  uint32_t ovf_stride;         // spill entries per lane behind the LDS stack, sized from the depth of the trees
  const MaterialDev* materials;   // row n4: MTL material table (NULL / n_materials == 0: the reference's hard-coded constants)
  const uint32_t* prim_material;  // material of every triangle of the index buffer (global primitive number = first_index / 3 + gl_PrimitiveID)
  int n_materials;
  const float* cover_boxes;    // object-space frontier boxes of every mesh (lo[3], hi[3]), InstanceDev::cover_first/count index them
};

struct FrameDev {
  float4* ray_o[2];            // ping-pong ray queues: (o.xyz, tmax)
  float4* ray_d[2];            //                        (d.xyz, sample id bits)
  float4* hit_a;               // closest-hit records: (t, u, v, prim bits)
  int32_t* hit_inst;           //                       instance index, -1 = miss
  float4* sh_o;                // shadow queue: (o.xyz, tmax = lightDistance)
  float4* sh_d;                //               (L.xyz, sample id bits)
  float4* sh_c;                //               (tmpColor if the light is visible: rgb, material tag bits)
  float4* sample_color;        // per-sample tmpColor (rgb, 1), index = i*(rows*W) + ly*W + x
  uint32_t* counters;          // rt::CNT_* layout; zero when the frame starts
  uint32_t* counters_next;     // the context's other counter block: k_resolve (the frame's last kernel) zeroes it for the next frame
  int32_t* ovf_stack;          // SceneDev::ovf_stride ints per persistent thread
  unsigned long long* stats_out;   // host-mapped StatSlot block, written by k_resolve (NULL: not wanted)
  uint32_t* fault_total;           // the context's never-reset count of frames whose k_tail gave up (device word)
  uint32_t* hint;              // host-mapped array [CNT_MAX_BOUNCES]: size of every bounce queue, written by k_resolve and
                               // read by the host, unsynchronised, as the launch-strategy hint for the next frame
  float4* out;                 // compact shard image (rows x W RGBA32F; rows x W RGBA8 when out_rgba8 is set)
  int out_rgba8;               // 0: RGBA32F, 1: R8G8B8A8, 2: B8G8R8A8
  uint32_t shard_cap;          // entries per queue shard (queues hold N_SHARDS * shard_cap rays)
  int width, height;           // full frame
  int rows;                    // rows rendered by this shard (compact)
  int band_rows, shard, n_shards;
  // Primary-ray coverage mask of this frame (NULL: off).  Word 0 != 0: every tile may be covered; bit t of the words from 1 on:
  // 8x8-pixel tile t = ty * cover_tiles_x + tx of the FULL frame may be touched by a mesh (k_cover); k_raygen shades the samples
  // of an unmarked tile as misses without testing anything.  cover_next/cover_words: the slot's other mask, zeroed by k_resolve.
  const uint32_t* cover;
  uint32_t* cover_next;
  uint32_t cover_words;
  int cover_tiles_x;
  // Entry lists of this frame's primary rays (NULL: off): record of local tile ty * tiles_x + tx, written by k_entry, read by
  // k_raygen (empty tiles) and by the closest-hit launch of bounce 0 (where the tile's rays start their walk).
  EntryRec* entry;
  // Entry lists of the shadow rays (NULL: off): record ((face * light_tiles) + ty) * light_tiles + tx of the cube around the
  // light; k_shade gives every shadow ray its record index (sh_e), the shadow traversal starts there.
  const EntryRec* light_entry;
  uint32_t* sh_e;              // shadow queue: record index | ENTRY_REVERSE, or ENTRY_FROM_ROOT
  int light_tiles;
  int far_possible;            // LaunchCfg::far of this frame (k_raygen's TLAS test)
  int settle_dead_shadow_rays; // k_shade does not queue a shadow ray whose outcome cannot change its sample (rt_set_param "dead_shadow_rays")
  // (ux, uy) of every sample of this frame size and shard layout (k_jitter_table; NULL: k_raygen evaluates the hash itself)
  const float2* jitter;
  // frame batch (rt_device.h BATCH_MAX): `rows` stays the rows of ONE frame's shard; the buffers hold batch_k of them back to back
  int batch_k;                 // 1: a single frame
  uint32_t out_frame_stride;   // pixels between the shard images of consecutive frames of a batch in `out` (>= rows * width)
  uint32_t cover_view_words;   // words of ONE view's coverage mask (frame k's camera mask starts k * cover_view_words into `cover`)
  // Tile blobs (rt_device.h; NULL: off).  tile_blob[local tile] = arena slot of the tile's blob or BLOB_NONE (k_blob): k_raygen leaves
  // the tiles that have one to k_tile, which generates their rays itself, walks them in LDS, puts them (ray direction + hit record) at the
  // TOP of their shard's region of bounce queue 0 (Q_TILE_RAYS; k_shade reads both ends) and appends the few that may still hit
  // another instance to queue 0 proper for the global walk.
  uint32_t* tile_blob;
  char* blob_arena;
  uint32_t blob_slots;         // arena capacity (slots of BLOB_SLOT_BYTES): N_SHARDS sub-arenas of blob_slots / N_SHARDS
  uint4* blob_list;            // BLOB_CLASSES x N_SHARDS parts of blob_slots / N_SHARDS entries
  // Pixel runs (kernels_beam.inc; 0: off): bounce queue 0 is NOT compacted — every covered tile with a traced sample owns a run of 64
  // slots per sample row of its k_raygen workgroup (slot = run + 64 * row + pixel of the tile), so that k_beam finds the samples of a
  // pixel together; a slot without a ray has a zero direction and gets HIT_DEAD as its hit record.
  int pixel_runs;
  // ... and the shadow ray of a primary hit takes the slot of its primary ray in sh_o / sh_d / sh_c / sh_e (zero direction: none), for
  // k_beam_shadow; the compact shadow queue of the later bounces then starts sh_base entries up in the same arrays (0: no shadow runs)
  uint32_t shadow_runs;        // 0 / 1
  uint32_t sh_base;
};

// camera of k_cover: the inverse of the basis (right, up, forward) maps a world offset v = P - position to (a.x, a.y, a.z) with
// the primary ray through P having ux = 2.5 a.x / a.z, uy = 2.5 a.y / a.z (src/shader.rgen:74-79)
struct CoverArgs {
  float cam[3];
  float inv[9];
  float kf;                    // 2.5 for the camera (src/shader.rgen:79), 1 for a face of the light cube
  float apex_radius;           // rays pass within this distance of cam (see EntryArgs)
  int width, height, tiles_x, tiles_y;
  int n_inst;
  uint32_t mask_offset;        // words from the start of the frame's mask block to this view's mask (word 0: everything marked)
  int inst_base;               // first instance record of this view (frame batches: the view is a frame, its instances follow the previous frame's)
};
struct CoverViews { CoverArgs v[MAX_VIEWS]; int n; };

// beam of a tile (k_entry): apex cam, directions ux * R + uy * U + kf * F with (a, b, c) = inv * (P - cam) the coordinates of a
// point in the basis (R, U, F); the camera: R, U, F = right, up, forward, kf = 2.5 (src/shader.rgen:79)
struct EntryArgs {
  float cam[3];
  float inv[9];
  float basis[9];              // R, U, F (three vectors): the centre ray of a tile orders its entries
  float kf;
  float apex_radius;           // the rays pass within this distance of cam (0: through it).  Shadow rays start 0.01 N off the surface
                               // point whose direction to the light they take (src/shader.rgen:107-110), so they end 0.01 N off the light
  int width, height;           // full frame
  int tiles_x, tile_rows;      // local tile grid of this shard (= the grid of k_raygen)
  int band_rows, shard, n_shards;
  EntryRec* records;
  const uint32_t* cover;       // coverage mask of this view (tiles it leaves unmarked get an empty record)
  int cover_tiles_x;
  int tlas_root_offset;        // frame batches: this view's (frame's) TLAS root is sc.tlas_root + tlas_root_offset
};
struct EntryViews { EntryArgs v[MAX_VIEWS]; int n; };

// per-frame camera and light of a batch (kernel argument of k_raygen / k_shade / k_tail; frame 0 = the uniform block proper)
struct BatchTab {
  float position[BATCH_MAX][4], right[BATCH_MAX][4], up[BATCH_MAX][4], forward[BATCH_MAX][4];
  float light[BATCH_MAX][4];
};

struct LaunchCfg {
  int trace_blocks;            // persistent grid of the traversal kernels (256 threads each)
  int shade_blocks;
  int rays_per_lane;           // device-side grid sizing of the traversal kernels (see k_trace)
  int min_blocks;
  int variant;                 // 0: BVH2, one lane per ray; 1: BVH4, four lanes per ray; 2: 4-ary records, one lane per ray
  int far;                     // 1: a ray of this frame may be FAR from a tree it walks (kernels.hip quant_far): launch the kernels that carry the far-ray logic
  int packet;                  // variant 0 only.  1: the primary rays (bounce 0) and the shadow rays are walked by k_packet, one wavefront per
                               // 64-ray chunk; 2: the record-level entry (rt_intersect) too (tests: incoherent rays through the packet kernel)
  int packet_blocks;           // persistent grid of k_packet (no LDS, few registers: eight workgroups per CU fit)
};

size_t raygen_block_count(int width, int rows, uint32_t spp, int batch_k);   // workgroups of k_raygen (each appends <= 256 rays to one shard)
void launch_raygen(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, hipStream_t s);
// the (ux, uy) table k_raygen reads through FrameDev::jitter: jitter_table_elems float2 for (f.width, f.rows, spp) and f's shard layout
size_t jitter_table_elems(int width, int rows, uint32_t spp);
void launch_jitter_table(const FrameDev& f, uint32_t spp, float2* table, hipStream_t s);
// bounce 0 of a frame with entry lists (f.entry) starts every ray at its tile's record
void launch_trace_closest(const SceneDev& sc, const FrameDev& f, int bounce, bool counting, const LaunchCfg& cfg, hipStream_t s);
// one lane per tile of the shard: the record of every tile the coverage mask marks
void launch_entry(const SceneDev& sc, const EntryViews& a, hipStream_t s);
// one wavefront per tile of the camera view `e` (the view the records were made for): the tile's blob
void launch_blob(const SceneDev& sc, const EntryArgs& e, const FrameDev& f, bool counting, hipStream_t s);
// the tiles with a blob: their primary rays generated and walked in LDS, one launch per size class
void launch_tile(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, bool counting, hipStream_t s);
// bounces first_bounce..maxBounceCount (traversal + shading) in one launch of TAIL_BLOCKS workgroups
void launch_tail(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, int first_bounce, bool counting, const LaunchCfg& cfg, int tail_blocks, hipStream_t s);
void launch_shade(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, int bounce, const LaunchCfg& cfg, hipStream_t s);
void launch_beam_shadow(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, bool counting, const LaunchCfg& cfg, hipStream_t s);
void launch_trace_shadow(const SceneDev& sc, const FrameDev& f, bool counting, const LaunchCfg& cfg, hipStream_t s);
void launch_resolve(const FrameDev& f, const UniformsDev& u, hipStream_t s);
// marks the tiles of `mask` (FrameDev::cover layout) that the frontier boxes of the instances project onto
void launch_cover(const SceneDev& sc, const CoverViews& a, uint32_t max_boxes_per_instance, uint32_t* mask_block, hipStream_t s);
// record-level traceRayEXT on raw rays: o = (o.xyz, tmin), d = (d.xyz, tmax); writes HitRec[n]
// counters must hold the ray count in cnt_tail(0, 0) and zeros elsewhere; rays form shard 0 of capacity shard_cap
void launch_trace_raw(const SceneDev& sc, const float4* ray_o, const float4* ray_d, HitRec* out, uint32_t shard_cap,
                      int32_t* ovf_stack, uint32_t* counters, bool any_hit, bool counting, const LaunchCfg& cfg, hipStream_t s);

// de-interleave n_shards gathered compact shards (shard_stride_px pixels apart) into the width x height frame
void launch_assemble(const void* gathered, void* out, int width, int height, int band_rows, int n_shards, size_t shard_stride_px, bool rgba8, hipStream_t s);
int trace_threads_per_block();
// true in librt_mi355x_alt.so (-DRT_ALT_KERNELS): trace_variant 1 / 2 and packet_trace exist; the product library has the one-lane BVH2 kernels only
bool alt_kernels_built();
// resident workgroups of k_tail per CU (occupancy query; <= 0 on failure)
int tail_blocks_per_cu();
// host-only sizing rules (rt_api.cpp)
size_t ovf_elems(int trace_blocks, int tail_blocks, uint32_t stride);
int tail_grid(int n_cu, int resident_blocks_per_cu, int live_slots);

}  // namespace rt
