// rt_api.cpp — the C ABI of include/rt_api.h: context, HBM residency, acceleration-structure
// management and the per-frame kernel pipeline.  Host counterpart of the Vulkan plumbing the
// reference inlines in main() (src/main.cpp:305-793 BLAS/TLAS, :1660-1729 buffers, :1847-1889 UBO,
// :2061-2412 cube map, :2620-2624 vkCmdTraceRaysKHR).  There is NO CPU fallback: without a HIP
// device every entry point fails with RT_ERR_NO_DEVICE / RT_ERR_DEVICE.
#include "../../include/rt_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "bvh_gpu.h"
#include "rt_device.h"
#include "rt_kernels.h"

using namespace rt;

static_assert(sizeof(rt_instance) == 64, "rt_instance must mirror VkAccelerationStructureInstanceKHR (64 B)");
static_assert(sizeof(rt_uniforms) == 104, "rt_uniforms must mirror UniformStructure (104 B)");
static_assert(sizeof(rt_uniforms) == sizeof(UniformsDev), "uniform mirrors out of sync");
static_assert(sizeof(rt_hit) == sizeof(HitRec), "hit mirrors out of sync");
static_assert(sizeof(rt_mesh_range) == 24, "rt_mesh_range layout");
static_assert(sizeof(rt_material) == sizeof(MaterialDev), "material mirrors out of sync");

namespace {

thread_local std::string g_create_error;

// Live contexts (frame slots) per device, over ALL scenes of the process: every one of them may have a k_tail launch in flight,
// and the workgroups of all of them must be co-resident for their grid barriers to complete (rt::tail_grid).
std::mutex g_live_mutex;
std::map<int, int> g_live_slots;
int live_slots_on(int device) { std::lock_guard<std::mutex> lk(g_live_mutex); auto it = g_live_slots.find(device); return it == g_live_slots.end() ? 0 : it->second; }
void live_slots_add(int device, int d) { std::lock_guard<std::mutex> lk(g_live_mutex); g_live_slots[device] += d; }

struct Mesh {
  rt_mesh_range range{};
  bool built = false;
  bool gpu_built = false;   // BLAS built on the device (bvh_gpu.hip): qnodes/tris hold its downloaded result, bvh/bvh4 stay empty
  Aabb bounds{};
  BuiltBvh bvh;
  Bvh4 bvh4;
  std::vector<BvhNodeQ> qnodes;   // quantized form of bvh.nodes
  float q_lo[3] = {0, 0, 0}, q_scale[3] = {1, 1, 1};
  std::vector<TriPacket> tris;
  int levels = 0;                 // interior levels of the BVH2 (bounds the traversal stack)
  int32_t node_base = 0, node_base4 = 0;   // position of this mesh's nodes / packets in the linked arrays
  int32_t root = 0;                        // index of the mesh's root node in the linked quantized array
  uint32_t tri_base = 0;
  uint32_t cover_first = 0, cover_count = 0;   // the mesh's frontier boxes in Scene::d_cover_boxes (primary-ray coverage mask)
};

struct TimedSpan { int cat; hipEvent_t a, b; };
enum { CAT_RAYGEN = 0, CAT_TRACE, CAT_SHADE, CAT_SHADOW, CAT_RESOLVE, CAT_FRAME, CAT_TAIL, CAT_N };

}  // namespace

// Everything a frame only READS, shared by the frame slots of one GPU: geometry (bindings 2/3), the linked BLAS nodes and
// triangle packets, the cube map (binding 5).  The reference shares exactly these across its swapchain images and duplicates
// only the per-image command buffer, fence and semaphores (src/main.cpp:2597, 2740-2749); a context created with
// rt_create_frame_slot is that per-image part.  Each slot owns a TLAS region behind the BLAS nodes (one base pointer for
// the kernels): slot k's TLAS nodes start at n_blas_nodes + k * tlas_cap.
constexpr int MAX_SLOTS = 16;          // = MAX_TAILS_IN_FLIGHT: frame slots per scene
// (ux, uy) of every sample of one frame size and shard layout (kernels.hip k_jitter_table): depends on neither camera nor scene, so
// it is computed once and shared by every frame slot that renders that size (cfg3: 66 MB instead of two binary64 sines per sample and frame)
struct JitterTable {
  int W = 0, H = 0, rows = 0, band_rows = 0, shard = 0, n_shards = 0; uint32_t spp = 0;
  float2* d = nullptr;
  hipEvent_t ready = nullptr;          // the fill kernel is done (frames on other streams wait for it once)
  uint64_t last_use = 0;
};
constexpr size_t MAX_JITTER_TABLES = 8;

struct Scene {
  int device = 0;
  std::vector<rt_ctx*> members;        // every context that renders from this scene
  uint32_t slot_mask = 0;              // TLAS regions in use
  std::vector<JitterTable> jitter_tables;
  uint64_t jitter_clock = 0;

  // geometry (bindings 2/3)
  std::vector<float> h_verts;
  std::vector<uint32_t> h_idx;
  float* d_verts = nullptr;
  uint32_t* d_idx = nullptr;
  std::vector<Mesh> meshes;
  bool blas_linked = false;
  // linked acceleration structures: BLAS part [0, n_blas*), then 2 * MAX_SLOTS TLAS regions (two per slot) of tlas_cap entries each
  std::vector<BvhNodeQ> h_blasq;
  BvhNodeQ* d_blas_nodes = nullptr;
  std::vector<Bvh4Node> h_blas4;
  Bvh4Node* d_nodes4 = nullptr;        // variant 1
  std::vector<WideNodeQ> h_wide;
  WideNodeQ* d_wide = nullptr;         // variant 2, same numbering as d_blas_nodes
  float4* d_tris = nullptr;
  size_t n_blas_nodes = 0, n_blas4 = 0, n_tris = 0;
  float* d_cover_boxes = nullptr;      // object-space frontier boxes of all meshes, 6 floats each (k_cover)
  uint32_t max_cover_count = 0;
  uint32_t tlas_cap = 1024;            // TLAS nodes per slot (grows when a sole owner needs more)
  bool arrays_ready = false;           // device node arrays allocated for (n_blas*, tlas_cap, variant)
  int variant = 0;                     // traversal variant the arrays were linked for

  // cube map (binding 5)
  uchar4* d_sky = nullptr;
  int sky_w = 0, sky_h = 0;

  // row n4: MTL materials (none: the reference's hard-coded constants)
  MaterialDev* d_materials = nullptr;
  uint32_t* d_prim_material = nullptr;
  int n_materials = 0;
  size_t n_prim_material = 0;
};

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  std::string info;
  int n_cu = 0;
  Scene* scene = nullptr;
  int slot = 0;                  // this context's TLAS region in the scene's node arrays

  // instances / TLAS (binding 0) of this slot
  std::vector<rt_instance> h_inst;
  std::vector<uint32_t> inst_types;   // rt_set_instance_types: per-instance object type (empty: src/shader.rgen:96's two-way switch)
  BuiltBvh tlas;
  Bvh4 tlas4;
  bool tlas_valid = false;
  // instance records and TLAS nodes are double-buffered (parity): the records of the next frame are written while the
  // frame enqueued before still reads its own, so a per-frame TLAS update never waits for the device
  InstanceDev* d_inst[2] = {nullptr, nullptr};
  size_t cap_inst[2] = {0, 0};
  int parity = 0;
  int tlas_node_count[2] = {0, 0};   // nodes of the TLAS uploaded for that parity
  // frame batch (rt_set_batch / rt_trace_shard_batch): batch_k frames go through one pass of the pipeline; h_inst then holds the
  // instances of all of them (frame k's at k * inst_per_frame), batch_uni their uniform blocks, and the slot's TLAS region their trees,
  // tlas_stride[parity] nodes apart, in one quantisation
  int batch_k = 1;
  uint32_t batch_out_stride = 0;   // pixels between the frames' shard images in the caller's buffer (0: compact), set per rt_trace_shard_batch call
  int inst_per_frame = 0;
  int tlas_stride[2] = {0, 0};
  std::vector<UniformsDev> batch_uni;
  hipEvent_t ev_frame[2] = {nullptr, nullptr};   // end of the last frame that read the buffers of that parity
  bool ev_frame_valid[2] = {false, false};
  float tlas_q_lo[3] = {0, 0, 0}, tlas_q_scale[3] = {1, 1, 1};
  uint32_t ovf_stride = STACK_OVF, ovf_alloc_stride = 0;
  int stack_need = 0;            // deepest traversal stack of this slot's trees (k_packet keeps its wave stack in two VGPRs: 128 entries)
  int ovf_alloc_blocks = 0;
  // per-frame TLAS update without a host stall: records are assembled in pinned memory and copied on the context's stream;
  // a frame on another stream waits for ev_upload on the device
  char* h_stage[2] = {nullptr, nullptr};
  size_t stage_bytes[2] = {0, 0};
  hipEvent_t ev_upload[2] = {nullptr, nullptr};   // the copies from the staging buffer of that parity are done
  bool upload_inflight[2] = {false, false};
  bool upload_pending = false;                    // the current parity's copies have not been waited for by a frame yet

  // uniforms (binding 1)
  UniformsDev uni{};
  bool have_uni = false;

  // frame state
  FrameDev frame{};
  size_t frame_capacity = 0;     // rays
  size_t out_capacity = 0;       // float4 pixels in lib-owned output
  float4* d_out_own = nullptr;
  float4* h_out_pinned = nullptr;   // rt_trace_async: pinned host copy of the frame
  size_t pinned_capacity = 0;      // float4 pixels
  bool out_rgba8 = false;          // "output_rgba8" / "output_bgra8": frames are stored as 8-bit (4 bytes per pixel) instead of RGBA32F
  bool out_bgra = false;           // ... in the byte order of B8G8R8A8 surfaces (the usual surfaceFormatList[0], src/main.cpp:1204, 1899)
  bool async_pending = false;
  int async_w = 0, async_h = 0;
  uint32_t* d_counters = nullptr;   // TWO counter blocks: frame k uses block k & 1 and its last kernel zeroes the other one for frame k + 1
  int cnt_parity = 0;
  int32_t* d_ovf = nullptr;
  LaunchCfg cfg{};
  bool grid_user_set = false;      // "trace_blocks_per_cu" was set explicitly: keep it
  // Per-launch caps of the persistent traversal grid (workgroups per CU).  -1 = automatic: with three or more frame slots on the GPU a launch
  // of at most GRID_SMALL_RAYS rays (the slot's previous frame tells) takes 2 per CU instead of 3 — closest hit always, shadow only when its
  // rays start from the kept records (a third fewer visits); measured on cfg3 (both meshes) -2.7 % / -1.6 %, cfg4's 6 M-ray launches and the
  // animated loop's shadow rays from the TLAS root want the 3 (profiles/r03_experiments.txt).  0 = no cap, 1..8 = that many.
  int closest_blocks_per_cu = -1, shadow_blocks_per_cu = -1;
  int tail_blocks = TAIL_BLOCKS;   // grid of k_tail, clamped so that the tails of every live slot on the device are always co-resident
  int tail_resident_per_cu = 0;
  int tail_mode = 1;             // 0: one launch per bounce and kernel; 1: k_tail when the last frame had few secondary rays; 2: always k_tail
  int jitter_table = 1;          // 1: k_raygen reads (ux, uy) from the scene's table of this frame size (bit-identical to evaluating the hash)
  const float2* jitter_waited = nullptr; hipStream_t jitter_waited_stream = nullptr;   // the table / stream this context last ordered itself behind
  int primary_cover = 1;         // 1: k_cover marks the screen tiles the meshes can project onto, k_raygen skips the others (result-identical)
  int tail_min_blocks = 1;       // smallest k_tail grid (experiments: RT_TAIL_MIN_BLOCKS; RT_TAIL_FULL_GRID=1 always launches tail_blocks)
  bool tail_full_grid = false;
  int entry_max_instances = 32;  // "entry_max_instances": scenes with more instances walk from the TLAS root (32 = what the kernels stage in LDS; cfg5's 17 instances: -1.3 %)
  int entry_points = 1;          // 1: k_entry gives every covered tile a list of deep subtrees and its primary rays start there (result-identical)
  int shadow_entry = 2;          // ... and every tile of a cube around the light one for the shadow rays (needs entry_points): 0 off; 1 rebuilt in every frame
                                 // (the six extra views cost every frame, and every 1/N shard of a frame, more k_cover / k_entry work than the shadow kernel
                                 // saves); 2 (default) kept while the light and the instances stand still (LightKey below): a static scene pays for them
                                 // once, a scene that moves every frame never
  int light_tiles = LIGHT_TILES_DEFAULT;   // tiles per side of a face of that cube
  EntryRec* d_entry = nullptr;   // one record per 8x8 tile of this slot's largest frame so far
  size_t entry_alloc_tiles = 0;
  // tile blobs (rt_device.h; kernels_tile.inc): 1 = k_blob writes, for every tile whose record names an instance, the nodes and triangle packets the tile's beam
  // can touch as one blob, and k_trace_tile walks the tile's primary rays through it in LDS (result-identical; needs entry_points)
  int camera_records = 1;   // entry records for the primary rays (k_entry's camera view); 0: their walks start at the TLAS root (the light-side records are not affected)
  int dead_shadow_rays = 1; // a shadow ray whose outcome cannot change its sample (diffuse and specular exactly 0) is settled in k_shade; 0: walked like the others
  int shadow_beams = 0; // ... and the shadow rays of the primary hits (k_beam_shadow): result-identical, a third of the node visits, and SLOWER (the rays of a pixel end
                        // at very different times — the first hit ends a ray — so most lanes of a wave wait: profiles/r04_experiments.txt); rt_set_param("shadow_beams", 1)
  bool sh_double = false;   // the shadow arrays of this context's frame have room for the shadow runs beside the compact queue
  int pixel_beams = 1;  // the primary rays of a pixel walked together (kernels_beam.inc): result-identical; rt_set_param("pixel_beams", 0) restores one walk per ray
  int tile_blobs = 0;   // OFF by default: result-identical and measured slower (profiles/r04_experiments.txt: a wave of k_tile spends two thirds of its life
                        // fetching its blob and storing its results, the walk itself is 2.6 x faster than the global one)
  uint32_t* d_tile_blob = nullptr;   // directory, one word per tile (allocated with d_entry)
  char* d_blob_arena = nullptr;
  uint32_t blob_slots = 0;
  uint4* d_blob_list = nullptr;
  size_t tile_dir_alloc = 0;
  EntryRec* d_light_entry = nullptr;   // 6 * light_tiles^2 records
  size_t light_alloc_tiles = 0;
  uint32_t* d_cover_mask = nullptr;   // TWO masks of cover_alloc_words: frame k uses one, its k_resolve clears the other
  uint32_t cover_alloc_words = 0;
  int cover_parity = 0;
  unsigned long long* h_stats = nullptr;   // pinned, device-visible StatSlot block written by k_resolve
  unsigned long long* d_stats = nullptr;
  bool last_empty = false;       // the last enqueued frame had no rows (nothing was launched)
  uint32_t* h_hint = nullptr;    // pinned, device-visible array [CNT_MAX_BOUNCES]: bounce-queue sizes of the most recent finished
  uint32_t* d_hint = nullptr;    // frame (written by k_resolve; a speed hint only, never affects results)
  int blas_builder = 1;          // 1: device builder (bvh_gpu.hip: binned SAH level by level; RT_GPU_BVH_ALGO 1 / 2 = LBVH / PLOC), default; 0: host binned-SAH (threaded)
  int timing = 0;                // 0 off, 1 HIP events around every kernel, 2 around the closest-hit traversal launches only
  bool counting = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<TimedSpan> spans;
  uint32_t timed_frames = 0;     // frames whose spans are waiting in `spans` (averaged by rt_get_stats)
  rt_stats last{};
  bool frame_pending = false;
  hipStream_t frame_stream = nullptr;
  // what the last enqueued frame was rendered from: a re-render after a k_tail fault must produce THAT frame, whatever
  // rt_set_uniforms / rt_set_instances did since (the instance records and TLAS nodes of `parity` are still the frame's own
  // as long as inst_gen[parity] has not moved: one later rt_set_instances writes the other parity)
  struct LastFrame { int W, H, band_rows, shard, n_shards; float4* d_out; bool counting; SceneDev sc; UniformsDev uni; int parity; uint64_t inst_gen; int batch_k; BatchTab bt; uint32_t out_frame_stride; } last_frame{};
  uint64_t inst_gen[2] = {0, 0};  // bumped whenever the records of that parity are rewritten
  // shadow_entry 2: the records of the cube around the light depend on the light, the instances and the trees only — like the TLAS they are
  // kept while those stand still: built in a context's first frame and in the second consecutive frame with a new key, used from then on, dropped when the key moves
  struct LightKey {
    float light[3] = {0, 0, 0}; uint64_t gen = 0; int parity = -1, n_inst = 0, tiles = 0;
    bool operator==(const LightKey& o) const { return light[0] == o.light[0] && light[1] == o.light[1] && light[2] == o.light[2] && gen == o.gen && parity == o.parity && n_inst == o.n_inst && tiles == o.tiles; }
  };
  LightKey light_key_seen, light_key_built;
  bool light_built = false, light_seen_valid = false;
  bool tail_disabled = false;    // a k_tail barrier gave up once: this context keeps to per-bounce launches from then on
  bool frame_rerendered = false; // collect_stats rendered the pending frame again (after a k_tail fault): copies of it are stale
  uint32_t tail_faults = 0;
  uint32_t* d_fault_total = nullptr;   // device word, never reset: frames of this context whose k_tail gave up (k_resolve adds to it and
  uint64_t faults_seen = 0;            // mirrors it into the statistics block, so a fault survives later frames' statistics)
  bool debug_force_tail_fault = false;   // rt_set_param "debug_force_tail_fault": treat the next k_tail frame as faulted (tests the fallback)
  uint32_t last_max_bounce = 0;
  uint64_t last_primary = 0;
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
  if (c) c->error = msg; else g_create_error = msg;
  return code;
}
#define HIP_TRY(c, expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(c, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE,            \
                  std::string("HIP runtime exception: return code ") + std::to_string((int)e_) +  \
                      " (" + hipGetErrorString(e_) + ") in " #expr);                              \
  } while (0)

// inverse of a row-major 3x4 affine map in binary64, rounded once (gl_WorldToObjectEXT)
void invert_affine(const float m[12], float out[12]) {
  double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  double tx = m[3], ty = m[7], tz = m[11];
  double c00 = e * i - f * h, c01 = c * h - b * i, c02 = b * f - c * e;
  double c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
  double c20 = d * h - e * g, c21 = b * g - a * h, c22 = a * e - b * d;
  double det = a * c00 + b * c10 + c * c20;
  double r = 1.0 / det;
  double n[9] = {c00 * r, c01 * r, c02 * r, c10 * r, c11 * r, c12 * r, c20 * r, c21 * r, c22 * r};
  out[0] = (float)n[0]; out[1] = (float)n[1]; out[2] = (float)n[2];
  out[4] = (float)n[3]; out[5] = (float)n[4]; out[6] = (float)n[5];
  out[8] = (float)n[6]; out[9] = (float)n[7]; out[10] = (float)n[8];
  out[3] = (float)(-(n[0] * tx + n[1] * ty + n[2] * tz));
  out[7] = (float)(-(n[3] * tx + n[4] * ty + n[5] * tz));
  out[11] = (float)(-(n[6] * tx + n[7] * ty + n[8] * tz));
}

// world-space box of an instance: 8 transformed corners of the BLAS bounds, padded so that the
// binary32 roundings of the transform can never make it non-conservative.
Aabb instance_world_box(const float o2w[12], const Aabb& b) {
  Aabb w; for (int k = 0; k < 3; k++) { w.lo[k] = 3.0e38f; w.hi[k] = -3.0e38f; }
  if (b.lo[0] > b.hi[0]) { for (int k = 0; k < 3; k++) w.lo[k] = w.hi[k] = 3.0e38f; return w; }  // empty mesh
  float mag = 0.f;
  for (int cx = 0; cx < 8; cx++) {
    double p[3] = {(cx & 1) ? b.hi[0] : b.lo[0], (cx & 2) ? b.hi[1] : b.lo[1], (cx & 4) ? b.hi[2] : b.lo[2]};
    for (int r = 0; r < 3; r++) {
      double v = o2w[4 * r] * p[0] + o2w[4 * r + 1] * p[1] + o2w[4 * r + 2] * p[2] + o2w[4 * r + 3];
      w.lo[r] = std::min(w.lo[r], (float)v); w.hi[r] = std::max(w.hi[r], (float)v);
      mag = std::max(mag, (float)std::fabs(v));
    }
  }
  float pad = 1e-5f * mag + 1e-30f;
  for (int k = 0; k < 3; k++) { w.lo[k] -= pad; w.hi[k] += pad; }
  return w;
}

}  // namespace

// Sizing rules with no device dependence (unit-tested on the CPU through rt_debug_sizing).
// Spill-stack elements: one area of `stride` entries per thread of the larger of the two persistent grids.
size_t rt::ovf_elems(int trace_blocks, int tail_blocks, uint32_t stride) {
  return (size_t)std::max(trace_blocks, tail_blocks) * 256u * (size_t)stride;
}
// Grid of k_tail.  Its workgroups spin in a grid barrier, so every k_tail that can be in flight at once (one per frame
// slot: MAX_TAILS_IN_FLIGHT per scene, and every scene of the process on that GPU counts — live_slots) must be co-resident:
// resident capacity / max(MAX_TAILS_IN_FLIGHT, live slots), a multiple of N_SHARDS, at most TAIL_BLOCKS.  0 = too many
// slots (or too small a device) for the tail kernel: per-bounce launches are used.
int rt::tail_grid(int n_cu, int resident_blocks_per_cu, int live_slots) {
  const long cap = (long)n_cu * std::max(0, resident_blocks_per_cu) / std::max(MAX_TAILS_IN_FLIGHT, live_slots);
  long g = std::min<long>(TAIL_BLOCKS, cap);
  g -= g % N_SHARDS;
  return g >= N_SHARDS ? (int)g : 0;
}

namespace {

// concatenate every built mesh into one node array / one packet array with global references (host side), then
// (re)allocate the device arrays: BLAS part + MAX_SLOTS TLAS regions, and upload the BLAS part
int alloc_scene_arrays(rt_ctx* c);

// Cache-line layout of a mesh's quantized nodes: four 32-byte nodes share a 128-byte line of the vector L1, and the step a walk takes
// most often is parent -> child.  Nodes are laid out in TREELETS of a node and its interior children (1-3 nodes, never split across a
// line; lines are filled greedily, treelets in depth-first order, a treelet's grandchildren are the roots of the next ones), so every
// second step of a descent stays inside the line the previous one fetched.  One-node treelets (a node over two leaves: half of all
// nodes, and always a line fetch of their own) fill the slots larger treelets leave empty, so the array grows by a fraction of a
// per cent only — the L2 of an XCD (4 MB) is the cache the deep levels live in.  A slot nothing fits repeats node 0 (never
// referenced, valid for walkers that scan the array).  The root stays node 0; any numbering gives the same hits.
static std::vector<BvhNodeQ> treelet_layout(const std::vector<BvhNodeQ>& in) {
  if (in.size() < 4) return in;
  std::vector<BvhNodeQ> out;
  out.reserve(in.size() + 8);
  std::vector<int32_t> new_of(in.size(), -1);
  std::vector<int32_t> todo;     // treelet roots, depth-first
  std::vector<int32_t> singles;  // treelets of ONE node (both children are leaves: half of all nodes) waiting for a slot a larger treelet left empty
  size_t singles_at = 0;
  auto place = [&](int32_t old) { new_of[old] = (int32_t)out.size(); out.push_back(in[old]); };
  todo.push_back(0);
  while (!todo.empty()) {
    const int32_t r = todo.back(); todo.pop_back();
    const BvhNodeQ& q = in[r];
    int32_t kids[2]; int nk = 0;
    if (q.child0 >= 0) kids[nk++] = q.child0;
    if (q.child1 >= 0 && q.child1 != q.child0) kids[nk++] = q.child1;
    const size_t size = 1 + (size_t)nk;
    if (size == 1 && r != 0) {
      if (out.size() & 3u) place(r); else singles.push_back(r);   // into the open line if there is one, else wait for a gap
      continue;
    }
    if ((out.size() & 3u) + size > 4u)
      while (out.size() & 3u) { if (singles_at < singles.size()) place(singles[singles_at++]); else out.push_back(in[0]); }
    place(r);
    for (int k = 0; k < nk; k++) place(kids[k]);
    for (int k = nk - 1; k >= 0; k--) {   // the grandchildren start treelets of their own (first child's subtree first)
      const BvhNodeQ& c = in[kids[k]];
      if (c.child1 >= 0 && c.child1 != c.child0) todo.push_back(c.child1);
      if (c.child0 >= 0) todo.push_back(c.child0);
    }
  }
  while (singles_at < singles.size()) place(singles[singles_at++]);
  for (BvhNodeQ& q : out) {
    if (q.child0 >= 0) q.child0 = new_of[q.child0];
    if (q.child1 >= 0) q.child1 = new_of[q.child1];
  }
  return out;
}

int link_blas(rt_ctx* c) {
  Scene* S = c->scene;
  size_t nn = 0, nt = 0, nn4 = 0;
  static const bool line_layout = [] { const char* e = getenv("RT_NODE_LAYOUT"); return e ? atoi(e) != 0 : true; }();
  std::vector<std::vector<BvhNodeQ>> laid(S->meshes.size());   // what is linked: the mesh's nodes in their final order
  std::vector<size_t> gap_from(S->meshes.size(), 0);
  for (size_t mi = 0; mi < S->meshes.size(); mi++) {
    Mesh& m = S->meshes[mi];
    if (!m.built) continue;
    if (!m.gpu_built) quantize_bvh2(m.bvh, m.qnodes, m.q_lo, m.q_scale);
    laid[mi] = line_layout ? treelet_layout(m.qnodes) : m.qnodes;
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "[link_blas] mesh %zu: %zu nodes, %zu slots in the cache-line layout\n", mi, m.qnodes.size(), laid[mi].size());
    gap_from[mi] = nn;
    nn = (nn + 3u) & ~(size_t)3u;   // every mesh starts on a line
    m.node_base = (int32_t)nn; m.tri_base = (uint32_t)nt; m.node_base4 = (int32_t)nn4;
    nn += laid[mi].size(); nt += m.tris.size(); nn4 += m.bvh4.nodes.size();
  }
  std::vector<BvhNodeQ> nodes(nn);
  std::vector<Bvh4Node> nodes4(nn4);
  std::vector<TriPacket> tris(nt);
  for (size_t mi = 0; mi < S->meshes.size(); mi++) {
    Mesh& m = S->meshes[mi];
    if (!m.built) continue;
    const std::vector<BvhNodeQ>& mq = laid[mi];
    for (size_t i = gap_from[mi]; i < (size_t)m.node_base; i++) nodes[i] = nodes[0];   // (alignment gap in front of this mesh: copies of a valid node)
    for (size_t i = 0; i < mq.size(); i++) {
      BvhNodeQ n = mq[i];
      auto fix = [&](int32_t ch) -> int32_t {
        if (ch >= 0) return ch + m.node_base;
        uint32_t ref = (uint32_t)(~ch);
        uint32_t first = (ref >> 3) + m.tri_base, cnt = ref & 7u;
        return ~(int32_t)((first << 3) | cnt);
      };
      n.child0 = fix(n.child0); n.child1 = fix(n.child1);
      nodes[m.node_base + i] = n;
    }
    for (size_t i = 0; i < m.bvh4.nodes.size(); i++) {
      Bvh4Node n = m.bvh4.nodes[i];
      for (int k = 0; k < 4; k++) {
        int32_t ch = n.c[k].ref;
        if (ch == 0x7FFFFFFF) continue;
        if (ch >= 0) n.c[k].ref = ch + m.node_base4;
        else { uint32_t ref = (uint32_t)(~ch); n.c[k].ref = ~(int32_t)((((ref >> 3) + m.tri_base) << 3) | (ref & 7u)); }
      }
      nodes4[m.node_base4 + i] = n;
    }
    if (!m.tris.empty()) memcpy(&tris[m.tri_base], m.tris.data(), m.tris.size() * sizeof(TriPacket));
    m.levels = bvh2_levels(mq.data(), mq.size(), 0);
    if (m.levels < 0) return fail(c, RT_ERR_DEVICE, "BLAS builder produced a node graph that is not a tree");
  }
  for (auto& m : S->meshes) m.root = m.node_base;
  // Frontier boxes for the primary-ray coverage mask (k_cover): walk every mesh breadth-first from its root until about
  // COVER_TARGET_BOXES child boxes are open, and keep those boxes (dequantized: the stored planes already contain the float
  // boxes with two quanta of margin) in object space.  A leaf met on the way contributes its box as it is.
  {
    std::vector<float> cover;
    S->max_cover_count = 0;
    size_t target = COVER_TARGET_BOXES;
    if (const char* e = getenv("RT_COVER_BOXES")) { const long v = atol(e); if (v >= 2 && v <= (1 << 20)) target = (size_t)v; }   // experiments
    for (auto& m : S->meshes) {
      m.cover_first = (uint32_t)(cover.size() / 6); m.cover_count = 0;
      if (!m.built || m.qnodes.empty()) continue;
      auto emit = [&](const BvhNodeQ& q, int k) {
        float lo[3], hi[3];
        for (int ax = 0; ax < 3; ax++) {
          const uint32_t w = q.w[3 * k + ax];
          const uint32_t pl = w & 0xFFFFu, ph = w >> 16;
          if (pl > ph) return;   // the absent child of a synthetic single-child root
          lo[ax] = m.q_lo[ax] + (float)pl * m.q_scale[ax]; hi[ax] = m.q_lo[ax] + (float)ph * m.q_scale[ax];
          const float pad = 1e-6f * (std::fabs(lo[ax]) + std::fabs(hi[ax]));
          lo[ax] -= pad; hi[ax] += pad;
        }
        cover.insert(cover.end(), lo, lo + 3); cover.insert(cover.end(), hi, hi + 3);
      };
      // open the LARGEST box first (object-space surface area) until `target` boxes are open: boxes of even size hug the
      // silhouette better than the boxes of one tree level
      struct Open { float area; int32_t parent; int k; int32_t ref; };
      auto area_of = [&](const BvhNodeQ& q, int k) -> float {
        float e[3];
        for (int ax = 0; ax < 3; ax++) {
          const uint32_t w = q.w[3 * k + ax];
          const uint32_t pl = w & 0xFFFFu, ph = w >> 16;
          if (pl > ph) return -1.0f;
          e[ax] = (float)(ph - pl) * m.q_scale[ax];
        }
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
      };
      auto cmp = [](const Open& x, const Open& y) { return x.area < y.area; };
      std::priority_queue<Open, std::vector<Open>, decltype(cmp)> open(cmp);
      std::vector<Open> closed;   // leaves: final as they are
      auto push_children = [&](int32_t n) {
        const BvhNodeQ& q = nodes[n];
        const int32_t ch[2] = {q.child0, q.child1};
        for (int k = 0; k < 2; k++) {
          const float ar = area_of(q, k);
          if (ar < 0.0f) continue;   // the absent child of a synthetic single-child root
          const Open o{ar, n, k, ch[k]};
          if (ch[k] >= 0) open.push(o); else closed.push_back(o);
        }
      };
      push_children(m.root);
      while (!open.empty() && open.size() + closed.size() + 1 <= target) {
        const Open o = open.top(); open.pop();
        push_children(o.ref);
      }
      for (const Open& o : closed) emit(nodes[o.parent], o.k);
      while (!open.empty()) { emit(nodes[open.top().parent], open.top().k); open.pop(); }
      m.cover_count = (uint32_t)(cover.size() / 6) - m.cover_first;
      S->max_cover_count = std::max(S->max_cover_count, m.cover_count);
    }
    if (S->d_cover_boxes) { HIP_TRY(c, hipFree(S->d_cover_boxes)); S->d_cover_boxes = nullptr; }
    HIP_TRY(c, hipMalloc((void**)&S->d_cover_boxes, std::max<size_t>(6, cover.size()) * sizeof(float)));
    if (!cover.empty()) HIP_TRY(c, hipMemcpy(S->d_cover_boxes, cover.data(), cover.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  S->variant = c->cfg.variant;
  std::vector<WideNodeQ> wide(S->variant == 2 ? nn : 0);   // only the variant that walks them pays for them
  if (S->variant == 2 && nn) widen_bvh2(nodes.data(), nn, 0, wide.data());
  S->h_wide.swap(wide);
  S->h_blas4.swap(nodes4); S->n_blas4 = nn4;
  S->h_blasq.swap(nodes);
  S->n_blas_nodes = nn; S->n_tris = nt;
  if (S->d_tris) { HIP_TRY(c, hipFree(S->d_tris)); S->d_tris = nullptr; }
  HIP_TRY(c, hipMalloc((void**)&S->d_tris, std::max<size_t>(1, nt) * sizeof(TriPacket)));
  if (nt) HIP_TRY(c, hipMemcpy(S->d_tris, tris.data(), nt * sizeof(TriPacket), hipMemcpyHostToDevice));
  S->arrays_ready = false;
  int r = alloc_scene_arrays(c); if (r) return r;
  S->blas_linked = true;
  return RT_OK;
}

int alloc_scene_arrays(rt_ctx* c) {
  Scene* S = c->scene;
  const size_t regions = (size_t)2 * MAX_SLOTS * S->tlas_cap;
  if (S->d_blas_nodes) { HIP_TRY(c, hipFree(S->d_blas_nodes)); S->d_blas_nodes = nullptr; }
  if (S->d_nodes4) { HIP_TRY(c, hipFree(S->d_nodes4)); S->d_nodes4 = nullptr; }
  if (S->d_wide) { HIP_TRY(c, hipFree(S->d_wide)); S->d_wide = nullptr; }
  HIP_TRY(c, hipMalloc((void**)&S->d_blas_nodes, (S->n_blas_nodes + regions) * sizeof(BvhNodeQ)));
  if (S->n_blas_nodes) HIP_TRY(c, hipMemcpy(S->d_blas_nodes, S->h_blasq.data(), S->n_blas_nodes * sizeof(BvhNodeQ), hipMemcpyHostToDevice));
  if (S->n_blas4) {   // quad traversal (host-built meshes only)
    HIP_TRY(c, hipMalloc((void**)&S->d_nodes4, (S->n_blas4 + regions) * sizeof(Bvh4Node)));
    HIP_TRY(c, hipMemcpy(S->d_nodes4, S->h_blas4.data(), S->n_blas4 * sizeof(Bvh4Node), hipMemcpyHostToDevice));
  }
  if (S->variant == 2) {
    HIP_TRY(c, hipMalloc((void**)&S->d_wide, (S->n_blas_nodes + regions) * sizeof(WideNodeQ)));
    if (S->n_blas_nodes) HIP_TRY(c, hipMemcpy(S->d_wide, S->h_wide.data(), S->n_blas_nodes * sizeof(WideNodeQ), hipMemcpyHostToDevice));
  }
  S->arrays_ready = true;
  return RT_OK;
}

// first node of this slot's TLAS region in the quantized / wide arrays and in the BVH4 array
inline size_t tlas_base(const rt_ctx* c) { return c->scene->n_blas_nodes + (size_t)(2 * c->slot + c->parity) * c->scene->tlas_cap; }
inline size_t tlas_base4(const rt_ctx* c) { return c->scene->n_blas4 + (size_t)(2 * c->slot + c->parity) * c->scene->tlas_cap; }

// Instance records and this slot's TLAS nodes go to the device WITHOUT stalling the host: they are assembled in pinned
// memory and copied on the context's stream; ev_upload orders frames on other streams behind the copies.  (The reference
// allocates two buffers, submits and blocks on vkWaitForFences every frame, src/main.cpp:672-696, 752-778.)
// frame_trees: the TLAS of every frame of a batch (same topology, refitted to that frame's boxes); NULL = the one tree c->tlas
int upload_instances(rt_ctx* c, const std::vector<InstanceDev>& inst_dev, const std::vector<BuiltBvh>* frame_trees = nullptr) {
  Scene* S = c->scene;
  const size_t n = inst_dev.size();
  std::vector<BvhNodeQ> tq;
  size_t stride = 0;
  if (frame_trees == nullptr) quantize_bvh2(c->tlas, tq, c->tlas_q_lo, c->tlas_q_scale);
  else {
    // one quantisation for all frames (the kernels carry ONE world-space dequantisation), the trees one after another
    double lo[3] = {3e38, 3e38, 3e38}, hi[3] = {-3e38, -3e38, -3e38};
    for (const BuiltBvh& t : *frame_trees) bvh2_bounds(t, lo, hi);
    quant_params(lo, hi, c->tlas_q_lo, c->tlas_q_scale);
    const size_t per_frame = inst_dev.size() / frame_trees->size();
    for (size_t k = 0; k < frame_trees->size(); k++) {
      std::vector<BvhNodeQ> one;
      quantize_bvh2_in((*frame_trees)[k], one, c->tlas_q_lo, c->tlas_q_scale);
      if (k == 0) stride = one.size();
      for (BvhNodeQ& nd : one) {   // links local to the frame's tree -> local to the slot's region; leaves name the frame's own instance records
        if (nd.child0 >= 0) nd.child0 += (int32_t)(k * stride); else nd.child0 = ~(int32_t)((uint32_t)(~nd.child0) + (uint32_t)(k * per_frame));
        if (nd.child1 >= 0) nd.child1 += (int32_t)(k * stride); else nd.child1 = ~(int32_t)((uint32_t)(~nd.child1) + (uint32_t)(k * per_frame));
      }
      tq.insert(tq.end(), one.begin(), one.end());
    }
  }
  c->tlas_node_count[c->parity] = (int)(frame_trees ? stride : tq.size());
  c->tlas_stride[c->parity] = frame_trees ? (int)stride : 0;
  const size_t need = std::max(tq.size(), c->tlas4.nodes.size());
  if (need > S->tlas_cap) {
    // a sole owner may grow the regions (nothing else reads the arrays); with several slots the arrays cannot move
    if (S->members.size() > 1)
      return fail(c, RT_ERR_INVALID_ARGUMENT, "TLAS of " + std::to_string(need) + " nodes exceeds the " + std::to_string(S->tlas_cap) +
                                                 " a frame slot of a shared scene can hold: set the instances before creating frame slots");
    S->tlas_cap = (uint32_t)((need + 1023) & ~(size_t)1023);
    int r = alloc_scene_arrays(c); if (r) return r;
  }
  const int par = c->parity;
  if (n > c->cap_inst[par]) {
    if (c->d_inst[par]) HIP_TRY(c, hipFree(c->d_inst[par]));
    c->d_inst[par] = nullptr; c->cap_inst[par] = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_inst[par], n * sizeof(InstanceDev)));
    c->cap_inst[par] = n;
  }
  const size_t base = tlas_base(c), base4 = tlas_base4(c);
  for (auto& nd : tq) {   // interior links are relative to the TLAS: rebase them to this slot's region
    if (nd.child0 >= 0) nd.child0 += (int32_t)base;
    if (nd.child1 >= 0) nd.child1 += (int32_t)base;
  }
  std::vector<Bvh4Node> t4;
  if (S->d_nodes4) {
    t4 = c->tlas4.nodes;
    for (auto& nd : t4)
      for (int k = 0; k < 4; k++)
        if (nd.c[k].ref >= 0 && nd.c[k].ref != 0x7FFFFFFF) nd.c[k].ref += (int32_t)base4;
  }
  std::vector<WideNodeQ> tw(S->variant == 2 ? tq.size() : 0);
  if (S->variant == 2) widen_bvh2(tq.data(), tq.size(), (int32_t)base, tw.data());
  // pinned staging: [instances][quantized TLAS][BVH4 TLAS][wide TLAS]
  const size_t b_inst = n * sizeof(InstanceDev), b_q = tq.size() * sizeof(BvhNodeQ), b_4 = t4.size() * sizeof(Bvh4Node), b_w = tw.size() * sizeof(WideNodeQ);
  const size_t total = b_inst + b_q + b_4 + b_w;
  if (total > c->stage_bytes[par]) {
    if (c->h_stage[par]) HIP_TRY(c, hipHostFree(c->h_stage[par]));
    c->h_stage[par] = nullptr; c->stage_bytes[par] = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_stage[par], total + 4096, hipHostMallocDefault));
    c->stage_bytes[par] = total + 4096;
  }
  if (!c->ev_upload[par]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload[par], hipEventDisableTiming));
  char* st = c->h_stage[par];
  memcpy(st, inst_dev.data(), b_inst);
  memcpy(st + b_inst, tq.data(), b_q);
  if (b_4) memcpy(st + b_inst + b_q, t4.data(), b_4);
  if (b_w) memcpy(st + b_inst + b_q + b_4, tw.data(), b_w);
  HIP_TRY(c, hipMemcpyAsync(c->d_inst[par], st, b_inst, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(S->d_blas_nodes + base, st + b_inst, b_q, hipMemcpyHostToDevice, c->stream));
  if (b_4) HIP_TRY(c, hipMemcpyAsync(S->d_nodes4 + base4, st + b_inst + b_q, b_4, hipMemcpyHostToDevice, c->stream));
  if (b_w) HIP_TRY(c, hipMemcpyAsync(S->d_wide + base, st + b_inst + b_q + b_4, b_w, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_upload[par], c->stream));
  c->upload_pending = true; c->upload_inflight[par] = true;
  return RT_OK;
}

SceneDev scene_dev(const rt_ctx* c) {
  const Scene* S = c->scene;
  SceneDev s{};
  s.nodes4 = S->d_nodes4; s.tlas_root4 = (int)tlas_base4(c);
  s.wide_nodes = S->d_wide; s.ovf_stride = c->ovf_stride;
  s.tlas_nodes = c->tlas_node_count[c->parity];
  s.tlas_stride = c->batch_k > 1 ? c->tlas_stride[c->parity] : 0;
  s.blas_nodes = S->d_blas_nodes; s.tlas_root = (int)tlas_base(c); s.tris = S->d_tris; s.inst = c->d_inst[c->parity];
  s.verts = S->d_verts; s.idx = S->d_idx; s.sky = S->d_sky; s.n_inst = (int)c->h_inst.size();
  s.sky_w = S->sky_w; s.sky_h = S->sky_h;
  s.materials = S->d_materials; s.prim_material = S->d_prim_material; s.n_materials = S->n_materials;
  s.cover_boxes = S->d_cover_boxes;
  for (int k = 0; k < 3; k++) { s.tlas_q_lo[k] = c->tlas_q_lo[k]; s.tlas_q_scale[k] = c->tlas_q_scale[k]; }
  return s;
}

int collect_stats(rt_ctx* c);

// Can a ray of a frame be FAR (kernels.hip quant_far: |q_lo - origin| / q_scale beyond ~2e6 quanta on some axis) from a tree it
// walks?  Origins are the camera (primary rays) and points of the scene itself (bounce and shadow rays start on surfaces: inside
// the TLAS bounds).  World space: against the TLAS quantisation; object space of every instance: the same points through w2o
// against the mesh's quantisation — the device's own test (quant_far_o) evaluated at the corners of the box that holds every
// possible origin.  False for every BASELINE workload, so their frames run the kernels without the far-ray logic.
bool far_possible(const rt_ctx* c, const UniformsDev& u) {
  const Scene* S = c->scene;
  const double K = 0.99 * 2097152.0;
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    // the TLAS bounds as its quantisation spans them (current after every build and refit), with the camera
    const double tlo = c->tlas_q_lo[k], thi = (double)c->tlas_q_lo[k] + 65535.0 * (double)c->tlas_q_scale[k];
    lo[k] = std::min(tlo, (double)u.position[k]); hi[k] = std::max(thi, (double)u.position[k]);
    if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]) || lo[k] > hi[k]) return true;
  }
  double smin = std::min({(double)c->tlas_q_scale[0], (double)c->tlas_q_scale[1], (double)c->tlas_q_scale[2]});
  for (int k = 0; k < 3; k++)
    if (std::max(std::fabs(c->tlas_q_lo[k] - lo[k]), std::fabs(c->tlas_q_lo[k] - hi[k])) > K * smin) return true;
  for (const rt_instance& in : c->h_inst) {
    const Mesh& m = S->meshes[in.mesh];
    if (!m.range.prim_count) continue;
    float w2o[12];
    invert_affine(in.transform, w2o);
    smin = std::min({(double)m.q_scale[0], (double)m.q_scale[1], (double)m.q_scale[2]});
    for (int cx = 0; cx < 8; cx++) {
      const double p[3] = {(cx & 1) ? hi[0] : lo[0], (cx & 2) ? hi[1] : lo[1], (cx & 4) ? hi[2] : lo[2]};
      for (int r = 0; r < 3; r++) {
        const double v = w2o[4 * r] * p[0] + w2o[4 * r + 1] * p[1] + w2o[4 * r + 2] * p[2] + w2o[4 * r + 3];
        if (!(std::fabs(m.q_lo[r] - v) <= K * smin)) return true;
      }
    }
  }
  return false;
}

// Calls that rewrite what a pending frame reads wait for that frame first (the reference waits on the frame's fence
// before it touches the TLAS or the uniform buffer again, src/main.cpp:772-778).
int quiesce(rt_ctx* c) {
  if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "a frame submitted with rt_trace_async is pending: call rt_trace_wait first");
  if (c->frame_pending) return collect_stats(c);
  return RT_OK;
}
// Calls that rewrite the SHARED scene wait for the frames of every slot that renders from it.
int quiesce_scene(rt_ctx* c) {
  for (rt_ctx* m : c->scene->members) {
    if (m->async_pending) return fail(c, RT_ERR_NOT_READY, "a frame slot of this scene has a frame pending (rt_trace_async): collect it with rt_trace_wait first");
    if (m->frame_pending) { int r = collect_stats(m); if (r) { if (m != c) c->error = m->error; return r; } }
    for (int k = 0; k < 2; k++)
      if (m->upload_inflight[k]) { HIP_TRY(c, hipEventSynchronize(m->ev_upload[k])); m->upload_inflight[k] = false; }
    m->upload_pending = false;
  }
  return RT_OK;
}
// the linked arrays changed: every slot has to set its instances again
void invalidate_tlas(Scene* S) { for (rt_ctx* m : S->members) m->tlas_valid = false; }

int ready_to_trace(rt_ctx* c, bool batch = false) {
  if (!c->scene->d_verts) return fail(c, RT_ERR_NOT_READY, "rt_upload_geometry has not been called");
  if (!c->tlas_valid) return fail(c, RT_ERR_NOT_READY, "rt_set_instances has not been called");
  if (!batch && c->batch_k != 1) return fail(c, RT_ERR_NOT_READY, "the context holds a frame batch (rt_set_batch): render it with rt_trace_shard_batch, or call rt_set_instances for a single frame");
  return RT_OK;
}

int ensure_common(rt_ctx* c) {
  if (!c->d_counters) {
    HIP_TRY(c, hipMalloc((void**)&c->d_counters, 2 * CNT_WORDS * sizeof(uint32_t)));
    // zeroed BEFORE any frame can be enqueued on whatever stream: a plain hipMemset of device memory may still be running
    // when it returns, and the frame's stream is not ordered behind the null stream
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 2 * CNT_WORDS * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->cnt_parity = 0;
  }
  if (!c->d_fault_total) {
    HIP_TRY(c, hipMalloc((void**)&c->d_fault_total, sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->d_fault_total, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->faults_seen = 0;
  }
  if (!c->h_hint) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_hint, CNT_MAX_BOUNCES * sizeof(uint32_t), hipHostMallocMapped));
    for (int b = 0; b < CNT_MAX_BOUNCES; b++) c->h_hint[b] = 0xFFFFFFFFu;   // unknown: the first frame takes one launch per bounce
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_hint, c->h_hint, 0));
  }
  if (!c->h_stats) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocMapped));
    memset(c->h_stats, 0, STAT_WORDS * sizeof(unsigned long long));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_stats, c->h_stats, 0));
  }
  if (c->d_ovf && (c->ovf_alloc_stride < c->ovf_stride || c->ovf_alloc_blocks < std::max(c->cfg.trace_blocks, c->tail_blocks))) { HIP_TRY(c, hipFree(c->d_ovf)); c->d_ovf = nullptr; }
  if (!c->d_ovf) {
    // one spill area per persistent thread of the LARGER grid: k_trace runs cfg.trace_blocks workgroups, k_tail tail_blocks
    HIP_TRY(c, hipMalloc((void**)&c->d_ovf, ovf_elems(c->cfg.trace_blocks, c->tail_blocks, c->ovf_stride) * sizeof(int32_t)));
    c->ovf_alloc_stride = c->ovf_stride; c->ovf_alloc_blocks = std::max(c->cfg.trace_blocks, c->tail_blocks);
  }
  return RT_OK;
}

int ensure_frame(rt_ctx* c, size_t capacity) {
  int r = ensure_common(c); if (r) return r;
  const bool want_double = c->shadow_beams != 0;
  if (capacity <= c->frame_capacity && (!want_double || c->sh_double)) return RT_OK;
  capacity = std::max(capacity, c->frame_capacity);
  FrameDev& f = c->frame;
  void** ptrs[] = {(void**)&f.ray_o[0], (void**)&f.ray_o[1], (void**)&f.ray_d[0], (void**)&f.ray_d[1], (void**)&f.hit_a, (void**)&f.sample_color};
  // (the shadow arrays hold two regions: the shadow runs of bounce 0 in their primary rays' slots — kernels_beam.inc — and the compact
  // queue of the later bounces above them)
  void** sh_ptrs[] = {(void**)&f.sh_o, (void**)&f.sh_d, (void**)&f.sh_c};
  for (void** p : ptrs) { if (*p) HIP_TRY(c, hipFree(*p)); *p = nullptr; }
  for (void** p : sh_ptrs) { if (*p) HIP_TRY(c, hipFree(*p)); *p = nullptr; }
  if (f.hit_inst) { HIP_TRY(c, hipFree(f.hit_inst)); f.hit_inst = nullptr; }
  if (f.sh_e) { HIP_TRY(c, hipFree(f.sh_e)); f.sh_e = nullptr; }
  c->frame_capacity = 0;
  for (void** p : ptrs) HIP_TRY(c, hipMalloc(p, capacity * sizeof(float4)));
  const size_t sh_cap = (want_double ? 2 : 1) * capacity;
  for (void** p : sh_ptrs) HIP_TRY(c, hipMalloc(p, sh_cap * sizeof(float4)));
  HIP_TRY(c, hipMalloc((void**)&f.hit_inst, capacity * sizeof(int32_t)));
  HIP_TRY(c, hipMalloc((void**)&f.sh_e, sh_cap * sizeof(uint32_t)));
  c->sh_double = want_double;
  c->frame_capacity = capacity;
  return RT_OK;
}

hipEvent_t take_event(rt_ctx* c) {
  if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->ev_pool.push_back(e); }
  return c->ev_pool[c->ev_used++];
}

// the scene's (ux, uy) table for this frame's size, sample count and shard layout; built on first use on stream s
int jitter_table_for(rt_ctx* c, const FrameDev& f, uint32_t spp, hipStream_t s, const float2** out) {
  Scene* S = c->scene;
  *out = nullptr;
  JitterTable* t = nullptr;
  for (JitterTable& k : S->jitter_tables)
    if (k.W == f.width && k.H == f.height && k.rows == f.rows && k.spp == spp && k.band_rows == f.band_rows && k.shard == f.shard && k.n_shards == f.n_shards) { t = &k; break; }
  if (!t) {
    if (S->jitter_tables.size() >= MAX_JITTER_TABLES) {
      // drop the table used longest ago; frames of any slot may still read it, so the device drains first (rare: a scene that
      // keeps changing its frame size)
      HIP_TRY(c, hipDeviceSynchronize());
      size_t lru = 0;
      for (size_t k = 1; k < S->jitter_tables.size(); k++) if (S->jitter_tables[k].last_use < S->jitter_tables[lru].last_use) lru = k;
      hipFree(S->jitter_tables[lru].d); hipEventDestroy(S->jitter_tables[lru].ready);
      S->jitter_tables.erase(S->jitter_tables.begin() + (long)lru);
      for (rt_ctx* m : S->members) m->jitter_waited = nullptr;
    }
    JitterTable n;
    n.W = f.width; n.H = f.height; n.rows = f.rows; n.spp = spp; n.band_rows = f.band_rows; n.shard = f.shard; n.n_shards = f.n_shards;
    HIP_TRY(c, hipMalloc((void**)&n.d, jitter_table_elems(f.width, f.rows, spp) * sizeof(float2)));
    if (hipEventCreateWithFlags(&n.ready, hipEventDisableTiming) != hipSuccess) { hipFree(n.d); return fail(c, RT_ERR_DEVICE, "hipEventCreate failed"); }
    launch_jitter_table(f, spp, n.d, s);
    HIP_TRY(c, hipEventRecord(n.ready, s));
    S->jitter_tables.push_back(n);
    t = &S->jitter_tables.back();
    c->jitter_waited = t->d; c->jitter_waited_stream = s;   // (stream order)
  }
  if (c->jitter_waited != t->d || c->jitter_waited_stream != s) {
    HIP_TRY(c, hipStreamWaitEvent(s, t->ready, 0));
    c->jitter_waited = t->d; c->jitter_waited_stream = s;
  }
  t->last_use = ++S->jitter_clock;
  *out = t->d;
  return RT_OK;
}

struct Span {
  rt_ctx* c; int cat; hipStream_t s; hipEvent_t a = nullptr;
  // An event record between two kernels costs ~10 us of idle GPU (measured: kernels of a frame run back to back without them),
  // so the light mode brackets only the dominant kernel.
  bool on() const { return c->timing == 1 || (c->timing == 2 && cat == CAT_TRACE); }
  Span(rt_ctx* c_, int cat_, hipStream_t s_) : c(c_), cat(cat_), s(s_) {
    if (on()) { a = take_event(c); hipEventRecord(a, s); }
  }
  ~Span() {
    if (on()) { hipEvent_t b = take_event(c); hipEventRecord(b, s); c->spans.push_back({cat, a, b}); }
  }
};

// `again`: render THIS frame once more (after a k_tail fault) from the state it was submitted with, without k_tail
int enqueue_frame(rt_ctx* c, int W, int H, int band_rows, int shard, int n_shards, float4* d_out, hipStream_t s, const rt_ctx::LastFrame* again = nullptr) {
  const int rows = rt_shard_rows(H, band_rows, shard, n_shards);   // of ONE frame's shard
  const bool no_tail = again != nullptr;
  // frame batch: K frames (their instance records and TLAS trees are in place: rt_set_batch) go through this one pass
  const int K = again ? again->batch_k : c->batch_k;
  SceneDev sc = again ? again->sc : scene_dev(c);
  const UniformsDev u = again ? again->uni : c->uni;
  BatchTab bt{};
  if (again) bt = again->bt;
  else if (K > 1)
    for (int k = 0; k < K; k++) {
      const UniformsDev& uk = c->batch_uni[k];
      for (int j = 0; j < 4; j++) { bt.position[k][j] = uk.position[j]; bt.right[k][j] = uk.right[j]; bt.up[k][j] = uk.up[j]; bt.forward[k][j] = uk.forward[j]; }
      for (int j = 0; j < 3; j++) bt.light[k][j] = uk.light_position[j];
    }
  if (K > 1) sc.batch_samples = (uint32_t)((size_t)u.samples_per_pixel * (size_t)rows * (size_t)W);
  const int n_inst1 = K > 1 ? sc.n_inst / K : sc.n_inst;   // instances of one frame
  const int frame_parity = again ? again->parity : c->parity;
  if (!again) { c->last_frame = {W, H, band_rows, shard, n_shards, d_out, c->counting, sc, u, c->parity, c->inst_gen[c->parity], K, bt, c->batch_out_stride ? c->batch_out_stride : (uint32_t)((size_t)rows * W)}; c->frame_rerendered = false; }
  if (u.samples_per_pixel == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "samplesPerPixel must be >= 1");
  if (u.max_bounce_count + 2 > (uint32_t)CNT_MAX_BOUNCES) return fail(c, RT_ERR_INVALID_ARGUMENT, "maxBounceCount too large (max 69)");
  const size_t tiles = (size_t)((W + 7) / 8) * (size_t)((rows + 7) / 8) * (size_t)K;
  const size_t samples = tiles * u.samples_per_pixel * 64;   // k_raygen threads
  if (samples >= 0xF0000000ull) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame too large for 32-bit sample ids");
  if (rows > 8 * 65535 || u.samples_per_pixel > 4u * 65535u) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame too large for the raygen grid");
  // k_raygen block b appends to shard b % 8, so a shard never receives more than this many rays; paths
  // stay in their shard, so the bound holds for every later queue as well
  if ((size_t)rows * K > 8u * 65535u) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame batch too tall for the raygen grid");
  const size_t raygen_blocks = raygen_block_count(W, rows, u.samples_per_pixel, K);
  // (with tile blobs a shard also holds the tile region — a fixed 256 slots per blob and group of four samples, whether a ray fills
  // them or not — while the rays k_tile hands on, at most as many again, go to queue 0 of the same shard; and k_tile sends ALL sample
  // groups of a tile to shard tile % 8 where k_raygen spreads them: twice the room covers both)
  const size_t shard_cap = std::max<size_t>(256, ((raygen_blocks + N_SHARDS - 1) / N_SHARDS) * 256) * (c->tile_blobs ? 2 : 1);
  const size_t capacity = shard_cap * N_SHARDS;
  int r = ensure_frame(c, capacity); if (r) return r;
  FrameDev f = c->frame;
  // the counters of this frame were zeroed by the previous frame's k_resolve (or at allocation); this frame's k_resolve zeroes the other block
  f.counters = c->d_counters + (size_t)c->cnt_parity * CNT_WORDS; f.counters_next = c->d_counters + (size_t)(c->cnt_parity ^ 1) * CNT_WORDS;
  f.ovf_stack = c->d_ovf; f.out = d_out; f.out_rgba8 = c->out_rgba8 ? (c->out_bgra ? 2 : 1) : 0; f.hint = c->d_hint; f.stats_out = c->d_stats; f.fault_total = c->d_fault_total;
  f.shard_cap = (uint32_t)shard_cap; f.width = W; f.height = H; f.rows = rows;
  f.band_rows = band_rows; f.shard = shard; f.n_shards = n_shards;
  f.batch_k = K;
  f.out_frame_stride = again ? again->out_frame_stride : (c->batch_out_stride ? c->batch_out_stride : (uint32_t)((size_t)rows * W));
  // Primary-ray coverage mask: on when the bands are whole 8x8 tiles, the camera basis is invertible and the scene has boxes.
  // The mask of this frame was cleared by the previous frame's k_resolve (or at allocation); this frame's clears the other.
  // Views: 0 = the camera; 1..6 = the faces of a cube around the light, when the shadow rays take entry lists (below).
  CoverViews cv{};
  CoverArgs& ca = cv.v[0];
  bool cover_on = c->primary_cover && rows > 0 && c->scene->max_cover_count > 0 && (n_shards == 1 || band_rows % 8 == 0) && c->cfg.variant != 1 &&
                  sc.n_inst <= 65535;   // (k_cover's grid has one row of workgroups per instance)
  const uint32_t cam_words = 1u + (uint32_t)((((size_t)((W + 7) / 8) * (size_t)((H + 7) / 8)) + 31) / 32);   // one camera view's mask
  // (a frame batch: every frame is a view of its own — its camera, its instance records, its part of the mask block)
  auto camera_view = [&](const float* position, const float* right, const float* up, const float* forward, CoverArgs& a) -> bool {
    const double R[9] = {right[0], up[0], forward[0], right[1], up[1], forward[1], right[2], up[2], forward[2]};   // columns right, up, forward
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    const double scale = std::fabs(R[0]) + std::fabs(R[1]) + std::fabs(R[2]) + std::fabs(R[3]) + std::fabs(R[4]) + std::fabs(R[5]) + std::fabs(R[6]) + std::fabs(R[7]) + std::fabs(R[8]);
    if (!(std::fabs(det) > 1e-6 * scale * scale * scale / 27.0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    const double inv[9] = {(R[4] * R[8] - R[5] * R[7]) * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id,
                           (R[5] * R[6] - R[3] * R[8]) * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[2] * R[3] - R[0] * R[5]) * id,
                           (R[3] * R[7] - R[4] * R[6]) * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
    for (int k = 0; k < 9; k++) a.inv[k] = (float)inv[k];
    for (int k = 0; k < 3; k++) a.cam[k] = position[k];
    a.kf = 2.5f;   // src/shader.rgen:79
    a.width = W; a.height = H; a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8; a.n_inst = n_inst1;
    a.mask_offset = 0; a.apex_radius = 0.0f; a.inst_base = 0;
    return true;
  };
  if (cover_on) {
    if (K == 1) { cover_on = camera_view(u.position, u.right, u.up, u.forward, ca); cv.n = 1; }
    else {
      for (int k = 0; k < K && cover_on; k++) {
        cover_on = camera_view(bt.position[k], bt.right[k], bt.up[k], bt.forward[k], cv.v[k]);
        cv.v[k].mask_offset = (uint32_t)k * cam_words; cv.v[k].inst_base = k * n_inst1;
      }
      cv.n = K;
    }
    if (!cover_on) cv.n = 0;
  }
  // Entry lists (k_entry) ride on the coverage mask: same tiles, same camera basis; the one-lane BVH2 kernel only.
  // (a record opens ONE instance's BLAS; where the beam of a tile meets many instances — cfg5's ring of 16 — the TLAS phase of k_entry costs
  // more than the records save: measured 1.86 vs 1.80 ms per frame, profiles/r03_experiments.txt — so records are for scenes of few instances)
  const bool entry_on = cover_on && c->entry_points && c->camera_records && c->cfg.variant == 0 && n_inst1 <= c->entry_max_instances;
  // far-ray logic in this frame's kernels only if some ray can be far (a re-render does not trust the context's current instance list: it carries the logic)
  bool far_frame = again != nullptr || far_possible(c, u);
  for (int k = 1; k < K && !far_frame; k++) far_frame = far_possible(c, c->batch_uni[k]);   // (against the instances of ALL frames: conservative)
  // ... and the tiles' blobs on the records (no far-ray logic in the LDS walk: such frames keep the global walk)
  const bool tile_on = entry_on && c->tile_blobs && !far_frame && K == 1;
  // ... or one walk per PIXEL from the records (k_beam: no far-ray logic either; the alternatives' queues are per ray)
  const bool beam_on = c->pixel_beams && !tile_on && !far_frame && c->cfg.variant == 0 && c->cfg.packet == 0;   // (without records the walks start at the TLAS root)
  // ... and for the shadow rays, which all end (within 0.01) at the light: a cube of light_tiles^2 tiles per face around it
  // (kept records are paid once, so they also serve scenes of more instances than the per-frame camera records are worth building for)
  const bool light_on = K == 1 && cover_on && c->entry_points && c->cfg.variant == 0 && (entry_on || c->shadow_entry == 2) && c->shadow_entry &&
                        std::isfinite(u.light_position[0]) && std::isfinite(u.light_position[1]) && std::isfinite(u.light_position[2]);
  const int LT = c->light_tiles;
  const uint32_t face_words = 1u + (uint32_t)(((size_t)LT * LT + 31) / 32);
  {
    const uint32_t words = std::max(cam_words + 6u * face_words, (uint32_t)K * cam_words);
    if (cover_on && words > c->cover_alloc_words) {
      if (c->d_cover_mask) { HIP_TRY(c, hipStreamSynchronize(c->stream)); if (s != c->stream) HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(c->d_cover_mask)); c->d_cover_mask = nullptr; c->cover_alloc_words = 0; }
      HIP_TRY(c, hipMalloc((void**)&c->d_cover_mask, 2 * (size_t)words * sizeof(uint32_t)));
      HIP_TRY(c, hipMemsetAsync(c->d_cover_mask, 0, 2 * (size_t)words * sizeof(uint32_t), c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->cover_alloc_words = words; c->cover_parity = 0;
    }
  }
  if (c->d_cover_mask && rows > 0) {
    // (a frame that does not use the mask still clears the next one, so that a later frame finds it clean)
    f.cover = cover_on ? c->d_cover_mask + (size_t)c->cover_parity * c->cover_alloc_words : nullptr;
    f.cover_next = c->d_cover_mask + (size_t)(c->cover_parity ^ 1) * c->cover_alloc_words;
    f.cover_words = c->cover_alloc_words; f.cover_tiles_x = (W + 7) / 8; f.cover_view_words = cam_words;
  }
  EntryViews ev{};
  if ((entry_on || light_on) && f.cover != nullptr) {
    const size_t tiles_local = entry_on ? (size_t)((W + 7) / 8) * (size_t)((rows + 7) / 8) : 0;   // of one frame
    if (tiles_local * K > c->entry_alloc_tiles) {
      if (c->d_entry) { HIP_TRY(c, hipStreamSynchronize(c->stream)); if (s != c->stream) HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(c->d_entry)); c->d_entry = nullptr; c->entry_alloc_tiles = 0; }
      HIP_TRY(c, hipMalloc((void**)&c->d_entry, tiles_local * K * sizeof(EntryRec)));
      c->entry_alloc_tiles = tiles_local * K;
    }
    f.entry = entry_on ? c->d_entry : nullptr;
    if (tile_on) {
      // directory (one word per tile), arena (N_SHARDS sub-arenas; a slot per tile that gets a blob — more tiles than slots: the rest take
      // the global walk) and the blob lists (one part per size class and sub-arena, a slot's worth of entries each)
      const uint32_t want_slots = (uint32_t)(((std::min<size_t>(tiles_local, tiles_local / 2 + 2048) + N_SHARDS - 1) / N_SHARDS + 8) * N_SHARDS);
      if (tiles_local > c->tile_dir_alloc || want_slots > c->blob_slots) {
        HIP_TRY(c, hipStreamSynchronize(c->stream)); if (s != c->stream) HIP_TRY(c, hipStreamSynchronize(s));
        if (c->d_tile_blob) HIP_TRY(c, hipFree(c->d_tile_blob));
        if (c->d_blob_arena) HIP_TRY(c, hipFree(c->d_blob_arena));
        if (c->d_blob_list) HIP_TRY(c, hipFree(c->d_blob_list));
        c->d_tile_blob = nullptr; c->d_blob_arena = nullptr; c->d_blob_list = nullptr; c->tile_dir_alloc = 0; c->blob_slots = 0;
        const size_t dir = std::max(tiles_local, c->entry_alloc_tiles);
        HIP_TRY(c, hipMalloc((void**)&c->d_tile_blob, dir * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void**)&c->d_blob_arena, (size_t)want_slots * BLOB_SLOT_BYTES));
        HIP_TRY(c, hipMalloc((void**)&c->d_blob_list, (size_t)BLOB_CLASSES * want_slots * sizeof(uint4)));
        c->tile_dir_alloc = dir; c->blob_slots = want_slots;
      }
      f.tile_blob = c->d_tile_blob; f.blob_arena = c->d_blob_arena; f.blob_slots = c->blob_slots; f.blob_list = c->d_blob_list;
    }
    // (without camera records view 0 stays empty: no tiles, no blocks with work; a frame batch: one view per frame)
    if (entry_on)
      for (int fk = 0; fk < K; fk++) {
        EntryArgs& ea = ev.v[fk];
        const CoverArgs& cak = cv.v[fk];
        const float* right = K > 1 ? bt.right[fk] : u.right; const float* up = K > 1 ? bt.up[fk] : u.up; const float* fwd = K > 1 ? bt.forward[fk] : u.forward;
        for (int k = 0; k < 3; k++) ea.cam[k] = cak.cam[k];
        for (int k = 0; k < 9; k++) ea.inv[k] = cak.inv[k];
        for (int k = 0; k < 3; k++) { ea.basis[k] = right[k]; ea.basis[3 + k] = up[k]; ea.basis[6 + k] = fwd[k]; }
        ea.kf = 2.5f; ea.apex_radius = 0.0f;
        ea.width = W; ea.height = H; ea.tiles_x = (W + 7) / 8; ea.tile_rows = (rows + 7) / 8;
        ea.band_rows = band_rows; ea.shard = shard; ea.n_shards = n_shards;
        ea.records = c->d_entry + (size_t)fk * tiles_local; ea.cover = f.cover + (size_t)fk * cam_words; ea.cover_tiles_x = f.cover_tiles_x;
        ea.tlas_root_offset = fk * sc.tlas_stride;
      }
    ev.n = entry_on ? K : 1;
    if (light_on) {
      const size_t light_tiles_total = (size_t)6 * LT * LT;
      if (light_tiles_total > c->light_alloc_tiles) {
        if (c->d_light_entry) { HIP_TRY(c, hipStreamSynchronize(c->stream)); if (s != c->stream) HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(c->d_light_entry)); c->d_light_entry = nullptr; c->light_alloc_tiles = 0; }
        HIP_TRY(c, hipMalloc((void**)&c->d_light_entry, light_tiles_total * sizeof(EntryRec)));
        c->light_alloc_tiles = light_tiles_total; c->light_built = false;
      }
      f.light_entry = c->d_light_entry; f.light_tiles = LT;   // (f.sh_e: allocated with the ray queues)
      // a shadow ray starts 0.01 N off the shaded point and runs parallel to the line from that point to the light
      // (src/shader.rgen:107-110): it ends within 0.01 (+ rounding) of the light, not in it
      const float apex_radius = 0.0102f;
      for (int face = 0; face < 6; face++) {
        const int axis = face >> 1, a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
        const float sgn = (face & 1) ? -1.0f : 1.0f;
        float basis[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        basis[a1] = 1.0f; basis[3 + a2] = 1.0f; basis[6 + axis] = sgn;   // R, U, F; orthonormal: the inverse is the transpose (rows R, U, F)
        CoverArgs& lc = cv.v[1 + face];
        EntryArgs& le = ev.v[1 + face];
        for (int k = 0; k < 3; k++) lc.cam[k] = le.cam[k] = u.light_position[k];
        for (int k = 0; k < 9; k++) { lc.inv[k] = le.inv[k] = basis[k]; le.basis[k] = basis[k]; }
        lc.kf = le.kf = 1.0f; lc.apex_radius = le.apex_radius = apex_radius;
        lc.width = lc.height = le.width = le.height = 8 * LT;
        lc.tiles_x = lc.tiles_y = LT; lc.n_inst = ca.n_inst;
        lc.mask_offset = cam_words + (uint32_t)face * face_words;
        le.tiles_x = le.tile_rows = LT; le.band_rows = 8 * LT; le.shard = 0; le.n_shards = 1;
        le.records = c->d_light_entry + (size_t)face * LT * LT;
        le.cover = f.cover + lc.mask_offset; le.cover_tiles_x = LT;
      }
      bool build = true;
      if (c->shadow_entry == 2) {
        rt_ctx::LightKey key;
        for (int k = 0; k < 3; k++) key.light[k] = u.light_position[k];
        key.gen = c->inst_gen[frame_parity]; key.parity = frame_parity; key.n_inst = sc.n_inst; key.tiles = LT;
        const bool have = c->light_built && key == c->light_key_built;
        // (the first frame of a context builds at once — an optimistic start: a scene that turns out to move drops the records at its second frame)
        const bool stable = !c->light_seen_valid || key == c->light_key_seen;
        if (!again) { c->light_key_seen = key; c->light_seen_valid = true; }
        if (again != nullptr) { build = false; if (!have) { f.light_entry = nullptr; f.light_tiles = 0; } }   // a re-render uses what exists, builds nothing
        else if (have) build = false;
        else if (stable) { c->light_key_built = key; c->light_built = true; }
        else { build = false; c->light_built = false; f.light_entry = nullptr; f.light_tiles = 0; }   // the scene moves: this frame's shadow rays walk from the TLAS root
      }
      if (build) { cv.n = ENTRY_VIEWS; ev.n = ENTRY_VIEWS; }
    }
  }
  // far-ray logic in this frame's kernels only if some ray can be far (a re-render decides again from the same inputs)
  LaunchCfg cfg = c->cfg;
  if (c->stack_need > 120) cfg.packet = 0;   // deeper than k_packet's 128-entry wave stack (a degenerate LBVH): the one-lane kernels spill to HBM instead
  cfg.far = far_frame ? 1 : 0;
  f.far_possible = cfg.far;
  f.settle_dead_shadow_rays = c->dead_shadow_rays;
  f.pixel_runs = beam_on ? 64 * (int)std::min<uint32_t>(u.samples_per_pixel, 4u) : 0;
  f.shadow_runs = (beam_on && c->shadow_beams && K == 1 && std::isfinite(u.light_position[0]) && std::isfinite(u.light_position[1]) && std::isfinite(u.light_position[2])) ? 1u : 0u;   // (a frame batch has a light per frame: one walk per shadow ray there)
  f.sh_base = f.shadow_runs ? (uint32_t)capacity : 0u;
  if (c->jitter_table && rows > 0) { r = jitter_table_for(c, f, u.samples_per_pixel, s, &f.jitter); if (r) return r; }
  // timing spans accumulate over frames until rt_get_stats reads (and averages) them; without a reader the
  // pool is recycled every 64 frames
  if (!c->timing || c->timed_frames >= 64) { c->ev_used = 0; c->spans.clear(); c->timed_frames = 0; }
  if (c->timing) c->timed_frames++;
  c->frame_stream = s; c->frame_pending = true;
  c->last_max_bounce = u.max_bounce_count;
  c->last_primary = (uint64_t)W * rows * u.samples_per_pixel * (uint64_t)K;
  c->last_empty = rows == 0;
  // the instance records / TLAS nodes of this slot were copied on the context's stream: a frame on another stream waits on the device
  if (c->upload_pending && s != c->stream) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_upload[c->parity], 0));   // (also orders a re-render behind newer uploads: harmless)
  if (rows == 0) return RT_OK;   // nothing is launched: both counter blocks stay zero
  c->cnt_parity ^= 1;
  if (f.cover_next) c->cover_parity ^= 1;   // (f.cover / f.cover_next were taken above)
  int cap_closest = c->closest_blocks_per_cu, cap_shadow = c->shadow_blocks_per_cu;
  {
    constexpr unsigned long long GRID_SMALL_RAYS = 2500000ull;
    const bool crowded = c->scene->members.size() >= 3 && !c->grid_user_set && !c->last_empty;
    const unsigned long long prev_q0 = ((volatile unsigned long long*)c->h_stats)[STAT_QUEUE0], prev_sh = ((volatile unsigned long long*)c->h_stats)[STAT_SHADOW];
    if (cap_closest < 0) cap_closest = (crowded && prev_q0 > 0 && prev_q0 <= GRID_SMALL_RAYS) ? 2 : 0;
    if (cap_shadow < 0) cap_shadow = (crowded && f.light_entry != nullptr && prev_sh > 0 && prev_sh <= GRID_SMALL_RAYS) ? 2 : 0;
  }
  {
    Span frame_span(c, CAT_FRAME, s);
    {
      Span sp(c, CAT_RAYGEN, s);
      if (f.cover) launch_cover(sc, cv, c->scene->max_cover_count, const_cast<uint32_t*>(f.cover), s);
      if (f.entry || ev.n > 1) launch_entry(sc, ev, s);
      if (f.tile_blob) launch_blob(sc, ev.v[0], f, c->counting, s);
      launch_raygen(sc, f, u, bt, s);
    }
    // k_tail takes over at the first bounce whose queue was small in the previous frame of this context (a hint:
    // either strategy gives the same image); bounces before it run on the full persistent grid
    uint32_t tail_start = 0xFFFFFFFFu;
    const bool tail_ok = c->cfg.variant != 1 && c->tail_blocks > 0 && !no_tail && !c->tail_disabled;
    if (tail_ok && c->tail_mode == 2) tail_start = 1;
    else if (tail_ok && c->tail_mode == 1)
      for (uint32_t b = 1; b <= u.max_bounce_count && b < (uint32_t)CNT_MAX_BOUNCES; b++)
        if (((volatile uint32_t*)c->h_hint)[b] <= TAIL_MAX_RAYS) { tail_start = b; break; }
    for (uint32_t b = 0; b <= u.max_bounce_count; b++) {
      if (b == tail_start) {
        // every later bounce in one launch (src/shader.rgen:84 loop), leaving as soon as a queue is empty
        // the grid follows the work: one lane per ray of the first tail bounce in the slot's previous frame, 1..tail_blocks
        // workgroups (any count works: the bodies steal from the other shards) — the workgroups of k_tail hold their CU
        // resources while they wait at the grid barriers, so a 64-workgroup grid for the 260 secondary rays of a 1/8 shard
        // mostly waits (1/8 shard with 16 slots in flight: 0.100 ms with 64 workgroups, 0.091 with 8, 0.088 with 2)
        // (contexts created since — other scenes on this GPU included — shrink every slot's share of the resident workgroups)
        int tb = std::min(c->tail_blocks, tail_grid(c->n_cu, c->tail_resident_per_cu, live_slots_on(c->device)));
        if (tb <= 0) { { Span sp(c, CAT_TRACE, s); launch_trace_closest(sc, f, (int)b, c->counting, cfg, s); } { Span sp(c, CAT_SHADE, s); launch_shade(sc, f, u, bt, (int)b, cfg, s); } continue; }
        const uint32_t expect = ((volatile uint32_t*)c->h_hint)[b];
        // (a lone slot keeps the full grid: nobody else needs the room and 64 workgroups finish 2 k rays in 43 us, 9 in 55)
        if (c->tail_mode == 1 && expect != 0xFFFFFFFFu && !c->tail_full_grid && c->scene->members.size() >= 3) {
          long want = ((long)expect + 255) / 256;
          const long lo = c->tail_min_blocks;
          if (lo >= N_SHARDS) want = (want + (N_SHARDS - 1)) / N_SHARDS * N_SHARDS;
          tb = (int)std::min<long>(tb, std::max<long>(lo, want));
        }
        Span sp(c, CAT_TAIL, s); launch_tail(sc, f, u, bt, (int)b, c->counting, cfg, tb, s);
        break;
      }
      { LaunchCfg cc = cfg; if (b == 0 && cap_closest > 0) cc.trace_blocks = std::min(cfg.trace_blocks, c->n_cu * cap_closest);
        Span sp(c, CAT_TRACE, s);
        if (b == 0 && f.tile_blob) launch_tile(sc, f, u, c->counting, s);   // the tiles with a blob: their rays generated and walked in LDS (it hands a few on to queue 0)
        launch_trace_closest(sc, f, (int)b, c->counting, cc, s); }
      { Span sp(c, CAT_SHADE, s); launch_shade(sc, f, u, bt, (int)b, cfg, s); }
      if (b >= 7 && (b & 3) == 3 && b < u.max_bounce_count) {
        // deep bounce budgets (the reference default is 63): stop launching once every path has ended
        uint32_t tails[N_SHARDS * CNT_STRIDE];
        HIP_TRY(c, hipMemcpyAsync(tails, f.counters + cnt_tail((int)b + 1, 0), sizeof(tails), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        uint32_t live = 0;
        for (int t = 0; t < N_SHARDS; t++) live += tails[t * CNT_STRIDE];
        if (live == 0) break;
      }
    }
    { LaunchCfg cs = cfg; if (cap_shadow > 0) cs.trace_blocks = std::min(cfg.trace_blocks, c->n_cu * cap_shadow);
      Span sp(c, CAT_SHADOW, s); launch_beam_shadow(sc, f, u, c->counting, cs, s); launch_trace_shadow(sc, f, c->counting, cs, s); }
    { Span sp(c, CAT_RESOLVE, s); launch_resolve(f, u, s); }
  }
  HIP_TRY(c, hipGetLastError());
  // the instance records / TLAS nodes of this parity are in use until here
  if (!c->ev_frame[frame_parity]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_frame[frame_parity], hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(c->ev_frame[frame_parity], s));
  c->ev_frame_valid[frame_parity] = true;
  return RT_OK;
}

int collect_stats(rt_ctx* c) {
  if (!c->frame_pending) return RT_OK;
  // k_resolve has written the frame's statistics block to host-mapped memory: nothing to copy.  (A device-to-host
  // copy of the counters here, behind the pixel copy of rt_trace_async, serialised the frames of other contexts.)
  HIP_TRY(c, hipStreamSynchronize(c->frame_stream));
  c->upload_pending = false;   // the frame waited for the copies: staging and device records may be rewritten
  const unsigned long long* hs = c->h_stats;
  rt_stats st{};
  st.rays_primary = c->last_primary;
  if (!c->last_empty) {
    st.rays_secondary = hs[STAT_SECONDARY];
    st.rays_shadow = hs[STAT_SHADOW]; st.rays_shadow_untraced = hs[STAT_SHADOW_UNTRACED];
    // k_tail faults: the never-reset total tells of EVERY frame of this context whose grid barrier gave up (its workgroups were not
    // co-resident: another process on the GPU, a partition smaller than the occupancy query promised) — also of frames enqueued
    // before the last one, whose own statistics a later k_resolve has overwritten.  Any fault keeps this context off k_tail from
    // now on; the frame that is re-rendered (one launch per bounce and kernel, from the state it was submitted with) is the LAST
    // one: earlier frames of the stream have been overwritten in d_out by their successors or were the caller's to collect one
    // by one (rt_api.h: "a frame is complete after the rt_synchronize / rt_get_stats / rt_trace_wait that follows it").
    const uint64_t total = hs[STAT_FAULT_TOTAL];
    const bool forced = c->debug_force_tail_fault && !c->tail_disabled;
    if (total != c->faults_seen || forced) {
      c->tail_faults += (uint32_t)(total - c->faults_seen) + ((forced && hs[STAT_FAULT] == 0) ? 1u : 0u);
      c->faults_seen = total;
      c->tail_disabled = true;
    }
    if (hs[STAT_FAULT] != 0 || forced) {
      c->debug_force_tail_fault = false;
      const rt_ctx::LastFrame lf = c->last_frame;
      if (c->inst_gen[lf.parity] != lf.inst_gen) {
        c->frame_pending = false;
        return fail(c, RT_ERR_DEVICE, "a k_tail grid barrier gave up and the frame cannot be rendered again: rt_set_instances replaced its instance records "
                                      "twice before the frame was collected (collect every frame, or update the instances at most once per frame)");
      }
      const bool counting = c->counting;
      c->counting = lf.counting;
      c->ev_used = 0; c->spans.clear(); c->timed_frames = 0;
      int r = enqueue_frame(c, lf.W, lf.H, lf.band_rows, lf.shard, lf.n_shards, lf.d_out, c->frame_stream, &lf);
      c->counting = counting;
      if (r) { c->frame_pending = false; return r; }
      c->frame_rerendered = true;
      return collect_stats(c);
    }
    // rays that went through the closest-hit traversal kernel: primary rays that survived the TLAS
    // test fused into k_raygen (queue 0) plus every secondary ray
    st.closest_rays = hs[STAT_QUEUE0] + st.rays_secondary;
    st.node_visits = hs[STAT_NODE_VISITS]; st.tri_tests = hs[STAT_TRI_TESTS];
    st.node_visits_shadow = hs[STAT_NODE_VISITS_SH]; st.tri_tests_shadow = hs[STAT_TRI_TESTS_SH];
    for (int k = 0; k < 6; k++) st.diag[k] = hs[STAT_DIAG + k];
    st.blob_tiles = hs[STAT_BLOB] + hs[STAT_BLOB + 1]; st.blob_tiles_large = hs[STAT_BLOB + 1]; st.blob_tiles_refused = hs[STAT_BLOB + 2]; st.blob_nodes = hs[STAT_BLOB + 3]; st.blob_tris = hs[STAT_BLOB + 4];
    st.tile_rays = hs[STAT_TILE_RAYS]; st.tile_rays_handed_on = hs[STAT_CONT_RAYS];
    for (int k = 0; k < 6; k++) st.tile_diag[k] = hs[STAT_TILE_DIAG + k];
  }
  st.bvh_node_bytes = c->cfg.variant == 1 ? sizeof(Bvh4Node) : c->cfg.variant == 2 ? sizeof(WideNodeQ) : sizeof(BvhNodeQ); st.bvh_tri_bytes = sizeof(TriPacket);
  for (auto& sp : c->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
    switch (sp.cat) {
      case CAT_RAYGEN: st.ms_raygen += ms; break;
      case CAT_TRACE: st.ms_trace_closest += ms; st.launches_trace_closest++; break;
      case CAT_SHADE: st.ms_shade += ms; break;
      case CAT_SHADOW: st.ms_trace_shadow += ms; break;
      case CAT_RESOLVE: st.ms_resolve += ms; break;
      case CAT_FRAME: st.ms_frame += ms; break;
      case CAT_TAIL: st.ms_tail += ms; break;
    }
    if (sp.cat != CAT_FRAME) st.launches_total++;
  }
  if (c->timed_frames > 1) {   // mean per frame over every frame recorded since the last read
    const float k = 1.0f / (float)c->timed_frames;
    st.ms_raygen *= k; st.ms_trace_closest *= k; st.ms_shade *= k; st.ms_trace_shadow *= k; st.ms_resolve *= k; st.ms_frame *= k; st.ms_tail *= k;
    st.launches_trace_closest /= c->timed_frames; st.launches_total /= c->timed_frames;
  }
  st.timed_frames = c->timed_frames;
  st.tail_faults = c->tail_faults;
  st.frames_rerendered = c->frame_rerendered ? 1u : 0u;
  c->ev_used = 0; c->spans.clear(); c->timed_frames = 0;
  c->last = st;
  c->frame_pending = false;
  return RT_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int rt_abi_version(void) { return 7; }   // 2: rt_trace_async / rt_trace_wait; 3: rt_stats::tail_faults, rt_debug_sizing; 4: frame slots, rt_assemble_shards, materials; 5: rt_stats::frames_rerendered, entry records, BGRA8

// Persistent traversal grid, workgroups per CU.  A lone context renders one frame at a time: the kernels are latency-bound and
// 5 workgroups per CU (all the LDS admits) are fastest (cfg3: 1.00 ms vs 1.03 at 4, 1.28 at 2).  With several frame slots the
// frames overlap, and a traversal kernel that parks fewer workgroups leaves registers and wave slots to the other frames'
// kernels: 3 per CU (cfg3 with 4 slots: 0.642 ms per frame vs 0.667 at 4; cfg4 -1 %, limbs mesh -2 %, cfg5 +1 %).
static void size_traversal_grids(Scene* S) {
  const int per_cu = S->members.size() >= 3 ? 3 : 5;
  for (rt_ctx* m : S->members)
    if (!m->grid_user_set) m->cfg.trace_blocks = m->n_cu * per_cu;
}

static int create_context(rt_ctx** out_ctx, int device_id, rt_ctx* parent) {
  if (!out_ctx) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device available (librt_mi355x has no CPU fallback)");
  if (device_id < 0 || device_id >= n) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "device_id out of range");
  HIP_TRY(nullptr, hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(nullptr, RT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library contains gfx950 code only");
  int slot = 0;
  if (parent) {
    while (slot < MAX_SLOTS && (parent->scene->slot_mask >> slot & 1u)) slot++;
    if (slot == MAX_SLOTS) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "a scene holds at most " + std::to_string(MAX_SLOTS) + " frame slots");
  }
  rt_ctx* c = new rt_ctx();
  c->device = device_id;
  c->n_cu = prop.multiProcessorCount;
  c->info = std::string("gfx950 ") + prop.name + " CUs=" + std::to_string(prop.multiProcessorCount);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, RT_ERR_DEVICE, "hipStreamCreate failed"); }
  // persistent grids: ~27 KB of LDS and <= 84 VGPRs per 256-thread block admit 5 blocks per CU; size_traversal_grids() below
  // picks 5 per CU for a lone frame slot and 3 once three or more slots share the GPU (the other frames' kernels need room)
  c->cfg.trace_blocks = c->n_cu * 4;
  c->cfg.shade_blocks = c->n_cu * 8;
  c->cfg.rays_per_lane = 4; c->cfg.min_blocks = c->n_cu;
  // default traversal kernel: 0 = one lane per ray over quantized BVH2 nodes (fastest measured); 1 = quad/BVH4
  c->cfg.variant = 0;
  // k_packet (one wavefront per 64-ray chunk) is built and tested but OFF: measured slower (cfg3 lone frame 1.47 vs 0.95 ms;
  // profiles/r03_experiments.txt) — a packet visits the union of its rays' walks, 63 nodes + 33 triangles per 64 primary rays,
  // and its wave-uniform control flow runs on the CU's single scalar unit
  c->cfg.packet = 0; c->cfg.packet_blocks = c->n_cu * 8;
  if (const char* env = getenv("RT_PACKET")) c->cfg.packet = alt_kernels_built() ? std::max(0, std::min(2, atoi(env))) : 0;
  if (const char* env = getenv("RT_PACKET_BLOCKS_PER_CU")) { const int v = atoi(env); if (v > 0 && v <= 16) c->cfg.packet_blocks = c->n_cu * v; }
  c->tail_resident_per_cu = tail_blocks_per_cu();
  live_slots_add(device_id, 1);
  c->tail_blocks = tail_grid(c->n_cu, c->tail_resident_per_cu, live_slots_on(device_id));   // 0: device too small for k_tail's co-residency guarantee
  if (const char* env = getenv("RT_BLAS_BUILDER")) c->blas_builder = atoi(env) ? 1 : 0;
  if (const char* env = getenv("RT_TAIL_MIN_BLOCKS")) c->tail_min_blocks = std::max(1, atoi(env));
  if (getenv("RT_TAIL_FULL_GRID")) c->tail_full_grid = true;
  if (const char* env = getenv("RT_TRACE_VARIANT")) { const int v = atoi(env); c->cfg.variant = (v >= 0 && v <= 2 && alt_kernels_built()) ? v : 0; }
  if (const char* env = getenv("RT_TRACE_BLOCKS_PER_CU")) { int v = atoi(env); if (v > 0 && v <= 8) c->cfg.trace_blocks = c->n_cu * v; }
  if (parent) {
    c->scene = parent->scene;
    c->cfg = parent->cfg; c->blas_builder = parent->blas_builder; c->tail_mode = parent->tail_mode; c->tail_min_blocks = parent->tail_min_blocks; c->tail_full_grid = parent->tail_full_grid; c->primary_cover = parent->primary_cover; c->jitter_table = parent->jitter_table; c->tile_blobs = parent->tile_blobs; c->pixel_beams = parent->pixel_beams; c->camera_records = parent->camera_records; c->dead_shadow_rays = parent->dead_shadow_rays; c->shadow_beams = parent->shadow_beams; c->entry_points = parent->entry_points; c->shadow_entry = parent->shadow_entry; c->entry_max_instances = parent->entry_max_instances; c->light_tiles = parent->light_tiles; c->out_rgba8 = parent->out_rgba8; c->out_bgra = parent->out_bgra;
  } else {
    c->scene = new Scene();
    c->scene->device = device_id;
  }
  c->slot = slot;
  c->scene->slot_mask |= 1u << slot;
  c->scene->members.push_back(c);
  if (getenv("RT_TRACE_BLOCKS_PER_CU")) c->grid_user_set = true;
  if (parent) c->grid_user_set = parent->grid_user_set;
  size_traversal_grids(c->scene);
  *out_ctx = c;
  return RT_OK;
}

int rt_create(rt_ctx** out_ctx, int device_id) { return create_context(out_ctx, device_id, nullptr); }

int rt_create_frame_slot(rt_ctx* parent, rt_ctx** out_ctx) {
  if (!parent) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "parent context is NULL");
  return create_context(out_ctx, parent->device, parent);
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  live_slots_add(c->device, -1);
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  FrameDev& f = c->frame;
  void* ptrs[] = {c->d_inst[0], c->d_inst[1], c->d_out_own, c->d_counters, c->d_ovf, c->d_cover_mask, c->d_entry, c->d_light_entry, f.sh_e, c->d_fault_total, c->d_tile_blob, c->d_blob_arena, c->d_blob_list,
                  f.ray_o[0], f.ray_o[1], f.ray_d[0], f.ray_d[1], f.hit_a, f.hit_inst, f.sh_o, f.sh_d, f.sh_c, f.sample_color};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->h_hint) hipHostFree(c->h_hint);
  if (c->h_out_pinned) hipHostFree(c->h_out_pinned);
  if (c->h_stats) hipHostFree(c->h_stats);
  for (int k = 0; k < 2; k++) { if (c->h_stage[k]) hipHostFree(c->h_stage[k]); if (c->ev_frame[k]) hipEventDestroy(c->ev_frame[k]); }
  for (int k = 0; k < 2; k++) if (c->ev_upload[k]) hipEventDestroy(c->ev_upload[k]);
  for (auto e : c->ev_pool) hipEventDestroy(e);
  if (c->stream) hipStreamDestroy(c->stream);
  Scene* S = c->scene;
  S->members.erase(std::remove(S->members.begin(), S->members.end(), c), S->members.end());
  S->slot_mask &= ~(1u << c->slot);
  size_traversal_grids(S);
  if (S->members.empty()) {   // the last context of a scene takes the shared arrays with it
    void* sp[] = {S->d_wide, S->d_nodes4, S->d_verts, S->d_idx, S->d_blas_nodes, S->d_tris, S->d_sky, S->d_materials, S->d_prim_material, S->d_cover_boxes};
    for (void* p : sp) if (p) hipFree(p);
    for (JitterTable& t : S->jitter_tables) { hipFree(t.d); hipEventDestroy(t.ready); }
    delete S;
  }
  delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }
const char* rt_device_info(const rt_ctx* c) { return c ? c->info.c_str() : ""; }

int rt_upload_geometry(rt_ctx* c, const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx,
                       const rt_mesh_range* ranges, int n_meshes) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (!verts6 || !idx || !ranges || n_meshes <= 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "null geometry pointers or no meshes");
  { int q = quiesce_scene(c); if (q) return q; }
  Scene* S = c->scene;
  if (n_floats % 6 != 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "vertex buffer must hold 6 floats per vertex");
  HIP_TRY(c, hipSetDevice(c->device));
  for (int m = 0; m < n_meshes; m++) {
    const rt_mesh_range& r = ranges[m];
    if (r.first_index + 3ull * r.prim_count > n_idx || r.first_float % 6 != 0 || r.first_float > n_floats)
      return fail(c, RT_ERR_INVALID_ARGUMENT, "mesh range " + std::to_string(m) + " exceeds the buffers");
    const uint64_t nv = (n_floats - r.first_float) / 6;
    for (uint64_t k = 0; k < 3ull * r.prim_count; k++)
      if (idx[r.first_index + k] >= nv) return fail(c, RT_ERR_INVALID_ARGUMENT, "index out of range in mesh " + std::to_string(m));
  }
  S->h_verts.assign(verts6, verts6 + n_floats);
  S->h_idx.assign(idx, idx + n_idx);
  if (S->d_verts) HIP_TRY(c, hipFree(S->d_verts));
  if (S->d_idx) HIP_TRY(c, hipFree(S->d_idx));
  S->d_verts = nullptr; S->d_idx = nullptr;
  HIP_TRY(c, hipMalloc((void**)&S->d_verts, std::max<size_t>(n_floats, 6) * sizeof(float)));
  HIP_TRY(c, hipMalloc((void**)&S->d_idx, std::max<size_t>(n_idx, 3) * sizeof(uint32_t)));
  HIP_TRY(c, hipMemcpy(S->d_verts, verts6, n_floats * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(S->d_idx, idx, n_idx * sizeof(uint32_t), hipMemcpyHostToDevice));
  S->meshes.assign(n_meshes, Mesh{});
  for (int m = 0; m < n_meshes; m++) S->meshes[m].range = ranges[m];
  S->blas_linked = false; invalidate_tlas(S);
  // the per-triangle material ids belong to the old index buffer
  if (S->d_materials) { HIP_TRY(c, hipFree(S->d_materials)); S->d_materials = nullptr; }
  if (S->d_prim_material) { HIP_TRY(c, hipFree(S->d_prim_material)); S->d_prim_material = nullptr; }
  S->n_materials = 0; S->n_prim_material = 0;
  return RT_OK;
}

int rt_build_blas(rt_ctx* c, int mesh) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  Scene* S = c->scene;
  if (mesh < 0 || mesh >= (int)S->meshes.size()) return fail(c, RT_ERR_INVALID_ARGUMENT, "mesh index out of range");
  { int q = quiesce_scene(c); if (q) return q; }
  Mesh& m = S->meshes[mesh];
  m.gpu_built = false;
  if (c->blas_builder == 1 && c->cfg.variant != 1 && m.range.prim_count >= 8) {
    // device build straight from the uploaded vertex/index buffers; the result is downloaded once so that the
    // linker treats every mesh alike
    HIP_TRY(c, hipSetDevice(c->device));
    GpuBlas g; std::string err;
    if (build_blas_gpu(S->d_verts + m.range.first_float, S->d_idx + m.range.first_index, m.range.prim_count, c->stream, g, err))
      return fail(c, RT_ERR_DEVICE, err);
    m.qnodes.resize(g.n_nodes); m.tris.resize(g.n_tris);
    hipError_t e1 = hipMemcpy(m.qnodes.data(), g.nodes, (size_t)g.n_nodes * sizeof(BvhNodeQ), hipMemcpyDeviceToHost);
    hipError_t e2 = hipMemcpy(m.tris.data(), g.tris, (size_t)g.n_tris * sizeof(TriPacket), hipMemcpyDeviceToHost);
    for (int k = 0; k < 3; k++) { m.q_lo[k] = g.q_lo[k]; m.q_scale[k] = g.q_scale[k]; m.bounds.lo[k] = g.bounds_lo[k]; m.bounds.hi[k] = g.bounds_hi[k]; }
    free_blas_gpu(g);
    HIP_TRY(c, e1); HIP_TRY(c, e2);
    m.bvh = BuiltBvh{}; m.bvh4 = Bvh4{};
    m.gpu_built = true;
  } else {
    build_blas(S->h_verts.data() + m.range.first_float, S->h_idx.data() + m.range.first_index, m.range.prim_count, m.bvh, m.tris);
    collapse_bvh4(m.bvh, true, false, m.bvh4);
    m.bounds = m.bvh.bounds;
  }
  m.built = true;
  S->blas_linked = false; invalidate_tlas(S);
  return RT_OK;
}

// the instances of K frames (K = 1: rt_set_instances; K > 1: a frame batch, frame k's n instances at inst + k * n)
static int set_instances_frames(rt_ctx* c, const rt_instance* inst, int n, int update, int K) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (!inst || n <= 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "no instances");
  HIP_TRY(c, hipSetDevice(c->device));
  Scene* S = c->scene;
  // Double-buffered records: this call fills the set the frame enqueued LAST does not read, so it waits only for the frame
  // before that one (usually long finished) — a per-frame update never stalls the host on the frame in flight.
  const int next_parity = c->parity ^ 1;
  if (c->ev_frame_valid[next_parity]) { HIP_TRY(c, hipEventSynchronize(c->ev_frame[next_parity])); c->ev_frame_valid[next_parity] = false; }
  if (c->upload_inflight[next_parity]) { HIP_TRY(c, hipEventSynchronize(c->ev_upload[next_parity])); c->upload_inflight[next_parity] = false; }   // its staging buffer is rewritten
  const int total = n * K;
  for (int i = 0; i < total; i++) {
    if (inst[i].mesh >= S->meshes.size()) return fail(c, RT_ERR_INVALID_ARGUMENT, "instance references an unknown mesh");
    if (!S->meshes[inst[i].mesh].built) return fail(c, RT_ERR_NOT_READY, "instance references a mesh whose BLAS is not built (rt_build_blas)");
  }
  if (update && (!c->tlas_valid || c->inst_per_frame != n))
    return fail(c, RT_ERR_INVALID_ARGUMENT, "TLAS update needs a previous build with the same instance count");
  if (K > 1 && c->cfg.variant != 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame batches need trace_variant 0");
  if (!S->blas_linked) {   // (re)linking moves the shared arrays: every slot of the scene has to be idle
    int q = quiesce_scene(c); if (q) return q;
    int r = link_blas(c); if (r) return r;
  }
  // from here on the slot's host-side TLAS state is being replaced: an error return leaves it INVALID (rt_set_instances
  // has to be called again) instead of half old, half new
  c->tlas_valid = false;
  c->h_inst.assign(inst, inst + total);
  std::vector<InstanceDev> inst_dev(total);
  std::vector<Aabb> boxes(total);
  for (int i = 0; i < total; i++) {
    InstanceDev& d = inst_dev[i];
    const Mesh& m = S->meshes[inst[i].mesh];
    memcpy(d.o2w, inst[i].transform, sizeof(d.o2w));
    invert_affine(d.o2w, d.w2o);
    d.blas_root = m.root;
    d.blas_root4 = m.node_base4;
    d.mask = m.range.prim_count ? (inst[i].custom_index_and_mask >> 24) : 0u;   // an empty mesh is never entered
    for (int k = 0; k < 3; k++) { d.q_lo[k] = m.q_lo[k]; d.q_scale[k] = m.q_scale[k]; }
    d.custom_index = (int32_t)(inst[i].custom_index_and_mask & 0xFFFFFFu);
    d.first_float = (uint32_t)m.range.first_float;
    d.first_index = (uint32_t)m.range.first_index;
    d.type = (size_t)(i % n) < c->inst_types.size() ? c->inst_types[i % n] : TYPE_BY_OBJECT_INDEX;
    d.cover_first = m.cover_first; d.cover_count = m.range.prim_count ? m.cover_count : 0u; d.pad = 0;
    boxes[i] = instance_world_box(d.o2w, m.bounds);
  }
  // frame 0 builds (or refits) the context's TLAS; the other frames of a batch are refits of that topology to their own boxes
  std::vector<BuiltBvh> frame_trees;
  if (update) { refit_bvh(boxes.data(), c->tlas); refit_bvh4(c->tlas, c->tlas4); }
  else { build_bvh(boxes.data(), (uint32_t)n, 1, 20, c->tlas); collapse_bvh4(c->tlas, false, true, c->tlas4); }
  if (K > 1) {
    frame_trees.resize(K);
    frame_trees[0] = c->tlas;
    for (int k = 1; k < K; k++) { frame_trees[k] = c->tlas; refit_bvh(boxes.data() + (size_t)k * n, frame_trees[k]); }
  }
  // the quad traversal keeps its whole stack in LDS: bottom sentinel + TLAS + marker + deepest BLAS
  int blas_need = 0, blas_levels = 0;
  for (int i = 0; i < total; i++) {
    const Mesh& m = S->meshes[inst[i].mesh];
    if (m.gpu_built && c->cfg.variant == 1) return fail(c, RT_ERR_INVALID_ARGUMENT, "device-built BLAS is not traversed by trace_variant 1: set trace_variant before rt_build_blas or use blas_builder 0");
    blas_need = std::max(blas_need, m.bvh4.stack_need);
    blas_levels = std::max(blas_levels, m.levels);
  }
  {
    // one-lane kernels: bottom sentinel + TLAS + leave-instance marker + deepest BLAS; the 4-ary records push up
    // to three links per two BVH2 levels.  Whatever exceeds the LDS part of the stack spills to ovf_stride entries.
    const int tl = c->tlas.depth + 1;
    const int need2 = 2 + tl + blas_levels, needw = 2 + 3 * ((tl + 1) / 2) + 3 * ((blas_levels + 1) / 2);
    c->ovf_stride = (uint32_t)std::max<int>(STACK_OVF, (std::max(need2, needw) + 7) & ~7);
    c->stack_need = need2 + ENTRY_WORDS + 2;   // (+ what an entry record puts on the stack before the walk starts)
  }
  if (1 + c->tlas4.stack_need + 1 + blas_need > STACK4_LDS)
    return fail(c, RT_ERR_INVALID_ARGUMENT, "acceleration structure needs " + std::to_string(2 + c->tlas4.stack_need + blas_need) +
                    " traversal-stack entries, more than the " + std::to_string((int)STACK4_LDS) + " the kernel keeps in LDS");
  c->parity = next_parity;
  c->inst_gen[next_parity]++;
  c->inst_per_frame = n; c->batch_k = K;
  int r = upload_instances(c, inst_dev, K > 1 ? &frame_trees : nullptr); if (r) return r;
  c->tlas_valid = true;
  return RT_OK;
}

int rt_set_instances(rt_ctx* c, const rt_instance* inst, int n, int update) { return set_instances_frames(c, inst, n, update, 1); }

int rt_set_batch(rt_ctx* c, int n_frames, const rt_instance* instances, int n, const rt_uniforms* uniforms, int update) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (n_frames < 1 || n_frames > BATCH_MAX) return fail(c, RT_ERR_INVALID_ARGUMENT, "a batch holds 1.." + std::to_string((int)BATCH_MAX) + " frames");
  if (!uniforms) return fail(c, RT_ERR_INVALID_ARGUMENT, "no uniforms");
  static_assert(sizeof(rt_uniforms) == sizeof(UniformsDev), "rt_uniforms layout");
  for (int k = 1; k < n_frames; k++)
    if (uniforms[k].max_bounce_count != uniforms[0].max_bounce_count || uniforms[k].samples_per_pixel != uniforms[0].samples_per_pixel ||
        uniforms[k].center_object_type != uniforms[0].center_object_type || uniforms[k].orbiting_object_type != uniforms[0].orbiting_object_type)
      return fail(c, RT_ERR_INVALID_ARGUMENT, "the frames of a batch share maxBounceCount, samplesPerPixel and the object types (camera, light and instances may differ)");
  int r = set_instances_frames(c, instances, n, update, n_frames); if (r) return r;
  c->batch_uni.resize(n_frames);
  memcpy(c->batch_uni.data(), uniforms, (size_t)n_frames * sizeof(UniformsDev));
  c->uni = c->batch_uni[0]; c->have_uni = true;
  return RT_OK;
}


// Row n4: the MTL materials the reference's loader parses and its renderer ignores (src/shader.rgen:51-55
// hard-codes kd, ks, 100, 1.52).  table[prim_material[g]] shades triangle g of the index buffer (g = first_index / 3 +
// gl_PrimitiveID).  n_materials == 0 removes the table: the reference's constants again.
int rt_set_materials(rt_ctx* c, const rt_material* table, int n_materials, const uint32_t* prim_material, size_t n_prims) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (n_materials < 0 || (n_materials > 0 && (!table || !prim_material))) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_set_materials arguments");
  { int q = quiesce_scene(c); if (q) return q; }
  HIP_TRY(c, hipSetDevice(c->device));
  Scene* S = c->scene;
  if (n_materials > 0) {
    if (!S->d_idx) return fail(c, RT_ERR_NOT_READY, "rt_upload_geometry has not been called");
    if (n_prims != S->h_idx.size() / 3) return fail(c, RT_ERR_INVALID_ARGUMENT, "prim_material needs one entry per triangle of the index buffer (" + std::to_string(S->h_idx.size() / 3) + ")");
    for (size_t k = 0; k < n_prims; k++)
      if (prim_material[k] >= (uint32_t)n_materials) return fail(c, RT_ERR_INVALID_ARGUMENT, "material index out of range at triangle " + std::to_string(k));
    for (int m = 0; m < n_materials; m++) {
      const rt_material& t = table[m];
      if (!(t.type <= 2u || t.type == TYPE_BY_INSTANCE)) return fail(c, RT_ERR_INVALID_ARGUMENT, "material " + std::to_string(m) + ": type must be 0, 1, 2 or RT_MATERIAL_TYPE_OF_INSTANCE");
      if (!(t.ni > 0.0f) || !(t.ns >= 0.0f)) return fail(c, RT_ERR_INVALID_ARGUMENT, "material " + std::to_string(m) + ": Ni must be > 0 and Ns >= 0");
    }
  }
  if (S->d_materials) { HIP_TRY(c, hipFree(S->d_materials)); S->d_materials = nullptr; }
  if (S->d_prim_material) { HIP_TRY(c, hipFree(S->d_prim_material)); S->d_prim_material = nullptr; }
  S->n_materials = 0; S->n_prim_material = 0;
  if (n_materials == 0) return RT_OK;
  std::vector<MaterialDev> dev(n_materials);
  for (int m = 0; m < n_materials; m++) {
    memcpy(&dev[m], &table[m], sizeof(MaterialDev));
    dev[m].ns = std::floor(std::min(std::max(table[m].ns, 0.0f), 1023.0f) + 0.5f);   // the exponent is applied as an integer power
  }
  HIP_TRY(c, hipMalloc((void**)&S->d_materials, dev.size() * sizeof(MaterialDev)));
  HIP_TRY(c, hipMalloc((void**)&S->d_prim_material, std::max<size_t>(1, n_prims) * sizeof(uint32_t)));
  HIP_TRY(c, hipMemcpy(S->d_materials, dev.data(), dev.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));
  if (n_prims) HIP_TRY(c, hipMemcpy(S->d_prim_material, prim_material, n_prims * sizeof(uint32_t), hipMemcpyHostToDevice));
  S->n_materials = n_materials; S->n_prim_material = n_prims;
  return RT_OK;
}

// Per-instance object type (0 diffuse, 1 mirror, 2 refractive) for this context's instances, replacing the reference's
// "objectIndex == 0 ? centerObjectType : orbitingObjectType" (src/shader.rgen:96, "Hardcoded as 2 objects" src/main.cpp:2425).
// Takes effect with the next rt_set_instances; n == 0 restores the two-way switch.
int rt_set_instance_types(rt_ctx* c, const uint32_t* types, int n) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (n < 0 || (n > 0 && !types)) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_set_instance_types arguments");
  for (int i = 0; i < n; i++) if (types[i] > 2u) return fail(c, RT_ERR_INVALID_ARGUMENT, "instance type must be 0, 1 or 2");
  c->inst_types.assign(types, types + n);
  if (c->tlas_valid && !c->h_inst.empty()) {   // re-issue the instance records with the new types
    std::vector<rt_instance> keep = c->h_inst;
    return rt_set_instances(c, keep.data(), (int)keep.size(), 1);
  }
  return RT_OK;
}

int rt_set_uniforms(rt_ctx* c, const rt_uniforms* u) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (!u) return fail(c, RT_ERR_INVALID_ARGUMENT, "uniforms pointer is NULL");
  memcpy(&c->uni, u, sizeof(UniformsDev));
  c->have_uni = true;
  return RT_OK;
}

int rt_set_skybox(rt_ctx* c, const uint8_t* const faces[6], int w, int h) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (!faces || w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad skybox arguments");
  for (int f = 0; f < 6; f++) if (!faces[f]) return fail(c, RT_ERR_INVALID_ARGUMENT, "skybox face is NULL");
  { int q = quiesce_scene(c); if (q) return q; }
  HIP_TRY(c, hipSetDevice(c->device));
  Scene* S = c->scene;
  const size_t face_bytes = (size_t)w * h * 4;
  if (S->d_sky) { HIP_TRY(c, hipFree(S->d_sky)); S->d_sky = nullptr; }
  HIP_TRY(c, hipMalloc((void**)&S->d_sky, 6 * face_bytes));
  for (int f = 0; f < 6; f++)
    HIP_TRY(c, hipMemcpy((uint8_t*)S->d_sky + f * face_bytes, faces[f], face_bytes, hipMemcpyHostToDevice));
  S->sky_w = w; S->sky_h = h;
  return RT_OK;
}

int rt_shard_rows(int height, int band_rows, int shard, int n_shards) {
  if (height <= 0 || band_rows <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards) return 0;
  const int n_bands = (height + band_rows - 1) / band_rows;
  int rows = 0;
  for (int b = shard; b < n_bands; b += n_shards) rows += std::min(band_rows, height - b * band_rows);
  return rows;
}

int rt_set_param(rt_ctx* c, const char* name, int value) {
  if (!c || !name) return RT_ERR_INVALID_ARGUMENT;
  std::string k(name);
  if (k == "trace_variant") {
    if (value < 0 || value > 2) return fail(c, RT_ERR_INVALID_ARGUMENT, "trace_variant must be 0, 1 or 2");
    if (value != 0 && !alt_kernels_built()) return fail(c, RT_ERR_INVALID_ARGUMENT, "trace_variant 1 / 2 are not in the product library (measured slower): build librt_mi355x_alt.so with `make alt` and load it (RT_LIB_VARIANT=alt)");
    { int q = quiesce_scene(c); if (q) return q; }
    Scene* S = c->scene;
    const int before = c->cfg.variant;
    for (rt_ctx* m : S->members) m->cfg.variant = value;   // the linked arrays are per scene: every slot walks the same ones
    bool relink = (value == 2 && before != 2 && S->blas_linked);   // the 4-ary records are derived from the linked BVH2 on demand
    if (value == 1) {
      // the quad kernel walks the BVH4 that only the host builder produces: rebuild device-built meshes on the host
      for (size_t mi = 0; mi < S->meshes.size(); mi++) {
        Mesh& m = S->meshes[mi];
        if (!m.built || !m.gpu_built) continue;
        build_blas(S->h_verts.data() + m.range.first_float, S->h_idx.data() + m.range.first_index, m.range.prim_count, m.bvh, m.tris);
        collapse_bvh4(m.bvh, true, false, m.bvh4);
        m.bounds = m.bvh.bounds; m.gpu_built = false; relink = true;
      }
    }
    if (relink) {
      S->blas_linked = false;
      // slots that had a TLAS get it back (same instances, fresh build) once the arrays are linked again
      for (rt_ctx* m : S->members) {
        if (!m->tlas_valid) continue;
        std::vector<rt_instance> keep = m->h_inst;
        m->tlas_valid = false;
        int r = rt_set_instances(m, keep.data(), (int)keep.size(), 0);
        if (r) { if (m != c) c->error = m->error; return r; }
      }
    }
    return RT_OK;
  }
  if (k == "closest_blocks_per_cu" || k == "shadow_blocks_per_cu") {
    if (value < -1 || value > 8) return fail(c, RT_ERR_INVALID_ARGUMENT, k + " must be -1 (automatic), 0 (the persistent grid) or 1..8");
    (k[0] == 'c' ? c->closest_blocks_per_cu : c->shadow_blocks_per_cu) = value; return RT_OK;
  }
  if (k == "trace_blocks_per_cu") {
    if (value < 1 || value > 8) return fail(c, RT_ERR_INVALID_ARGUMENT, "trace_blocks_per_cu must be 1..8");
    { int q = quiesce(c); if (q) return q; }   // the spill stacks are re-sized by the next frame (ensure_common)
    c->cfg.trace_blocks = c->n_cu * value; c->grid_user_set = true; return RT_OK;
  }
  if (k == "output_rgba8") {
    if (value != 0 && value != 1) return fail(c, RT_ERR_INVALID_ARGUMENT, "output_rgba8 must be 0 or 1");
    if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "output_rgba8 cannot change while a frame is pending");
    c->out_rgba8 = value != 0; c->out_bgra = false; return RT_OK;
  }
  if (k == "output_bgra8") {
    if (value != 0 && value != 1) return fail(c, RT_ERR_INVALID_ARGUMENT, "output_bgra8 must be 0 or 1");
    if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "output_bgra8 cannot change while a frame is pending");
    c->out_rgba8 = value != 0; c->out_bgra = value != 0; return RT_OK;
  }
  if (k == "primary_cover") { c->primary_cover = value != 0; return RT_OK; }
  if (k == "jitter_table") { c->jitter_table = value != 0; return RT_OK; }
  if (k == "tile_blobs") { c->tile_blobs = value != 0; return RT_OK; }
  if (k == "pixel_beams") { c->pixel_beams = value != 0; return RT_OK; }
  if (k == "camera_records") { c->camera_records = value != 0; return RT_OK; }
  if (k == "dead_shadow_rays") { c->dead_shadow_rays = value != 0; return RT_OK; }
  if (k == "shadow_beams") { c->shadow_beams = value != 0; return RT_OK; }

  if (k == "entry_points") { c->entry_points = value != 0; return RT_OK; }
  if (k == "packet_trace") {
    if (value < 0 || value > 2) return fail(c, RT_ERR_INVALID_ARGUMENT, "packet_trace must be 0, 1 or 2");
    if (value != 0 && !alt_kernels_built()) return fail(c, RT_ERR_INVALID_ARGUMENT, "packet_trace is not in the product library (measured slower): build librt_mi355x_alt.so with `make alt` and load it (RT_LIB_VARIANT=alt)");
    c->cfg.packet = value; return RT_OK;
  }
  if (k == "packet_blocks_per_cu") { if (value < 1 || value > 16) return fail(c, RT_ERR_INVALID_ARGUMENT, "packet_blocks_per_cu must be 1..16"); c->cfg.packet_blocks = c->n_cu * value; return RT_OK; }
  if (k == "shadow_entry") { if (value < 0 || value > 2) return fail(c, RT_ERR_INVALID_ARGUMENT, "shadow_entry must be 0, 1 or 2"); if (value != c->shadow_entry) { c->light_built = false; c->light_seen_valid = false; } c->shadow_entry = value; return RT_OK; }
  if (k == "entry_max_instances") { if (value < 1 || value >= (int)ENTRY_NO_INST) return fail(c, RT_ERR_INVALID_ARGUMENT, "entry_max_instances out of range"); c->entry_max_instances = value; return RT_OK; }
  if (k == "light_tiles") {
    if (value < 8 || value > 512) return fail(c, RT_ERR_INVALID_ARGUMENT, "light_tiles must be 8..512");
    { int q = quiesce(c); if (q) return q; }
    c->light_tiles = value; c->light_built = false; return RT_OK;
  }
  if (k == "tail_kernel") { if (value < 0 || value > 2) return fail(c, RT_ERR_INVALID_ARGUMENT, "tail_kernel must be 0 (off), 1 (auto) or 2 (always)"); c->tail_mode = value; return RT_OK; }
  if (k == "debug_force_tail_fault") { c->debug_force_tail_fault = value != 0; if (value == 2) c->tail_disabled = false; return RT_OK; }
  if (k == "blas_builder") { if (value != 0 && value != 1) return fail(c, RT_ERR_INVALID_ARGUMENT, "blas_builder must be 0 (host SAH) or 1 (device)"); for (rt_ctx* m : c->scene->members) m->blas_builder = value; return RT_OK; }
  if (k == "trace_rays_per_lane") { if (value < 1 || value > 64) return fail(c, RT_ERR_INVALID_ARGUMENT, "trace_rays_per_lane must be 1..64"); c->cfg.rays_per_lane = value; return RT_OK; }
  if (k == "trace_min_blocks") { if (value < 8) return fail(c, RT_ERR_INVALID_ARGUMENT, "trace_min_blocks must be >= 8"); c->cfg.min_blocks = value; return RT_OK; }
  if (k == "shade_blocks_per_cu") { if (value < 1 || value > 16) return fail(c, RT_ERR_INVALID_ARGUMENT, "shade_blocks_per_cu must be 1..16"); c->cfg.shade_blocks = c->n_cu * value; return RT_OK; }
  return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown parameter " + k);
}

// Host-only self check of the acceleration-structure builders (no device needed): builds the BVH2 of an indexed
// triangle mesh exactly as rt_build_blas(blas_builder 0) does, collapses it to the BVH4, quantizes it, and verifies
// the structural invariants.  out[0..7] = nodes, leaves, depth, max leaf size, bvh4 nodes, bvh4 stack need,
// violations found, triangles reached.  Returns 0 when every invariant holds.
int rt_debug_check_builders(const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx, uint64_t* out) {
  if (!verts6 || !idx || !out || n_idx % 3 != 0) return RT_ERR_INVALID_ARGUMENT;
  const uint32_t n = (uint32_t)(n_idx / 3);
  for (size_t k = 0; k < n_idx; k++) if ((size_t)idx[k] * 6 + 5 >= n_floats) return RT_ERR_INVALID_ARGUMENT;
  BuiltBvh bvh; std::vector<TriPacket> tris; Bvh4 b4; std::vector<BvhNodeQ> q; float q_lo[3], q_scale[3];
  build_blas(verts6, idx, n, bvh, tris);
  collapse_bvh4(bvh, true, false, b4);
  quantize_bvh2(bvh, q, q_lo, q_scale);
  uint64_t violations = 0, reached = 0, max_leaf = 0;
  std::vector<uint8_t> seen(n, 0);
  auto tri_box = [&](uint32_t leaf_pos, Aabb& b) {
    const TriPacket& t = tris[leaf_pos];
    for (int k = 0; k < 3; k++) {
      float a = t.v0[k], c1 = t.v0[k] + t.e1[k], c2 = t.v0[k] + t.e2[k];
      b.lo[k] = std::min(a, std::min(c1, c2)); b.hi[k] = std::max(a, std::max(c1, c2));
    }
  };
  // walk the float BVH2 and its quantized twin together
  struct Item { int32_t ref; Aabb box; int depth; };
  std::vector<Item> stack;
  auto child_box = [&](const BvhNode& nd, int which) { Aabb b; if (which == 0) { b.lo[0] = nd.a[0]; b.hi[0] = nd.a[1]; b.lo[1] = nd.a[2]; b.hi[1] = nd.a[3]; b.lo[2] = nd.c[0]; b.hi[2] = nd.c[1]; }
                                                        else { b.lo[0] = nd.b[0]; b.hi[0] = nd.b[1]; b.lo[1] = nd.b[2]; b.hi[1] = nd.b[3]; b.lo[2] = nd.c[2]; b.hi[2] = nd.c[3]; } return b; };
  Aabb all; for (int k = 0; k < 3; k++) { all.lo[k] = -3e38f; all.hi[k] = 3e38f; }
  stack.push_back({0, all, 0});
  int depth = 0;
  while (!stack.empty()) {
    Item it = stack.back(); stack.pop_back();
    depth = std::max(depth, it.depth);
    if (it.ref >= 0) {
      if ((size_t)it.ref >= bvh.nodes.size()) { violations++; continue; }
      const BvhNode& nd = bvh.nodes[it.ref]; const BvhNodeQ& nq = q[it.ref];
      for (int w = 0; w < 2; w++) {
        Aabb cb = child_box(nd, w);
        const int32_t ref = w ? nd.child1 : nd.child0;
        if (cb.lo[0] > 1e37f) continue;   // missing child of a synthetic root
        for (int k = 0; k < 3; k++) {
          if (cb.lo[k] < it.box.lo[k] - 1e-4f || cb.hi[k] > it.box.hi[k] + 1e-4f) violations++;      // child inside parent
          const uint32_t wq = nq.w[(w ? 3 : 0) + k];
          const double qlo = (double)q_lo[k] + (double)(wq & 0xFFFFu) * q_scale[k], qhi = (double)q_lo[k] + (double)(wq >> 16) * q_scale[k];
          if (qlo > cb.lo[k] || qhi < cb.hi[k]) violations++;                                         // quantized box contains the float box
        }
        if ((w ? nq.child1 : nq.child0) != ref) violations++;
        stack.push_back({ref, cb, it.depth + 1});
      }
    } else {
      const uint32_t r = (uint32_t)(~it.ref), first = r >> 3, count = (r & 7u) + 1u;
      max_leaf = std::max<uint64_t>(max_leaf, count);
      for (uint32_t k = 0; k < count; k++) {
        if (first + k >= n) { violations++; continue; }
        const uint32_t prim = tris[first + k].prim;
        if (prim >= n || seen[prim]) violations++; else { seen[prim] = 1; reached++; }
        Aabb tb; tri_box(first + k, tb);
        for (int a = 0; a < 3; a++) if (tb.lo[a] < it.box.lo[a] - 1e-4f || tb.hi[a] > it.box.hi[a] + 1e-4f) violations++;   // triangle inside its leaf box
      }
    }
  }
  if (reached != n) violations++;
  // BVH4: every triangle reachable exactly once as well
  std::vector<uint8_t> seen4(n, 0); uint64_t reached4 = 0;
  std::vector<int32_t> st4; if (!b4.nodes.empty()) st4.push_back(0);
  while (!st4.empty()) {
    int32_t ref = st4.back(); st4.pop_back();
    if (ref >= 0) { for (int k = 0; k < 4; k++) { int32_t c = b4.nodes[ref].c[k].ref; if (c != 0x7FFFFFFF) st4.push_back(c); } }
    else { const uint32_t r = (uint32_t)(~ref), first = r >> 3, count = (r & 7u) + 1u; for (uint32_t k = 0; k < count && first + k < n; k++) { uint32_t p = tris[first + k].prim; if (p < n && !seen4[p]) { seen4[p] = 1; reached4++; } else violations++; } }
  }
  if (reached4 != n) violations++;
  out[0] = bvh.nodes.size(); out[1] = bvh.leaves; out[2] = (uint64_t)depth; out[3] = max_leaf; out[4] = b4.nodes.size(); out[5] = (uint64_t)b4.stack_need;
  out[6] = violations; out[7] = reached;
  return violations ? RT_ERR_INVALID_ARGUMENT : RT_OK;
}

// Host-only view of the sizing rules (no device needed): out[0] = grid of k_tail for a device with n_cu compute units that
// can hold resident_per_cu of its workgroups each (0 = k_tail unusable), out[1] = int32 elements of the spill-stack
// allocation for that grid, a traversal grid of trace_blocks and ovf_stride entries per thread.
int rt_debug_sizing(int n_cu, int resident_per_cu, int trace_blocks, uint32_t ovf_stride, uint64_t* out) {
  if (!out || n_cu <= 0 || trace_blocks <= 0) return RT_ERR_INVALID_ARGUMENT;
  const int g = tail_grid(n_cu, resident_per_cu, 0);
  out[0] = (uint64_t)g; out[1] = (uint64_t)ovf_elems(trace_blocks, g, ovf_stride);
  return RT_OK;
}

int rt_set_timing(rt_ctx* c, int enabled) { if (!c) return RT_ERR_INVALID_ARGUMENT; c->timing = enabled == 2 ? 2 : (enabled != 0); return RT_OK; }

int rt_trace_shard(rt_ctx* c, int W, int H, int band_rows, int shard, int n_shards, void* d_out, size_t out_capacity_bytes, void* hip_stream) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (W <= 0 || H <= 0 || band_rows <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards || !d_out)
    return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_trace_shard arguments");
  if (!c->have_uni) return fail(c, RT_ERR_NOT_READY, "rt_set_uniforms has not been called");
  if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "a frame submitted with rt_trace_async is pending: call rt_trace_wait first");
  HIP_TRY(c, hipSetDevice(c->device));
  int r = ready_to_trace(c); if (r) return r;
  const int rows = rt_shard_rows(H, band_rows, shard, n_shards);
  if ((size_t)rows * W * (c->out_rgba8 ? 4 : 16) > out_capacity_bytes) return fail(c, RT_ERR_INVALID_ARGUMENT, "output buffer too small for this shard");
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  // one set of queues and counters per context: a frame on another stream must not start while the previous one runs
  if (c->frame_pending && c->frame_stream != s) { int q = collect_stats(c); if (q) return q; }
  return enqueue_frame(c, W, H, band_rows, shard, n_shards, (float4*)d_out, s);
}

// Root-side step of a multi-GPU frame: n_shards compact shards (as rt_trace_shard writes them, each padded to
// shard_stride_bytes) lie back to back in d_gathered after the gather; this writes the width x height frame to d_frame on
// hip_stream (NULL = the context's stream).  Pixel format = the context's ("output_rgba8").  Asynchronous.
int rt_trace_shard_batch(rt_ctx* c, int W, int H, int band_rows, int shard, int n_shards, void* d_out, size_t frame_stride_bytes, size_t out_capacity_bytes, void* hip_stream) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (W <= 0 || H <= 0 || band_rows <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards || !d_out)
    return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_trace_shard_batch arguments");
  if (!c->have_uni) return fail(c, RT_ERR_NOT_READY, "rt_set_batch has not been called");
  if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "a frame submitted with rt_trace_async is pending: call rt_trace_wait first");
  HIP_TRY(c, hipSetDevice(c->device));
  int r = ready_to_trace(c, true); if (r) return r;
  const int rows = rt_shard_rows(H, band_rows, shard, n_shards);
  const size_t px_bytes = c->out_rgba8 ? 4 : 16, shard_bytes = (size_t)rows * W * px_bytes;
  if (frame_stride_bytes == 0) frame_stride_bytes = shard_bytes;
  if (frame_stride_bytes < shard_bytes || frame_stride_bytes % px_bytes != 0 || frame_stride_bytes / px_bytes > 0xFFFFFFFFull)
    return fail(c, RT_ERR_INVALID_ARGUMENT, "frame_stride_bytes must be 0 (compact) or a whole number of pixels >= one shard");
  if (frame_stride_bytes * (size_t)(c->batch_k - 1) + shard_bytes > out_capacity_bytes) return fail(c, RT_ERR_INVALID_ARGUMENT, "output buffer too small for the shards of this batch");
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  if (c->frame_pending && c->frame_stream != s) { int q = collect_stats(c); if (q) return q; }
  c->batch_out_stride = (uint32_t)(frame_stride_bytes / px_bytes);
  r = enqueue_frame(c, W, H, band_rows, shard, n_shards, (float4*)d_out, s);
  c->batch_out_stride = 0;
  return r;
}

int rt_assemble_shards(rt_ctx* c, const void* d_gathered, int n_shards, size_t shard_stride_bytes, int W, int H, int band_rows,
                       void* d_frame, size_t frame_capacity_bytes, void* hip_stream) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  const size_t bpp = c->out_rgba8 ? 4 : 16;
  if (!d_gathered || !d_frame || n_shards <= 0 || W <= 0 || H <= 0 || band_rows <= 0 || shard_stride_bytes % bpp != 0)
    return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_assemble_shards arguments");
  if ((size_t)W * H * bpp > frame_capacity_bytes) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame buffer too small");
  for (int s = 0; s < n_shards; s++)
    if ((size_t)rt_shard_rows(H, band_rows, s, n_shards) * W * bpp > shard_stride_bytes) return fail(c, RT_ERR_INVALID_ARGUMENT, "shard stride smaller than a shard");
  HIP_TRY(c, hipSetDevice(c->device));
  launch_assemble(d_gathered, d_frame, W, H, band_rows, n_shards, shard_stride_bytes / bpp, c->out_rgba8, hip_stream ? (hipStream_t)hip_stream : c->stream);
  HIP_TRY(c, hipGetLastError());
  return RT_OK;
}

int rt_synchronize(rt_ctx* c) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->frame_pending) return collect_stats(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_get_stats(rt_ctx* c, rt_stats* st) {
  if (!c || !st) return RT_ERR_INVALID_ARGUMENT;
  int r = collect_stats(c); if (r) return r;
  *st = c->last;
  return RT_OK;
}

static int trace_host(rt_ctx* c, int W, int H, float* out, rt_stats* stats, bool counting) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (W <= 0 || H <= 0 || !out) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_trace arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  { int q = quiesce(c); if (q) return q; }
  const size_t px = (size_t)W * H;
  if (px > c->out_capacity) {
    if (c->d_out_own) HIP_TRY(c, hipFree(c->d_out_own));
    c->d_out_own = nullptr; c->out_capacity = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_out_own, px * sizeof(float4)));
    c->out_capacity = px;
  }
  c->counting = counting;
  int r = rt_trace_shard(c, W, H, H, 0, 1, c->d_out_own, px * sizeof(float4), nullptr);
  c->counting = false;
  if (r) return r;
  r = collect_stats(c); if (r) return r;
  HIP_TRY(c, hipMemcpy(out, c->d_out_own, px * (c->out_rgba8 ? 4 : sizeof(float4)), hipMemcpyDeviceToHost));
  if (stats) *stats = c->last;
  return RT_OK;
}

int rt_trace(rt_ctx* c, int W, int H, float* out, rt_stats* stats) { return trace_host(c, W, H, out, stats, false); }

// vkQueueSubmit + fence of the reference's frame loop (src/main.cpp:2905-2967): the frame and its copy to a pinned host
// buffer are enqueued on the context's stream and the call returns; a host keeps several contexts in flight.
int rt_trace_async(rt_ctx* c, int W, int H) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (W <= 0 || H <= 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad rt_trace_async arguments");
  if (c->async_pending) return fail(c, RT_ERR_NOT_READY, "rt_trace_async: the previous frame of this context has not been collected (rt_trace_wait)");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t px = (size_t)W * H;
  if (px > c->out_capacity) {
    if (c->d_out_own) HIP_TRY(c, hipFree(c->d_out_own));
    c->d_out_own = nullptr; c->out_capacity = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_out_own, px * sizeof(float4)));
    c->out_capacity = px;
  }
  if (px > c->pinned_capacity) {
    if (c->h_out_pinned) HIP_TRY(c, hipHostFree(c->h_out_pinned));
    c->h_out_pinned = nullptr; c->pinned_capacity = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_out_pinned, px * sizeof(float4), hipHostMallocDefault));
    c->pinned_capacity = px;
  }
  if (c->frame_pending) { int q = collect_stats(c); if (q) return q; }   // d_out_own may still be written by an rt_trace_shard frame
  int r = rt_trace_shard(c, W, H, H, 0, 1, c->d_out_own, px * sizeof(float4), nullptr);
  if (r) return r;
  HIP_TRY(c, hipMemcpyAsync(c->h_out_pinned, c->d_out_own, px * (c->out_rgba8 ? 4 : sizeof(float4)), hipMemcpyDeviceToHost, c->stream));
  c->async_pending = true; c->async_w = W; c->async_h = H;
  return RT_OK;
}

// vkWaitForFences (src/main.cpp:772-778) for the frame submitted with rt_trace_async; *pixels stays valid until the
// next rt_trace_async on this context.
int rt_trace_wait(rt_ctx* c, const void** pixels, rt_stats* stats) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if (!c->async_pending) return fail(c, RT_ERR_NOT_READY, "rt_trace_wait without rt_trace_async");
  HIP_TRY(c, hipSetDevice(c->device));
  c->async_pending = false;
  c->frame_rerendered = false;
  int r = collect_stats(c); if (r) return r;          // waits for the frame's kernels
  if (c->frame_rerendered)   // the copy enqueued by rt_trace_async took the discarded frame
    HIP_TRY(c, hipMemcpyAsync(c->h_out_pinned, c->d_out_own, (size_t)c->async_w * c->async_h * (c->out_rgba8 ? 4 : sizeof(float4)), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));        // ... and for the copy behind them
  if (pixels) *pixels = c->h_out_pinned;
  if (stats) *stats = c->last;
  return RT_OK;
}
int rt_trace_counting(rt_ctx* c, int W, int H, float* out, rt_stats* stats) { return trace_host(c, W, H, out, stats, true); }

int rt_intersect(rt_ctx* c, size_t n, const float* rays8, int any_hit, rt_hit* out, int counting, rt_stats* stats) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  if ((!rays8 || !out) && n) return fail(c, RT_ERR_INVALID_ARGUMENT, "null ray/hit pointers");
  if (n >= 0xFFFFFF00ull) return fail(c, RT_ERR_INVALID_ARGUMENT, "too many rays for one call");
  HIP_TRY(c, hipSetDevice(c->device));
  { int q = quiesce(c); if (q) return q; }   // the counters and spill stacks below are the pending frame's
  int r = ready_to_trace(c); if (r) return r;
  r = ensure_common(c); if (r) return r;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return RT_OK;
  std::vector<float4> ho(n), hd(n);
  for (size_t i = 0; i < n; i++) {
    const float* p = rays8 + 8 * i;
    ho[i] = make_float4(p[0], p[1], p[2], p[3]);
    hd[i] = make_float4(p[4], p[5], p[6], p[7]);
  }
  float4 *d_o = nullptr, *d_d = nullptr; HitRec* d_h = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  struct Guard {   // every exit path below releases the temporaries
    float4 *&o, *&d; HitRec*& h; hipEvent_t &a, &b;
    ~Guard() { if (o) hipFree(o); if (d) hipFree(d); if (h) hipFree(h); if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
  } guard{d_o, d_d, d_h, e0, e1};
  HIP_TRY(c, hipMalloc((void**)&d_o, n * sizeof(float4)));
  HIP_TRY(c, hipMalloc((void**)&d_d, n * sizeof(float4)));
  HIP_TRY(c, hipMalloc((void**)&d_h, n * sizeof(HitRec)));
  HIP_TRY(c, hipMemcpy(d_o, ho.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(d_d, hd.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, CNT_WORDS * sizeof(uint32_t), c->stream));
  uint32_t n32 = (uint32_t)n;
  HIP_TRY(c, hipMemcpyAsync(c->d_counters + cnt_tail(0, 0), &n32, sizeof(n32), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
  hipEventRecord(e0, c->stream);
  LaunchCfg raw_cfg = c->cfg;
  if (c->stack_need > 120) raw_cfg.packet = 0;   // (see enqueue_frame)
  launch_trace_raw(scene_dev(c), d_o, d_d, d_h, n32, c->d_ovf, c->d_counters, any_hit != 0, counting != 0, raw_cfg, c->stream);
  hipEventRecord(e1, c->stream);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpy(out, d_h, n * sizeof(HitRec), hipMemcpyDeviceToHost));
  struct Rezero {   // frames expect both counter blocks zeroed, on whatever stream they are enqueued next
    rt_ctx* c;
    ~Rezero() { (void)hipMemsetAsync(c->d_counters, 0, 2 * CNT_WORDS * sizeof(uint32_t), c->stream); (void)hipStreamSynchronize(c->stream); c->cnt_parity = 0; }
  } rezero{c};
  if (stats) {
    uint32_t cnt[CNT_TAILS];
    HIP_TRY(c, hipMemcpy(cnt, c->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
    memcpy(&stats->node_visits, &cnt[any_hit ? CNT_NODE_VISITS_SH : CNT_NODE_VISITS], 8);
    memcpy(&stats->tri_tests, &cnt[any_hit ? CNT_TRI_TESTS_SH : CNT_TRI_TESTS], 8);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (any_hit) stats->ms_trace_shadow = ms; else stats->ms_trace_closest = ms;
    stats->closest_rays = any_hit ? 0 : n; stats->rays_shadow = any_hit ? n : 0;
    stats->bvh_node_bytes = c->cfg.variant == 1 ? sizeof(Bvh4Node) : c->cfg.variant == 2 ? sizeof(WideNodeQ) : sizeof(BvhNodeQ); stats->bvh_tri_bytes = sizeof(TriPacket);
  }
  return RT_OK;
}

}  // extern "C"
