/* rt_api.h — C ABI of librt_mi355x.so, the MI355X (gfx950) replacement for the reference's
 * VK_KHR_ray_tracing_pipeline stage.
 *
 * The reference (mcan1999/vulkan-raytracing) reaches its ray-tracing stage through raw Vulkan calls
 * inlined in main(); it has no plugin/FFI interface.  This header cuts the seam at the Vulkan
 * objects the host creates for that stage: one export per Vulkan interaction on the path.  Each
 * declaration cites the reference call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every call returns 0 on success (mirrors VK_SUCCESS); non-zero = rt_status below, message via
 *     rt_last_error().  No exception crosses the boundary (the reference throws
 *     std::runtime_error("Vulkan API exception...") from throwExceptionVulkanAPI, src/main.cpp:138-147;
 *     the C++ host wrapper host/rt_host.hpp re-throws to keep that behaviour).
 *   - the caller owns every host array; the library copies on upload (as copyData does,
 *     src/main.cpp:203-219).  Handles are opaque; destroy is explicit.
 *   - one context drives one GPU and is not re-entrant (the reference is single-threaded with one
 *     queue and a blocking fence after every build/upload); the contexts of one scene family (rt_create_frame_slot)
 *     are driven from one host thread as well.
 *   - plain pointers and sizes only; no torch / HIP types in signatures (streams are void*).
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = 1,
  RT_ERR_NOT_READY = 2,      /* call order violated (e.g. trace before geometry/instances/uniforms) */
  RT_ERR_DEVICE = 3,         /* a HIP runtime call failed; see rt_last_error */
  RT_ERR_OUT_OF_MEMORY = 4,
  RT_ERR_NO_DEVICE = 5       /* no usable gfx950 device: the library never falls back to the CPU */
};

/* One object of the shared vertex/index buffers (generalises orbitingObjectVertexOffset /
 * orbitingObjectPrimitiveOffset, src/main.cpp:1872-1873). */
typedef struct rt_mesh_range {
  uint64_t first_float;   /* offset into verts6, in floats (= vertexOffset of src/shader.rchit:55) */
  uint64_t first_index;   /* offset into idx, in uint32s (= 3*primitiveOffset) */
  uint32_t prim_count;    /* triangles (src/main.cpp:1644) */
  uint32_t reserved;
} rt_mesh_range;

/* 64-byte mirror of VkAccelerationStructureInstanceKHR as filled by createInstance
 * (src/main.cpp:538-551): a maintainer can memcpy the Vulkan struct and overwrite the last field. */
typedef struct rt_instance {
  float transform[12];              /* row-major 3x4 object->world (glmToVulkan, src/main.cpp:245-249) */
  uint32_t custom_index_and_mask;   /* instanceCustomIndex:24 (low) | mask:8 (high) */
  uint32_t sbt_offset_and_flags;    /* instanceShaderBindingTableRecordOffset:24 | flags:8 (ignored: the
                                       reference always uses offset 0 / TRIANGLE_FACING_CULL_DISABLE) */
  uint64_t mesh;                    /* index into rt_mesh_range[] — replaces accelerationStructureReference.
                                       RULE: the closest-hit stage reads the index/vertex range of THIS mesh.  The reference
                                       picks the range from instanceCustomIndex instead (src/shader.rchit:52-58: index 0 -> no
                                       offset, otherwise the orbiting object's offsets), which is the same thing whenever
                                       customIndex k is given to instances of mesh k, as src/main.cpp:1810-1816 does; an
                                       instance of mesh 1 created with customIndex 0 would read the wrong triangles there and
                                       the right ones here.  customIndex still selects the material type (src/shader.rgen:96) */
} rt_instance;

/* The 104-byte UniformStructure, field for field (src/main.cpp:1847-1866; src/shader.rgen:22-46). */
typedef struct rt_uniforms {
  float position[4];
  float right[4];
  float up[4];
  float forward[4];
  float light_position[3];
  float light_intensity;
  uint32_t max_bounce_count;
  uint32_t samples_per_pixel;
  uint32_t center_object_type;      /* 0 diffuse, 1 mirror, 2 refractive (include/config.h:9-16) */
  uint32_t orbiting_object_type;
  uint32_t orbiting_object_primitive_offset;  /* informational: ranges[] is authoritative */
  uint32_t orbiting_object_vertex_offset;
} rt_uniforms;

/* Result of one traceRayEXT (src/shader.rgen:86-87 / 111-112): what the driver hands to rchit/rmiss. */
typedef struct rt_hit {
  float t, u, v;        /* gl_HitTEXT and hitAttributeEXT vec2 (barycentrics of vertices B, C) */
  int32_t prim;         /* gl_PrimitiveID; -1 on miss */
  int32_t inst;         /* index into the instance array (gl_InstanceID); -1 on miss */
} rt_hit;

typedef struct rt_stats {
  uint64_t rays_primary;     /* one count per traceRayEXT-equivalent, by class */
  uint64_t rays_secondary;
  uint64_t rays_shadow;
  uint64_t node_visits;      /* closest-hit kernel; filled only by the instrumented (counting) kernels */
  uint64_t tri_tests;
  uint64_t node_visits_shadow; /* any-hit kernel, same */
  uint64_t tri_tests_shadow;
  uint64_t diag[6];          /* counting build: loop trips, busy quad-trips, wave cycles (closest; shadow) */
  uint64_t closest_rays;     /* rays through the closest-hit traversal kernel (primary+secondary) */
  float ms_frame;            /* HIP-event time of the whole frame pipeline on the trace stream */
  float ms_raygen;
  float ms_trace_closest;    /* sum over its launches of the closest-hit traversal kernel k_trace (not k_tail) */
  float ms_trace_shadow;     /* any-hit traversal kernel */
  float ms_shade;
  float ms_resolve;
  uint32_t launches_trace_closest;   /* per frame */
  uint32_t launches_total;
  uint32_t timed_frames;     /* the ms_* fields are means over this many frames (all frames enqueued with timing on
                                since the previous rt_get_stats / rt_trace) */
  float ms_tail;             /* k_tail: bounces 1..maxBounceCount in one launch (0 when the per-bounce launches ran) */
  uint32_t bvh_node_bytes;   /* S_node, S_tri of the roofline formula */
  uint32_t bvh_tri_bytes;
  uint32_t tail_faults;      /* frames of this context that were rendered again with per-bounce launches because a k_tail grid
                                barrier gave up (its workgroups were not co-resident); the context stays off k_tail afterwards */
  uint32_t frames_rerendered; /* 1 when the frame these statistics belong to had to be rendered a second time (k_tail fault): a copy of
                                the frame taken BEFORE this call returned (a gather or memcpy enqueued behind rt_trace_shard) is stale */
  /* ABI 6 — tile blobs (rt_set_param "tile_blobs"): the nodes and triangle packets a screen tile's rays can touch, staged through LDS */
  uint64_t blob_tiles;          /* tiles of this frame whose primary rays were walked in LDS */
  uint64_t blob_tiles_large;    /* ... of which needed the large size class (counting builds fill the blob_* fields) */
  uint64_t blob_tiles_refused;  /* tiles whose blob fit no size class (nodes, packets, depth, arena): their rays took the global walk */
  uint64_t blob_nodes;          /* nodes / triangle packets in all blobs of the frame */
  uint64_t blob_tris;
  uint64_t tile_rays;           /* primary rays walked in LDS (part of closest_rays) */
  uint64_t tile_rays_handed_on; /* ... of which went on to the global walk because they could still hit another instance */
  uint64_t tile_diag[6];        /* counting builds: wave cycles of k_tile before the walk (blob copy, ray generation), in the walk, in all; then from the
                                   start of the workgroup until: its list entry is there, the blob is in LDS, the ray is set up */
  uint64_t rays_shadow_untraced; /* ABI 7 — of rays_shadow: shadow rays whose outcome cannot change their sample (the surface and the half vector face away from
                                    the light: diffuse and specular are exactly 0, src/shader.rgen:113-128) — settled in k_shade, not walked (rt_set_param "dead_shadow_rays") */
} rt_stats;

/* Device/queue/pipeline creation (src/main.cpp:928-1102, 1578-1601).  device_id = HIP ordinal. */
int rt_create(rt_ctx** out_ctx, int device_id);
/* A second (third, ...) frame in flight on the same GPU: a context that SHARES the parent's scene — geometry, BLAS, cube map,
 * everything rt_upload_geometry / rt_build_blas / rt_set_skybox created, before or after this call — and owns only what one
 * frame needs: its instance records and TLAS, its uniform block, its ray queues, counters and stream.  This is the
 * reference's per-swapchain-image state (command buffer, fence, semaphores: src/main.cpp:2597, 2740-2749) next to its
 * shared buffers and acceleration structures.  Scene-building calls on ANY context of the family wait for the frames of
 * all of them, rebuild the shared scene and invalidate every slot's TLAS (call rt_set_instances again).  At most 16
 * contexts share a scene; rt_destroy on the last one frees it. */
int rt_create_frame_slot(rt_ctx* parent, rt_ctx** out_ctx);
/* Cleanup (src/main.cpp:2977-3060). */
void rt_destroy(rt_ctx* ctx);

/* buildBuffer for the shared vertex and index buffers (src/main.cpp:1684-1697, 1713-1726).
 * verts6 = interleaved [px py pz nx ny nz] (src/main.cpp:1673-1682), idx = object-local uint32. */
int rt_upload_geometry(rt_ctx* ctx, const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx,
                       const rt_mesh_range* ranges, int n_meshes);

/* createBLASGeometry + createBLAS + createBLASScratchBuffer + buildBLAS (src/main.cpp:305-536, called
 * :1734-1799).  Synchronous like the reference's fence wait (:525-527). */
int rt_build_blas(rt_ctx* ctx, int mesh);

/* createInstance + createTLAS (src/main.cpp:538-793; called :1818-1835 with update=false and every
 * frame :2848-2861 with update=true).  update!=0 keeps the TLAS topology and refits boxes (Vulkan
 * UPDATE mode, src=dst); it requires the same instance count as the last build.  Never waits for the frame in flight: the
 * instance records and TLAS nodes are double-buffered, the new set travels on the context's stream (pinned staging) and
 * the next frame is ordered behind it; only the frame before the last is waited for, if it is still running (the reference
 * allocates buffers and blocks on a fence here every frame, src/main.cpp:672-696, 752-778). */
int rt_set_instances(rt_ctx* ctx, const rt_instance* instances, int n, int update);

/* ---- MTL materials and a per-instance type table -------------------------------------------------
 * The reference's loader parses Kd/Ks/Ns/Ni/illum (include/tiny_obj_loader.h:565 GetMaterials) and its renderer ignores
 * them: src/shader.rgen:51-55 hard-codes ka (.1,.3,.1), kd (.2,1,.2), ks .8, exponent 100, index of refraction 1.52, and
 * src/shader.rgen:96 knows two object types ("Hardcoded as 2 objects", src/main.cpp:2425).  Both calls are optional:
 * without them every frame is the reference's, bit for bit. */
typedef struct rt_material {
  float ka[3]; float ns;        /* Ka;  Ns = specular exponent, applied as an integer power (rounded, 0..1023) */
  float kd[3]; float ni;        /* Kd;  Ni = index of refraction of a refractive surface */
  float ks[3]; uint32_t type;   /* Ks;  0 diffuse, 1 mirror, 2 refractive, or RT_MATERIAL_TYPE_OF_INSTANCE */
} rt_material;
#define RT_MATERIAL_TYPE_OF_INSTANCE 0xFFFFFFFFu   /* the surface's type is its instance's (rt_set_instance_types / the uniform block) */
/* table[prim_material[g]] shades triangle g of the shared index buffer, g = first_index / 3 + gl_PrimitiveID (one entry per
 * triangle: n_prims = n_idx / 3).  Scene state, shared by all frame slots; n_materials == 0 removes the table.  With a table the
 * diffuse branch evaluates Iamb*ka, kd, ks, pow(., Ns) and the refractive branch Ni from the hit triangle's material. */
int rt_set_materials(rt_ctx* ctx, const rt_material* table, int n_materials, const uint32_t* prim_material, size_t n_prims);
/* types[i] in {0,1,2} for instance i of this context's next rt_set_instances (and of the current ones, re-issued at once);
 * n == 0 restores `objectIndex == 0 ? centerObjectType : orbitingObjectType` (src/shader.rgen:96). */
int rt_set_instance_types(rt_ctx* ctx, const uint32_t* types, int n);

/* Uniform buffer copyData (src/main.cpp:1887-1889, 2901-2903). */
int rt_set_uniforms(rt_ctx* ctx, const rt_uniforms* u);

/* Cube map creation + upload (src/main.cpp:2073-2412): 6 RGBA8 faces in the order
 * right,left,top,bottom,front,back (:2064-2071) = +X,-X,+Y,-Y,+Z,-Z, all w x h. */
int rt_set_skybox(rt_ctx* ctx, const uint8_t* const faces_rgba8[6], int w, int h);

/* vkCmdTraceRaysKHR(W,H,1) + the image copy (src/main.cpp:2620-2624, 2683-2686).  Blocking; writes the
 * whole frame (row 0 = top, RGBA32F, the shader's declared rgba32f format src/shader.rgen:48) to host. */
int rt_trace(rt_ctx* ctx, int width, int height, float* out_rgba32f_host, rt_stats* stats);

/* Sharded, asynchronous form used for multi-GPU tiling (one process per GPU).  d_out must stay valid, and is only guaranteed
 * complete, after the rt_synchronize / rt_get_stats that follows the call: should a grid barrier of the bounce kernel give
 * up (rt_stats.tail_faults), that call renders the frame again into d_out — from the uniforms and instance records the
 * frame was submitted with — and reports rt_stats.frames_rerendered = 1; work the caller enqueued behind the frame on its
 * own stream (a gather, a copy) has then consumed the incomplete frame and must be repeated.  Frames enqueued back to back
 * on one context without collecting each: a fault in an earlier one is still reported and switches the context to
 * per-bounce launches, but only the most recent frame is rendered again.  Renders the row bands
 * {b : b % n_shards == shard} of band_rows rows each and writes them COMPACTLY (band after band, each
 * band_rows x width x 4 floats; the last band of the frame may be short) into d_out, a DEVICE pointer
 * owned by the caller (e.g. a torch tensor), enqueued on hip_stream (NULL = the context's stream).
 * Returns immediately; use rt_synchronize / the stream to wait.  out_capacity_bytes guards d_out. */
int rt_trace_shard(rt_ctx* ctx, int width, int height, int band_rows, int shard, int n_shards,
                   void* d_out, size_t out_capacity_bytes, void* hip_stream);
/* Frame batches: K <= 8 CONSECUTIVE frames through ONE pass of the pipeline.  A rank of an N-GPU split renders only its bands, and a
 * 1/8 shard of one frame is eight kernel launches at their latency floors (measured: 0.084 ms per shard frame against 0.058 = 1/8 of a
 * whole frame) — the bands of K frames together are launches of whole-frame size again, and the N-GPU host makes one gather per batch.
 * This is the multi-GPU form of the reference's frames in flight (swapchain images, src/main.cpp:1203, 2905-2967): the frames of a batch
 * are rendered together and complete together.
 * rt_set_batch replaces rt_set_instances + rt_set_uniforms for the K frames: instances = n_frames x n records (frame k's at
 * instances + k * n, each frame a createTLAS(update) of the same topology, src/main.cpp:2848-2861), uniforms = n_frames blocks — camera
 * and light may differ from frame to frame, maxBounceCount / samplesPerPixel / object types are the batch's.  update as in rt_set_instances.
 * rt_trace_shard_batch is rt_trace_shard for the batch: frame k's compact shard lands frame_stride_bytes behind frame k - 1's (0: back to
 * back, rows * width pixels apart); statistics are sums over the batch.  Results are those of the K frames rendered one by one, bit for bit (tested). */
int rt_set_batch(rt_ctx* ctx, int n_frames, const rt_instance* instances, int n, const rt_uniforms* uniforms, int update);
int rt_trace_shard_batch(rt_ctx* ctx, int width, int height, int band_rows, int shard, int n_shards,
                         void* d_out, size_t frame_stride_bytes, size_t out_capacity_bytes, void* hip_stream);
/* Frames in flight from a plain C/C++ host: rt_trace_async enqueues the frame and its copy to a pinned host buffer owned
 * by the context and returns (vkQueueSubmit with a fence, src/main.cpp:2905-2967); rt_trace_wait blocks until that frame
 * is complete (vkWaitForFences, src/main.cpp:772-778) and hands out the pixels (W*H*4 floats, or W*H*4 bytes with
 * rt_set_param "output_rgba8" 1; valid until the next rt_trace_async on this context) and the frame's counters.  One frame per context may be pending; a host that wants P
 * frames in flight keeps P contexts, as the reference keeps one command buffer, fence and image per swapchain image. */
int rt_trace_async(rt_ctx* ctx, int width, int height);
int rt_trace_wait(rt_ctx* ctx, const void** pixels, rt_stats* stats);

/* Root-side step of a multi-GPU frame (the analogue of the reference's vkCmdCopyImage into the presented image,
 * src/main.cpp:2683-2686): after the gather, n_shards compact shards (as rt_trace_shard writes them, shard s at byte
 * s * shard_stride_bytes) lie in d_gathered; this de-interleaves them into the width x height frame at d_frame (device
 * pointers of ctx's GPU; pixel format = ctx's, RGBA32F or RGBA8), enqueued on hip_stream (NULL = the context's). */
int rt_assemble_shards(rt_ctx* ctx, const void* d_gathered, int n_shards, size_t shard_stride_bytes, int width, int height,
                       int band_rows, void* d_frame, size_t frame_capacity_bytes, void* hip_stream);

/* number of rows rt_trace_shard writes for (height, band_rows, shard, n_shards) */
int rt_shard_rows(int height, int band_rows, int shard, int n_shards);

/* Wait for the last enqueued frame and read its counters / HIP-event timings. */
int rt_synchronize(rt_ctx* ctx);
int rt_get_stats(rt_ctx* ctx, rt_stats* stats);
/* Per-kernel hipEvent timing (default off): 1 = events around every kernel of a frame (each record between two kernels costs
 * ~10 us of idle GPU), 2 = around the closest-hit traversal launches only (rt_stats.ms_trace_closest; the other times read 0). */
int rt_set_timing(rt_ctx* ctx, int enabled);

/* Tunables (no reference counterpart): "trace_variant" 0 = quantized BVH2 / one lane per ray (default), 1 = BVH4 /
 * four lanes per ray, 2 = 4-ary records / one lane per ray; "tail_kernel" 0 = one launch per bounce and kernel, 1 = bounces
 * 1..maxBounceCount in one launch when the previous frame had few secondary rays (default), 2 = always;
 * "output_rgba8" 1 = every entry point that returns a frame stores 8-bit RGBA (clamp to [0,1], x255, round; the format
 * the reference's storage image really has, src/main.cpp:1899) instead of RGBA32F — a quarter of the PCIe bytes;
 * "trace_blocks_per_cu" 1..8; "shade_blocks_per_cu" 1..16; "blas_builder" 1 = device builder (default:
 * rt_build_blas builds on the GPU, as the reference's DEVICE build type does, src/main.cpp:345-357 — a binned-SAH tree made level by
 * level; the environment variable RT_GPU_BVH_ALGO = 1 / 2 selects the LBVH / PLOC builders instead), 0 = host binned-SAH (threaded);
 * "trace_rays_per_lane", "trace_min_blocks" size the persistent grids; "closest_blocks_per_cu" / "shadow_blocks_per_cu" cap one launch's share of
 * it (-1 = automatic, the default: 2 per CU for launches of at most 2.5 M rays when three or more frame slots share the GPU, 0 = no cap, 1..8); "primary_cover" 1 (default) = before ray generation the
 * frontier boxes of every instance's BLAS are projected onto 8x8-pixel screen tiles and the samples of tiles no mesh can
 * project onto are shaded as misses without any box test, 0 = every primary ray is tested against the TLAS;
 * "entry_points" 1 (default; needs primary_cover) = every marked tile gets an entry record — the handful of deep subtrees of the
 * TLAS / the nearest instance's BLAS that the tile's beam of primary rays can touch — and its primary rays start their walk there
 * instead of at the TLAS root; "shadow_entry" = the same for the shadow rays, from the tiles of a cube of "light_tiles" (8..512,
 * default 256) tiles per side around the light: 0 = off, 1 = rebuilt in every frame (measured: costs more than it saves), 2 (default) =
 * built once the light and the instances have stood still for two frames and kept until either moves (they do not depend on the camera); "packet_trace" 1 = primary and shadow rays are
 * walked by the packet kernel (one wavefront per 64-ray chunk; default 0: measured slower), 2 = rt_intersect's rays too;
 * "output_bgra8" 1 = like "output_rgba8" in the byte order of a B8G8R8A8 surface (surfaceFormatList[0] is normally that,
 * src/main.cpp:1204, 1899), so that a frame can be compared byte for byte with a screenshot of the original's swapchain image;
 * round 4: "pixel_beams" 1 (default) = the primary rays of a pixel share one walk (k_beam: boxes against the beam of the pixel's samples, triangles
 * per ray; frames without far rays), 0 = one walk per ray; "camera_records" 0 = no entry records for the primary rays (the light-side records stay);
 * "dead_shadow_rays" 1 (default) = a shadow ray whose outcome cannot change its sample (diffuse and specular exactly 0: the same bits lit or
 * shadowed) is settled when it is made and not walked — it still counts in rt_stats::rays_shadow, rays_shadow_untraced says how many — 0 = every
 * shadow ray is walked; "shadow_beams" 1 = the shadow rays of the primary hits in beams as well (k_beam_shadow; default 0: measured slower);
 * "jitter_table" 1 (default) = k_raygen reads the sample positions of a frame size from a table made once, 0 = evaluates the hash per sample and frame;
 * "tile_blobs" 1 = the nodes and triangle packets of a screen tile staged through LDS (k_blob / k_tile; default 0: measured slower).
 * Results do not depend on any of them. */
int rt_set_param(rt_ctx* ctx, const char* name, int value);

/* Record-level entry for traceRayEXT alone (rows a10/a14): n rays of 8 floats (o.xyz, tmin, d.xyz, tmax)
 * from host memory; any_hit!=0 = TerminateOnFirstHit|SkipClosestHit (src/shader.rgen:67). Blocking.
 * counting!=0 runs the instrumented kernel and fills stats->node_visits / tri_tests. */
int rt_intersect(rt_ctx* ctx, size_t n, const float* rays8_host, int any_hit, rt_hit* out_host, int counting,
                 rt_stats* stats);

/* Same frame as rt_trace but through the instrumented traversal kernels (visit counters). */
int rt_trace_counting(rt_ctx* ctx, int width, int height, float* out_rgba32f_host, rt_stats* stats);

/* Host-only self check of the acceleration-structure builders (needs no GPU): builds the BVH2 / BVH4 / quantized nodes of
 * an indexed mesh as rt_build_blas(blas_builder 0) does and verifies their invariants (every triangle in exactly one leaf,
 * children inside parents, quantized boxes containing float boxes, depth and stack bounds).  out[8] = nodes, leaves,
 * depth, max leaf size, BVH4 nodes, BVH4 stack need, violations, triangles reached.  0 = all invariants hold. */
int rt_debug_check_builders(const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx, uint64_t* out8);
/* Host-only view of the launch/allocation sizing rules (needs no GPU): out2[0] = workgroups of the k_tail grid on a device
 * of n_cu compute units holding resident_per_cu of them each (0 = k_tail is not used), out2[1] = int32 elements of the
 * spill-stack allocation for that grid, a traversal grid of trace_blocks workgroups and ovf_stride entries per thread. */
int rt_debug_sizing(int n_cu, int resident_per_cu, int trace_blocks, uint32_t ovf_stride, uint64_t* out2);

/* Message of the last failing call on this context (or of rt_create when ctx==NULL). */
const char* rt_last_error(const rt_ctx* ctx);
/* "gfx950 <device name> CUs=<n>" of the bound device. */
const char* rt_device_info(const rt_ctx* ctx);
/* ABI version: bumped on any signature/layout change. */
int rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
