/* rt_multi.h — C ABI of librt_multi.so: the ray-tracing stage on SEVERAL GPUs of one node, driven by ONE host process.
 *
 * The reference binds a single device (physicalDeviceHandleList[0], src/main.cpp:928; deviceMask 0, :1108) and copies the
 * traced image into the presented one (src/main.cpp:2683-2686).  Here the frame is split into interleaved 8-row bands
 * (band b -> device b mod N), every device holds its own copy of the scene (one rt_ctx scene + frame slots per GPU,
 * include/rt_api.h), renders its bands (rt_trace_shard) and the compact shards are GATHERED on the first device over
 * xGMI with RCCL called directly from C++ (ncclCommInitAll + one ncclGather per device inside ncclGroupStart/End, the
 * single-process form of the collective), then de-interleaved there (rt_assemble_shards).  Pixels are independent, the
 * builders are deterministic, so the N-GPU frame equals the 1-GPU frame bit for bit.
 *
 * Conventions are those of rt_api.h: 0 on success, message through rtm_last_error, caller owns host arrays, one host
 * thread drives a context.  `slot` selects one of the frames in flight (0 .. frames_in_flight-1).
 */
#ifndef RT_MULTI_H
#define RT_MULTI_H

#include <stddef.h>
#include <stdint.h>

#include "rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtm_ctx rtm_ctx;

enum rtm_flags {
  RTM_LOOPBACK = 1   /* no RCCL: shards move with device-to-device copies.  Allows several LOGICAL devices on one physical
                        GPU (device_ids may repeat) — how the N > 1 path is tested where only one GPU exists */
};

/* Device/queue creation (src/main.cpp:928-1102) on n_devices GPUs; device_ids[0] is the root that assembles the frame.
 * frames_in_flight frame slots per device (the reference: swapchain image count, src/main.cpp:1203). */
int rtm_create(rtm_ctx** out_ctx, int n_devices, const int* device_ids, int frames_in_flight, int flags);
void rtm_destroy(rtm_ctx* ctx);

/* Scene building, replicated on every device (src/main.cpp:1684-1726, 1734-1799, 2073-2412). */
int rtm_upload_geometry(rtm_ctx* ctx, const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx,
                        const rt_mesh_range* ranges, int n_meshes);
int rtm_build_blas(rtm_ctx* ctx, int mesh);
int rtm_set_skybox(rtm_ctx* ctx, const uint8_t* const faces_rgba8[6], int w, int h);
/* rt_set_param on every slot of every device; plus this library's own "host_copy" (default 1; 0: rtm_trace_wait hands out no
 * host pixels, the assembled frame stays on the root device: rtm_frame_device). */
int rtm_set_param(rtm_ctx* ctx, const char* name, int value);
/* SURVEY.md §8(f) row n4 on several GPUs: the MTL material table (scene state, replicated) and the per-instance types of one
 * slot — rt_set_materials / rt_set_instance_types on every device (src/shader.rgen:51-55, :96). */
int rtm_set_materials(rtm_ctx* ctx, const rt_material* table, int n_materials, const uint32_t* prim_material, size_t n_prims);
int rtm_set_instance_types(rtm_ctx* ctx, int slot, const uint32_t* types, int n);
/* rt_set_timing on the ROOT device's slots: rtm_trace_wait's stats carry that device's kernel times. */
int rtm_set_timing(rtm_ctx* ctx, int enabled);

/* Per-frame state of one slot, pushed to every device (src/main.cpp:2848-2861, 2901-2903). */
int rtm_set_instances(rtm_ctx* ctx, int slot, const rt_instance* instances, int n, int update);
int rtm_set_uniforms(rtm_ctx* ctx, int slot, const rt_uniforms* u);

/* vkQueueSubmit of one frame (src/main.cpp:2933-2949): every device enqueues its bands, the gather and, on the root, the
 * de-interleave and the copy to a pinned host buffer; returns at once. */
/* Frame batches (include/rt_api.h rt_set_batch): the slot's next rtm_trace_async renders n_frames CONSECUTIVE frames — each with its own
 * instances (frame k's n records at instances + k * n), camera and light — in ONE pass on every device, with ONE gather for all of
 * them; rtm_trace_wait then hands out n_frames frames back to back (width x height pixels each), rtm_frame_device likewise, and the
 * statistics are sums over the pass.  rtm_set_instances puts the slot back to single frames. */
int rtm_set_batch(rtm_ctx* ctx, int slot, int n_frames, const rt_instance* instances, int n, const rt_uniforms* uniforms, int update);
int rtm_trace_async(rtm_ctx* ctx, int slot, int width, int height);
/* vkWaitForFences for that frame: pixels = width*height RGBA32F (or RGBA8 after rtm_set_param "output_rgba8" 1), valid
 * until the next rtm_trace_async on the slot (NULL with "host_copy" 0); stats = ray counts summed over the devices, kernel
 * times and visit counters of the root device.  If a device had to render its bands a second time (rt_stats.frames_rerendered,
 * a k_tail fault) the gather, de-interleave and copy of the slot are repeated here before the call returns. */
int rtm_trace_wait(rtm_ctx* ctx, int slot, const void** pixels, rt_stats* stats);

/* Device pointer (root GPU) of the slot's assembled width*height frame, valid after rtm_trace_wait until the next
 * rtm_trace_async on the slot (and across buffer growth only until then). */
const void* rtm_frame_device(const rtm_ctx* ctx, int slot);
int rtm_device_count(const rtm_ctx* ctx);
const char* rtm_last_error(const rtm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_MULTI_H */
