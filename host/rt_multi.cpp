// rt_multi.cpp — librt_multi.so (include/rt_multi.h): one host process, several MI355X of one node.
// Plain C++ over the C ABI of librt_mi355x.so (include/rt_api.h) and RCCL (/opt/rocm/include/rccl/rccl.h), called directly:
// no Python, no torch.  What it replaces in the reference: the single-device setup (src/main.cpp:928, 1108) and the copy of
// the traced image into the presented one (src/main.cpp:2683-2686).
//
// Per frame and slot: every device renders its interleaved 8-row bands on the slot's stream (rt_trace_shard), then ONE
// gather brings the compact shards to the root over xGMI — ncclGather per device inside ncclGroupStart/End; the frame slots
// share MAX_COMM_SETS communicator sets round-robin, so that frames in flight rarely queue behind one another on a
// communicator without creating slots x devices of them — and the root de-interleaves (rt_assemble_shards) and copies the
// frame to pinned host memory ("host_copy" 0: leaves it on the root device).  Nothing on the data path waits on the host
// between those steps.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_multi.h"

namespace {

constexpr int BAND_ROWS = 8;
constexpr int MAX_COMM_SETS = 4;   // communicator sets (each: one communicator per device); slot j uses set j % n
thread_local std::string g_create_error;

struct Device {
  int id = 0;
  std::vector<rt_ctx*> slots;        // slots[0] owns the scene, the others are frame slots on it
  std::vector<hipStream_t> streams;  // one per slot
  std::vector<void*> shard;          // compact shard of each slot (device memory)
  std::vector<hipEvent_t> done;      // loopback: the shard of a slot is complete
};

}  // namespace

struct rtm_ctx {
  std::vector<Device> dev;
  int P = 1;
  int flags = 0;
  bool rgba8 = false;
  bool host_copy = true;                       // rtm_set_param "host_copy": copy every assembled frame to pinned host memory
  std::vector<std::vector<ncclComm_t>> comm;   // [set][device], min(frames in flight, MAX_COMM_SETS) sets
  // root side, per slot
  std::vector<void*> gathered, frame;
  std::vector<void*> h_frame;
  std::vector<int> pending_w, pending_h, pending_k;
  std::vector<int> batch_k;                   // frames per pass of every slot (rtm_set_batch; 1 after rtm_set_instances)
  size_t stride = 0, frame_bytes = 0;         // current allocation (bytes per shard, bytes per frame)
  int alloc_k = 1;                            // frames per pass the buffers hold
  std::string error;
};

namespace {

int fail(rtm_ctx* c, int code, const std::string& msg) {
  if (c) c->error = msg; else g_create_error = msg;
  return code;
}
#define HIPM(c, expr)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess)                                                                                     \
      return fail(c, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE,                        \
                  std::string("HIP runtime exception: return code ") + std::to_string((int)e_) + " (" +       \
                      hipGetErrorString(e_) + ") in " #expr);                                                 \
  } while (0)
#define NCCLM(c, expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess)                                                                                    \
      return fail(c, RT_ERR_DEVICE, std::string("RCCL exception: return code ") + std::to_string((int)r_) +   \
                                        " (" + ncclGetErrorString(r_) + ") in " #expr);                       \
  } while (0)
// a failing rt_* call on one device: keep its message
#define RTM(c, d, s, expr)                                                                                    \
  do {                                                                                                        \
    int r_ = (expr);                                                                                          \
    if (r_) return fail(c, r_, std::string("device ") + std::to_string((c)->dev[d].id) + ": " + rt_last_error((c)->dev[d].slots[s])); \
  } while (0)

int max_shard_rows(int H, int n) {
  int m = 0;
  for (int s = 0; s < n; s++) m = std::max(m, rt_shard_rows(H, BAND_ROWS, s, n));
  return m;
}

// K: frames per pass (frame batches): every buffer holds K frames' worth — a device's K compact shards `stride` apart, the root's
// gathered block [device][frame], K assembled frames back to back
int ensure_buffers(rtm_ctx* c, int W, int H, int K) {
  const size_t bpp = c->rgba8 ? 4 : 16;
  const int n = (int)c->dev.size();
  size_t stride = (size_t)max_shard_rows(H, n) * W * bpp, frame_bytes = (size_t)W * H * bpp;
  if (stride <= c->stride && frame_bytes <= c->frame_bytes && K <= c->alloc_k) return RT_OK;
  stride = std::max(stride, c->stride); frame_bytes = std::max(frame_bytes, c->frame_bytes); K = std::max(K, c->alloc_k);
  // growing: nothing may be in flight on the old buffers
  for (auto& d : c->dev) { HIPM(c, hipSetDevice(d.id)); for (auto s : d.streams) HIPM(c, hipStreamSynchronize(s)); }
  for (auto& d : c->dev) {
    HIPM(c, hipSetDevice(d.id));
    for (auto& p : d.shard) { if (p) HIPM(c, hipFree(p)); p = nullptr; HIPM(c, hipMalloc(&p, stride * K)); }
  }
  HIPM(c, hipSetDevice(c->dev[0].id));
  for (int j = 0; j < c->P; j++) {
    if (c->gathered[j]) HIPM(c, hipFree(c->gathered[j]));
    if (c->frame[j]) HIPM(c, hipFree(c->frame[j]));
    if (c->h_frame[j]) HIPM(c, hipHostFree(c->h_frame[j]));
    c->gathered[j] = c->frame[j] = c->h_frame[j] = nullptr;
    HIPM(c, hipMalloc(&c->gathered[j], stride * n * K));
    HIPM(c, hipMalloc(&c->frame[j], frame_bytes * K));
    HIPM(c, hipHostMalloc(&c->h_frame[j], frame_bytes * K, hipHostMallocDefault));
  }
  c->stride = stride; c->frame_bytes = frame_bytes; c->alloc_k = K;
  return RT_OK;
}

// steps 2 and 3 of a frame, enqueued behind the devices' bands: ONE gather of the compact shards to the root, the de-interleave
// and (host_copy) the copy to pinned host memory
int gather_and_assemble(rtm_ctx* c, int slot, int W, int H, int K) {
  const int n = (int)c->dev.size();
  const size_t bpp = c->rgba8 ? 4 : 16, frame_bytes = (size_t)W * H * bpp;
  const size_t part = c->stride * (size_t)K;     // what one device contributes: its K shards
  hipStream_t root_stream = c->dev[0].streams[slot];
  if (c->flags & RTM_LOOPBACK) {
    for (int d = 0; d < n; d++) {
      Device& D = c->dev[d];
      HIPM(c, hipSetDevice(D.id));
      if (d != 0) { HIPM(c, hipEventRecord(D.done[slot], D.streams[slot])); }
    }
    HIPM(c, hipSetDevice(c->dev[0].id));
    for (int d = 0; d < n; d++) {
      if (d != 0) HIPM(c, hipStreamWaitEvent(root_stream, c->dev[d].done[slot], 0));
      HIPM(c, hipMemcpyAsync((char*)c->gathered[slot] + (size_t)d * part, c->dev[d].shard[slot], part, hipMemcpyDeviceToDevice, root_stream));
    }
  } else {
    std::vector<ncclComm_t>& set = c->comm[(size_t)slot % c->comm.size()];
    NCCLM(c, ncclGroupStart());
    for (int d = 0; d < n; d++) {
      Device& D = c->dev[d];
      HIPM(c, hipSetDevice(D.id));
      NCCLM(c, ncclGather(D.shard[slot], d == 0 ? c->gathered[slot] : nullptr, part, ncclUint8, 0, set[d], D.streams[slot]));
    }
    NCCLM(c, ncclGroupEnd());
  }
  // the root de-interleaves and hands the frame to the host
  HIPM(c, hipSetDevice(c->dev[0].id));
  // (frame k of a pass: its shards lie `part` apart in the gathered block, k * stride in; the assembled frames go back to back)
  for (int k = 0; k < K; k++)
    RTM(c, 0, slot, rt_assemble_shards(c->dev[0].slots[slot], (const char*)c->gathered[slot] + (size_t)k * c->stride, n, part, W, H, BAND_ROWS,
                                       (char*)c->frame[slot] + (size_t)k * frame_bytes, c->frame_bytes, root_stream));
  if (c->host_copy) HIPM(c, hipMemcpyAsync(c->h_frame[slot], c->frame[slot], frame_bytes * (size_t)K, hipMemcpyDeviceToHost, root_stream));
  return RT_OK;
}

}  // namespace

extern "C" {

int rtm_create(rtm_ctx** out, int n_devices, const int* device_ids, int frames_in_flight, int flags) {
  if (!out || n_devices <= 0 || !device_ids || frames_in_flight <= 0 || frames_in_flight > 16)
    return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "bad rtm_create arguments (1..16 frames in flight)");
  *out = nullptr;
  if (!(flags & RTM_LOOPBACK))
    for (int a = 0; a < n_devices; a++)
      for (int b = a + 1; b < n_devices; b++)
        if (device_ids[a] == device_ids[b]) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "a device may appear once (RCCL ranks are distinct GPUs); use RTM_LOOPBACK for logical shards on one GPU");
  rtm_ctx* c = new rtm_ctx();
  c->P = frames_in_flight; c->flags = flags;
  c->dev.resize(n_devices);
  auto bail = [&](int code) { g_create_error = c->error; rtm_destroy(c); return code; };
  for (int r = 0; r < n_devices; r++) {
    Device& d = c->dev[r];
    d.id = device_ids[r];
    rt_ctx* root = nullptr;
    int rc = rt_create(&root, d.id);
    if (rc) { c->error = rt_last_error(nullptr); return bail(rc); }
    d.slots.push_back(root);
    for (int j = 1; j < c->P; j++) {
      rt_ctx* s = nullptr;
      rc = rt_create_frame_slot(root, &s);
      if (rc) { c->error = rt_last_error(nullptr); return bail(rc); }
      d.slots.push_back(s);
    }
    if (hipSetDevice(d.id) != hipSuccess) { c->error = "hipSetDevice failed"; return bail(RT_ERR_DEVICE); }
    d.streams.resize(c->P); d.shard.assign(c->P, nullptr); d.done.resize(c->P);
    for (int j = 0; j < c->P; j++) {
      if (hipStreamCreateWithFlags(&d.streams[j], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d.done[j], hipEventDisableTiming) != hipSuccess) {
        c->error = "stream/event creation failed"; return bail(RT_ERR_DEVICE);
      }
    }
  }
  c->gathered.assign(c->P, nullptr); c->frame.assign(c->P, nullptr); c->h_frame.assign(c->P, nullptr);
  c->pending_w.assign(c->P, 0); c->pending_h.assign(c->P, 0); c->pending_k.assign(c->P, 1); c->batch_k.assign(c->P, 1);
  if (!(flags & RTM_LOOPBACK)) {
    // a few communicator sets shared by the frame slots: collectives of frames in flight rarely queue behind one another, and the
    // number of communicators does not grow with slots x devices (16 x 8 = 128 of them took seconds to create and pinned buffers each)
    const int n_sets = std::min(c->P, MAX_COMM_SETS);
    c->comm.assign(n_sets, std::vector<ncclComm_t>(n_devices, nullptr));
    for (int j = 0; j < n_sets; j++) {
      ncclResult_t r = ncclCommInitAll(c->comm[j].data(), n_devices, device_ids);
      if (r != ncclSuccess) { c->error = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); return bail(RT_ERR_DEVICE); }
    }
  }
  *out = c;
  return RT_OK;
}

void rtm_destroy(rtm_ctx* c) {
  if (!c) return;
  for (auto& d : c->dev) {
    (void)hipSetDevice(d.id);
    for (auto s : d.streams) if (s) (void)hipStreamSynchronize(s);
  }
  for (auto& cs : c->comm) for (auto cm : cs) if (cm) ncclCommDestroy(cm);
  if (!c->dev.empty()) {
    (void)hipSetDevice(c->dev[0].id);
    for (auto p : c->gathered) if (p) (void)hipFree(p);
    for (auto p : c->frame) if (p) (void)hipFree(p);
    for (auto p : c->h_frame) if (p) (void)hipHostFree(p);
  }
  for (auto& d : c->dev) {
    (void)hipSetDevice(d.id);
    for (auto p : d.shard) if (p) (void)hipFree(p);
    for (auto e : d.done) if (e) (void)hipEventDestroy(e);
    for (auto s : d.streams) if (s) (void)hipStreamDestroy(s);
    for (size_t j = d.slots.size(); j-- > 0;) rt_destroy(d.slots[j]);   // frame slots first, the scene owner last
  }
  delete c;
}

int rtm_device_count(const rtm_ctx* c) { return c ? (int)c->dev.size() : 0; }
const char* rtm_last_error(const rtm_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int rtm_upload_geometry(rtm_ctx* c, const float* verts6, size_t n_floats, const uint32_t* idx, size_t n_idx, const rt_mesh_range* ranges, int n_meshes) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, 0, rt_upload_geometry(c->dev[d].slots[0], verts6, n_floats, idx, n_idx, ranges, n_meshes));
  return RT_OK;
}
int rtm_build_blas(rtm_ctx* c, int mesh) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, 0, rt_build_blas(c->dev[d].slots[0], mesh));
  return RT_OK;
}
int rtm_set_skybox(rtm_ctx* c, const uint8_t* const faces[6], int w, int h) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, 0, rt_set_skybox(c->dev[d].slots[0], faces, w, h));
  return RT_OK;
}
int rtm_set_param(rtm_ctx* c, const char* name, int value) {
  if (!c || !name) return RT_ERR_INVALID_ARGUMENT;
  if (std::string(name) == "host_copy") { c->host_copy = value != 0; return RT_OK; }   // this library's own knob (not an rt_set_param)
  for (size_t d = 0; d < c->dev.size(); d++)
    for (int j = 0; j < c->P; j++) RTM(c, d, j, rt_set_param(c->dev[d].slots[j], name, value));
  if (std::string(name) == "output_rgba8" || std::string(name) == "output_bgra8") { c->rgba8 = value != 0; c->stride = 0; c->frame_bytes = 0; }
  return RT_OK;
}
int rtm_set_materials(rtm_ctx* c, const rt_material* table, int n_materials, const uint32_t* prim_material, size_t n_prims) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, 0, rt_set_materials(c->dev[d].slots[0], table, n_materials, prim_material, n_prims));
  return RT_OK;
}
int rtm_set_instance_types(rtm_ctx* c, int slot, const uint32_t* types, int n) {
  if (!c || slot < 0 || slot >= c->P) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, slot, rt_set_instance_types(c->dev[d].slots[slot], types, n));
  return RT_OK;
}
int rtm_set_timing(rtm_ctx* c, int enabled) {
  if (!c) return RT_ERR_INVALID_ARGUMENT;
  for (int j = 0; j < c->P; j++) RTM(c, 0, j, rt_set_timing(c->dev[0].slots[j], enabled));
  return RT_OK;
}
const void* rtm_frame_device(const rtm_ctx* c, int slot) { return (c && slot >= 0 && slot < c->P) ? c->frame[slot] : nullptr; }

int rtm_set_instances(rtm_ctx* c, int slot, const rt_instance* inst, int n, int update) {
  if (!c || slot < 0 || slot >= c->P) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, slot, rt_set_instances(c->dev[d].slots[slot], inst, n, update));
  c->batch_k[slot] = 1;
  return RT_OK;
}
int rtm_set_batch(rtm_ctx* c, int slot, int n_frames, const rt_instance* instances, int n, const rt_uniforms* uniforms, int update) {
  if (!c || slot < 0 || slot >= c->P) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, slot, rt_set_batch(c->dev[d].slots[slot], n_frames, instances, n, uniforms, update));
  c->batch_k[slot] = n_frames;
  return RT_OK;
}
int rtm_set_uniforms(rtm_ctx* c, int slot, const rt_uniforms* u) {
  if (!c || slot < 0 || slot >= c->P) return RT_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < c->dev.size(); d++) RTM(c, d, slot, rt_set_uniforms(c->dev[d].slots[slot], u));
  return RT_OK;
}

int rtm_trace_async(rtm_ctx* c, int slot, int W, int H) {
  if (!c || slot < 0 || slot >= c->P || W <= 0 || H <= 0) return c ? fail(c, RT_ERR_INVALID_ARGUMENT, "bad rtm_trace_async arguments") : RT_ERR_INVALID_ARGUMENT;
  if (c->pending_w[slot]) return fail(c, RT_ERR_NOT_READY, "rtm_trace_async: the previous frame of this slot has not been collected (rtm_trace_wait)");
  const int K = c->batch_k[slot];
  int r = ensure_buffers(c, W, H, K); if (r) return r;
  const int n = (int)c->dev.size();
  // 1. every device renders its bands on the slot's stream (of the K frames of the slot's pass: rtm_set_batch)
  for (int d = 0; d < n; d++) {
    Device& D = c->dev[d];
    if (K == 1) RTM(c, d, slot, rt_trace_shard(D.slots[slot], W, H, BAND_ROWS, d, n, D.shard[slot], c->stride, D.streams[slot]));
    else RTM(c, d, slot, rt_trace_shard_batch(D.slots[slot], W, H, BAND_ROWS, d, n, D.shard[slot], c->stride, c->stride * (size_t)K, D.streams[slot]));
  }
  r = gather_and_assemble(c, slot, W, H, K); if (r) return r;
  c->pending_w[slot] = W; c->pending_h[slot] = H; c->pending_k[slot] = K;
  return RT_OK;
}

int rtm_trace_wait(rtm_ctx* c, int slot, const void** pixels, rt_stats* stats) {
  if (!c || slot < 0 || slot >= c->P) return RT_ERR_INVALID_ARGUMENT;
  if (!c->pending_w[slot]) return fail(c, RT_ERR_NOT_READY, "rtm_trace_wait without rtm_trace_async");
  const int W = c->pending_w[slot], H = c->pending_h[slot];
  c->pending_w[slot] = c->pending_h[slot] = 0;
  rt_stats sum{};
  bool again = false;
  for (size_t d = 0; d < c->dev.size(); d++) {
    rt_stats st{};
    RTM(c, d, slot, rt_get_stats(c->dev[d].slots[slot], &st));   // waits for the device's part of the frame
    if (d == 0) sum = st;                                         // kernel times, visit counters: the root device's
    else { sum.rays_primary += st.rays_primary; sum.rays_secondary += st.rays_secondary; sum.rays_shadow += st.rays_shadow; sum.closest_rays += st.closest_rays;
           sum.tail_faults += st.tail_faults; sum.frames_rerendered += st.frames_rerendered; }
    again = again || st.frames_rerendered != 0;
  }
  if (again) {
    // a device rendered its bands a second time (k_tail fault, rt_api.h rt_trace_shard): the gather enqueued behind the first
    // attempt took the incomplete shard — gather, de-interleave and copy this slot's frame again
    int r = gather_and_assemble(c, slot, W, H, c->pending_k[slot]); if (r) return r;
    if (!(c->flags & RTM_LOOPBACK))
      for (auto& D : c->dev) { HIPM(c, hipSetDevice(D.id)); HIPM(c, hipStreamSynchronize(D.streams[slot])); }
  }
  HIPM(c, hipSetDevice(c->dev[0].id));
  HIPM(c, hipStreamSynchronize(c->dev[0].streams[slot]));        // gather, de-interleave, copy
  if (pixels) *pixels = c->host_copy ? c->h_frame[slot] : nullptr;
  if (stats) *stats = sum;
  return RT_OK;
}

}  // extern "C"
