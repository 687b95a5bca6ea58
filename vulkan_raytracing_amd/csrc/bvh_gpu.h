// bvh_gpu.h — device-side BLAS builders (binned SAH level by level — the default —, LBVH, PLOC), see bvh_gpu.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "rt_device.h"

namespace rt {

struct GpuBlas {
  BvhNodeQ* nodes = nullptr;   // device, n_nodes entries, indices local to the mesh, root = node 0
  float4* tris = nullptr;      // device, 3 float4 per triangle, leaf order
  uint32_t n_nodes = 0, n_tris = 0;
  float q_lo[3] = {0, 0, 0}, q_scale[3] = {1, 1, 1};
  float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
};

// verts6 / idx are DEVICE pointers to the mesh's first vertex float and first index; n >= 8 triangles.
// Synchronous (like the reference's fence wait after vkCmdBuildAccelerationStructuresKHR, src/main.cpp:525-527).
int build_blas_gpu(const float* d_verts6, const uint32_t* d_idx, uint32_t n, hipStream_t s, GpuBlas& out, std::string& err);
void free_blas_gpu(GpuBlas& b);
void launch_rebase_nodes(const BvhNodeQ* src, BvhNodeQ* dst, uint32_t n, int node_base, uint32_t tri_base, hipStream_t s);

}  // namespace rt
