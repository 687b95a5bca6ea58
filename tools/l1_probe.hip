// l1_probe.hip — what does ONE divergent 16-byte lane request cost a CU's vector-memory path on gfx950?
// The traversal kernels fetch a 32-byte node per lane and visit (two global_load_dwordx4 from a per-lane address).  This probe
// runs that access pattern with nothing else around it — a dependent chain per lane, the next index derived from the loaded words —
// at the traversal kernels' occupancy (5 workgroups of 256 per CU) and varies (a) requests per step (1 or 2 x 16 B, or 2 x 16 B of
// two different nodes), (b) how many neighbouring lanes share a node (1, 4, 16, 64), (c) the working set (L1-resident, L2-resident,
// beyond L2).  Output: nanoseconds and CU cycles per wave-step, and lane requests per CU cycle.
//   hipcc -O3 --offload-arch=gfx950 -o l1_probe tools/l1_probe.hip && ./l1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: one 16-byte request per step; 1: two requests to the same 32-byte node; 2: two requests to two different nodes;
// 3: MODE 1 plus `valu` dependent fmas per step (the box tests); 4: two requests per step like MODE 1, but issued by lane PAIRS: in the
// first instruction both lanes of a pair fetch the two halves of the even lane's node, in the second those of the odd lane's node
// (adjacent 16-byte pieces from adjacent lanes: one 32-byte access per pair if the address unit merges them), halves exchanged by DPP (quad_perm 1,0,3,2)
// MODE 5: the same chain with the nodes in LDS (the first 512 nodes, 16 KB, copied by the workgroup): two ds_read_b128 per step — what a
// walk over LDS-staged subtrees would pay per visit instead
template <int MODE>
__global__ __launch_bounds__(256) void k_chase(const uint4* __restrict__ nodes, uint32_t mask, uint32_t share_shift, int steps, int valu, uint32_t* out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  // lanes that share a node start from the same index and follow the same chain
  uint32_t cur = ((tid >> share_shift) * 2654435761u) & mask;
  uint32_t acc = 0; float f = 1.0f;
  __shared__ uint4 s_nodes[MODE == 5 ? 1024 : 1];
  if (MODE == 5) {
    for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) s_nodes[i] = nodes[i];
    __syncthreads();
    cur &= 511u;
    for (int s = 0; s < steps; s++) {
      const uint4 a = s_nodes[2u * cur], b = s_nodes[2u * cur + 1u];
      acc += a.y ^ b.z ^ a.z ^ a.w ^ b.x ^ b.y;
      cur = (a.x + b.w) & 511u;
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (tid == 0) out[1] = cur;
    return;
  }
  for (int s = 0; s < steps; s++) {
    const uint4 a = nodes[2u * cur];
    uint4 b = make_uint4(0, 0, 0, 0);
    if (MODE == 1 || MODE == 3) b = nodes[2u * cur + 1u];
    if (MODE == 2) b = nodes[2u * ((cur * 40503u + 977u) & mask) + 1u];
    if (MODE == 4) {
      const uint32_t odd = threadIdx.x & 1u, other = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, 0xB1, 0xF, 0xF, true);
      const uint32_t even_node = odd ? other : cur, odd_node = odd ? cur : other;
      const uint4 p = nodes[2u * even_node + odd], q = nodes[2u * odd_node + odd];
      // lane 2k needs (p of 2k, p of 2k+1); lane 2k+1 needs (q of 2k, q of 2k+1)
      const uint32_t give_x = odd ? p.x : q.x, give_w = odd ? p.w : q.w, give_y = odd ? p.y : q.y, give_z = odd ? p.z : q.z;
      const uint32_t got_x = (uint32_t)__builtin_amdgcn_mov_dpp((int)give_x, 0xB1, 0xF, 0xF, true), got_w = (uint32_t)__builtin_amdgcn_mov_dpp((int)give_w, 0xB1, 0xF, 0xF, true);
      const uint32_t got_y = (uint32_t)__builtin_amdgcn_mov_dpp((int)give_y, 0xB1, 0xF, 0xF, true), got_z = (uint32_t)__builtin_amdgcn_mov_dpp((int)give_z, 0xB1, 0xF, 0xF, true);
      const uint4 lo = odd ? make_uint4(got_x, got_y, got_z, got_w) : p, hi = odd ? q : make_uint4(got_x, got_y, got_z, got_w);
      acc += lo.y ^ hi.z ^ lo.z ^ lo.w ^ hi.x ^ hi.y; cur = (lo.x + hi.w) & mask;
      continue;
    }
    if (MODE == 3) { for (int v = 0; v < valu; v++) f = __builtin_fmaf(f, 1.0000001f, __uint_as_float(a.y & 0x007FFFFFu)); }
    acc += a.y ^ b.z ^ a.z ^ a.w ^ b.x ^ b.y;
    cur = (a.x + b.w) & mask;   // node words hold random next indices
  }
  if (acc == 0x12345678u) out[0] = acc;   // keeps every loaded word alive
  if (f == 3.0f) out[2] = 1u;
  if (tid == 0) out[1] = cur;
}

int main() {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int n_cu = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
  printf("# %s, %d CUs, %.2f GHz; 5 workgroups of 256 per CU (20 waves per CU), dependent chain of 2000 steps per lane\n", prop.gcnArchName, n_cu, ghz);
  const size_t max_nodes = (size_t)1 << 25;   // 1 GiB of 32-byte nodes
  std::vector<uint32_t> h(max_nodes * 8);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
  uint4* d; uint32_t* d_out;
  CHECK(hipMalloc((void**)&d, max_nodes * 32)); CHECK(hipMalloc((void**)&d_out, 64));
  CHECK(hipMemcpy(d, h.data(), max_nodes * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int steps = 2000;
  auto run = [&](int mode, uint32_t log2_nodes, uint32_t share_shift, int wg_per_cu, int valu) {
    const uint32_t mask = (1u << log2_nodes) - 1u;
    const int grid = n_cu * wg_per_cu;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      switch (mode) {
        case 0: k_chase<0><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
        case 1: k_chase<1><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
        case 2: k_chase<2><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
        case 4: k_chase<4><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
        case 5: k_chase<5><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
        default: k_chase<3><<<grid, 256>>>(d, mask, share_shift, steps, valu, d_out); break;
      }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double ns_step = best * 1e6 / steps;                      // every wave does `steps` steps, all waves concurrently
    const double cyc = ns_step * ghz;
    const int req = mode == 0 ? 1 : 2;   // (mode 5: two LDS reads)
    const double lane_req_per_cu_cycle = (double)wg_per_cu * 4 * 64 * req / cyc;
    printf("mode %d  set %8.2f MB  lanes/node %2u  wg/CU %d  valu %3d : %8.1f ns = %7.0f cycles per wave-step, %.3f lane requests per CU cycle\n",
           mode, (double)(1u << log2_nodes) * 32 / 1048576.0, 1u << share_shift, wg_per_cu, valu, ns_step, cyc, lane_req_per_cu_cycle);
  };
  for (uint32_t lg : {9u, 15u, 18u, 22u, 25u})            // 16 KB (L1), 1 MB, 8 MB (L2), 128 MB (Infinity Cache), 1 GB (HBM)
    for (int mode : {0, 1, 2, 4}) run(mode, lg, 0, 5, 0);
  for (uint32_t sh : {2u, 4u, 6u}) for (int mode : {0, 1}) run(mode, 18, sh, 5, 0);   // 4 / 16 / 64 lanes share a node
  for (int wg : {1, 2, 3, 4, 5, 6, 8}) run(1, 18, 0, wg, 0);                          // occupancy
  for (int valu : {0, 32, 64, 96, 128}) run(3, 18, 0, 5, valu);                       // + dependent VALU per step
  for (int valu : {0, 64}) run(3, 9, 0, 5, valu);
  for (int wg : {1, 3, 5}) run(5, 9, 0, wg, 0);                                        // nodes in LDS
  return 0;
}
