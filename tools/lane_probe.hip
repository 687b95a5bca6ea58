// lane_probe.hip — does a divergent 16-byte load cost the CU's vector-memory path per ACTIVE LANE, per quad of lanes, or per wave
// instruction?  The traversal kernels run their node fetches with 36 of 64 lanes active on average (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU);
// TCP_TOTAL_ACCESSES counts 64 per wave load whatever the exec mask.  Same chain as tools/l1_probe.hip (mode 1: two 16-byte requests
// to one 32-byte node per step, next index from the loaded words), 5 workgroups of 256 per CU, with only the lanes of `lane_mask` alive.
//   hipcc -O3 --offload-arch=gfx950 -o lane_probe tools/lane_probe.hip && ./lane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_chase(const uint4* __restrict__ nodes, uint32_t mask, unsigned long long lane_mask, int steps, uint32_t* out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  uint32_t cur = (tid * 2654435761u) & mask;
  uint32_t acc = 0;
  if ((lane_mask >> (threadIdx.x & 63u)) & 1ull) {
    for (int s = 0; s < steps; s++) {
      const uint4 a = nodes[2u * cur], b = nodes[2u * cur + 1u];
      acc += a.y ^ b.z ^ a.z ^ a.w ^ b.x ^ b.y;
      cur = (a.x + b.w) & mask;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
  if (tid == 0) out[1] = cur;
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
  printf("# %s, %d CUs, %.2f GHz; 5 workgroups of 256 per CU, two 16-byte requests per lane and step, 2000 dependent steps\n", prop.gcnArchName, n_cu, ghz);
  const size_t max_nodes = (size_t)1 << 18;   // 8 MB of 32-byte nodes (L2)
  std::vector<uint32_t> h(max_nodes * 8);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
  uint4* d; uint32_t* d_out;
  CHECK(hipMalloc((void**)&d, max_nodes * 32)); CHECK(hipMalloc((void**)&d_out, 64));
  CHECK(hipMemcpy(d, h.data(), max_nodes * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int steps = 2000;
  struct Case { const char* name; unsigned long long m; };
  unsigned long long rnd36 = 0; { uint64_t y = 12345; int n = 0; while (n < 36) { y = y * 6364136223846793005ull + 1442695040888963407ull; int b = (int)(y >> 58); if (!((rnd36 >> b) & 1)) { rnd36 |= 1ull << b; n++; } } }
  const Case cases[] = {
    {"64 lanes", ~0ull}, {"48 contiguous", (1ull << 48) - 1}, {"36 contiguous", (1ull << 36) - 1}, {"36 random", rnd36},
    {"32 contiguous", 0xFFFFFFFFull}, {"32 even lanes", 0x5555555555555555ull}, {"32 = 2 of every quad (lanes 0,1)", 0x3333333333333333ull},
    {"16 contiguous", 0xFFFFull}, {"16 = 1 of every quad", 0x1111111111111111ull}, {"16 = every 4th quad whole", 0x000F000F000F000Full},
    {"8 contiguous", 0xFFull}, {"8 = 1 of every 8", 0x0101010101010101ull}, {"1 lane", 1ull}};
  for (uint32_t lg : {9u, 18u}) {
    const uint32_t mask = (1u << lg) - 1u;
    for (const Case& c : cases) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        k_chase<<<n_cu * 5, 256>>>(d, mask, c.m, steps, d_out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double cyc = best * 1e6 / steps * ghz;
      const int lanes = __builtin_popcountll(c.m);
      printf("set %5.2f MB  %-36s : %7.0f cycles per wave-step, %.3f ACTIVE lane requests per CU cycle, %.3f wave loads per 100 CU cycles\n",
             (double)(1u << lg) * 32 / 1048576.0, c.name, cyc, 20.0 * lanes * 2 / cyc, 20.0 * 2 * 100 / cyc);
    }
  }
  return 0;
}
