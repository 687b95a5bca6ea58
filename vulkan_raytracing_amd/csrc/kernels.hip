// kernels.hip — gfx950 (MI355X, CDNA4) kernels of the ray-tracing stage.  Hand-written HIP for one
// target only: 64-lane wavefronts, per-wave LDS traversal stacks, ballot/popcount queue compaction.
//
// Replaces, in the reference (paths relative to its root):
//   k_raygen        src/shader.rgen:57-79     jitter hash + primary ray
//   k_trace<...>    traceRayEXT, src/shader.rgen:86-87 (closest hit) and :111-112 (any hit, flags 13);
//                   the traversal itself is driver code in the reference
//   k_shade         src/shader.rchit:50-96, src/shader.rmiss:11, src/shader.rgen:90-177
//   shadow epilogue src/shader_shadow.rmiss:6 + src/shader.rgen:114-129
//   k_resolve       src/shader.rgen:64,180-185
//
// Arithmetic follows the canonical definition stated in DESIGN.md ("Canonical arithmetic"): IEEE
// binary32/64 +,-,*,/,sqrt and explicit fma only, in a fixed order; this file is compiled with
// -ffp-contract=off so nothing fuses unless written as __builtin_fmaf.  Box tests are the one
// exception — they only have to be conservative, so they use v_rcp_f32 and a slack factor.
#include <hip/hip_runtime.h>

#include "rt_kernels.h"

namespace rt {

// ------------------------------------------------------------------------------------------------
// small vector helpers (canonical forms)
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 mul3(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 neg3(F3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
  return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ float length3(F3 v) { return __builtin_sqrtf(dot3(v, v)); }
__device__ __forceinline__ F3 normalize3(F3 v) { float inv = 1.0f / length3(v); return mul3(v, inv); }
__device__ __forceinline__ F3 fma3(float s, F3 a, F3 b) { return mk3(__builtin_fmaf(s, a.x, b.x), __builtin_fmaf(s, a.y, b.y), __builtin_fmaf(s, a.z, b.z)); }
__device__ __forceinline__ F3 reflect3(F3 I, F3 N) { float k = 2.0f * dot3(N, I); return fma3(-k, N, I); }

__device__ __forceinline__ F3 xform_point(const float* m, F3 p) {
  return mk3(__builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, m[0] * p.x)) + m[3],
             __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, m[4] * p.x)) + m[7],
             __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, m[8] * p.x)) + m[11]);
}
__device__ __forceinline__ F3 xform_vec(const float* m, F3 p) {
  return mk3(__builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, m[0] * p.x)),
             __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, m[4] * p.x)),
             __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, m[8] * p.x)));
}
// vec3 * mat4x3 of src/shader.rchit:94
__device__ __forceinline__ F3 xform_normal(const float* w, F3 n) {
  return mk3(__builtin_fmaf(w[8], n.z, __builtin_fmaf(w[4], n.y, w[0] * n.x)),
             __builtin_fmaf(w[9], n.z, __builtin_fmaf(w[5], n.y, w[1] * n.x)),
             __builtin_fmaf(w[10], n.z, __builtin_fmaf(w[6], n.y, w[2] * n.x)));
}

// ------------------------------------------------------------------------------------------------
// canonical binary64 sine (same constants and operation order as oracle/rt_oracle.cpp canon_sin)
__device__ __forceinline__ double poly_sin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = __builtin_fma(z, S6, S5);
  p = __builtin_fma(z, p, S4);
  p = __builtin_fma(z, p, S3);
  p = __builtin_fma(z, p, S2);
  p = __builtin_fma(z, p, S1);
  return __builtin_fma(r * z, p, r);
}
__device__ __forceinline__ double poly_cos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = __builtin_fma(z, C6, C5);
  p = __builtin_fma(z, p, C4);
  p = __builtin_fma(z, p, C3);
  p = __builtin_fma(z, p, C2);
  p = __builtin_fma(z, p, C1);
  return __builtin_fma(z * z, p, __builtin_fma(z, -0.5, 1.0));
}
__device__ __forceinline__ double canon_sin(double x) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  double k = __builtin_rint(x * TWO_OVER_PI);
  double r = __builtin_fma(-k, PIO2_HI, x);
  r = __builtin_fma(-k, PIO2_LO, r);
  int q = (int)((long long)k & 3);
  double s = poly_sin(r), c = poly_cos(r);
  double v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
// src/shader.rgen:57-59
__device__ __forceinline__ float jitter_hash(float px, float py, float seed) {
  float d = px * 12.9898f + py * 78.233f;
  float a = d + 1113.1f * seed;
  float s = (float)canon_sin((double)a);
  float x = s * 43758.5453f;
  return x - __builtin_floorf(x);
}

__device__ __forceinline__ float pow100(float x) {
  float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32;
  return (x64 * x32) * x4;
}

// ------------------------------------------------------------------------------------------------
// wave-level helpers (wave64)
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t prefix_rank(uint64_t mask) {  // # set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Wavefront ballot compaction: lanes with `want` get consecutive slots from *counter.
__device__ __forceinline__ uint32_t wave_alloc(bool want, uint32_t* counter) {
  uint64_t mask = __ballot(want);
  uint32_t base = 0;
  if (mask != 0) {
    uint32_t leader = (uint32_t)__builtin_ctzll(mask);
    if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
    base = __shfl(base, (int)leader);
  }
  return base + prefix_rank(mask);
}

// ------------------------------------------------------------------------------------------------
// k_raygen: one thread per (8x8 pixel tile, sample, lane).  src/shader.rgen:62-79.
__global__ __launch_bounds__(256) void k_raygen(FrameDev f, UniformsDev u) {
  const uint32_t tiles_x = ((uint32_t)f.width + 7u) >> 3;
  const uint32_t tiles_y = ((uint32_t)f.rows + 7u) >> 3;
  const uint32_t spp = u.samples_per_pixel;
  const uint32_t total = tiles_x * tiles_y * spp * 64u;
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0) f.counters[CNT_QUEUE0] = total;
  if (q >= total) return;
  const uint32_t lane = q & 63u;
  const uint32_t ts = q >> 6;
  const uint32_t i = ts % spp;
  const uint32_t tile = ts / spp;
  const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u);
  const uint32_t ly = (tile / tiles_x) * 8u + (lane >> 3);
  if (x >= (uint32_t)f.width || ly >= (uint32_t)f.rows) {
    f.ray_o[0][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    f.ray_d[0][q] = make_float4(0.f, 0.f, 1.f, __uint_as_float(SID_DEAD));
    return;
  }
  const uint32_t band = ly / (uint32_t)f.band_rows;
  const uint32_t y = (band * (uint32_t)f.n_shards + (uint32_t)f.shard) * (uint32_t)f.band_rows + (ly % (uint32_t)f.band_rows);
  const float fx = (float)x, fy = (float)y;
  const float seed0 = (float)(spp + i), seed1 = seed0 + 0.5f;
  float ux = (fx + jitter_hash(fx, fy, seed0)) / (float)f.width;
  float uy = (fy + jitter_hash(fx, fy, seed1)) / (float)f.height;
  ux = __builtin_fmaf(ux, 2.0f, -1.0f);
  uy = -__builtin_fmaf(uy, 2.0f, -1.0f);
  F3 right = mk3(u.right[0], u.right[1], u.right[2]), up = mk3(u.up[0], u.up[1], u.up[2]), fwd = mk3(u.forward[0], u.forward[1], u.forward[2]);
  F3 d = normalize3(fma3(2.5f, fwd, fma3(uy, up, mul3(right, ux))));
  const uint32_t sid = i * (uint32_t)(f.rows * f.width) + ly * (uint32_t)f.width + x;
  f.ray_o[0][q] = make_float4(u.position[0], u.position[1], u.position[2], 10000.0f);
  f.ray_d[0][q] = make_float4(d.x, d.y, d.z, __uint_as_float(sid));
}

// ------------------------------------------------------------------------------------------------
// Traversal.  Two-level BVH2, one ray per lane, per-wave LDS stack [entry][lane] (bank = lane, so
// pushes/pops never conflict), spill to HBM above STACK_LDS entries.
struct TraceArgs {
  SceneDev sc;
  const float4* ray_o;
  const float4* ray_d;
  const uint32_t* n_ptr;
  // closest-hit pipeline outputs
  float4* hit_a;
  int32_t* hit_inst;
  // shadow pipeline
  const float4* sh_c;
  float4* sample_color;
  // raw mode
  HitRec* raw_out;
  int32_t* ovf_stack;
  uint32_t* counters;
  float tmin;
};

constexpr int MODE_CLOSEST = 0;  // pipeline closest hit: o.w = tmax, d.w = sid
constexpr int MODE_SHADOW = 1;   // pipeline any hit + shading epilogue
constexpr int MODE_RAW = 2;      // o.w = tmin, d.w = tmax; writes HitRec
constexpr int STACK_MARK = 0x7FFFFFFE;  // "return to world space" marker

__device__ __forceinline__ float safe_rcp(float d) {
  const float eps = 1e-20f;
  float a = __builtin_fabsf(d) < eps ? __builtin_copysignf(eps, d) : d;
  return __builtin_amdgcn_rcpf(a);
}

template <int MODE, bool ANY, bool COUNT>
__global__ __launch_bounds__(256) void k_trace(TraceArgs a) {
  __shared__ int s_stack[4][STACK_LDS][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  int(*stk)[64] = s_stack[wave];
  int32_t* ovf = a.ovf_stack + (size_t)(blockIdx.x * 256u + threadIdx.x) * STACK_OVF;
  const uint32_t n = *a.n_ptr;
  const uint32_t n_waves = gridDim.x * 4u;
  uint64_t cnt_nodes = 0, cnt_tris = 0;

  for (uint32_t base = (blockIdx.x * 4u + wave) * 64u; base < n; base += n_waves * 64u) {
    const uint32_t q = base + lane;
    bool active = q < n;
    float4 ro = make_float4(0.f, 0.f, 0.f, 0.f), rd = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { ro = a.ray_o[q]; rd = a.ray_d[q]; }
    float tmin, tmax;
    uint32_t sid = 0;
    if (MODE == MODE_RAW) { tmin = ro.w; tmax = rd.w; }
    else { tmin = a.tmin; tmax = ro.w; sid = __float_as_uint(rd.w); if (sid == SID_DEAD) active = false; }

    // world-space ray (kept for the return from an instance) and current-space ray
    const F3 wo = mk3(ro.x, ro.y, ro.z), wd = mk3(rd.x, rd.y, rd.z);
    F3 co = wo, cd = wd;
    F3 id = mk3(safe_rcp(cd.x), safe_rcp(cd.y), safe_rcp(cd.z));
    float best_t = tmax, best_u = 0.f, best_v = 0.f;
    int best_prim = -1, best_inst = -1;
    int cur_inst = -1;
    const BvhNode* nodes = a.sc.tlas_nodes;
    int sp = 0;
    int cur = 0;               // TLAS root (always interior)
    bool done = !active;

    auto push = [&](int v) {
      if (sp < STACK_LDS) stk[sp][lane] = v; else ovf[sp - STACK_LDS] = v;
      sp++;
    };
    auto pop = [&]() {
      for (;;) {
        if (sp == 0) { done = true; return; }
        sp--;
        cur = (sp < STACK_LDS) ? stk[sp][lane] : ovf[sp - STACK_LDS];
        if (cur != STACK_MARK) return;
        // leave the instance: back to the world-space ray and the TLAS
        co = wo; cd = wd;
        id = mk3(safe_rcp(cd.x), safe_rcp(cd.y), safe_rcp(cd.z));
        nodes = a.sc.tlas_nodes; cur_inst = -1;
      }
    };

    while (!done) {
      if (cur >= 0) {
        // ---- interior node: 56 useful bytes of one 64-byte record
        const float4* np = reinterpret_cast<const float4*>(nodes + cur);
        const float4 A = np[0], B = np[1], C = np[2];
        const int2 ch = *reinterpret_cast<const int2*>(np + 3);
        if (COUNT) cnt_nodes++;
        const float lim = best_t;
        float t0, t1;
        bool h0, h1;
        {
          float x0 = (A.x - co.x) * id.x, x1 = (A.y - co.x) * id.x;
          float y0 = (A.z - co.y) * id.y, y1 = (A.w - co.y) * id.y;
          float z0 = (C.x - co.z) * id.z, z1 = (C.y - co.z) * id.z;
          float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
          float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), lim));
          h0 = tn <= tf * 1.00002f; t0 = tn;
        }
        {
          float x0 = (B.x - co.x) * id.x, x1 = (B.y - co.x) * id.x;
          float y0 = (B.z - co.y) * id.y, y1 = (B.w - co.y) * id.y;
          float z0 = (C.z - co.z) * id.z, z1 = (C.w - co.z) * id.z;
          float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
          float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), lim));
          h1 = tn <= tf * 1.00002f; t1 = tn;
        }
        if (h0 && h1) {
          const bool swap = t1 < t0;
          push(swap ? ch.x : ch.y);
          cur = swap ? ch.y : ch.x;
        } else if (h0) cur = ch.x;
        else if (h1) cur = ch.y;
        else pop();
      } else if (cur_inst < 0) {
        // ---- TLAS leaf: enter an instance (ray -> object space, t preserved)
        const int ii = ~cur;
        const InstanceDev* I = a.sc.inst + ii;
        if ((I->mask & 0xFFu) == 0u) { pop(); continue; }
        float m[12];
        const float4* mp = reinterpret_cast<const float4*>(I->w2o);
        float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
        m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
        m[8] = m2.x; m[9] = m2.y; m[10] = m2.z; m[11] = m2.w;
        co = xform_point(m, wo); cd = xform_vec(m, wd);
        id = mk3(safe_rcp(cd.x), safe_rcp(cd.y), safe_rcp(cd.z));
        push(STACK_MARK);
        cur_inst = ii; nodes = a.sc.blas_nodes; cur = I->blas_root;
      } else {
        // ---- BLAS leaf: Möller–Trumbore on 48-byte packets (canonical form, see oracle tri_test)
        const uint32_t ref = (uint32_t)(~cur);
        const uint32_t first = ref >> 3, count = (ref & 7u) + 1u;
        for (uint32_t k = 0; k < count; k++) {
          const float4* tp = a.sc.tris + (size_t)(first + k) * 3;
          const float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
          if (COUNT) cnt_tris++;
          const F3 v0 = mk3(T0.x, T0.y, T0.z), e1 = mk3(T0.w, T1.x, T1.y), e2 = mk3(T1.z, T1.w, T2.x);
          const F3 p = cross3(cd, e2);
          const float det = dot3(e1, p);
          const F3 s = sub3(co, v0);
          float un = dot3(s, p);
          const F3 qv = cross3(s, e1);
          float vn = dot3(cd, qv);
          float tn = dot3(e2, qv);
          const float da = __builtin_fabsf(det);
          if (det < 0.0f) { un = -un; vn = -vn; tn = -tn; }
          if ((un >= 0.0f) && (vn >= 0.0f) && (un + vn <= da) && (da > 0.0f)) {
            const float inv = 1.0f / da;
            const float tt = tn * inv;
            if ((tt > tmin) && (tt < tmax)) {
              const int prim = (int)__float_as_uint(T2.y);
              const bool better = (best_inst < 0) || (tt < best_t) ||
                                  (tt == best_t && (cur_inst < best_inst || (cur_inst == best_inst && prim < best_prim)));
              if (better) { best_t = tt; best_u = un * inv; best_v = vn * inv; best_prim = prim; best_inst = cur_inst; }
            }
          }
        }
        if (ANY && best_inst >= 0) { done = true; }
        else pop();
      }
    }

    // ---- epilogue
    if (MODE == MODE_CLOSEST) {
      if (q < n) {
        a.hit_a[q] = make_float4(best_t, best_u, best_v, __uint_as_float((uint32_t)best_prim));
        a.hit_inst[q] = best_inst;
      }
    } else if (MODE == MODE_SHADOW) {
      if (active) {
        // src/shader_shadow.rmiss:6 + src/shader.rgen:114-129: lit iff nothing was hit
        const float4 c = a.sh_c[q];
        float r = 0.08f, g = 0.24f, b = 0.08f;
        if (best_inst < 0) { r = __builtin_fmaf(c.w, c.x, r); g = __builtin_fmaf(c.w, c.y, g); b = __builtin_fmaf(c.w, c.z, b); }
        a.sample_color[sid] = make_float4(r, g, b, 1.0f);
      }
    } else {
      if (q < n) {
        HitRec h;
        h.t = best_t; h.u = best_u; h.v = best_v; h.prim = best_prim; h.inst = best_inst;
        a.raw_out[q] = h;
      }
    }
  }
  if (COUNT) {
    // wave-reduce then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
      cnt_nodes += __shfl_down((unsigned long long)cnt_nodes, off);
      cnt_tris += __shfl_down((unsigned long long)cnt_tris, off);
    }
    if (lane == 0) {
      const int off = ANY ? (CNT_NODE_VISITS_SH - CNT_NODE_VISITS) : 0;
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + CNT_NODE_VISITS + off), (unsigned long long)cnt_nodes);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + CNT_TRI_TESTS + off), (unsigned long long)cnt_tris);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Cube-map lookup (LINEAR, CLAMP_TO_EDGE per face, RGBA8 UNORM) — same arithmetic as oracle sample_sky.
__device__ __forceinline__ F3 sample_sky(const SceneDev& sc, F3 r) {
  if (sc.sky_w == 0) return mk3(0.f, 0.f, 0.f);
  const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
  int layer; float s, t, ma;
  if (az >= ax && az >= ay) { ma = az; if (r.z >= 0.f) { layer = 4; s = r.x; t = -r.y; } else { layer = 5; s = -r.x; t = -r.y; } }
  else if (ay >= ax)        { ma = ay; if (r.y >= 0.f) { layer = 2; s = r.x; t = r.z; } else { layer = 3; s = r.x; t = -r.z; } }
  else                      { ma = ax; if (r.x >= 0.f) { layer = 0; s = -r.z; t = -r.y; } else { layer = 1; s = r.z; t = -r.y; } }
  const float fs = 0.5f * (s / ma + 1.0f), ft = 0.5f * (t / ma + 1.0f);
  const float u = fs * (float)sc.sky_w - 0.5f, v = ft * (float)sc.sky_h - 0.5f;
  const float fu0 = __builtin_floorf(u), fv0 = __builtin_floorf(v);
  const float wu = u - fu0, wv = v - fv0;
  int x0 = (int)fu0, y0 = (int)fv0, x1 = x0 + 1, y1 = y0 + 1;
  x0 = max(0, min(x0, sc.sky_w - 1)); x1 = max(0, min(x1, sc.sky_w - 1));
  y0 = max(0, min(y0, sc.sky_h - 1)); y1 = max(0, min(y1, sc.sky_h - 1));
  const uchar4* base = sc.sky + (size_t)layer * sc.sky_w * sc.sky_h;
  const uchar4 c00 = base[(size_t)y0 * sc.sky_w + x0], c10 = base[(size_t)y0 * sc.sky_w + x1];
  const uchar4 c01 = base[(size_t)y1 * sc.sky_w + x0], c11 = base[(size_t)y1 * sc.sky_w + x1];
  const float iu = 1.0f - wu, iv = 1.0f - wv;
  float ra = __builtin_fmaf((float)c10.x, wu, (float)c00.x * iu), rb = __builtin_fmaf((float)c11.x, wu, (float)c01.x * iu);
  float ga = __builtin_fmaf((float)c10.y, wu, (float)c00.y * iu), gb = __builtin_fmaf((float)c11.y, wu, (float)c01.y * iu);
  float ba = __builtin_fmaf((float)c10.z, wu, (float)c00.z * iu), bb = __builtin_fmaf((float)c11.z, wu, (float)c01.z * iu);
  return mk3(__builtin_fmaf(rb, wv, ra * iv) / 255.0f, __builtin_fmaf(gb, wv, ga * iv) / 255.0f, __builtin_fmaf(bb, wv, ba * iv) / 255.0f);
}

// ------------------------------------------------------------------------------------------------
// k_shade: closest-hit / miss shading and path continuation for one bounce.
struct ShadeArgs {
  SceneDev sc;
  FrameDev f;
  UniformsDev u;
  int bounce;
};

__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) {
  const FrameDev& f = a.f;
  const UniformsDev& U = a.u;
  const int cur = a.bounce & 1, nxt = cur ^ 1;
  const uint32_t n = f.counters[CNT_QUEUE0 + a.bounce];
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_round = (n + 63u) & ~63u;  // whole waves stay together for the ballots
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_round; q += stride) {
    bool push_next = false, push_shadow = false;
    F3 no = mk3(0, 0, 0), nd = mk3(0, 0, 1);
    float sh_tmax = 0.f; F3 sh_c = mk3(0, 0, 0); float sh_w = 0.f;
    uint32_t sid = SID_DEAD;
    if (q < n) {
      const float4 rd = f.ray_d[cur][q];
      sid = __float_as_uint(rd.w);
      if (sid != SID_DEAD) {
        const F3 d = mk3(rd.x, rd.y, rd.z);
        const int inst = f.hit_inst[q];
        if (inst < 0) {
          // src/shader.rmiss:11 + src/shader.rgen:90-94
          const F3 c = sample_sky(a.sc, mk3(d.x, d.y, -d.z));
          f.sample_color[sid] = make_float4(c.x, c.y, c.z, 1.0f);
        } else {
          // src/shader.rchit:50-96
          const float4 h = f.hit_a[q];
          const InstanceDev* I = a.sc.inst + inst;
          const uint32_t prim = __float_as_uint(h.w);
          const uint32_t* ix = a.sc.idx + I->first_index + 3u * prim;
          const uint32_t ia = ix[0], ib = ix[1], ic = ix[2];
          const float* vb = a.sc.verts + I->first_float;
          const float bx = (1.0f - h.y) - h.z, by = h.y, bz = h.z;
          const float* pa = vb + 6u * ia; const float* pb = vb + 6u * ib; const float* pc = vb + 6u * ic;
          const F3 pos = fma3(bz, mk3(pc[0], pc[1], pc[2]), fma3(by, mk3(pb[0], pb[1], pb[2]), mul3(mk3(pa[0], pa[1], pa[2]), bx)));
          const F3 nrm = fma3(bz, mk3(pc[3], pc[4], pc[5]), fma3(by, mk3(pb[3], pb[4], pb[5]), mul3(mk3(pa[3], pa[4], pa[5]), bx)));
          const F3 P = xform_point(I->o2w, pos);
          F3 N = normalize3(xform_normal(I->w2o, nrm));
          const int objectIndex = I->custom_index;
          const uint32_t type = objectIndex == 0 ? U.center_object_type : U.orbiting_object_type;
          const bool last = (uint32_t)a.bounce >= U.max_bounce_count;
          if (type == 0u) {
            // src/shader.rgen:97-131
            if (dot3(d, N) >= 0.0f) {
              f.sample_color[sid] = make_float4(0.08f, 0.24f, 0.08f, 1.0f);
            } else {
              no = fma3(0.01f, N, P);
              const F3 toL = sub3(mk3(U.light_position[0], U.light_position[1], U.light_position[2]), P);
              const float dist = length3(toL);
              const F3 L = mul3(toL, 1.0f / dist);
              const F3 Hh = normalize3(add3(L, neg3(d)));
              const float NdotL = dot3(N, L), NdotH = dot3(N, Hh);
              const float dl = fmaxf(0.0f, NdotL), sp = pow100(fmaxf(0.0f, NdotH));
              const uint32_t i = sid / (uint32_t)(f.rows * f.width);
              float w = 1.0f;
              for (uint32_t k = 0; k < i; k++) w = w * 0.9f;
              const float Iv = U.light_intensity;
              const F3 diff = mk3((Iv * 0.2f) * dl, (Iv * 1.0f) * dl, (Iv * 0.2f) * dl);
              const float sv = (Iv * 0.8f) * sp;
              sh_c = add3(diff, mk3(sv, sv, sv)); sh_w = w;
              nd = L; sh_tmax = dist;
              push_shadow = true;
            }
          } else if (type == 1u) {
            // src/shader.rgen:132-138
            no = fma3(0.01f, N, P);
            nd = reflect3(d, N);
            push_next = true;
          } else if (type == 2u) {
            // src/shader.rgen:139-165
            float ndoti = dot3(d, N);
            const bool outwards = ndoti > 0.0f;
            if (outwards) { N = neg3(N); ndoti = -ndoti; }
            const float ratio = outwards ? 1.52f : (1.0f / 1.52f);
            const float k = 1.0f - (ratio * ratio) * (1.0f - ndoti * ndoti);
            if (k < 0.0f) { nd = reflect3(d, N); no = fma3(0.01f, N, P); }
            else {
              const float c = __builtin_fmaf(ratio, ndoti, __builtin_sqrtf(k));
              nd = normalize3(fma3(-c, N, mul3(d, ratio)));
              no = fma3(-0.01f, N, P);
            }
            push_next = true;
          } else {
            // unknown type: the reference loop re-traces the unchanged ray until the bounce budget ends
            const float4 ro = f.ray_o[cur][q];
            no = mk3(ro.x, ro.y, ro.z); nd = d; push_next = true;
          }
          if (push_next && last) {
            // loop of src/shader.rgen:84 ends: tmpColor keeps Iamb*ka
            push_next = false;
            f.sample_color[sid] = make_float4(0.08f, 0.24f, 0.08f, 1.0f);
          }
        }
      }
    }
    // wavefront ballot compaction into the next-bounce queue / the shadow queue
    const uint32_t slot_n = wave_alloc(push_next, f.counters + CNT_QUEUE0 + a.bounce + 1);
    if (push_next) {
      f.ray_o[nxt][slot_n] = make_float4(no.x, no.y, no.z, 10000.0f);
      f.ray_d[nxt][slot_n] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(sid));
    }
    const uint32_t slot_s = wave_alloc(push_shadow, f.counters + CNT_SHADOW);
    if (push_shadow) {
      f.sh_o[slot_s] = make_float4(no.x, no.y, no.z, sh_tmax);
      f.sh_d[slot_s] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(sid));
      f.sh_c[slot_s] = make_float4(sh_c.x, sh_c.y, sh_c.z, sh_w);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_resolve: src/shader.rgen:64,180-185 — ordered sum over samples, divide, store.
__global__ __launch_bounds__(256) void k_resolve(FrameDev f, UniformsDev u) {
  const uint32_t npx = (uint32_t)(f.rows * f.width);
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npx) return;
  float r = 0.f, g = 0.f, b = 0.f, al = 0.f;
  for (uint32_t i = 0; i < u.samples_per_pixel; i++) {
    const float4 c = f.sample_color[(size_t)i * npx + p];
    r += c.x; g += c.y; b += c.z; al += c.w;
  }
  const float nn = (float)u.samples_per_pixel;
  f.out[p] = make_float4(r / nn, g / nn, b / nn, al / nn);
}

// ------------------------------------------------------------------------------------------------
// launchers
int trace_threads_per_block() { return 256; }

void launch_raygen(const FrameDev& f, const UniformsDev& u, hipStream_t s) {
  const uint32_t tiles = (((uint32_t)f.width + 7u) >> 3) * (((uint32_t)f.rows + 7u) >> 3);
  const uint32_t total = tiles * u.samples_per_pixel * 64u;
  hipLaunchKernelGGL(k_raygen, dim3((total + 255u) / 256u), dim3(256), 0, s, f, u);
}

static TraceArgs make_args(const SceneDev& sc) {
  TraceArgs a{};
  a.sc = sc;
  a.tmin = 0.001f;  // src/shader.rgen:87,112
  return a;
}

void launch_trace_closest(const SceneDev& sc, const FrameDev& f, int bounce, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc);
  a.ray_o = f.ray_o[bounce & 1]; a.ray_d = f.ray_d[bounce & 1];
  a.n_ptr = f.counters + CNT_QUEUE0 + bounce;
  a.hit_a = f.hit_a; a.hit_inst = f.hit_inst;
  a.ovf_stack = f.ovf_stack; a.counters = f.counters;
  if (counting) hipLaunchKernelGGL((k_trace<MODE_CLOSEST, false, true>), dim3(cfg.trace_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_trace<MODE_CLOSEST, false, false>), dim3(cfg.trace_blocks), dim3(256), 0, s, a);
}

void launch_trace_shadow(const SceneDev& sc, const FrameDev& f, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc);
  a.ray_o = f.sh_o; a.ray_d = f.sh_d; a.sh_c = f.sh_c;
  a.n_ptr = f.counters + CNT_SHADOW;
  a.sample_color = f.sample_color;
  a.ovf_stack = f.ovf_stack; a.counters = f.counters;
  if (counting) hipLaunchKernelGGL((k_trace<MODE_SHADOW, true, true>), dim3(cfg.trace_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_trace<MODE_SHADOW, true, false>), dim3(cfg.trace_blocks), dim3(256), 0, s, a);
}

void launch_trace_raw(const SceneDev& sc, const float4* ray_o, const float4* ray_d, HitRec* out, const uint32_t* n_ptr,
                      int32_t* ovf_stack, uint32_t* counters, bool any_hit, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc);
  a.ray_o = ray_o; a.ray_d = ray_d; a.n_ptr = n_ptr; a.raw_out = out; a.ovf_stack = ovf_stack; a.counters = counters;
  dim3 g(cfg.trace_blocks), b(256);
  if (any_hit) {
    if (counting) hipLaunchKernelGGL((k_trace<MODE_RAW, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace<MODE_RAW, true, false>), g, b, 0, s, a);
  } else {
    if (counting) hipLaunchKernelGGL((k_trace<MODE_RAW, false, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace<MODE_RAW, false, false>), g, b, 0, s, a);
  }
}

void launch_shade(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, int bounce, const LaunchCfg& cfg, hipStream_t s) {
  ShadeArgs a{sc, f, u, bounce};
  hipLaunchKernelGGL(k_shade, dim3(cfg.shade_blocks), dim3(256), 0, s, a);
}

void launch_resolve(const FrameDev& f, const UniformsDev& u, hipStream_t s) {
  const uint32_t npx = (uint32_t)(f.rows * f.width);
  hipLaunchKernelGGL(k_resolve, dim3((npx + 255u) / 256u), dim3(256), 0, s, f, u);
}

}  // namespace rt
