// kernels.hip — gfx950 (MI355X, CDNA4) kernels of the ray-tracing stage.  Hand-written HIP for one
// target only: 64-lane wavefronts, per-wave LDS traversal stacks, ballot/popcount queue compaction.
//
// Replaces, in the reference (paths relative to its root):
//   k_cover         (none: what the driver's traversal does implicitly) which screen tiles can a mesh project onto
//   k_entry         (none) per tile of a view — the camera; optionally a cube around the light — the deep subtrees its rays can hit
//   k_raygen        src/shader.rgen:57-79     jitter hash + primary ray
//   k_trace<...>    traceRayEXT, src/shader.rgen:86-87 (closest hit) and :111-112 (any hit, flags 13);
//                   the traversal itself is driver code in the reference (k_packet: the same query, one wavefront per
//                   64-ray chunk — an alternative kept for comparison, off by default)
//   k_shade         src/shader.rchit:50-96, src/shader.rmiss:11, src/shader.rgen:90-177
//   shadow epilogue src/shader_shadow.rmiss:6 + src/shader.rgen:114-129
//   k_resolve       src/shader.rgen:64,180-185
//
// Arithmetic follows the canonical definition stated in DESIGN.md ("Canonical arithmetic"): IEEE
// binary32/64 +,-,*,/,sqrt and explicit fma only, in a fixed order; this file is compiled with
// -ffp-contract=off so nothing fuses unless written as __builtin_fmaf.  Box tests are the one
// exception — they only have to be conservative, so they use v_rcp_f32 and a slack factor.
#include <hip/hip_runtime.h>
#include <stdlib.h>

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
// x^n for an integer exponent 0..1023 (row n4: the MTL's Ns): the powers x^(2^k) by repeated squaring, multiplied together
// from the HIGHEST set bit down — for n = 100 exactly pow100's (x^64 * x^32) * x^4.  Same order in the oracle.
__device__ __forceinline__ float pow_int(float x, uint32_t n) {
  float p[10];
  p[0] = x;
#pragma unroll
  for (int k = 1; k < 10; k++) p[k] = p[k - 1] * p[k - 1];
  float acc = 1.0f;
  bool first = true;
#pragma unroll
  for (int k = 9; k >= 0; k--)
    if (n >> k & 1u) { acc = first ? p[k] : acc * p[k]; first = false; }
  return acc;
}
// Iamb * ka of the material a shadow-queue entry was tagged with (MATERIAL_NONE: the folded constant of shader.rgen.spv)
__device__ __forceinline__ F3 ambient_of(const SceneDev& sc, uint32_t mat) {
  if (mat == MATERIAL_NONE) return mk3(0.08f, 0.24f, 0.08f);
  const MaterialDev* M = sc.materials + mat;
  return mk3(0.8f * M->ka[0], 0.8f * M->ka[1], 0.8f * M->ka[2]);
}

// ------------------------------------------------------------------------------------------------
// wave-level helpers (wave64)
// Streamed data — ray queues, hit records, per-sample colours: written once by one kernel, read once by the next — can carry the
// non-temporal hint so that it does not push the acceleration structure out of the 4-MB L2 of its XCD (experiment: -DRT_NT_STREAMS).
typedef float rt_v4f __attribute__((ext_vector_type(4)));
#ifndef RT_NT_STREAMS
#define RT_NT_STREAMS 1   /* bit 0: loads (default), bit 1: stores, bit 2: the cube-map taps */
#endif
__device__ __forceinline__ float4 ld_stream(const float4* p) {
  if (RT_NT_STREAMS & 1) { const rt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const rt_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  return *p;
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
  if (RT_NT_STREAMS & 2) { rt_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<rt_v4f*>(p)); }
  else *p = v;
}
__device__ __forceinline__ int ld_stream(const int* p) { return (RT_NT_STREAMS & 1) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_stream(int* p, int v) { if (RT_NT_STREAMS & 2) __builtin_nontemporal_store(v, p); else *p = v; }
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

// frame of a sample id in a frame batch (<= BATCH_MAX = 8 frames of S sample ids each): three comparisons instead of a division
__device__ __forceinline__ uint32_t frame_of(uint32_t sid, uint32_t S) {
  uint32_t fi = sid >= 4u * S ? 4u : 0u;
  fi += sid >= (fi + 2u) * S ? 2u : 0u;
  fi += sid >= (fi + 1u) * S ? 1u : 0u;
  return fi;
}

// Queue cursors are read with agent scope: inside k_tail a queue is filled by other workgroups of the same launch.
__device__ __forceinline__ uint32_t ld_cursor(const uint32_t* p) {
  return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float safe_rcp(float d) {
  const float eps = 1e-20f;
  float a = __builtin_fabsf(d) < eps ? __builtin_copysignf(eps, d) : d;
  return __builtin_amdgcn_rcpf(a);
}

// Conservative slab test of one box (box tests need not be bit-reproducible, only never to reject a
// box the canonical triangle test would hit: sub-mul form, relative slack 2^-15.6).
__device__ __forceinline__ bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, F3 o, F3 id, float tmin, float tlim, float& tn) {
  float x0 = (lox - o.x) * id.x, x1 = (hix - o.x) * id.x;
  float y0 = (loy - o.y) * id.y, y1 = (hiy - o.y) * id.y;
  float z0 = (loz - o.z) * id.z, z1 = (hiz - o.z) * id.z;
  tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
  float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tlim));
  return tn <= tf * 1.00002f;
}

// Slab test on a quantized box (rt_device.h BvhNodeQ): w = lo | hi << 16 per axis; the ray space carries
// qs = q_scale/d and qb = (q_lo - o)/d, so a plane distance is one cvt + one fma.  Which plane of an axis is
// the near one depends only on the sign of the ray direction, so instead of a min and a max per axis the
// word is rotated by rot (0, or 16 for a negative direction) and its low half is the near plane.
__device__ __forceinline__ bool slab_q(uint32_t wx, uint32_t wy, uint32_t wz, F3 qs, F3 qb, uint3 rot, float tmin, float tlim, float& tn) {
  wx = __builtin_amdgcn_alignbit(wx, wx, rot.x); wy = __builtin_amdgcn_alignbit(wy, wy, rot.y); wz = __builtin_amdgcn_alignbit(wz, wz, rot.z);
  const float x0 = __builtin_fmaf((float)(wx & 0xFFFFu), qs.x, qb.x), x1 = __builtin_fmaf((float)(wx >> 16), qs.x, qb.x);
  const float y0 = __builtin_fmaf((float)(wy & 0xFFFFu), qs.y, qb.y), y1 = __builtin_fmaf((float)(wy >> 16), qs.y, qb.y);
  const float z0 = __builtin_fmaf((float)(wz & 0xFFFFu), qs.z, qb.z), z1 = __builtin_fmaf((float)(wz >> 16), qs.z, qb.z);
  tn = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
  const float tf = fminf(fminf(x1, y1), fminf(z1, tlim));
  return tn <= tf * 1.00002f;
}
// The one ABSOLUTE error of the plane distances: qb = (q_lo - o) / d loses what the subtraction q_lo - o rounds away, up to half an
// ulp of |q_lo - o| IN SPACE — nothing while the origin is near the tree, but from thousands of units away (the pipeline's tmax
// is 10000, src/shader.rgen:86-87, and the camera flies freely) it exceeds the two quanta of margin in the stored planes, and on
// an axis the ray is nearly perpendicular to no relative slack in t covers it (tools/far_probe.py: 1 record in 40 000 wrong from
// 20 000 units).  A ray is FAR when that rounding can reach a quarter of a quantum on some axis; far rays take the generic visit
// (interior_step), whose slab test moves every near plane down and every far plane up by 2^-22 |qb| on its own axis.  The
// headline's rays are never far (|q_lo - o| / q_scale ~ 2.5e5 quanta against the threshold of 2.1e6).
__device__ __forceinline__ bool quant_far(F3 qs, F3 qb) {
  const float K = 2097152.0f;   // 0.25 quantum / 2^-23
  return __builtin_fabsf(qb.x) > K * __builtin_fabsf(qs.x) || __builtin_fabsf(qb.y) > K * __builtin_fabsf(qs.y) || __builtin_fabsf(qb.z) > K * __builtin_fabsf(qs.z);
}
// the same test before the division by d: |q_lo - o| > K q_scale on some axis, against the smallest scale (3 VALU per ray space)
__device__ __forceinline__ bool quant_far_o(F3 o, const float* q_lo, const float* q_scale) {
  const float r = 2097152.0f * fminf(fminf(q_scale[0], q_scale[1]), q_scale[2]);
  return fmaxf(fmaxf(__builtin_fabsf(q_lo[0] - o.x), __builtin_fabsf(q_lo[1] - o.y)), __builtin_fabsf(q_lo[2] - o.z)) > r;
}
__device__ __forceinline__ bool slab_q_far(uint32_t wx, uint32_t wy, uint32_t wz, F3 qs, F3 qb, uint3 rot, float tmin, float tlim, float& tn) {
  const float ex = 2.4e-7f * __builtin_fabsf(qb.x), ey = 2.4e-7f * __builtin_fabsf(qb.y), ez = 2.4e-7f * __builtin_fabsf(qb.z);
  wx = __builtin_amdgcn_alignbit(wx, wx, rot.x); wy = __builtin_amdgcn_alignbit(wy, wy, rot.y); wz = __builtin_amdgcn_alignbit(wz, wz, rot.z);
  const float x0 = __builtin_fmaf((float)(wx & 0xFFFFu), qs.x, qb.x) - ex, x1 = __builtin_fmaf((float)(wx >> 16), qs.x, qb.x) + ex;
  const float y0 = __builtin_fmaf((float)(wy & 0xFFFFu), qs.y, qb.y) - ey, y1 = __builtin_fmaf((float)(wy >> 16), qs.y, qb.y) + ey;
  const float z0 = __builtin_fmaf((float)(wz & 0xFFFFu), qs.z, qb.z) - ez, z1 = __builtin_fmaf((float)(wz >> 16), qs.z, qb.z) + ez;
  tn = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
  const float tf = fminf(fminf(x1, y1), fminf(z1, tlim));
  return tn <= tf * 1.00002f;
}
// (qs, qb, rot) of a ray in the space of a tree with dequantisation (q_lo, q_scale)
__device__ __forceinline__ void quant_space(F3 o, F3 d, const float* q_lo, const float* q_scale, F3& qs, F3& qb, uint3& rot) {
  const F3 id = mk3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
  qs = mk3(q_scale[0] * id.x, q_scale[1] * id.y, q_scale[2] * id.z);
  qb = mk3((q_lo[0] - o.x) * id.x, (q_lo[1] - o.y) * id.y, (q_lo[2] - o.z) * id.z);
  rot = make_uint3(qs.x < 0.0f ? 16u : 0u, qs.y < 0.0f ? 16u : 0u, qs.z < 0.0f ? 16u : 0u);   // q_scale > 0: the sign of 1/d
}

// Canonical Moller-Trumbore on one 48-byte packet (see oracle tri_test): two-sided, division-free
// rejection, one IEEE reciprocal for an accepted candidate, accept iff tmin < t < tmax.
__device__ __forceinline__ bool tri_test(const float4 T0, const float4 T1, const float4 T2, F3 co, F3 cd, float tmin, float tmax, float& t, float& u, float& v) {
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
  if (!((un >= 0.0f) && (vn >= 0.0f) && (un + vn <= da) && (da > 0.0f))) return false;
  const float inv = 1.0f / da;
  const float tt = tn * inv;
  if (!((tt > tmin) && (tt < tmax))) return false;
  t = tt; u = un * inv; v = vn * inv;
  return true;
}

// x / 255.0f without the ten-instruction IEEE division sequence: q = x*rc, one fma for the exact remainder, one fma
// to correct q (rc = RN(1/255)).  The result equals the IEEE quotient — which is what the canonical definition and
// the oracle use — for EVERY binary32 x in [0, 256]: proved exhaustively by tools/check_div255.c (1.13e9 values,
// sampled in tests/test_oracle.py); the filter below only produces values in [0, 255].
__device__ __forceinline__ float div255(float x) {
  const float rc = 0x1.010102p-8f;   // 0x3b808081
  const float q = x * rc;
  const float r = __builtin_fmaf(-q, 255.0f, x);
  return __builtin_fmaf(r, rc, q);
}

// ------------------------------------------------------------------------------------------------
// Cube-map lookup (LINEAR on a cube view, RGBA8 UNORM, src/main.cpp:2393-2406) — same arithmetic as oracle sample_sky.
// Footprint texels that fall off the selected face come from the neighbouring face (Vulkan cube map edge handling); the
// tap beyond a corner is the mean of the other three.  Only ~1/W of all lookups touch an edge, so that path lives in its
// own function and the common path stays four plain loads.
struct SkyTap { int layer, x, y; };
__device__ __forceinline__ SkyTap sky_neighbour(int layer, int x, int y, int W) {
  const int S = 2 * x + 1 - W, T = 2 * y + 1 - W;
  int px, py, pz;
  switch (layer) {
    case 0: px = W; py = -T; pz = -S; break;
    case 1: px = -W; py = -T; pz = S; break;
    case 2: px = S; py = W; pz = T; break;
    case 3: px = S; py = -W; pz = -T; break;
    case 4: px = S; py = -T; pz = W; break;
    default: px = -S; py = -T; pz = -W; break;
  }
  const int ax = abs(px), ay = abs(py), az = abs(pz);
  int nl, sc, tc, ma;
  if (az >= ax && az >= ay) { ma = az; if (pz >= 0) { nl = 4; sc = px; tc = -py; } else { nl = 5; sc = -px; tc = -py; } }
  else if (ay >= ax)        { ma = ay; if (py >= 0) { nl = 2; sc = px; tc = pz; } else { nl = 3; sc = px; tc = -pz; } }
  else                      { ma = ax; if (px >= 0) { nl = 0; sc = -pz; tc = -py; } else { nl = 1; sc = pz; tc = -py; } }
  SkyTap t;
  t.layer = nl; t.x = min(((sc + ma) * W) / (2 * ma), W - 1); t.y = min(((tc + ma) * W) / (2 * ma), W - 1);
  return t;
}
// texel index (in uchar4 units from the start of the cube map) of tap (x, y) of `layer`, either coordinate possibly one
// texel beyond the face; SKY_CORNER when both are (no unique neighbour).  Out of line and returning one register: only
// ~1/W of all lookups come here, and the common path must not pay registers or scratch for it.
constexpr uint32_t SKY_CORNER = 0xFFFFFFFFu;
__device__ __noinline__ uint32_t sky_tap_index(int layer, int x, int y, int W, int H) {
  const bool ox = x < 0 || x >= W, oy = y < 0 || y >= H;
  SkyTap t; t.layer = layer; t.x = x; t.y = y;
  if (W != H) { t.x = max(0, min(x, W - 1)); t.y = max(0, min(y, H - 1)); }   // not a cube: per-face clamp
  else if (ox && oy) return SKY_CORNER;
  else if (ox || oy) t = sky_neighbour(layer, x, y, W);
  return ((uint32_t)t.layer * (uint32_t)H + (uint32_t)t.y) * (uint32_t)W + (uint32_t)t.x;
}

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
  const int W = sc.sky_w, H = sc.sky_h;
  int x0 = (int)fu0, y0 = (int)fv0;
  x0 = max(-1, min(x0, W - 1)); y0 = max(-1, min(y0, H - 1));
  uint32_t i00 = ((uint32_t)layer * (uint32_t)H + (uint32_t)y0) * (uint32_t)W + (uint32_t)x0, i10 = i00 + 1u, i01 = i00 + (uint32_t)W, i11 = i01 + 1u;
  const bool edge = !(x0 >= 0 && x0 + 1 < W && y0 >= 0 && y0 + 1 < H);
  if (edge) {
    i00 = sky_tap_index(layer, x0, y0, W, H); i10 = sky_tap_index(layer, x0 + 1, y0, W, H);
    i01 = sky_tap_index(layer, x0, y0 + 1, W, H); i11 = sky_tap_index(layer, x0 + 1, y0 + 1, W, H);
  }
  auto tap = [&](uint32_t i) -> uchar4 {
    if (RT_NT_STREAMS & 4) { const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(sc.sky) + i); uchar4 r; r.x = w & 255u; r.y = (w >> 8) & 255u; r.z = (w >> 16) & 255u; r.w = w >> 24; return r; }
    return sc.sky[i];
  };
  const uchar4 a = tap(i00 == SKY_CORNER ? 0u : i00), b = tap(i10 == SKY_CORNER ? 0u : i10);
  const uchar4 c = tap(i01 == SKY_CORNER ? 0u : i01), d = tap(i11 == SKY_CORNER ? 0u : i11);
  float c00x = (float)a.x, c00y = (float)a.y, c00z = (float)a.z, c10x = (float)b.x, c10y = (float)b.y, c10z = (float)b.z;
  float c01x = (float)c.x, c01y = (float)c.y, c01z = (float)c.z, c11x = (float)d.x, c11y = (float)d.y, c11z = (float)d.z;
  if (edge) {
    // the tap beyond a cube corner = mean of the other three, summed in footprint order starting behind the corner tap
    // (taps are numbered 00, 10, 01, 11; oracle: ((t[k+1] + t[k+2]) + t[k+3]) / 3)
    if (i00 == SKY_CORNER) { c00x = ((c10x + c01x) + c11x) / 3.0f; c00y = ((c10y + c01y) + c11y) / 3.0f; c00z = ((c10z + c01z) + c11z) / 3.0f; }
    if (i10 == SKY_CORNER) { c10x = ((c01x + c11x) + c00x) / 3.0f; c10y = ((c01y + c11y) + c00y) / 3.0f; c10z = ((c01z + c11z) + c00z) / 3.0f; }
    if (i01 == SKY_CORNER) { c01x = ((c11x + c00x) + c10x) / 3.0f; c01y = ((c11y + c00y) + c10y) / 3.0f; c01z = ((c11z + c00z) + c10z) / 3.0f; }
    if (i11 == SKY_CORNER) { c11x = ((c00x + c10x) + c01x) / 3.0f; c11y = ((c00y + c10y) + c01y) / 3.0f; c11z = ((c00z + c10z) + c01z) / 3.0f; }
  }
  const float iu = 1.0f - wu, iv = 1.0f - wv;
  float ra = __builtin_fmaf(c10x, wu, c00x * iu), rb = __builtin_fmaf(c11x, wu, c01x * iu);
  float ga = __builtin_fmaf(c10y, wu, c00y * iu), gb = __builtin_fmaf(c11y, wu, c01y * iu);
  float ba = __builtin_fmaf(c10z, wu, c00z * iu), bb = __builtin_fmaf(c11z, wu, c01z * iu);
  return mk3(div255(__builtin_fmaf(rb, wv, ra * iv)), div255(__builtin_fmaf(gb, wv, ga * iv)), div255(__builtin_fmaf(bb, wv, ba * iv)));
}

// src/shader.rgen:185 imageStore: the pixel in the frame's format — RGBA32F (the shader's declared rgba32f), or the 8-bit view the
// reference's storage image really has (src/main.cpp:1899): clamp to [0,1], scale, round; R8G8B8A8 or the byte order of a B8G8R8A8 surface
__device__ __forceinline__ void store_pixel(const FrameDev& f, uint32_t p, float4 px) {
  if (f.out_rgba8) {
    auto q = [](float v) -> unsigned char { v = fminf(fmaxf(v, 0.0f), 1.0f); return (unsigned char)(v * 255.0f + 0.5f); };
    reinterpret_cast<uchar4*>(f.out)[p] = f.out_rgba8 == 2 ? make_uchar4(q(px.z), q(px.y), q(px.x), q(px.w)) : make_uchar4(q(px.x), q(px.y), q(px.z), q(px.w));
  } else {
    f.out[p] = px;
  }
}
constexpr float PIXEL_DONE = 2.0f;   // alpha of sample 0's colour slot when k_raygen has already resolved the pixel (all its samples missed)

// ------------------------------------------------------------------------------------------------
// k_raygen: one thread per (8x8 pixel tile, sample, lane).  src/shader.rgen:62-79, fused with the
// first step every traceRayEXT performs: the ray is tested against the boxes of the TLAS root.  A
// ray that enters none of them is a miss (src/shader.rmiss:11), so its sky colour
// (src/shader.rgen:90-94) is written here and it never touches a queue; the survivors are compacted
// into bounce queue 0 (shard blockIdx % 8) with a wavefront ballot.  On the headline frame ~80 % of
// the primary rays end here, which removes their ray/hit records from HBM traffic altogether.
// Primary-ray coverage mask.  All primary rays leave one point, so "which rays can touch a mesh at all" has a cheap conservative
// answer in SCREEN space: every frontier box of every instance (the child boxes of the BLAS nodes a few levels below the root,
// rt_api link_blas) is a convex body in front of the camera, its image is the convex hull of its eight projected corners, and
// a ray can only enter it through a pixel inside the bounding rectangle of those corners.  One thread per (instance, box) marks
// the 8x8-pixel tiles of that rectangle (plus one pixel of margin for the jitter and the rounding); a box with a corner at or
// behind the camera plane, or with a rectangle of more than COVER_MAX_TILES tiles, marks the whole frame.  k_raygen then shades
// the samples of unmarked tiles as misses (src/shader.rmiss:11) without touching the TLAS: on cfg3 the rays handed to the
// traversal kernel drop from 37 % of the samples (inside the instances' boxes) to about what really grazes the meshes.
__global__ __launch_bounds__(256) void k_cover(SceneDev sc, CoverViews views, uint32_t* mask_block) {
  // blockIdx.z = view: 0 the camera, 1..6 the faces of the cube around the light (entry lists of the shadow rays)
  const CoverArgs& a = views.v[blockIdx.z];
  uint32_t* const mask = mask_block + a.mask_offset;
  const InstanceDev* I = sc.inst + a.inst_base + blockIdx.y;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= I->cover_count || (I->mask & 0xFFu) == 0u) return;
  const float* bx = sc.cover_boxes + 6u * (size_t)(I->cover_first + b);
  float x0 = 3e38f, x1 = -3e38f, y0 = 3e38f, y1 = -3e38f, az_min = 3e38f;
  bool all = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const F3 p = mk3(bx[(k & 1) ? 3 : 0], bx[(k & 2) ? 4 : 1], bx[(k & 4) ? 5 : 2]);
    const F3 w = xform_point(I->o2w, p);
    const F3 v = mk3(w.x - a.cam[0], w.y - a.cam[1], w.z - a.cam[2]);
    const float ax = a.inv[0] * v.x + a.inv[1] * v.y + a.inv[2] * v.z;
    const float ay = a.inv[3] * v.x + a.inv[4] * v.y + a.inv[5] * v.z;
    const float az = a.inv[6] * v.x + a.inv[7] * v.y + a.inv[8] * v.z;
    if (!(az > 1e-20f)) { all = true; continue; }
    az_min = fminf(az_min, az);
    const float ux = a.kf * ax / az, uy = a.kf * ay / az;
    const float px = (ux + 1.0f) * 0.5f * (float)a.width, py = (1.0f - uy) * 0.5f * (float)a.height;
    if (!(__builtin_fabsf(px) < 1e9f) || !(__builtin_fabsf(py) < 1e9f)) { all = true; continue; }
    x0 = fminf(x0, px); x1 = fmaxf(x1, px); y0 = fminf(y0, py); y1 = fmaxf(y1, py);
  }
  if (!all) {
    // one pixel of margin on every side, and a relative one for the rounding of the projection
    float mx = 1.0f + 1e-4f * (__builtin_fabsf(x0) + __builtin_fabsf(x1)), my = 1.0f + 1e-4f * (__builtin_fabsf(y0) + __builtin_fabsf(y1));
    if (a.apex_radius > 0.0f) {
      // rays that pass within apex_radius of the view point instead of through it (shadow rays at the light): a point of the box
      // at depth az is then seen up to apex_radius (1 + |u|) / (az - apex_radius) away in u, u = kf ax / az (orthonormal light bases)
      if (!(az_min > 4.0f * a.apex_radius)) all = true;
      else {
        const float g = a.apex_radius / (az_min - a.apex_radius) * 1.01f;
        const float uxm = fmaxf(__builtin_fabsf(2.0f * x0 / (float)a.width - 1.0f), __builtin_fabsf(2.0f * x1 / (float)a.width - 1.0f));
        const float uym = fmaxf(__builtin_fabsf(2.0f * y0 / (float)a.height - 1.0f), __builtin_fabsf(2.0f * y1 / (float)a.height - 1.0f));
        mx += g * (a.kf + uxm) * 0.5f * (float)a.width; my += g * (a.kf + uym) * 0.5f * (float)a.height;
      }
    }
    const int ix0 = (int)floorf(x0 - mx), ix1 = (int)floorf(x1 + mx), iy0 = (int)floorf(y0 - my), iy1 = (int)floorf(y1 + my);
    if (ix1 < 0 || iy1 < 0 || ix0 >= a.width || iy0 >= a.height) return;   // the box is off screen
    const int tx0 = max(ix0, 0) >> 3, tx1 = min(ix1, a.width - 1) >> 3, ty0 = max(iy0, 0) >> 3, ty1 = min(iy1, a.height - 1) >> 3;
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > COVER_MAX_TILES) all = true;
    else {
      // the tiles of one row are consecutive bits: one OR per mask word the span touches.  The results are not used, so the
      // atomics are issued back to back without waiting for any of them.
      for (int ty = ty0; ty <= ty1; ty++) {
        const uint32_t t0 = (uint32_t)(ty * a.tiles_x + tx0), t1 = (uint32_t)(ty * a.tiles_x + tx1);
        for (uint32_t w = t0 >> 5; w <= (t1 >> 5); w++) {
          const uint32_t lo = w == (t0 >> 5) ? (t0 & 31u) : 0u, hi = w == (t1 >> 5) ? (t1 & 31u) : 31u;
          const uint32_t bits = (hi == 31u ? 0xFFFFFFFFu : ((1u << (hi + 1u)) - 1u)) & ~((1u << lo) - 1u);
          __hip_atomic_fetch_or(mask + 1u + w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (all) atomicOr(mask, 1u);
}

constexpr int REF_DONE = (int)0x80000000;   // bottom-of-stack sentinel: the ray is finished
constexpr int REF_MARK = (int)0x80000001;   // "leave the instance" marker (both negative: not interior)

// ------------------------------------------------------------------------------------------------
// k_entry: entry-point search for the primary rays of one 8x8-pixel tile (all rays of a tile leave one point through one small
// rectangle of directions, so what they can hit is decided ONCE per tile instead of once per ray).  One lane per tile walks the
// TLAS, then the BLAS of the nearest instance, with the tile's BEAM — the cone {c > 0, ux0 c <= k a <= ux1 c, uy0 c <= k b <= uy1 c}
// in camera coordinates (a, b, c) of src/shader.rgen:74-79 — instead of a ray: a child box that lies entirely on the outer side
// of one of the beam's five planes cannot be entered by any ray of the tile and is dropped; a node with one surviving child is
// replaced by that child; a node with two is split while the tile's list has room.  What is left is a handful of subtrees deep
// in the tree (ENTRY_WORDS + 1 words) that together contain everything a ray of the tile can hit: the closest-hit kernel starts
// the tile's rays there (trace_body<..., ENTRY>) instead of at the TLAS root.  The rule is conservative in the same sense as the
// box tests of the traversal itself (quantized boxes contain the float boxes; every comparison carries a slack that covers the
// binary32 evaluation), and the closest hit is the minimum over all candidates with a fixed tie rule, so frames are identical
// with and without it (tested) — it only removes the ~11 upper levels of the walk from every primary ray.
//
// Record of a tile (EntryRec, 32 bytes): w[0] = number of stack words | instance << 8 (ENTRY_NO_INST: the rays start in world
// space), w[1] = first node to visit, w[2..] = the words to push under it, bottom first: [TLAS words of the other instances /
// unopened TLAS nodes, far to near] [REF_MARK] [BLAS nodes of the instance, far to near].  w[0] = ENTRY_EMPTY: the beam touches
// nothing — k_raygen shades the tile's samples as misses.
struct Plane5 {
  float nx[5], ny[5], nz[5], d[5], slack[5];   // L_j(X) = nx X.x + ny X.y + nz X.z + d on QUANTIZED coordinates X
  F3 qs, qb; uint3 rot;                        // the tile's CENTRE ray in the same space (slab_q form): orders the entries
};

// planes of the beam in world space: v = P - apex, (a, b, c) = inv * v
__device__ __forceinline__ void beam_world(const EntryArgs& e, float ux0, float ux1, float uy0, float uy1, float wn[5][3], float wd[5], float wmag[5]) {
  const float* ia = e.inv; const float* ib = e.inv + 3; const float* ic = e.inv + 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    wn[0][k] = e.kf * ia[k] - ux0 * ic[k];
    wn[1][k] = ux1 * ic[k] - e.kf * ia[k];
    wn[2][k] = e.kf * ib[k] - uy0 * ic[k];
    wn[3][k] = uy1 * ic[k] - e.kf * ib[k];
    wn[4][k] = ic[k];
  }
#pragma unroll
  for (int j = 0; j < 5; j++) {
    wd[j] = -(wn[j][0] * e.cam[0] + wn[j][1] * e.cam[1] + wn[j][2] * e.cam[2]);
    wmag[j] = __builtin_fabsf(wn[j][0] * e.cam[0]) + __builtin_fabsf(wn[j][1] * e.cam[1]) + __builtin_fabsf(wn[j][2] * e.cam[2]);   // what wd[j] was summed from
  }
}
// the same planes on the quantized coordinates of a tree: P = m * (q_lo + X * q_scale) + t  (m = NULL: world space, P = q_lo + X q_scale)
__device__ __forceinline__ void beam_fold(const float wn[5][3], const float wd[5], const float wmag[5], float apex_radius, const float* m, const float* q_lo, const float* q_scale, Plane5& P) {
#pragma unroll
  for (int j = 0; j < 5; j++) {
    float n0 = wn[j][0], n1 = wn[j][1], n2 = wn[j][2], d = wd[j], mag = wmag[j];
    if (m) {   // n_obj = M^T n, d_obj = d + n . t
      const float a0 = wn[j][0] * m[0] + wn[j][1] * m[4] + wn[j][2] * m[8];
      const float a1 = wn[j][0] * m[1] + wn[j][1] * m[5] + wn[j][2] * m[9];
      const float a2 = wn[j][0] * m[2] + wn[j][1] * m[6] + wn[j][2] * m[10];
      d = wd[j] + (wn[j][0] * m[3] + wn[j][1] * m[7] + wn[j][2] * m[11]);
      mag += __builtin_fabsf(wn[j][0] * m[3]) + __builtin_fabsf(wn[j][1] * m[7]) + __builtin_fabsf(wn[j][2] * m[11]);
      n0 = a0; n1 = a1; n2 = a2;
    }
    const float t0 = n0 * q_lo[0], t1 = n1 * q_lo[1], t2 = n2 * q_lo[2];
    P.nx[j] = n0 * q_scale[0]; P.ny[j] = n1 * q_scale[1]; P.nz[j] = n2 * q_scale[2];
    P.d[j] = d + (t0 + t1 + t2);
    mag += __builtin_fabsf(t0) + __builtin_fabsf(t1) + __builtin_fabsf(t2) + 65535.0f * (__builtin_fabsf(P.nx[j]) + __builtin_fabsf(P.ny[j]) + __builtin_fabsf(P.nz[j]));
    // >> the rounding of every binary32 step above and of the evaluation below (a few 1e-7 of mag); plus, for rays that only pass
    // within apex_radius of the view point instead of through it, every plane moved outward by that distance
    P.slack[j] = 2e-5f * mag + apex_radius * __builtin_sqrtf(wn[j][0] * wn[j][0] + wn[j][1] * wn[j][1] + wn[j][2] * wn[j][2]) * 1.0001f;
  }
}
// can a ray of the beam enter the quantized box (wx, wy, wz)?  key = where the tile's centre ray passes the last of the box's
// near planes (its entry distance when it hits the box): the order of the entries, the same measure the traversal orders by
__device__ __forceinline__ bool beam_box(const Plane5& P, uint32_t wx, uint32_t wy, uint32_t wz, float& key) {
  if ((wx & 0xFFFFu) > (wx >> 16)) return false;   // the absent child of a synthetic single-child root
  {
    const uint32_t rx = __builtin_amdgcn_alignbit(wx, wx, P.rot.x), ry = __builtin_amdgcn_alignbit(wy, wy, P.rot.y), rz = __builtin_amdgcn_alignbit(wz, wz, P.rot.z);
    key = fmaxf(fmaxf(__builtin_fmaf((float)(rx & 0xFFFFu), P.qs.x, P.qb.x), __builtin_fmaf((float)(ry & 0xFFFFu), P.qs.y, P.qb.y)),
                fmaxf(__builtin_fmaf((float)(rz & 0xFFFFu), P.qs.z, P.qb.z), 0.0f));
  }
  const float lx = (float)(wx & 0xFFFFu), hx = (float)(wx >> 16), ly = (float)(wy & 0xFFFFu), hy = (float)(wy >> 16), lz = (float)(wz & 0xFFFFu), hz = (float)(wz >> 16);
  bool in = true;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const float mx = P.nx[j] * (P.nx[j] >= 0.0f ? hx : lx) + P.ny[j] * (P.ny[j] >= 0.0f ? hy : ly) + P.nz[j] * (P.nz[j] >= 0.0f ? hz : lz) + P.d[j];
    in = in && (mx >= -P.slack[j]);
  }
  return in;
}

constexpr int ENTRY_TLAS_CAP = 4;       // TLAS words kept per tile (instances / unopened TLAS nodes)
constexpr int ENTRY_MAX_ROUNDS = 72;    // levels one tree can be opened (BLAS depth <= 40, TLAS <= 20; whatever is open then is a valid entry set)
constexpr int ENTRY_FREE = 0x7FFFFFFF;  // an unused slot of a tile's list

// EIGHT LANES PER TILE, one per slot of the tile's list, level-synchronous: in every round each lane that holds an unvisited
// interior node fetches it and tests both children against the beam (the lanes of a tile work in parallel, so a round costs one
// node fetch whatever the length of the list: the search is as deep as the tree, ~15 dependent fetches, instead of one per node
// opened).  No child hit: the slot becomes free.  One: the slot descends.  Two: the lane asks for a free slot of its tile —
// requests are ranked (the ones waiting longest, i.e. the shallowest nodes, first) against the free slots with two ballots, the
// second child travels through LDS; a request that finds no room keeps its (unsplit) node, which stays a valid entry.
// Returns when no lane of the wave has anything left to open.  TLAS: a leaf whose instance the ray mask cannot see is dropped.
template <bool TLAS>
__device__ __forceinline__ void entry_open(const SceneDev& sc, const Plane5& P, int& w, float& key, uint32_t cap, int (*s_xw)[8], float (*s_xk)[8]) {
  const char* const node_bytes = reinterpret_cast<const char*>(sc.blas_nodes);
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, grp = lane >> 3, gsh = lane & 56u;
  const uint32_t below = (1u << sub) - 1u;
  bool pend = false, visited = false;   // pend: the node splits (both children stashed) and waits for a free slot
  int c0 = 0, c1 = 0; float k0 = 0.f, k1 = 0.f;
  for (int round = 0; round < ENTRY_MAX_ROUNDS; round++) {
    const bool act = w >= 0 && w != ENTRY_FREE && !visited;
    const bool old_req = pend;
    if (act) {
      const uint4* np = reinterpret_cast<const uint4*>(node_bytes + ((uint32_t)w << 5));
      const uint4 Q0 = np[0], Q1 = np[1];
      bool h0 = beam_box(P, Q0.x, Q0.y, Q0.z, k0), h1 = beam_box(P, Q0.w, Q1.x, Q1.y, k1);
      c0 = (int)Q1.z; c1 = (int)Q1.w;
      if (TLAS) {
        if (h0 && c0 < 0 && (sc.inst[~c0].mask & 0xFFu) == 0u) h0 = false;
        if (h1 && c1 < 0 && (sc.inst[~c1].mask & 0xFFu) == 0u) h1 = false;
      }
      if (h0 && h1) { pend = true; visited = true; }
      else if (h0 || h1) { w = h0 ? c0 : c1; key = h0 ? k0 : k1; }   // one child: descend (the new node is unvisited)
      else w = ENTRY_FREE;
    }
    // ---- free slots against split requests, per tile
    const uint32_t free_m = (uint32_t)(__ballot(w == ENTRY_FREE && sub < cap) >> gsh) & 0xFFu;
    const uint32_t old_m = (uint32_t)(__ballot(old_req) >> gsh) & 0xFFu, new_m = (uint32_t)(__ballot(pend && !old_req) >> gsh) & 0xFFu;
    const uint32_t n_free = (uint32_t)__builtin_popcount(free_m);
    const uint32_t my_req = old_req ? (uint32_t)__builtin_popcount(old_m & below) : (uint32_t)__builtin_popcount(old_m) + (uint32_t)__builtin_popcount(new_m & below);
    const bool granted = pend && my_req < n_free;
    const uint32_t n_granted = min(n_free, (uint32_t)__builtin_popcount(old_m | new_m));
    if (granted) { s_xw[grp][my_req] = c1; s_xk[grp][my_req] = k1; w = c0; key = k0; pend = false; visited = false; }
    __syncthreads();
    const uint32_t my_free = (uint32_t)__builtin_popcount(free_m & below);
    if (w == ENTRY_FREE && sub < cap && my_free < n_granted) { w = s_xw[grp][my_free]; key = s_xk[grp][my_free]; visited = false; }
    __syncthreads();
    if (__ballot((w >= 0 && w != ENTRY_FREE && !visited)) == 0ull) break;   // nothing left to open in this wave (grants make unvisited words)
  }
}

// one wave = 8 tiles x 8 list slots
__global__ __launch_bounds__(64) void k_entry(SceneDev sc, EntryViews views) {
  __shared__ int s_xw[8][8];  __shared__ float s_xk[8][8];   // split exchange / all-to-all of a tile's list
  const EntryArgs& e = views.v[blockIdx.y];                  // view: 0 the camera, 1..6 the faces of the cube around the light
  const uint32_t lane = threadIdx.x, sub = lane & 7u, grp = lane >> 3;
  const uint32_t t = blockIdx.x * 8u + grp;
  const uint32_t n_tiles = (uint32_t)(e.tiles_x * e.tile_rows);
  bool work = t < n_tiles;
  const uint32_t tc = work ? t : 0u;
  const uint32_t tx = tc % (uint32_t)e.tiles_x, lty = tc / (uint32_t)e.tiles_x;
  uint32_t fty = lty;   // tile row of the full frame (bands are whole tiles when entry lists are on)
  if (e.n_shards != 1) {
    const uint32_t tiles_per_band = (uint32_t)e.band_rows >> 3;
    const uint32_t band = lty / tiles_per_band, sb = lty - band * tiles_per_band;
    fty = (band * (uint32_t)e.n_shards + (uint32_t)e.shard) * tiles_per_band + sb;
  }
  EntryRec* const rec = e.records + tc;
  if (work && e.cover != nullptr && e.cover[0] == 0u) {
    const uint32_t ct = fty * (uint32_t)e.cover_tiles_x + tx;
    if (((e.cover[1u + (ct >> 5)] >> (ct & 31u)) & 1u) == 0u) {   // no mesh projects onto this tile
      work = false;
      if (sub == 0u) rec->w[0] = (int)ENTRY_EMPTY;
    }
  }
  if (__ballot(work) == 0ull) return;   // (wave-uniform: the barriers below are reached by all 64 lanes or by none)
  // the tile's rectangle of (ux, uy), src/shader.rgen:74-75, with a quarter of a pixel of margin (jitter in [0, 1] after
  // rounding, binary32 evaluation of ux/uy and of the normalised direction: all far below it)
  const float m = 0.25f;
  const float ux0 = 2.0f * ((float)(8u * tx) - m) / (float)e.width - 1.0f, ux1 = 2.0f * ((float)(8u * tx + 8u) + m) / (float)e.width - 1.0f;
  const float uy1 = 1.0f - 2.0f * ((float)(8u * fty) - m) / (float)e.height, uy0 = 1.0f - 2.0f * ((float)(8u * fty + 8u) + m) / (float)e.height;
  float wn[5][3], wd[5], wmag[5];
  beam_world(e, ux0, ux1, uy0, uy1, wn, wd, wmag);
  Plane5 P;
  beam_fold(wn, wd, wmag, e.apex_radius, nullptr, sc.tlas_q_lo, sc.tlas_q_scale, P);
  // the tile's centre ray (direction not normalised: the keys of one tile only have to be comparable with one another)
  const float uxc = 0.5f * (ux0 + ux1), uyc = 0.5f * (uy0 + uy1);
  const F3 cam = mk3(e.cam[0], e.cam[1], e.cam[2]);
  const F3 dc = mk3(uxc * e.basis[0] + uyc * e.basis[3] + e.kf * e.basis[6], uxc * e.basis[1] + uyc * e.basis[4] + e.kf * e.basis[7], uxc * e.basis[2] + uyc * e.basis[5] + e.kf * e.basis[8]);
  quant_space(cam, dc, sc.tlas_q_lo, sc.tlas_q_scale, P.qs, P.qb, P.rot);
  // ---- the TLAS: instances (and, beyond ENTRY_TLAS_CAP, unopened TLAS nodes) the beam can touch
  int w = (work && sub == 0u) ? sc.tlas_root + e.tlas_root_offset : ENTRY_FREE;
  float key = 0.0f;
  entry_open<true>(sc, P, w, key, ENTRY_TLAS_CAP, s_xw, s_xk);
  // every lane of the tile reads the whole TLAS list and orders it (near first) the same way
  if (sub < (uint32_t)ENTRY_TLAS_CAP) { s_xw[grp][sub] = w; s_xk[grp][sub] = key; }
  __syncthreads();
  int tw[ENTRY_TLAS_CAP]; float tk[ENTRY_TLAS_CAP];
#pragma unroll
  for (int i = 0; i < ENTRY_TLAS_CAP; i++) { tw[i] = s_xw[grp][i]; tk[i] = tw[i] == ENTRY_FREE ? 3.0e38f : s_xk[grp][i]; }
  __syncthreads();
#define RT_CSWAP(i_, j_) { const bool sw_ = tk[j_] < tk[i_]; const int a_ = sw_ ? tw[j_] : tw[i_], b_ = sw_ ? tw[i_] : tw[j_]; const float ka_ = sw_ ? tk[j_] : tk[i_], kb_ = sw_ ? tk[i_] : tk[j_]; tw[i_] = a_; tw[j_] = b_; tk[i_] = ka_; tk[j_] = kb_; }
  RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
  int nt = 0;
#pragma unroll
  for (int i = 0; i < ENTRY_TLAS_CAP; i++) nt += tw[i] != ENTRY_FREE ? 1 : 0;
  // the nearest instance (TLAS leaf) of the list is opened further; the other words stay as they are
  int ia = -1;
#pragma unroll
  for (int i = ENTRY_TLAS_CAP - 1; i >= 0; i--) if (i < nt && tw[i] < 0) ia = i;
  uint32_t inst = ENTRY_NO_INST;
  int rest[ENTRY_TLAS_CAP];   // the other words, near first
  int n_rest = 0;
#pragma unroll
  for (int i = 0; i < ENTRY_TLAS_CAP; i++) rest[i] = ENTRY_FREE;
  {
    int o = 0;
#pragma unroll
    for (int i = 0; i < ENTRY_TLAS_CAP; i++)
      if (i < nt && i != ia) {
#pragma unroll
        for (int j = 0; j < ENTRY_TLAS_CAP; j++) if (j == o) rest[j] = tw[i];
        o++;
      }
    n_rest = o;
  }
  // ---- the BLAS of that instance
  w = ENTRY_FREE; key = 0.0f;
  uint32_t cap_b = 0;
  if (work && ia >= 0) {
    int wa = ENTRY_FREE;
#pragma unroll
    for (int i = 0; i < ENTRY_TLAS_CAP; i++) if (i == ia) wa = tw[i];
    inst = (uint32_t)(~wa);
    const InstanceDev* I = sc.inst + inst;
    float o2w[12];
#pragma unroll
    for (int k = 0; k < 12; k++) o2w[k] = I->o2w[k];
    beam_fold(wn, wd, wmag, e.apex_radius, o2w, I->q_lo, I->q_scale, P);
    quant_space(xform_point(I->w2o, cam), xform_vec(I->w2o, dc), I->q_lo, I->q_scale, P.qs, P.qb, P.rot);
    if (sub == 0u) w = I->blas_root;
    cap_b = (uint32_t)(ENTRY_WORDS - n_rest);   // stack words: n_rest + REF_MARK + (nb - 1) <= ENTRY_WORDS
  }
  entry_open<false>(sc, P, w, key, cap_b, s_xw, s_xk);
  // rank of this lane's word among the tile's BLAS words (near first; ties by slot)
  s_xw[grp][sub] = w; s_xk[grp][sub] = key;
  __syncthreads();
  int nb = 0, rank = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int wi = s_xw[grp][i]; const float ki = s_xk[grp][i];
    if (wi != ENTRY_FREE) { nb++; if (w != ENTRY_FREE && (ki < key || (ki == key && (uint32_t)i < sub))) rank++; }
  }
  if (!work) return;
  if (nb == 0) inst = ENTRY_NO_INST;
  if (nb == 0 && n_rest == 0) { if (sub == 0u) rec->w[0] = (int)ENTRY_EMPTY; return; }
  // record: w[0] header, w[1] first node, w[2..] stack words bottom first: [rest far -> near] [REF_MARK] [BLAS far -> near]
  if (nb > 0) {
    if (w != ENTRY_FREE) rec->w[rank == 0 ? 1 : 2 + n_rest + 1 + (nb - 1 - rank)] = w;
    if (sub == 0u) {
#pragma unroll
      for (int i = 0; i < ENTRY_TLAS_CAP; i++) if (i < n_rest) rec->w[2 + (n_rest - 1 - i)] = rest[i];
      rec->w[2 + n_rest] = REF_MARK;
      rec->w[0] = (int)((uint32_t)(n_rest + nb) | ((uint32_t)(n_rest + 1) << 4) | (inst << 8));
    }
  } else if (sub == 0u) {
    rec->w[1] = rest[0];
#pragma unroll
    for (int i = 1; i < ENTRY_TLAS_CAP; i++) if (i < n_rest) rec->w[2 + (n_rest - 1 - i)] = rest[i];
    rec->w[0] = (int)((uint32_t)(n_rest - 1) | ((uint32_t)(n_rest - 1) << 4) | (ENTRY_NO_INST << 8));
  }
}

// src/shader.rgen:62-75: where sample i of pixel (x, y) of a W x H frame crosses the image plane, (ux, uy) in [-1, 1].  A function of the
// pixel, the sample index and the frame size only — neither camera nor scene enter — so it is computed ONCE per (W, H, spp, shard
// layout) into a table (k_jitter_table; SURVEY.md §8(c)(2) proposes exactly this) and k_raygen reads 8 bytes per sample instead of
// evaluating two binary64 sines and two IEEE divisions per sample and frame.  Same function, same bits.
__device__ __forceinline__ float2 sample_uv(uint32_t x, uint32_t y, uint32_t i, uint32_t spp, int width, int height) {
  const float fx = (float)x, fy = (float)y;
  const float seed0 = (float)(spp + i), seed1 = seed0 + 0.5f;
  float ux = (fx + jitter_hash(fx, fy, seed0)) / (float)width;
  float uy = (fy + jitter_hash(fx, fy, seed1)) / (float)height;
  ux = __builtin_fmaf(ux, 2.0f, -1.0f);
  uy = -__builtin_fmaf(uy, 2.0f, -1.0f);
  return make_float2(ux, uy);
}
// local row of a shard's compact image -> row of the full frame (interleaved bands)
__device__ __forceinline__ uint32_t frame_row(const FrameDev& f, uint32_t ly) {
  if (f.n_shards == 1) return ly;
  const uint32_t band = ly / (uint32_t)f.band_rows, within = ly - band * (uint32_t)f.band_rows;
  return (band * (uint32_t)f.n_shards + (uint32_t)f.shard) * (uint32_t)f.band_rows + within;
}
// table entry of (tile, sample, lane): tile-major, so that a wavefront of k_raygen reads 512 consecutive bytes
__device__ __forceinline__ size_t jitter_index(uint32_t tile, uint32_t i, uint32_t spp, uint32_t lane) { return ((size_t)tile * spp + i) * 64u + lane; }

// same grid as k_raygen
__global__ __launch_bounds__(256) void k_jitter_table(FrameDev f, uint32_t spp, float2* table) {
  const uint32_t lane = threadIdx.x;
  const uint32_t i = blockIdx.z * blockDim.y + threadIdx.y;
  const uint32_t x = blockIdx.x * 8u + (lane & 7u), ly = blockIdx.y * 8u + (lane >> 3);
  if (i >= spp) return;
  float2 uv = make_float2(0.f, 0.f);
  if (x < (uint32_t)f.width && ly < (uint32_t)f.rows) uv = sample_uv(x, frame_row(f, ly), i, spp, f.width, f.height);
  table[jitter_index(blockIdx.y * gridDim.x + blockIdx.x, i, spp, lane)] = uv;
}

__global__ __launch_bounds__(256) void k_raygen(SceneDev sc, FrameDev f, UniformsDev u, BatchTab bt) {
  // grid (tiles_x, tiles_y [x frames of a batch], sample groups), block (64 lanes = one 8x8 tile, up to 4 samples): no index division
  const uint32_t spp = u.samples_per_pixel;
  const uint32_t lane = threadIdx.x;
  const uint32_t i = blockIdx.z * blockDim.y + threadIdx.y;
  const uint32_t x = blockIdx.x * 8u + (lane & 7u);
  // frame batch: the frames' tile rows are stacked in grid.y; fi = this workgroup's frame, by = its tile row in that frame (uniform)
  uint32_t fi = 0, by = blockIdx.y;
  if (f.batch_k > 1) { const uint32_t tile_rows1 = ((uint32_t)f.rows + 7u) >> 3; fi = blockIdx.y / tile_rows1; by = blockIdx.y - fi * tile_rows1; }
  const uint32_t npx1 = (uint32_t)(f.rows * f.width);
  const uint32_t ly = by * 8u + (lane >> 3);
  const bool live = i < spp && x < (uint32_t)f.width && ly < (uint32_t)f.rows;
  // coverage mask (uniform per workgroup): can any mesh touch this tile?  The local tile row maps to a tile row of the full
  // frame because bands are whole tiles when the mask is on (rt_api enables it only for band heights that are multiples of 8).
  bool covered = true;
  const uint32_t* const cover = f.cover != nullptr ? f.cover + (size_t)fi * f.cover_view_words : nullptr;
  if (cover != nullptr && cover[0] == 0u) {
    uint32_t fty = by;
    if (f.n_shards != 1) {
      const uint32_t tiles_per_band = (uint32_t)f.band_rows >> 3;
      const uint32_t band = by / tiles_per_band, sub = by - band * tiles_per_band;
      fty = (band * (uint32_t)f.n_shards + (uint32_t)f.shard) * tiles_per_band + sub;
    }
    const uint32_t t = fty * (uint32_t)f.cover_tiles_x + blockIdx.x;
    covered = ((cover[1u + (t >> 5)] >> (t & 31u)) & 1u) != 0u;
  }
  // entry lists (k_entry): the record of this tile says whether its beam touches anything at all
  const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;   // (over all frames of a batch: the records of frame k follow those of frame k - 1)
  const uint32_t tile1 = by * gridDim.x + blockIdx.x;          // within its frame
  if (f.entry != nullptr && covered && (uint32_t)f.entry[tile].w[0] == ENTRY_EMPTY) covered = false;
  // tile blobs (k_blob): the rays of a tile that has one are walked in LDS by k_trace_tile
  if (f.tile_blob != nullptr && covered && f.tile_blob[tile] != BLOB_NONE) return;   // (uniform over the workgroup) k_tile generates and walks this tile's rays
  bool survive = false;
  F3 d = mk3(0.f, 0.f, 1.f);
  uint32_t sid = 0;
  float4 miss_col = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    // (the table of this frame size and shard layout, when the host has one: rt_api jitter tables)
    const float2 uv = f.jitter != nullptr ? f.jitter[jitter_index(tile1, i, spp, lane)] : sample_uv(x, frame_row(f, ly), i, spp, f.width, f.height);
    const float ux = uv.x, uy = uv.y;
    F3 right = mk3(u.right[0], u.right[1], u.right[2]), up = mk3(u.up[0], u.up[1], u.up[2]), fwd = mk3(u.forward[0], u.forward[1], u.forward[2]);
    if (f.batch_k > 1) { right = mk3(bt.right[fi][0], bt.right[fi][1], bt.right[fi][2]); up = mk3(bt.up[fi][0], bt.up[fi][1], bt.up[fi][2]); fwd = mk3(bt.forward[fi][0], bt.forward[fi][1], bt.forward[fi][2]); }
    d = normalize3(fma3(2.5f, fwd, fma3(uy, up, mul3(right, ux))));
    sid = (fi * spp + i) * npx1 + ly * (uint32_t)f.width + x;
    if (covered) {
      const F3 o = f.batch_k > 1 ? mk3(bt.position[fi][0], bt.position[fi][1], bt.position[fi][2]) : mk3(u.position[0], u.position[1], u.position[2]);
      F3 qs, qb; uint3 rot;
      quant_space(o, d, sc.tlas_q_lo, sc.tlas_q_scale, qs, qb, rot);
      const bool far = f.far_possible != 0 && quant_far(qs, qb);   // a camera hundreds of TLAS extents from the scene: the TLAS does not cull (see quant_far)
      // two levels of the TLAS: the boxes of the root and, where a child of the root is interior, of its children
      const uint4* rp = reinterpret_cast<const uint4*>(sc.blas_nodes + sc.tlas_root + (int)fi * sc.tlas_stride);
      const uint4 Q0 = rp[0], Q1 = rp[1];
      float tn;
      const bool h0 = far || slab_q(Q0.x, Q0.y, Q0.z, qs, qb, rot, 0.001f, 10000.0f, tn);
      const bool h1 = (far || slab_q(Q0.w, Q1.x, Q1.y, qs, qb, rot, 0.001f, 10000.0f, tn)) && Q1.w != Q1.z;
      const int c0 = (int)Q1.z, c1 = (int)Q1.w;
      survive = (h0 && c0 < 0) || (h1 && c1 < 0);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int ch = k ? c1 : c0;
        if ((k ? h1 : h0) && ch >= 0 && !survive) {
          const uint4* np = reinterpret_cast<const uint4*>(sc.blas_nodes + ch);
          const uint4 N0 = np[0], N1 = np[1];
          survive = far || slab_q(N0.x, N0.y, N0.z, qs, qb, rot, 0.001f, 10000.0f, tn) || slab_q(N0.w, N1.x, N1.y, qs, qb, rot, 0.001f, 10000.0f, tn);
        }
      }
    }
    if (!survive) {
      const F3 c = sample_sky(sc, mk3(d.x, d.y, -d.z));
      miss_col = make_float4(c.x, c.y, c.z, 1.0f);
    }
  }
  // A pixel ALL of whose samples are misses is resolved right here (src/shader.rgen:180-185: the same ordered sum and division as
  // k_resolve) when the workgroup holds all of them (spp <= 4): 4/5 of the headline's pixels are sky, and for them the per-sample
  // colours never travel to HBM and back — one pixel and one marker are stored instead of four colours, k_resolve reads the marker only.
  const bool missed = live && !survive;
  const bool fuse = gridDim.z == 1u;
  __shared__ float4 s_col[4][64];
  __shared__ unsigned long long s_miss[4];
  if (fuse && !covered) {
    // a tile no mesh can touch (uniform over the workgroup; 4/5 of the headline's tiles): every sample is a miss, so there is no queue
    // run to allocate and no vote to take — exchange the colours, sum, store, done
    s_col[threadIdx.y][lane] = miss_col;
    __syncthreads();
    if (threadIdx.y == 0 && live) {
      float r = 0.f, g = 0.f, b = 0.f, al = 0.f;
      for (uint32_t w = 0; w < blockDim.y; w++) { const float4 c = s_col[w][lane]; r += c.x; g += c.y; b += c.z; al += c.w; }
      const float nn = (float)spp;
      const uint32_t p = ly * (uint32_t)f.width + x;
      store_pixel(f, fi * f.out_frame_stride + p, make_float4(r / nn, g / nn, b / nn, al / nn));
      st_stream(&f.sample_color[fi * spp * npx1 + p], make_float4(0.f, 0.f, 0.f, PIXEL_DONE));
    }
    return;
  }
  if (fuse) {
    s_col[threadIdx.y][lane] = miss_col;
    const uint64_t mm = __ballot(missed);
    if (lane == 0) s_miss[threadIdx.y] = mm;
  } else if (missed) st_stream(&f.sample_color[sid], miss_col);
  // workgroups are handed to the XCDs round-robin in linear order
  const uint32_t shard = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & (N_SHARDS - 1);
  // ONE allocation per workgroup: the survivors of a tile's (up to four) samples form one run of the queue, so the 64-ray chunks
  // the traversal kernels pull hold rays of one tile (two where a run ends) — what the packet kernel's coherence rests on
  __shared__ uint32_t s_run[5];
  const uint64_t smask = __ballot(survive);
  if (lane == 0) s_run[threadIdx.y] = (uint32_t)__builtin_popcountll(smask);
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < blockDim.y; w++) { const uint32_t c = s_run[w]; s_run[w] = tot; tot += c; }
    if (f.pixel_runs && tot) {
      // pixel runs (kernels_beam.inc): the tile's samples keep their places — 64 slots per sample row, traced or not
      const uint32_t run = 64u * blockDim.y;
      __hip_atomic_fetch_add(f.counters + cnt_tail(Q_DEAD, (int)shard), run - tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tot = run;
    }
    s_run[4] = tot ? atomicAdd(f.counters + cnt_tail(0, (int)shard), tot) : 0u;
    s_run[0] = f.pixel_runs ? tot : s_run[0];   // (pixel runs: row 0's offset is not needed; the word says whether the tile has a run)
  }
  __syncthreads();
  const bool run_tile = f.pixel_runs && s_run[0] != 0u;
  const uint32_t slot = f.pixel_runs ? s_run[4] + 64u * threadIdx.y + lane : s_run[4] + s_run[threadIdx.y] + prefix_rank(smask);
  if (fuse) {   // (s_col / s_miss were written before the first barrier above)
    bool all = true;
    for (uint32_t w = 0; w < blockDim.y; w++) all = all && ((s_miss[w] >> lane) & 1ull) != 0ull;
    if (all) {
      if (threadIdx.y == 0) {
        float r = 0.f, g = 0.f, b = 0.f, al = 0.f;
        for (uint32_t w = 0; w < blockDim.y; w++) { const float4 c = s_col[w][lane]; r += c.x; g += c.y; b += c.z; al += c.w; }
        const float nn = (float)spp;
        const uint32_t p = ly * (uint32_t)f.width + x;     // = the sample id of sample 0 (in its frame)
        store_pixel(f, fi * f.out_frame_stride + p, make_float4(r / nn, g / nn, b / nn, al / nn));
        st_stream(&f.sample_color[fi * spp * npx1 + p], make_float4(0.f, 0.f, 0.f, PIXEL_DONE));
      }
    } else if (missed) st_stream(&f.sample_color[sid], miss_col);
  }
  if (survive || run_tile) {
    const uint32_t v = shard * f.shard_cap + slot;
    // (with entry lists the ray carries its tile instead of tmax, which is the constant 10000 of src/shader.rgen:87)
    const F3 o = f.batch_k > 1 ? mk3(bt.position[fi][0], bt.position[fi][1], bt.position[fi][2]) : mk3(u.position[0], u.position[1], u.position[2]);
    st_stream(&f.ray_o[0][v], make_float4(o.x, o.y, o.z, f.entry != nullptr ? __uint_as_float(tile) : 10000.0f));
    st_stream(&f.ray_d[0][v], survive ? make_float4(d.x, d.y, d.z, __uint_as_float(sid)) : make_float4(0.f, 0.f, 0.f, __uint_as_float(SID_DEAD)));   // (a zero direction: no ray in this slot)
  }
}

// ------------------------------------------------------------------------------------------------
// Traversal kernels.
struct TraceArgs {
  SceneDev sc;
  const float4* ray_o;
  const float4* ray_d;
  const uint32_t* tails;       // counters + cnt_tail(queue, 0): entries per shard, CNT_STRIDE apart
  uint32_t* work;              // counters + cnt_work(queue, 0): chunk cursors, CNT_STRIDE apart
  uint32_t shard_cap;
  float4* hit_a;               // closest-hit pipeline outputs
  int32_t* hit_inst;
  const float4* sh_c;          // shadow pipeline
  float4* sample_color;
  HitRec* raw_out;             // raw mode
  int32_t* ovf_stack;
  uint32_t* counters;
  float tmin;
  uint32_t rays_per_lane;      // device-side grid sizing (variant 0): blocks beyond total/(256*rays_per_lane) exit
  uint32_t min_blocks;
  const EntryRec* entry;       // ENTRY kernels: the tile records of k_entry (closest hit: the ray carries its tile in o.w;
  const uint32_t* sh_e;        // shadow: record index | ENTRY_REVERSE of every shadow-queue entry, written by k_shade)
};

constexpr int MODE_CLOSEST = 0;  // pipeline closest hit: o.w = tmax, d.w = sid
constexpr int MODE_SHADOW = 1;   // pipeline any hit + shading epilogue
constexpr int MODE_RAW = 2;      // o.w = tmin, d.w = tmax; writes HitRec


// ---- variant 0: BVH2, ONE LANE PER RAY, persistent threads with per-lane refill.
//   * 64-ray chunks from the sharded cursors; the next chunk is in flight into registers while the
//     current one sits in LDS; idle lanes take rays from LDS by ballot + prefix rank once enough of
//     them wait (REFILL_MIN), so the wave stays dense without paying the refill code per ray;
//   * phased loop: the interior-node loop runs while most live lanes are at interior nodes; lanes
//     that reach a leaf, an instance or the end of their ray wait there, then each of those phases
//     runs ONCE for all waiting lanes — the expensive, rarer bodies execute densely instead of being
//     dragged through every trip;
//   * per-lane stack in LDS [entry][lane] (bank = lane, conflict free), spill to HBM beyond it;
//   * results staged in LDS and flushed in bursts (stores share the in-order vmcnt with node loads).
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16
#endif
#ifndef RT_KEEP_NUM
#define RT_KEEP_NUM 5   /* interior loop continues while >= RT_KEEP_NUM/8 of the live lanes are interior */
#endif
#ifndef RT_INTERIOR_UNROLL
#define RT_INTERIOR_UNROLL 2
#endif
#ifndef RT_WIDE_UNROLL
#define RT_WIDE_UNROLL 1
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 5    /* waves per SIMD the shipped traversal kernels are register-allocated for (= workgroups per CU) */
#endif
#ifndef RT_LEAF_CHAIN
#define RT_LEAF_CHAIN 1      /* leaves tested per visit of the leaf phase when leaf follows leaf on the stack */
#endif
#ifndef RT_STACK2_LDS
#define RT_STACK2_LDS 12   /* 12 entries in LDS (3 KB per wave) let 6 blocks share a CU; deeper paths spill to HBM */
#endif
constexpr int STACK2_LDS = RT_STACK2_LDS;
constexpr uint32_t REFILL_MIN = RT_REFILL_MIN;

// experiment builds (-DRT_EXP_PHASE_SEL=k): diag = (times region k ran, wave cycles spent in it, wave cycles); regions:
// 1 leaf, 2 leave-instance, 3 enter-instance, 4 finish/flush, 5 refill
#ifdef RT_EXP_PHASE_SEL
#define PH_BEGIN(k) uint64_t ph_b##k = 0; if (COUNT && RT_EXP_PHASE_SEL == k) ph_b##k = __builtin_readcyclecounter();
#define PH_END(k) if (COUNT && RT_EXP_PHASE_SEL == k && lane == 0) { diag_iters++; diag_busy += __builtin_readcyclecounter() - ph_b##k; }
#else
#define PH_BEGIN(k)
#define PH_END(k)
#endif

// FAR: the kernel carries the far-ray logic (quant_far).  The host launches the FAR = false instantiations whenever no ray of the
// launch can be far — camera, scene extent and instance scales decide that per frame (rt_api far_possible) — so the headline pays
// nothing for it; the record-level entry point (arbitrary origins) and k_tail always carry it.
// CONT (closest hit, ENTRY): the queue may hold rays k_tile handed on (CONT_FLAG in o.w) — walked in LDS through their tile's blob
// already, they carry their incumbent hit (in their own hit record) and start from the REST words of their tile's record that
// their ray can still reach (mask in o.w).
template <int MODE, bool ANY, bool COUNT, bool WIDE, bool ENTRY = false, bool FAR = true, bool CONT = false>
__device__ __forceinline__ void trace_body(const TraceArgs& a) {
  __shared__ int s_stack[4][STACK2_LDS + 1][64];   // + one scratch row: lanes that do not push write there (fast_step)
  __shared__ float4 s_rays[4][2][64];
  __shared__ float4 s_out[4][64];
  __shared__ int2 s_outq[4][64];
  __shared__ uint32_t s_ent[(ENTRY && MODE == MODE_SHADOW) ? 4 : 1][64];   // entry record of every ray of the current chunk
  // Instance records staged through LDS: what "enter the instance" reads (world->object rows, dequantisation, root, mask)
  // for the first LDS_INSTANCES instances, 80 bytes each.  That phase runs for a quarter of the lanes at a time and was
  // spending ~1200 cycles per pass on the global-memory latency of these few, shared records.
  __shared__ float4 s_inst[LDS_INSTANCES ? LDS_INSTANCES : 1][5];
  // Grid sizing on the device: the launch always has the full persistent grid, but a queue that holds only a
  // few rays per lane runs faster on fewer, less contended waves that refill (every ray costs ~25-40 dependent
  // trips, and a trip is quickest with 1-2 waves per SIMD) — surplus blocks leave at once.
  {
    uint32_t total = 0;
#pragma unroll
    for (int t = 0; t < N_SHARDS; t++) total += ld_cursor(a.tails + t * CNT_STRIDE);
    uint32_t want = (total + 256u * a.rays_per_lane - 1u) / (256u * a.rays_per_lane);
    want = (want + (N_SHARDS - 1)) & ~(uint32_t)(N_SHARDS - 1);
    if (want < a.min_blocks) want = a.min_blocks;
    if (blockIdx.x >= want) return;
  }
  const int n_lds_inst = a.sc.n_inst < LDS_INSTANCES ? a.sc.n_inst : LDS_INSTANCES;
  if ((int)threadIdx.x < n_lds_inst) {
    const InstanceDev* I = a.sc.inst + threadIdx.x;
    const float4* mp = reinterpret_cast<const float4*>(I->w2o);
    s_inst[threadIdx.x][0] = mp[0]; s_inst[threadIdx.x][1] = mp[1]; s_inst[threadIdx.x][2] = mp[2];
    s_inst[threadIdx.x][3] = make_float4(I->q_lo[0], I->q_lo[1], I->q_lo[2], __int_as_float(I->blas_root));
    s_inst[threadIdx.x][4] = make_float4(I->q_scale[0], I->q_scale[1], I->q_scale[2], __uint_as_float(I->mask));
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  int* const stk = &s_stack[wave][0][lane];                 // entry e at stk[e * 64]
  int32_t* const ovf = a.ovf_stack + (size_t)(blockIdx.x * 256u + threadIdx.x) * a.sc.ovf_stride;
  uint64_t cnt_nodes = 0, cnt_tris = 0, diag_iters = 0, diag_busy = 0;
  const uint64_t diag_t0 = COUNT ? __builtin_readcyclecounter() : 0;

  // ---- work distribution (wave-uniform): prefetched chunk in registers, current chunk in LDS
  uint32_t shard = blockIdx.x & (N_SHARDS - 1), tried = 0;
  uint32_t pf_base = 0, pf_count = 0;
  float4 pf_o = make_float4(0, 0, 0, 0), pf_d = pf_o;
  uint32_t pf_e = ENTRY_FROM_ROOT;
  uint32_t chunk_base = 0, chunk_count = 0, chunk_pos = 0;
  auto prefetch = [&]() {
    pf_count = 0;
    while (tried < (uint32_t)N_SHARDS) {
      const uint32_t size = ld_cursor(a.tails + shard * CNT_STRIDE);
      uint32_t off = 0;
      if (lane == 0 && size) off = atomicAdd(a.work + shard * CNT_STRIDE, 64u);
      off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
      if (size && off < size) { pf_base = shard * a.shard_cap + off; pf_count = (size - off) < 64u ? (size - off) : 64u; break; }
      shard = (shard + 1u) & (N_SHARDS - 1); tried++;
    }
    if (lane < pf_count) {
      pf_o = ld_stream(&a.ray_o[pf_base + lane]); pf_d = ld_stream(&a.ray_d[pf_base + lane]);
      if (ENTRY && MODE == MODE_SHADOW) pf_e = (uint32_t)ld_stream(reinterpret_cast<const int*>(a.sh_e) + pf_base + lane);
    }
  };
  auto promote = [&]() {
    s_rays[wave][0][lane] = pf_o; s_rays[wave][1][lane] = pf_d;
    if (ENTRY && MODE == MODE_SHADOW) s_ent[wave][lane] = pf_e;
    chunk_base = pf_base; chunk_count = pf_count; chunk_pos = 0;
    prefetch();
  };
  prefetch();
  promote();

  uint32_t out_count = 0;
  auto flush = [&]() {
    if (lane < out_count) {
      const float4 r = s_out[wave][lane];
      const int2 k = s_outq[wave][lane];
      if (MODE == MODE_CLOSEST) { st_stream(&a.hit_a[k.x], r); st_stream(&a.hit_inst[k.x], k.y); }
      else if (MODE == MODE_SHADOW) {
        // src/shader_shadow.rmiss:6 + src/shader.rgen:114-129: lit iff nothing was hit.  The light term of the
        // shadow-queue entry is fetched here, once per result burst, instead of riding along in registers.
        // (the entry holds tmpColor for the lit case; the shadowed case keeps Iamb*ka — of the tagged material when a table is set)
        float cr = 0.08f, cg = 0.24f, cb = 0.08f;
        if (r.x != 0.0f || a.sc.n_materials != 0) {
          const float4 shc = ld_stream(&a.sh_c[(uint32_t)k.y]);
          if (r.x != 0.0f) { cr = shc.x; cg = shc.y; cb = shc.z; }
          else { const F3 amb = ambient_of(a.sc, __float_as_uint(shc.w)); cr = amb.x; cg = amb.y; cb = amb.z; }
        }
        st_stream(&a.sample_color[(uint32_t)k.x], make_float4(cr, cg, cb, 1.0f));
      }
      else { HitRec h; h.t = r.x; h.u = r.y; h.v = r.z; h.prim = (int)__float_as_uint(r.w); h.inst = k.y; a.raw_out[k.x] = h; }
    }
    out_count = 0;
  };

  // ---- per-lane ray state
  bool need = true;
  uint32_t q = 0, sid = 0;
  float tmin_ray = 0.f, tmax = 0.f;   // tmin is a per-ray value only in the raw mode; the pipeline uses one constant
  F3 wo = mk3(0, 0, 0), wd = mk3(0, 0, 1), co = wo, cd = wd, qs = mk3(1, 1, 1), qb = mk3(0, 0, 0);
  uint3 rot = make_uint3(0u, 0u, 0u);
  bool far = false;   // quant_far() of the current space: this lane's visits take the generic path with the widened slab test
  float best_t = 0.f, best_u = 0.f, best_v = 0.f;
  int best_prim = -1, best_inst = -1, cur_inst = -1, sp = 0, cur = REF_DONE;
  // BLAS nodes, then the TLAS nodes (WIDE: the 64-byte 4-ary records with the same numbering)
  const char* const node_bytes = WIDE ? reinterpret_cast<const char*>(a.sc.wide_nodes) : reinterpret_cast<const char*>(a.sc.blas_nodes);

  auto push = [&](int v) {
    if (sp < STACK2_LDS) stk[sp * 64] = v;
    else *reinterpret_cast<volatile int32_t*>(ovf + (sp - STACK2_LDS)) = v;
    sp++;
  };
  auto pop = [&]() {
    sp--;
    if (sp < STACK2_LDS) cur = stk[sp * 64];
    else cur = *reinterpret_cast<volatile int32_t*>(ovf + (sp - STACK2_LDS));   // volatile: never merged with the LDS load into a flat_load
  };

  // ray -> object space of instance ii (t preserved); returns (root of its BLAS, instance mask)
  auto to_instance = [&](int ii, int& root, uint32_t& imask) {
    float4 m0, m1, m2, ql, qsc;   // w2o rows, (q_lo, root), (q_scale, mask)
    if (ii < n_lds_inst) { m0 = s_inst[ii][0]; m1 = s_inst[ii][1]; m2 = s_inst[ii][2]; ql = s_inst[ii][3]; qsc = s_inst[ii][4]; }
    else {
      const InstanceDev* I = a.sc.inst + ii;
      const float4* mp = reinterpret_cast<const float4*>(I->w2o);
      m0 = mp[0]; m1 = mp[1]; m2 = mp[2];
      ql = make_float4(I->q_lo[0], I->q_lo[1], I->q_lo[2], __int_as_float(I->blas_root));
      qsc = make_float4(I->q_scale[0], I->q_scale[1], I->q_scale[2], __uint_as_float(I->mask));
    }
    root = __float_as_int(ql.w); imask = __float_as_uint(qsc.w);
    if ((imask & 0xFFu) != 0u) {
      float m[12];
      m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
      m[8] = m2.x; m[9] = m2.y; m[10] = m2.z; m[11] = m2.w;
      co = xform_point(m, wo); cd = xform_vec(m, wd);
      const float qlo3[3] = {ql.x, ql.y, ql.z}, qsc3[3] = {qsc.x, qsc.y, qsc.z};
      quant_space(co, cd, qlo3, qsc3, qs, qb, rot); far = FAR && quant_far_o(co, qlo3, qsc3);
    }
  };

  for (;;) {
    // ---- (A) refill
    PH_BEGIN(5)
    const uint64_t need_mask = __ballot(need);
    const uint32_t n_need = (uint32_t)__builtin_popcountll(need_mask);
    if (n_need >= REFILL_MIN || n_need == 64u) {
      if (chunk_pos == chunk_count && pf_count > 0) promote();
      if (chunk_pos < chunk_count) {
        const uint32_t rank = prefix_rank(need_mask);
        const uint32_t avail = chunk_count - chunk_pos;
        if (need && rank < avail) {
          const uint32_t ci = chunk_pos + rank;
          q = chunk_base + ci;
          const float4 ro = s_rays[wave][0][ci], rd = s_rays[wave][1][ci];
          if (MODE == MODE_RAW) { tmin_ray = ro.w; tmax = rd.w; }
          else { tmax = ro.w; sid = __float_as_uint(rd.w); }
          wo = mk3(ro.x, ro.y, ro.z); wd = mk3(rd.x, rd.y, rd.z);
          co = wo; cd = wd;
          cur_inst = -1;
          stk[0] = REF_DONE;
          bool handed_on = false;
          if (CONT && (__float_as_uint(ro.w) & CONT_FLAG) != 0u) {
            // handed on by k_tile: the surviving REST words of the tile's record (far to near on the stack), the incumbent hit
            handed_on = true;
            const uint32_t ent = __float_as_uint(ro.w);
            const uint32_t tile = ent & 0xFFFFFFu, mask = (ent >> 24) & 15u;
            const int4* rp = reinterpret_cast<const int4*>(a.entry + tile);
            const int4 r0 = rp[0], r1 = rp[1];
            const uint32_t n_rest = (((uint32_t)r0.x >> 4) & 15u) - 1u;
            const int wk[4] = {r0.z, r0.w, r1.x, r1.y};   // rest words, far to near (bottom first)
            sp = 1;
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
              if (k < n_rest && ((mask >> (n_rest - 1u - k)) & 1u) != 0u) { stk[sp * 64] = wk[k]; sp++; }
            sp--; cur = stk[sp * 64];
            quant_space(co, cd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = false;
            const float4 inc = a.hit_a[q];
            best_t = inc.x; best_u = inc.y; best_v = inc.z; best_prim = (int)__float_as_uint(inc.w); best_inst = a.hit_inst[q];
          } else
          if (ENTRY) {
            // the walk starts at the record of the ray's tile (k_entry): its words go on the stack, its first node becomes the
            // current one, and the ray enters the record's instance here instead of in phase (C)
            uint32_t ent;
            if (MODE == MODE_SHADOW) ent = s_ent[wave][ci];
            else ent = __float_as_uint(ro.w);   // (o.w carries the tile; tmax is the constant 10000 of src/shader.rgen:87)
            const bool rev_flag = MODE == MODE_SHADOW && (ent & ENTRY_REVERSE) != 0u;
            if (MODE == MODE_SHADOW) ent &= ~ENTRY_REVERSE;
            int4 r0 = make_int4((int)ENTRY_EMPTY, REF_DONE, 0, 0), r1 = make_int4(0, 0, 0, 0);
            if (ent != ENTRY_FROM_ROOT) {
              const int4* rp = reinterpret_cast<const int4*>(a.entry + ent);
              r0 = rp[0]; r1 = rp[1];
            }
            const uint32_t hdr = (uint32_t)r0.x;
            if (ent == ENTRY_FROM_ROOT) {
              quant_space(co, cd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = FAR && quant_far_o(co, a.sc.tlas_q_lo, a.sc.tlas_q_scale);
              sp = 1; cur = a.sc.tlas_root + (a.sc.batch_samples ? (int)frame_of(sid, a.sc.batch_samples) * a.sc.tlas_stride : 0);
            } else if (hdr == ENTRY_EMPTY) {
              sp = 1; cur = REF_DONE;   // nothing a ray of this tile can hit: the ray is finished (a miss)
            } else {
              const uint32_t nw = hdr & 15u, n_rm = (hdr >> 4) & 15u, ia = hdr >> 8;
              // ENTRY_REVERSE: the instance's subtrees in the opposite order (the farthest from the view point first)
              const bool rev = MODE == MODE_SHADOW && rev_flag && nw > n_rm;   // (camera records are never reversed: no such code in the closest-hit kernel)
              int cur_new = r0.y;
              const int wk[6] = {r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
              for (uint32_t k = 0; k < 6u; k++)
                if (k < nw) {
                  if (rev && k == n_rm) cur_new = wk[k];
                  else stk[((!rev || k < n_rm) ? 1u + k : 1u + nw + n_rm - k) * 64u] = wk[k];
                }
              if (rev) stk[(1u + n_rm) * 64u] = r0.y;
              sp = 1 + (int)nw;
              cur = cur_new;
              __builtin_amdgcn_sched_barrier(0);   // the record's words are on the stack before the instance record is fetched (register peak of the kernel)
              if (ia != ENTRY_NO_INST) { int root; uint32_t imask; to_instance((int)ia, root, imask); cur_inst = (int)ia; }   // (k_entry never names an invisible instance)
              else { quant_space(co, cd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = FAR && quant_far_o(co, a.sc.tlas_q_lo, a.sc.tlas_q_scale); }   // (to_instance set `far` against the MESH's quantisation: it must survive)
            }
          } else {
            quant_space(co, cd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = FAR && quant_far_o(co, a.sc.tlas_q_lo, a.sc.tlas_q_scale);
            sp = 1;
            cur = a.sc.tlas_root;   // TLAS root (always interior); in a frame batch: the root of the ray's frame
            if (MODE != MODE_RAW && a.sc.batch_samples) cur += (int)frame_of(sid, a.sc.batch_samples) * a.sc.tlas_stride;
          }
          if (!(CONT && handed_on)) { best_t = (ENTRY && MODE == MODE_CLOSEST) ? 10000.0f : tmax; best_u = 0.f; best_v = 0.f; best_prim = -1; best_inst = -1; }
          need = false;
        }
        chunk_pos += n_need < avail ? n_need : avail;
      } else if (n_need == 64u) { flush(); break; }   // queue drained and every lane idle
    }
    PH_END(5)
    const uint32_t live = 64u - (uint32_t)__builtin_popcountll(__ballot(need));
    const uint32_t keep_going = (live * RT_KEEP_NUM + 7u) >> 3;   // share of the live lanes that must still be interior

    // ---- (B) interior-node loop: runs while most live lanes are at interior nodes.  The exit vote (ballot, popcount,
    // compare) is taken every RT_INTERIOR_UNROLL trips; in between, lanes that left the interior state just idle.
    auto interior_step = [&]() {
      if (WIDE) {
        if (cur >= 0) {
          const uint4* np = reinterpret_cast<const uint4*>(node_bytes + ((uint32_t)cur << 6));   // 64-byte record: four requests
          const uint4 X = np[0], Y = np[1], Z = np[2], R = np[3];
          if (COUNT) cnt_nodes++;
          float t0, t1, t2, t3;
          bool h0 = far ? slab_q_far(X.x, Y.x, Z.x, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0) : slab_q(X.x, Y.x, Z.x, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0);
          bool h1 = far ? slab_q_far(X.y, Y.y, Z.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1) : slab_q(X.y, Y.y, Z.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1);
          bool h2 = far ? slab_q_far(X.z, Y.z, Z.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t2) : slab_q(X.z, Y.z, Z.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t2);
          bool h3 = far ? slab_q_far(X.w, Y.w, Z.w, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t3) : slab_q(X.w, Y.w, Z.w, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t3);
          if (far && cur_inst < 0) { h0 = h1 = h2 = h3 = true; t0 = t1 = t2 = t3 = 0.0f; }   // far ray in world space: the TLAS does not cull (see the BVH2 visit below)
          // entry distance with the entry number in its two low mantissa bits; a miss sorts last
          uint32_t k0 = h0 ? (__float_as_uint(t0) & ~3u) : 0xFFFFFFFFu;
          uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xFFFFFFFFu;
          uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xFFFFFFFFu;
          uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xFFFFFFFFu;
          const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
#define RT_CAS(a_, b_) { const uint32_t lo_ = min(a_, b_), hi_ = max(a_, b_); a_ = lo_; b_ = hi_; }
          RT_CAS(k0, k1) RT_CAS(k2, k3) RT_CAS(k0, k2) RT_CAS(k1, k3) RT_CAS(k1, k2)
#undef RT_CAS
          auto link = [&](uint32_t key) -> int {
            const uint32_t e = key & 3u;
            return (int)(e == 0u ? R.x : e == 1u ? R.y : e == 2u ? R.z : R.w);
          };
          if (nh == 0) pop();
          else {
            if (nh > 1) {   // far entries first, so the nearest of them is popped first
              if (nh > 2) {
                if (nh > 3) push(link(k3));
                push(link(k2));
              }
              push(link(k1));
            }
            cur = link(k0);
          }
        }
        return;
      }
      if (cur >= 0) {   // idle lanes hold REF_DONE
        const uint4* np = reinterpret_cast<const uint4*>(node_bytes + ((uint32_t)cur << 5));   // 32-byte node: two requests
        const uint4 Q0 = np[0], Q1 = np[1];
        const int2 ch = make_int2((int)Q1.z, (int)Q1.w);
        if (COUNT) cnt_nodes++;
        float t0, t1;
        // A far ray in WORLD space (quant_far): its canonical object-space twin (rounded once per instance) can deviate from it by
        // more than the padding of the instance boxes, so the TLAS is not allowed to cull for it — every instance is entered and
        // the BLAS tests, made on the object-space ray itself with the widened slabs, decide.
        const bool open_all = far && cur_inst < 0;
        bool h0 = open_all ? (Q0.x & 0xFFFFu) <= (Q0.x >> 16) : (far ? slab_q_far(Q0.x, Q0.y, Q0.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0) : slab_q(Q0.x, Q0.y, Q0.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0));
        bool h1 = open_all ? (Q0.w & 0xFFFFu) <= (Q0.w >> 16) : (far ? slab_q_far(Q0.w, Q1.x, Q1.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1) : slab_q(Q0.w, Q1.x, Q1.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1));
        if (open_all) { t0 = 0.0f; t1 = 0.0f; }
        if (h0 && h1) {
          const bool swap = t1 < t0;
          push(swap ? ch.x : ch.y);
          cur = swap ? ch.y : ch.x;
        } else if (h0) cur = ch.x;
        else if (h1) cur = ch.y;
        else pop();
      }
    };
    // The same visit without a branch, for the common case that the stack stays inside its LDS rows: the top of the
    // stack is read while the node is still in flight (a pop then costs nothing on the dependent chain), the far child
    // is written unconditionally (lanes that do not push write to the scratch row), and the next node is a chain of
    // selects.  The visit is the unit of the kernel's dependent instruction chain, which is what bounds it.
    typedef __attribute__((address_space(3))) volatile int lds_vint;   // volatile AND still an LDS pointer (ds_read/ds_write)
    lds_vint* const stk_lds = (lds_vint*)stk;
#ifdef RT_EXP_VISIT_STAMPS
    // measurement build (instrumented kernels only; tools/visit_stamps.sh): what a wave's fast visit is made of — RT_EXP_VISIT_STAMPS 1: from
    // the issue of the node fetch to its arrival (forced s_waitcnt), 2: from there to the end of the visit (box tests, selects, stack
    // store), 3: the whole visit.  profiles/r03_visit_stamps.txt: 1 670 + 408 = 2 118 cycles (closest hit), 1 043 + 445 = 1 494 (shadow).
    auto fast_step = [&]() {
      const uint64_t vs_t0 = COUNT ? __builtin_readcyclecounter() : 0;
      uint4 Q0 = make_uint4(0, 0, 0, 0), Q1 = Q0; int top = 0;
      if (cur >= 0) {
        const uint4* np = reinterpret_cast<const uint4*>(node_bytes + ((uint32_t)cur << 5));
        Q0 = np[0]; Q1 = np[1];
        top = stk_lds[(sp - 1) * 64];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const uint64_t vs_t1 = COUNT ? __builtin_readcyclecounter() : 0;
      if (cur >= 0) {
        const int2 ch = make_int2((int)Q1.z, (int)Q1.w);
        if (COUNT) cnt_nodes++;
        float t0, t1;
        const bool h0 = slab_q(Q0.x, Q0.y, Q0.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0);
        const bool h1 = slab_q(Q0.w, Q1.x, Q1.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1);
        const bool both = h0 && h1, none = !(h0 || h1), swap = t1 < t0;
        const uint32_t pm = 0u - (uint32_t)both;
        stk_lds[(((uint32_t)sp & pm) | ((uint32_t)STACK2_LDS & ~pm)) * 64u] = swap ? ch.x : ch.y;
        const int one = h0 ? ch.x : ch.y;
        cur = both ? (swap ? ch.y : ch.x) : (none ? top : one);
        sp += (both ? 1 : 0) - (none ? 1 : 0);
      }
      if (COUNT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t vs_t2 = __builtin_readcyclecounter();
        if (lane == 0) { diag_iters++; diag_busy += RT_EXP_VISIT_STAMPS == 1 ? vs_t1 - vs_t0 : RT_EXP_VISIT_STAMPS == 2 ? vs_t2 - vs_t1 : vs_t2 - vs_t0; }
      }
    };
#else
    auto fast_step = [&]() {
      if (cur >= 0) {
        const uint4* np = reinterpret_cast<const uint4*>(node_bytes + ((uint32_t)cur << 5));
        const uint4 Q0 = np[0], Q1 = np[1];
        // volatile: the read has to be issued here, under the node fetch, not sunk into a branch after the box tests
        const int top = stk_lds[(sp - 1) * 64];
        const int2 ch = make_int2((int)Q1.z, (int)Q1.w);
        if (COUNT) cnt_nodes++;
        float t0, t1;
        const bool h0 = slab_q(Q0.x, Q0.y, Q0.z, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t0);
        const bool h1 = slab_q(Q0.w, Q1.x, Q1.y, qs, qb, rot, (MODE == MODE_RAW ? tmin_ray : a.tmin), best_t, t1);
        const bool both = h0 && h1, none = !(h0 || h1), swap = t1 < t0;
        const uint32_t pm = 0u - (uint32_t)both;   // all ones when pushing
        stk_lds[(((uint32_t)sp & pm) | ((uint32_t)STACK2_LDS & ~pm)) * 64u] = swap ? ch.x : ch.y;   // the far child: one store, no branch
        const int one = h0 ? ch.x : ch.y;
        cur = both ? (swap ? ch.y : ch.x) : (none ? top : one);
        sp += (both ? 1 : 0) - (none ? 1 : 0);
      }
    };
#endif
    constexpr int UNROLL = WIDE ? RT_WIDE_UNROLL : RT_INTERIOR_UNROLL;
    for (;;) {
      const uint32_t n_int = (uint32_t)__builtin_popcountll(__ballot(cur >= 0));
      if (n_int == 0 || n_int < keep_going) break;
#if !defined(RT_EXP_PHASE_SEL) && !defined(RT_EXP_VISIT_STAMPS)
      if (COUNT && lane == 0) { diag_iters++; diag_busy += n_int; }
#endif
      // fast visits need every stack they touch inside the LDS rows: sp - 1 >= 0 always holds for a live ray, and
      // UNROLL pushes must fit below row STACK2_LDS
      const bool deep = cur >= 0 && (sp + UNROLL > STACK2_LDS || far);   // far rays: the generic visit has the widened slab test
      if (!WIDE && __ballot(deep) == 0) {
#pragma unroll
        for (int r = 0; r < UNROLL; r++) fast_step();
      } else {
#pragma unroll
        for (int r = 0; r < UNROLL; r++) interior_step();
      }
    }

    // ---- (C) the rarer bodies, each run once for all lanes that wait at them.  They are chained (leaf, then
    // leave-instance, then enter-instance) so that a lane can finish a leaf, leave its instance and enter the
    // next one in the same pass instead of waiting a whole pass for each step.
    PH_BEGIN(1)
    if (cur < 0 && cur > REF_MARK && cur_inst >= 0) {
      // BLAS leaf: Moller-Trumbore on 48-byte packets.  The entry popped after a leaf is often a leaf again (the far child of a
      // bottom node whose two children were both hit): up to RT_LEAF_CHAIN leaves are tested in one visit of this phase, instead of
      // one outer pass each — passes, not node visits, are what a ray's life is counted in.
      for (int chain = 0; chain < RT_LEAF_CHAIN; chain++) {
        const uint32_t ref = (uint32_t)(~cur);
        const uint32_t first = ref >> 3, count = (ref & 7u) + 1u;
        for (uint32_t k = 0; k < count; k++) {
          const float4* tp = a.sc.tris + (size_t)(first + k) * 3;
          const float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
          if (COUNT) cnt_tris++;
          float tt, uu, vv;
          // (the closest-hit pipeline's primary rays all have tmax 10000, src/shader.rgen:87: a constant for the ENTRY kernel, not a register)
          if (tri_test(T0, T1, T2, co, cd, (MODE == MODE_RAW ? tmin_ray : a.tmin), (ANY ? best_t : ((ENTRY && MODE == MODE_CLOSEST) ? 10000.0f : tmax)), tt, uu, vv)) {
            if (ANY && MODE == MODE_SHADOW) best_inst = cur_inst;   // the shadow pipeline only asks WHETHER something was hit: no record to keep
            else {
              const int prim = (int)__float_as_uint(T2.y);
              const bool better = (best_inst < 0) || (tt < best_t) ||
                                  (tt == best_t && (cur_inst < best_inst || (cur_inst == best_inst && prim < best_prim)));
              if (better) { best_t = tt; best_u = uu; best_v = vv; best_prim = prim; best_inst = cur_inst; }
            }
          }
        }
        if (ANY && best_inst >= 0) { cur = REF_DONE; break; }   // any hit ends the ray (flags 13, src/shader.rgen:67)
        pop();
        if (!(cur < 0 && cur > REF_MARK)) break;
      }
    }
    PH_END(1)
    PH_BEGIN(2)
    if (cur == REF_MARK) {
      // leave the instance: back to the TLAS.  The world-space ray space is needed again only if the next entry is
      // an interior TLAS node (a TLAS leaf sets up its own space, and the bottom sentinel ends the ray)
      cur_inst = -1;
      pop();
      if (cur >= 0) { quant_space(wo, wd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = FAR && quant_far_o(wo, a.sc.tlas_q_lo, a.sc.tlas_q_scale); }
    }
    PH_END(2)
    PH_BEGIN(3)
    if (cur < 0 && cur > REF_MARK && cur_inst < 0) {
      // TLAS leaf: enter the instance (ray -> object space, t preserved)
      const int ii = ~cur;
      int root; uint32_t imask;
      to_instance(ii, root, imask);
      if ((imask & 0xFFu) == 0u) {
        pop();   // invisible to the ray mask 0xFF; the ray space may still be that of the instance left before
        if (cur >= 0) { quant_space(wo, wd, a.sc.tlas_q_lo, a.sc.tlas_q_scale, qs, qb, rot); far = FAR && quant_far_o(wo, a.sc.tlas_q_lo, a.sc.tlas_q_scale); }
      } else {
        push(REF_MARK);
        cur_inst = ii; cur = root;
      }
    }

    PH_END(3)
    // ---- (D) finished rays: append the result to the wave's LDS out-list
    PH_BEGIN(4)
    const bool fin = !need && cur == REF_DONE;
    const uint64_t fin_mask = __ballot(fin);
    if (fin_mask != 0) {
      const uint32_t n_fin = (uint32_t)__builtin_popcountll(fin_mask);
      if (out_count + n_fin > 64u) flush();   // the out-list holds 64 results
      if (fin) {
        const uint32_t slot = out_count + prefix_rank(fin_mask);
        if (MODE == MODE_SHADOW) {
          s_out[wave][slot].x = best_inst < 0 ? 1.0f : 0.0f;   // lit?
          s_outq[wave][slot] = make_int2((int)sid, (int)q);
        } else {
          s_out[wave][slot] = make_float4(best_t, best_u, best_v, __uint_as_float((uint32_t)best_prim));
          s_outq[wave][slot] = make_int2((int)q, best_inst);
        }
        need = true;
      }
      out_count += n_fin;
    }
    PH_END(4)
  }
  if (COUNT) {
    for (int off = 32; off > 0; off >>= 1) {
      cnt_nodes += __shfl_down((unsigned long long)cnt_nodes, off);
      cnt_tris += __shfl_down((unsigned long long)cnt_tris, off);
    }
    if (lane == 0) {
      const int off = ANY ? (CNT_NODE_VISITS_SH - CNT_NODE_VISITS) : 0;
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + CNT_NODE_VISITS + off), (unsigned long long)cnt_nodes);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + CNT_TRI_TESTS + off), (unsigned long long)cnt_tris);
      const int dg = ANY ? CNT_DIAG_SH : CNT_DIAG;
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + dg), (unsigned long long)diag_iters);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + dg + 2), (unsigned long long)diag_busy);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.counters + dg + 4), (unsigned long long)(__builtin_readcyclecounter() - diag_t0));
    }
  }
}

// (five 256-thread workgroups per CU = five waves per SIMD is what the LDS admits: the register allocator of the shipped kernels
// is held to that; the instrumented ones (k_trace_count) may take more registers rather than spill — a spill reload would
// distort the very phase timings they exist to measure)
template <int MODE, bool ANY, bool WIDE, bool ENTRY = false, bool FAR = true, bool CONT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU))) void k_trace(TraceArgs a) { trace_body<MODE, ANY, false, WIDE, ENTRY, FAR, CONT>(a); }
template <int MODE, bool ANY, bool WIDE, bool ENTRY = false, bool FAR = true, bool CONT = false>
__global__ __launch_bounds__(256) void k_trace_count(TraceArgs a) { trace_body<MODE, ANY, true, WIDE, ENTRY, FAR, CONT>(a); }

#include "kernels_tile.inc"   // k_blob, k_trace_tile: the tile's nodes and triangle packets staged through LDS
#include "kernels_beam.inc"   // k_beam: the primary rays of a pixel walked together

#ifdef RT_ALT_KERNELS
#include "kernels_alt.inc"   // k_packet, k_trace4: alternatives measured slower, only in librt_mi355x_alt.so
#endif

// ------------------------------------------------------------------------------------------------
// k_shade: closest-hit / miss shading and path continuation for one bounce.  Persistent grid; a wave
// takes 64-entry batches interleaved over the shards and appends continuation / shadow rays to the
// SAME shard with one wave-aggregated atomic each (wavefront ballot compaction).
struct ShadeArgs {
  SceneDev sc;
  FrameDev f;
  UniformsDev u;
  int bounce;
  BatchTab bt;
};

// TILE: bounce 0 of a frame with tile blobs — the hit records lie in two regions per shard (kernels_tile.inc)
// BATCH: the frame is one of a frame batch — the light is the one of the sample's frame (the single-frame instantiations are untouched)
// RUNS: bounce 0 of a frame with shadow runs (kernels_beam.inc) — the shadow ray of a primary hit goes into the slot of its primary ray
template <bool TILE = false, bool BATCH = false, bool RUNS = false>
__device__ __forceinline__ void shade_body(const ShadeArgs& a, const int bounce) {
  const FrameDev& f = a.f;
  const UniformsDev& U = a.u;
  const int cur = bounce & 1, nxt = cur ^ 1;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // bounce 0 of a frame with tile blobs has its rays in two regions per shard: queue 0 at the bottom, the tile rays (walked by
  // k_trace_tile) at the top — a second pass with the other count and offset
  const int n_pass = TILE ? 2 : 1;
  for (int pass = 0; pass < n_pass; pass++) {
  uint32_t cnt[N_SHARDS], maxb = 0;
#pragma unroll
  for (int t = 0; t < N_SHARDS; t++) {
    if (TILE && pass) {   // the tile region: a fixed number of slots per blob of the shard's lists (kernels_tile.inc)
      uint32_t blobs = 0;
#pragma unroll
      for (int k = 0; k < BLOB_CLASSES; k++) blobs += ld_cursor(f.counters + cnt_tail(Q_BLOB_LIST + k, t));
      cnt[t] = blobs * (256u * ((U.samples_per_pixel + 3u) / 4u));
    } else cnt[t] = ld_cursor(f.counters + cnt_tail(bounce, t));
    maxb = max(maxb, (cnt[t] + 63u) >> 6);
  }
  const uint32_t n_waves = gridDim.x * 4u;
  for (uint32_t g = blockIdx.x * 4u + wave; g < maxb * N_SHARDS; g += n_waves) {
    const uint32_t shard = g & (N_SHARDS - 1), base = (g >> 3) << 6;
    uint32_t n = 0;
#pragma unroll
    for (int t = 0; t < N_SHARDS; t++) n = (shard == (uint32_t)t) ? cnt[t] : n;
    if (base >= n) continue;
    const uint32_t q = shard * f.shard_cap + ((TILE && pass) ? f.shard_cap - n : 0u) + base + lane;
    bool push_next = false, push_shadow = false, settled = false;   // settled: a shadow ray whose outcome cannot change the sample
    F3 no = mk3(0, 0, 0), nd = mk3(0, 0, 1);
    float sh_tmax = 0.f; F3 sh_c = mk3(0, 0, 0); float sh_w = 0.f;
    uint32_t sh_ent = ENTRY_FROM_ROOT;
    uint32_t sid = SID_DEAD;
    int inst = HIT_DEAD;
    if (base + lane < n) inst = ld_stream(&f.hit_inst[q]);
    if (inst != HIT_DEAD) {   // (a slot of the tile region without a ray: nothing to shade)
      const float4 rd = ld_stream(&f.ray_d[cur][q]);
      sid = __float_as_uint(rd.w);
      const F3 d = mk3(rd.x, rd.y, rd.z);
      if (inst < 0) {
        // src/shader.rmiss:11 + src/shader.rgen:90-94
        const F3 c = sample_sky(a.sc, mk3(d.x, d.y, -d.z));
        st_stream(&f.sample_color[sid], make_float4(c.x, c.y, c.z, 1.0f));
      } else {
        // src/shader.rchit:50-96
        const float4 h = ld_stream(&f.hit_a[q]);
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
        // src/shader.rgen:96, generalised (row n4): a per-instance type replaces the two-way switch when the host set one,
        // and an MTL material may fix its own type (illum)
        uint32_t type = I->type != TYPE_BY_OBJECT_INDEX ? I->type : (objectIndex == 0 ? U.center_object_type : U.orbiting_object_type);
        const MaterialDev* M = nullptr;
        uint32_t mat = MATERIAL_NONE;
        if (a.sc.n_materials != 0) {
          mat = a.sc.prim_material[I->first_index / 3u + prim];
          M = a.sc.materials + mat;
          if (M->type != TYPE_BY_INSTANCE) type = M->type;
        }
        const bool last = (uint32_t)bounce >= U.max_bounce_count;
        if (type == 0u) {
          // src/shader.rgen:97-131
          if (dot3(d, N) >= 0.0f) {
            st_stream(&f.sample_color[sid], make_float4(0.08f, 0.24f, 0.08f, 1.0f));
          } else {
            no = fma3(0.01f, N, P);
            F3 light = mk3(U.light_position[0], U.light_position[1], U.light_position[2]);
            uint32_t fi = 0;   // frame batch: the light of the sample's frame
            if (BATCH) { fi = frame_of(sid, a.sc.batch_samples); light = mk3(a.bt.light[fi][0], a.bt.light[fi][1], a.bt.light[fi][2]); }
            const F3 toL = sub3(light, P);
            const float dist = length3(toL);
            const F3 L = mul3(toL, 1.0f / dist);
            const F3 Hh = normalize3(add3(L, neg3(d)));
            const float NdotL = dot3(N, L), NdotH = dot3(N, Hh);
            const float dl = fmaxf(0.0f, NdotL);
            const float sp = M ? pow_int(fmaxf(0.0f, NdotH), (uint32_t)M->ns) : pow100(fmaxf(0.0f, NdotH));
            const uint32_t i = sid / (uint32_t)(f.rows * f.width) - fi * U.samples_per_pixel;   // sample index in its pixel
            float w = 1.0f;
            for (uint32_t k = 0; k < i; k++) w = w * 0.9f;
            const float Iv = U.light_intensity;
            const F3 kd = M ? mk3(M->kd[0], M->kd[1], M->kd[2]) : mk3(0.2f, 1.0f, 0.2f);
            const F3 ks = M ? mk3(M->ks[0], M->ks[1], M->ks[2]) : mk3(0.8f, 0.8f, 0.8f);
            const F3 diff = mk3((Iv * kd.x) * dl, (Iv * kd.y) * dl, (Iv * kd.z) * dl);
            const F3 spec = mk3((Iv * ks.x) * sp, (Iv * ks.y) * sp, (Iv * ks.z) * sp);
            // tmpColor += pow(0.9, i) * (diffuse + specular) on top of Iamb*ka; the shadow kernel writes it if the light is visible
            const F3 amb = ambient_of(a.sc, mat);
            sh_c = fma3(w, add3(diff, spec), amb); sh_w = __uint_as_float(mat);
            nd = L; sh_tmax = dist;
            push_shadow = true;
            // A surface that faces away from the light (and whose half vector does too) gets NOTHING from it: diffuse and specular are
            // exactly 0 and tmpColor stays Iamb*ka whether the shadow ray finds the light or not (src/shader.rgen:113-128) — the sample's
            // colour is the same bits either way, so the ray need not be walked.  It still counts as a shadow ray (the reference issues
            // the traceRayEXT); rt_stats::rays_shadow_untraced says how many were settled here.
            if (f.settle_dead_shadow_rays && __float_as_uint(sh_c.x) == __float_as_uint(amb.x) && __float_as_uint(sh_c.y) == __float_as_uint(amb.y) &&
                __float_as_uint(sh_c.z) == __float_as_uint(amb.z)) {
              st_stream(&f.sample_color[sid], make_float4(amb.x, amb.y, amb.z, 1.0f));
              push_shadow = false; settled = true;
            }
            if (f.light_entry != nullptr) {
              // which tile of the cube around the light does this ray belong to?  Seen from the light the ray's ORIGIN lies in
              // direction v; the ray then runs to within 0.01 of the light (k_entry's beams are widened by that much).
              const F3 v = sub3(no, light);
              const float ax = __builtin_fabsf(v.x), ay = __builtin_fabsf(v.y), az = __builtin_fabsf(v.z);
              const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
              const float vc = axis == 0 ? v.x : (axis == 1 ? v.y : v.z), va = axis == 0 ? v.y : (axis == 1 ? v.z : v.x), vb = axis == 0 ? v.z : (axis == 1 ? v.x : v.y);
              const float c = __builtin_fabsf(vc);
              if (c > 0.0f) {
                const float T8 = (float)(8 * f.light_tiles);
                const float px = (va / c + 1.0f) * 0.5f * T8, py = (1.0f - vb / c) * 0.5f * T8;
                const int tx = min(max((int)(px * 0.125f), 0), f.light_tiles - 1), ty = min(max((int)(py * 0.125f), 0), f.light_tiles - 1);
                const int face = 2 * axis + (vc < 0.0f ? 1 : 0);
                // a surface that faces away from the light is almost always shadowed by its own neighbourhood: start there
                sh_ent = (uint32_t)((face * f.light_tiles + ty) * f.light_tiles + tx) | (NdotL < 0.0f ? ENTRY_REVERSE : 0u);
              }
            }
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
          const float ratio = M ? (outwards ? M->ni : 1.0f / M->ni) : (outwards ? 1.52f : (1.0f / 1.52f));
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
          const float4 ro = ld_stream(&f.ray_o[cur][q]);
          no = mk3(ro.x, ro.y, ro.z); nd = d; push_next = true;
        }
        if (push_next && last) {
          // loop of src/shader.rgen:84 ends: tmpColor keeps Iamb*ka
          push_next = false;
          st_stream(&f.sample_color[sid], make_float4(0.08f, 0.24f, 0.08f, 1.0f));
        }
      }
    }
    {
      const uint64_t m_st = __ballot(settled);   // (statistics: non-returning atomics)
      if (lane == 0 && m_st != 0ull) {
        __hip_atomic_fetch_add(f.counters + cnt_tail(Q_SHADOW0, (int)shard), (uint32_t)__builtin_popcountll(m_st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(f.counters + cnt_work(Q_DEAD, (int)shard), (uint32_t)__builtin_popcountll(m_st), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // wavefront ballot compaction into the next-bounce queue / the shadow queue of the same shard
    const uint32_t slot_n = wave_alloc(push_next, f.counters + cnt_tail(bounce + 1, (int)shard));
    if (push_next) {
      const uint32_t v = shard * f.shard_cap + slot_n;
      st_stream(&f.ray_o[nxt][v], make_float4(no.x, no.y, no.z, 10000.0f));
      st_stream(&f.ray_d[nxt][v], make_float4(nd.x, nd.y, nd.z, __uint_as_float(sid)));
    }
    if (RUNS) {
      // shadow runs: no allocation — the ray's place is its primary ray's; every slot of the run says whether it holds one
      if (base + lane < n) {
        if (push_shadow) {
          st_stream(&f.sh_o[q], make_float4(no.x, no.y, no.z, sh_tmax));
          st_stream(&f.sh_c[q], make_float4(sh_c.x, sh_c.y, sh_c.z, sh_w));
          if (f.sh_e != nullptr) f.sh_e[q] = sh_ent;
        }
        st_stream(&f.sh_d[q], push_shadow ? make_float4(nd.x, nd.y, nd.z, __uint_as_float(sid)) : make_float4(0.f, 0.f, 0.f, __uint_as_float(SID_DEAD)));
      }
      const uint64_t m_sh = __ballot(push_shadow);   // (statistics: a non-returning atomic)
      if (lane == 0 && m_sh != 0ull) __hip_atomic_fetch_add(f.counters + cnt_tail(Q_SHADOW0, (int)shard), (uint32_t)__builtin_popcountll(m_sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const uint32_t slot_s = wave_alloc(push_shadow, f.counters + cnt_tail(Q_SHADOW, (int)shard));
      if (push_shadow) {
        const uint32_t v = f.sh_base + shard * f.shard_cap + slot_s;   // (sh_base: above the shadow runs of bounce 0, when the frame has them)
        st_stream(&f.sh_o[v], make_float4(no.x, no.y, no.z, sh_tmax));
        st_stream(&f.sh_d[v], make_float4(nd.x, nd.y, nd.z, __uint_as_float(sid)));
        st_stream(&f.sh_c[v], make_float4(sh_c.x, sh_c.y, sh_c.z, sh_w));
        if (f.sh_e != nullptr) f.sh_e[v] = sh_ent;
      }
    }
  }
  }
}

template <bool TILE, bool BATCH, bool RUNS = false>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) { shade_body<TILE, BATCH, RUNS>(a, a.bounce); }

// ------------------------------------------------------------------------------------------------
// k_tail: bounces first..maxBounceCount of one frame in ONE launch.  After the first bounce a frame usually
// has few live paths (mirror / glass pixels only) but the loop of src/shader.rgen:84 may run up to 63 more
// times; one launch per bounce and kernel would be 2 x 63 nearly empty launches (or host polls).  Here a
// small persistent grid alternates traversal and shading, separated by a grid-wide barrier, and leaves as
// soon as a bounce queue is empty — the same test the reference loop makes per pixel.  The grid is small
// (rt::tail_grid(): at most TAIL_BLOCKS, and never more than 1/MAX_TAILS_IN_FLIGHT of the workgroups the device can
// hold) so that the tails of every frame in flight are co-resident and a barrier can always complete;
// bounces whose queue held many rays in the previous frame of the context take the per-bounce launches on the full
// grid first (rt_api decides where the tail starts from the queue sizes k_resolve reports).
struct TailArgs {
  TraceArgs tr;          // closest-hit arguments; ray queue / cursors are set per bounce
  ShadeArgs sh;
  uint32_t first_bounce;
  uint32_t* barrier;     // counters + CNT_BARRIER (zeroed with the counters at frame start)
  uint32_t* fault;       // counters + CNT_FAULT: set when a barrier gave up
};

__device__ __forceinline__ void grid_barrier(uint32_t* bar, uint32_t* fault, uint32_t& phase) {
  __syncthreads();
  phase++;
  if (threadIdx.x == 0) {
    __threadfence();                                   // release this workgroup's queue writes
    const uint32_t target = phase * gridDim.x;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) { __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never hang the device
    }
    __threadfence();                                   // acquire the other workgroups' writes
  }
  __syncthreads();
}

template <bool COUNT, bool WIDE, bool BATCH = false>
__global__ __launch_bounds__(256) void k_tail(TailArgs t) {
  uint32_t phase = 0;
  const uint32_t max_bounce = t.sh.u.max_bounce_count;
  for (uint32_t b = t.first_bounce; b <= max_bounce; b++) {
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < N_SHARDS; k++) live += ld_cursor(t.tr.counters + cnt_tail((int)b, k));
    if (live == 0 || ld_cursor(t.fault) != 0u) break;   // uniform over the grid: the queue is final since the last barrier
    TraceArgs a = t.tr;
    a.ray_o = t.sh.f.ray_o[b & 1u]; a.ray_d = t.sh.f.ray_d[b & 1u];
    a.tails = t.tr.counters + cnt_tail((int)b, 0);
    a.work = t.tr.counters + cnt_work((int)b, 0);
    trace_body<MODE_CLOSEST, false, COUNT, WIDE>(a);
    grid_barrier(t.barrier, t.fault, phase);
    // A barrier that gave up means some workgroup may still be tracing: its hit records are not final, so nobody may
    // shade them.  Every workgroup leaves as soon as it sees the flag (the host re-renders the frame without k_tail).
    if (ld_cursor(t.fault) != 0u) return;
    shade_body<false, BATCH>(t.sh, (int)b);
    grid_barrier(t.barrier, t.fault, phase);
    if (ld_cursor(t.fault) != 0u) return;
  }
}

// ------------------------------------------------------------------------------------------------
// k_resolve: src/shader.rgen:64,180-185 — ordered sum over samples, divide, store.
__global__ __launch_bounds__(256) void k_resolve(FrameDev f, UniformsDev u) {
  const uint32_t npx1 = (uint32_t)(f.rows * f.width), npx = npx1 * (uint32_t)f.batch_k;   // (a frame batch: the frames back to back)
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t fi = f.batch_k > 1 ? p / npx1 : 0u, p1 = p - fi * npx1;
  const size_t s0 = (size_t)fi * u.samples_per_pixel * npx1 + p1;   // sample 0 of this pixel
  if (p < npx && f.sample_color[s0].w != PIXEL_DONE) {   // (PIXEL_DONE: every sample of the pixel was a miss and k_raygen stored the pixel)
    float r = 0.f, g = 0.f, b = 0.f, al = 0.f;
    for (uint32_t i = 0; i < u.samples_per_pixel; i++) {
      const float4 c = ld_stream(&f.sample_color[s0 + (size_t)i * npx1]);
      r += c.x; g += c.y; b += c.z; al += c.w;
    }
    const float nn = (float)u.samples_per_pixel;
    store_pixel(f, fi * f.out_frame_stride + p1, make_float4(r / nn, g / nn, b / nn, al / nn));
  }
  // The next frame of this context finds its counters zeroed (no memset dispatch per frame): it uses the other block.
  if (f.counters_next)
    for (uint32_t i = p; i < (uint32_t)CNT_WORDS; i += gridDim.x * blockDim.x) f.counters_next[i] = 0u;
  if (f.cover_next)   // and its coverage mask cleared
    for (uint32_t i = p; i < f.cover_words; i += gridDim.x * blockDim.x) f.cover_next[i] = 0u;
  // This is the last kernel of the frame, so every counter is final: workgroup 0 condenses them into the host-mapped
  // statistics block (rt_device.h StatSlot).
  if (blockIdx.x == 0 && f.stats_out) {
    __shared__ unsigned long long s_q[N_QUEUES];
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)N_QUEUES) {
      unsigned long long n = 0;
      for (int k = 0; k < N_SHARDS; k++) n += ld_cursor(f.counters + cnt_tail((int)t, k));
      s_q[t] = n;
    }
    __syncthreads();
    // launch-strategy hint for the next frame of this context: the size of every bounce queue
    if (f.hint && t >= 1u && t < (uint32_t)CNT_MAX_BOUNCES)
      __hip_atomic_store(f.hint + t, (uint32_t)(s_q[t] > 0xFFFFFFFFull ? 0xFFFFFFFFull : s_q[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // every slot is written by its own thread: the stores cross PCIe and would cost ~0.3 us each one after another
    auto cnt64 = [&](int word) { return (unsigned long long)ld_cursor(f.counters + word) | ((unsigned long long)ld_cursor(f.counters + word + 1) << 32); };
    if (t < (uint32_t)STAT_WORDS) {
      unsigned long long v = 0;
      if (t == STAT_QUEUE0) { v = s_q[0] + s_q[Q_TILE_RAYS] - s_q[Q_DEAD]; for (int k = 0; k < N_SHARDS; k++) v -= ld_cursor(f.counters + cnt_work(Q_TILE_RAYS, k)); }   // (a ray handed on counts once)
      else if (t == STAT_TILE_RAYS) v = s_q[Q_TILE_RAYS];
      else if (t == STAT_CONT_RAYS) { for (int k = 0; k < N_SHARDS; k++) v += ld_cursor(f.counters + cnt_work(Q_TILE_RAYS, k)); }
      else if (t >= STAT_BLOB && t < STAT_BLOB + 5) v = ld_cursor(f.counters + CNT_BLOB_STATS + (int)(t - STAT_BLOB));
      else if (t >= STAT_TILE_DIAG && t < STAT_TILE_DIAG + 6) v = cnt64(CNT_TILE_DIAG + 2 * (int)(t - STAT_TILE_DIAG));
      else if (t == STAT_SECONDARY) { for (uint32_t b = 1; b <= u.max_bounce_count && b < (uint32_t)CNT_MAX_BOUNCES; b++) v += s_q[b]; }
      else if (t == STAT_SHADOW) v = s_q[Q_SHADOW] + s_q[Q_SHADOW0];
      else if (t == STAT_SHADOW_UNTRACED) { for (int k = 0; k < N_SHARDS; k++) v += ld_cursor(f.counters + cnt_work(Q_DEAD, k)); }
      else if (t == STAT_QUEUE1) v = s_q[1];
      else if (t == STAT_FAULT) v = ld_cursor(f.counters + CNT_FAULT);
      else if (t == STAT_FAULT_TOTAL) {
        // sticky: the frames of a context run one after another on its stream, so this is the only writer at any time
        uint32_t total = f.fault_total ? f.fault_total[0] : 0u;
        if (f.fault_total && ld_cursor(f.counters + CNT_FAULT) != 0u) { total++; f.fault_total[0] = total; }
        v = total;
      }
      else if (t == STAT_NODE_VISITS) v = cnt64(CNT_NODE_VISITS);
      else if (t == STAT_TRI_TESTS) v = cnt64(CNT_TRI_TESTS);
      else if (t == STAT_NODE_VISITS_SH) v = cnt64(CNT_NODE_VISITS_SH);
      else if (t == STAT_TRI_TESTS_SH) v = cnt64(CNT_TRI_TESTS_SH);
      else if (t >= STAT_DIAG && t < STAT_DIAG + 3) v = cnt64(CNT_DIAG + 2 * (int)(t - STAT_DIAG));
      else if (t >= STAT_DIAG + 3 && t < STAT_DIAG + 6) v = cnt64(CNT_DIAG_SH + 2 * (int)(t - STAT_DIAG - 3));
      __hip_atomic_store(f.stats_out + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
int trace_threads_per_block() { return 256; }

// k_raygen grid: one workgroup per (8x8 tile, group of up to 4 samples)
static dim3 raygen_grid(int width, int rows, uint32_t spp, int batch_k, dim3& block) {
  const uint32_t wpb = spp < 4u ? spp : 4u;
  block = dim3(64, wpb);
  return dim3(((uint32_t)width + 7u) >> 3, (((uint32_t)rows + 7u) >> 3) * (uint32_t)batch_k, (spp + wpb - 1u) / wpb);
}
size_t raygen_block_count(int width, int rows, uint32_t spp, int batch_k) {
  dim3 b; const dim3 g = raygen_grid(width, rows, spp, batch_k, b);
  return (size_t)g.x * g.y * g.z;
}
void launch_raygen(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, hipStream_t s) {
  dim3 b; const dim3 g = raygen_grid(f.width, f.rows, u.samples_per_pixel, f.batch_k, b);
  hipLaunchKernelGGL(k_raygen, g, b, 0, s, sc, f, u, bt);
}
size_t jitter_table_elems(int width, int rows, uint32_t spp) {
  return (size_t)(((uint32_t)width + 7u) >> 3) * (size_t)(((uint32_t)rows + 7u) >> 3) * spp * 64u;
}
void launch_jitter_table(const FrameDev& f, uint32_t spp, float2* table, hipStream_t s) {
  dim3 b; const dim3 g = raygen_grid(f.width, f.rows, spp, 1, b);
  hipLaunchKernelGGL(k_jitter_table, g, b, 0, s, f, spp, table);
}

void launch_cover(const SceneDev& sc, const CoverViews& a, uint32_t max_boxes_per_instance, uint32_t* mask_block, hipStream_t s) {
  if (a.n <= 0 || a.v[0].n_inst <= 0 || max_boxes_per_instance == 0) return;
  hipLaunchKernelGGL(k_cover, dim3((max_boxes_per_instance + 255u) / 256u, (unsigned)a.v[0].n_inst, (unsigned)a.n), dim3(256), 0, s, sc, a, mask_block);
}

static TraceArgs make_args(const SceneDev& sc, uint32_t* counters, int queue, uint32_t shard_cap, int32_t* ovf) {
  TraceArgs a{};
  a.sc = sc;
  a.tmin = 0.001f;  // src/shader.rgen:87,112
  a.counters = counters;
  a.tails = counters + cnt_tail(queue, 0);
  a.work = counters + cnt_work(queue, 0);
  a.shard_cap = shard_cap;
  a.ovf_stack = ovf;
  a.rays_per_lane = 4; a.min_blocks = 8;
  return a;
}

template <int MODE, bool ANY>
static void launch_trace(const TraceArgs& a_in, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = a_in;
  a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
  const dim3 g(cfg.trace_blocks), b(256);
#ifdef RT_ALT_KERNELS
  if (cfg.variant == 2) {
    if (counting) hipLaunchKernelGGL((k_trace_count<MODE, ANY, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace<MODE, ANY, true>), g, b, 0, s, a);
    return;
  }
  if (cfg.variant == 1) {
    if (counting) hipLaunchKernelGGL((k_trace4<MODE, ANY, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace4<MODE, ANY, false>), g, b, 0, s, a);
    return;
  }
#endif
  // (the product library has the one-lane BVH2 kernel only; rt_set_param refuses the other variants there: alt_kernels_built())
  if (counting) hipLaunchKernelGGL((k_trace_count<MODE, ANY, false>), g, b, 0, s, a);
  else if (cfg.far || MODE == MODE_RAW) hipLaunchKernelGGL((k_trace<MODE, ANY, false, false, true>), g, b, 0, s, a);
  else hipLaunchKernelGGL((k_trace<MODE, ANY, false, false, false>), g, b, 0, s, a);
}

bool alt_kernels_built() {
#ifdef RT_ALT_KERNELS
  return true;
#else
  return false;
#endif
}

void launch_trace_closest(const SceneDev& sc, const FrameDev& f, int bounce, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc, f.counters, bounce, f.shard_cap, f.ovf_stack);
  a.ray_o = f.ray_o[bounce & 1]; a.ray_d = f.ray_d[bounce & 1];
  a.hit_a = f.hit_a; a.hit_inst = f.hit_inst;
#ifdef RT_ALT_KERNELS
  if (bounce == 0 && cfg.packet != 0 && cfg.variant == 0) {
    // one wavefront per chunk of primary rays (k_packet); with entry records the walk of each tile's rays starts at its record
    a.entry = f.entry;
    a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
    const dim3 g(cfg.packet_blocks), b(256);
    if (f.entry != nullptr) {
      if (counting) hipLaunchKernelGGL((k_packet<MODE_CLOSEST, false, true, true>), g, b, 0, s, a);
      else hipLaunchKernelGGL((k_packet<MODE_CLOSEST, false, false, true>), g, b, 0, s, a);
    } else {
      if (counting) hipLaunchKernelGGL((k_packet<MODE_CLOSEST, false, true, false>), g, b, 0, s, a);
      else hipLaunchKernelGGL((k_packet<MODE_CLOSEST, false, false, false>), g, b, 0, s, a);
    }
    return;
  }
#endif
  if (bounce == 0 && f.pixel_runs && cfg.variant == 0) {   // one walk per pixel (kernels_beam.inc); f.pixel_runs = slots per run
    a.entry = f.entry;   // (NULL: the walks start at the TLAS root)
    const dim3 g(cfg.trace_blocks), b(256);
    if (counting) hipLaunchKernelGGL(k_beam_count, g, b, 0, s, a, (uint32_t)f.pixel_runs);
    else hipLaunchKernelGGL(k_beam, g, b, 0, s, a, (uint32_t)f.pixel_runs);
    return;
  }
  if (bounce == 0 && f.entry != nullptr && cfg.variant == 0) {
    // primary rays start at their tile's entry record (k_entry)
    a.entry = f.entry;
    a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
    const dim3 g(cfg.trace_blocks), b(256);

    if (f.tile_blob != nullptr) {   // queue 0 may hold rays k_tile handed on (tile blobs are off in frames with far rays)
      if (counting) hipLaunchKernelGGL((k_trace_count<MODE_CLOSEST, false, false, true, false, true>), g, b, 0, s, a);
      else hipLaunchKernelGGL((k_trace<MODE_CLOSEST, false, false, true, false, true>), g, b, 0, s, a);
      return;
    }
    if (counting) hipLaunchKernelGGL((k_trace_count<MODE_CLOSEST, false, false, true>), g, b, 0, s, a);
    else if (cfg.far) hipLaunchKernelGGL((k_trace<MODE_CLOSEST, false, false, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace<MODE_CLOSEST, false, false, true, false>), g, b, 0, s, a);
    return;
  }
  launch_trace<MODE_CLOSEST, false>(a, counting, cfg, s);
}

void launch_blob(const SceneDev& sc, const EntryArgs& e, const FrameDev& f, bool counting, hipStream_t s) {
  const uint32_t n = (uint32_t)(e.tiles_x * e.tile_rows);
  if (n == 0 || f.tile_blob == nullptr) return;
  BlobArgs a{sc, e, f.tile_blob, f.blob_arena, f.blob_slots / N_SHARDS, f.blob_list, f.counters};
  if (counting) hipLaunchKernelGGL((k_blob<true>), dim3(n), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_blob<false>), dim3(n), dim3(64), 0, s, a);
}

template <int CLS>
static void launch_tile_class(const TileArgs& t, uint32_t blocks, bool counting, hipStream_t s) {
  if (counting) hipLaunchKernelGGL((k_tile<CLS, true>), dim3(blocks), dim3(256), 0, s, t);
  else hipLaunchKernelGGL((k_tile<CLS, false>), dim3(blocks), dim3(256), 0, s, t);
}
void launch_tile(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, bool counting, hipStream_t s) {
  if (f.tile_blob == nullptr) return;
  TileArgs t{};
  t.sc = sc; t.f = f; t.u = u; t.sub_slots = f.blob_slots / N_SHARDS; t.tmin = 0.001f;   // src/shader.rgen:87
  const uint32_t blocks = f.blob_slots / N_SHARDS * N_SHARDS;
  launch_tile_class<0>(t, blocks, counting, s);
  launch_tile_class<1>(t, blocks, counting, s);
  launch_tile_class<2>(t, blocks, counting, s);
  static const bool twice = getenv("RT_EXP_TILE_TWICE") != nullptr;   // experiment: the same launches again (warm TLB / caches): how much of k_tile is cold misses?
  if (twice) { launch_tile_class<0>(t, blocks, counting, s); launch_tile_class<1>(t, blocks, counting, s); launch_tile_class<2>(t, blocks, counting, s); }
}

void launch_entry(const SceneDev& sc, const EntryViews& a, hipStream_t s) {
  uint32_t n = 0;
  for (int v = 0; v < a.n; v++) n = max(n, (uint32_t)(a.v[v].tiles_x * a.v[v].tile_rows));
  if (n == 0) return;
  hipLaunchKernelGGL(k_entry, dim3((n + 7u) / 8u, (unsigned)a.n), dim3(64), 0, s, sc, a);   // 8 lanes per tile; blockIdx.y = view
}

void launch_trace_shadow(const SceneDev& sc, const FrameDev& f, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc, f.counters, Q_SHADOW, f.shard_cap, f.ovf_stack);
  a.ray_o = f.sh_o + f.sh_base; a.ray_d = f.sh_d + f.sh_base; a.sh_c = f.sh_c + f.sh_base;   // (sh_base: above the shadow runs of bounce 0, which k_beam_shadow walks)
  a.sample_color = f.sample_color;
#ifdef RT_ALT_KERNELS
  if (cfg.packet != 0 && cfg.variant == 0) {
    // one wavefront per chunk of shadow rays (k_packet); optionally from the records of the cube around the light
    const bool le = f.light_entry != nullptr && f.sh_e != nullptr;
    a.entry = le ? f.light_entry : nullptr; a.sh_e = le ? f.sh_e + f.sh_base : nullptr;
    a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
    const dim3 g(cfg.packet_blocks), b(256);
    if (le) {
      if (counting) hipLaunchKernelGGL((k_packet<MODE_SHADOW, true, true, true>), g, b, 0, s, a);
      else hipLaunchKernelGGL((k_packet<MODE_SHADOW, true, false, true>), g, b, 0, s, a);
    } else {
      if (counting) hipLaunchKernelGGL((k_packet<MODE_SHADOW, true, true, false>), g, b, 0, s, a);
      else hipLaunchKernelGGL((k_packet<MODE_SHADOW, true, false, false>), g, b, 0, s, a);
    }
    return;
  }
#endif
  if (f.light_entry != nullptr && f.sh_e != nullptr && cfg.variant == 0) {
    // shadow rays start at the record of their tile of the cube around the light (k_entry, k_shade)
    a.entry = f.light_entry; a.sh_e = f.sh_e + f.sh_base;
    a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
    const dim3 g(cfg.trace_blocks), b(256);
    if (counting) hipLaunchKernelGGL((k_trace_count<MODE_SHADOW, true, false, true>), g, b, 0, s, a);
    else if (cfg.far) hipLaunchKernelGGL((k_trace<MODE_SHADOW, true, false, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace<MODE_SHADOW, true, false, true, false>), g, b, 0, s, a);
    return;
  }
  launch_trace<MODE_SHADOW, true>(a, counting, cfg, s);
}

// the shadow rays of the primary hits, one walk per pixel (kernels_beam.inc): the runs of bounce queue 0 in the shadow arrays
void launch_beam_shadow(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  if (!f.shadow_runs || !f.pixel_runs) return;
  BeamShadowArgs A{};
  A.t = make_args(sc, f.counters, 0, f.shard_cap, f.ovf_stack);
  A.t.work = f.counters + cnt_work(Q_SHADOW0, 0);
  A.t.ray_o = f.sh_o; A.t.ray_d = f.sh_d; A.t.sh_c = f.sh_c; A.t.sample_color = f.sample_color;
  const bool le = f.light_entry != nullptr && f.sh_e != nullptr;
  A.t.entry = le ? f.light_entry : nullptr; A.t.sh_e = le ? f.sh_e : nullptr;
  for (int k = 0; k < 3; k++) A.light[k] = u.light_position[k];
  A.run = (uint32_t)f.pixel_runs;
  const dim3 g(cfg.trace_blocks), b(256);
  if (counting) hipLaunchKernelGGL(k_beam_shadow_count, g, b, 0, s, A);
  else hipLaunchKernelGGL(k_beam_shadow, g, b, 0, s, A);
}

void launch_trace_raw(const SceneDev& sc, const float4* ray_o, const float4* ray_d, HitRec* out, uint32_t shard_cap,
                      int32_t* ovf_stack, uint32_t* counters, bool any_hit, bool counting, const LaunchCfg& cfg, hipStream_t s) {
  TraceArgs a = make_args(sc, counters, 0, shard_cap, ovf_stack);
  a.ray_o = ray_o; a.ray_d = ray_d; a.raw_out = out;
#ifdef RT_ALT_KERNELS
  if (cfg.packet >= 2 && cfg.variant == 0) {
    a.rays_per_lane = (uint32_t)cfg.rays_per_lane; a.min_blocks = (uint32_t)cfg.min_blocks;
    const dim3 g(cfg.packet_blocks), b(256);
    if (any_hit) { if (counting) hipLaunchKernelGGL((k_packet<MODE_RAW, true, true, false>), g, b, 0, s, a); else hipLaunchKernelGGL((k_packet<MODE_RAW, true, false, false>), g, b, 0, s, a); }
    else { if (counting) hipLaunchKernelGGL((k_packet<MODE_RAW, false, true, false>), g, b, 0, s, a); else hipLaunchKernelGGL((k_packet<MODE_RAW, false, false, false>), g, b, 0, s, a); }
    return;
  }
#endif
  if (any_hit) launch_trace<MODE_RAW, true>(a, counting, cfg, s);
  else launch_trace<MODE_RAW, false>(a, counting, cfg, s);
}

int tail_blocks_per_cu() {
  // the smallest over the instantiations: any of them may be the one in flight (counting; 4-ary records in the alt build)
  int n = 1 << 30, v = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_tail<false, false>, 256, 0) != hipSuccess) return 0;
  n = min(n, v);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_tail<true, false>, 256, 0) != hipSuccess) return 0;
  n = min(n, v);
#ifdef RT_ALT_KERNELS
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_tail<false, true>, 256, 0) != hipSuccess) return 0;
  n = min(n, v);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_tail<true, true>, 256, 0) != hipSuccess) return 0;
  n = min(n, v);
#endif
  return n;
}

void launch_tail(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, int first_bounce, bool counting, const LaunchCfg& cfg, int tail_blocks, hipStream_t s) {
  TailArgs t{};
  t.tr = make_args(sc, f.counters, first_bounce, f.shard_cap, f.ovf_stack);
  t.tr.hit_a = f.hit_a; t.tr.hit_inst = f.hit_inst;
  t.tr.rays_per_lane = 1u; t.tr.min_blocks = 8u;   // few rays: one per lane, the bounce costs one ray lifetime
  t.sh = ShadeArgs{sc, f, u, first_bounce, bt};
  t.first_bounce = (uint32_t)first_bounce;
  t.barrier = f.counters + CNT_BARRIER; t.fault = f.counters + CNT_FAULT;
  const dim3 g((unsigned)tail_blocks), b(256);
#ifdef RT_ALT_KERNELS
  if (cfg.variant == 2) {
    if (counting) hipLaunchKernelGGL((k_tail<true, true>), g, b, 0, s, t);
    else hipLaunchKernelGGL((k_tail<false, true>), g, b, 0, s, t);
    return;
  }
#endif
  if (f.batch_k > 1) {
    if (counting) hipLaunchKernelGGL((k_tail<true, false, true>), g, b, 0, s, t);
    else hipLaunchKernelGGL((k_tail<false, false, true>), g, b, 0, s, t);
    return;
  }
  if (counting) hipLaunchKernelGGL((k_tail<true, false>), g, b, 0, s, t);
  else hipLaunchKernelGGL((k_tail<false, false>), g, b, 0, s, t);
}

void launch_shade(const SceneDev& sc, const FrameDev& f, const UniformsDev& u, const BatchTab& bt, int bounce, const LaunchCfg& cfg, hipStream_t s) {
  ShadeArgs a{sc, f, u, bounce, bt};
  if (f.batch_k > 1) hipLaunchKernelGGL((k_shade<false, true>), dim3(cfg.shade_blocks), dim3(256), 0, s, a);
  else if (bounce == 0 && f.shadow_runs) hipLaunchKernelGGL((k_shade<false, false, true>), dim3(cfg.shade_blocks), dim3(256), 0, s, a);
  else if (bounce == 0 && f.tile_blob != nullptr) hipLaunchKernelGGL((k_shade<true, false>), dim3(cfg.shade_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_shade<false, false>), dim3(cfg.shade_blocks), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// k_assemble: the de-interleave step after the multi-GPU gather (the reference's image copy to the swapchain,
// src/main.cpp:2683-2686, is the single-GPU analogue).  `gathered` holds n_shards compact shards back to back
// (shard_stride units each: band after band of shard s = bands s, s + n, s + 2n, ...); row y of the frame is row
// (y / band_rows / n_shards) * band_rows + y % band_rows of shard (y / band_rows) % n_shards.  One thread per 16-byte
// (RGBA32F pixel) or 4-byte (RGBA8 pixel) unit: coalesced row copies, no arithmetic on the pixels.
template <typename T>
__global__ __launch_bounds__(256) void k_assemble(const T* __restrict__ gathered, T* __restrict__ out, uint32_t width, uint32_t height,
                                                   uint32_t band_rows, uint32_t n_shards, size_t shard_stride) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= width || y >= height) return;
  const uint32_t band = y / band_rows, within = y - band * band_rows;
  const uint32_t shard = band % n_shards, local = (band / n_shards) * band_rows + within;
  out[(size_t)y * width + x] = gathered[(size_t)shard * shard_stride + (size_t)local * width + x];
}

void launch_assemble(const void* gathered, void* out, int width, int height, int band_rows, int n_shards, size_t shard_stride_px, bool rgba8, hipStream_t s) {
  const dim3 g(((uint32_t)width + 255u) / 256u, (uint32_t)height), b(256);
  if (rgba8) hipLaunchKernelGGL((k_assemble<uint32_t>), g, b, 0, s, (const uint32_t*)gathered, (uint32_t*)out, (uint32_t)width, (uint32_t)height, (uint32_t)band_rows, (uint32_t)n_shards, shard_stride_px);
  else hipLaunchKernelGGL((k_assemble<float4>), g, b, 0, s, (const float4*)gathered, (float4*)out, (uint32_t)width, (uint32_t)height, (uint32_t)band_rows, (uint32_t)n_shards, shard_stride_px);
}

void launch_resolve(const FrameDev& f, const UniformsDev& u, hipStream_t s) {
  const uint32_t npx = (uint32_t)(f.rows * f.width) * (uint32_t)f.batch_k;
  hipLaunchKernelGGL(k_resolve, dim3((npx + 255u) / 256u), dim3(256), 0, s, f, u);
}

}  // namespace rt
