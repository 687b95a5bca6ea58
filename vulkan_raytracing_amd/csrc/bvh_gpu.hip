// bvh_gpu.hip — BLAS build on the GPU (what the driver does behind vkCmdBuildAccelerationStructuresKHR in
// the reference, src/main.cpp:495-498, with VK_ACCELERATION_STRUCTURE_BUILD_TYPE_DEVICE_KHR, :345-357).
//
// Linear BVH: triangle boxes + bounds -> 30-bit Morton codes of the centroids -> radix sort (rocPRIM through
// hipcub; the sort is plumbing) -> binary radix tree built in parallel, one thread per internal node (Karras,
// "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012) -> bottom-up box
// propagation with one atomic arrival flag per node -> emit: subtrees of <= max_leaf triangles become leaves (default 1), every
// surviving internal node is written as a 32-byte quantized BvhNodeQ (rt_device.h), triangles as 48-byte packets
// in sorted order.  Output indices are local to the mesh (root = node 0); rt_api.cpp rebases them when linking.
//
// Any valid BVH yields the same hits (the tie rule makes results independent of traversal order), so images from
// this builder are bit-identical to those from the host SAH builder — tested in tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <string>

#include "bvh_gpu.h"

namespace rt {
namespace {

// monotone float <-> uint mapping so that atomicMin/atomicMax on uints order floats
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

struct Box { float lo[3], hi[3]; };

__global__ void k_init_bounds(uint32_t* b) {
  if (threadIdx.x < 3) b[threadIdx.x] = 0xFFFFFFFFu;       // min accumulators
  else if (threadIdx.x < 6) b[threadIdx.x] = 0u;           // max accumulators
}

// per triangle: box, centroid bounds (block-reduced, then 6 atomics per block)
__global__ __launch_bounds__(256) void k_tri_boxes(const float* verts6, const uint32_t* idx, uint32_t n, Box* boxes, uint32_t* cbounds) {
  __shared__ uint32_t s_b[6];
  if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    Box b;
    for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; }
    for (int c = 0; c < 3; c++) {
      const float* v = verts6 + 6ull * idx[3ull * p + c];
      for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], v[k]); b.hi[k] = fmaxf(b.hi[k], v[k]); }
    }
    boxes[p] = b;
    for (int k = 0; k < 3; k++) {
      const float cen = 0.5f * b.lo[k] + 0.5f * b.hi[k];
      atomicMin(&s_b[k], f2ord(cen)); atomicMax(&s_b[3 + k], f2ord(cen));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&cbounds[threadIdx.x], s_b[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&cbounds[threadIdx.x], s_b[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {   // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(256) void k_morton(const Box* boxes, uint32_t n, const uint32_t* cbounds, uint32_t* keys, uint32_t* vals) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t code = 0;
  for (int k = 0; k < 3; k++) {
    const float lo = ord2f(cbounds[k]), hi = ord2f(cbounds[3 + k]);
    const float ext = hi - lo;
    const float cen = 0.5f * boxes[p].lo[k] + 0.5f * boxes[p].hi[k];
    float t = ext > 0.f ? (cen - lo) / ext : 0.f;
    t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
    code |= spread3((uint32_t)t) << (2 - k);
  }
  keys[p] = code; vals[p] = p;
}

// common-prefix length of sorted keys i and j (ties broken by the index), -1 outside the array
__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = keys[i], b = keys[j];
  if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clz(a ^ b);
}

// One thread per internal node i in [0, n-1): range, split, children, parents (Karras 2012, algorithm 1).
// child encoding here: >= 0 internal node, < 0 leaf ~sorted_index
__global__ __launch_bounds__(256) void k_radix_tree(const uint32_t* keys, int n, int2* children, int2* ranges, int* parent_internal, int* parent_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const int left = (lo == gamma) ? ~gamma : gamma;
  const int right = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
  children[i] = make_int2(left, right);
  ranges[i] = make_int2(lo, hi);
  if (left >= 0) parent_internal[left] = i; else parent_leaf[~left] = i;
  if (right >= 0) parent_internal[right] = i; else parent_leaf[~right] = i;
  if (i == 0) parent_internal[0] = -1;
}

// bottom-up: each leaf climbs; the second arrival at a node merges the children's boxes and continues
__device__ __forceinline__ Box box_union(const Box& a, const Box& b) {
  Box m;
  for (int k = 0; k < 3; k++) { m.lo[k] = fminf(a.lo[k], b.lo[k]); m.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
  return m;
}
__device__ __forceinline__ float box_half_area(const Box& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

// Bottom-up box propagation: leaves climb, the second arrival at a node owns the finished subtree below it.
// ROTATE: before a node's box is stored, the owner tries the four tree rotations that exchange one child with a
// grandchild of the other side (Kensler 2008) and keeps the one that shrinks the surface area of the re-formed
// child most — the SAH cost of a tree with one triangle per leaf is the sum of its internal nodes' areas, and a
// rotation changes exactly one of them.  Nothing outside the owned subtree is touched, so the pass is race free.
template <bool ROTATE>
__global__ __launch_bounds__(256) void k_propagate(const Box* tri_boxes, const uint32_t* sorted_ids, int n, int2* children, int* parent_internal,
                                                   int* parent_leaf, Box* node_boxes, uint32_t* flags) {
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n) return;
  auto box_of = [&](int ref) -> Box { return ref >= 0 ? node_boxes[ref] : tri_boxes[sorted_ids[~ref]]; };
  auto set_parent = [&](int ref, int p) { if (ref >= 0) parent_internal[ref] = p; else parent_leaf[~ref] = p; };
  int node = parent_leaf[leaf];
  while (node >= 0) {
    __threadfence();
    if (atomicAdd(&flags[node], 1u) == 0u) return;   // first arrival: the sibling subtree is not finished yet
    __threadfence();
    int2 ch = children[node];
    // (the acquire fence above invalidated this CU's L1, so plain loads see the sibling subtree's boxes)
    Box a = box_of(ch.x), b = box_of(ch.y);
    if (ROTATE) {
      float best = -1e-7f * box_half_area(box_union(a, b));   // only real improvements
      int pick = 0;
      Box nb{}, nb2{};   // boxes of the re-formed child(ren)
      int2 l = make_int2(0, 0), r = make_int2(0, 0);
      if (ch.y >= 0) {
        r = children[ch.y];
        const float old = box_half_area(b);
        const Box c1 = box_union(a, box_of(r.y)), c2 = box_union(a, box_of(r.x));   // left <-> right.x / right.y
        const float d1 = box_half_area(c1) - old, d2 = box_half_area(c2) - old;
        if (d1 < best) { best = d1; pick = 1; nb = c1; }
        if (d2 < best) { best = d2; pick = 2; nb = c2; }
      }
      if (ch.x >= 0) {
        l = children[ch.x];
        const float old = box_half_area(a);
        const Box c3 = box_union(b, box_of(l.y)), c4 = box_union(b, box_of(l.x));   // right <-> left.x / left.y
        const float d3 = box_half_area(c3) - old, d4 = box_half_area(c4) - old;
        if (d3 < best) { best = d3; pick = 3; nb = c3; }
        if (d4 < best) { best = d4; pick = 4; nb = c4; }
      }
      if (ch.x >= 0 && ch.y >= 0) {
        // grandchild <-> grandchild: the pairings {LL,RL}+{LR,RR} and {LL,RR}+{RL,LR}; two boxes change at once
        const Box bll = box_of(l.x), blr = box_of(l.y), brl = box_of(r.x), brr = box_of(r.y);
        const float old = box_half_area(a) + box_half_area(b);
        const Box p5 = box_union(bll, brl), q5 = box_union(blr, brr);
        const Box p6 = box_union(bll, brr), q6 = box_union(brl, blr);
        const float d5 = box_half_area(p5) + box_half_area(q5) - old, d6 = box_half_area(p6) + box_half_area(q6) - old;
        if (d5 < best) { best = d5; pick = 5; nb = p5; nb2 = q5; }
        if (d6 < best) { best = d6; pick = 6; nb = p6; nb2 = q6; }
      }
      if (pick == 5 || pick == 6) {
        // left keeps LL and takes RL (5) or RR (6); right takes LR and keeps the other one
        const int take = pick == 5 ? r.x : r.y, keep = pick == 5 ? r.y : r.x;
        children[ch.x] = make_int2(l.x, take); set_parent(take, ch.x);
        children[ch.y] = pick == 5 ? make_int2(l.y, keep) : make_int2(keep, l.y); set_parent(l.y, ch.y);
        node_boxes[ch.x] = nb; node_boxes[ch.y] = nb2;
      } else if (pick == 1 || pick == 2) {
        // the left child goes down into the right child, the right child's x (pick 1) or y (pick 2) comes up
        const int up = pick == 1 ? r.x : r.y, stay = pick == 1 ? r.y : r.x;
        children[ch.y] = make_int2(ch.x, stay); set_parent(ch.x, ch.y);
        node_boxes[ch.y] = nb;
        ch = make_int2(up, ch.y); set_parent(up, node);
        children[node] = ch;
      } else if (pick == 3 || pick == 4) {
        const int up = pick == 3 ? l.x : l.y, stay = pick == 3 ? l.y : l.x;
        children[ch.x] = make_int2(stay, ch.y); set_parent(ch.y, ch.x);
        node_boxes[ch.x] = nb;
        ch = make_int2(ch.x, up); set_parent(up, node);
        children[node] = ch;
      }
      if (pick) { a = box_of(ch.x); b = box_of(ch.y); }
    }
    node_boxes[node] = box_union(a, b);
    node = parent_internal[node];
  }
}

__global__ void k_quant_params(const Box* node_boxes, float* qparams /* lo[3], scale[3], bounds lo[3], hi[3] */) {
  if (threadIdx.x >= 3) return;
  const int k = threadIdx.x;
  const float lo = node_boxes[0].lo[k], hi = node_boxes[0].hi[k];
  const float ext = hi - lo;
  const float scale = ext > 0.f ? ext * 1.00001f / 65520.0f : 1e-30f;
  qparams[k] = lo - 4.0f * scale;       // quanta 0..3 stay below every stored plane
  qparams[3 + k] = scale;
  qparams[6 + k] = lo; qparams[9 + k] = hi;
}

__device__ __forceinline__ uint32_t quant_box_axis(float lo, float hi, float base, float scale) {
  // two quanta of margin on each side cover the float rounding of the division
  float ql = floorf((lo - base) / scale) - 2.0f, qh = ceilf((hi - base) / scale) + 2.0f;
  ql = fminf(fmaxf(ql, 0.0f), 65535.0f); qh = fminf(fmaxf(qh, 0.0f), 65535.0f);
  return (uint32_t)ql | ((uint32_t)qh << 16);
}

// emit: internal node i with more than 4 triangles becomes BvhNodeQ[i]; children with <= 4 triangles become leaves
__global__ __launch_bounds__(256) void k_emit_nodes(const Box* tri_boxes, const uint32_t* sorted_ids, int n, const int2* children, const int2* ranges,
                                                    const Box* node_boxes, const float* qparams, BvhNodeQ* out, int max_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int2 r = ranges[i];
  BvhNodeQ q{};
  if (r.y - r.x + 1 > max_leaf) {
    const int2 ch = children[i];
    int refs[2]; Box bx[2];
    const int c[2] = {ch.x, ch.y};
    for (int k = 0; k < 2; k++) {
      if (c[k] >= 0) {
        const int2 cr = ranges[c[k]];
        const int cnt = cr.y - cr.x + 1;
        bx[k] = node_boxes[c[k]];
        refs[k] = cnt <= max_leaf ? ~(int)(((uint32_t)cr.x << 3) | (uint32_t)(cnt - 1)) : c[k];
      } else {
        const int leaf = ~c[k];
        bx[k] = tri_boxes[sorted_ids[leaf]];
        refs[k] = ~(int)(((uint32_t)leaf << 3) | 0u);
      }
    }
    for (int a = 0; a < 3; a++) {
      q.w[a] = quant_box_axis(bx[0].lo[a], bx[0].hi[a], qparams[a], qparams[3 + a]);
      q.w[3 + a] = quant_box_axis(bx[1].lo[a], bx[1].hi[a], qparams[a], qparams[3 + a]);
    }
    q.child0 = refs[0]; q.child1 = refs[1];
  }
  out[i] = q;
}

__global__ __launch_bounds__(256) void k_emit_tris(const float* verts6, const uint32_t* idx, const uint32_t* sorted_ids, uint32_t n, float4* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = sorted_ids[i];
  const float* v0 = verts6 + 6ull * idx[3ull * p + 0];
  const float* v1 = verts6 + 6ull * idx[3ull * p + 1];
  const float* v2 = verts6 + 6ull * idx[3ull * p + 2];
  // e1 = v1 - v0, e2 = v2 - v0 rounded once in binary32, exactly as the oracle and the host builder do
  const float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
  const float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
  out[3ull * i + 0] = make_float4(v0[0], v0[1], v0[2], e1x);
  out[3ull * i + 1] = make_float4(e1y, e1z, e2x, e2y);
  out[3ull * i + 2] = make_float4(e2z, __uint_as_float(p), 0.f, 0.f);
}

// copy a mesh's local nodes into the linked array, rebasing interior links and leaf ranges
__global__ __launch_bounds__(256) void k_rebase_nodes(const BvhNodeQ* src, BvhNodeQ* dst, uint32_t n, int node_base, uint32_t tri_base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  BvhNodeQ q = src[i];
  int* ch[2] = {&q.child0, &q.child1};
  for (int k = 0; k < 2; k++) {
    const int c = *ch[k];
    if (c >= 0) *ch[k] = c + node_base;
    else { const uint32_t ref = (uint32_t)(~c); *ch[k] = ~(int)((((ref >> 3) + tri_base) << 3) | (ref & 7u)); }
  }
  dst[i] = q;
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding
// Volume Hierarchy Construction", TVCG 2018).  Clusters start as the Morton-sorted triangles; every round each
// cluster looks at its 2r neighbours in the sorted order for the partner that minimises the surface area of the merged
// box, mutually-nearest pairs merge into a new node, survivors are compacted in order.  Bottom-up, so every node is
// born with the boxes of both children; the result is close to a top-down SAH tree at a fraction of its build time.
struct Cluster { Box box; int id; int pad; };        // id >= 0: internal node, < 0: leaf ~sorted position
struct PlocNode { Box b0, b1; int c0, c1; };          // 56 bytes

__device__ __forceinline__ float merged_area(const Box& a, const Box& b) {
  const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(256) void k_ploc_init(const Box* tri_boxes, const uint32_t* sorted_ids, int n, Cluster* c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Cluster cl; cl.box = tri_boxes[sorted_ids[i]]; cl.id = ~i; cl.pad = 0;
  c[i] = cl;
}

__global__ __launch_bounds__(256) void k_ploc_nn(const Cluster* c, int m, int r, int* nn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Box me = c[i].box;
  float best = 3.0e38f; int bj = i == 0 ? 1 : i - 1;
  const int lo = max(0, i - r), hi = min(m - 1, i + r);
  for (int j = lo; j <= hi; j++) {
    if (j == i) continue;
    const float a = merged_area(me, c[j].box);
    if (a < best) { best = a; bj = j; }     // ascending j: ties keep the smaller index, so the choice is deterministic
  }
  nn[i] = bj;
}

// flags: bit 0 = cluster survives (possibly as a merged one), bit 1 = this slot creates a new node
__global__ __launch_bounds__(256) void k_ploc_flags(int m, const int* nn, uint32_t* survive, uint32_t* creates) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = nn[j] == i;
  survive[i] = (mutual && i > j) ? 0u : 1u;
  creates[i] = (mutual && i < j) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_ploc_merge(const Cluster* c, int m, const int* nn, const uint32_t* survive, const uint32_t* pos, const uint32_t* creates,
                                                    const uint32_t* cpos, int node_base, PlocNode* nodes, Cluster* next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || !survive[i]) return;
  Cluster out = c[i];
  if (creates[i]) {
    const Cluster o = c[nn[i]];
    const int id = node_base + (int)cpos[i];          // ids follow the sorted order: the numbering is deterministic
    PlocNode nd; nd.b0 = out.box; nd.b1 = o.box; nd.c0 = out.id; nd.c1 = o.id;
    nodes[id] = nd;
    for (int k = 0; k < 3; k++) { out.box.lo[k] = fminf(out.box.lo[k], o.box.lo[k]); out.box.hi[k] = fmaxf(out.box.hi[k], o.box.hi[k]); }
    out.id = id;
  }
  next[pos[i]] = out;
}

__global__ void k_quant_params_ploc(const PlocNode* nodes, int root, float* qparams) {
  if (threadIdx.x >= 3) return;
  const int k = threadIdx.x;
  const float lo = fminf(nodes[root].b0.lo[k], nodes[root].b1.lo[k]), hi = fmaxf(nodes[root].b0.hi[k], nodes[root].b1.hi[k]);
  const float ext = hi - lo;
  const float scale = ext > 0.f ? ext * 1.00001f / 65520.0f : 1e-30f;
  qparams[k] = lo - 4.0f * scale;
  qparams[3 + k] = scale;
  qparams[6 + k] = lo; qparams[9 + k] = hi;
}

// node id k (creation order, root = n-2) -> array slot (n-2) - k, so the root lands in slot 0
__global__ __launch_bounds__(256) void k_emit_ploc(const PlocNode* nodes, int n_internal, const float* qparams, BvhNodeQ* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_internal) return;
  const PlocNode nd = nodes[k];
  BvhNodeQ q{};
  for (int a = 0; a < 3; a++) {
    q.w[a] = quant_box_axis(nd.b0.lo[a], nd.b0.hi[a], qparams[a], qparams[3 + a]);
    q.w[3 + a] = quant_box_axis(nd.b1.lo[a], nd.b1.hi[a], qparams[a], qparams[3 + a]);
  }
  q.child0 = nd.c0 >= 0 ? (n_internal - 1 - nd.c0) : ~(int)(((uint32_t)(~nd.c0) << 3) | 0u);
  q.child1 = nd.c1 >= 0 ? (n_internal - 1 - nd.c1) : ~(int)(((uint32_t)(~nd.c1) << 3) | 0u);
  out[n_internal - 1 - k] = q;
}

#define GB_TRY(expr)                                                                                             \
  do {                                                                                                           \
    hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess) { err = std::string("HIP runtime exception: return code ") + std::to_string((int)e_) + \
                                  " (" + hipGetErrorString(e_) + ") in " #expr; cleanup(); return 1; }            \
  } while (0)

}  // namespace

int build_blas_gpu(const float* d_verts6, const uint32_t* d_idx, uint32_t n, hipStream_t s, GpuBlas& out, std::string& err) {
  out = GpuBlas{};
  if (n < 8) { err = "build_blas_gpu needs at least 8 triangles"; return 1; }
  int max_leaf = 1;   // subtrees of at most this many triangles become leaves (1 measured best: 1.15 ms/frame vs 1.22 at 4)
  if (const char* e = getenv("RT_LBVH_MAX_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 8) max_leaf = v; }
  Box *tri_boxes = nullptr, *node_boxes = nullptr;
  uint32_t *cbounds = nullptr, *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *flags = nullptr;
  int2 *children = nullptr, *ranges = nullptr;
  int *parent_internal = nullptr, *parent_leaf = nullptr;
  float* qparams = nullptr;
  void* tmp = nullptr;
  bool done = false;   // set once the results in `out` are complete: until then cleanup() frees them as well
  auto cleanup = [&]() {
    for (void* p : {(void*)tri_boxes, (void*)node_boxes, (void*)cbounds, (void*)keys, (void*)keys2, (void*)vals, (void*)vals2, (void*)flags, (void*)children,
                    (void*)ranges, (void*)parent_internal, (void*)parent_leaf, (void*)qparams, tmp})
      if (p) hipFree(p);
    if (!done) {
      if (out.nodes) hipFree(out.nodes);
      if (out.tris) hipFree(out.tris);
      out.nodes = nullptr; out.tris = nullptr;
    }
  };
  const uint32_t nb = (n + 255u) / 256u;
  GB_TRY(hipMalloc((void**)&tri_boxes, n * sizeof(Box)));
  GB_TRY(hipMalloc((void**)&node_boxes, n * sizeof(Box)));
  GB_TRY(hipMalloc((void**)&cbounds, 6 * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&keys, n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&keys2, n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&vals, n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&vals2, n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&flags, n * sizeof(uint32_t)));
  GB_TRY(hipMalloc((void**)&children, n * sizeof(int2)));
  GB_TRY(hipMalloc((void**)&ranges, n * sizeof(int2)));
  GB_TRY(hipMalloc((void**)&parent_internal, n * sizeof(int)));
  GB_TRY(hipMalloc((void**)&parent_leaf, n * sizeof(int)));
  GB_TRY(hipMalloc((void**)&qparams, 12 * sizeof(float)));
  GB_TRY(hipMalloc((void**)&out.nodes, (size_t)(n - 1) * sizeof(BvhNodeQ)));
  GB_TRY(hipMalloc((void**)&out.tris, (size_t)n * 3 * sizeof(float4)));

  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, s, cbounds);
  hipLaunchKernelGGL(k_tri_boxes, dim3(nb), dim3(256), 0, s, d_verts6, d_idx, n, tri_boxes, cbounds);
  hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, s, tri_boxes, n, cbounds, keys, vals);
  size_t tmp_bytes = 0;
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 30, s));
  GB_TRY(hipMalloc(&tmp, tmp_bytes));
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 30, s));
  int algo = 1;   // 1: LBVH (Karras radix tree, default — 1.17 ms/frame on cfg3), 2: PLOC (1.20 ms/frame on this smooth mesh)
  if (const char* e = getenv("RT_GPU_BVH_ALGO")) { int v = atoi(e); if (v == 1 || v == 2) algo = v; }
  if (algo == 2) {
    int radius = 16;
    if (const char* e = getenv("RT_PLOC_RADIUS")) { int v = atoi(e); if (v >= 1 && v <= 128) radius = v; }
    Cluster *ca = nullptr, *cb = nullptr; PlocNode* pn = nullptr; int* nn = nullptr; uint32_t *sv = nullptr, *cr = nullptr, *ps = nullptr, *cp = nullptr; void* scan_tmp = nullptr;
    auto cleanup2 = [&]() { for (void* p : {(void*)ca, (void*)cb, (void*)pn, (void*)nn, (void*)sv, (void*)cr, (void*)ps, (void*)cp, scan_tmp}) if (p) hipFree(p); };
#define PL_TRY(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { err = std::string("HIP runtime exception: return code ") + std::to_string((int)e2_) + " in " #expr; cleanup2(); cleanup(); return 1; } } while (0)
    PL_TRY(hipMalloc((void**)&ca, n * sizeof(Cluster))); PL_TRY(hipMalloc((void**)&cb, n * sizeof(Cluster)));
    PL_TRY(hipMalloc((void**)&pn, (size_t)(n - 1) * sizeof(PlocNode))); PL_TRY(hipMalloc((void**)&nn, n * sizeof(int)));
    PL_TRY(hipMalloc((void**)&sv, n * sizeof(uint32_t))); PL_TRY(hipMalloc((void**)&cr, n * sizeof(uint32_t)));
    PL_TRY(hipMalloc((void**)&ps, n * sizeof(uint32_t))); PL_TRY(hipMalloc((void**)&cp, n * sizeof(uint32_t)));
    size_t scan_bytes = 0;
    PL_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, sv, ps, (int)n, s));
    PL_TRY(hipMalloc(&scan_tmp, scan_bytes));
    hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(256), 0, s, tri_boxes, vals2, (int)n, ca);
    int m = (int)n, node_base = 0, rounds = 0;
    while (m > 1) {
      const uint32_t mb = ((uint32_t)m + 255u) / 256u;
      hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(256), 0, s, ca, m, radius, nn);
      hipLaunchKernelGGL(k_ploc_flags, dim3(mb), dim3(256), 0, s, m, nn, sv, cr);
      PL_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, sv, ps, m, s));
      PL_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, cr, cp, m, s));
      hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(256), 0, s, ca, m, nn, sv, ps, cr, cp, node_base, pn, cb);
      uint32_t last[4];   // pos[m-1], survive[m-1], cpos[m-1], creates[m-1]
      PL_TRY(hipMemcpyAsync(&last[0], ps + (m - 1), 4, hipMemcpyDeviceToHost, s)); PL_TRY(hipMemcpyAsync(&last[1], sv + (m - 1), 4, hipMemcpyDeviceToHost, s));
      PL_TRY(hipMemcpyAsync(&last[2], cp + (m - 1), 4, hipMemcpyDeviceToHost, s)); PL_TRY(hipMemcpyAsync(&last[3], cr + (m - 1), 4, hipMemcpyDeviceToHost, s));
      PL_TRY(hipStreamSynchronize(s));
      const int created = (int)(last[2] + last[3]);
      if (created == 0) { err = "PLOC made no progress"; cleanup2(); cleanup(); return 1; }
      m = (int)(last[0] + last[1]); node_base += created;
      std::swap(ca, cb);
      if (++rounds > 4096) { err = "PLOC did not converge"; cleanup2(); cleanup(); return 1; }
    }
    if (node_base != (int)n - 1) { err = "PLOC produced " + std::to_string(node_base) + " nodes for " + std::to_string(n) + " triangles"; cleanup2(); cleanup(); return 1; }
    hipLaunchKernelGGL(k_quant_params_ploc, dim3(1), dim3(64), 0, s, pn, (int)n - 2, qparams);
    hipLaunchKernelGGL(k_emit_ploc, dim3(nb), dim3(256), 0, s, pn, (int)n - 1, qparams, out.nodes);
    hipLaunchKernelGGL(k_emit_tris, dim3(nb), dim3(256), 0, s, d_verts6, d_idx, vals2, n, out.tris);
    float h2[12];
    PL_TRY(hipMemcpyAsync(h2, qparams, sizeof(h2), hipMemcpyDeviceToHost, s));
    PL_TRY(hipStreamSynchronize(s));
    PL_TRY(hipGetLastError());
    for (int k = 0; k < 3; k++) { out.q_lo[k] = h2[k]; out.q_scale[k] = h2[3 + k]; out.bounds_lo[k] = h2[6 + k]; out.bounds_hi[k] = h2[9 + k]; }
    out.n_nodes = n - 1; out.n_tris = n;
    done = true;
    cleanup2(); cleanup();
    return 0;
#undef PL_TRY
  }
  GB_TRY(hipMemsetAsync(flags, 0, n * sizeof(uint32_t), s));
  hipLaunchKernelGGL(k_radix_tree, dim3(nb), dim3(256), 0, s, keys2, (int)n, children, ranges, parent_internal, parent_leaf);
  // box propagation; with one triangle per leaf, RT_LBVH_ROTATE passes (default 2) also apply tree rotations
  int rotate_passes = max_leaf == 1 ? 2 : 0;
  if (const char* e = getenv("RT_LBVH_ROTATE")) { const int v = atoi(e); if (v >= 0 && v <= 16 && max_leaf == 1) rotate_passes = v; }
  if (rotate_passes == 0)
    hipLaunchKernelGGL(k_propagate<false>, dim3(nb), dim3(256), 0, s, tri_boxes, vals2, (int)n, children, parent_internal, parent_leaf, node_boxes, flags);
  for (int pass = 0; pass < rotate_passes; pass++) {
    if (pass) GB_TRY(hipMemsetAsync(flags, 0, n * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_propagate<true>, dim3(nb), dim3(256), 0, s, tri_boxes, vals2, (int)n, children, parent_internal, parent_leaf, node_boxes, flags);
  }
  hipLaunchKernelGGL(k_quant_params, dim3(1), dim3(64), 0, s, node_boxes, qparams);
  hipLaunchKernelGGL(k_emit_nodes, dim3(nb), dim3(256), 0, s, tri_boxes, vals2, (int)n, children, ranges, node_boxes, qparams, out.nodes, max_leaf);
  hipLaunchKernelGGL(k_emit_tris, dim3(nb), dim3(256), 0, s, d_verts6, d_idx, vals2, n, out.tris);
  float h[12];
  GB_TRY(hipMemcpyAsync(h, qparams, sizeof(h), hipMemcpyDeviceToHost, s));
  GB_TRY(hipStreamSynchronize(s));
  GB_TRY(hipGetLastError());
  for (int k = 0; k < 3; k++) { out.q_lo[k] = h[k]; out.q_scale[k] = h[3 + k]; out.bounds_lo[k] = h[6 + k]; out.bounds_hi[k] = h[9 + k]; }
  out.n_nodes = n - 1; out.n_tris = n;
  done = true;
  cleanup();
  return 0;
}

void free_blas_gpu(GpuBlas& b) {
  if (b.nodes) hipFree(b.nodes);
  if (b.tris) hipFree(b.tris);
  b = GpuBlas{};
}

void launch_rebase_nodes(const BvhNodeQ* src, BvhNodeQ* dst, uint32_t n, int node_base, uint32_t tri_base, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_rebase_nodes, dim3((n + 255u) / 256u), dim3(256), 0, s, src, dst, n, node_base, tri_base);
}

}  // namespace rt
