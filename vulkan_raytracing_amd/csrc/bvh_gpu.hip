// bvh_gpu.hip — BLAS build on the GPU (what the driver does behind vkCmdBuildAccelerationStructuresKHR in
// the reference, src/main.cpp:495-498, with VK_ACCELERATION_STRUCTURE_BUILD_TYPE_DEVICE_KHR, :345-357).
//
// Three builders over the same front end (triangle boxes + centroid bounds) and back end (32-byte quantized BvhNodeQ nodes,
// 48-byte triangle packets in leaf order, one triangle per leaf; indices local to the mesh, root = node 0, rt_api.cpp rebases them
// when linking):
//   3 (default) binned SAH, top-down, level by level — "k_sah_*" below: the tree quality of the host builder (csrc/bvh_build.cpp)
//     at device speed; frames render 2-3 % faster than from the LBVH tree (profiles/r03_experiments.txt);
//   1 LBVH: 30-bit Morton codes of the centroids -> radix sort (rocPRIM through hipcub; the sort is plumbing) -> binary radix
//     tree, one thread per internal node (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees",
//     HPG 2012) -> bottom-up box propagation with one atomic arrival flag per node and SAH-driven tree rotations -> emit;
//   2 PLOC (Meister & Bittner 2018) over the same Morton order.
// RT_GPU_BVH_ALGO selects; RT_LBVH_MAX_LEAF > 1 (leaves of several triangles) exists for the LBVH only.
//
// Any valid BVH yields the same hits (the tie rule makes results independent of traversal order), so images from
// this builder are bit-identical to those from the host SAH builder — tested in tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
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

// ---- binned SAH, top-down, level by level (RT_GPU_BVH_ALGO=3).  The tree the host builder of bvh_build.cpp makes (32 bins per axis,
// exact sweep for nodes of at most SAH_SWEEP references, one triangle per leaf), built breadth-first on the device: all nodes of a
// level are split together.  References live in one array in which every node owns a contiguous segment; a level is
//   k_sah_level_flags + two scans   which nodes split / which are binned: child numbers and bin-block numbers by prefix sum (deterministic)
//   k_sah_bin                       one thread per reference of a binned node: 3 axes x (count, box) into the node's bins, atomics
//                                   (order-independent quantities only: the tree is the same in every run)
//   k_sah_split                     one thread per node: binned nodes evaluate the 3 x 31 split candidates from their bins; small nodes
//                                   sort their <= 16 references per axis in scratch, sweep every split position and write their two
//                                   sub-segments themselves; both create their children (segment, box)
//   k_sah_side + scan + k_sah_scatter   binned nodes: stable partition of the segment by a prefix sum over "goes left"; the centroid
//                                   bounds of the children are accumulated on the way (ordered-uint atomic min/max)
// with ONE 8-byte read-back per level (how many nodes split, how many are binned).  Nodes are numbered level by level, so the array
// starts with the top of the tree.  After level 48 every split is a position median (bounded depth on adversarial input).
constexpr int SAH_BINS = 32, SAH_SWEEP = 16, SAH_BIN_WORDS = 3 * SAH_BINS * 7;   // per bin: count, lo[3], hi[3] (ordered uints)

__global__ __launch_bounds__(256) void k_sah_root(const Box* tri_boxes, uint32_t n, uint32_t* root_box /* 6 ordered words */, uint32_t* ids, int* pos_node) {
  __shared__ uint32_t s_b[6];
  if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    ids[p] = p; pos_node[p] = 0;
    const Box b = tri_boxes[p];
    for (int k = 0; k < 3; k++) { atomicMin(&s_b[k], f2ord(b.lo[k])); atomicMax(&s_b[3 + k], f2ord(b.hi[k])); }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&root_box[threadIdx.x], s_b[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&root_box[threadIdx.x], s_b[threadIdx.x]);
}

__global__ void k_sah_root_node(const uint32_t* root_box, const uint32_t* cbounds, uint32_t n, Box* nbox, uint32_t* ncb, int2* nseg, int2* nkids) {
  if (threadIdx.x == 0) { nseg[0] = make_int2(0, (int)n); nkids[0] = make_int2(-1, -1); }
  if (threadIdx.x < 3) { nbox[0].lo[threadIdx.x] = ord2f(root_box[threadIdx.x]); nbox[0].hi[threadIdx.x] = ord2f(root_box[3 + threadIdx.x]); }
  if (threadIdx.x < 6) ncb[threadIdx.x] = cbounds[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_sah_level_flags(const int2* nseg, int lb, int le, uint32_t* act, uint32_t* big) {
  const int i = lb + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= le) return;
  const int c = nseg[i].y;
  act[i - lb] = c >= 2 ? 1u : 0u; big[i - lb] = c > SAH_SWEEP ? 1u : 0u;
}
__global__ void k_sah_level_totals(const uint32_t* act, const uint32_t* act_pre, const uint32_t* big, const uint32_t* big_pre, int count, uint32_t* tot) {
  if (threadIdx.x == 0) { tot[0] = act_pre[count - 1] + act[count - 1]; tot[1] = big_pre[count - 1] + big[count - 1]; }
}
__global__ __launch_bounds__(256) void k_sah_bins_init(uint32_t* bins, uint32_t words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  const uint32_t w = i % 7u;
  bins[i] = w == 0u ? 0u : (w <= 3u ? 0xFFFFFFFFu : 0u);
}

__device__ __forceinline__ int sah_bin_of(float c, float lo, float scale) {
  int b = (int)((c - lo) * scale);
  return b < 0 ? 0 : (b >= SAH_BINS ? SAH_BINS - 1 : b);
}
struct SahFrame { float lo[3], scale[3]; bool live[3]; };
__device__ __forceinline__ SahFrame sah_frame(const uint32_t* cb) {
  SahFrame f;
  for (int a = 0; a < 3; a++) {
    const float lo = ord2f(cb[a]), hi = ord2f(cb[3 + a]);
    const float ext = hi - lo;
    f.lo[a] = lo; f.live[a] = ext > 0.f; f.scale[a] = f.live[a] ? (float)SAH_BINS / ext : 0.f;
  }
  return f;
}

// A workgroup whose 256 references all belong to ONE binned node (the rule on the upper levels, where the atomics on a node's few
// hundred bin words would otherwise serialise: 4.5 ms for the root level alone) accumulates in LDS and adds its non-empty bins once.
__global__ __launch_bounds__(256) void k_sah_bin(const Box* tri_boxes, const uint32_t* ids, const int* pos_node, uint32_t n, const int2* nseg, const uint32_t* ncb,
                                                 int lb, const uint32_t* big_pre, uint32_t* bins) {
  __shared__ uint32_t s_bins[SAH_BIN_WORDS];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const int node0 = pos_node[blockIdx.x * blockDim.x];
  const int node = p < n ? pos_node[p] : node0;
  const bool uniform = __syncthreads_and(node == node0 ? 1 : 0) != 0;
  if (uniform) {
    if (nseg[node0].y <= SAH_SWEEP) return;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SAH_BIN_WORDS; i += 256u) { const uint32_t w = i % 7u; s_bins[i] = w == 0u ? 0u : (w <= 3u ? 0xFFFFFFFFu : 0u); }
    __syncthreads();
  }
  if (p < n && nseg[node].y > SAH_SWEEP) {
    const SahFrame f = sah_frame(ncb + 6 * (size_t)node);
    const Box b = tri_boxes[ids[p]];
    uint32_t* nb = uniform ? s_bins : bins + (size_t)big_pre[node - lb] * SAH_BIN_WORDS;
    for (int a = 0; a < 3; a++) {
      if (!f.live[a]) continue;
      const int bin = sah_bin_of(0.5f * b.lo[a] + 0.5f * b.hi[a], f.lo[a], f.scale[a]);
      uint32_t* w = nb + (a * SAH_BINS + bin) * 7;
      atomicAdd(w, 1u);
      for (int k = 0; k < 3; k++) { atomicMin(w + 1 + k, f2ord(b.lo[k])); atomicMax(w + 4 + k, f2ord(b.hi[k])); }
    }
  }
  if (uniform) {
    __syncthreads();
    uint32_t* nb = bins + (size_t)big_pre[node0 - lb] * SAH_BIN_WORDS;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SAH_BIN_WORDS; i += 256u) {
      const uint32_t w = i % 7u, v = s_bins[i];
      if (w == 0u) { if (v) atomicAdd(nb + i, v); }
      else if (w <= 3u) { if (v != 0xFFFFFFFFu) atomicMin(nb + i, v); }
      else if (v != 0u) atomicMax(nb + i, v);
    }
  }
}

__device__ __forceinline__ void box_empty(Box& b) { for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; } }
__device__ __forceinline__ void box_add(Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], o.lo[k]); b.hi[k] = fmaxf(b.hi[k], o.hi[k]); } }
__device__ __forceinline__ float box_area0(const Box& b) {   // an empty box has no area
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return (dx < 0.f || dy < 0.f || dz < 0.f) ? 0.f : dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ Box bin_box(const uint32_t* w) {
  Box b; for (int k = 0; k < 3; k++) { b.lo[k] = ord2f(w[1 + k]); b.hi[k] = ord2f(w[4 + k]); } return b;
}

// nsplit[node - lb] = (axis 0..2 | 3 = position median, last bin that goes left, references that go left, 0)
__global__ __launch_bounds__(64) void k_sah_split(const Box* tri_boxes, const uint32_t* ids, uint32_t* ids_next, int* pos_node_next, int2* nseg, int2* nkids, Box* nbox,
                                                  uint32_t* ncb, int lb, int le, const uint32_t* act_pre, const uint32_t* big_pre, const uint32_t* bins, int4* nsplit,
                                                  int force_median) {
  const int node = lb + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (node >= le) return;
  const int2 seg = nseg[node];
  if (seg.y < 2) return;
  const int left = le + 2 * (int)act_pre[node - lb], right = left + 1;
  nkids[node] = make_int2(left, right);
  nkids[left] = make_int2(-1, -1); nkids[right] = make_int2(-1, -1);
  const SahFrame f = sah_frame(ncb + 6 * (size_t)node);
  Box lbx, rbx; box_empty(lbx); box_empty(rbx);
  int nl = seg.y / 2;
  if (seg.y > SAH_SWEEP) {
    // ---- binned node: the candidates are the 31 bin boundaries of each live axis
    const uint32_t* nb = bins + (size_t)big_pre[node - lb] * SAH_BIN_WORDS;
    float best = 3.0e38f; int best_axis = -1, best_bin = -1, best_nl = 0;
    if (!force_median)
      for (int a = 0; a < 3; a++) {
        if (!f.live[a]) continue;
        float la[SAH_BINS]; uint32_t lc[SAH_BINS];
        Box acc; box_empty(acc); uint32_t c = 0;
        for (int b = 0; b < SAH_BINS; b++) {
          const uint32_t* w = nb + (a * SAH_BINS + b) * 7;
          if (w[0]) { c += w[0]; box_add(acc, bin_box(w)); }
          la[b] = box_area0(acc); lc[b] = c;
        }
        box_empty(acc); c = 0;
        for (int b = SAH_BINS - 1; b >= 1; b--) {
          const uint32_t* w = nb + (a * SAH_BINS + b) * 7;
          if (w[0]) { c += w[0]; box_add(acc, bin_box(w)); }
          const uint32_t cl = lc[b - 1];
          if (cl == 0u || c == 0u) continue;
          const float cost = la[b - 1] * (float)cl + box_area0(acc) * (float)c;
          if (cost < best) { best = cost; best_axis = a; best_bin = b - 1; best_nl = (int)cl; }
        }
      }
    if (best_axis >= 0) {
      for (int b = 0; b < SAH_BINS; b++) {
        const uint32_t* w = nb + (best_axis * SAH_BINS + b) * 7;
        if (w[0]) box_add(b <= best_bin ? lbx : rbx, bin_box(w));
      }
      nl = best_nl;
      nsplit[node - lb] = make_int4(best_axis, best_bin, nl, 0);
    } else {
      // no axis separates the centroids (or the depth guard is on): the first half of the segment goes left; the children keep the
      // parent's box (conservative), their centroid bounds are accumulated by the scatter as always
      lbx = nbox[node]; rbx = lbx;
      nsplit[node - lb] = make_int4(3, 0, nl, 0);
    }
    for (int k = 0; k < 6; k++) { ncb[6 * (size_t)left + k] = k < 3 ? 0xFFFFFFFFu : 0u; ncb[6 * (size_t)right + k] = k < 3 ? 0xFFFFFFFFu : 0u; }
  } else {
    // ---- small node: every split position of the references sorted by centroid, per axis (exact sweep); this thread also moves them
    Box bx[SAH_SWEEP]; float cen[SAH_SWEEP][3]; uint32_t id[SAH_SWEEP];
    const int cnt = seg.y;
    for (int i = 0; i < cnt; i++) {
      id[i] = ids[seg.x + i]; bx[i] = tri_boxes[id[i]];
      for (int k = 0; k < 3; k++) cen[i][k] = 0.5f * bx[i].lo[k] + 0.5f * bx[i].hi[k];
    }
    float best = 3.0e38f; int best_k = -1;
    uint8_t best_ord[SAH_SWEEP];
    for (int i = 0; i < cnt; i++) best_ord[i] = (uint8_t)i;
    if (!force_median)
      for (int a = 0; a < 3; a++) {
        if (!f.live[a]) continue;
        uint8_t ord[SAH_SWEEP];
        for (int i = 0; i < cnt; i++) {   // insertion sort by (centroid, id)
          int j = i;
          while (j > 0) {
            const int q = ord[j - 1];
            if (cen[q][a] < cen[i][a] || (cen[q][a] == cen[i][a] && id[q] < id[i])) break;
            ord[j] = ord[j - 1]; j--;
          }
          ord[j] = (uint8_t)i;
        }
        float ra[SAH_SWEEP];
        Box acc; box_empty(acc);
        for (int i = cnt - 1; i >= 1; i--) { box_add(acc, bx[ord[i]]); ra[i] = box_area0(acc); }
        box_empty(acc);
        for (int k = 1; k < cnt; k++) {   // the first k references go left
          box_add(acc, bx[ord[k - 1]]);
          const float cost = box_area0(acc) * (float)k + ra[k] * (float)(cnt - k);
          if (cost < best) { best = cost; best_k = k; for (int i = 0; i < cnt; i++) best_ord[i] = ord[i]; }
        }
      }
    if (best_k > 0) nl = best_k;
    uint32_t cl[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u}, cr[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    for (int i = 0; i < cnt; i++) {
      const int q = best_ord[i];
      ids_next[seg.x + i] = id[q]; pos_node_next[seg.x + i] = i < nl ? left : right;
      box_add(i < nl ? lbx : rbx, bx[q]);
      uint32_t* c = i < nl ? cl : cr;
      for (int k = 0; k < 3; k++) { const uint32_t o = f2ord(cen[q][k]); c[k] = min(c[k], o); c[3 + k] = max(c[3 + k], o); }
    }
    for (int k = 0; k < 6; k++) { ncb[6 * (size_t)left + k] = cl[k]; ncb[6 * (size_t)right + k] = cr[k]; }
  }
  nseg[left] = make_int2(seg.x, nl); nseg[right] = make_int2(seg.x + nl, seg.y - nl);
  nbox[left] = lbx; nbox[right] = rbx;
}

__global__ __launch_bounds__(256) void k_sah_side(const Box* tri_boxes, const uint32_t* ids, const int* pos_node, uint32_t n, const int2* nseg, const uint32_t* ncb,
                                                  int lb, const int4* nsplit, uint32_t* side) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int node = pos_node[p];
  const int2 seg = nseg[node];
  uint32_t goes_left = 0u;
  if (seg.y > SAH_SWEEP) {
    const int4 sp = nsplit[node - lb];
    if (sp.x == 3) goes_left = ((int)p - seg.x) < sp.z ? 1u : 0u;
    else {
      const SahFrame f = sah_frame(ncb + 6 * (size_t)node);
      const Box b = tri_boxes[ids[p]];
      goes_left = sah_bin_of(0.5f * b.lo[sp.x] + 0.5f * b.hi[sp.x], f.lo[sp.x], f.scale[sp.x]) <= sp.y ? 1u : 0u;
    }
  }
  side[p] = goes_left;
}

__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o)); return v; }

// (a wavefront whose 64 references all move inside ONE binned node reduces their centroid bounds per side first: twelve atomics
// per wave instead of six per reference — on the root level 174 000 atomics per word took 20 ms)
__global__ __launch_bounds__(256) void k_sah_scatter(const Box* tri_boxes, const uint32_t* ids, const int* pos_node, uint32_t n, const int2* nseg, const int2* nkids,
                                                     int lb, const int4* nsplit, const uint32_t* side, const uint32_t* side_pre, uint32_t* ids_next, int* pos_node_next,
                                                     uint32_t* ncb) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = p < n;
  const int node = in ? pos_node[p] : -1;
  const int2 seg = in ? nseg[node] : make_int2(0, 0);
  const bool moves = in && seg.y > SAH_SWEEP;   // (a small node's references are moved by k_sah_split)
  if (in && seg.y == 1) { ids_next[p] = ids[p]; pos_node_next[p] = node; }   // a finished leaf stays where it is
  bool l = false; int2 kids = make_int2(-1, -1);
  uint32_t o[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (moves) {
    const int nl = nsplit[node - lb].z;
    const uint32_t lefts_before = side_pre[p] - side_pre[seg.x];
    l = side[p] != 0u;
    const uint32_t np = (uint32_t)seg.x + (l ? lefts_before : (uint32_t)nl + (p - (uint32_t)seg.x - lefts_before));
    kids = nkids[node];
    const uint32_t id = ids[p];
    ids_next[np] = id; pos_node_next[np] = l ? kids.x : kids.y;
    const Box b = tri_boxes[id];
    for (int k = 0; k < 3; k++) { o[k] = f2ord(0.5f * b.lo[k] + 0.5f * b.hi[k]); o[3 + k] = o[k]; }
  }
  const int node_first = __builtin_amdgcn_readfirstlane(node);
  const bool wave_uniform = __ballot(!moves || node != node_first) == 0ull;   // (the whole wave is active here)
  if (wave_uniform) {
    uint32_t lo_l[3], hi_l[3], lo_r[3], hi_r[3];
    for (int k = 0; k < 3; k++) {
      lo_l[k] = wave_min_u(l ? o[k] : 0xFFFFFFFFu); hi_l[k] = wave_max_u(l ? o[3 + k] : 0u);
      lo_r[k] = wave_min_u(l ? 0xFFFFFFFFu : o[k]); hi_r[k] = wave_max_u(l ? 0u : o[3 + k]);
    }
    if ((threadIdx.x & 63u) == 0u) {
      uint32_t* cl = ncb + 6 * (size_t)kids.x; uint32_t* cr = ncb + 6 * (size_t)kids.y;
      for (int k = 0; k < 3; k++) {
        if (lo_l[k] != 0xFFFFFFFFu) { atomicMin(cl + k, lo_l[k]); atomicMax(cl + 3 + k, hi_l[k]); }
        if (lo_r[k] != 0xFFFFFFFFu) { atomicMin(cr + k, lo_r[k]); atomicMax(cr + 3 + k, hi_r[k]); }
      }
    }
  } else if (moves) {
    uint32_t* c = ncb + 6 * (size_t)(l ? kids.x : kids.y);
    for (int k = 0; k < 3; k++) { atomicMin(c + k, o[k]); atomicMax(c + 3 + k, o[3 + k]); }
  }
}

__global__ __launch_bounds__(256) void k_sah_internal_flags(const int2* nseg, int n_nodes, uint32_t* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes) flag[i] = nseg[i].y >= 2 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_sah_emit(const int2* nseg, const int2* nkids, const Box* nbox, const uint32_t* iidx, int n_nodes, const float* qparams, BvhNodeQ* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || nseg[i].y < 2) return;
  const int2 ch = nkids[i];
  BvhNodeQ q{};
  const Box b0 = nbox[ch.x], b1 = nbox[ch.y];
  for (int a = 0; a < 3; a++) {
    q.w[a] = quant_box_axis(b0.lo[a], b0.hi[a], qparams[a], qparams[3 + a]);
    q.w[3 + a] = quant_box_axis(b1.lo[a], b1.hi[a], qparams[a], qparams[3 + a]);
  }
  q.child0 = nseg[ch.x].y >= 2 ? (int)iidx[ch.x] : ~(int)(((uint32_t)nseg[ch.x].x << 3) | 0u);
  q.child1 = nseg[ch.y].y >= 2 ? (int)iidx[ch.y] : ~(int)(((uint32_t)nseg[ch.y].x << 3) | 0u);
  out[iidx[i]] = q;
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
  const auto t_begin = std::chrono::steady_clock::now();
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
  int algo = 3;   // 3: binned SAH, level by level (default: the host builder's tree quality, ~3-5 % faster frames than the LBVH's); 1: LBVH (Karras radix tree) + rotations; 2: PLOC
  if (const char* e = getenv("RT_GPU_BVH_ALGO")) { int v = atoi(e); if (v >= 1 && v <= 3) algo = v; }
  if (max_leaf != 1 && algo == 3) algo = 1;   // (the SAH builder makes one-triangle leaves only)
  if (algo == 3) {
    const size_t cap = 2 * (size_t)n;
    uint32_t *ids[2] = {nullptr, nullptr}, *root_box = nullptr, *ncb = nullptr, *act = nullptr, *big = nullptr, *act_pre = nullptr, *big_pre = nullptr, *tot = nullptr,
             *bins = nullptr, *side = nullptr, *side_pre = nullptr, *iidx = nullptr;
    int* pos_node[2] = {nullptr, nullptr};
    Box* nbox = nullptr; int2 *nseg = nullptr, *nkids = nullptr; int4* nsplit = nullptr; void* scan_tmp = nullptr;
    auto cleanup3 = [&]() {
      for (void* p : {(void*)ids[0], (void*)ids[1], (void*)root_box, (void*)ncb, (void*)act, (void*)big, (void*)act_pre, (void*)big_pre, (void*)tot, (void*)bins, (void*)side,
                      (void*)side_pre, (void*)iidx, (void*)pos_node[0], (void*)pos_node[1], (void*)nbox, (void*)nseg, (void*)nkids, (void*)nsplit, scan_tmp})
        if (p) hipFree(p);
    };
#define S3_TRY(expr) do { hipError_t e3_ = (expr); if (e3_ != hipSuccess) { err = std::string("HIP runtime exception: return code ") + std::to_string((int)e3_) + " (" + hipGetErrorString(e3_) + ") in " #expr; cleanup3(); cleanup(); return 1; } } while (0)
    const size_t max_big = (size_t)n / (SAH_SWEEP + 1) + 2;
    for (int k = 0; k < 2; k++) { S3_TRY(hipMalloc((void**)&ids[k], n * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&pos_node[k], n * sizeof(int))); }
    S3_TRY(hipMalloc((void**)&root_box, 6 * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&ncb, cap * 6 * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&act, cap * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&big, cap * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&act_pre, cap * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&big_pre, cap * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&tot, 2 * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&bins, max_big * SAH_BIN_WORDS * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&side, n * sizeof(uint32_t))); S3_TRY(hipMalloc((void**)&side_pre, n * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&iidx, cap * sizeof(uint32_t)));
    S3_TRY(hipMalloc((void**)&nbox, cap * sizeof(Box))); S3_TRY(hipMalloc((void**)&nseg, cap * sizeof(int2))); S3_TRY(hipMalloc((void**)&nkids, cap * sizeof(int2)));
    S3_TRY(hipMalloc((void**)&nsplit, cap * sizeof(int4)));
    size_t scan_bytes = 0;
    S3_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, act, act_pre, (int)cap, s));
    S3_TRY(hipMalloc(&scan_tmp, scan_bytes));
    S3_TRY(hipMemsetAsync(ids[1], 0, n * sizeof(uint32_t), s)); S3_TRY(hipMemsetAsync(pos_node[1], 0, n * sizeof(int), s));
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, s, root_box);
    hipLaunchKernelGGL(k_sah_root, dim3(nb), dim3(256), 0, s, tri_boxes, n, root_box, ids[0], pos_node[0]);
    hipLaunchKernelGGL(k_sah_root_node, dim3(1), dim3(64), 0, s, root_box, cbounds, n, nbox, ncb, nseg, nkids);
    int lb = 0, le = 1, cur = 0, level = 0;
    for (;; level++) {
      const int cnt = le - lb;
      const uint32_t lvb = ((uint32_t)cnt + 255u) / 256u;
      hipLaunchKernelGGL(k_sah_level_flags, dim3(lvb), dim3(256), 0, s, nseg, lb, le, act, big);
      S3_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, act, act_pre, cnt, s));
      S3_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, big, big_pre, cnt, s));
      hipLaunchKernelGGL(k_sah_level_totals, dim3(1), dim3(64), 0, s, act, act_pre, big, big_pre, cnt, tot);
      uint32_t h_tot[2];
      S3_TRY(hipMemcpyAsync(h_tot, tot, sizeof(h_tot), hipMemcpyDeviceToHost, s));
      S3_TRY(hipStreamSynchronize(s));
      const int n_act = (int)h_tot[0], n_big = (int)h_tot[1];
      if (n_act == 0) break;
      if ((size_t)n_big > max_big || (size_t)le + 2 * (size_t)n_act > cap || level > 4096) { err = "SAH builder: inconsistent level (" + std::to_string(n_act) + " splits, " + std::to_string(n_big) + " binned)"; cleanup3(); cleanup(); return 1; }
      if (n_big) {
        const uint32_t words = (uint32_t)n_big * SAH_BIN_WORDS;
        hipLaunchKernelGGL(k_sah_bins_init, dim3((words + 255u) / 256u), dim3(256), 0, s, bins, words);
        hipLaunchKernelGGL(k_sah_bin, dim3(nb), dim3(256), 0, s, tri_boxes, ids[cur], pos_node[cur], n, nseg, ncb, lb, big_pre, bins);
      }
      hipLaunchKernelGGL(k_sah_split, dim3(((uint32_t)cnt + 63u) / 64u), dim3(64), 0, s, tri_boxes, ids[cur], ids[cur ^ 1], pos_node[cur ^ 1], nseg, nkids, nbox, ncb, lb, le,
                         act_pre, big_pre, bins, nsplit, level >= 48 ? 1 : 0);
      hipLaunchKernelGGL(k_sah_side, dim3(nb), dim3(256), 0, s, tri_boxes, ids[cur], pos_node[cur], n, nseg, ncb, lb, nsplit, side);
      S3_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, side, side_pre, (int)n, s));
      hipLaunchKernelGGL(k_sah_scatter, dim3(nb), dim3(256), 0, s, tri_boxes, ids[cur], pos_node[cur], n, nseg, nkids, lb, nsplit, side, side_pre, ids[cur ^ 1], pos_node[cur ^ 1], ncb);
      cur ^= 1; lb = le; le += 2 * n_act;
    }
    const int n_nodes = le;
    if (n_nodes != 2 * (int)n - 1) { err = "SAH builder produced " + std::to_string(n_nodes) + " nodes for " + std::to_string(n) + " triangles"; cleanup3(); cleanup(); return 1; }
    const uint32_t nnb = ((uint32_t)n_nodes + 255u) / 256u;
    hipLaunchKernelGGL(k_sah_internal_flags, dim3(nnb), dim3(256), 0, s, nseg, n_nodes, act);
    S3_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, act, iidx, n_nodes, s));
    hipLaunchKernelGGL(k_quant_params, dim3(1), dim3(64), 0, s, nbox, qparams);
    hipLaunchKernelGGL(k_sah_emit, dim3(nnb), dim3(256), 0, s, nseg, nkids, nbox, iidx, n_nodes, qparams, out.nodes);
    hipLaunchKernelGGL(k_emit_tris, dim3(nb), dim3(256), 0, s, d_verts6, d_idx, ids[cur], n, out.tris);
    float h3[12];
    S3_TRY(hipMemcpyAsync(h3, qparams, sizeof(h3), hipMemcpyDeviceToHost, s));
    S3_TRY(hipStreamSynchronize(s));
    S3_TRY(hipGetLastError());
    for (int k = 0; k < 3; k++) { out.q_lo[k] = h3[k]; out.q_scale[k] = h3[3 + k]; out.bounds_lo[k] = h3[6 + k]; out.bounds_hi[k] = h3[9 + k]; }
    out.n_nodes = n - 1; out.n_tris = n;
    if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "[bvh_gpu] SAH: %u triangles, %d levels, %.2f ms\n", n, level, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    done = true;
    cleanup3(); cleanup();
    return 0;
#undef S3_TRY
  }
  hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, s, tri_boxes, n, cbounds, keys, vals);
  size_t tmp_bytes = 0;
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 30, s));
  GB_TRY(hipMalloc(&tmp, tmp_bytes));
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 30, s));
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
  if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "[bvh_gpu] LBVH: %u triangles, %.2f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
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
