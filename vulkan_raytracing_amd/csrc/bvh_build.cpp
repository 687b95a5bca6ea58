// bvh_build.cpp — binned-SAH BVH2 builder producing the 64-byte two-child-box node layout of
// rt_device.h.  Replaces what the Vulkan driver does behind vkCmdBuildAccelerationStructuresKHR
// (reference src/main.cpp:495-498 for BLAS, :730-733 for TLAS build/update).  Deterministic: the
// same input always yields the same tree, so every GPU of a multi-GPU job holds identical BVHs.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rt {
namespace {

constexpr float BIG = 3.0e38f;
constexpr int NBINS = 32;

struct Ref {
  Aabb box;
  float c[3];
  uint32_t id;
};

inline void box_reset(Aabb& b) {
  for (int k = 0; k < 3; k++) { b.lo[k] = BIG; b.hi[k] = -BIG; }
}
inline void box_grow(Aabb& b, const Aabb& o) {
  for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); }
}
inline float half_area(const Aabb& b) {
  float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
  return dx * dy + dy * dz + dz * dx;
}
// minimum number of levels a subtree of n primitives needs below its root
inline int levels_needed(uint32_t n, int max_leaf) {
  uint32_t leaves = (n + (uint32_t)max_leaf - 1) / (uint32_t)max_leaf;
  int l = 0;
  while ((1ull << l) < leaves) l++;
  return l;
}

// One node of the top-down build: bounds, the best binned-SAH split over the three axes (all three binned in one pass over the
// references), the partition.  Nodes of at least PAR_MIN references do each of those passes on `threads` threads (chunked; the
// partition is a stable two-pass scatter through `tmp`), smaller subtrees are whole tasks for the same threads.  The tree depends
// only on the SETS of references on either side of every split — bin boxes and counts are order-independent — so it is the same
// for every thread count; node numbers in `topo` are not (they are allocated from an atomic counter), and nothing reads them:
// the emitter walks the links depth-first.
struct Builder {
  std::vector<Ref> refs, tmp;
  int max_leaf, max_depth;
  float trav_cost = 1.0f;   // SAH: cost of visiting an interior node relative to one triangle test
  BuiltBvh* out;
  int threads = 1;
  // Node numbers, leaf count and depth are kept per task (one cache line hammered by every thread at every node made eight
  // threads slower than one): a subtree over c references owns the next 2c - 2 node numbers below its root, reserved in one step.
  struct Local { int next = 0; uint32_t leaves = 0; int depth = 0; };
  std::atomic<int> next_node{0};
  static constexpr uint32_t PAR_MIN = 16384;
  static constexpr uint32_t SWEEP_MAX = 12;   // nodes of at most this many references take the exact sweep

  struct Bins {
    uint32_t cnt[3][NBINS];
    Aabb bb[3][NBINS];
    void reset() { for (int a = 0; a < 3; a++) for (int b = 0; b < NBINS; b++) { cnt[a][b] = 0; box_reset(bb[a][b]); } }
  };
  struct Task { int node; uint32_t first, count; int depth; };

  template <class F>
  void chunks(uint32_t first, uint32_t count, int nt, F&& f) {   // f(chunk index, chunk first, chunk count)
    if (nt <= 1) { f(0, first, count); return; }
    std::vector<std::thread> th;
    const uint32_t per = (count + (uint32_t)nt - 1) / (uint32_t)nt;
    for (int t = 1; t < nt; t++) {
      const uint32_t b = std::min(count, per * (uint32_t)t), e = std::min(count, per * (uint32_t)(t + 1));
      th.emplace_back([&f, t, first, b, e]() { f(t, first + b, e - b); });
    }
    f(0, first, std::min(count, per));
    for (auto& x : th) x.join();
  }

  // splits node `me`; returns false for a leaf, else the two children (node, first, count)
  bool split(int me, uint32_t first, uint32_t count, int depth, int nt, Local& lc, Task& L, Task& R) {
    lc.depth = std::max(lc.depth, depth);
    if (count < PAR_MIN) nt = 1;
    Aabb bounds, cb;
    if (nt == 1) {
      box_reset(bounds); box_reset(cb);
      for (uint32_t i = first; i < first + count; i++) {
        box_grow(bounds, refs[i].box);
        for (int k = 0; k < 3; k++) { cb.lo[k] = std::min(cb.lo[k], refs[i].c[k]); cb.hi[k] = std::max(cb.hi[k], refs[i].c[k]); }
      }
    } else {
      std::vector<Aabb> pb((size_t)nt), pc((size_t)nt);
      chunks(first, count, nt, [&](int t, uint32_t f, uint32_t n) {
        Aabb b, c; box_reset(b); box_reset(c);
        for (uint32_t i = f; i < f + n; i++) {
          box_grow(b, refs[i].box);
          for (int k = 0; k < 3; k++) { c.lo[k] = std::min(c.lo[k], refs[i].c[k]); c.hi[k] = std::max(c.hi[k], refs[i].c[k]); }
        }
        pb[t] = b; pc[t] = c;
      });
      bounds = pb[0]; cb = pc[0];
      for (int t = 1; t < nt; t++) { box_grow(bounds, pb[t]); box_grow(cb, pc[t]); }
    }
    BuildNode& node = out->topo[me];
    node.box = bounds; node.first = first;
    if (count <= 1) { node.count = count; lc.leaves++; return false; }

    // ---- best SAH split over 3 axes: NBINS bins each, or — for the few references of the bottom levels, where a bin grid is
    // mostly empty bins to reset and merge — every split position of the references sorted by centroid (the exact sweep)
    float scale[3]; bool live[3];
    for (int a = 0; a < 3; a++) { const float ext = cb.hi[a] - cb.lo[a]; live[a] = ext > 0.f; scale[a] = live[a] ? (float)NBINS / ext : 0.f; }
    float best_cost = BIG; int best_axis = -1, best_bin = -1;
    const bool sweep = count <= SWEEP_MAX;
    uint8_t sweep_order[SWEEP_MAX];   // the references in the order of the best axis
    if (sweep) {
      for (int axis = 0; axis < 3; axis++) {
        if (!live[axis]) continue;
        uint8_t ord[SWEEP_MAX];
        for (uint32_t i = 0; i < count; i++) {   // insertion sort by (centroid, id)
          const Ref& r = refs[first + i];
          uint32_t j = i;
          while (j > 0) {
            const Ref& q = refs[first + ord[j - 1]];
            if (q.c[axis] < r.c[axis] || (q.c[axis] == r.c[axis] && q.id < r.id)) break;
            ord[j] = ord[j - 1]; j--;
          }
          ord[j] = (uint8_t)i;
        }
        float ra[SWEEP_MAX];
        Aabb acc; box_reset(acc);
        for (uint32_t i = count; i-- > 1;) { box_grow(acc, refs[first + ord[i]].box); ra[i] = half_area(acc); }
        box_reset(acc);
        for (uint32_t k = 1; k < count; k++) {   // the first k references go left
          box_grow(acc, refs[first + ord[k - 1]].box);
          const float cost = half_area(acc) * (float)k + ra[k] * (float)(count - k);
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = (int)k; for (uint32_t i = 0; i < count; i++) sweep_order[i] = ord[i]; }
        }
      }
    } else {
      Bins stack_bins;
      std::vector<Bins> heap_bins;
      Bins* pbins = &stack_bins;
      if (nt > 1) { heap_bins.resize((size_t)nt); pbins = heap_bins.data(); }
      chunks(first, count, nt, [&](int t, uint32_t f, uint32_t n) {
        Bins& B = pbins[t]; B.reset();
        for (uint32_t i = f; i < f + n; i++)
          for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            int b = (int)((refs[i].c[a] - cb.lo[a]) * scale[a]);
            b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
            B.cnt[a][b]++; box_grow(B.bb[a][b], refs[i].box);
          }
      });
      Bins& B = pbins[0];
      for (int t = 1; t < nt; t++)
        for (int a = 0; a < 3; a++) for (int b = 0; b < NBINS; b++) { B.cnt[a][b] += pbins[t].cnt[a][b]; box_grow(B.bb[a][b], pbins[t].bb[a][b]); }
      for (int axis = 0; axis < 3; axis++) {
        if (!live[axis]) continue;
        float la[NBINS]; uint32_t lc[NBINS];
        Aabb acc; box_reset(acc); uint32_t c = 0;
        for (int b = 0; b < NBINS; b++) { c += B.cnt[axis][b]; box_grow(acc, B.bb[axis][b]); la[b] = half_area(acc); lc[b] = c; }
        box_reset(acc); c = 0;
        for (int b = NBINS - 1; b >= 1; b--) {
          c += B.cnt[axis][b]; box_grow(acc, B.bb[axis][b]);
          uint32_t nl = lc[b - 1], nr = c;
          if (nl == 0 || nr == 0) continue;
          float cost = la[b - 1] * (float)nl + half_area(acc) * (float)nr;
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b - 1; }
        }
      }
    }
    float pa = half_area(bounds);
    if (count <= (uint32_t)max_leaf) {
      float leaf_cost = pa * (float)count;
      float split_cost = best_axis >= 0 ? pa * trav_cost + best_cost : BIG;
      if (leaf_cost <= split_cost) { node.count = count; lc.leaves++; return false; }
    }
    uint32_t mid = first;
    bool ok = false;
    int budget = max_depth - depth - 1;  // levels available below each child
    if (best_axis >= 0 && sweep) {
      Ref sorted[SWEEP_MAX];
      for (uint32_t i = 0; i < count; i++) sorted[i] = refs[first + sweep_order[i]];
      for (uint32_t i = 0; i < count; i++) refs[first + i] = sorted[i];
      mid = first + (uint32_t)best_bin;
      uint32_t nl = mid - first, nr = count - nl;
      ok = levels_needed(nl, max_leaf) <= budget && levels_needed(nr, max_leaf) <= budget;
    } else if (best_axis >= 0) {
      const float sc = scale[best_axis], lo = cb.lo[best_axis]; const int ax = best_axis, bbin = best_bin;
      auto left = [&](const Ref& r) {
        int b = (int)((r.c[ax] - lo) * sc);
        b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
        return b <= bbin;
      };
      if (nt > 1) {   // stable scatter through tmp: lefts of all chunks, then rights of all chunks
        std::vector<uint32_t> nl((size_t)nt + 1, 0), nr((size_t)nt + 1, 0);
        chunks(first, count, nt, [&](int t, uint32_t f, uint32_t n) {
          uint32_t l = 0;
          for (uint32_t i = f; i < f + n; i++) l += left(refs[i]) ? 1u : 0u;
          nl[(size_t)t + 1] = l; nr[(size_t)t + 1] = n - l;
        });
        for (int t = 0; t < nt; t++) { nl[(size_t)t + 1] += nl[t]; nr[(size_t)t + 1] += nr[t]; }
        const uint32_t total_left = nl[nt];
        chunks(first, count, nt, [&](int t, uint32_t f, uint32_t n) {
          uint32_t l = first + nl[t], r = first + total_left + nr[t];
          for (uint32_t i = f; i < f + n; i++) { if (left(refs[i])) tmp[l++] = refs[i]; else tmp[r++] = refs[i]; }
        });
        chunks(first, count, nt, [&](int, uint32_t f, uint32_t n) { std::copy(tmp.begin() + f, tmp.begin() + f + n, refs.begin() + f); });
        mid = first + total_left;
      } else {
        auto it = std::partition(refs.begin() + first, refs.begin() + first + count, left);
        mid = (uint32_t)(it - refs.begin());
      }
      uint32_t nl = mid - first, nr = count - nl;
      ok = nl > 0 && nr > 0 && levels_needed(nl, max_leaf) <= budget && levels_needed(nr, max_leaf) <= budget;
    }
    if (!ok) {  // balanced median split on the widest centroid axis (keeps the depth bound)
      int ax = 0;
      for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
      mid = first + count / 2;
      std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + first + count,
                       [ax](const Ref& a, const Ref& b) { return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.id < b.id); });
    }
    L = Task{lc.next++, first, mid - first, depth + 1};
    R = Task{lc.next++, mid, first + count - mid, depth + 1};
    node.left = L.node; node.right = R.node;
    return true;
  }

  void subtree(const Task& t, Local& lc) {   // one thread, depth-first
    Task L, R;
    if (!split(t.node, t.first, t.count, t.depth, 1, lc, L, R)) return;
    subtree(L, lc); subtree(R, lc);
  }

  void run(uint32_t n) {
    out->topo.assign(2 * (size_t)n, BuildNode{});   // a binary tree over n references has at most 2n - 1 nodes
    if (threads > 1) tmp.resize(n);
    std::vector<Task> big, small;
    Local top; top.next = 1;   // node 0 = the root
    big.push_back(Task{0, 0, n, 0});
    while (!big.empty()) {   // the large nodes one after the other, every pass over their references on all threads
      const Task t = big.back(); big.pop_back();
      if (threads <= 1 || t.count < PAR_MIN) { small.push_back(t); continue; }
      Task L, R;
      if (split(t.node, t.first, t.count, t.depth, threads, top, L, R)) { big.push_back(R); big.push_back(L); }
    }
    next_node = top.next;
    // the rest: whole subtrees, largest first, handed out from an atomic cursor
    std::sort(small.begin(), small.end(), [](const Task& a, const Task& b) { return a.count > b.count || (a.count == b.count && a.first < b.first); });
    std::atomic<size_t> cursor{0};
    const int nw = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, small.size()));
    std::vector<Local> res((size_t)nw);
    auto worker = [&](int w) {
      Local acc;
      for (size_t i; (i = cursor.fetch_add(1)) < small.size();) {
        Local lc; lc.next = next_node.fetch_add(2 * (int)small[i].count - 2);   // (its root has a number already)
        subtree(small[i], lc);
        acc.leaves += lc.leaves; acc.depth = std::max(acc.depth, lc.depth);
      }
      res[(size_t)w] = acc;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nw; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
    out->topo.resize((size_t)std::min<long long>(next_node.load(), 2 * (long long)n));
    out->depth = top.depth; out->leaves = top.leaves;
    for (const Local& r : res) { out->depth = std::max(out->depth, r.depth); out->leaves += r.leaves; }
  }
};

inline void set_child_box(BvhNode& n, int which, const Aabb& b) {
  if (which == 0) {
    n.a[0] = b.lo[0]; n.a[1] = b.hi[0]; n.a[2] = b.lo[1]; n.a[3] = b.hi[1]; n.c[0] = b.lo[2]; n.c[1] = b.hi[2];
  } else {
    n.b[0] = b.lo[0]; n.b[1] = b.hi[0]; n.b[2] = b.lo[1]; n.b[3] = b.hi[1]; n.c[2] = b.lo[2]; n.c[3] = b.hi[2];
  }
}
inline Aabb missing_box() {
  Aabb b; for (int k = 0; k < 3; k++) { b.lo[k] = BIG; b.hi[k] = BIG; } return b;
}

struct Emitter {
  BuiltBvh* bvh;
  bool direct_ids;
  int32_t leaf_ref(const BuildNode& t) const {
    if (direct_ids) return ~(int32_t)bvh->order[t.first];
    return ~(int32_t)((t.first << 3) | (t.count - 1));
  }
  int32_t emit(int ti) {
    const BuildNode t = bvh->topo[ti];
    if (t.left < 0) return leaf_ref(t);
    int32_t e = (int32_t)bvh->nodes.size();
    bvh->nodes.push_back(BvhNode{});
    bvh->emit_of[ti] = e;
    set_child_box(bvh->nodes[e], 0, bvh->topo[t.left].box);
    set_child_box(bvh->nodes[e], 1, bvh->topo[t.right].box);
    int32_t c0 = emit(t.left);
    int32_t c1 = emit(t.right);
    bvh->nodes[e].child0 = c0; bvh->nodes[e].child1 = c1;
    return e;
  }
};

}  // namespace

// Threads of the host builder: RT_BUILD_THREADS, else min(cores visible, the cgroup's CPU quota rounded up, 16).
static int build_threads() {
  if (const char* e = getenv("RT_BUILD_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 64); }
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0; char word[32] = {0};
    if (fscanf(f, "%31s %lld", word, &period) == 2 && strcmp(word, "max") != 0 && period > 0) {
      quota = atoll(word);
      if (quota > 0) n = std::min<int>(n, (int)((quota + period - 1) / period));
    }
    fclose(f);
  }
  return std::max(1, std::min(n, 16));
}

static void build_bvh_impl(const Aabb* prim_boxes, uint32_t n, int max_leaf, int max_depth, bool direct_ids, BuiltBvh& out, float trav_cost = 1.0f) {
  out = BuiltBvh{};
  box_reset(out.bounds);
  Builder b;
  b.max_leaf = std::max(1, std::min(8, max_leaf));
  b.max_depth = std::max(max_depth, levels_needed(std::max(1u, n), b.max_leaf) + 1);
  b.out = &out;
  b.trav_cost = trav_cost;
  b.refs.resize(n);
  for (uint32_t i = 0; i < n; i++) {
    b.refs[i].box = prim_boxes[i]; b.refs[i].id = i;
    for (int k = 0; k < 3; k++) b.refs[i].c[k] = 0.5f * prim_boxes[i].lo[k] + 0.5f * prim_boxes[i].hi[k];
    box_grow(out.bounds, prim_boxes[i]);
  }
  b.threads = n >= 2 * Builder::PAR_MIN ? build_threads() : 1;
  const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto t_a = now();
  if (n > 0) b.run(n);
  auto t_b = now();
  out.order.resize(n);
  for (uint32_t i = 0; i < n; i++) out.order[i] = b.refs[i].id;
  out.emit_of.assign(out.topo.size(), -1);
  Emitter em{&out, direct_ids};
  if (n == 0 || out.topo[0].left < 0) {
    // root must be interior: wrap the single leaf (or nothing) in a synthetic node
    BvhNode root{};
    set_child_box(root, 0, n ? out.topo[0].box : missing_box());
    set_child_box(root, 1, missing_box());
    root.child0 = n ? em.leaf_ref(out.topo[0]) : ~0;
    root.child1 = ~0;
    out.nodes.push_back(root);
  } else {
    em.emit(0);
  }
  if (timing && n > 1000) fprintf(stderr, "[bvh_build] n %u threads %d: tree %.1f ms, emit %.1f ms\n", n, b.threads,
                                  std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(now() - t_b).count());
}

void build_bvh(const Aabb* prim_boxes, uint32_t n, int max_leaf, int max_depth, BuiltBvh& out) {
  build_bvh_impl(prim_boxes, n, max_leaf, max_depth, max_leaf == 1, out);
}

static void refit_rec(const Aabb* prim_boxes, BuiltBvh& bvh, int ti) {
  BuildNode& t = bvh.topo[ti];
  if (t.left < 0) {
    box_reset(t.box);
    for (uint32_t i = t.first; i < t.first + t.count; i++) box_grow(t.box, prim_boxes[bvh.order[i]]);
    return;
  }
  refit_rec(prim_boxes, bvh, t.left);
  refit_rec(prim_boxes, bvh, t.right);
  box_reset(t.box);
  box_grow(t.box, bvh.topo[t.left].box);
  box_grow(t.box, bvh.topo[t.right].box);
  int e = bvh.emit_of[ti];
  if (e >= 0) {
    set_child_box(bvh.nodes[e], 0, bvh.topo[t.left].box);
    set_child_box(bvh.nodes[e], 1, bvh.topo[t.right].box);
  }
}

void refit_bvh(const Aabb* prim_boxes, BuiltBvh& bvh) {
  if (bvh.topo.empty()) return;
  refit_rec(prim_boxes, bvh, 0);
  bvh.bounds = bvh.topo[0].box;
  if (bvh.topo[0].left < 0) set_child_box(bvh.nodes[0], 0, bvh.topo[0].box);
}

// ---------------------------------------------------------------------------------------------
namespace {
inline void set_child4(Bvh4Child& c, const Aabb& b, int32_t ref) {
  c.lo[0] = b.lo[0]; c.lo[1] = b.lo[1]; c.lo[2] = b.lo[2];
  c.hix = b.hi[0]; c.hiy = b.hi[1]; c.hiz = b.hi[2];
  c.ref = ref; c.pad = 0;
}
struct Collapser {
  const BuiltBvh* b2; Bvh4* out; bool area_driven, direct_ids;
  int32_t leaf_ref(const BuildNode& t) const {
    if (direct_ids) return ~(int32_t)b2->order[t.first];
    return ~(int32_t)((t.first << 3) | (t.count - 1));
  }
  // worst case of the traversal stack below node e: all k children hit -> k-1 pushed while inside one
  int stack_need(int32_t e) const {
    int k = 0, deepest = 0;
    for (int c = 0; c < 4; c++) {
      const int32_t ref = out->nodes[e].c[c].ref;
      if (ref == 0x7FFFFFFF) continue;
      k++;
      if (ref >= 0) deepest = std::max(deepest, stack_need(ref));
    }
    return std::max(0, k - 1) + deepest;
  }
  int32_t emit(int ti, int depth) {
    out->depth = std::max(out->depth, depth);
    int kids[4]; int nk = 2;
    kids[0] = b2->topo[ti].left; kids[1] = b2->topo[ti].right;
    while (nk < 4) {
      int pick = -1; float best = -1.f;
      for (int k = 0; k < nk; k++) {
        const BuildNode& c = b2->topo[kids[k]];
        if (c.left < 0) continue;
        if (!area_driven) { pick = k; break; }
        float a = half_area(c.box);
        if (a > best) { best = a; pick = k; }
      }
      if (pick < 0) break;
      const BuildNode& c = b2->topo[kids[pick]];
      for (int k = nk; k > pick + 1; k--) kids[k] = kids[k - 1];
      kids[pick + 1] = c.right; kids[pick] = c.left;
      nk++;
    }
    int32_t e = (int32_t)out->nodes.size();
    out->nodes.push_back(Bvh4Node{});
    out->child_topo.resize(out->child_topo.size() + 4, -1);
    for (int k = 0; k < 4; k++) {
      if (k < nk) {
        const BuildNode& c = b2->topo[kids[k]];
        out->child_topo[4 * (size_t)e + k] = kids[k];
        int32_t ref = c.left < 0 ? leaf_ref(c) : emit(kids[k], depth + 1);
        set_child4(out->nodes[e].c[k], c.box, ref);
      } else {
        set_child4(out->nodes[e].c[k], missing_box(), 0x7FFFFFFF);
      }
    }
    return e;
  }
};
}  // namespace

void collapse_bvh4(const BuiltBvh& b2, bool area_driven, bool direct_ids, Bvh4& out) {
  out = Bvh4{};
  Collapser c{&b2, &out, area_driven, direct_ids};
  if (b2.topo.empty() || b2.topo[0].left < 0) {
    Bvh4Node root{};
    out.child_topo.assign(4, -1);
    for (int k = 0; k < 4; k++) set_child4(root.c[k], missing_box(), 0x7FFFFFFF);
    if (!b2.topo.empty()) { set_child4(root.c[0], b2.topo[0].box, c.leaf_ref(b2.topo[0])); out.child_topo[0] = 0; }
    out.nodes.push_back(root);
    return;
  }
  c.emit(0, 0);
  out.stack_need = c.stack_need(0);
}

void refit_bvh4(const BuiltBvh& b2, Bvh4& b4) {
  for (size_t e = 0; e < b4.nodes.size(); e++)
    for (int k = 0; k < 4; k++) {
      int32_t t = b4.child_topo[4 * e + k];
      if (t < 0) continue;
      const Aabb& b = b2.topo[t].box;
      Bvh4Child& c = b4.nodes[e].c[k];
      c.lo[0] = b.lo[0]; c.lo[1] = b.lo[1]; c.lo[2] = b.lo[2]; c.hix = b.hi[0]; c.hiy = b.hi[1]; c.hiz = b.hi[2];
    }
}

static bool box_valid(const float* bx) { return bx[0] < 1e37f; }

void bvh2_bounds(const BuiltBvh& bvh, double lo[3], double hi[3]) {
  // bounds over every child box actually stored (refits can move them); accumulates into lo / hi
  for (const BvhNode& n : bvh.nodes) {
    const float c0[6] = {n.a[0], n.a[1], n.a[2], n.a[3], n.c[0], n.c[1]}, c1[6] = {n.b[0], n.b[1], n.b[2], n.b[3], n.c[2], n.c[3]};
    for (const float* c : {c0, c1}) {
      if (!box_valid(c)) continue;
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)c[2 * k]); hi[k] = std::max(hi[k], (double)c[2 * k + 1]); }
    }
  }
}

void quant_params(const double lo_in[3], const double hi_in[3], float q_lo[3], float q_scale[3]) {
  for (int k = 0; k < 3; k++) {
    double lo = lo_in[k], hi = hi_in[k];
    if (lo > hi) { lo = hi = 0.0; }
    const double ext = hi - lo;
    const double scale = ext > 0 ? ext * (1.0 + 1e-6) / 65530.0 : 1e-30;
    q_lo[k] = (float)(lo - 2.0 * scale);           // quanta 0,1 stay below every stored plane
    q_scale[k] = (float)scale;
  }
}

void quantize_bvh2(const BuiltBvh& bvh, std::vector<BvhNodeQ>& out, float q_lo[3], float q_scale[3]) {
  double lo[3] = {3e38, 3e38, 3e38}, hi[3] = {-3e38, -3e38, -3e38};
  bvh2_bounds(bvh, lo, hi);
  quant_params(lo, hi, q_lo, q_scale);
  quantize_bvh2_in(bvh, out, q_lo, q_scale);
}

void quantize_bvh2_in(const BuiltBvh& bvh, std::vector<BvhNodeQ>& out, const float q_lo[3], const float q_scale[3]) {
  auto valid = box_valid;
  // the kernel uses the float values: the doubles are re-derived from them so that rounding of base/scale
  // cannot make a box non-conservative
  double scale[3], base[3];
  for (int k = 0; k < 3; k++) { base[k] = q_lo[k]; scale[k] = q_scale[k]; }
  auto qdn = [&](double x, int k) { double q = std::floor((x - base[k]) / scale[k]) - 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, q)); };
  auto qup = [&](double x, int k) { double q = std::ceil((x - base[k]) / scale[k]) + 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, q)); };
  out.resize(bvh.nodes.size());
  for (size_t i = 0; i < bvh.nodes.size(); i++) {
    const BvhNode& n = bvh.nodes[i];
    float c0[6] = {n.a[0], n.a[1], n.a[2], n.a[3], n.c[0], n.c[1]}, c1[6] = {n.b[0], n.b[1], n.b[2], n.b[3], n.c[2], n.c[3]};
    int32_t r0 = n.child0, r1 = n.child1;
    const bool v0 = valid(c0), v1 = valid(c1);
    // The absent child of a synthetic single-child root (one-instance TLAS, single-leaf BLAS) gets an INVERTED box
    // (lower plane 0xFFFF, upper plane 0): no ray can enter it, so the sibling is never visited twice.  Its link
    // repeats the sibling's so that every link in the array stays a valid reference for the tree walkers.
    const uint32_t kNever = 0x0000FFFFu;
    if (!v1) r1 = r0;
    if (!v0) r0 = r1;
    BvhNodeQ q{};
    for (int k = 0; k < 3; k++) {
      q.w[k] = v0 ? (qdn(c0[2 * k], k) | (qup(c0[2 * k + 1], k) << 16)) : kNever;
      q.w[3 + k] = v1 ? (qdn(c1[2 * k], k) | (qup(c1[2 * k + 1], k) << 16)) : kNever;
    }
    q.child0 = r0; q.child1 = r1;
    out[i] = q;
  }
}

void widen_bvh2(const BvhNodeQ* nodes, size_t count, int32_t base, WideNodeQ* out) {
  for (size_t i = 0; i < count; i++) {
    const BvhNodeQ& n = nodes[i];
    uint32_t ex[4], ey[4], ez[4]; int32_t er[4]; int ne = 0;
    auto add = [&](const uint32_t* w, int32_t ref) {
      for (int j = 0; j < ne; j++) if (er[j] == ref) return;   // a missing child is stored as a copy of its sibling
      ex[ne] = w[0]; ey[ne] = w[1]; ez[ne] = w[2]; er[ne] = ref; ne++;
    };
    auto expand = [&](const uint32_t* w, int32_t ref) {
      if (ref >= 0) { const BvhNodeQ& c = nodes[ref - base]; add(c.w, c.child0); add(c.w + 3, c.child1); }
      else add(w, ref);
    };
    expand(n.w, n.child0);
    if (n.child1 != n.child0) expand(n.w + 3, n.child1);
    WideNodeQ o;
    for (int k = 0; k < 4; k++) {
      if (k < ne) { o.x[k] = ex[k]; o.y[k] = ey[k]; o.z[k] = ez[k]; o.ref[k] = er[k]; }
      else { o.x[k] = o.y[k] = o.z[k] = 0u; o.ref[k] = er[0]; }   // point box at quantum 0: below every stored plane
    }
    out[i] = o;
  }
}

int bvh2_levels(const BvhNodeQ* nodes, size_t count, int32_t root) {
  if (count == 0) return 0;
  std::vector<std::pair<int32_t, int>> todo;
  todo.push_back({root, 1});
  int levels = 0; size_t seen = 0;
  while (!todo.empty()) {
    auto [i, d] = todo.back(); todo.pop_back();
    if (i < 0 || (size_t)i >= count || ++seen > count) return -1;
    levels = std::max(levels, d);
    const BvhNodeQ& n = nodes[i];
    if (n.child0 >= 0) todo.push_back({n.child0, d + 1});
    if (n.child1 >= 0 && n.child1 != n.child0) todo.push_back({n.child1, d + 1});
  }
  return levels;
}

void build_blas(const float* verts6, const uint32_t* idx, uint32_t n_prims, BuiltBvh& bvh, std::vector<TriPacket>& tris) {
  std::vector<Aabb> boxes(n_prims);
  for (uint32_t p = 0; p < n_prims; p++) {
    Aabb b; box_reset(b);
    for (int c = 0; c < 3; c++) {
      const float* v = verts6 + 6ull * idx[3ull * p + c];
      for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], v[k]); b.hi[k] = std::max(b.hi[k], v[k]); }
    }
    boxes[p] = b;
  }
  // experiment knobs (defaults are the measured best): RT_BVH_MAX_LEAF 1..8, RT_BVH_TRAV_COST
  int max_leaf = 4; float trav_cost = 1.0f;
  if (const char* e = getenv("RT_BVH_MAX_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 8) max_leaf = v; }
  if (const char* e = getenv("RT_BVH_TRAV_COST")) { float v = (float)atof(e); if (v > 0.f) trav_cost = v; }
  build_bvh_impl(boxes.data(), n_prims, max_leaf, BLAS_MAX_DEPTH, false, bvh, trav_cost);
  tris.resize(n_prims);
  for (uint32_t i = 0; i < n_prims; i++) {
    uint32_t p = bvh.order[i];
    const float* v0 = verts6 + 6ull * idx[3ull * p + 0];
    const float* v1 = verts6 + 6ull * idx[3ull * p + 1];
    const float* v2 = verts6 + 6ull * idx[3ull * p + 2];
    TriPacket& t = tris[i];
    for (int k = 0; k < 3; k++) {
      t.v0[k] = v0[k];
      t.e1[k] = v1[k] - v0[k];   // one binary32 rounding, identical to the oracle's v1 - v0
      t.e2[k] = v2[k] - v0[k];
    }
    t.prim = p; t.pad[0] = t.pad[1] = 0;
  }
}

}  // namespace rt
