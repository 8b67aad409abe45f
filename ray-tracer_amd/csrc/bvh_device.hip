// The acceleration structure built on the GPU: a linear BVH over Morton-sorted item boxes (Karras,
// "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012), collapsed into
// the 4-wide nodes the traversal walks (rt_types.h BvhNode).  The host builder (bvh_build.cpp, binned
// SAH) gives the better tree and stays the default for scenes that are built once; this one is for
// scenes that are built often or are large: a million triangles in milliseconds.
//
//   1. Morton key of every item's box centre (21 bits per axis), radix sort of (key, item)
//   2. the binary radix tree over the sorted keys: one thread per internal node
//   3. boxes and the kinds of items below every internal node: unions over its key range, read from a
//      reduction tree over the sorted leaves (no thread waits for another)
//   4. where the triangles / quadrics of a range start in the per-kind primitive orders (prefix sums):
//      a subtree of at most four items of one kind becomes one primitive run
//   5. the 4-wide nodes, top down, one launch pair per level: a node starts from the two children of a
//      binary node and keeps replacing the child with the largest box by its two children until four
//      slots are used; the children that are binary nodes themselves become the next level's work.
//      Node numbers are handed out by a prefix sum over the level, so the build is reproducible.
//   6. the traversal stack a ray can need, bottom up over the levels
//
// Any conservative hierarchy over the same leaves returns the same hits (the reference scans every
// child of a group, src/shape.rs:1213-1219), so images do not depend on which builder ran.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"
#include "bvh_build.h"
#include "rt_error.h"

namespace rt {
namespace {

#define BD_TRY(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      rc = rt_fail_message(RT_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str());    \
      goto done;                                                                                            \
    }                                                                                                       \
  } while (0)

constexpr int THREADS = 256;
constexpr uint32_t LEAF_BIT = 0x80000000u;  // in a binary child link: the rest is a sorted leaf position
constexpr uint32_t MAX_RUN_ITEMS = 4;       // items of one kind a primitive run holds at most (as the host builder)

struct Tree {
  // items, by item id
  const float* box;         // lo xyz, hi xyz
  const uint8_t* kind;      // KIND_TRI / KIND_QUAD / KIND_GATE
  const uint32_t* index;    // source triangle / quadric, or gate number
  const uint32_t* gate_alt; // by gate number: what a ray that may skip gates walks instead (REF_NONE entries: the gate itself)
  const int32_t* gate_need; // by gate number: stack entries the subtree behind the gate can need
  uint32_t n_gate_alt, n_gate_need;
  uint32_t n;               // items
  // sorted leaves
  const unsigned long long* key;
  const uint32_t* item;     // item id at every sorted position
  const uint32_t* tri_pre;  // triangles before every sorted position (exclusive), n + 1 entries
  const uint32_t* quad_pre;
  // binary internal nodes 0 .. n - 2, root 0
  uint32_t *left, *right, *first, *last, *kmask;
  float* ibox;
};

__device__ __forceinline__ int delta(const Tree& t, int i, int j) {
  if (j < 0 || j >= (int)t.n) return -1;
  const unsigned long long a = t.key[i], b = t.key[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz(i ^ j);  // equal keys: the positions decide
}

__global__ void __launch_bounds__(THREADS) k_morton(const float* box, uint32_t n, float3 lo, float3 inv_extent, unsigned long long* key,
                                                    uint32_t* item) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n) return;
  const float* b = box + (size_t)i * 6;
  const float c[3] = {0.5f * (b[0] + b[3]), 0.5f * (b[1] + b[4]), 0.5f * (b[2] + b[5])};
  const float l[3] = {lo.x, lo.y, lo.z}, s[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
  unsigned long long k = 0;
  uint32_t q[3];
  for (int a = 0; a < 3; a++) {
    float u = (c[a] - l[a]) * s[a];
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    q[a] = (uint32_t)fminf(u * 2097152.0f, 2097151.0f);  // 21 bits
  }
  for (int bit = 20; bit >= 0; bit--)
    for (int a = 0; a < 3; a++) k = (k << 1) | ((q[a] >> bit) & 1u);
  key[i] = k;
  item[i] = i;
}

__global__ void __launch_bounds__(THREADS) k_kind_flags(Tree t, uint32_t* tri_flag, uint32_t* quad_flag) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i > t.n) return;
  const uint8_t k = i < t.n ? t.kind[t.item[i]] : 0;
  tri_flag[i] = k == KIND_TRI ? 1u : 0u;
  quad_flag[i] = k == KIND_QUAD ? 1u : 0u;
}

// Step 2 (Karras 2012, figure 4): the range of keys internal node i covers and where it splits.
__global__ void __launch_bounds__(THREADS) k_radix_tree(Tree t) {
  const int i = (int)(blockIdx.x * THREADS + threadIdx.x);
  if (i >= (int)t.n - 1) return;
  const int d = delta(t, i, i + 1) - delta(t, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(t, i, i - d);
  int lmax = 2;
  while (delta(t, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int step = lmax / 2; step >= 1; step /= 2)
    if (delta(t, i, i + (l + step) * d) > dmin) l += step;
  const int j = i + l * d;
  const int dnode = delta(t, i, j);
  int s = 0;
  for (int div = 2;; div *= 2) {
    const int step = (l + div - 1) / div;
    if (delta(t, i, i + (s + step) * d) > dnode) s += step;
    if (step <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const uint32_t left = lo == gamma ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
  const uint32_t right = hi == gamma + 1 ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  t.left[i] = left;
  t.right[i] = right;
  t.first[i] = (uint32_t)lo;
  t.last[i] = (uint32_t)hi;
}

__device__ __forceinline__ uint32_t kind_bit(uint8_t k) { return k == KIND_TRI ? 1u : (k == KIND_QUAD ? 2u : 4u); }

// Step 3.  The box of a binary node is the union of the leaf boxes of its key range.  A complete
// reduction tree over the sorted leaves (level k holds the unions of 2^k consecutive leaves; one plain
// launch per level, so no thread waits for another and nothing has to be made visible between compute
// units inside a launch) answers any range with at most two entries per level.
struct Seg {  // one entry: box and the kinds of items below
  float lo[3], hi[3];
  uint32_t mask;
};
__device__ __forceinline__ void seg_join(Seg& a, const Seg& b) {
  for (int k = 0; k < 3; k++) {
    a.lo[k] = fminf(a.lo[k], b.lo[k]);
    a.hi[k] = fmaxf(a.hi[k], b.hi[k]);
  }
  a.mask |= b.mask;
}

__global__ void __launch_bounds__(THREADS) k_seg_leaves(Tree t, Seg* level0) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= t.n) return;
  const uint32_t id = t.item[i];
  const float* b = t.box + (size_t)id * 6;
  Seg s;
  for (int a = 0; a < 3; a++) {
    s.lo[a] = b[a];
    s.hi[a] = b[3 + a];
  }
  s.mask = kind_bit(t.kind[id]);
  level0[i] = s;
}

__global__ void __launch_bounds__(THREADS) k_seg_level(const Seg* below, uint32_t n_below, Seg* level, uint32_t n_level) {
  const uint32_t j = blockIdx.x * THREADS + threadIdx.x;
  if (j >= n_level) return;
  Seg s = below[2 * j];
  if (2 * j + 1 < n_below) seg_join(s, below[2 * j + 1]);
  level[j] = s;
}

constexpr int SEG_LEVELS = 28;  // 2^27 items at most
struct SegTree {
  const Seg* level[SEG_LEVELS];
};

__global__ void __launch_bounds__(THREADS) k_boxes(Tree t, SegTree seg) {
  const uint32_t node = blockIdx.x * THREADS + threadIdx.x;
  if (node + 1 >= t.n) return;
  uint32_t l = t.first[node], r = t.last[node] + 1u;
  Seg acc{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}, 0u};
  for (int k = 0; l < r; k++) {
    if (l & 1u) seg_join(acc, seg.level[k][l++]);
    if (r & 1u) seg_join(acc, seg.level[k][--r]);
    l >>= 1;
    r >>= 1;
  }
  float* out = t.ibox + (size_t)node * 6;
  for (int a = 0; a < 3; a++) {
    out[a] = acc.lo[a];
    out[3 + a] = acc.hi[a];
  }
  t.kmask[node] = acc.mask;
}

// a binary link that ends the 4-wide hierarchy: a leaf, or a subtree that becomes one primitive run
__device__ __forceinline__ bool is_run_root(const Tree& t, uint32_t inner) {
  const uint32_t m = t.kmask[inner];
  return (m == 1u || m == 2u) && t.last[inner] - t.first[inner] + 1u <= MAX_RUN_ITEMS;
}
__device__ __forceinline__ bool expandable(const Tree& t, uint32_t link) { return !(link & LEAF_BIT) && !is_run_root(t, link); }

__device__ __forceinline__ const float* link_box(const Tree& t, uint32_t link) {
  return (link & LEAF_BIT) ? t.box + (size_t)t.item[link & ~LEAF_BIT] * 6 : t.ibox + (size_t)link * 6;
}

__device__ uint32_t link_ref(const Tree& t, uint32_t link) {  // for links that are not expandable
  uint32_t first, count, kmask;
  if (link & LEAF_BIT) {
    first = link & ~LEAF_BIT;
    count = 1;
    const uint32_t id = t.item[first];
    const uint8_t k = t.kind[id];
    if (k == KIND_GATE) return (KIND_GATE << 30) | t.index[id];
    kmask = kind_bit(k);
  } else {
    first = t.first[link];
    count = t.last[link] - first + 1u;
    kmask = t.kmask[link];
  }
  return kmask == 1u ? (KIND_TRI << 30) | ((count - 1u) << 27) | t.tri_pre[first]
                     : (KIND_QUAD << 30) | ((count - 1u) << 27) | t.quad_pre[first];
}

struct Work {
  uint32_t inner;  // binary node
  uint32_t node;   // 4-wide node it becomes
};

// Step 5a: the slots of every node of this level, and how many of them are nodes of the next level.
__global__ void __launch_bounds__(THREADS) k_collapse_plan(Tree t, const Work* work, uint32_t n_work, uint4* slots, uint32_t* n_inner) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w > n_work) return;
  if (w == n_work) {  // one entry past the end so that the exclusive sum ends in the total
    n_inner[w] = 0;
    return;
  }
  const uint32_t b = work[w].inner;
  uint32_t s[4] = {t.left[b], t.right[b], 0xFFFFFFFFu, 0xFFFFFFFFu};
  int ns = 2;
  while (ns < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < ns; k++)
      if (expandable(t, s[k])) {
        const float* bx = link_box(t, s[k]);
        const float dx = bx[3] - bx[0], dy = bx[4] - bx[1], dz = bx[5] - bx[2];
        const float area = dx * dy + dy * dz + dz * dx;
        if (area > best_area) {
          best_area = area;
          best = k;
        }
      }
    if (best < 0) break;
    const uint32_t e = s[best];
    s[best] = s[ns - 1];
    s[ns - 1] = t.left[e];
    s[ns] = t.right[e];
    ns++;
  }
  uint32_t inner = 0;
  for (int k = 0; k < ns; k++) inner += expandable(t, s[k]) ? 1u : 0u;
  slots[w] = make_uint4(s[0], s[1], s[2], s[3]);
  n_inner[w] = inner;
}

// Step 5b: writes the nodes of this level and the work of the next one.
__global__ void __launch_bounds__(THREADS) k_collapse_emit(Tree t, const Work* work, uint32_t n_work, const uint4* slots,
                                                           const uint32_t* offset, uint32_t next_base, BvhNode* nodes, Work* next) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w >= n_work) return;
  const uint4 sl = slots[w];
  const uint32_t s[4] = {sl.x, sl.y, sl.z, sl.w};
  BvhNode& out = nodes[work[w].node];
  uint32_t k_inner = 0;
  for (int c = 0; c < 4; c++) {
    if (s[c] == 0xFFFFFFFFu) {
      for (int a = 0; a < 3; a++) {
        out.plane[a][0][c] = INFINITY;
        out.plane[a][1][c] = -INFINITY;
      }
      out.child[c] = out.alt[c] = REF_NONE;
      continue;
    }
    const float* bx = link_box(t, s[c]);
    for (int a = 0; a < 3; a++) {
      out.plane[a][0][c] = bx[a];
      out.plane[a][1][c] = bx[3 + a];
    }
    uint32_t ref, alt;
    if (expandable(t, s[c])) {
      const uint32_t idx = next_base + offset[w] + k_inner;
      next[offset[w] + k_inner] = Work{s[c], idx};
      k_inner++;
      ref = alt = (KIND_INNER << 30) | idx;
    } else {
      ref = alt = link_ref(t, s[c]);
      if ((ref >> 30) == KIND_GATE) {
        const uint32_t g = ref & 0x3FFFFFFFu;
        if (g < t.n_gate_alt) alt = t.gate_alt[g];
      }
    }
    out.child[c] = ref;
    out.alt[c] = alt;
  }
}

// Step 6: stack entries a traversal below each node of one level can hold at once (deeper levels done).
__global__ void __launch_bounds__(THREADS) k_stack_need(Tree t, const Work* work, uint32_t n_work, const BvhNode* nodes, int32_t* need) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w >= n_work) return;
  const BvhNode& nd = nodes[work[w].node];
  int children = 0, deepest = 0;
  for (int c = 0; c < 4; c++) {
    const uint32_t ref = nd.child[c];
    if (ref == REF_NONE) continue;
    children++;
    int below = 0;
    if ((ref >> 30) == KIND_INNER) below = need[ref & 0x3FFFFFFFu];
    else if ((ref >> 30) == KIND_GATE && (ref & 0x3FFFFFFFu) < t.n_gate_need) below = t.gate_need[ref & 0x3FFFFFFFu];
    deepest = max(deepest, below);
  }
  need[work[w].node] = (children - 1) + deepest;
}

struct Scan {  // exclusive sums with one reusable workspace
  void* tmp = nullptr;
  size_t bytes = 0;
  hipError_t run(const uint32_t* in, uint32_t* out, size_t n) {
    size_t need = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n);
    if (e != hipSuccess) return e;
    if (need > bytes) {
      if (tmp) (void)hipFree(tmp);
      tmp = nullptr;
      bytes = 0;
      e = hipMalloc(&tmp, need);
      if (e != hipSuccess) return e;
      bytes = need;
    }
    return hipcub::DeviceScan::ExclusiveSum(tmp, need, in, out, (int)n);
  }
  ~Scan() {
    if (tmp) (void)hipFree(tmp);
  }
};

// Every temporary of one build comes out of one device allocation.
struct Arena {
  char* base = nullptr;
  size_t used = 0, capacity = 0;
  bool measuring = true;  // first pass adds the sizes up, second pass hands out the pieces
  template <class T>
  void take(T** p, size_t n) {
    const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
    *p = measuring ? nullptr : reinterpret_cast<T*>(base + used);
    used += bytes;
  }
  ~Arena() {
    if (base) (void)hipFree(base);
  }
};

}  // namespace

int build_bvh_device(const BuildItem* items, size_t n_items, const std::vector<uint32_t>& gate_alt, const std::vector<int>& gate_need,
                     DeviceBvh* out) {
  out->nodes.clear();
  out->tri_order.clear();
  out->quad_order.clear();
  out->root = REF_NONE;
  out->stack_need = 0;
  if (n_items < 2 || n_items > (1u << 27)) return rt_fail_message(RT_ERR_UNSUPPORTED, "device build: item count out of range");
  const uint32_t n = (uint32_t)n_items;
  int rc = RT_OK;
  const auto t0 = std::chrono::steady_clock::now();
  static const bool timing = getenv("RT_AMD_BUILD_TIMING") != nullptr;
  auto mark = [&, last = t0](const char* what) mutable {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[rt_amd]   device build: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  };
  // ---- host: boxes rounded outward to f32, the box of the centres
  // plain arrays, not vectors: nothing fills them before the slices below do (a vector would zero 30 MB first)
  std::unique_ptr<float[]> h_box(new float[(size_t)n * 6]);
  std::unique_ptr<uint8_t[]> h_kind(new uint8_t[n]);
  std::unique_ptr<uint32_t[]> h_index(new uint32_t[n]);
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  {
    // one slice of the items per thread; a float moved one step outward when the rounding went inward
    const unsigned workers = n < 65536 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::array<float, 6>> part(workers, std::array<float, 6>{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY});
    auto slice = [&](unsigned w) {
      const uint32_t lo_i = (uint32_t)((uint64_t)n * w / workers), hi_i = (uint32_t)((uint64_t)n * (w + 1) / workers);
      std::array<float, 6>& acc = part[w];
      for (uint32_t i = lo_i; i < hi_i; i++) {
        const BuildItem& it = items[i];
        for (int a = 0; a < 3; a++) {
          float l = (float)it.lo[a], h = (float)it.hi[a];
          if ((double)l > it.lo[a]) l = std::nextafterf(l, -INFINITY);
          if ((double)h < it.hi[a]) h = std::nextafterf(h, INFINITY);
          h_box[(size_t)i * 6 + a] = l;
          h_box[(size_t)i * 6 + 3 + a] = h;
          const float c = 0.5f * (l + h);
          acc[a] = std::fmin(acc[a], c);
          acc[3 + a] = std::fmax(acc[3 + a], c);
        }
        h_kind[i] = (uint8_t)it.kind;
        h_index[i] = it.index;
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 1; w < workers; w++) pool.emplace_back(slice, w);
    slice(0);
    for (std::thread& th : pool) th.join();
    for (const std::array<float, 6>& acc : part)
      for (int a = 0; a < 3; a++) {
        clo[a] = std::fmin(clo[a], acc[a]);
        chi[a] = std::fmax(chi[a], acc[3 + a]);
      }
  }
  std::vector<int32_t> h_need(gate_need.begin(), gate_need.end());
  // One scale for the three axes: an axis along which the items hardly spread (the height of a terrain)
  // must not take a third of the key's leading bits, or the tree splits where nothing is separated.
  float3 lo = make_float3(clo[0], clo[1], clo[2]);
  const float widest = std::fmax(std::fmax(chi[0] - clo[0], chi[1] - clo[1]), chi[2] - clo[2]);
  const float inv = widest > 0.0f && std::isfinite(widest) ? 1.0f / widest : 0.0f;
  float3 inv_extent = make_float3(inv, inv, inv);
  mark("boxes on the host");
  float *d_box = nullptr, *d_ibox = nullptr;
  uint8_t* d_kind = nullptr;
  uint32_t *d_index = nullptr, *d_gate_alt = nullptr, *d_item = nullptr, *d_item2 = nullptr, *d_tri_flag = nullptr, *d_quad_flag = nullptr,
           *d_tri_pre = nullptr, *d_quad_pre = nullptr, *d_left = nullptr, *d_right = nullptr, *d_first = nullptr, *d_last = nullptr,
           *d_kmask = nullptr, *d_ninner = nullptr, *d_offset = nullptr;
  int32_t *d_gate_need = nullptr, *d_need = nullptr;
  unsigned long long *d_key = nullptr, *d_key2 = nullptr;
  void* d_sort_tmp = nullptr;
  BvhNode* d_nodes = nullptr;
  Seg* d_seg = nullptr;    // the reduction tree over the sorted leaves: n + n / 2 + ... entries
  Work* d_work = nullptr;  // all levels, one after the other
  uint4* d_slots = nullptr;
  Scan scan;
  Arena arena;
  std::vector<uint32_t> level_first, level_count;
  uint32_t n_nodes = 1, totals[2] = {0, 0};
  const dim3 block(THREADS);
  auto grid = [](size_t k) { return dim3((unsigned)((k + THREADS - 1) / THREADS)); };

  for (int pass = 0; pass < 2; pass++) {
    arena.used = 0;
    arena.take(&d_box, (size_t)n * 6);
    arena.take(&d_kind, n);
    arena.take(&d_index, n);
    arena.take(&d_gate_alt, gate_alt.size());
    arena.take(&d_gate_need, h_need.size());
    arena.take(&d_key, n);
    arena.take(&d_key2, n);
    arena.take(&d_item, n);
    arena.take(&d_item2, n);
    arena.take(&d_tri_flag, (size_t)n + 1);
    arena.take(&d_quad_flag, (size_t)n + 1);
    arena.take(&d_tri_pre, (size_t)n + 1);
    arena.take(&d_quad_pre, (size_t)n + 1);
    arena.take(&d_left, n);
    arena.take(&d_right, n);
    arena.take(&d_first, n);
    arena.take(&d_last, n);
    arena.take(&d_kmask, n);
    arena.take(&d_ibox, (size_t)n * 6);
    arena.take(&d_nodes, n);       // fewer 4-wide nodes than binary ones
    arena.take(&d_work, (size_t)n + 1);
    arena.take(&d_slots, n);
    arena.take(&d_ninner, (size_t)n + 1);
    arena.take(&d_offset, (size_t)n + 1);
    arena.take(&d_need, n);
    arena.take(&d_seg, 2 * (size_t)n + SEG_LEVELS);
    if (pass == 0) {
      BD_TRY(hipMalloc((void**)&arena.base, arena.used));
      arena.capacity = arena.used;
      arena.measuring = false;
    }
  }
  BD_TRY(hipMemcpy(d_box, h_box.get(), (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
  BD_TRY(hipMemcpy(d_kind, h_kind.get(), n, hipMemcpyHostToDevice));
  BD_TRY(hipMemcpy(d_index, h_index.get(), (size_t)n * 4, hipMemcpyHostToDevice));
  if (!gate_alt.empty()) BD_TRY(hipMemcpy(d_gate_alt, gate_alt.data(), gate_alt.size() * 4, hipMemcpyHostToDevice));
  if (!h_need.empty()) BD_TRY(hipMemcpy(d_gate_need, h_need.data(), h_need.size() * 4, hipMemcpyHostToDevice));
  mark("allocations, upload");
  {
    // ---- 1. keys, sort
    hipLaunchKernelGGL(k_morton, grid(n), block, 0, 0, (const float*)d_box, n, lo, inv_extent, d_key, d_item);
    size_t sort_bytes = 0;
    BD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_key, d_key2, d_item, d_item2, (int)n, 0, 63));
    BD_TRY(hipMalloc(&d_sort_tmp, sort_bytes ? sort_bytes : 1));
    BD_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, sort_bytes, d_key, d_key2, d_item, d_item2, (int)n, 0, 63));
    mark("keys, sort");
    Tree t{};
    t.box = d_box; t.kind = d_kind; t.index = d_index;
    t.gate_alt = d_gate_alt; t.gate_need = d_gate_need;
    t.n_gate_alt = (uint32_t)gate_alt.size(); t.n_gate_need = (uint32_t)h_need.size();
    t.n = n;
    t.key = d_key2; t.item = d_item2; t.tri_pre = d_tri_pre; t.quad_pre = d_quad_pre;
    t.left = d_left; t.right = d_right; t.first = d_first; t.last = d_last;
    t.kmask = d_kmask; t.ibox = d_ibox;
    // ---- 4 (needs only the order). primitive orders: where every kind's items of a range start
    hipLaunchKernelGGL(k_kind_flags, grid((size_t)n + 1), block, 0, 0, t, d_tri_flag, d_quad_flag);
    BD_TRY(scan.run(d_tri_flag, d_tri_pre, (size_t)n + 1));
    BD_TRY(scan.run(d_quad_flag, d_quad_pre, (size_t)n + 1));
    // ---- 2, 3. the binary tree and its boxes
    mark("kind flags, prefix sums");
    hipLaunchKernelGGL(k_radix_tree, grid(n - 1), block, 0, 0, t);
    mark("radix tree");
    {
      SegTree seg{};
      Seg* base = d_seg;
      uint32_t n_level = n;
      hipLaunchKernelGGL(k_seg_leaves, grid(n), block, 0, 0, t, base);
      seg.level[0] = base;
      for (int k = 1; k < SEG_LEVELS && n_level > 1; k++) {
        const uint32_t n_below = n_level;
        Seg* below = base;
        base += n_below;
        n_level = (n_below + 1) / 2;
        hipLaunchKernelGGL(k_seg_level, grid(n_level), block, 0, 0, (const Seg*)below, n_below, base, n_level);
        seg.level[k] = base;
      }
      hipLaunchKernelGGL(k_boxes, grid(n - 1), block, 0, 0, t, seg);
    }
    BD_TRY(hipGetLastError());
    mark("boxes");
    // ---- 5. the 4-wide nodes, level by level.  The root binary node always becomes node 0: a range of
    // two to four items of one kind could be a single run, but then there would be no node to hold its box.
    const Work root_work{0u, 0u};
    BD_TRY(hipMemcpy(d_work, &root_work, sizeof(Work), hipMemcpyHostToDevice));
    uint32_t first = 0, count = 1;
    while (count > 0) {
      level_first.push_back(first);
      level_count.push_back(count);
      if (level_first.size() > 200) {
        rc = rt_fail_message(RT_ERR_UNSUPPORTED, "device build: hierarchy deeper than 200 levels");
        goto done;
      }
      hipLaunchKernelGGL(k_collapse_plan, grid((size_t)count + 1), block, 0, 0, t, (const Work*)(d_work + first), count, d_slots, d_ninner);
      BD_TRY(scan.run(d_ninner, d_offset, (size_t)count + 1));
      uint32_t next_count = 0;
      BD_TRY(hipMemcpy(&next_count, d_offset + count, 4, hipMemcpyDeviceToHost));
      if ((size_t)n_nodes + next_count > (size_t)n) {
        rc = rt_fail_message(RT_ERR_DEVICE, "device build: more nodes than items");
        goto done;
      }
      hipLaunchKernelGGL(k_collapse_emit, grid(count), block, 0, 0, t, (const Work*)(d_work + first), count, (const uint4*)d_slots,
                         (const uint32_t*)d_offset, n_nodes, d_nodes, d_work + first + count);
      BD_TRY(hipGetLastError());
      first += count;
      n_nodes += next_count;
      count = next_count;
    }
    mark("collapse to 4-wide nodes");
    // ---- 6. stack need, deepest level first
    for (size_t l = level_first.size(); l-- > 0;)
      hipLaunchKernelGGL(k_stack_need, grid(level_count[l]), block, 0, 0, t, (const Work*)(d_work + level_first[l]), level_count[l],
                         (const BvhNode*)d_nodes, d_need);
    BD_TRY(hipGetLastError());
    int32_t root_need = 0;
    BD_TRY(hipMemcpy(&root_need, d_need, 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals, d_tri_pre + n, 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals + 1, d_quad_pre + n, 4, hipMemcpyDeviceToHost));
    mark("stack need");
    // ---- results back to the host builder's vectors
    out->nodes.resize(n_nodes);
    BD_TRY(hipMemcpy(out->nodes.data(), d_nodes, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    std::vector<uint32_t> order(n);
    BD_TRY(hipMemcpy(order.data(), d_item2, (size_t)n * 4, hipMemcpyDeviceToHost));
    out->tri_order.reserve(totals[0]);
    out->quad_order.reserve(totals[1]);
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t id = order[i];
      if (h_kind[id] == KIND_TRI) out->tri_order.push_back(h_index[id]);
      else if (h_kind[id] == KIND_QUAD) out->quad_order.push_back(h_index[id]);
    }
    mark("download, primitive orders");
    out->root = make_ref(KIND_INNER, 0);
    out->stack_need = root_need;
    out->levels = (int)level_first.size();
  }
done:
  if (d_sort_tmp) (void)hipFree(d_sort_tmp);
  out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

}  // namespace rt
