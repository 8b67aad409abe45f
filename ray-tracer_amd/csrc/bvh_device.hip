// The acceleration structure built on the GPU: parallel locally-ordered clustering over Morton-sorted item
// boxes (PLOC: Meister and Bittner, "Parallel locally-ordered clustering for bounding volume hierarchy
// construction", 2018), collapsed into the 4-wide nodes the traversal walks (rt_types.h BvhNode).  The host
// builder (bvh_build.cpp, binned SAH) stays the default for scenes that are built once; this one is for scenes
// that are built often or are large: a million triangles in milliseconds.
//
//   1. Morton key of every item's box centre (21 bits per axis), radix sort of (key, item)
//   2. the binary tree, bottom up: the clusters (at first the sorted items) stay in one array; in every round
//      each cluster looks for the neighbour within PLOC_RADIUS places whose union with it has the smallest
//      surface, pairs that chose each other merge into a new node, and the array is compacted.  Clusters that
//      are close in space are close in the array, so the merges follow the surface-area heuristic where a
//      plain radix tree over the keys (the first version of this builder) splits at bit boundaries: on a
//      terrain of a million triangles a ray visits a third fewer nodes and triangles.
//   3. where every item lands in its kind's primitive order: the depth-first order of the tree, found by every
//      leaf and node walking up to the root and adding up what lies left of its path
//   4. the 4-wide nodes, top down, one launch pair per level: a node starts from the two children of a
//      binary node and keeps replacing the child with the largest box by its two children until four
//      slots are used; a subtree of at most four triangles becomes one primitive run (quadrics stay one per
//      leaf, as in the host builder); the children that are binary nodes themselves become the next level's
//      work.  Node numbers are handed out by a prefix sum over the level, so the build is reproducible.
//   5. the traversal stack a ray can need, bottom up over the levels
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
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;

struct Tree {
  // items, by item id
  const float* box;         // lo xyz, hi xyz
  const uint8_t* kind;      // KIND_TRI / KIND_QUAD / KIND_GATE
  const uint32_t* index;    // source triangle / quadric, or gate number
  const uint32_t* gate_alt; // by gate number: what a ray that may skip gates walks instead (REF_NONE entries: the gate itself)
  const int32_t* gate_need; // by gate number: stack entries the subtree behind the gate can need
  uint32_t n_gate_alt, n_gate_need;
  uint32_t n;               // items
  uint32_t max_run_tris;    // triangles a primitive run holds at most; quadrics stay one per leaf, as in the host builder
  // sorted leaves
  const uint32_t* item;     // item id at every sorted position
  uint32_t* leaf_parent;    // binary node above every sorted position
  uint32_t* leaf_kpos;      // where the item lands in its kind's primitive order
  // binary internal nodes 0 .. n - 2 in the order they were made, root n - 2
  uint32_t *left, *right, *parent, *kmask;
  uint32_t *n_tri, *n_quad;          // triangles / quadrics below
  uint32_t *tri_first, *quad_first;  // primitive-order position of the first triangle / quadric below
  float* ibox;
};

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

__device__ __forceinline__ uint32_t kind_bit(uint8_t k) { return k == KIND_TRI ? 1u : (k == KIND_QUAD ? 2u : 4u); }

// ---- step 2: the clusters of one round are (link, box) pairs in array order
__global__ void __launch_bounds__(THREADS) k_ploc_init(Tree t, uint32_t* cid, float* cbox) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= t.n) return;
  cid[i] = LEAF_BIT | i;
  const float* b = t.box + (size_t)t.item[i] * 6;
  for (int a = 0; a < 6; a++) cbox[(size_t)i * 6 + a] = b[a];
  t.leaf_parent[i] = NO_NODE;
}

// the neighbour whose union with cluster i has the smallest surface; ties go to the lower place
__global__ void __launch_bounds__(THREADS) k_ploc_nearest(const float* cbox, uint32_t n, uint32_t radius, uint32_t* nn) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n) return;
  float b[6];
  for (int a = 0; a < 6; a++) b[a] = cbox[(size_t)i * 6 + a];
  const uint32_t lo = i > radius ? i - radius : 0u, hi = min(n - 1u, i + radius);
  float best = INFINITY;
  uint32_t best_j = i;
  for (uint32_t j = lo; j <= hi; j++) {
    if (j == i) continue;
    const float* c = cbox + (size_t)j * 6;
    const float dx = fmaxf(b[3], c[3]) - fminf(b[0], c[0]), dy = fmaxf(b[4], c[4]) - fminf(b[1], c[1]),
                dz = fmaxf(b[5], c[5]) - fminf(b[2], c[2]);
    const float area = dx * dy + dy * dz + dz * dx;
    if (area < best) {
      best = area;
      best_j = j;
    }
  }
  nn[i] = best_j;
}

// per cluster: does it survive the round (high word), does it become a new node (low word); one entry past
// the end so that the exclusive sums end in the totals
__global__ void __launch_bounds__(THREADS) k_ploc_flags(const uint32_t* nn, uint32_t n, unsigned long long* flags) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flags[i] = 0ull;
    return;
  }
  const uint32_t j = nn[i];
  const bool pair = j != i && nn[j] == i;
  const unsigned long long keep = (pair && j < i) ? 0ull : 1ull, merge = (pair && i < j) ? 1ull : 0ull;
  flags[i] = (keep << 32) | merge;
}

__device__ __forceinline__ void link_counts(const Tree& t, uint32_t link, uint32_t& tris, uint32_t& quads, uint32_t& mask) {
  if (link & LEAF_BIT) {
    const uint8_t k = t.kind[t.item[link & ~LEAF_BIT]];
    tris = k == KIND_TRI ? 1u : 0u;
    quads = k == KIND_QUAD ? 1u : 0u;
    mask = kind_bit(k);
  } else {
    tris = t.n_tri[link];
    quads = t.n_quad[link];
    mask = t.kmask[link];
  }
}

__global__ void __launch_bounds__(THREADS) k_ploc_merge(Tree t, const uint32_t* cid, const float* cbox, const uint32_t* nn,
                                                        const unsigned long long* offset, uint32_t n, uint32_t first_node,
                                                        uint32_t* cid_out, float* cbox_out) {
  const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = nn[i];
  const bool pair = j != i && nn[j] == i;
  if (pair && j < i) return;  // lives on in its partner
  const uint32_t place = (uint32_t)(offset[i] >> 32);
  float b[6];
  for (int a = 0; a < 6; a++) b[a] = cbox[(size_t)i * 6 + a];
  uint32_t link = cid[i];
  if (pair) {
    const uint32_t node = first_node + (uint32_t)(offset[i] & 0xFFFFFFFFull), other = cid[j];
    const float* c = cbox + (size_t)j * 6;
    for (int a = 0; a < 3; a++) {
      b[a] = fminf(b[a], c[a]);
      b[3 + a] = fmaxf(b[3 + a], c[3 + a]);
    }
    uint32_t lt, lq, lm, rt_, rq, rm;
    link_counts(t, link, lt, lq, lm);
    link_counts(t, other, rt_, rq, rm);
    t.left[node] = link;
    t.right[node] = other;
    t.parent[node] = NO_NODE;
    t.kmask[node] = lm | rm;
    t.n_tri[node] = lt + rt_;
    t.n_quad[node] = lq + rq;
    for (int a = 0; a < 6; a++) t.ibox[(size_t)node * 6 + a] = b[a];
    if (link & LEAF_BIT) t.leaf_parent[link & ~LEAF_BIT] = node;
    else t.parent[link] = node;
    if (other & LEAF_BIT) t.leaf_parent[other & ~LEAF_BIT] = node;
    else t.parent[other] = node;
    link = node;
  }
  cid_out[place] = link;
  for (int a = 0; a < 6; a++) cbox_out[(size_t)place * 6 + a] = b[a];
}

// ---- step 3: depth-first positions.  What precedes a leaf or a subtree in its kind's primitive order is
// what hangs to the left of its path to the root.
__global__ void __launch_bounds__(THREADS) k_positions(Tree t, uint32_t* tri_order, uint32_t* quad_order) {
  const uint32_t x = blockIdx.x * THREADS + threadIdx.x;
  if (x >= 2u * t.n - 1u) return;
  const bool leaf = x < t.n;
  uint32_t link = leaf ? (LEAF_BIT | x) : x - t.n;
  uint32_t up = leaf ? t.leaf_parent[x] : t.parent[x - t.n];
  uint32_t tris = 0, quads = 0;
  while (up != NO_NODE) {
    if (t.right[up] == link) {
      uint32_t lt, lq, lm;
      link_counts(t, t.left[up], lt, lq, lm);
      tris += lt;
      quads += lq;
    }
    link = up;
    up = t.parent[up];
  }
  if (leaf) {
    const uint32_t id = t.item[x];
    const uint8_t k = t.kind[id];
    t.leaf_kpos[x] = k == KIND_TRI ? tris : quads;
    if (k == KIND_TRI) tri_order[tris] = t.index[id];
    else if (k == KIND_QUAD) quad_order[quads] = t.index[id];
  } else {
    t.tri_first[x - t.n] = tris;
    t.quad_first[x - t.n] = quads;
  }
}

// a binary link that ends the 4-wide hierarchy: a leaf, or a subtree of triangles that becomes one primitive run
__device__ __forceinline__ bool is_run_root(const Tree& t, uint32_t inner) {
  return t.kmask[inner] == 1u && t.n_tri[inner] <= t.max_run_tris;
}
__device__ __forceinline__ bool expandable(const Tree& t, uint32_t link) { return !(link & LEAF_BIT) && !is_run_root(t, link); }

__device__ __forceinline__ const float* link_box(const Tree& t, uint32_t link) {
  return (link & LEAF_BIT) ? t.box + (size_t)t.item[link & ~LEAF_BIT] * 6 : t.ibox + (size_t)link * 6;
}

__device__ uint32_t link_ref(const Tree& t, uint32_t link) {  // for links that are not expandable
  if (link & LEAF_BIT) {
    const uint32_t at = link & ~LEAF_BIT, id = t.item[at];
    const uint8_t k = t.kind[id];
    if (k == KIND_GATE) return (KIND_GATE << 30) | t.index[id];
    return ((uint32_t)k << 30) | t.leaf_kpos[at];  // a run of one
  }
  return (KIND_TRI << 30) | ((t.n_tri[link] - 1u) << 27) | t.tri_first[link];
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
  hipError_t run64(const unsigned long long* in, unsigned long long* out, size_t n) {
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
  // places to either side a cluster looks for its partner, and the longest triangle run (experiment knobs)
  const uint32_t radius = getenv("RT_AMD_PLOC_RADIUS") ? (uint32_t)std::max(1, atoi(getenv("RT_AMD_PLOC_RADIUS"))) : 12u;
  const uint32_t max_run = getenv("RT_AMD_DEVICE_TRI_RUN") ? (uint32_t)std::min(8, std::max(1, atoi(getenv("RT_AMD_DEVICE_TRI_RUN")))) : 4u;
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
  float *d_box = nullptr, *d_ibox = nullptr, *d_cbox[2] = {nullptr, nullptr};
  uint8_t* d_kind = nullptr;
  uint32_t *d_index = nullptr, *d_gate_alt = nullptr, *d_item = nullptr, *d_item2 = nullptr, *d_left = nullptr, *d_right = nullptr,
           *d_parent = nullptr, *d_leaf_parent = nullptr, *d_leaf_kpos = nullptr, *d_kmask = nullptr, *d_n_tri = nullptr, *d_n_quad = nullptr,
           *d_tri_first = nullptr, *d_quad_first = nullptr, *d_cid[2] = {nullptr, nullptr}, *d_nn = nullptr, *d_tri_order = nullptr,
           *d_quad_order = nullptr, *d_ninner = nullptr, *d_offset = nullptr;
  int32_t *d_gate_need = nullptr, *d_need = nullptr;
  unsigned long long *d_key = nullptr, *d_key2 = nullptr, *d_flags = nullptr, *d_sums = nullptr;
  void* d_sort_tmp = nullptr;
  BvhNode* d_nodes = nullptr;
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
    arena.take(&d_left, n);
    arena.take(&d_right, n);
    arena.take(&d_parent, n);
    arena.take(&d_leaf_parent, n);
    arena.take(&d_leaf_kpos, n);
    arena.take(&d_kmask, n);
    arena.take(&d_n_tri, n);
    arena.take(&d_n_quad, n);
    arena.take(&d_tri_first, n);
    arena.take(&d_quad_first, n);
    arena.take(&d_ibox, (size_t)n * 6);
    for (int k = 0; k < 2; k++) {
      arena.take(&d_cid[k], n);
      arena.take(&d_cbox[k], (size_t)n * 6);
    }
    arena.take(&d_nn, n);
    arena.take(&d_flags, (size_t)n + 1);
    arena.take(&d_sums, (size_t)n + 1);
    arena.take(&d_tri_order, n);
    arena.take(&d_quad_order, n);
    arena.take(&d_nodes, n);       // fewer 4-wide nodes than binary ones
    arena.take(&d_work, (size_t)n + 1);
    arena.take(&d_slots, n);
    arena.take(&d_ninner, (size_t)n + 1);
    arena.take(&d_offset, (size_t)n + 1);
    arena.take(&d_need, n);
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
    t.max_run_tris = max_run;
    t.item = d_item2; t.leaf_parent = d_leaf_parent; t.leaf_kpos = d_leaf_kpos;
    t.left = d_left; t.right = d_right; t.parent = d_parent; t.kmask = d_kmask;
    t.n_tri = d_n_tri; t.n_quad = d_n_quad; t.tri_first = d_tri_first; t.quad_first = d_quad_first;
    t.ibox = d_ibox;
    // ---- 2. the binary tree, bottom up
    {
      hipLaunchKernelGGL(k_ploc_init, grid(n), block, 0, 0, t, d_cid[0], d_cbox[0]);
      uint32_t clusters = n, made = 0;
      int cur = 0, rounds = 0;
      while (clusters > 1) {
        if (++rounds > 4096) {  // every round merges at least the closest pair
          rc = rt_fail_message(RT_ERR_DEVICE, "device build: the clustering does not end");
          goto done;
        }
        hipLaunchKernelGGL(k_ploc_nearest, grid(clusters), block, 0, 0, (const float*)d_cbox[cur], clusters, radius, d_nn);
        hipLaunchKernelGGL(k_ploc_flags, grid((size_t)clusters + 1), block, 0, 0, (const uint32_t*)d_nn, clusters, d_flags);
        BD_TRY(scan.run64(d_flags, d_sums, (size_t)clusters + 1));
        unsigned long long total = 0;
        BD_TRY(hipMemcpy(&total, d_sums + clusters, sizeof total, hipMemcpyDeviceToHost));
        const uint32_t kept = (uint32_t)(total >> 32), merged = (uint32_t)(total & 0xFFFFFFFFull);
        if (merged == 0 || kept + merged != clusters || made + merged > n - 1) {
          rc = rt_fail_message(RT_ERR_DEVICE, "device build: inconsistent clustering round");
          goto done;
        }
        hipLaunchKernelGGL(k_ploc_merge, grid(clusters), block, 0, 0, t, (const uint32_t*)d_cid[cur], (const float*)d_cbox[cur],
                           (const uint32_t*)d_nn, (const unsigned long long*)d_sums, clusters, made, d_cid[cur ^ 1], d_cbox[cur ^ 1]);
        made += merged;
        clusters = kept;
        cur ^= 1;
      }
      BD_TRY(hipGetLastError());
      if (made != n - 1) {
        rc = rt_fail_message(RT_ERR_DEVICE, "device build: the tree is not complete");
        goto done;
      }
      if (timing) fprintf(stderr, "[rt_amd]   device build: %d clustering rounds\n", rounds);
    }
    mark("clustering");
    // ---- 3. depth-first positions and the primitive orders
    hipLaunchKernelGGL(k_positions, grid(2 * (size_t)n - 1), block, 0, 0, t, d_tri_order, d_quad_order);
    BD_TRY(hipGetLastError());
    mark("positions");
    // ---- 4. the 4-wide nodes, level by level.  The root binary node always becomes node 0: a range of
    // two to four items of one kind could be a single run, but then there would be no node to hold its box.
    const Work root_work{n - 2u, 0u};  // the node made last
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
    // ---- 5. stack need, deepest level first
    for (size_t l = level_first.size(); l-- > 0;)
      hipLaunchKernelGGL(k_stack_need, grid(level_count[l]), block, 0, 0, t, (const Work*)(d_work + level_first[l]), level_count[l],
                         (const BvhNode*)d_nodes, d_need);
    BD_TRY(hipGetLastError());
    int32_t root_need = 0;
    BD_TRY(hipMemcpy(&root_need, d_need, 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals, d_n_tri + (n - 2u), 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals + 1, d_n_quad + (n - 2u), 4, hipMemcpyDeviceToHost));
    mark("stack need");
    // ---- results back to the host builder's vectors
    out->nodes.resize(n_nodes);
    BD_TRY(hipMemcpy(out->nodes.data(), d_nodes, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    out->tri_order.resize(totals[0]);
    out->quad_order.resize(totals[1]);
    if (totals[0]) BD_TRY(hipMemcpy(out->tri_order.data(), d_tri_order, (size_t)totals[0] * 4, hipMemcpyDeviceToHost));
    if (totals[1]) BD_TRY(hipMemcpy(out->quad_order.data(), d_quad_order, (size_t)totals[1] * 4, hipMemcpyDeviceToHost));
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
