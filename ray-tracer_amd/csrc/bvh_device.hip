// The acceleration structure built on the GPU: a binary tree from binned surface-area splits, made top
// down a level per round, collapsed into the 4-wide nodes the traversal walks (rt_types.h BvhNode).  Same
// splitting rule as the host builder (bvh_build.cpp), so a ray visits as many nodes through either
// hierarchy; a million triangles take 20 ms here against 180 ms on sixteen host cores.
//
//   1. the binary tree.  Every node owns a stretch of the item array.  Per level: sixteen bins per axis
//      across each node's box, filled by all items at once (integer atomics on an order-preserving encoding
//      of the floats; a workgroup whose 256 items share a node gathers in LDS first); one thread per node
//      picks the axis and bin boundary with the smallest area(left) * count(left) + area(right) * count(right);
//      a stable partition of every stretch through one prefix sum over the whole array.  A stretch of at
//      most four triangles becomes one primitive run (quadrics stay one per leaf, as in the host builder).
//      Nodes are numbered by prefix sums over the level, so the build is reproducible.
//   2. the items lie in depth-first order now: a prefix sum over their kinds gives every item's place in
//      its kind's primitive order and every node's counts; boxes bottom up over the levels
//   3. the 4-wide nodes, top down, one launch pair per level: a node starts from the two children of a
//      binary node and keeps replacing the child with the largest box by its two children until four
//      slots are used; the children that are binary nodes themselves become the next level's work
//   4. the traversal stack a ray can need, bottom up over the levels
//
// Two bottom-up builders came before this one and were dropped for the hierarchies they make: a radix tree
// over Morton keys (Karras 2012; a ray through a terrain of a million triangles visited 17.8 nodes and
// tested 15.8 triangles, against 11.1 and 9.8 here) and locally-ordered clustering (Meister and Bittner
// 2018: 16.2 and 11.1; built in 12.7 ms, traversed 1.75 times slower than this one).
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
constexpr uint32_t LEAF_BIT = 0x80000000u;  // in a binary child link: the rest is a place in the item array

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
  // the item array in its final order
  const uint32_t* item;     // item id at every place
  uint32_t* leaf_kpos;      // where the item lands in its kind's primitive order
  // binary internal nodes 0 .. n - 2, root 0
  uint32_t *left, *right, *kmask;
  uint32_t *n_tri, *n_quad;          // triangles / quadrics below
  uint32_t *tri_first, *quad_first;  // primitive-order position of the first triangle / quadric below
  float* ibox;
};

// ---- step 1: binned surface-area splits, one level of the binary tree per round.
// Every item knows the node it is in (node_of, by place in the item array); a node owns a stretch of
// that array.  Per level: sixteen bins per axis and node, across the node's box, with the count and box of
// the items whose centre falls in them; per node the split (axis, bin) with the smallest
// area(left) * count(left) + area(right) * count(right), the rule of the host builder (bvh_build.cpp,
// which lays its bins across the centres' bounds; across the box the children's ranges follow from the
// parent's bins and need no pass of their own); a stable partition of every node's stretch.  Floats are accumulated with integer
// atomic min / max on an order-preserving encoding, so the result does not depend on the order of arrival.
constexpr uint32_t INACTIVE = 0xFFFFFFFFu;
constexpr int SAH_BINS = 16;
constexpr int BIN_WORDS = 7;  // count, lo xyz, hi xyz
constexpr int NODE_BIN_WORDS = 3 * SAH_BINS * BIN_WORDS;
constexpr uint32_t SPLIT_RUN = 0xFFFFFFFEu;     // nsplit: the node is the root of one primitive run
constexpr uint32_t SPLIT_MEDIAN = 0xFFFFFFFDu;  // nsplit: the first nleft items go left
constexpr int SAH_MEDIAN_LEVEL = 40;            // from this level on the stretches are halved, so the depth stays bounded

__device__ __forceinline__ uint32_t ordered(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

struct Sah {
  uint32_t n;
  const float* box;
  const uint8_t* kind;
  uint32_t *nstart, *ncount, *nsplit, *nleft;  // by binary node
  float* nbox;                                 // by node: the range its bins are laid across, lo xyz hi xyz
  uint32_t* bins;                              // by node of the level: [axis][bin][BIN_WORDS]
  float* halves;                               // by node of the level: the boxes of the two sides of its split
  uint32_t level_first, level_count;           // the nodes being split: level_first .. level_first + level_count - 1
  uint32_t max_run_tris;
};

__device__ __forceinline__ void sah_new_node(const Sah& s, uint32_t node, uint32_t start, uint32_t count, uint32_t split, const float* range) {
  s.nstart[node] = start;
  s.ncount[node] = count;
  s.nsplit[node] = split;
  s.nleft[node] = 0;
  for (int a = 0; a < 6; a++) s.nbox[(size_t)node * 6 + a] = range[a];
}

struct SahRange {
  float v[6];
};
__global__ void __launch_bounds__(THREADS) k_sah_init(Sah s, SahRange all, uint32_t* item, uint32_t* node_of) {
  const uint32_t p = blockIdx.x * THREADS + threadIdx.x;
  if (p >= s.n) return;
  item[p] = p;
  node_of[p] = 0u;
  if (p == 0) sah_new_node(s, 0u, 0u, s.n, 0u, all.v);
}

__global__ void __launch_bounds__(THREADS) k_sah_clear_bins(uint32_t* bins, size_t words) {
  const size_t w = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (w >= words) return;
  const uint32_t k = (uint32_t)(w % BIN_WORDS);
  bins[w] = (k >= 1 && k <= 3) ? 0xFFFFFFFFu : 0u;
}

// which bin of `axis` a centre falls in; false when the node has no extent along that axis
__device__ __forceinline__ bool sah_bin(const Sah& s, uint32_t node, int axis, float centre, int& bin) {
  const float lo = s.nbox[(size_t)node * 6 + axis], hi = s.nbox[(size_t)node * 6 + 3 + axis];
  const float ext = hi - lo;
  if (!(ext > 0.0f)) return false;
  const float scale = (float)SAH_BINS / ext;
  bin = min(SAH_BINS - 1, max(0, (int)((centre - lo) * scale)));
  return true;
}

__global__ void __launch_bounds__(THREADS) k_sah_bins(Sah s, const uint32_t* item, const uint32_t* node_of) {
  __shared__ uint32_t acc[NODE_BIN_WORDS];
  __shared__ uint32_t first;
  const uint32_t p = blockIdx.x * THREADS + threadIdx.x;
  const uint32_t node = p < s.n ? node_of[p] : INACTIVE;
  if (threadIdx.x == 0) first = node;
  for (int w = threadIdx.x; w < NODE_BIN_WORDS; w += THREADS) {
    const int k = w % BIN_WORDS;
    acc[w] = (k >= 1 && k <= 3) ? 0xFFFFFFFFu : 0u;
  }
  __syncthreads();
  const bool uniform = __syncthreads_and(node == first) != 0;
  if (uniform && first == INACTIVE) return;
  if (node != INACTIVE) {
    const float* b = s.box + (size_t)item[p] * 6;
    uint32_t lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      lo[a] = ordered(b[a]);
      hi[a] = ordered(b[3 + a]);
    }
    uint32_t* base = uniform ? acc : s.bins + (size_t)(node - s.level_first) * NODE_BIN_WORDS;
    for (int a = 0; a < 3; a++) {
      int bin;
      if (!sah_bin(s, node, a, 0.5f * (b[a] + b[3 + a]), bin)) continue;
      uint32_t* q = base + (a * SAH_BINS + bin) * BIN_WORDS;
      atomicAdd(q, 1u);
      for (int k = 0; k < 3; k++) {  // a plain look first: most items do not move their bin's box
        if (lo[k] < *(volatile uint32_t*)(q + 1 + k)) atomicMin(q + 1 + k, lo[k]);
        if (hi[k] > *(volatile uint32_t*)(q + 4 + k)) atomicMax(q + 4 + k, hi[k]);
      }
    }
  }
  if (!uniform) return;
  __syncthreads();
  uint32_t* out = s.bins + (size_t)(first - s.level_first) * NODE_BIN_WORDS;
  for (int w = threadIdx.x; w < NODE_BIN_WORDS; w += THREADS) {
    const int k = w % BIN_WORDS;
    const uint32_t v = acc[w];
    if (k == 0) {
      if (v) atomicAdd(out + w, v);
    } else if (k <= 3) {
      if (v != 0xFFFFFFFFu) atomicMin(out + w, v);
    } else if (v != 0u) {
      atomicMax(out + w, v);
    }
  }
}

struct SahBox {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  __device__ void grow(const uint32_t* bin) {  // a bin's lo xyz, hi xyz
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], unordered(bin[1 + a]));
      hi[a] = fmaxf(hi[a], unordered(bin[4 + a]));
    }
  }
  __device__ float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

// The split of every node of the level, and how many nodes it adds: in the low word the children that are
// split in the next round, in the high word the nodes below the root of a primitive run.
__global__ void __launch_bounds__(THREADS) k_sah_split(Sah s, const uint32_t* item, int median, unsigned long long* adds) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w > s.level_count) return;
  if (w == s.level_count) {  // one entry past the end so that the exclusive sums end in the totals
    adds[w] = 0ull;
    return;
  }
  const uint32_t node = s.level_first + w, start = s.nstart[node], count = s.ncount[node];
  bool run = count <= s.max_run_tris;  // triangles that fit one run are not split by cost (as in the host builder)
  for (uint32_t i = 0; run && i < count; i++) run = s.kind[item[start + i]] == KIND_TRI;
  if (run) {
    s.nsplit[node] = SPLIT_RUN;
    adds[w] = (unsigned long long)(count - 2u) << 32;
    return;
  }
  float best_cost = INFINITY;
  int best_axis = -1, best_bin = -1;
  uint32_t best_left = 0;
  const uint32_t* bins = s.bins + (size_t)w * NODE_BIN_WORDS;
  for (int a = 0; a < 3 && !median; a++) {
    const uint32_t* row = bins + a * SAH_BINS * BIN_WORDS;
    float right_area[SAH_BINS];
    uint32_t right_count[SAH_BINS];
    SahBox acc;
    uint32_t c = 0;
    for (int b = SAH_BINS - 1; b > 0; b--) {
      if (row[b * BIN_WORDS]) acc.grow(row + b * BIN_WORDS);
      c += row[b * BIN_WORDS];
      right_area[b] = acc.half_area();
      right_count[b] = c;
    }
    SahBox left;
    uint32_t lc = 0;
    for (int b = 0; b < SAH_BINS - 1; b++) {
      if (row[b * BIN_WORDS]) left.grow(row + b * BIN_WORDS);
      lc += row[b * BIN_WORDS];
      if (lc == 0 || right_count[b + 1] == 0) continue;
      const float cost = left.half_area() * (float)lc + right_area[b + 1] * (float)right_count[b + 1];
      if (cost < best_cost) {
        best_cost = cost;
        best_axis = a;
        best_bin = b;
        best_left = lc;
      }
    }
  }
  uint32_t nl;
  float* halves = s.halves + (size_t)w * 12;
  if (best_axis < 0) {  // the centres coincide, or the level is past the depth bound: both halves keep the node's range
    s.nsplit[node] = SPLIT_MEDIAN;
    nl = count / 2u;
    for (int a = 0; a < 6; a++) halves[a] = halves[6 + a] = s.nbox[(size_t)node * 6 + a];
  } else {
    s.nsplit[node] = ((uint32_t)best_axis << 8) | (uint32_t)best_bin;
    nl = best_left;
    const uint32_t* row = bins + best_axis * SAH_BINS * BIN_WORDS;
    SahBox side[2];
    for (int b = 0; b < SAH_BINS; b++)
      if (row[b * BIN_WORDS]) side[b <= best_bin ? 0 : 1].grow(row + b * BIN_WORDS);
    for (int h = 0; h < 2; h++)
      for (int a = 0; a < 3; a++) {
        halves[h * 6 + a] = side[h].lo[a];
        halves[h * 6 + 3 + a] = side[h].hi[a];
      }
  }
  s.nleft[node] = nl;
  adds[w] = (nl >= 2u ? 1ull : 0ull) + (count - nl >= 2u ? 1ull : 0ull);
}

// Writes the links of the level's nodes and sets up their children.  New nodes that are split in the next
// round are numbered upward from next_first in the order of their parents; the nodes inside primitive runs
// are numbered downward from chain_top, so the nodes of one level stay one range.
__global__ void __launch_bounds__(THREADS) k_sah_emit(Sah s, Tree t, const uint32_t* item, const unsigned long long* offset,
                                                      uint32_t next_first, uint32_t chain_top) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w >= s.level_count) return;
  const uint32_t node = s.level_first + w, start = s.nstart[node], count = s.ncount[node];
  const uint32_t split = s.nsplit[node];
  const uint32_t off_split = (uint32_t)(offset[w] & 0xFFFFFFFFull), off_chain = (uint32_t)(offset[w] >> 32);
  if (split == SPLIT_RUN) {
    // node -> (item, next) -> (item, next) ... -> (item, item); boxes from the last item backwards
    uint32_t ids[8];
    ids[0] = node;
    for (uint32_t i = 1; i + 1 < count; i++) {
      ids[i] = chain_top - off_chain - (i - 1u);
      sah_new_node(s, ids[i], start + i, count - i, SPLIT_RUN, s.nbox + (size_t)node * 6);
    }
    for (uint32_t i = 0; i + 1 < count; i++) {
      const uint32_t cur = ids[i];
      t.left[cur] = LEAF_BIT | (start + i);
      t.right[cur] = i + 2 == count ? (LEAF_BIT | (start + i + 1u)) : ids[i + 1];
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = count; i-- > 0;) {
      const float* b = t.box + (size_t)item[start + i] * 6;
      for (int a = 0; a < 3; a++) {
        lo[a] = fminf(lo[a], b[a]);
        hi[a] = fmaxf(hi[a], b[3 + a]);
      }
      if (i + 1 < count)
        for (int a = 0; a < 3; a++) {
          t.ibox[(size_t)ids[i] * 6 + a] = lo[a];
          t.ibox[(size_t)ids[i] * 6 + 3 + a] = hi[a];
        }
    }
    return;
  }
  const uint32_t nl = s.nleft[node], nr = count - nl;
  uint32_t k = 0;
  auto child = [&](uint32_t first, uint32_t cnt, const float* range) -> uint32_t {
    if (cnt == 1u) return LEAF_BIT | first;
    const uint32_t id = next_first + off_split + k++;
    sah_new_node(s, id, first, cnt, 0u, range);
    return id;
  };
  t.left[node] = child(start, nl, s.halves + (size_t)w * 12);
  t.right[node] = child(start + nl, nr, s.halves + (size_t)w * 12 + 6);
}

// 1 for the items that go to the left child of their node
__global__ void __launch_bounds__(THREADS) k_sah_sides(Sah s, const uint32_t* item, const uint32_t* node_of, uint32_t* side) {
  const uint32_t p = blockIdx.x * THREADS + threadIdx.x;
  if (p > s.n) return;
  uint32_t left = 0;
  const uint32_t node = p < s.n ? node_of[p] : INACTIVE;
  if (node != INACTIVE) {
    const uint32_t split = s.nsplit[node];
    if (split == SPLIT_MEDIAN) {
      left = p - s.nstart[node] < s.nleft[node] ? 1u : 0u;
    } else if (split != SPLIT_RUN) {
      const int axis = (int)(split >> 8);
      const float* b = s.box + (size_t)item[p] * 6;
      int bin = 0;
      (void)sah_bin(s, node, axis, 0.5f * (b[axis] + b[3 + axis]), bin);
      left = bin <= (int)(split & 0xFFu) ? 1u : 0u;
    }
  }
  side[p] = left;
}

// the stable partition: sums are the exclusive sums of side
__global__ void __launch_bounds__(THREADS) k_sah_scatter(Sah s, Tree t, const uint32_t* item, const uint32_t* node_of, const uint32_t* sums,
                                                         uint32_t* item_out, uint32_t* node_of_out) {
  const uint32_t p = blockIdx.x * THREADS + threadIdx.x;
  if (p >= s.n) return;
  const uint32_t node = node_of[p];
  if (node == INACTIVE || s.nsplit[node] == SPLIT_RUN) {
    item_out[p] = item[p];
    node_of_out[p] = INACTIVE;
    return;
  }
  const uint32_t start = s.nstart[node], before = sums[p] - sums[start];  // items of the node ahead of p that go left
  const bool left = sums[p + 1] != sums[p];
  const uint32_t to = left ? start + before : start + s.nleft[node] + (p - start - before);
  const uint32_t link = left ? t.left[node] : t.right[node];
  item_out[to] = item[p];
  node_of_out[to] = (link & LEAF_BIT) ? INACTIVE : link;
}

// With the items in their final places: a triangle / quadric count ahead of every place (high / low word) ...
__global__ void __launch_bounds__(THREADS) k_sah_kinds(Tree t, unsigned long long* flags) {
  const uint32_t p = blockIdx.x * THREADS + threadIdx.x;
  if (p > t.n) return;
  unsigned long long f = 0ull;
  if (p < t.n) {
    const uint8_t k = t.kind[t.item[p]];
    f = k == KIND_TRI ? (1ull << 32) : (k == KIND_QUAD ? 1ull : 0ull);
  }
  flags[p] = f;
}
// ... from which follow every item's place in its kind's primitive order and, per node, its counts by kind
// and where its triangles and quadrics begin in those orders
__global__ void __launch_bounds__(THREADS) k_sah_orders(Tree t, Sah s, const unsigned long long* ahead, uint32_t* tri_order, uint32_t* quad_order) {
  const uint32_t x = blockIdx.x * THREADS + threadIdx.x;
  if (x < t.n) {
    const uint32_t id = t.item[x];
    const uint8_t k = t.kind[id];
    const uint32_t tris = (uint32_t)(ahead[x] >> 32), quads = (uint32_t)(ahead[x] & 0xFFFFFFFFull);
    t.leaf_kpos[x] = k == KIND_TRI ? tris : quads;
    if (k == KIND_TRI) tri_order[tris] = t.index[id];
    else if (k == KIND_QUAD) quad_order[quads] = t.index[id];
  }
  if (x + 1 < t.n) {  // binary node x
    const uint32_t start = s.nstart[x], count = s.ncount[x];
    const unsigned long long a = ahead[start], b = ahead[start + count];
    const uint32_t tris = (uint32_t)(b >> 32) - (uint32_t)(a >> 32), quads = (uint32_t)(b & 0xFFFFFFFFull) - (uint32_t)(a & 0xFFFFFFFFull);
    t.n_tri[x] = tris;
    t.n_quad[x] = quads;
    t.kmask[x] = (tris ? 1u : 0u) | (quads ? 2u : 0u) | (count - tris - quads ? 4u : 0u);
    t.tri_first[x] = (uint32_t)(a >> 32);
    t.quad_first[x] = (uint32_t)(a & 0xFFFFFFFFull);
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

// boxes of the split nodes of one level of the top-down build, from their children's (deeper levels done)
__global__ void __launch_bounds__(THREADS) k_sah_boxes(Tree t, Sah s) {
  const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
  if (w >= s.level_count) return;
  const uint32_t node = s.level_first + w;
  if (s.nsplit[node] == SPLIT_RUN) return;  // made with the run
  const float *l = link_box(t, t.left[node]), *r = link_box(t, t.right[node]);
  for (int a = 0; a < 3; a++) {
    t.ibox[(size_t)node * 6 + a] = fminf(l[a], r[a]);
    t.ibox[(size_t)node * 6 + 3 + a] = fmaxf(l[3 + a], r[3 + a]);
  }
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
  // the longest triangle run (experiment knob)
  const uint32_t max_run = getenv("RT_AMD_DEVICE_TRI_RUN") ? (uint32_t)std::min(8, std::max(1, atoi(getenv("RT_AMD_DEVICE_TRI_RUN")))) : 4u;
  auto mark = [&, last = t0](const char* what) mutable {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[rt_amd]   device build: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  };
  // ---- host: boxes rounded outward to f32, and the box of them all
  // plain arrays, not vectors: nothing fills them before the slices below do (a vector would zero 30 MB first)
  std::unique_ptr<float[]> h_box(new float[(size_t)n * 6]);
  std::unique_ptr<uint8_t[]> h_kind(new uint8_t[n]);
  std::unique_ptr<uint32_t[]> h_index(new uint32_t[n]);
  float all_box[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};  // of the items' boxes
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
          acc[a] = std::fmin(acc[a], l);
          acc[3 + a] = std::fmax(acc[3 + a], h);
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
        all_box[a] = std::fmin(all_box[a], acc[a]);
        all_box[3 + a] = std::fmax(all_box[3 + a], acc[3 + a]);
      }
  }
  std::vector<int32_t> h_need(gate_need.begin(), gate_need.end());
  mark("boxes on the host");
  float *d_box = nullptr, *d_ibox = nullptr, *d_nbox = nullptr;
  uint8_t* d_kind = nullptr;
  uint32_t *d_index = nullptr, *d_gate_alt = nullptr, *d_item = nullptr, *d_item2 = nullptr, *d_left = nullptr, *d_right = nullptr,
           *d_leaf_kpos = nullptr, *d_kmask = nullptr, *d_n_tri = nullptr, *d_n_quad = nullptr,
           *d_tri_first = nullptr, *d_quad_first = nullptr, *d_tri_order = nullptr,
           *d_quad_order = nullptr, *d_ninner = nullptr, *d_offset = nullptr;
  uint32_t *d_node_of[2] = {nullptr, nullptr}, *d_nstart = nullptr, *d_ncount = nullptr, *d_nsplit = nullptr, *d_nleft = nullptr,
           *d_bins = nullptr;
  size_t bins_nodes = 0;  // nodes d_bins has room for
  int32_t *d_gate_need = nullptr, *d_need = nullptr;
  unsigned long long *d_flags = nullptr, *d_sums = nullptr;
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
    arena.take(&d_item, n);
    arena.take(&d_item2, n);
    arena.take(&d_left, n);
    arena.take(&d_right, n);
    arena.take(&d_leaf_kpos, n);
    arena.take(&d_kmask, n);
    arena.take(&d_n_tri, n);
    arena.take(&d_n_quad, n);
    arena.take(&d_tri_first, n);
    arena.take(&d_quad_first, n);
    arena.take(&d_ibox, (size_t)n * 6);
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
    for (int k = 0; k < 2; k++) arena.take(&d_node_of[k], n);
    arena.take(&d_nstart, n);
    arena.take(&d_ncount, n);
    arena.take(&d_nsplit, n);
    arena.take(&d_nleft, n);
    arena.take(&d_nbox, (size_t)n * 6);
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
    Tree t{};
    t.box = d_box; t.kind = d_kind; t.index = d_index;
    t.gate_alt = d_gate_alt; t.gate_need = d_gate_need;
    t.n_gate_alt = (uint32_t)gate_alt.size(); t.n_gate_need = (uint32_t)h_need.size();
    t.n = n;
    t.max_run_tris = max_run;
    t.item = d_item; t.leaf_kpos = d_leaf_kpos;
    t.left = d_left; t.right = d_right; t.kmask = d_kmask;
    t.n_tri = d_n_tri; t.n_quad = d_n_quad; t.tri_first = d_tri_first; t.quad_first = d_quad_first;
    t.ibox = d_ibox;
    {
      // ---- 1. the binary tree, top down, a level per round
      Sah sah{};
      sah.n = n; sah.box = d_box; sah.kind = d_kind;
      sah.nstart = d_nstart; sah.ncount = d_ncount; sah.nsplit = d_nsplit; sah.nleft = d_nleft;
      sah.nbox = d_nbox;
      sah.max_run_tris = max_run;
      uint32_t* items[2] = {d_item, d_item2};
      int cur = 0;
      SahRange all;
      for (int a = 0; a < 6; a++) all.v[a] = all_box[a];
      hipLaunchKernelGGL(k_sah_init, grid(n), block, 0, 0, sah, all, items[0], d_node_of[0]);
      std::vector<std::pair<uint32_t, uint32_t>> rounds;  // first node and node count of every level
      uint32_t lvl_first = 0, lvl_count = 1, next_id = 1, chain_top = n - 2u;
      while (lvl_count > 0) {
        if (rounds.size() > 160) {
          rc = rt_fail_message(RT_ERR_DEVICE, "device build: the splitting does not end");
          goto done;
        }
        if (lvl_count > bins_nodes) {  // room for the level's bins; levels grow by at most a factor of two
          if (d_bins) (void)hipFree(d_bins);
          d_bins = nullptr;
          bins_nodes = std::min<size_t>((size_t)n / 2 + 1, std::max<size_t>(std::max<size_t>(4096, n / 64), (size_t)lvl_count * 4));
          BD_TRY(hipMalloc((void**)&d_bins, bins_nodes * (NODE_BIN_WORDS + 12) * sizeof(uint32_t)));
        }
        sah.bins = d_bins;
        sah.halves = reinterpret_cast<float*>(d_bins + bins_nodes * NODE_BIN_WORDS);
        sah.level_first = lvl_first;
        sah.level_count = lvl_count;
        const size_t bin_words = (size_t)lvl_count * NODE_BIN_WORDS;
        hipLaunchKernelGGL(k_sah_clear_bins, grid(bin_words), block, 0, 0, d_bins, bin_words);
        hipLaunchKernelGGL(k_sah_bins, grid(n), block, 0, 0, sah, (const uint32_t*)items[cur], (const uint32_t*)d_node_of[cur]);
        hipLaunchKernelGGL(k_sah_split, grid((size_t)lvl_count + 1), block, 0, 0, sah, (const uint32_t*)items[cur],
                           rounds.size() >= (size_t)SAH_MEDIAN_LEVEL ? 1 : 0, d_flags);
        BD_TRY(scan.run64(d_flags, d_sums, (size_t)lvl_count + 1));
        unsigned long long total = 0;
        BD_TRY(hipMemcpy(&total, d_sums + lvl_count, sizeof total, hipMemcpyDeviceToHost));
        const uint32_t split_nodes = (uint32_t)(total & 0xFFFFFFFFull), chain_nodes = (uint32_t)(total >> 32);
        if ((uint64_t)next_id + split_nodes + chain_nodes > (uint64_t)chain_top + 1u) {
          rc = rt_fail_message(RT_ERR_DEVICE, "device build: more binary nodes than a tree over the items has");
          goto done;
        }
        hipLaunchKernelGGL(k_sah_emit, grid(lvl_count), block, 0, 0, sah, t, (const uint32_t*)items[cur],
                           (const unsigned long long*)d_sums, next_id, chain_top);
        hipLaunchKernelGGL(k_sah_sides, grid((size_t)n + 1), block, 0, 0, sah, (const uint32_t*)items[cur], (const uint32_t*)d_node_of[cur], d_ninner);
        BD_TRY(scan.run(d_ninner, d_offset, (size_t)n + 1));
        hipLaunchKernelGGL(k_sah_scatter, grid(n), block, 0, 0, sah, t, (const uint32_t*)items[cur], (const uint32_t*)d_node_of[cur],
                           (const uint32_t*)d_offset, items[cur ^ 1], d_node_of[cur ^ 1]);
        cur ^= 1;
        rounds.emplace_back(lvl_first, lvl_count);
        lvl_first = next_id;
        lvl_count = split_nodes;
        next_id += split_nodes;
        chain_top -= chain_nodes;
      }
      BD_TRY(hipGetLastError());
      if (next_id != chain_top + 1u) {
        rc = rt_fail_message(RT_ERR_DEVICE, "device build: the tree is not complete");
        goto done;
      }
      if (timing) fprintf(stderr, "[rt_amd]   device build: %zu levels of splits\n", rounds.size());
      mark("surface-area splits");
      // ---- 2. primitive orders, counts and boxes: the items lie in depth-first order already
      t.item = items[cur];
      hipLaunchKernelGGL(k_sah_kinds, grid((size_t)n + 1), block, 0, 0, t, d_flags);
      BD_TRY(scan.run64(d_flags, d_sums, (size_t)n + 1));
      hipLaunchKernelGGL(k_sah_orders, grid(n), block, 0, 0, t, sah, (const unsigned long long*)d_sums, d_tri_order, d_quad_order);
      for (size_t l = rounds.size(); l-- > 0;) {
        sah.level_first = rounds[l].first;
        sah.level_count = rounds[l].second;
        hipLaunchKernelGGL(k_sah_boxes, grid(rounds[l].second), block, 0, 0, t, sah);
      }
      BD_TRY(hipGetLastError());
      mark("orders, boxes");
    }
    // ---- 3. the 4-wide nodes, level by level.  The root binary node (0) always becomes node 0: a range of
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
    // ---- 4. stack need, deepest level first
    for (size_t l = level_first.size(); l-- > 0;)
      hipLaunchKernelGGL(k_stack_need, grid(level_count[l]), block, 0, 0, t, (const Work*)(d_work + level_first[l]), level_count[l],
                         (const BvhNode*)d_nodes, d_need);
    BD_TRY(hipGetLastError());
    int32_t root_need = 0;
    BD_TRY(hipMemcpy(&root_need, d_need, 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals, d_n_tri, 4, hipMemcpyDeviceToHost));
    BD_TRY(hipMemcpy(totals + 1, d_n_quad, 4, hipMemcpyDeviceToHost));
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
  if (d_bins) (void)hipFree(d_bins);
  out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

}  // namespace rt
