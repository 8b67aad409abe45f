// Binned-SAH BVH over f64 boxes.  The reference tracer scans every child of a
// group linearly (src/shape.rs:1213-1219); any conservative hierarchy over the
// same leaves returns the same closest hit, so this structure is free to be
// whatever traverses fastest.
#include "bvh_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <limits>
#include <thread>

namespace rt {
namespace {

constexpr int BINS = 16;
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
const int MAX_TRI_LEAF = env_int("RT_AMD_TRI_LEAF", 4);     // experiment knobs
// One quadric per leaf: with boxes that fit the round shapes, a second primitive in a run is tested far
// more often than it is hit (100k primitives: 4.2 tests per ray with runs of up to 4, 2.4 with single
// primitives, 11 % less frame time for 8 % more node visits).
const int MAX_QUAD_LEAF = env_int("RT_AMD_QUAD_LEAF", 1);
constexpr int SOFT_DEPTH = 36;  // beyond this, split at the median so depth stays bounded

const double INF = std::numeric_limits<double>::infinity();

struct Box {
  double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
  void grow(const double* l, const double* h) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], l[a]);
      hi[a] = std::max(hi[a], h[a]);
    }
  }
  void grow_point(const double* p) { grow(p, p); }
  double half_area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0;
    return dx * dy + dy * dz + dz * dx;
  }
};

inline void centroid(const BuildItem& it, double c[3]) {
  for (int a = 0; a < 3; a++) c[a] = 0.5 * (it.lo[a] + it.hi[a]);
}

double env_double(const char* name, double dflt) {
  const char* v = getenv(name);
  return v && *v ? atof(v) : dflt;
}
// Cost of visiting one more node relative to one primitive test; < 0 turns the cost test off
// (every range that fits a run becomes a leaf).
const double SAH_NODE_TRI = env_double("RT_AMD_SAH_NODE_TRI", -1.0);
const double SAH_NODE_QUAD = env_double("RT_AMD_SAH_NODE_QUAD", 1.5);

// Can the range be one primitive run at all?
bool leaf_ok(const BuildItem* items, size_t n) {
  if (n == 1) return true;
  uint32_t k = items[0].kind;
  if (k == KIND_GATE) return false;
  for (size_t i = 1; i < n; i++)
    if (items[i].kind != k) return false;
  return n <= (size_t)(k == KIND_TRI ? MAX_TRI_LEAF : MAX_QUAD_LEAF);
}

}  // namespace

uint32_t BvhBuilder::make_leaf(BuildItem* items, size_t n) {
  uint32_t k = items[0].kind;
  if (k == KIND_GATE) return make_ref(KIND_GATE, items[0].index);
  std::vector<uint32_t>& order = k == KIND_TRI ? tri_order : quad_order;
  uint32_t first = (uint32_t)order.size();
  for (size_t i = 0; i < n; i++) order.push_back(items[i].index);
  return make_run(k, first, (uint32_t)n);
}

// Appends a finished sub-builder's binary nodes and primitive orders to this builder and returns `ref`
// (a reference inside the sub-builder) as a reference inside this one.
uint32_t BvhBuilder::splice(const BvhBuilder& sub, uint32_t ref) {
  const uint32_t bin_off = (uint32_t)bin_.size(), tri_off = (uint32_t)tri_order.size(),
                 quad_off = (uint32_t)quad_order.size();
  auto moved = [&](uint32_t r) -> uint32_t {
    if (r == REF_NONE) return r;
    switch (r >> 30) {
      case KIND_INNER: return r + bin_off;
      case KIND_TRI: return r + tri_off;    // the run's first slot sits in the low bits
      case KIND_QUAD: return r + quad_off;
      default: return r;                    // gates are numbered by the caller
    }
  };
  for (BinNode nd : sub.bin_) {
    nd.c[0] = moved(nd.c[0]);
    nd.c[1] = moved(nd.c[1]);
    bin_.push_back(nd);
  }
  tri_order.insert(tri_order.end(), sub.tri_order.begin(), sub.tri_order.end());
  quad_order.insert(quad_order.end(), sub.quad_order.begin(), sub.quad_order.end());
  return moved(ref);
}

constexpr size_t PARALLEL_MIN_ITEMS = 16384;  // below this a range is not worth a thread

uint32_t BvhBuilder::build_range(BuildItem* items, size_t n, int depth, double lo[3], double hi[3], int threads) {
  Box box, cbox;
  for (size_t i = 0; i < n; i++) {
    box.grow(items[i].lo, items[i].hi);
    double c[3];
    centroid(items[i], c);
    cbox.grow_point(c);
  }
  for (int a = 0; a < 3; a++) {
    lo[a] = box.lo[a];
    hi[a] = box.hi[a];
  }
  const bool may_leaf = leaf_ok(items, n);
  const double node_cost = may_leaf && n > 1 ? (items[0].kind == KIND_TRI ? SAH_NODE_TRI : SAH_NODE_QUAD) : -1.0;
  if (may_leaf && (n == 1 || node_cost < 0.0)) return make_leaf(items, n);

  size_t mid = 0;
  bool split = false;
  if (depth < SOFT_DEPTH) {
    double best_cost = INF;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; a++) {
      double ext = cbox.hi[a] - cbox.lo[a];
      if (!(ext > 0)) continue;
      Box bb[BINS];
      size_t cnt[BINS] = {0};
      double scale = BINS / ext;
      for (size_t i = 0; i < n; i++) {
        double c = 0.5 * (items[i].lo[a] + items[i].hi[a]);
        int b = std::min(BINS - 1, std::max(0, (int)((c - cbox.lo[a]) * scale)));
        bb[b].grow(items[i].lo, items[i].hi);
        cnt[b]++;
      }
      double right_area[BINS];
      size_t right_cnt[BINS];
      Box acc;
      size_t c = 0;
      for (int b = BINS - 1; b > 0; b--) {
        acc.grow(bb[b].lo, bb[b].hi);
        c += cnt[b];
        right_area[b] = acc.half_area();
        right_cnt[b] = c;
      }
      Box left;
      size_t lc = 0;
      for (int b = 0; b < BINS - 1; b++) {
        left.grow(bb[b].lo, bb[b].hi);
        lc += cnt[b];
        if (lc == 0 || right_cnt[b + 1] == 0) continue;
        double cost = left.half_area() * (double)lc + right_area[b + 1] * (double)right_cnt[b + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = a;
          best_bin = b;
        }
      }
    }
    // surface-area heuristic: splitting pays when the expected tests of the two halves plus the
    // extra node cost less than testing all n primitives
    if (may_leaf && !(best_cost + node_cost * box.half_area() < (double)n * box.half_area())) return make_leaf(items, n);
    if (best_axis >= 0) {
      int a = best_axis;
      double scale = BINS / (cbox.hi[a] - cbox.lo[a]);
      double clo = cbox.lo[a];
      BuildItem* m = std::partition(items, items + n, [&](const BuildItem& it) {
        double c = 0.5 * (it.lo[a] + it.hi[a]);
        int b = std::min(BINS - 1, std::max(0, (int)((c - clo) * scale)));
        return b <= best_bin;
      });
      mid = (size_t)(m - items);
      split = mid > 0 && mid < n;
    }
  }
  if (!split && may_leaf) return make_leaf(items, n);
  if (!split) {
    int a = 0;
    double ext = -1;
    for (int k = 0; k < 3; k++)
      if (cbox.hi[k] - cbox.lo[k] > ext) {
        ext = cbox.hi[k] - cbox.lo[k];
        a = k;
      }
    mid = n / 2;
    std::nth_element(items, items + mid, items + n, [a](const BuildItem& x, const BuildItem& y) {
      return x.lo[a] + x.hi[a] < y.lo[a] + y.hi[a];
    });
  }

  uint32_t idx = (uint32_t)bin_.size();
  bin_.emplace_back();
  double l0[3], h0[3], l1[3], h1[3];
  uint32_t c0, c1;
  if (threads > 1 && n >= PARALLEL_MIN_ITEMS) {
    BvhBuilder left, right;
    const int left_threads = std::max(1, (int)((double)threads * (double)mid / (double)n + 0.5));
    const int right_threads = std::max(1, threads - left_threads);
    auto other = std::async(std::launch::async, [&] {
      return left.build_range(items, mid, depth + 1, l0, h0, left_threads);
    });
    uint32_t r1 = right.build_range(items + mid, n - mid, depth + 1, l1, h1, right_threads);
    uint32_t r0 = other.get();
    c0 = splice(left, r0);  // left first: the order of a single-threaded build
    c1 = splice(right, r1);
  } else {
    c0 = build_range(items, mid, depth + 1, l0, h0, 1);
    c1 = build_range(items + mid, n - mid, depth + 1, l1, h1, 1);
  }
  BinNode& nd = bin_[idx];
  for (int a = 0; a < 3; a++) {
    nd.lo[0][a] = l0[a]; nd.hi[0][a] = h0[a];
    nd.lo[1][a] = l1[a]; nd.hi[1][a] = h1[a];
  }
  nd.c[0] = c0;
  nd.c[1] = c1;
  return make_ref(KIND_INNER, idx);
}

namespace {
float round_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}
double half_area3(const double* lo, const double* hi) {
  double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
}  // namespace

// Collapses the binary tree under bin_ref into 4-wide nodes: start from the two children and keep
// replacing the inner child with the largest box by its own two children until four slots are used.
//
// *need receives the number of stack entries a traversal of the subtree can hold at once: a node
// keeps at most (children - 1) siblings waiting while one child's subtree is walked, a gate hands
// over to the subtree behind it, a primitive run needs nothing.
uint32_t BvhBuilder::collapse(uint32_t bin_ref, int* need) {
  *need = 0;
  if (bin_ref == REF_NONE) return bin_ref;
  if ((bin_ref >> 30) == KIND_GATE) {
    uint32_t g = bin_ref & 0x3FFFFFFFu;
    *need = g < gate_need.size() ? gate_need[g] : 0;
    return bin_ref;
  }
  if ((bin_ref >> 30) != KIND_INNER) return bin_ref;
  struct Slot {
    double lo[3], hi[3];
    uint32_t ref;
  };
  Slot slots[4];
  int n = 0;
  auto add = [&](const BinNode& b, int k) {
    for (int a = 0; a < 3; a++) {
      slots[n].lo[a] = b.lo[k][a];
      slots[n].hi[a] = b.hi[k][a];
    }
    slots[n].ref = b.c[k];
    n++;
  };
  {
    const BinNode& b = bin_[bin_ref & 0x3FFFFFFFu];
    add(b, 0);
    add(b, 1);
  }
  while (n < 4) {
    int best = -1;
    double best_area = -1;
    for (int i = 0; i < n; i++)
      if ((slots[i].ref >> 30) == KIND_INNER) {
        double ar = half_area3(slots[i].lo, slots[i].hi);
        if (ar > best_area) {
          best_area = ar;
          best = i;
        }
      }
    if (best < 0) break;
    const BinNode b = bin_[slots[best].ref & 0x3FFFFFFFu];
    slots[best] = slots[n - 1];
    n--;
    add(b, 0);
    add(b, 1);
  }
  uint32_t idx = (uint32_t)nodes.size();
  nodes.emplace_back();
  uint32_t refs[4];
  int deepest = 0;
  for (int i = 0; i < n; i++) {
    int child_need = 0;
    refs[i] = collapse(slots[i].ref, &child_need);
    deepest = std::max(deepest, child_need);
  }
  *need = (n - 1) + deepest;
  BvhNode& out = nodes[idx];
  for (int c = 0; c < 4; c++) {
    for (int a = 0; a < 3; a++) {
      out.plane[a][0][c] = c < n ? round_down(slots[c].lo[a]) : INFINITY;
      out.plane[a][1][c] = c < n ? round_up(slots[c].hi[a]) : -INFINITY;
    }
    out.child[c] = c < n ? refs[c] : REF_NONE;
    out.alt[c] = c < n ? alt(refs[c]) : REF_NONE;
  }
  return make_ref(KIND_INNER, idx);
}

uint32_t BvhBuilder::build(std::vector<BuildItem>& items, double lo[3], double hi[3]) {
  if (items.empty()) {
    for (int a = 0; a < 3; a++) {
      lo[a] = INF;
      hi[a] = -INF;
    }
    stack_need = 0;
    return REF_NONE;
  }
  if (on_device && items.size() >= device_min_items && items.size() >= 2) {
    DeviceBvh d;
    if (build_bvh_device(items.data(), items.size(), gate_alt, gate_need, &d) == 0) {
      const uint32_t node_off = (uint32_t)nodes.size(), tri_off = (uint32_t)tri_order.size(), quad_off = (uint32_t)quad_order.size();
      auto moved = [&](uint32_t r) -> uint32_t {
        if (r == REF_NONE) return r;
        switch (r >> 30) {
          case KIND_INNER: return r + node_off;
          case KIND_TRI: return r + tri_off;
          case KIND_QUAD: return r + quad_off;
          default: return r;
        }
      };
      for (BvhNode nd : d.nodes) {
        for (int c = 0; c < 4; c++) {
          const bool gate = nd.child[c] != REF_NONE && (nd.child[c] >> 30) == KIND_GATE;
          nd.child[c] = moved(nd.child[c]);
          if (!gate) nd.alt[c] = nd.child[c];  // a gate's alternative is a reference of this builder already
        }
        nodes.push_back(nd);
      }
      tri_order.insert(tri_order.end(), d.tri_order.begin(), d.tri_order.end());
      quad_order.insert(quad_order.end(), d.quad_order.begin(), d.quad_order.end());
      for (int a = 0; a < 3; a++) {
        lo[a] = INF;
        hi[a] = -INF;
      }
      for (const BuildItem& it : items)
        for (int a = 0; a < 3; a++) {
          lo[a] = std::min(lo[a], it.lo[a]);
          hi[a] = std::max(hi[a], it.hi[a]);
        }
      stack_need = d.stack_need;
      device_seconds += d.seconds;
      device_builds++;
      if (getenv("RT_AMD_BUILD_TIMING"))
        fprintf(stderr, "[rt_amd] device build of %zu items: %.2f ms, %zu nodes in %d levels, stack need %d\n", items.size(),
                d.seconds * 1e3, d.nodes.size(), d.levels, d.stack_need);
      return moved(d.root);
    }
    // refused (too deep, too large): the host builder takes over
  }
  bin_.clear();
  static const int default_threads = [] {
    int t = env_int("RT_AMD_BUILD_THREADS", 0);
    if (t <= 0) t = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return t;
  }();
  static const bool timing = getenv("RT_AMD_BUILD_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  uint32_t root = build_range(items.data(), items.size(), 1, lo, hi, threads > 0 ? threads : default_threads);
  auto t1 = std::chrono::steady_clock::now();
  uint32_t out = collapse(root, &stack_need);
  if (timing && items.size() > 1000)
    fprintf(stderr, "[rt_amd] build of %zu items: split %.1f ms, collapse %.1f ms\n", items.size(),
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return out;
}

}  // namespace rt
