// Binned-SAH BVH over f64 boxes.  The reference tracer scans every child of a
// group linearly (src/shape.rs:1213-1219); any conservative hierarchy over the
// same leaves returns the same closest hit, so this structure is free to be
// whatever traverses fastest.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rt {
namespace {

constexpr int BINS = 16;
constexpr int MAX_TRI_LEAF = 4;
constexpr int MAX_QUAD_LEAF = 2;
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

uint32_t BvhBuilder::build_range(BuildItem* items, size_t n, int depth, double lo[3], double hi[3]) {
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
  max_depth = std::max(max_depth, depth);
  if (leaf_ok(items, n)) return make_leaf(items, n);

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

  uint32_t idx = (uint32_t)nodes.size();
  nodes.emplace_back();
  double l0[3], h0[3], l1[3], h1[3];
  uint32_t c0 = build_range(items, mid, depth + 1, l0, h0);
  uint32_t c1 = build_range(items + mid, n - mid, depth + 1, l1, h1);
  BvhNode& nd = nodes[idx];
  for (int a = 0; a < 3; a++) {
    nd.lo0[a] = l0[a]; nd.hi0[a] = h0[a];
    nd.lo1[a] = l1[a]; nd.hi1[a] = h1[a];
  }
  nd.c0 = c0;
  nd.c1 = c1;
  for (uint32_t& p : nd.pad) p = 0;
  return make_ref(KIND_INNER, idx);
}

uint32_t BvhBuilder::build(std::vector<BuildItem>& items, double lo[3], double hi[3]) {
  if (items.empty()) {
    for (int a = 0; a < 3; a++) {
      lo[a] = INF;
      hi[a] = -INF;
    }
    return REF_NONE;
  }
  return build_range(items.data(), items.size(), 1, lo, hi);
}

}  // namespace rt
