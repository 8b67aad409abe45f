// What a traversal is asked: the closest hit, any hit before the light, or the shapes that contain
// the ray's origin (Intersections::hit, World::is_shadowed, refraction_indices of the reference).
#pragma once
#include "rt_math.h"

namespace rt {

// Closest hit = first finite t >= 0 of the stably sorted list
// (src/intersect.rs:71-89); equal t resolves to the lower depth-first leaf.
struct ClosestQuery {
  static constexpr bool BEHIND = false;
  double best_t;
  int best_leaf;
  int band_leaf;  // the best hit when the cones' parallel-ray list had been looked through
  double u, v;
  DEV void init() {
    best_t = INFINITY;
    best_leaf = band_leaf = 0x7FFFFFFF;
    u = v = 0.0;
  }
  DEV void after_bands() { band_leaf = best_leaf; }
  // The hit came from a cone's parallel-ray branch (src/shape.rs:1089-1103), which can lie anywhere on the
  // cone's infinite extension.  (A ray meets a cone either there or in the hierarchy, never in both.)
  DEV bool on_cone_extension() const { return best_leaf != 0x7FFFFFFF && best_leaf == band_leaf; }
  DEV double hi_d() const { return best_t; }
  DEV bool done() const { return false; }
  DEV void offer(double t, int leaf, double uu, double vv) {
    if (t >= 0.0 && isfinite(t) && (t < best_t || (t == best_t && leaf < best_leaf))) {
      best_t = t;
      best_leaf = leaf;
      u = uu;
      v = vv;
    }
  }
  DEV void leaf_ts(int leaf, int n, const double* t) {
    for (int i = 0; i < n; i++) offer(t[i], leaf, 0.0, 0.0);
  }
  DEV void leaf_tri(int leaf, double t, double uu, double vv) { offer(t, leaf, uu, vv); }
};

// World::is_shadowed (src/world.rs:123-137): some finite 0 <= t < distance exists.
struct ShadowQuery {
  static constexpr bool BEHIND = false;
  double distance;
  bool found;
  DEV double hi_d() const { return distance; }
  DEV bool done() const { return found; }
  DEV void after_bands() {}
  DEV void leaf_ts(int, int n, const double* t) {
    for (int i = 0; i < n; i++)
      if (t[i] >= 0.0 && t[i] < distance) found = true;  // a finite bound keeps out +inf; NaN fails both
  }
  DEV void leaf_tri(int, double t, double, double) {
    if (t >= 0.0 && t < distance) found = true;
  }
};

// refraction_indices (src/intersect.rs:191-234) as a streaming query: the list
// walk up to the hit only sees intersections with t < 0.  A shape is still in
// `containers` there iff it has an odd number of them, and its position in the
// vector is decided by its largest negative t (ties: depth-first order).
struct ContainerQuery {
  static constexpr bool BEHIND = true;  // wants t < 0: the boxes are walked along the reversed ray
  int hit_leaf;
  bool hit_inside;
  double hit_key;
  double other_key;
  int other_leaf;
  DEV void init(int leaf) {
    hit_leaf = leaf;
    hit_inside = false;
    hit_key = -INFINITY;
    other_key = -INFINITY;
    other_leaf = -1;
  }
  DEV double hi_d() const { return INFINITY; }
  DEV bool done() const { return false; }
  DEV void after_bands() {}
  DEV void container(int leaf, int count, double key) {
    if (!(count & 1)) return;
    if (leaf == hit_leaf) {
      hit_inside = true;
      hit_key = key;
    } else if (key > other_key || (key == other_key && leaf > other_leaf)) {
      other_key = key;
      other_leaf = leaf;
    }
  }
  DEV void leaf_ts(int leaf, int n, const double* t) {
    int count = 0;
    double key = -INFINITY;
    for (int i = 0; i < n; i++)
      if (t[i] < 0.0 && isfinite(t[i])) {
        count++;
        key = fmax(key, t[i]);
      }
    container(leaf, count, key);
  }
  DEV void leaf_tri(int leaf, double t, double, double) {
    if (t < 0.0 && isfinite(t)) container(leaf, 1, t);
  }
};

}  // namespace rt
