// Primitive intersection routines of the reference (src/shape.rs:839-1455), the group gate
// (src/shape.rs:1191-1211, src/geometry.rs:160-179) and the level-by-level ray transform.
#pragma once
#include "rt_math.h"

namespace rt {

DEV void check_cube_axis(double origin, double direction, double& tmin, double& tmax) {  // 1367-1385
  double tmin_numerator = -1.0 - origin;
  double tmax_numerator = 1.0 - origin;
  if (fabs(direction) >= FEQ_EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INFINITY;
    tmax = tmax_numerator * INFINITY;
  }
  if (tmin > tmax) {
    double s = tmin;
    tmin = tmax;
    tmax = s;
  }
}

// Intersections of an object-space ray with one quadric-type primitive, in the
// reference's emission order.  Returns their count (at most 4).
// type may carry QUAD_SKIP_PARALLEL / QUAD_ONLY_PARALLEL (cones only): the two copies of a bounded
// cone split its rays by the reference's own |a| < FEQ_EPSILON branch, so each ray meets one of them.
DEV int intersect_quadric(int type_flags, int closed, double minimum, double maximum, const Ray& r, double t[4]) {
  int n = 0;
  const int type = type_flags & QUAD_TYPE_MASK;
  if (type == RT_PLANE) {  // 891-908
    if (fabs(r.d.y) <= FEQ_EPSILON) return 0;
    t[0] = -r.o.y / r.d.y;
    return 1;
  }
  if (type == RT_CUBE) {  // 925-953
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_cube_axis(r.o.x, r.d.x, xtmin, xtmax);
    check_cube_axis(r.o.y, r.d.y, ytmin, ytmax);
    check_cube_axis(r.o.z, r.d.z, ztmin, ztmax);
    double tmin = fmax(fmax(xtmin, ytmin), ztmin);  // NaN-ignoring like f64::max
    double tmax = fmin(fmin(xtmax, ytmax), ztmax);
    if (tmin > tmax) return 0;
    t[0] = tmin;
    t[1] = tmax;
    return 2;
  }
  if (type == RT_SPHERE || type == RT_CYLINDER || type == RT_CONE) {
    // One quadratic a t^2 + b t + c = 0 for the three shapes (839-866, 981-1040, 1071-1144), so that
    // the lanes of a wave holding different shapes share the square root and the two divisions.
    // The y terms enter with the sign sy = +1 (sphere), 0 (cylinder), -1 (cone); multiplying by
    // +-1 or 0 and doubling are exact, so every shape still sees its own operations in its own order:
    //   sphere   a = dot(d, d)             b = 2 dot(d, o)                 c = dot(o, o) - 1
    //   cylinder a = dx^2 + dz^2           b = 2 ox dx + 2 oz dz           c = ox^2 + oz^2 - 1
    //   cone     a = dx^2 - dy^2 + dz^2    b = 2 ox dx - 2 oy dy + 2 oz dz c = ox^2 - oy^2 + oz^2
    const bool sphere = type == RT_SPHERE, cone = type == RT_CONE;
    const double sy = sphere ? 1.0 : (cone ? -1.0 : 0.0);
    const double a = (r.d.x * r.d.x + sy * (r.d.y * r.d.y)) + r.d.z * r.d.z;
    const double b = 2.0 * ((r.o.x * r.d.x + sy * (r.o.y * r.d.y)) + r.o.z * r.d.z);
    const double c = ((r.o.x * r.o.x + sy * (r.o.y * r.o.y)) + r.o.z * r.o.z) - (cone ? 0.0 : 1.0);
    // Cylinder caps exist when closed; cone caps when NOT closed (1396 vs 1431).
    const bool caps = !sphere && (cone ? !closed : (closed != 0)) && !(fabs(r.d.y) < FEQ_EPSILON);
    bool walls = true;
    if (cone) {
      if (fabs(a) < FEQ_EPSILON) {  // 1089-1103: caps first, then the single root
        if (type_flags & QUAD_SKIP_PARALLEL) return 0;
        walls = false;
      } else if (type_flags & QUAD_ONLY_PARALLEL) {
        return 0;
      }
    } else if (!sphere && a < FEQ_EPSILON) {
      walls = false;  // 990-994: caps only
    }
    if (walls) {
      double disc = b * b - 4.0 * a * c;
      if (disc < 0.0) return 0;  // 848-850 / 1004-1006 / 1108-1110: no caps either
      double sq = sqrt(disc);
      double t0 = (-b - sq) / (2.0 * a);
      double t1 = (-b + sq) / (2.0 * a);
      if (!sphere && t0 > t1) {
        double s = t0;
        t0 = t1;
        t1 = s;
      }
      double y0 = r.o.y + t0 * r.d.y;
      if (sphere || (minimum < y0 && y0 < maximum)) t[n++] = t0;
      double y1 = r.o.y + t1 * r.d.y;
      if (sphere || (minimum < y1 && y1 < maximum)) t[n++] = t1;
    }
    if (caps) {  // 1388-1455
      double tl = (minimum - r.o.y) / r.d.y;
      double xl = r.o.x + tl * r.d.x, zl = r.o.z + tl * r.d.z;
      double tu = (maximum - r.o.y) / r.d.y;
      double xu = r.o.x + tu * r.d.x, zu = r.o.z + tu * r.d.z;
      double rl = cone ? minimum * minimum : 1.0;
      double ru = cone ? maximum * maximum : 1.0;
      if (xl * xl + zl * zl <= rl) t[n++] = tl;
      if (xu * xu + zu * zu <= ru) t[n++] = tu;
    }
    if (cone && !walls && !(fabs(b) < FEQ_EPSILON)) t[n++] = -c / (2.0 * b);  // 1097-1102
    return n;
  }
  return 0;  // RT_EMPTY
}

// Moller-Trumbore as the reference writes it (src/shape.rs:1225-1260, 1273-1315).
DEV bool intersect_triangle(const TriIsect& tr, const Ray& r, double& t, double& u, double& v) {
  V3 e1 = v3(tr.e1[0], tr.e1[1], tr.e1[2]), e2 = v3(tr.e2[0], tr.e2[1], tr.e2[2]);
  V3 dir_cross_e2 = cross(r.d, e2);
  double det = dot(e1, dir_cross_e2);
  if (fabs(det) < FEQ_EPSILON) return false;
  double f = 1.0 / det;
  V3 p1_to_origin = r.o - v3(tr.p1[0], tr.p1[1], tr.p1[2]);
  u = f * dot(p1_to_origin, dir_cross_e2);
  if (u < 0.0) return false;
  V3 origin_cross_e1 = cross(p1_to_origin, e1);
  v = f * dot(r.d, origin_cross_e1);
  if (v < 0.0 || u + v > 1.0) return false;
  t = f * dot(e2, origin_cross_e1);
  return true;
}

// The bounding-box gate of intersect_group (src/shape.rs:1191-1211) with
// Bounds::check_axis (src/geometry.rs:160-179), in the group's own space.
DEV void gate_axis(double mn, double mx, double origin, double direction, double& tmin, double& tmax) {
  double tmin_numerator = mn - origin;
  double tmax_numerator = mx - origin;
  if (fabs(direction) >= FEQ_EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INFINITY;
    tmax = tmax_numerator * INFINITY;
  }
  if (tmin > tmax) {
    double s = tmin;
    tmin = tmax;
    tmax = s;
  }
}
// A ray taken through a transform chain: one local inverse after the other, outermost first, as the
// nested calls of shape::intersect do (src/shape.rs:1467-1470).
DEV Ray chain_ray(const SceneDev& S, uint32_t first, uint32_t len, Ray r) {
  for (uint32_t k = 0; k < len; k++) r = transform_ray(S.chain_mats + 12 * (size_t)(first + k), r);
  return r;
}
// GENERAL: the kernels built for scenes with CSG operators or transform chains.
template <bool GENERAL>
DEV Ray quad_ray(const SceneDev& S, const QuadPrim& q, const Ray& ray) {
  if constexpr (GENERAL) {
    if (q.type & QUAD_CHAIN) {
      const uint32_t x = (uint32_t)S.leaves[q.leaf].xform;
      return chain_ray(S, S.xform_chain[2 * x], S.xform_chain[2 * x + 1], ray);
    }
  }
  return transform_ray(q.m, ray);
}
template <bool GENERAL>
DEV Ray tri_ray(const SceneDev& S, const TriIsect& tr, const Ray& ray) {  // tr.xform >= 0
  if constexpr (GENERAL) {
    if (tr.xform & TRI_XFORM_CHAIN) {
      const uint32_t x = (uint32_t)(tr.xform & ~TRI_XFORM_CHAIN);
      return chain_ray(S, S.xform_chain[2 * x], S.xform_chain[2 * x + 1], ray);
    }
  }
  return transform_ray(S.xforms[tr.xform].ray_inv, ray);
}
template <bool GENERAL>
DEV bool gate_passes(const SceneDev& S, const Gate& g, const Ray& world_ray) {
  Ray r;
  if (GENERAL && g.chain_len) r = chain_ray(S, g.chain_first, g.chain_len, world_ray);
  else r = transform_ray(g.m, world_ray);
  double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
  gate_axis(g.bmin[0], g.bmax[0], r.o.x, r.d.x, xtmin, xtmax);
  gate_axis(g.bmin[1], g.bmax[1], r.o.y, r.d.y, ytmin, ytmax);
  gate_axis(g.bmin[2], g.bmax[2], r.o.z, r.d.z, ztmin, ztmax);
  double tmin = fmax(fmax(xtmin, ytmin), ztmin);
  double tmax = fmin(fmin(xtmax, ytmax), ztmax);
  return !(tmin > tmax);
}

}  // namespace rt
