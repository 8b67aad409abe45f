// HIP kernels of the MI355X (gfx950) render path and the C ABI around them.
//
// The reference renders each pixel by recursion: Camera::ray_for_pixel
// (src/camera.rs:76-95) -> World::color_at (src/world.rs:249-261) -> shade_hit
// (src/world.rs:220-246) with a shadow query, a reflection ray and a refraction
// ray per hit.  Here the recursion is unrolled into generations of rays held in
// HBM (struct-of-arrays, one record per ray):
//
//   [ generation kernel ] per generation -> resolve (bottom-up) -> output
//
// generation  one ray per lane, start to finish: camera ray or queued ray ->
//             closest hit against a world-space BVH (plus the refractive-index
//             query for hits on transparent materials) -> hit point, normal,
//             Phong terms, pattern lookup -> reflection / refraction rays
//             appended to the next generation's queue (wave-level ballot +
//             prefix, one atomic per wave) -> shadow query towards the light
//             -> surface colour.  Ray, hit and lighting terms never leave
//             registers; a scene without reflective or transparent materials
//             finishes every pixel in this one kernel.
// resolve     walks the generations backwards and combines each ray's surface
//             colour with its children's colours in the reference's own
//             expression order, so the f64 result equals the recursive evaluation
//
// All arithmetic is f64 with the reference's operation order; build with
// -ffp-contract=off (Rust never fuses a multiply with an add).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "band_grid.h"
#include "band_grid_test.h"
#include "bvh_build.h"
#include "rt_error.h"
#include "rt_types.h"

using namespace rt;

// The layout of every structure that crosses the C ABI, as a binding in another language has to
// declare it (`#[repr(C)]` in the reference's Rust): sizes and field offsets in bytes.
#define RT_ABI_FIELD(T, f, off) static_assert(offsetof(T, f) == (off), #T "." #f " moved: the C ABI changed")
#define RT_ABI_SIZE(T, n) static_assert(sizeof(T) == (n), #T " changed size: the C ABI changed")
RT_ABI_SIZE(rt_xform, 384); RT_ABI_FIELD(rt_xform, ray_inv, 0); RT_ABI_FIELD(rt_xform, full_inv, 128); RT_ABI_FIELD(rt_xform, fwd, 256);
RT_ABI_SIZE(rt_triangle, 168); RT_ABI_FIELD(rt_triangle, p1, 0); RT_ABI_FIELD(rt_triangle, e1, 24); RT_ABI_FIELD(rt_triangle, e2, 48);
RT_ABI_FIELD(rt_triangle, normal, 72); RT_ABI_FIELD(rt_triangle, n1, 96); RT_ABI_FIELD(rt_triangle, n2, 120); RT_ABI_FIELD(rt_triangle, n3, 144);
RT_ABI_SIZE(rt_leaf, 72); RT_ABI_FIELD(rt_leaf, type, 0); RT_ABI_FIELD(rt_leaf, closed, 4); RT_ABI_FIELD(rt_leaf, material, 8);
RT_ABI_FIELD(rt_leaf, xform, 12); RT_ABI_FIELD(rt_leaf, gate, 16); RT_ABI_FIELD(rt_leaf, tri, 20); RT_ABI_FIELD(rt_leaf, csg, 24);
RT_ABI_FIELD(rt_leaf, csg_left, 28); RT_ABI_FIELD(rt_leaf, minimum, 32); RT_ABI_FIELD(rt_leaf, maximum, 40); RT_ABI_FIELD(rt_leaf, plane_normal, 48);
RT_ABI_SIZE(rt_gate, 184); RT_ABI_FIELD(rt_gate, parent, 0); RT_ABI_FIELD(rt_gate, csg, 4); RT_ABI_FIELD(rt_gate, to_local, 8);
RT_ABI_FIELD(rt_gate, bmin, 136); RT_ABI_FIELD(rt_gate, bmax, 160);
RT_ABI_SIZE(rt_csg, 24); RT_ABI_FIELD(rt_csg, op, 0); RT_ABI_FIELD(rt_csg, parent, 4); RT_ABI_FIELD(rt_csg, gate, 8);
RT_ABI_FIELD(rt_csg, first_leaf, 12); RT_ABI_FIELD(rt_csg, n_leaves, 16); RT_ABI_FIELD(rt_csg, reserved, 20);
RT_ABI_SIZE(rt_material, 88); RT_ABI_FIELD(rt_material, color, 0); RT_ABI_FIELD(rt_material, ambient, 24); RT_ABI_FIELD(rt_material, diffuse, 32);
RT_ABI_FIELD(rt_material, specular, 40); RT_ABI_FIELD(rt_material, shininess, 48); RT_ABI_FIELD(rt_material, reflective, 56);
RT_ABI_FIELD(rt_material, refractive_index, 64); RT_ABI_FIELD(rt_material, transparency, 72); RT_ABI_FIELD(rt_material, pattern, 80);
RT_ABI_SIZE(rt_pattern, 192); RT_ABI_FIELD(rt_pattern, type, 0); RT_ABI_FIELD(rt_pattern, first, 4); RT_ABI_FIELD(rt_pattern, second, 8);
RT_ABI_FIELD(rt_pattern, inv, 16); RT_ABI_FIELD(rt_pattern, a, 144); RT_ABI_FIELD(rt_pattern, b, 168);
RT_ABI_SIZE(rt_light, 48); RT_ABI_FIELD(rt_light, intensity, 0); RT_ABI_FIELD(rt_light, position, 24);
RT_ABI_SIZE(rt_camera, 160); RT_ABI_FIELD(rt_camera, hsize, 0); RT_ABI_FIELD(rt_camera, vsize, 4); RT_ABI_FIELD(rt_camera, half_width, 8);
RT_ABI_FIELD(rt_camera, half_height, 16); RT_ABI_FIELD(rt_camera, pixel_size, 24); RT_ABI_FIELD(rt_camera, inv, 32);
RT_ABI_SIZE(rt_chain_ref, 8); RT_ABI_FIELD(rt_chain_ref, first, 0); RT_ABI_FIELD(rt_chain_ref, count, 4);
RT_ABI_SIZE(rt_scene_desc, 160); RT_ABI_FIELD(rt_scene_desc, light, 0); RT_ABI_FIELD(rt_scene_desc, n_leaves, 48);
RT_ABI_FIELD(rt_scene_desc, n_gates, 52); RT_ABI_FIELD(rt_scene_desc, n_xforms, 56); RT_ABI_FIELD(rt_scene_desc, n_triangles, 60);
RT_ABI_FIELD(rt_scene_desc, n_materials, 64); RT_ABI_FIELD(rt_scene_desc, n_patterns, 68); RT_ABI_FIELD(rt_scene_desc, n_csg, 72);
RT_ABI_FIELD(rt_scene_desc, n_chain_mats, 76); RT_ABI_FIELD(rt_scene_desc, csg, 80); RT_ABI_FIELD(rt_scene_desc, leaves, 88);
RT_ABI_FIELD(rt_scene_desc, gates, 96); RT_ABI_FIELD(rt_scene_desc, xforms, 104); RT_ABI_FIELD(rt_scene_desc, triangles, 112);
RT_ABI_FIELD(rt_scene_desc, materials, 120); RT_ABI_FIELD(rt_scene_desc, patterns, 128); RT_ABI_FIELD(rt_scene_desc, chain_mats, 136);
RT_ABI_FIELD(rt_scene_desc, xform_chain, 144); RT_ABI_FIELD(rt_scene_desc, gate_chain, 152);
RT_ABI_SIZE(rt_stats, 128); RT_ABI_FIELD(rt_stats, primary, 0); RT_ABI_FIELD(rt_stats, render_ms, 32); RT_ABI_FIELD(rt_stats, kernel_ms, 40);
RT_ABI_FIELD(rt_stats, kernel_launches, 48); RT_ABI_FIELD(rt_stats, generations, 56); RT_ABI_FIELD(rt_stats, node_visits, 64);
RT_ABI_FIELD(rt_stats, shadow_node_visits, 96);
#undef RT_ABI_FIELD
#undef RT_ABI_SIZE

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int rt_fail_message(int code, const char* msg) { return fail(code, msg ? msg : ""); }
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* reported here: a later launch check must not see it again */ \
      return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    }                                                                                       \
  } while (0)

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#define DEV __device__ __forceinline__

struct V3 {
  double x, y, z;
};
struct Ray {
  V3 o, d;
};

DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
// Tuple4D::dot with both w == 0 (src/tuple.rs:145-150): the w product adds +0.0
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) {  // src/tuple.rs:160-167
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEV double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // src/tuple.rs:111-118
DEV V3 normalize(V3 a) {  // src/tuple.rs:126-135: multiply by the reciprocal
  double mag = magnitude(a);
  return v3(a.x * (1.0 / mag), a.y * (1.0 / mag), a.z * (1.0 / mag));
}
DEV V3 reflect(V3 v, V3 n) { return v - ((n * 2.0) * dot(v, n)); }  // src/tuple.rs:170-172
DEV bool feq(double a, double b) { return fabs(a - b) < FEQ_EPSILON; }

// Matrix4D * point (w = 1) and * vector (w = 0), rows 0..2 (src/matrix.rs:638-653)
DEV V3 mul_point(const double* m, V3 p) {
  return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
DEV V3 mul_vector(const double* m, V3 v) {
  return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
DEV Ray transform_ray(const double* m, const Ray& r) {  // src/ray.rs:50-55
  return Ray{mul_point(m, r.o), mul_vector(m, r.d)};
}

struct Counters {
  unsigned long long node_visits, tri_tests, quad_tests, gate_tests;
#ifdef RT_STAMPS
  unsigned long long* sink;  // four sums: unbounded + setup, node loop, triangle runs, quadric runs + gates
  unsigned long long tt;
#endif
#if defined(RT_WAVE_CLOCK_IMAGE) && RT_WAVE_CLOCK_IMAGE == 2
  unsigned wi[4];  // this lane's share of the wave-level loop iterations (it counts those it leads)
#endif
};
#ifdef RT_STAMPS
// one count per wave and loop iteration (by the first active lane): lane events / (64 x these) = lane utilisation
#define RT_WAVE_ITER(c, k) do { if ((c).sink && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd((c).sink + 4 + (k), 1ull); } while (0)
#define RT_TBEGIN(c) (c).tt = __builtin_amdgcn_s_memtime()
#define RT_TSTAMP(c, k) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if ((c).sink && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd((c).sink + (k), now_ - (c).tt); \
    (c).tt = now_; } while (0)
#elif defined(RT_WAVE_CLOCK_IMAGE) && RT_WAVE_CLOCK_IMAGE == 2
#define RT_TBEGIN(c) do { } while (0)
#define RT_TSTAMP(c, k) do { } while (0)
#define RT_WAVE_ITER(c, k) do { if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) (c).wi[k]++; } while (0)
#else
#define RT_TBEGIN(c) do { } while (0)
#define RT_TSTAMP(c, k) do { } while (0)
#define RT_WAVE_ITER(c, k) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// primitive intersection, reference src/shape.rs:839-1455
// ---------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------
// query policies
// ---------------------------------------------------------------------------
// Closest hit = first finite t >= 0 of the stably sorted list
// (src/intersect.rs:71-89); equal t resolves to the lower depth-first leaf.
struct ClosestQuery {
  static constexpr bool BEHIND = false;
  double best_t;
  int best_leaf;
  double u, v;
  DEV void init() {
    best_t = INFINITY;
    best_leaf = 0x7FFFFFFF;
    u = v = 0.0;
  }
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

// ---------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------
// Per-lane stack in LDS: entry e of thread t lives at stack[e * BLOCK + t], so
// the 64 lanes of a wave hit 64 consecutive banks.
#ifndef RT_PRIVATE_STACK
template <int STACK, int BLOCK, bool CSG = false, bool UNIFORM = false, bool TRIPF = false>
struct LdsStack {
  static constexpr bool WITH_CSG = CSG;  // the kernels built for scenes with CSG operators carry eval_csg
  static constexpr bool UNIFORM_NODES = UNIFORM;  // ... and those for large hierarchies the scalar node path
  static constexpr bool TRI_PREFETCH = TRIPF;  // ... and those for large meshes fetch a run's next triangle early
  static constexpr int THREADS = BLOCK;
  static constexpr int ENTRIES = STACK;
  uint32_t* base;
  uint32_t* csg;  // this lane's column of the CSG hit list (dynamic LDS), or null when the scene has no CSG
  int sp;
  DEV void init(uint32_t* lds, uint32_t* csg_lds = nullptr) {
    const uint32_t tid = BLOCK == 64 ? (uint32_t)__lane_id() : threadIdx.x;
    base = lds + tid;
    csg = csg_lds ? csg_lds + tid : nullptr;
    sp = 0;
  }
  DEV void push(uint32_t r) {
    if (sp < STACK) base[(sp++) * BLOCK] = r;
  }
  // Unconditional store, conditional advance: the slot above the top is scratch.  The host sizes
  // the stack three entries above the deepest need, so the store never leaves the lane's column.
  DEV void push_if(uint32_t r, bool take) {
    base[sp * BLOCK] = r;
    sp += take ? 1 : 0;
  }
  DEV uint32_t pop() { return base[(--sp) * BLOCK]; }
  DEV uint32_t peek(int i) const { return base[i * BLOCK]; }  // entry i of this lane's column, used as a queue
  DEV void poke(int i, uint32_t r) { base[i * BLOCK] = r; }
  DEV uint32_t pop_or_none() { return sp > 0 ? base[(--sp) * BLOCK] : REF_NONE; }
  DEV bool empty() const { return sp == 0; }
  DEV void clear() { sp = 0; }
};
#define RT_STACK_LDS(n) (n)
#else
// Experiment: per-lane stack in private (scratch) memory instead of LDS.
template <int STACK, int BLOCK, bool CSG = false, bool UNIFORM = false, bool TRIPF = false>
struct LdsStack {
  static constexpr bool WITH_CSG = CSG;
  static constexpr bool UNIFORM_NODES = UNIFORM;
  static constexpr bool TRI_PREFETCH = TRIPF;
  static constexpr int THREADS = BLOCK;
  static constexpr int ENTRIES = STACK;
  uint32_t slot[STACK];
  uint32_t* csg;
  int sp;
  DEV void init(uint32_t*, uint32_t* csg_lds = nullptr) {
    csg = csg_lds ? csg_lds + threadIdx.x : nullptr;
    sp = 0;
  }
  DEV void push(uint32_t r) {
    if (sp < STACK) slot[sp++] = r;
  }
  DEV void push_if(uint32_t r, bool take) {
    slot[sp] = r;
    sp += take ? 1 : 0;
  }
  DEV uint32_t pop() { return slot[--sp]; }
  DEV uint32_t peek(int i) const { return slot[i]; }
  DEV void poke(int i, uint32_t r) { slot[i] = r; }
  DEV uint32_t pop_or_none() { return sp > 0 ? slot[--sp] : REF_NONE; }
  DEV bool empty() const { return sp == 0; }
  DEV void clear() { sp = 0; }
};
#define RT_STACK_LDS(n) 1
#endif

// What a ray needs for the f32 box tests.  Per axis t(plane) = (plane - origin) / direction; the
// near plane of an axis is the lower one when the direction is positive.  Only t >= 0 matters
// (queries behind the origin walk the reversed ray), so a test stays conservative when every near
// value is computed too small and every far value too large:
//   inv          1 / direction from v_rcp_f32 on the direction rounded to f32 (relative error below
//                2^-22); origin * inv is formed with that same inv, so the error of inv only scales t
//   near / far   inv and origin * inv scaled by (1 -/+ 1.5e-6), which covers the error of inv, the
//                rounding of the scaled values and that of the box test's multiply-add (2^-24 |t| each)
//   c_n / c_f    additionally moved outward by 1e-6 |origin * inv|: covers the f32 roundings of the
//                origin, of origin * inv and of the scaled value, which are relative to
//                |origin * inv|, not to t (2^-24 each, 2^-22 in all)
// A box test can therefore pass wrongly but never fail wrongly.
struct RayBox {
  float inv_n[3], c_n[3], inv_f[3], c_f[3];
  uint32_t off_near[3];  // byte offset of the near plane of each axis in a BvhNode; far = near ^ 16
  bool tiny;             // some |direction| component is below FEQ_EPSILON
};

// Rounding of the reference's quadratic for round shapes, relative to |object-space origin|^2 (derived at
// round_shape_noise() below), with a factor of four in hand.
constexpr double ROUND_NOISE = 4.0 * 6.6e-16;

// widen: a distance (world units) by which every box is to be taken larger for this ray, see round_shape_noise().
DEV RayBox make_raybox(const Ray& r, bool reversed, float widen = 0.0f) {
  RayBox b;
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  b.tiny = false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double dir = reversed ? -d[a] : d[a];
    if (fabs(dir) < FEQ_EPSILON) b.tiny = true;
    float dirf = (float)dir;
    if (!(fabsf(dirf) >= 1e-20f)) dirf = copysignf(1e-20f, dirf);  // parallel: a huge finite slope, no inf * 0
    // Everything below in single precision.  inv is within 2^-22 of 1 / direction (rounding of the
    // direction, v_rcp_f32), oinv within 2^-22 |oinv| of origin / direction (rounding of the origin,
    // of inv, of the product); scaling by (1 -/+ 1.5e-6) covers the first and the roundings of the
    // scaled values and of the box test's multiply-add, the slack of 1e-6 |oinv| the second.
    const float inv = __builtin_amdgcn_rcpf(dirf);
    const float oinv = (float)o[a] * inv;
    const float slack = fabsf(oinv) * 1e-6f + 1e-30f + widen * fabsf(inv) * (1.0f + 1e-6f);
    const float lo_s = 1.0f - 1.5e-6f, hi_s = 1.0f + 1.5e-6f;
    b.inv_n[a] = inv * lo_s;
    b.inv_f[a] = inv * hi_s;
    b.c_n[a] = fmaf(oinv, lo_s, slack);
    b.c_f[a] = fmaf(oinv, hi_s, -slack);
    b.off_near[a] = 32u * a + (inv < 0.0f ? 16u : 0u);
  }
  return b;
}

struct F4 {
  float v[4];
};
DEV F4 load_f4(const char* base, uint32_t off) {
  float4 q = *reinterpret_cast<const float4*>(base + off);
  return F4{{q.x, q.y, q.z, q.w}};
}

// The same 16 bytes for a wave-uniform address: read through the constant address space, so the
// load goes through the scalar cache into SGPRs and not through the per-CU vector data path, whose
// return bandwidth (64 B per clock, 1 KB per wave instruction whatever the addresses) bounds the
// node loop.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEV F4 load_f4_uniform(const char* base, uint32_t off) {
  f32x4 q = *(const __attribute__((address_space(4))) f32x4*)(base + off);
  return F4{{q.x, q.y, q.z, q.w}};
}

// Slab tests of the four children of one node: entry distance (INFINITY when missed) per child,
// and the references of missed children cleared.
DEV void slab4(const RayBox& rb, float qhi, const F4& nx, const F4& ny, const F4& nz, const F4& fx, const F4& fy,
               const F4& fz, float tn[4], uint32_t c[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float tnear = fmaxf(fmaxf(fmaf(nx.v[k], rb.inv_n[0], -rb.c_n[0]), fmaf(ny.v[k], rb.inv_n[1], -rb.c_n[1])),
                        fmaf(nz.v[k], rb.inv_n[2], -rb.c_n[2]));
    float tfar = fminf(fminf(fmaf(fx.v[k], rb.inv_f[0], -rb.c_f[0]), fmaf(fy.v[k], rb.inv_f[1], -rb.c_f[1])),
                       fmaf(fz.v[k], rb.inv_f[2], -rb.c_f[2]));
    bool hit = fmaxf(tnear, 0.0f) <= fminf(tfar, qhi);
    tn[k] = hit ? tnear : INFINITY;
    if (!hit) c[k] = REF_NONE;
  }
}

struct __attribute__((packed, aligned(4))) StripWords {  // four strip words of a cell's list, at any word address
  uint32_t x, y, z, w;
};

// The intersections of one primitive with a world-space ray, its enclosing groups' gates included when
// asked for: how many (0..4), written to t in the reference's emission order.
template <bool GENERAL>
DEV int eval_quad(const SceneDev& S, const QuadPrim& q, const Ray& ray, double t[4], bool check_gates) {
  if (check_gates) {
    for (int g = q.gate; g >= 0; g = S.gates[g].parent)
      if (!gate_passes<GENERAL>(S, S.gates[g], ray)) return 0;
  }
#ifndef RT_NO_PLANE_FAST
  if (q.type == RT_PLANE) {
    // only the y row of the transformed ray enters intersect_plane (src/shape.rs:891-908)
    const double oy = q.m[4] * ray.o.x + q.m[5] * ray.o.y + q.m[6] * ray.o.z + q.m[7];
    const double dy = q.m[4] * ray.d.x + q.m[5] * ray.d.y + q.m[6] * ray.d.z;
    if (fabs(dy) <= FEQ_EPSILON) return 0;
    t[0] = -oy / dy;
    return 1;
  }
#endif
  Ray local = quad_ray<GENERAL>(S, q, ray);  // src/shape.rs:1468
  return intersect_quadric(q.type, q.closed, q.minimum, q.maximum, local, t);
}
template <class Query, bool GENERAL>
DEV void test_quad(const SceneDev& S, const QuadPrim& q, const Ray& ray, Query& Q, bool check_gates) {
  double t[4];
  const int n = eval_quad<GENERAL>(S, q, ray, t, check_gates);
  if (n) Q.leaf_ts(q.leaf, n, t);
}

// intersect_csg + filter_intersections (src/shape.rs:1339-1354, src/intersect.rs:278-328) for one
// root operator tree.  The reference intersects both operands, sorts, and lets an intersection
// through depending on which operand it came from and whether the ray is inside the other; nested
// operators filter their own lists first.  Here all leaf intersections below the root go into one
// list sorted by (t, leaf order) -- the order every stable sort of the reference yields -- and the
// operators run innermost first over the entries that are theirs and still alive.
// The list lives in LDS: entry e of a lane at words (3 e + {0, 1, 2}) * BLOCK: t low, t high, leaf slot.
template <class Query, class Stack, int BLOCK_>
DEV void eval_csg(const SceneDev& S, uint32_t root_index, const Ray& ray, Query& Q, Stack& st, bool outer_gates) {
  const CsgRoot R = S.csg_roots[root_index];
  uint32_t* list = st.csg;
  if (!list) return;
  if (outer_gates)
    for (int g = R.gate; g >= 0; g = S.gates[g].parent)
      if (!gate_passes<true>(S, S.gates[g], ray)) return;
  auto put = [&](int e, double t, uint32_t who) {
    list[(3 * e + 0) * BLOCK_] = (uint32_t)__double2loint(t);
    list[(3 * e + 1) * BLOCK_] = (uint32_t)__double2hiint(t);
    list[(3 * e + 2) * BLOCK_] = who;
  };
  auto get_t = [&](int e) { return __hiloint2double((int)list[(3 * e + 1) * BLOCK_], (int)list[(3 * e + 0) * BLOCK_]); };
  auto get_who = [&](int e) { return list[(3 * e + 2) * BLOCK_]; };
  int n = 0;
  for (uint32_t j = 0; j < R.n_leaves; j++) {
    const CsgLeaf L = S.csg_leaves[R.first_leaf + j];
    bool open = true;
    for (int g = L.gate; g != R.gate && g >= 0; g = S.gates[g].parent)  // groups between the root and the leaf
      if (!gate_passes<true>(S, S.gates[g], ray)) {
        open = false;
        break;
      }
    if (!open) continue;
    double t[4];
    int m = 0;
    const uint32_t idx = L.ref & MAX_RUN_FIRST;
    if ((L.ref >> 30) == KIND_TRI) {
      const TriIsect& tr = S.tris[idx];
      double u, v;
      bool hit = tr.xform >= 0 ? intersect_triangle(tr, tri_ray<true>(S, tr, ray), t[0], u, v)
                               : intersect_triangle(tr, ray, t[0], u, v);
      m = hit ? 1 : 0;
    } else {
      const QuadPrim& q = S.quads[idx];
      m = intersect_quadric(q.type, q.closed, q.minimum, q.maximum, quad_ray<true>(S, q, ray), t);
    }
    for (int k = 0; k < m; k++) {
      if (!(t[k] == t[k]) || n >= S.csg_hits) continue;  // NaN never orders (the reference's sort would panic)
      int e = n;
      while (e > 0 && get_t(e - 1) > t[k]) {  // stable: equal t stays behind earlier leaves
        put(e, get_t(e - 1), get_who(e - 1));
        e--;
      }
      put(e, t[k], j);
      n++;
    }
  }
  if (n == 0) return;
  static_assert(CSG_MAX_HITS <= 64, "one bit of `alive` per list entry");
  unsigned long long alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  for (int c = (int)R.n_nodes - 1; c >= 0; c--) {  // children carry larger indices than their parents
    const int op = S.csg_nodes[R.first_node + c].op;
    bool inl = false, inr = false;
    for (int e = 0; e < n; e++) {
      if (!((alive >> e) & 1ull)) continue;
      const CsgLeaf& L = S.csg_leaves[R.first_leaf + get_who(e)];
      int a = L.node, level = 0;
      while (a > c) {
        a = S.csg_nodes[R.first_node + a].parent;
        level++;
      }
      if (a != c) continue;  // not below this operator
      const bool lhit = (L.left_mask >> level) & 1u;
      bool allowed;  // intersection_allowed (src/intersect.rs:278-290)
      if (op == RT_CSG_UNION) allowed = lhit ? !inr : !inl;
      else if (op == RT_CSG_INTERSECTION) allowed = lhit ? inr : inl;
      else allowed = lhit ? !inr : inl;
      if (lhit) inl = !inl;
      else inr = !inr;
      if (!allowed) alive &= ~(1ull << e);
    }
  }
  if (!alive) return;
  for (uint32_t j = 0; j < R.n_leaves; j++) {
    double t[4];
    int m = 0;
    for (int e = 0; e < n; e++)
      if (((alive >> e) & 1ull) && get_who(e) == j && m < 4) t[m++] = get_t(e);
    if (!m) continue;
    const CsgLeaf L = S.csg_leaves[R.first_leaf + j];
    const uint32_t idx = L.ref & MAX_RUN_FIRST;
    if ((L.ref >> 30) == KIND_TRI) {
      const TriIsect& tr = S.tris[idx];
      double tt, u, v;  // again, for the barycentric coordinates of a smooth triangle
      bool hit = tr.xform >= 0 ? intersect_triangle(tr, tri_ray<true>(S, tr, ray), tt, u, v)
                               : intersect_triangle(tr, ray, tt, u, v);
      if (hit) Q.leaf_tri(tr.leaf, tt, u, v);
    } else {
      Q.leaf_ts(S.quads[idx].leaf, m, t);
    }
  }
}

// World::intersect (src/world.rs:107-117) over the acceleration structure: first the primitives
// without a finite box, then the 4-wide BVH.  While-while traversal: every lane first descends
// inner nodes (the nearest hit child stays in a register, the others go on the LDS stack), and
// only when all lanes of the wave have reached a primitive run or a gate is that work done.
#ifndef RT_BAND_BATCH
#define RT_BAND_BATCH 2  // cone-band list entries fetched together
#endif
template <class Query, class Stack, bool COUNT>
DEV void traverse(const SceneDev& S, const Ray& ray, Query& Q, Stack& st, Counters& cnt) {
  RT_TBEGIN(cnt);
#pragma unroll
  for (int i = 0; i < MAX_INLINE_PLANES; i++) {
    if (i < S.n_inline_planes) {  // intersect_plane (src/shape.rs:891-908) on the y row of the transformed ray
      const InlinePlane& P = S.inline_planes[i];
      if (COUNT) cnt.quad_tests++;
      const double oy = P.row[0] * ray.o.x + P.row[1] * ray.o.y + P.row[2] * ray.o.z + P.row[3];
      const double dy = P.row[0] * ray.d.x + P.row[1] * ray.d.y + P.row[2] * ray.d.z;
      if (!(fabs(dy) <= FEQ_EPSILON)) {
        double t[1] = {-oy / dy};
        Q.leaf_ts(P.leaf, 1, t);
      }
    }
  }
  for (int i = 0; i < S.n_unbounded; i++) {
    if (COUNT) cnt.quad_tests++;
    test_quad<Query, Stack::WITH_CSG>(S, S.unbounded[i], ray, Q, true);
  }
  if constexpr (Stack::WITH_CSG)
    for (int i = 0; i < S.n_csg_unbounded; i++) eval_csg<Query, Stack, Stack::THREADS>(S, S.csg_unbounded[i], ray, Q, st, true);
  // The cones' parallel-ray branch (src/shape.rs:1089-1103): the cones whose band holds the ray's direction.
  auto band_list = [&]() {
    const double dd = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
    if (S.band_grid_n > 0 && dd >= 1.0 - 1e-6) {
      // cell of the direction on the sign-folded cube map; the grid was built for |d| >= 1 - 1e-6
      const int n = S.band_grid_n;
      const float fx = (float)ray.d.x, fy = (float)ray.d.y, fz = (float)ray.d.z;
      const float fxx = fx * fx, fyy = fy * fy, fzz = fz * fz, fxy = fx * fy, fxz = fx * fz, fyz = fy * fz;
      int m = fzz > fyy ? 2 : 1;
      m = fxx >= (m == 2 ? fzz : fyy) ? 0 : m;
      const float dm = m == 0 ? fx : (m == 1 ? fy : fz);
      const float da = m == 0 ? fy : (m == 1 ? fz : fx);
      const float dc = m == 0 ? fz : (m == 1 ? fx : fy);
      const float inv = __builtin_amdgcn_rcpf(dm);
      const float half_n = 0.5f * (float)n;
      int iu, iv;
      float ls, lt;  // where in its cell the direction lies, -1..1 per axis
      band_strip_coords((da * inv + 1.0f) * half_n, (dc * inv + 1.0f) * half_n, n, iu, iv, ls, lt);
      const uint32_t cell = ((uint32_t)m * n + iv) * n + iu;
      // A cell lists every cone whose band crosses it, and with each entry a word that says where in the
      // cell the band lies (a strip, band_grid_test.h).  Three passes, each over all lanes in step, because
      // what is rare for one ray happens in every wave:
      //   1. the strip words of the cell, four per load from one contiguous array: one multiply-add and a
      //      compare per entry, no look-up; the few entries whose strip holds the direction are queued on
      //      the traversal stack (free until the hierarchy is walked);
      //   2. the queued entries: band number, then the band's quadratic form in single precision (it passes
      //      slightly more than the double-precision band); survivors stay queued as primitive numbers;
      //   3. the cone's own test (src/shape.rs:1089) for what is left: it runs as often as the fullest
      //      queue is long, not once per entry some lane happens to pass.  (Dealing these tests evenly over
      //      the lanes of the wave through LDS was measured and is no faster: with 25 000 cones the tests
      //      are 1.3 ms of a 15.7 ms frame.)
      const uint32_t end = S.band_start[cell + 1];
#ifdef RT_DIAG_NO_BANDS  // timing builds (wrong frames): the passes below stop after the cell look-up (0) .. run in full (3)
      const int band_passes = S.band_diag_passes;
#else
      constexpr int band_passes = 3;
#endif
      const float fdd = (fxx + fyy + fzz) * (1.0f + 1e-6f);
      const uint16_t* items16 = reinterpret_cast<const uint16_t*>(S.band_items);
      st.clear();
      auto flush = [&]() {
        if (band_passes <= 1) { st.sp = 0; return; }
        int wp = 0;  // survivors are written over the entries already read (wp <= rp)
        for (int rp = 0; __any(rp < st.sp); rp += RT_BAND_BATCH) {
          uint32_t id[RT_BAND_BATCH];
          ConeBandF32 f[RT_BAND_BATCH];
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) {
            const uint32_t at = rp + j < st.sp ? st.peek(rp + j) : 0u;  // entry 0 exists: the index is not empty
            id[j] = S.band_items16 ? (uint32_t)items16[at] : S.band_items[at];
          }
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) f[j] = S.bands32[id[j]];
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) {
            const float a = fmaf(f[j].q[0], fxx, fmaf(f[j].q[1], fyy, fmaf(f[j].q[2], fzz,
                            fmaf(f[j].q[3], fxy, fmaf(f[j].q[4], fxz, f[j].q[5] * fyz)))));
            if (rp + j < st.sp && fabsf(a) < 1.0001e-4f + f[j].slack * fdd) st.poke(wp++, f[j].prim);
          }
        }
        st.sp = wp;
        if (band_passes <= 2) st.sp = 0;
#ifdef RT_DIAG_NO_BANDS
        Query dropped = Q;  // RT_AMD_TIMING_BAND_PASSES=4: the tests run, their intersections are dropped
#endif
        while (__any(st.sp > 0)) {
          if (st.sp > 0) {
            const uint32_t prim = st.pop();
            if (COUNT) cnt.quad_tests++;
#ifdef RT_DIAG_NO_BANDS
            if (band_passes == 4) test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[prim], ray, dropped, true);
            else
#endif
            test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[prim], ray, Q, true);
          }
        }
      };
      const uint32_t begin = S.band_start[cell];
      uint32_t stop = end;
      if (band_passes == 0 && end != 0xFFFFFFFFu) stop = begin;  // the offsets are still fetched
      // all lanes stay in the loop until the longest list is through: flush() works on the whole wave
      const float ls1024 = ls * 1024.0f, lt1024 = lt * 1024.0f;
      auto words_at = [&](uint32_t k) {  // four strip words of the list, or padding behind its end (four words)
        return *reinterpret_cast<const StripWords*>(S.band_strip + (k < stop ? k : stop));
      };
      StripWords next = words_at(begin);
      for (uint32_t k = begin; __any(k < stop); k += 4) {
        const StripWords w = next;
        next = words_at(k + 4);  // in flight while these four are looked at
        if (__any(st.sp + 4 > Stack::ENTRIES)) flush();
        st.push_if(k, k < stop && band_strip_passes(w.x, ls, lt, ls1024, lt1024));  // room for four was made above
        st.push_if(k + 1, k + 1 < stop && band_strip_passes(w.y, ls, lt, ls1024, lt1024));
        st.push_if(k + 2, k + 2 < stop && band_strip_passes(w.z, ls, lt, ls1024, lt1024));
        st.push_if(k + 3, k + 3 < stop && band_strip_passes(w.w, ls, lt, ls1024, lt1024));
      }
      flush();
    } else {
      const double xx = ray.d.x * ray.d.x, yy = ray.d.y * ray.d.y, zz = ray.d.z * ray.d.z;
      const double xy = ray.d.x * ray.d.y, xz = ray.d.x * ray.d.z, yz = ray.d.y * ray.d.z;
      for (int i = 0; i < S.n_bands; i++) {
        const ConeBand& cb = S.bands[i];
        double a = fma(cb.q[0], xx, fma(cb.q[1], yy, fma(cb.q[2], zz, fma(cb.q[3], xy, fma(cb.q[4], xz, cb.q[5] * yz)))));
        if (fabs(a) < FEQ_EPSILON * (1.0 + 1e-6) + cb.slack * dd) {
          if (COUNT) cnt.quad_tests++;
          test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[cb.prim], ray, Q, true);
        }
      }
    }
  };
  // The order in which a query meets the shapes does not change its answer; the bands come first because
  // their hits shorten the walk.  (A shadow query that walks the hierarchy first and looks the bands up
  // only when it found nothing was measured: correct, but the shadow ray of a hit far out on a cone's
  // extension runs back along that cone, and the band list ends it at once.)
  if (S.n_bands > 0) band_list();
  auto hierarchy = [&]() {
  if (S.root == REF_NONE || Q.done()) return;
  const char* nodes = reinterpret_cast<const char*>(S.nodes);
  // The reference finds the roots of a round shape in object space, where b^2 - 4ac cancels for a far
  // origin: with |origin|^2 = 1e14 a unit sphere's silhouette moves by a hundredth of its radius, and the
  // reference's hits then lie outside the shape's box.  The boxes are taken larger by that much for this
  // ray (nothing measurable for a ray that starts within 1e5 units of the scene).
  float widen = 0.0f;
  if (S.box_noise[0] > 0.0 || S.box_noise[1] > 0.0) {
    const double q = 2.0 * (ray.o.x * ray.o.x + ray.o.y * ray.o.y + ray.o.z * ray.o.z + S.box_noise[2]);
    if (q > 1e9) {  // within 2e4 units the rounding stays below the boxes' own pads on every scene tested
      const double w = 0.5 * ROUND_NOISE * q * S.box_noise[0] + sqrt(ROUND_NOISE * q) * S.box_noise[1];
      widen = w < 1e30 ? __double2float_ru(w) : 1e30f;  // NaN origins: the box tests see NaN either way
    }
  }
  const RayBox rb = make_raybox(ray, Query::BEHIND, widen);
  auto hi_f = [&]() { return __double2float_ru(Q.hi_d()); };  // the query's far bound, rounded up
  float qhi = hi_f();
  st.clear();
  // rays without a near-zero direction component read the child references in which skippable
  // gates are already replaced by their subtrees
  const uint32_t child_off = rb.tiny ? 96u : 112u;
  uint32_t cur = rb.tiny ? S.root : S.root_alt;
  // do all lanes that traverse pick the same near planes and the same child list?
  uint32_t un[3] = {0, 0, 0}, uchild = 0;
  bool same_signs = false;
  if constexpr (Stack::UNIFORM_NODES) {
    for (int a = 0; a < 3; a++) un[a] = (uint32_t)__builtin_amdgcn_readfirstlane(rb.off_near[a]);
    uchild = __builtin_amdgcn_readfirstlane(child_off);
    same_signs = __builtin_amdgcn_ballot_w64(rb.off_near[0] != un[0] || rb.off_near[1] != un[1] ||
                                             rb.off_near[2] != un[2] || child_off != uchild) == 0;
  }
  RT_TSTAMP(cnt, 0);
  while (cur != REF_NONE) {
    while ((cur >> 30) == KIND_INNER) {
      const uint32_t nd = cur << 7;  // byte offset of the node (inner kind bits are zero)
      if (COUNT) cnt.node_visits++;
      RT_WAVE_ITER(cnt, 0);
      float tn[4];
      uint32_t c[4];
      bool together = false;
      uint32_t ucur = 0;
      if constexpr (Stack::UNIFORM_NODES) {
        ucur = __builtin_amdgcn_readfirstlane(cur);
        together = same_signs && __builtin_amdgcn_ballot_w64(cur != ucur) == 0;
      }
      if (together) {
        // every lane still in the loop is at this node and picks the same near / far planes
        const char* np = nodes + ((size_t)ucur << 7);
        F4 nx = load_f4_uniform(np, un[0]), ny = load_f4_uniform(np, un[1]), nz = load_f4_uniform(np, un[2]);
        F4 fx = load_f4_uniform(np, un[0] ^ 16u), fy = load_f4_uniform(np, un[1] ^ 16u), fz = load_f4_uniform(np, un[2] ^ 16u);
        u32x4 ch = *(const __attribute__((address_space(4))) u32x4*)(np + uchild);
        c[0] = ch.x; c[1] = ch.y; c[2] = ch.z; c[3] = ch.w;
        slab4(rb, qhi, nx, ny, nz, fx, fy, fz, tn, c);
      } else {
        F4 nx = load_f4(nodes, nd + rb.off_near[0]), ny = load_f4(nodes, nd + rb.off_near[1]),
           nz = load_f4(nodes, nd + rb.off_near[2]);
        F4 fx = load_f4(nodes, nd + (rb.off_near[0] ^ 16u)), fy = load_f4(nodes, nd + (rb.off_near[1] ^ 16u)),
           fz = load_f4(nodes, nd + (rb.off_near[2] ^ 16u));
        uint4 ch = *reinterpret_cast<const uint4*>(nodes + nd + child_off);
        c[0] = ch.x; c[1] = ch.y; c[2] = ch.z; c[3] = ch.w;
        slab4(rb, qhi, nx, ny, nz, fx, fy, fz, tn, c);
      }
      // nearest hit child stays in a register, the other hit children go on the stack
      bool lt = tn[1] < tn[0];
      float best = lt ? tn[1] : tn[0];
      uint32_t bc = lt ? c[1] : c[0], o1 = lt ? c[0] : c[1];
      lt = tn[2] < best;
      best = lt ? tn[2] : best;
      uint32_t o2 = lt ? bc : c[2];
      bc = lt ? c[2] : bc;
      lt = tn[3] < best;
      uint32_t o3 = lt ? bc : c[3];
      bc = lt ? c[3] : bc;
      st.push_if(o1, o1 != REF_NONE);
      st.push_if(o2, o2 != REF_NONE);
      st.push_if(o3, o3 != REF_NONE);
      cur = bc != REF_NONE ? bc : st.pop_or_none();
    }
    RT_TSTAMP(cnt, 1);
    RT_WAVE_ITER(cnt, 3);  // rounds of the outer loop
    if (cur == REF_NONE) break;
    const uint32_t kind = cur >> 30;
    if (kind == KIND_GATE) {
      const Gate& g = S.gates[cur & 0x3FFFFFFFu];
      if (g.csg) {
        if constexpr (Stack::WITH_CSG) eval_csg<Query, Stack, Stack::THREADS>(S, g.csg - 1u, ray, Q, st, false);
        if (Q.done()) break;
        qhi = hi_f();
        cur = st.pop_or_none();
        RT_TSTAMP(cnt, 3);
        continue;
      }
      bool pass;
      if (g.skippable && !rb.tiny) {
        pass = true;
      } else {
        if (COUNT) cnt.gate_tests++;
        pass = gate_passes<Stack::WITH_CSG>(S, g, ray);
      }
      cur = pass ? g.root : st.pop_or_none();
      RT_TSTAMP(cnt, 3);
      continue;
    }
    const uint32_t first = cur & MAX_RUN_FIRST, count = ((cur >> 27) & 7u) + 1u;
    if (kind == KIND_TRI) {
      int32_t local_of = -1;  // the transform `local` was made with: the triangles of a mesh share one
      Ray local = ray;
      // Kernels for large meshes hold the next record of the run in registers while this one is tested.
      TriIsect ahead;
      if constexpr (Stack::TRI_PREFETCH) ahead = S.tris[first];
      for (uint32_t i = 0; i < count; i++) {
        TriIsect held;
        if constexpr (Stack::TRI_PREFETCH) {
          held = ahead;
          if (i + 1 < count) ahead = S.tris[first + i + 1];
        }
        const TriIsect& tr = Stack::TRI_PREFETCH ? held : S.tris[first + i];
        if (COUNT) cnt.tri_tests++;
        RT_WAVE_ITER(cnt, 1);
        double t, u, v;
        bool hit;
        if (tr.xform >= 0) {
          if (tr.xform != local_of) {  // the same ray, the same matrix: the same object-space ray, made once
            local = tri_ray<Stack::WITH_CSG>(S, tr, ray);
            local_of = tr.xform;
          }
          hit = intersect_triangle(tr, local, t, u, v);
        } else {
          hit = intersect_triangle(tr, ray, t, u, v);
        }
        if (hit) Q.leaf_tri(tr.leaf, t, u, v);
      }
      RT_TSTAMP(cnt, 2);
    } else {
      for (uint32_t i = 0; i < count; i++) {
        if (COUNT) cnt.quad_tests++;
        RT_WAVE_ITER(cnt, 2);
        test_quad<Query, Stack::WITH_CSG>(S, S.quads[first + i], ray, Q, false);
      }
      RT_TSTAMP(cnt, 3);
    }
    if (Q.done()) return;
    qhi = hi_f();
    cur = st.pop_or_none();
  }
  };
  hierarchy();
}

// ---------------------------------------------------------------------------
// ray records: one generation = struct of arrays in HBM
// ---------------------------------------------------------------------------
struct GenView {
  double *ox, *oy, *oz, *dx, *dy, *dz;              // ray (generations >= 1 and caller-supplied rays)
  double *cr, *cg, *cb;                             // surface colour, then the ray's final colour
  double *reflective, *transparency, *reflectance;  // mix factors of this hit
  double* t;                                        // hit distance
  uint32_t *child_reflect, *child_refract;          // indices into the next generation or REF_NONE
  uint32_t* flags;                                  // bit0: hit, bit1: Schlick mix
  int32_t* leaf;                                    // hit leaf or -1
};

DEV Ray load_ray(const GenView& g, uint32_t i) {
  return Ray{v3(g.ox[i], g.oy[i], g.oz[i]), v3(g.dx[i], g.dy[i], g.dz[i])};
}

struct CameraDev {
  uint32_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double inv[12];
};

struct ImageOut {
  uint8_t* rgb8;
  double* rgb64;
  int32_t* hit_leaf;
  uint32_t compact;  // rgb8 holds this part's pixels in ray order (3 bytes at 3 i) instead of at y * width + x
};

constexpr int TILE = 32;  // multi-GPU partition granule, pixels per side
constexpr int TILE_PIXELS = TILE * TILE;
#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif
constexpr int BLOCK = RT_BLOCK;  // threads per workgroup of every kernel
// Thread index within the workgroup.  With one wave per workgroup it is the lane id, which costs two
// instructions to recompute wherever it is needed instead of occupying a register for the whole kernel.
DEV uint32_t thread_in_block() { return BLOCK == 64 ? (uint32_t)__lane_id() : threadIdx.x; }

// Device statistics block (unsigned long long): [0..3] closest-hit traversal work,
// [4..7] shadow traversal work, then three ray counters spread over 64 slots each
// (shaded hits = shadow rays, reflection rays, refraction rays) so that the per-wave
// adds do not all land on one address.
constexpr int STAT_SLOTS = 64;
constexpr int QUEUE_SEGMENTS = 64;
constexpr int COUNT_WORDS = (MAX_GENERATIONS + 1) * QUEUE_SEGMENTS;  // queue counters of one frame
constexpr int STAT_RAYS = 8;
#ifdef RT_STAMPS
// Diagnostic build: per-wave clock stamps between the sections of k_generation, summed per section.
constexpr int STAMP_WORDS = 24;  // 0..4 sections of k_generation, 8..11 / 16..19 closest, 12..15 / 20..23 shadow
#else
constexpr int STAMP_WORDS = 0;
#endif
constexpr int STAT_WORDS = STAT_RAYS + 3 * STAT_SLOTS + STAMP_WORDS;
#ifdef RT_STAMPS
#define RT_STAMP(k) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if ((thread_in_block() & 63) == 0) atomicAdd(&stats[STAT_RAYS + 3 * STAT_SLOTS + (k)], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define RT_STAMP(k) do { } while (0)
#endif

// Ray index -> pixel: tile k of this part, then 8x8 pixel blocks per wave so the
// 64 rays of a wave stay close in screen space.
DEV bool ray_pixel(const uint32_t* tiles, uint32_t tiles_x, const CameraDev& cam, uint32_t i, uint32_t& x,
                   uint32_t& y) {
  uint32_t tile = tiles ? tiles[i / TILE_PIXELS] : i / TILE_PIXELS;  // no table: the part owns every tile, in order
  uint32_t in_tile = i % TILE_PIXELS;
  uint32_t wave = in_tile >> 6, lane = in_tile & 63u;
  x = (tile % tiles_x) * TILE + (wave & 3u) * 8u + (lane & 7u);
  y = (tile / tiles_x) * TILE + (wave >> 2) * 8u + (lane >> 3);
  return x < cam.hsize && y < cam.vsize;
}

// Camera::ray_for_pixel (src/camera.rs:76-95) with the inverse precomputed.
DEV Ray camera_ray(const CameraDev& cam, uint32_t px, uint32_t py) {
  double xoffset = ((double)px + 0.5) * cam.pixel_size;
  double yoffset = ((double)py + 0.5) * cam.pixel_size;
  double world_x = cam.half_width - xoffset;
  double world_y = cam.half_height - yoffset;
  V3 pixel = mul_point(cam.inv, v3(world_x, world_y, -1.0));
  V3 origin = mul_point(cam.inv, v3(0.0, 0.0, 0.0));
  V3 direction = normalize(pixel - origin);
  return Ray{origin, direction};
}

// Rays supplied by the caller (rt_color_at and friends): Ray4D::new keeps the
// three components as given.
__global__ void __launch_bounds__(BLOCK) k_load_rays(GenView g, uint32_t n, const double* rays6) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double* r = rays6 + (size_t)i * 6;
  g.ox[i] = r[0]; g.oy[i] = r[1]; g.oz[i] = r[2];
  g.dx[i] = r[3]; g.dy[i] = r[4]; g.dz[i] = r[5];
}

DEV void flush_counters(const Counters& c, unsigned long long* out) {
  // one atomic per wave and counter
  unsigned long long v[4] = {c.node_visits, c.tri_tests, c.quad_tests, c.gate_tests};
  for (int k = 0; k < 4; k++) {
    unsigned long long s = v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((thread_in_block() & 63) == 0 && s) atomicAdd(&out[k], s);
  }
}

// ---------------------------------------------------------------------------
// shading
// ---------------------------------------------------------------------------
DEV double rem_euclid2(double x) {  // f64::rem_euclid(2.0)
  double r = fmod(x, 2.0);
  return r < 0.0 ? r + 2.0 : r;
}

// Pattern::pattern_at (src/pattern.rs:97-189).  Blend operands are evaluated at
// the same point with no transform of their own (src/pattern.rs:185-186).
template <int DEPTH>
DEV void pattern_at(const SceneDev& S, int idx, V3 p, double& r, double& g, double& b) {
  const PatternDev& pt = S.patterns[idx];
  bool primary = true;
  switch (pt.type) {
    case RT_PAT_NULL: r = g = b = 0.0; return;
    case RT_PAT_POINT: r = p.x; g = p.y; b = p.z; return;
    case RT_PAT_IDENTITY: r = pt.a[0]; g = pt.a[1]; b = pt.a[2]; return;
    case RT_PAT_STRIPE: primary = feq(rem_euclid2(floor(p.x)), 0.0); break;
    case RT_PAT_RING: primary = feq(rem_euclid2(floor(sqrt(p.x * p.x + p.z * p.z))), 0.0); break;
    case RT_PAT_CHECKER: primary = feq(rem_euclid2(floor(p.x) + floor(p.y) + floor(p.z)), 0.0); break;
    case RT_PAT_GRADIENT: {
      double fraction = p.x - floor(p.x);
      r = pt.a[0] + (pt.b[0] - pt.a[0]) * fraction;
      g = pt.a[1] + (pt.b[1] - pt.a[1]) * fraction;
      b = pt.a[2] + (pt.b[2] - pt.a[2]) * fraction;
      return;
    }
    case RT_PAT_BLEND: {
      if constexpr (DEPTH > 0) {
        double r1, g1, b1, r2, g2, b2;
        pattern_at<DEPTH - 1>(S, pt.first, p, r1, g1, b1);
        pattern_at<DEPTH - 1>(S, pt.second, p, r2, g2, b2);
        r = (r1 + r2) / 2.0; g = (g1 + g2) / 2.0; b = (b1 + b2) / 2.0;  // src/color.rs:169-175
      } else {
        r = g = b = 0.0;
      }
      return;
    }
    default: r = g = b = 0.0; return;
  }
  r = primary ? pt.a[0] : pt.b[0];
  g = primary ? pt.a[1] : pt.b[1];
  b = primary ? pt.a[2] : pt.b[2];
}
constexpr int MAX_BLEND_DEPTH = 4;

// local_normal_at (src/shape.rs:805-829, per-type routines 880-1173, 1263-1332)
DEV V3 local_normal(const SceneDev& S, const LeafShade& L, V3 at, double u, double v) {
  switch (L.type) {
    case RT_PLANE: return v3(L.plane_normal[0], L.plane_normal[1], L.plane_normal[2]);
    case RT_CUBE: {
      double xa = fabs(at.x), ya = fabs(at.y), za = fabs(at.z);
      double max_component = fmax(fmax(xa, ya), za);
      if (max_component == xa) return v3(at.x, 0.0, 0.0);
      if (max_component == ya) return v3(0.0, at.y, 0.0);
      return v3(0.0, 0.0, at.z);
    }
    case RT_CYLINDER:
    case RT_CONE: {
      double dist = at.x * at.x + at.z * at.z;
      if (dist < 1.0 && at.y >= L.maximum - FEQ_EPSILON) return v3(0.0, 1.0, 0.0);
      if (dist < 1.0 && at.y <= L.minimum + FEQ_EPSILON) return v3(0.0, -1.0, 0.0);
      if (L.type == RT_CYLINDER) return v3(at.x, 0.0, at.z);
      double y = sqrt(dist);
      if (at.y > 0.0) y = -y;
      return v3(at.x, y, at.z);
    }
    case RT_TRIANGLE: {
      const TriShade& ts = S.tri_shade[L.tri];
      return v3(ts.normal[0], ts.normal[1], ts.normal[2]);
    }
    case RT_SMOOTH_TRIANGLE: {
      const TriShade& ts = S.tri_shade[L.tri];
      double w = 1.0 - u - v;  // n2 * u + n3 * v + n1 * (1 - u - v)
      return v3(ts.n2[0] * u + ts.n3[0] * v + ts.n1[0] * w, ts.n2[1] * u + ts.n3[1] * v + ts.n1[1] * w,
                ts.n2[2] * u + ts.n3[2] * v + ts.n1[2] * w);
    }
    default: return at;  // sphere, empty: the point with w = 0
  }
}

// Appends `mine` items per lane to a queue: one atomic per wave, prefix by ballot.
DEV uint32_t wave_append(uint32_t* counter, bool mine) {
  unsigned long long mask = __ballot(mine);
  uint32_t lane = thread_in_block() & 63u;
  uint32_t total = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  if (lane == 0 && total) base = atomicAdd(counter, total);
  base = __shfl(base, 0);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// The rays a generation spawns are queued in QUEUE_SEGMENTS segments, each with its own counter and
// its own stretch of `segcap` record slots (segment s holds slots [s * segcap, (s + 1) * segcap)): a
// wave appends to segment (wave number % QUEUE_SEGMENTS).  One counter for all waves serialises the
// returning atomics at one L2 address, about 10 ns each, which for a generation of half a million rays
// (15 000 atomics) is more than the tracing itself takes.
//
// The next generation runs one thread per queued ray: thread i finds its segment and record slot from
// the running sum of the segment counters.  Every lane of the wave must call this (it shuffles).
DEV bool queue_slot(const uint32_t* counts, uint32_t segcap, uint32_t i, uint32_t& slot) {
  static_assert(QUEUE_SEGMENTS == 64, "one segment counter per lane of a wave");
  const uint32_t lane = thread_in_block() & 63u;
  const uint32_t c = min(counts[lane], segcap);
  uint32_t p = c;  // inclusive running sum over the segments
  for (uint32_t d = 1; d < 64; d <<= 1) {
    uint32_t v = __shfl_up(p, d, 64);
    if (lane >= d) p += v;
  }
  uint32_t seg = 0;  // the number of segments that end at or before ray i, at most 63
  for (uint32_t step = 32; step > 0; step >>= 1) {
    uint32_t v = __shfl(p, (int)(seg + step - 1), 64);
    if (v <= i) seg += step;
  }
  const uint32_t end = __shfl(p, (int)seg, 64), len = __shfl(c, (int)seg, 64);
  slot = seg * segcap + (i - (end - len));
  return i < end;
}

// reflect_dot_eye.powf(shininess) (src/light.rs:168).  A whole-number exponent up to 1024 (the
// default material's is 200) is raised by repeated squaring: about ten multiplications, each
// correctly rounded, so the result stays within a few ulps of the libm value; anything else goes
// through pow().
DEV double specular_power(double x, double shininess) {
  if (shininess >= 0.0 && shininess <= 1024.0 && shininess == floor(shininess)) {
    uint32_t n = (uint32_t)shininess;
    double result = 1.0, base = x;
    while (n) {
      if (n & 1u) result *= base;
      base *= base;
      n >>= 1;
    }
    return result;
  }
  return pow(x, shininess);
}

// Canvas::save quantisation (src/canvas.rs:60-62): ceil(clamp(c * 255, 0, 255)); NaN -> 0
DEV uint8_t quantise(double c) {
  double v = c * 255.0;
  if (v < 0.0) v = 0.0;
  else if (v > 255.0) v = 255.0;
  v = ceil(v);
  return (v == v) ? (uint8_t)v : (uint8_t)0;
}

// Canvas::write_pixel for the 8x8 pixel block of a wave; all 64 lanes call it together.  The bytes
// of one block row (8 pixels, 24 bytes) are gathered across lanes and stored as six dwords, so the
// framebuffer leaves the chip as full dwords instead of three single-byte stores per pixel.  In the
// compact layout the 64 pixels of the wave are contiguous (192 bytes, 48 dwords).
DEV void write_pixel(const ImageOut& img, uint32_t hsize, uint32_t x, uint32_t y, uint32_t i, bool active, double r,
                     double g, double b, int leaf) {
  const size_t p = (size_t)y * hsize + x;
  if (active) {
    if (img.rgb64) {
      img.rgb64[p * 3 + 0] = r; img.rgb64[p * 3 + 1] = g; img.rgb64[p * 3 + 2] = b;
    }
    if (img.hit_leaf) img.hit_leaf[p] = leaf;
  }
  if (!img.rgb8) return;
  const uint32_t v = (uint32_t)quantise(r) | ((uint32_t)quantise(g) << 8) | ((uint32_t)quantise(b) << 16);
  const uint32_t lane = thread_in_block() & 63u;
  if (img.compact) {
    // dword `lane` (0..47) of the wave's 192 bytes holds bytes 4 lane .. 4 lane + 3: pixels p0 and p0 + 1
    const uint32_t p0 = (4u * lane) / 3u, shift = 8u * (4u * lane - 3u * p0);
    const uint32_t lo = __shfl(v, (int)(p0 < 63u ? p0 : 63u));
    const uint32_t hi = __shfl(v, (int)(p0 + 1u < 63u ? p0 + 1u : 63u));
    if (((uintptr_t)img.rgb8 & 3u) == 0u) {  // wave-uniform; i - lane is a multiple of 64
      if (lane < 48u) {
        const unsigned long long cat = (unsigned long long)lo | ((unsigned long long)hi << 24);
        reinterpret_cast<uint32_t*>(img.rgb8 + (size_t)(i - lane) * 3u)[lane] = (uint32_t)(cat >> shift);
      }
    } else {
      img.rgb8[(size_t)i * 3 + 0] = (uint8_t)v; img.rgb8[(size_t)i * 3 + 1] = (uint8_t)(v >> 8);
      img.rgb8[(size_t)i * 3 + 2] = (uint8_t)(v >> 16);
    }
    return;
  }
  const uint32_t col = lane & 7u, row_base = lane & ~7u;
  // dword `col` (0..5) of the row holds bytes 4 col .. 4 col + 3, which belong to pixels p0 and p0 + 1
  const uint32_t p0 = (4u * col) / 3u, shift = 8u * (4u * col - 3u * p0);
  const uint32_t lo = __shfl(v, (int)(row_base + (p0 < 7u ? p0 : 7u)));
  const uint32_t hi = __shfl(v, (int)(row_base + (p0 + 1u < 7u ? p0 + 1u : 7u)));
  const uint32_t x0 = x - col;
  const bool whole_rows = (hsize & 3u) == 0u && x0 + 8u <= hsize && ((uintptr_t)img.rgb8 & 3u) == 0u;  // wave-uniform
  if (whole_rows) {
    if (active && col < 6u) {
      const unsigned long long cat = (unsigned long long)lo | ((unsigned long long)hi << 24);
      uint32_t* dst = reinterpret_cast<uint32_t*>(img.rgb8 + ((size_t)y * hsize + x0) * 3u) + col;
      *dst = (uint32_t)(cat >> shift);
    }
  } else if (active) {
    img.rgb8[p * 3 + 0] = (uint8_t)v; img.rgb8[p * 3 + 1] = (uint8_t)(v >> 8); img.rgb8[p * 3 + 2] = (uint8_t)(v >> 16);
  }
}

// One generation of rays, start to finish, one ray per lane:
//   ray (camera or queue) -> closest hit (World::intersect + Intersections::hit)
//   -> IntersectionComputation::new (src/intersect.rs:152-184) -> lighting terms
//   (src/light.rs:125-174) -> reflection / refraction rays appended to the next
//   generation (src/world.rs:142-215) -> shadow query (src/world.rs:123-137)
//   -> surface colour.
// MODE 0: camera rays of a scene that never spawns rays: the pixel is final, written straight
//         to the image.  MODE 1: camera rays with ray records.  MODE 2: rays read from g.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_XCD_RUN
#define RT_XCD_RUN 16  // pixel blocks (waves) per run on one XCD: 16 = one 32x32 tile; 1 = plain order
#endif
#ifndef RT_MIN_WAVES_SPAWN
#define RT_MIN_WAVES_SPAWN 3
#endif
#ifndef RT_SHADOW_ALWAYS
#define RT_SHADOW_ALWAYS 0  // 1: traverse every shadow query, also those whose answer cannot reach the colour (A/B builds)
#endif
template <int STACK, bool COUNT, int MODE, bool CSG, bool UNIFORM, bool TRIPF = false>
// Waves per SIMD a kernel is compiled for.  The ray-spawning modes need about 170 registers: at three waves
// nothing much spills, and small scenes, whose records come from L1, run fastest that way (glass 4K: 0.67 ms
// against 0.88 at four waves, 0.76 at two).  The kernels for large hierarchies wait on L2 and beyond, and a
// fourth wave hides more of that than the spilled registers cost (100k primitives 8K: 11.1 -> 10.7 ms).
__global__ void __launch_bounds__(BLOCK, MODE == 0 || UNIFORM ? RT_MIN_WAVES : RT_MIN_WAVES_SPAWN)
    k_generation(SceneDev S, GenView g, uint32_t n, GenView next, uint32_t next_segcap, int remaining,
                 uint32_t* next_counts, unsigned long long* stats, CameraDev cam, const uint32_t* tiles,
                 uint32_t tiles_x, ImageOut img, const uint32_t* queue_counts, uint32_t segcap) {
  __shared__ uint32_t lds[RT_STACK_LDS(STACK * BLOCK)];
  LdsStack<STACK, BLOCK, CSG, UNIFORM, TRIPF> st;
  extern __shared__ uint32_t csg_lds[];  // SceneDev::csg_hits * 3 * BLOCK words when the scene has CSG operators, else nothing
  st.init(lds, CSG ? csg_lds : nullptr);
#ifdef RT_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef RT_WAVE_CLOCK_IMAGE  // diagnostic build: the frame shows how long each wave ran instead of the colours
  const unsigned long long wave_begin_ = __builtin_amdgcn_s_memtime();
#endif
  // i: the thread's ray; rec: where the ray's record lives.  Camera rays and rays handed in by the
  // caller are dense (rec == i, n of them); spawned rays are found through the segment counters the
  // generation before left in queue_counts, and n only bounds the grid.
  // Camera rays: which 8x8 pixel block a wave takes.  Workgroups are dealt round-robin over the eight
  // XCDs (blocks b and b + 8 share one, each XCD with its own L2), so consecutive block numbers
  // would spread the sixteen blocks of a tile over all eight L2s: every L2 would hold the nodes and
  // triangles behind every part of the image, and every 128-byte line of the frame would leave
  // several L2s as partial writes.  Instead the blocks that share an XCD take runs of XCD_RUN
  // consecutive pixel blocks: one 32 x 32 tile.  Measured on the 1M-triangle mesh at 4K: 2.96 ms in
  // plain order, 2.74 ms with runs of 16, 2.77 with 64, 3.14 with 256 (longer runs leave the XCDs
  // unevenly loaded: the teapot frame gets 5 % slower at 64 and is unchanged at 16).
  uint32_t wave_index = blockIdx.x;
  if (MODE != 2) {
    constexpr uint32_t XCD_RUN = RT_XCD_RUN, XCDS = 8;
    const uint32_t full = gridDim.x - gridDim.x % (XCDS * XCD_RUN);  // the rest keeps its order
    if (XCD_RUN > 1 && wave_index < full) {
      const uint32_t xcd = wave_index % XCDS, k = wave_index / XCDS;
      wave_index = ((k / XCD_RUN) * XCDS + xcd) * XCD_RUN + k % XCD_RUN;
    }
  }
  const uint32_t i = wave_index * BLOCK + thread_in_block();
  uint32_t rec = i;
  bool active = i < n;
  if (MODE == 2 && queue_counts) active = queue_slot(queue_counts, segcap, i, rec) && active;
  const bool has_record = active;
  uint32_t px = 0, py = 0;
  Ray ray{v3(0, 0, 0), v3(0, 0, 1)};
  if (MODE != 2) {
    active = active && ray_pixel(tiles, tiles_x, cam, i, px, py);
    if (active) ray = camera_ray(cam, px, py);
  } else if (active) {
    ray = load_ray(g, rec);
  }

  Counters ct{}, cs{};
#ifdef RT_STAMPS
  ct.sink = stats + STAT_RAYS + 3 * STAT_SLOTS + 8;   // time sums at sink[0..3], wave iterations at sink[4..7]:
  cs.sink = stats + STAT_RAYS + 3 * STAT_SLOTS + 16;  // two blocks of eight
#endif
  ClosestQuery Q;
  Q.init();
  RT_STAMP(0);  // ray generation
  if (active) traverse<ClosestQuery, LdsStack<STACK, BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, ray, Q, st, ct);
  RT_STAMP(1);  // closest-hit traversal
  const bool hit = active && Q.best_leaf != 0x7FFFFFFF;
  const int leaf = hit ? Q.best_leaf : -1;

  bool want_reflect = false, want_refract = false;
  Ray reflect_ray{v3(0, 0, 0), v3(0, 0, 0)}, refract_ray{v3(0, 0, 0), v3(0, 0, 0)};
  V3 over_point = v3(0, 0, 0);
  double ar = 0, ag = 0, ab = 0, lr = 0, lg = 0, lb = 0;
  double k_reflective = 0, k_transparency = 0, reflectance = 0;
  uint32_t flags = hit ? 1u : 0u;
  if (hit) {
    const LeafShade L = S.leaves[leaf];
    const MaterialDev m = S.materials[L.material];
    const XformDev& X = S.xforms[L.xform];
    const double t = Q.best_t;
    V3 point = ray.o + (ray.d * t);  // Ray4D::position
    V3 eyev = -ray.d;
    // normal_at (src/shape.rs:1478-1484)
    V3 local_point = mul_point(X.full_inv, point);
    V3 ln = local_normal(S, L, local_point, Q.u, Q.v);
    const double* fi = X.full_inv;  // transposition() * normal, w dropped, normalize
    V3 normalv = normalize(v3(fi[0] * ln.x + fi[4] * ln.y + fi[8] * ln.z, fi[1] * ln.x + fi[5] * ln.y + fi[9] * ln.z,
                              fi[2] * ln.x + fi[6] * ln.y + fi[10] * ln.z));
    if (dot(normalv, eyev) < 0.0) normalv = -normalv;
    over_point = point + normalv * FEQ_EPSILON;

    // lighting(): evaluated at over_point, pattern included (src/world.rs:223-227)
    double cr, cg, cb;
    if (m.pattern >= 0) {
      V3 object_point = mul_point(X.full_inv, over_point);
      V3 pattern_point = mul_point(S.patterns[m.pattern].inv, object_point);
      pattern_at<MAX_BLEND_DEPTH>(S, m.pattern, pattern_point, cr, cg, cb);
    } else {
      cr = m.color[0]; cg = m.color[1]; cb = m.color[2];
    }
    double er = cr * S.light_intensity[0], eg = cg * S.light_intensity[1], eb = cb * S.light_intensity[2];
    V3 lightpos = v3(S.light_position[0], S.light_position[1], S.light_position[2]);
    V3 lightv = normalize(lightpos - over_point);
    ar = er * m.ambient; ag = eg * m.ambient; ab = eb * m.ambient;
    double dr = 0.0, dg = 0.0, db = 0.0, sr = 0.0, sg = 0.0, sb = 0.0;
    double light_dot_normal = dot(lightv, normalv);
    if (!(light_dot_normal < 0.0)) {
      dr = er * m.diffuse * light_dot_normal;
      dg = eg * m.diffuse * light_dot_normal;
      db = eb * m.diffuse * light_dot_normal;
      V3 rv = reflect(-lightv, normalv);
      double reflect_dot_eye = dot(rv, eyev);
      if (!(reflect_dot_eye <= 0.0)) {
        double factor = specular_power(reflect_dot_eye, m.shininess);
        sr = S.light_intensity[0] * m.specular * factor;
        sg = S.light_intensity[1] * m.specular * factor;
        sb = S.light_intensity[2] * m.specular * factor;
      }
    }
    lr = ar + dr + sr; lg = ag + dg + sg; lb = ab + db + sb;

    if (MODE != 0) {
      k_reflective = m.reflective;
      k_transparency = m.transparency;
      // reflected_color (src/world.rs:142-159)
      if (remaining != 0 && (m.flags & 1)) {
        want_reflect = true;
        reflect_ray = Ray{over_point, reflect(ray.d, normalv)};
      }
      double n1 = 1.0, n2 = 1.0;
      if (m.flags & 8) {  // n1 / n2 are consumed: refraction or Schlick (src/intersect.rs:191-234)
        ContainerQuery C;
        C.init(leaf);
        traverse<ContainerQuery, LdsStack<STACK, BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, ray, C, st, ct);
        double ri_hit = m.refractive_index;
        double ri_other = C.other_leaf >= 0 ? S.materials[S.leaves[C.other_leaf].material].refractive_index : 1.0;
        if (C.hit_inside) {
          bool hit_last = C.other_leaf < 0 || C.hit_key > C.other_key ||
                          (C.hit_key == C.other_key && C.hit_leaf > C.other_leaf);
          n1 = hit_last ? ri_hit : ri_other;
          n2 = ri_other;
        } else {
          n1 = ri_other;
          n2 = ri_hit;
        }
      }
      // refracted_color (src/world.rs:181-215)
      if (remaining != 0 && (m.flags & 2)) {
        double n_ratio = n1 / n2;
        double cos_i = dot(eyev, normalv);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
          double cos_t = sqrt(1.0 - sin2_t);
          V3 under_point = point - normalv * FEQ_EPSILON;
          V3 direction = normalv * (n_ratio * cos_i - cos_t) - eyev * n_ratio;
          want_refract = true;
          refract_ray = Ray{under_point, direction};
        }
      }
      if (m.flags & 4) {  // schlick (src/intersect.rs:240-260)
        double cosv = dot(eyev, normalv);
        bool total = false;
        if (n1 > n2) {
          double nn = n1 / n2;
          double sin2_t = (nn * nn) * (1.0 - cosv * cosv);
          if (sin2_t > 1.0) total = true;
          else cosv = sqrt(1.0 - sin2_t);
        }
        if (total) {
          reflectance = 1.0;
        } else {
          double q = (n1 - n2) / (n1 + n2);
          double r0 = q * q;
          double x = 1.0 - cosv;
          double x2 = x * x;
          reflectance = r0 + (1.0 - r0) * (x * (x2 * x2));  // powi(5)
        }
        flags = 3u;
      }
    }
  }

  bool keep_record = false;
  if (MODE != 0) {
    // next generation: one atomic per wave that has something to append
    uint32_t child_reflect = REF_NONE, child_refract = REF_NONE;
    // this wave's segment of the next generation's queue
    const uint32_t seg = (blockIdx.x * (BLOCK / 64) + (thread_in_block() >> 6)) % QUEUE_SEGMENTS;
    uint32_t ri = wave_append(next_counts + seg, want_reflect);
    if (want_reflect && ri < next_segcap) {
      ri += seg * next_segcap;
      next.ox[ri] = reflect_ray.o.x; next.oy[ri] = reflect_ray.o.y; next.oz[ri] = reflect_ray.o.z;
      next.dx[ri] = reflect_ray.d.x; next.dy[ri] = reflect_ray.d.y; next.dz[ri] = reflect_ray.d.z;
      child_reflect = ri;
    }
    uint32_t ti = wave_append(next_counts + seg, want_refract);
    if (want_refract && ti < next_segcap) {
      ti += seg * next_segcap;
      next.ox[ti] = refract_ray.o.x; next.oy[ti] = refract_ray.o.y; next.oz[ti] = refract_ray.o.z;
      next.dx[ti] = refract_ray.d.x; next.dy[ti] = refract_ray.d.y; next.dz[ti] = refract_ray.d.z;
      child_refract = ti;
    }
    // Camera rays write their pixel below; only those with children are looked at again
    // (k_resolve_pixels), so the others leave nothing but their flags.
    if (child_reflect != REF_NONE || child_refract != REF_NONE) flags |= 4u;
    keep_record = has_record && (MODE == 2 || (flags & 4u));
    if (has_record) g.flags[rec] = flags;
    if (keep_record) {
      g.leaf[rec] = leaf;
      g.child_reflect[rec] = child_reflect;
      g.child_refract[rec] = child_refract;
      if (hit) {
        g.t[rec] = Q.best_t;
        g.reflective[rec] = k_reflective;
        g.transparency[rec] = k_transparency;
        g.reflectance[rec] = reflectance;
      }
    }
  }

  // World::is_shadowed (src/world.rs:123-137), then the in_shadow branch of lighting
  double out_r = 0.0, out_g = 0.0, out_b = 0.0;  // World::color_at: no hit => black
  RT_STAMP(2);  // shading, spawning
  if (hit) {
    V3 lightpos = v3(S.light_position[0], S.light_position[1], S.light_position[2]);
    V3 v = lightpos - over_point;
    ShadowQuery SQ;
    SQ.distance = magnitude(v);
    SQ.found = false;
    Ray shadow_ray{over_point, normalize(v)};
    // lighting() returns `ambient` in shadow and `ambient + diffuse + specular` otherwise
    // (src/light.rs:135-173).  Where the two are the same bits -- the light behind the surface, or a
    // material without diffuse and specular terms -- the answer of is_shadowed cannot reach the
    // colour, and the query (still counted as a ray) is not traversed.
    const bool decides = __double_as_longlong(ar) != __double_as_longlong(lr) ||
                         __double_as_longlong(ag) != __double_as_longlong(lg) ||
                         __double_as_longlong(ab) != __double_as_longlong(lb);
    if (RT_SHADOW_ALWAYS || decides)
      traverse<ShadowQuery, LdsStack<STACK, BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, shadow_ray, SQ, st, cs);
    out_r = SQ.found ? ar : lr;
    out_g = SQ.found ? ag : lg;
    out_b = SQ.found ? ab : lb;
  }
  RT_STAMP(3);  // shadow traversal
#ifdef RT_WAVE_CLOCK_IMAGE
  {
#if RT_WAVE_CLOCK_IMAGE == 2  // wave-level iterations instead: node visits in the low 12 bits, triangle tests above
    unsigned nodes_ = ct.wi[0] + cs.wi[0], tris_ = ct.wi[1] + cs.wi[1];
    for (int off = 32; off > 0; off >>= 1) {
      nodes_ += __shfl_xor(nodes_, off);
      tris_ += __shfl_xor(tris_, off);
    }
    const unsigned long long ticks = (nodes_ < 4095u ? nodes_ : 4095u) | ((tris_ < 4095u ? tris_ : 4095u) << 12);
#else
    const unsigned long long ticks = __builtin_amdgcn_s_memtime() - wave_begin_;  // 24 bits into the three bytes
#endif
    const unsigned b0 = ticks & 255u, b1 = (ticks >> 8) & 255u, b2 = (ticks >> 16) & 255u;
    out_r = b0 ? (b0 - 0.5) / 255.0 : 0.0;
    out_g = b1 ? (b1 - 0.5) / 255.0 : 0.0;
    out_b = b2 ? (b2 - 0.5) / 255.0 : 0.0;
  }
#endif
  if (MODE != 2) write_pixel(img, cam.hsize, px, py, i, active, out_r, out_g, out_b, leaf);
  if (MODE != 0 && keep_record) {
    g.cr[rec] = out_r; g.cg[rec] = out_g; g.cb[rec] = out_b;
  }

  // ray statistics, one add per wave and kind, spread over STAT_SLOTS addresses
  unsigned long long hm = __ballot(hit), rm = __ballot(want_reflect), fm = __ballot(want_refract);
  if ((thread_in_block() & 63) == 0) {
    uint32_t slot = (blockIdx.x * (BLOCK / 64) + (thread_in_block() >> 6)) % STAT_SLOTS;
    if (hm) atomicAdd(&stats[STAT_RAYS + 0 * STAT_SLOTS + slot], (unsigned long long)__popcll(hm));
    if (rm) atomicAdd(&stats[STAT_RAYS + 1 * STAT_SLOTS + slot], (unsigned long long)__popcll(rm));
    if (fm) atomicAdd(&stats[STAT_RAYS + 2 * STAT_SLOTS + slot], (unsigned long long)__popcll(fm));
  }
  if (COUNT) {
    flush_counters(ct, stats);
    flush_counters(cs, stats + 4);
  }
  RT_STAMP(4);  // output
}

// shade_hit's final expression (src/world.rs:236-245) for record i, its children resolved already.
// A record without children keeps its surface colour (returns false, nothing written).
DEV bool resolve_record(const GenView& g, const GenView& next, uint32_t i, double* out_r = nullptr,
                        double* out_g = nullptr, double* out_b = nullptr) {
  const uint32_t flags = g.flags[i];
  if (!(flags & 4u)) return false;
  uint32_t cr = g.child_reflect[i], cf = g.child_refract[i];
  double rr = 0.0, rg = 0.0, rb = 0.0, fr = 0.0, fg = 0.0, fb = 0.0;
  if (cr != REF_NONE) {
    double k = g.reflective[i];
    rr = next.cr[cr] * k; rg = next.cg[cr] * k; rb = next.cb[cr] * k;
  }
  if (cf != REF_NONE) {
    double k = g.transparency[i];
    fr = next.cr[cf] * k; fg = next.cg[cf] * k; fb = next.cb[cf] * k;
  }
  double r = g.cr[i], gg = g.cg[i], b = g.cb[i];
  if (flags & 2u) {
    double R = g.reflectance[i];
    r = r + rr * R + fr * (1.0 - R);
    gg = gg + rg * R + fg * (1.0 - R);
    b = b + rb * R + fb * (1.0 - R);
  } else {
    r = r + rr + fr;
    gg = gg + rg + fg;
    b = b + rb + fb;
  }
  if (out_r) {
    *out_r = r; *out_g = gg; *out_b = b;
  } else {
    g.cr[i] = r; g.cg[i] = gg; g.cb[i] = b;
  }
  return true;
}

// Children first: generation g is resolved after generation g + 1.
__global__ void __launch_bounds__(BLOCK)
    k_resolve(GenView g, uint32_t n, GenView next, const uint32_t* queue_counts, uint32_t segcap) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  bool active = i < n;
  if (queue_counts) {  // spawned rays: records found through the segment counters
    uint32_t rec;
    active = queue_slot(queue_counts, segcap, i, rec) && active;
    i = rec;
  }
  if (!active) return;
  resolve_record(g, next, i);
}

// The last step of a camera frame with secondary rays: the pixels whose primary ray had children get
// their final colour (the kernel of generation 0 wrote every pixel with the surface colour alone).
__global__ void __launch_bounds__(BLOCK)
    k_resolve_pixels(GenView g, uint32_t n, GenView next, CameraDev cam, const uint32_t* tiles, uint32_t tiles_x,
                     ImageOut img) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  double r, gg, b;
  if (!resolve_record(g, next, i, &r, &gg, &b)) return;
  uint32_t x = 0, y = 0;
  if (!ray_pixel(tiles, tiles_x, cam, i, x, y)) return;
  const size_t p = (size_t)y * cam.hsize + x;
  if (img.rgb64) {
    img.rgb64[p * 3 + 0] = r; img.rgb64[p * 3 + 1] = gg; img.rgb64[p * 3 + 2] = b;
  }
  if (img.rgb8) {
    uint8_t* dst = img.rgb8 + (img.compact ? (size_t)i : p) * 3;
    dst[0] = quantise(r); dst[1] = quantise(gg); dst[2] = quantise(b);
  }
}

__global__ void __launch_bounds__(BLOCK)
    k_output_rays(GenView g, uint32_t n, double* rgb, int32_t* leaf, double* t) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (rgb) {
    rgb[(size_t)i * 3 + 0] = g.cr[i]; rgb[(size_t)i * 3 + 1] = g.cg[i]; rgb[(size_t)i * 3 + 2] = g.cb[i];
  }
  int l = g.leaf[i];
  if (leaf) leaf[i] = l;
  if (t) t[i] = l >= 0 ? g.t[i] : NAN;
}

// World::intersect + Intersections::hit only (rt_intersect_hit).
template <int STACK, bool CSG>
__global__ void __launch_bounds__(BLOCK) k_intersect(SceneDev S, GenView g, uint32_t n) {
  __shared__ uint32_t lds[RT_STACK_LDS(STACK * BLOCK)];
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  LdsStack<STACK, BLOCK, CSG> st;
  extern __shared__ uint32_t csg_lds[];  // SceneDev::csg_hits * 3 * BLOCK words when the scene has CSG operators, else nothing
  st.init(lds, CSG ? csg_lds : nullptr);
  Counters cnt{};
  ClosestQuery Q;
  Q.init();
  traverse<ClosestQuery, LdsStack<STACK, BLOCK, CSG>, false>(S, load_ray(g, i), Q, st, cnt);
  bool hit = Q.best_leaf != 0x7FFFFFFF;
  g.leaf[i] = hit ? Q.best_leaf : -1;
  if (hit) g.t[i] = Q.best_t;
}

// World::is_shadowed for caller-supplied points.
template <int STACK, bool CSG>
__global__ void __launch_bounds__(BLOCK) k_points_shadowed(SceneDev S, uint32_t n, const double* pts, uint8_t* out) {
  __shared__ uint32_t lds[RT_STACK_LDS(STACK * BLOCK)];
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  LdsStack<STACK, BLOCK, CSG> st;
  extern __shared__ uint32_t csg_lds[];  // SceneDev::csg_hits * 3 * BLOCK words when the scene has CSG operators, else nothing
  st.init(lds, CSG ? csg_lds : nullptr);
  Counters cnt{};
  V3 p = v3(pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]);
  V3 v = v3(S.light_position[0], S.light_position[1], S.light_position[2]) - p;
  ShadowQuery Q;
  Q.distance = magnitude(v);
  Q.found = false;
  Ray ray{p, normalize(v)};
  traverse<ShadowQuery, LdsStack<STACK, BLOCK, CSG>, false>(S, ray, Q, st, cnt);
  out[i] = Q.found ? 1 : 0;
}

__global__ void __launch_bounds__(BLOCK) k_pixel_rays(CameraDev cam, uint32_t n, const uint32_t* xy, double* rays6) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  Ray r = camera_ray(cam, xy[i * 2], xy[i * 2 + 1]);
  double* o = rays6 + (size_t)i * 6;
  o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
  o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

// ---------------------------------------------------------------------------
// refit: new boxes for a hierarchy whose primitives moved (rt_scene_update)
// ---------------------------------------------------------------------------
// The reference lets any shape move between renders (Shape::set_transform re-propagates to the
// children, src/shape.rs:592-656).  The topology of the hierarchy stays; every box is recomputed bottom
// up from the boxes of the primitive slots.  One thread starts at every node none of whose children is
// another node (directly or behind a gate); a thread that finishes a node writes the node's box into
// its slot of the parent and goes on with the parent if it was the last child to arrive.
struct RefitView {
  BvhNode* nodes;
  const float* tri_box;    // 6 floats per TriIsect slot: lo xyz, hi xyz (rounded outward)
  const float* quad_box;   // the same per QuadPrim slot
  const Gate* gates;
  const uint32_t* parent;  // per node: (parent node << 2 | slot), REF_NONE for a node without one
  const uint32_t* waits;   // per node: children whose box another node delivers
  uint32_t* arrived;       // per node, zeroed before the launch
  uint32_t n_nodes;
};

// A plane another thread (possibly on another compute unit) wrote during this launch: read past the
// vector L1, which is not coherent between compute units.
DEV float refit_read(const float* p) {
  return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

DEV void refit_run_box(const RefitView& v, uint32_t ref, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; a++) {
    lo[a] = INFINITY;
    hi[a] = -INFINITY;
  }
  if (ref == REF_NONE) return;
  const uint32_t kind = ref >> 30;
  if (kind != KIND_TRI && kind != KIND_QUAD) return;
  const float* boxes = kind == KIND_TRI ? v.tri_box : v.quad_box;
  const uint32_t first = ref & MAX_RUN_FIRST, count = ((ref >> 27) & 7u) + 1u;
  for (uint32_t i = 0; i < count; i++)
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], boxes[(size_t)(first + i) * 6 + a]);
      hi[a] = fmaxf(hi[a], boxes[(size_t)(first + i) * 6 + 3 + a]);
    }
}

__global__ void __launch_bounds__(256) k_refit(RefitView v) {
  uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= v.n_nodes || v.waits[n] != 0) return;
  for (;;) {
    BvhNode& node = v.nodes[n];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 4; c++) {
      uint32_t ref = node.child[c];
      if (ref != REF_NONE && (ref >> 30) == KIND_GATE) ref = v.gates[ref & 0x3FFFFFFFu].root;  // a run right behind a gate
      if (ref != REF_NONE && ((ref >> 30) == KIND_TRI || (ref >> 30) == KIND_QUAD)) {
        float l[3], h[3];
        refit_run_box(v, ref, l, h);
        for (int a = 0; a < 3; a++) {
          node.plane[a][0][c] = l[a];
          node.plane[a][1][c] = h[a];
        }
      }
      if (node.child[c] != REF_NONE)  // slots of other nodes were written by the threads that finished them
        for (int a = 0; a < 3; a++) {
          lo[a] = fminf(lo[a], refit_read(&node.plane[a][0][c]));
          hi[a] = fmaxf(hi[a], refit_read(&node.plane[a][1][c]));
        }
      node.alt[c] = node.child[c];  // after an update every gate is tested (see rt_scene_update)
    }
    const uint32_t up = v.parent[n];
    if (up == REF_NONE) return;
    BvhNode& p = v.nodes[up >> 2];
    const uint32_t slot = up & 3u;
    for (int a = 0; a < 3; a++) {
      p.plane[a][0][slot] = lo[a];
      p.plane[a][1][slot] = hi[a];
    }
    __threadfence();
    const uint32_t before = atomicAdd(&v.arrived[up >> 2], 1u);
    if (before + 1u != v.waits[up >> 2]) return;  // a sibling subtree is still on its way: its thread goes on
    __threadfence();
    n = up >> 2;
  }
}

// ---------------------------------------------------------------------------
// host side: scene object
// ---------------------------------------------------------------------------
namespace {

struct GenBuf {
  void* slab = nullptr;
  size_t capacity = 0;
  GenView view;
};

constexpr int N_DOUBLE_ARRAYS = 13;
constexpr int N_U32_ARRAYS = 4;

size_t gen_bytes(size_t cap) {
  size_t c = (cap + 63) & ~(size_t)63;
  return c * (N_DOUBLE_ARRAYS * sizeof(double) + N_U32_ARRAYS * sizeof(uint32_t));
}

void carve(GenBuf& b) {
  size_t c = (b.capacity + 63) & ~(size_t)63;
  double* d = (double*)b.slab;
  GenView& v = b.view;
  double** dptr[N_DOUBLE_ARRAYS] = {&v.ox, &v.oy, &v.oz, &v.dx, &v.dy, &v.dz, &v.cr,
                                    &v.cg, &v.cb, &v.reflective, &v.transparency, &v.reflectance, &v.t};
  for (int k = 0; k < N_DOUBLE_ARRAYS; k++) *dptr[k] = d + (size_t)k * c;
  uint32_t* u = (uint32_t*)(d + (size_t)N_DOUBLE_ARRAYS * c);
  v.child_reflect = u;
  v.child_refract = u + c;
  v.flags = u + 2 * c;
  v.leaf = (int32_t*)(u + 3 * c);
}

}  // namespace

struct rt_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  // Asynchronous frames go round-robin to `stream` and the first n_frame_streams - 1 of these.  Three
  // render streams and the copy stream make four, the hardware queues a process has by default
  // (GPU_MAX_HW_QUEUES): a fourth render stream shares a queue and measured no better than two.  A
  // whole 1080p frame fills the GPU from one stream pair; a part of a frame (one GPU of several) is
  // bound by its slowest waves, and a third kernel in flight cut its frame period by a quarter.
  static constexpr int FRAME_STREAMS = 4;
  hipStream_t frame_stream[FRAME_STREAMS - 1] = {nullptr, nullptr, nullptr};
  int n_frame_streams = 3;
  unsigned frame_seq = 0;
  // work enqueued on stream / frame_stream[k] that the copy stream has not been ordered after
  bool stream_used = true, frame_stream_used[FRAME_STREAMS - 1] = {false, false, false};
  hipStream_t last_stream = nullptr;   // where the most recent frame was enqueued
  hipStream_t copy_stream = nullptr;   // rt_download
  hipEvent_t copy_order = nullptr, copy_order2[FRAME_STREAMS - 1] = {nullptr, nullptr, nullptr};
  static constexpr int COPY_FENCES = 8;
  hipEvent_t copy_fence[COPY_FENCES] = {};  // rt_copy_fence / rt_copy_wait
  bool copy_fence_set[COPY_FENCES] = {};
  SceneDev dev{};
  // device arrays of the scene data: address, size, and which pointer of `dev` holds the address
  // (byte offset inside SceneDev), so that rt_scene_clone can rebuild `dev` on another GPU
  struct Alloc {
    void* p;
    size_t bytes;
    size_t field;
  };
  std::vector<Alloc> allocs;
  uint64_t scene_bytes = 0;
  double build_seconds = 0;
  double device_build_seconds = 0;  // of build_seconds: inside the GPU builder (upload, kernels, download)
  size_t device_builds = 0;         // subtrees the GPU builder made
  int stack_need = 1;
  bool has_csg = false;
  // queue lengths per generation of the last multi-generation frame: the next frame is launched on them
  // without asking the device after every generation
  uint32_t hint[MAX_GENERATIONS + 2] = {0};  // [g]: the fullest queue segment of generation g
  uint32_t hint_key[2] = {0, 0};             // n0 and remaining0 the hints belong to
  bool hint_valid = false;
  uint32_t holdoff = 0, holdoff_next = 1;  // frames to render exactly after a speculative frame did not fit
  bool uniform_nodes = false;  // large hierarchy: kernels with the scalar path for nodes a whole wave visits
  bool tri_prefetch = false;   // large mesh: kernels that fetch the next triangle of a run while one is tested
  int max_children = 0;  // rays a hit can spawn: reflective? + transparent?
  GenBuf gens[MAX_GENERATIONS + 1];
  uint32_t* d_counts = nullptr;           // [gen][QUEUE_SEGMENTS]: rays queued for the next generation, per segment
  unsigned long long* d_stats = nullptr;  // STAT_WORDS counters
  uint32_t* h_counts = nullptr;           // pinned mirrors
  unsigned long long* h_stats = nullptr;
  uint32_t* d_tiles = nullptr;
  size_t tiles_capacity = 0;
  std::vector<uint32_t> h_tiles;
  uint32_t tiles_key[4] = {0, 0, 0, 0};  // hsize, vsize, part, n_parts of the uploaded tile table
  hipEvent_t ev[2 * MAX_GENERATIONS] = {};
  // frames enqueued under RT_FLAG_ASYNC and not collected yet: a small ring, each slot with its own
  // statistics block so consecutive frames do not share counters
  struct FrameSlot {
    hipEvent_t begin = nullptr, end = nullptr;
    bool pending = false;
    uint64_t primary = 0;
  };
  // Counters of the asynchronous frames: every such frame adds to the same device words (nothing per frame
  // crosses PCIe or needs a launch of its own); they are read and cleared when the frames are collected.
  unsigned long long* d_async_stats = nullptr;
  unsigned long long* h_async_stats = nullptr;
  bool async_counters_dirty = false;
  static constexpr int FRAME_SLOTS = 8;
  FrameSlot slots[FRAME_SLOTS];
  int next_slot = 0;
  rt_stats async_total{};
  // staging for the host-buffer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // What rt_scene_update needs again: the records in device order, the structure of the description
  // they were made from, and the device arrays of the refit.
  struct Host {
    std::vector<TriIsect> tris;
    std::vector<QuadPrim> quads, unbounded, band_prims;
    std::vector<Gate> gates;
    std::vector<rt_leaf> leaves;            // structure check: type, gate, csg, material / xform / triangle slots
    std::vector<int32_t> gate_parent;
    std::vector<rt_chain_ref> xform_chain, gate_chain;
    std::vector<uint8_t> boxed;             // per leaf: lives in the hierarchy (1) or is tested for every ray (0)
    std::vector<int32_t> inline_plane_leaf;
    int32_t n_xforms = 0, n_triangles = 0, n_materials = 0, n_patterns = 0, n_chain_mats = 0, n_csg = 0;
    uint32_t n_nodes = 0;
    float *d_tri_box = nullptr, *d_quad_box = nullptr;
    uint32_t *d_parent = nullptr, *d_waits = nullptr, *d_arrived = nullptr;
    std::vector<uint32_t> parent, waits;    // host copies (rt_scene_clone uploads them again)
    bool updated = false;                   // every gate is tested since the first update
  } host;
};

// The scene's main stream for an enqueue; the copy stream is ordered after it at the next download.
static inline hipStream_t main_stream(rt_scene* s) {
  s->stream_used = true;
  return s->stream;
}

namespace {

// `out` points at one of the pointers inside s->dev
template <class T>
int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
  *out = nullptr;
  if (v.empty()) return RT_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  s->allocs.push_back(rt_scene::Alloc{p, v.size() * sizeof(T), (size_t)((const char*)out - (const char*)&s->dev)});
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  s->scene_bytes += v.size() * sizeof(T);
  *out = (const T*)p;
  return RT_OK;
}

bool affine(const double* m) {
  return fabs(m[12]) < 1e-9 && fabs(m[13]) < 1e-9 && fabs(m[14]) < 1e-9 && fabs(m[15] - 1.0) < 1e-9;
}
bool is_identity(const double* m) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++)
      if (m[r * 4 + c] != (r == c ? 1.0 : 0.0)) return false;  // -0.0 == 0.0
  return true;
}

void box_of_points(const double* fwd, const double (*pts)[3], int n, double lo[3], double hi[3]) {
  for (int a = 0; a < 3; a++) {
    lo[a] = INFINITY;
    hi[a] = -INFINITY;
  }
  for (int i = 0; i < n; i++) {
    for (int a = 0; a < 3; a++) {
      double v = fwd[a * 4 + 0] * pts[i][0] + fwd[a * 4 + 1] * pts[i][1] + fwd[a * 4 + 2] * pts[i][2] + fwd[a * 4 + 3];
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
  }
  // Pad: the reference finds hits by rounded object-space arithmetic, so a hit can sit a few
  // ulps outside the exact shape; the pad also absorbs the rounding of the slab test.
  for (int a = 0; a < 3; a++) {
    double pad = 1e-7 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-9;
    lo[a] -= pad;
    hi[a] += pad;
  }
}

// Rounding of the reference's quadratic (src/shape.rs:848-865, 987-1010, 1077-1112) for a ray whose
// object-space origin o is far from the shape: b = 2 o.d, c = o.o - 1 and b^2 - 4ac each carry a few
// relative roundings of |o|^2 |d|^2, about 24 * 1.1e-16 |o|^2 |d|^2 in the discriminant, which is
// 4 |d|^2 (1 - p^2) for a line at distance p from the centre (axis): the reference reports hits up to
// p^2 <= 1 + 6.6e-16 |o|^2.  ROUND_NOISE is that with a factor of four in hand.  For a shape with
// object -> world matrix F and world -> object matrix M, |o| <= |M| (|origin| + |centre|) and a distance
// in object space is at most |F| times as long in world space, so the silhouette moves by at most
//   sphere, cylinder (radius 1):  |F| (sqrt(1 + x) - 1) <= |F| x / 2,   x = ROUND_NOISE |M|^2 q,
//   cone (radius |y|, down to 0): |F| sqrt(x),
// with q = 2 (|origin|^2 + |centre|^2) and Frobenius norms.  round_shape_noise() takes the largest
// |F| |M|^2, |F| |M| and |centre|^2 over the scene's round shapes; the traversal widens its box tests by
// the two terms for every ray (rt_types.h: SceneDev::box_noise).
void round_shape_noise(const rt_scene_desc* d, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (int32_t i = 0; i < d->n_leaves; i++) {
    const rt_leaf& L = d->leaves[i];
    if (L.type != RT_SPHERE && L.type != RT_CYLINDER && L.type != RT_CONE) continue;
    const rt_xform& X = d->xforms[L.xform];
    double f2 = 0, m2 = 0, c2 = 0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) {
        f2 += X.fwd[r * 4 + c] * X.fwd[r * 4 + c];
        m2 += X.ray_inv[r * 4 + c] * X.ray_inv[r * 4 + c];
      }
      c2 += X.fwd[r * 4 + 3] * X.fwd[r * 4 + 3];
    }
    if (!(f2 < 1e300) || !(m2 < 1e300) || !(c2 < 1e300)) continue;  // degenerate matrices: such shapes have no box
    const double f = sqrt(f2), m = sqrt(m2);
    if (L.type == RT_CONE) out[1] = fmax(out[1], f * m);
    else out[0] = fmax(out[0], f * m2);
    out[2] = fmax(out[2], c2);
  }
}

// World-space box of a leaf, or false when it has to be tested for every ray.
bool leaf_box(const rt_leaf& L, const rt_xform& X, const rt_triangle* tri, double lo[3], double hi[3]) {
  const double* f = X.fwd;
  double scale = 0;
  for (int c = 0; c < 3; c++)
    scale = fmax(scale, sqrt(f[c] * f[c] + f[4 + c] * f[4 + c] + f[8 + c] * f[8 + c]));
  double x0, x1, y0, y1;
  switch (L.type) {
    case RT_SPHERE: x0 = y0 = -1; x1 = y1 = 1; break;
    case RT_CUBE:
      // A cube treats a ray whose object-space direction component is below 1e-4 as parallel to
      // that slab (src/shape.rs:1372-1377); for a very large cube that widens its silhouette by
      // far more than any pad, so such cubes are tested for every ray.
      if (scale > 100.0) return false;
      x0 = y0 = -1 - 1e-3; x1 = y1 = 1 + 1e-3;
      break;
    case RT_CYLINDER:
      if (!std::isfinite(L.minimum) || !std::isfinite(L.maximum)) return false;
      x0 = -1; x1 = 1; y0 = L.minimum; y1 = L.maximum;
      break;
    case RT_CONE: {
      if (!std::isfinite(L.minimum) || !std::isfinite(L.maximum) || scale > 100.0) return false;
      double r = fmax(fabs(L.minimum), fabs(L.maximum));
      x0 = -r; x1 = r; y0 = L.minimum; y1 = L.maximum;
      break;
    }
    case RT_TRIANGLE:
    case RT_SMOOTH_TRIANGLE: {
      double pts[3][3];
      for (int a = 0; a < 3; a++) {
        pts[0][a] = tri->p1[a];
        pts[1][a] = tri->p1[a] + tri->e1[a];
        pts[2][a] = tri->p1[a] + tri->e2[a];
      }
      box_of_points(f, pts, 3, lo, hi);
      return true;
    }
    default: return false;  // plane
  }
  static const bool corner_boxes = getenv("RT_AMD_CORNER_BOXES") != nullptr;  // the looser boxes, for comparison
  if (L.type != RT_CUBE && !corner_boxes) {
    // Round shapes get their own extent along each world axis instead of the box of their object-space
    // box's corners (up to sqrt(3) wider per axis under a rotation).  A sphere reaches
    // sqrt(f_i0^2 + f_i1^2 + f_i2^2) from its centre along axis i.  A cylinder or cone lies in the hull of
    // its two end discs (a cone's lines pass through the apex from one rim to the other), and a disc of
    // radius r at height y reaches r sqrt(f_i0^2 + f_i2^2) from its centre f_i1 y + f_i3.
    const bool sphere = L.type == RT_SPHERE;
    const double ry0 = L.type == RT_CONE ? fabs(y0) : 1.0, ry1 = L.type == RT_CONE ? fabs(y1) : 1.0;
    for (int a = 0; a < 3; a++) {
      const double fx = f[a * 4 + 0], fy = f[a * 4 + 1], fz = f[a * 4 + 2], c = f[a * 4 + 3];
      if (sphere) {
        const double h = sqrt(fx * fx + fy * fy + fz * fz);
        lo[a] = c - h;
        hi[a] = c + h;
      } else {
        const double rho = sqrt(fx * fx + fz * fz);
        lo[a] = c + fmin(fy * y0 - ry0 * rho, fy * y1 - ry1 * rho);
        hi[a] = c + fmax(fy * y0 + ry0 * rho, fy * y1 + ry1 * rho);
      }
      // the same pad as box_of_points, and the rounding of the extent itself
      const double pad = 2e-7 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-9;
      lo[a] -= pad;
      hi[a] += pad;
    }
    return true;
  }
  double pts[8][3];
  for (int k = 0; k < 8; k++) {
    pts[k][0] = (k & 1) ? x1 : x0;
    pts[k][1] = (k & 2) ? y1 : y0;
    pts[k][2] = (k & 4) ? x1 : x0;
  }
  box_of_points(f, pts, 8, lo, hi);
  return true;
}

void copy3x4(double* dst, const double* src16) { memcpy(dst, src16, 12 * sizeof(double)); }

// The tables the shading code reads, from a description.
struct ShadingTables {
  std::vector<LeafShade> leaves;
  std::vector<XformDev> xforms;
  std::vector<TriShade> tri_shade;
  std::vector<MaterialDev> mats;
  std::vector<PatternDev> pats;
  int max_children = 0;  // rays a hit can spawn: reflective? + transparent?
  void build(const rt_scene_desc* d) {
    leaves.resize(d->n_leaves);
    for (int i = 0; i < d->n_leaves; i++) {
      const rt_leaf& L = d->leaves[i];
      LeafShade& o = leaves[i];
      o.type = L.type; o.closed = L.closed; o.material = L.material; o.xform = L.xform;
      o.tri = L.tri; o.pad = 0; o.minimum = L.minimum; o.maximum = L.maximum;
      for (int a = 0; a < 3; a++) o.plane_normal[a] = L.plane_normal[a];
    }
    xforms.resize(d->n_xforms);
    for (int i = 0; i < d->n_xforms; i++) {
      copy3x4(xforms[i].ray_inv, d->xforms[i].ray_inv);
      copy3x4(xforms[i].full_inv, d->xforms[i].full_inv);
    }
    tri_shade.resize(d->n_triangles);
    for (int i = 0; i < d->n_triangles; i++) {
      const rt_triangle& T = d->triangles[i];
      for (int a = 0; a < 3; a++) {
        tri_shade[i].normal[a] = T.normal[a];
        tri_shade[i].n1[a] = T.n1[a];
        tri_shade[i].n2[a] = T.n2[a];
        tri_shade[i].n3[a] = T.n3[a];
      }
    }
    mats.resize(d->n_materials);
    bool any_reflect = false, any_refract = false;
    for (int i = 0; i < d->n_materials; i++) {
      const rt_material& M = d->materials[i];
      MaterialDev& o = mats[i];
      for (int a = 0; a < 3; a++) o.color[a] = M.color[a];
      o.ambient = M.ambient; o.diffuse = M.diffuse; o.specular = M.specular; o.shininess = M.shininess;
      o.reflective = M.reflective; o.refractive_index = M.refractive_index; o.transparency = M.transparency;
      o.pattern = M.pattern;
      bool refl = !(fabs(M.reflective - 0.0) < FEQ_EPSILON);      // world.rs:151
      bool refr = !(fabs(M.transparency - 0.0) < FEQ_EPSILON);    // world.rs:190
      bool schlick = M.reflective > 0.0 && M.transparency > 0.0;  // world.rs:237
      o.flags = (refl ? 1 : 0) | (refr ? 2 : 0) | (schlick ? 4 : 0) | ((refr || schlick) ? 8 : 0);
      any_reflect |= refl;
      any_refract |= refr;
    }
    max_children = (any_reflect ? 1 : 0) + (any_refract ? 1 : 0);
    pats.resize(d->n_patterns);
    for (int i = 0; i < d->n_patterns; i++) {
      const rt_pattern& P = d->patterns[i];
      pats[i].type = P.type; pats[i].first = P.first; pats[i].second = P.second; pats[i].pad = 0;
      copy3x4(pats[i].inv, P.inv);
      for (int a = 0; a < 3; a++) {
        pats[i].a[a] = P.a[a];
        pats[i].b[a] = P.b[a];
      }
    }
  }
};

// The direction pre-filter of a bounded cone's parallel-ray branch (ConeBand) from its world -> object
// matrix: a(d) = |M d|^2 under diag(1, -1, 1), M = the linear part.
ConeBand make_cone_band(const QuadPrim& q, uint32_t prim) {
  const double* m = q.m;
  const double sg[3] = {1.0, -1.0, 1.0};
  double Q[3][3], sum = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Q[i][j] = 0.0;
      for (int r = 0; r < 3; r++) Q[i][j] += sg[r] * m[r * 4 + i] * m[r * 4 + j];
    }
  ConeBand cb{};
  cb.q[0] = Q[0][0]; cb.q[1] = Q[1][1]; cb.q[2] = Q[2][2];
  cb.q[3] = 2.0 * Q[0][1]; cb.q[4] = 2.0 * Q[0][2]; cb.q[5] = 2.0 * Q[1][2];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) sum += m[r * 4 + c] * m[r * 4 + c];
  cb.slack = 1e-11 * sum;
  cb.prim = prim;
  return cb;
}

// Single-precision copies: the direction's components are rounded to single (2^-24 each, so 2^-23 per
// product), then six products and five sums within 2^-24 relative each, on top of the rounding of the
// coefficients: below 1e-6 * sum |q| in all; 4e-6 * sum |q| is added to the slack.
std::vector<ConeBandF32> cone_bands_f32(const std::vector<ConeBand>& bands) {
  std::vector<ConeBandF32> out(bands.size());
  for (size_t i = 0; i < bands.size(); i++) {
    double sum = 0;
    for (int c = 0; c < 6; c++) {
      out[i].q[c] = (float)bands[i].q[c];
      sum += fabs(bands[i].q[c]);
    }
    out[i].slack = std::nextafterf((float)(bands[i].slack + 4e-6 * sum), INFINITY);
    out[i].prim = bands[i].prim;
  }
  return out;
}

struct SceneBuild {
  const rt_scene_desc* d;
  BvhBuilder bvh;
  std::vector<std::vector<int>> gate_children, gate_leaves;
  std::vector<int> top_gates, top_leaves;
  std::vector<Gate> gates;
  std::vector<QuadPrim> quad_src, unbounded, band_prims;
  std::vector<ConeBand> bands;
  // CSG: one root per maximal operator tree
  std::vector<std::vector<int>> gate_csg;  // roots directly inside each group
  std::vector<int> top_csg;
  std::vector<CsgRoot> csg_roots;
  std::vector<CsgNode> csg_nodes;
  std::vector<CsgLeaf> csg_leaves;         // ref filled in once the BVH orders are final
  std::vector<QuadPrim> csg_quads;
  std::vector<TriIsect> csg_tris;
  std::vector<uint32_t> csg_unbounded;
  std::vector<int> csg_root_of;            // operator index -> root index (for roots), else -1
  int csg_hits = CSG_MIN_HITS;             // the longest hit list a root of this scene can produce
  std::vector<TriIsect> tri_src;
  std::vector<int> tri_xform;  // device xform index or -1 (identity)
  int stack_need = 1;

  bool xform_has_chain(int x) const { return d->xform_chain && d->xform_chain[x].count > 0; }
  void gate_chain(int gi, Gate& g) const {
    g.chain_first = g.chain_len = 0;
    if (d->gate_chain && d->gate_chain[gi].count > 0) {
      g.chain_first = (uint32_t)d->gate_chain[gi].first;
      g.chain_len = (uint32_t)d->gate_chain[gi].count;
    }
  }

  QuadPrim make_quad(int li) const {
    const rt_leaf& L = d->leaves[li];
    QuadPrim q{};
    copy3x4(q.m, d->xforms[L.xform].ray_inv);
    q.minimum = L.minimum;
    q.maximum = L.maximum;
    q.type = L.type | (xform_has_chain(L.xform) ? QUAD_CHAIN : 0);
    q.closed = L.closed;
    q.leaf = li;
    q.gate = L.gate;
    return q;
  }

  TriIsect make_tri(int li) const {
    const rt_leaf& L = d->leaves[li];
    const rt_triangle& T = d->triangles[L.tri];
    TriIsect ti{};
    for (int a = 0; a < 3; a++) {
      ti.p1[a] = T.p1[a];
      ti.e1[a] = T.e1[a];
      ti.e2[a] = T.e2[a];
    }
    ti.leaf = li;
    ti.xform = is_identity(d->xforms[L.xform].ray_inv) ? -1 : L.xform;
    if (xform_has_chain(L.xform)) ti.xform = L.xform | TRI_XFORM_CHAIN;
    return ti;
  }

  // Records of one CSG root (operator `top` of the description and everything below it) and its item
  // for the enclosing level; a root without a finite box goes on the list tested for every ray.
  void csg_items(const std::vector<int>& roots, std::vector<BuildItem>& items) {
    for (int top : roots) {
      const rt_csg& C = d->csg[top];
      CsgRoot R{};
      R.first_leaf = (uint32_t)csg_leaves.size();
      R.first_node = (uint32_t)csg_nodes.size();
      R.gate = C.gate;
      int last = top;  // operators below `top` follow it directly (parents first)
      while (last + 1 < d->n_csg) {
        int a = last + 1;
        while (a > top) a = d->csg[a].parent;
        if (a != top) break;
        last++;
      }
      for (int k = top; k <= last; k++) csg_nodes.push_back(CsgNode{d->csg[k].op, k == top ? -1 : d->csg[k].parent - top});
      R.n_nodes = (uint32_t)(last - top + 1);
      BuildItem it{};
      for (int a = 0; a < 3; a++) {
        it.lo[a] = INFINITY;
        it.hi[a] = -INFINITY;
      }
      bool bounded = true;
      int possible = 0;
      for (int li = C.first_leaf; li < C.first_leaf + C.n_leaves; li++) {
        const rt_leaf& L = d->leaves[li];
        if (L.type == RT_EMPTY) continue;
        bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
        double lo[3], hi[3];
        // a cone's parallel-ray root can lie anywhere (src/shape.rs:1097-1102): no box bounds the operator then
        if (L.type == RT_CONE || !leaf_box(L, d->xforms[L.xform], tri ? &d->triangles[L.tri] : nullptr, lo, hi)) bounded = false;
        else
          for (int a = 0; a < 3; a++) {
            it.lo[a] = fmin(it.lo[a], lo[a]);
            it.hi[a] = fmax(it.hi[a], hi[a]);
          }
        CsgLeaf cl{};
        if (tri) {
          cl.ref = make_run(KIND_TRI, (uint32_t)csg_tris.size(), 1);
          csg_tris.push_back(make_tri(li));
          possible += 1;
        } else {
          cl.ref = make_run(KIND_QUAD, (uint32_t)csg_quads.size(), 1);
          csg_quads.push_back(make_quad(li));
          possible += (L.type == RT_SPHERE || L.type == RT_CUBE) ? 2 : (L.type == RT_PLANE ? 1 : 4);
        }
        cl.node = L.csg - top;
        cl.left_mask = L.csg_left;
        cl.gate = L.gate;
        csg_leaves.push_back(cl);
      }
      R.n_leaves = (uint32_t)csg_leaves.size() - R.first_leaf;
      if (possible > CSG_MAX_HITS)
        throw std::length_error("a CSG operator tree whose shapes can yield more than 60 intersections per ray");
      csg_hits = std::max(csg_hits, possible);
      const uint32_t root_index = (uint32_t)csg_roots.size();
      csg_roots.push_back(R);
      if (R.n_leaves == 0) continue;
      if (!bounded) {
        csg_unbounded.push_back(root_index);
        continue;
      }
      Gate g{};  // stands for the root in the BVH
      g.root = REF_NONE;
      g.parent = C.gate;
      g.csg = root_index + 1;
      it.kind = KIND_GATE;
      it.index = (uint32_t)gates.size();
      gates.push_back(g);
      if (bvh.gate_alt.size() < gates.size()) bvh.gate_alt.resize(gates.size(), REF_NONE);
      bvh.gate_alt[it.index] = make_ref(KIND_GATE, it.index);
      if (bvh.gate_need.size() < gates.size()) bvh.gate_need.resize(gates.size(), 0);
      items.push_back(it);
    }
  }

  // Items of the leaves directly under one group (or the top level).
  void leaf_items(const std::vector<int>& leaves, std::vector<BuildItem>& items) {
    for (int li : leaves) {
      const rt_leaf& L = d->leaves[li];
      if (L.type == RT_EMPTY) continue;  // never intersects (src/shape.rs:789)
      const rt_xform& X = d->xforms[L.xform];
      BuildItem it{};
      bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
      if (!leaf_box(L, X, tri ? &d->triangles[L.tri] : nullptr, it.lo, it.hi)) {
        unbounded.push_back(make_quad(li));
        continue;
      }
      if (tri) {
        it.kind = KIND_TRI;
        it.index = (uint32_t)tri_src.size();
        tri_src.push_back(make_tri(li));
      } else {
        it.kind = KIND_QUAD;
        it.index = (uint32_t)quad_src.size();
        QuadPrim q = make_quad(li);
        if (L.type == RT_CONE) {
          ConeBand cb = make_cone_band(q, (uint32_t)band_prims.size());
          bands.push_back(cb);
          QuadPrim only = q;
          only.type |= QUAD_ONLY_PARALLEL;
          band_prims.push_back(only);
          q.type |= QUAD_SKIP_PARALLEL;
        }
        quad_src.push_back(q);
      }
      items.push_back(it);
    }
  }

  // Builds the subtree of gate gi and its item for the enclosing level.
  void build_gate(int gi, BuildItem* out_item, bool* has_item) {
    std::vector<BuildItem> items;
    for (int c : gate_children[gi]) {
      BuildItem it{};
      bool ok = false;
      build_gate(c, &it, &ok);
      if (ok) items.push_back(it);
    }
    leaf_items(gate_leaves[gi], items);
    csg_items(gate_csg[gi], items);
    const rt_gate& G = d->gates[gi];
    Gate& g = gates[gi];
    copy3x4(g.m, G.to_local);
    gate_chain(gi, g);
    for (int a = 0; a < 3; a++) {
      g.bmin[a] = G.bmin[a];
      g.bmax[a] = G.bmax[a];
    }
    g.parent = G.parent;
    g.csg = 0;
    double lo[3], hi[3];
    g.root = bvh.build(items, lo, hi);
    if ((size_t)gi >= bvh.gate_need.size()) bvh.gate_need.resize(gi + 1, 0);
    bvh.gate_need[gi] = bvh.stack_need + 1;
    // The exact test can be skipped for rays without a near-zero direction component when the
    // group space is world space and the reference box holds everything below (boxes are padded
    // by about 1e-7 relative, hence the tolerance).
    bool contains = g.root != REF_NONE;
    for (int a = 0; a < 3 && contains; a++) {
      double tol = 1e-6 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-8;
      if (!(G.bmin[a] <= lo[a] + tol) || !(G.bmax[a] >= hi[a] - tol)) contains = false;
    }
    g.skippable = (contains && is_identity(G.to_local)) ? 1u : 0u;
    if ((size_t)gi >= bvh.gate_alt.size()) bvh.gate_alt.resize(gi + 1, REF_NONE);
    bvh.gate_alt[gi] = g.skippable ? bvh.alt(g.root) : make_ref(KIND_GATE, (uint32_t)gi);
    *has_item = g.root != REF_NONE;
    if (*has_item) {
      out_item->kind = KIND_GATE;
      out_item->index = (uint32_t)gi;
      for (int a = 0; a < 3; a++) {
        out_item->lo[a] = lo[a];
        out_item->hi[a] = hi[a];
      }
    }
  }
};

// 40 entries x 64 lanes x 4 B = 10 KB per wave: sixteen waves still fit the 160 KB of LDS of a CU,
// so the stack does not limit occupancy; deeper scenes take the 64-entry kernels (10 waves per CU).
#ifndef RT_STACK_SMALL
#define RT_STACK_SMALL 40
#endif
constexpr int STACK_SMALL = RT_STACK_SMALL;
int pick_stack(int need) { return need <= STACK_SMALL ? STACK_SMALL : 64; }

}  // namespace

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }

extern "C" int rt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

extern "C" void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  // frames may still run on any render stream and copies on the copy stream: drain them all
  // before any of the memory they use is released
  for (hipStream_t st : {s->stream, s->frame_stream[0], s->frame_stream[1], s->frame_stream[2], s->copy_stream})
    if (st) (void)hipStreamSynchronize(st);
  for (const rt_scene::Alloc& a : s->allocs) (void)hipFree(a.p);
  for (GenBuf& g : s->gens)
    if (g.slab) (void)hipFree(g.slab);
  if (s->d_counts) (void)hipFree(s->d_counts);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  if (s->d_tiles) (void)hipFree(s->d_tiles);
  if (s->stage) (void)hipFree(s->stage);
  for (void* p : {(void*)s->host.d_tri_box, (void*)s->host.d_quad_box, (void*)s->host.d_parent, (void*)s->host.d_waits,
                  (void*)s->host.d_arrived})
    if (p) (void)hipFree(p);
  for (hipEvent_t e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->d_async_stats) (void)hipFree(s->d_async_stats);
  if (s->h_async_stats) (void)hipHostFree(s->h_async_stats);
  for (rt_scene::FrameSlot& f : s->slots)
    for (hipEvent_t e : {f.begin, f.end})
      if (e) (void)hipEventDestroy(e);
  if (s->copy_stream) {
    (void)hipStreamSynchronize(s->copy_stream);
    (void)hipStreamDestroy(s->copy_stream);
  }
  if (s->copy_order) (void)hipEventDestroy(s->copy_order);
  for (hipEvent_t e : s->copy_order2)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->copy_fence)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : s->frame_stream)
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// Streams, events and the per-frame counters of a scene on `device` (which becomes current).
static int scene_runtime_init(rt_scene* s, int device) {
  HIP_TRY(hipSetDevice(device));
  s->device = device;
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  for (hipStream_t& st : s->frame_stream) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  if (const char* e = getenv("RT_AMD_FRAME_STREAMS")) s->n_frame_streams = std::min(std::max(atoi(e), 1), (int)rt_scene::FRAME_STREAMS);
  s->last_stream = s->stream;
  HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&s->copy_order, hipEventDisableTiming));
  for (hipEvent_t& e : s->copy_order2) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : s->copy_fence) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipMalloc((void**)&s->d_counts, COUNT_WORDS * sizeof(uint32_t)));
  HIP_TRY(hipMalloc((void**)&s->d_stats, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc((void**)&s->h_counts, COUNT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&s->h_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  for (hipEvent_t& e : s->ev) HIP_TRY(hipEventCreate(&e));
  HIP_TRY(hipMalloc((void**)&s->d_async_stats, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(s->d_async_stats, 0, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc((void**)&s->h_async_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  for (rt_scene::FrameSlot& f : s->slots) {
    HIP_TRY(hipEventCreate(&f.begin));
    HIP_TRY(hipEventCreate(&f.end));
  }
  return RT_OK;
}

static int attach_band_grid(rt_scene* s, size_t n_bands);

static int scene_create_impl(const rt_scene_desc* d, int device, uint32_t build_flags, rt_scene* s) {
  // ---- validate
  if (d->n_leaves < 0 || d->n_gates < 0 || d->n_xforms < 0 || d->n_triangles < 0 || d->n_materials < 0 ||
      d->n_patterns < 0)
    return fail(RT_ERR_INVALID, "negative count in scene description");
  if ((d->n_leaves > 0 && !d->leaves) || (d->n_gates > 0 && !d->gates) || (d->n_xforms > 0 && !d->xforms) ||
      (d->n_triangles > 0 && !d->triangles) || (d->n_materials > 0 && !d->materials) || (d->n_patterns > 0 && !d->patterns))
    return fail(RT_ERR_INVALID, "null array with a positive count in scene description");
  for (int i = 0; i < d->n_xforms; i++) {
    if (!affine(d->xforms[i].ray_inv) || !affine(d->xforms[i].full_inv) || !affine(d->xforms[i].fwd))
      return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) shape transform: last row must be 0 0 0 1");
  }
  for (int i = 0; i < d->n_gates; i++) {
    if (!affine(d->gates[i].to_local)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) group transform");
    if (d->gates[i].parent >= i || d->gates[i].parent < -1)
      return fail(RT_ERR_INVALID, "gate parent must precede its children");
  }
  if (d->n_csg < 0 || (d->n_csg > 0 && !d->csg)) return fail(RT_ERR_INVALID, "bad CSG operator list");
  if (d->n_chain_mats < 0 || (d->n_chain_mats > 0 && !d->chain_mats)) return fail(RT_ERR_INVALID, "bad transform chain list");
  for (int i = 0; i < d->n_chain_mats; i++)
    if (!affine(d->chain_mats + 16 * (size_t)i)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) matrix in a transform chain");
  bool any_chain = false;
  for (int pass = 0; pass < 2; pass++) {
    const rt_chain_ref* refs = pass ? d->gate_chain : d->xform_chain;
    const int n = pass ? d->n_gates : d->n_xforms;
    for (int i = 0; refs && i < n; i++) {
      if (refs[i].count < 0 || refs[i].first < 0 || (int64_t)refs[i].first + refs[i].count > d->n_chain_mats)
        return fail(RT_ERR_INVALID, "transform chain out of range");
      any_chain = any_chain || refs[i].count > 0;
    }
  }
  for (int i = 0; i < d->n_csg; i++) {
    const rt_csg& c = d->csg[i];
    if (c.op < RT_CSG_UNION || c.op > RT_CSG_DIFFERENCE) return fail(RT_ERR_INVALID, "unknown CSG operator");
    if (c.parent >= i || c.parent < -1) return fail(RT_ERR_INVALID, "CSG parent must precede its children");
    if (c.gate < -1 || c.gate >= d->n_gates) return fail(RT_ERR_INVALID, "CSG gate out of range");
    if (c.first_leaf < 0 || c.n_leaves < 0 || (int64_t)c.first_leaf + c.n_leaves > d->n_leaves)
      return fail(RT_ERR_INVALID, "CSG leaf range out of bounds");
    if (c.parent >= 0) {
      const rt_csg& p = d->csg[c.parent];
      if (c.first_leaf < p.first_leaf || c.first_leaf + c.n_leaves > p.first_leaf + p.n_leaves)
        return fail(RT_ERR_INVALID, "CSG leaf range not inside its parent's");
    }
  }
  for (int i = 0; i < d->n_gates; i++) {
    const rt_gate& G = d->gates[i];
    if (G.csg < -1 || G.csg >= d->n_csg) return fail(RT_ERR_INVALID, "gate CSG index out of range");
    // A group below an operator lies below the same operator as its parent group, or below one nested in it
    // (an operator between the two groups); a group outside every operator has no parent inside one.
    if (G.parent >= 0) {
      const int pc = d->gates[G.parent].csg;
      int a = G.csg;
      while (a >= 0 && a != pc) a = d->csg[a].parent;
      if (a != pc) return fail(RT_ERR_INVALID, "gate CSG operator does not lie below its parent gate's");
    }
  }
  for (int i = 0; i < d->n_patterns; i++) {
    const rt_pattern& p = d->patterns[i];
    if (!affine(p.inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) pattern transform");
    if (p.type < RT_PAT_NULL || p.type > RT_PAT_BLEND) return fail(RT_ERR_INVALID, "unknown pattern type");
    if (p.type == RT_PAT_BLEND && (p.first < 0 || p.first >= d->n_patterns || p.second < 0 || p.second >= d->n_patterns))
      return fail(RT_ERR_INVALID, "blend pattern operand out of range");
  }
  for (int i = 0; i < d->n_materials; i++)
    if (d->materials[i].pattern < -1 || d->materials[i].pattern >= d->n_patterns)
      return fail(RT_ERR_INVALID, "material pattern index out of range");
  for (int i = 0; i < d->n_leaves; i++) {
    const rt_leaf& L = d->leaves[i];
    if (L.type < RT_EMPTY || L.type > RT_SMOOTH_TRIANGLE) return fail(RT_ERR_INVALID, "unknown leaf type");
    if (L.material < 0 || L.material >= d->n_materials) return fail(RT_ERR_INVALID, "leaf material out of range");
    if (L.xform < 0 || L.xform >= d->n_xforms) return fail(RT_ERR_INVALID, "leaf xform out of range");
    if (L.gate < -1 || L.gate >= d->n_gates) return fail(RT_ERR_INVALID, "leaf gate out of range");
    if (L.csg < -1 || L.csg >= d->n_csg) return fail(RT_ERR_INVALID, "leaf CSG index out of range");
    if (L.csg >= 0 && (i < d->csg[L.csg].first_leaf || i >= d->csg[L.csg].first_leaf + d->csg[L.csg].n_leaves))
      return fail(RT_ERR_INVALID, "leaf outside the leaf range of its CSG operator");
    if (L.gate >= 0) {
      // the leaf's innermost operator is its group's, or one nested below the group's
      const int gc = d->gates[L.gate].csg;
      int a = L.csg;
      while (a >= 0 && a != gc) a = d->csg[a].parent;
      if (a != gc) return fail(RT_ERR_INVALID, "leaf CSG operator does not lie below its gate's");
    }
    bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
    if (tri && (L.tri < 0 || L.tri >= d->n_triangles)) return fail(RT_ERR_INVALID, "leaf triangle out of range");
  }
  if ((uint64_t)d->n_triangles > MAX_RUN_FIRST || (uint64_t)d->n_leaves > (1u << 24))
    return fail(RT_ERR_UNSUPPORTED, "too many primitives");  // node byte offsets are 32-bit

  int rci = scene_runtime_init(s, device);
  if (rci) return rci;

  // ---- acceleration structure
  auto t0 = std::chrono::steady_clock::now();
  SceneBuild b;
  b.d = d;
  {  // where the hierarchy is built: RT_BUILD_DEVICE, or RT_AMD_DEVICE_BVH=1 / 0 in the environment
    bool on_device = (build_flags & RT_BUILD_DEVICE) != 0;
    if (const char* e = getenv("RT_AMD_DEVICE_BVH")) on_device = atoi(e) != 0;
    b.bvh.on_device = on_device;
    if (const char* e = getenv("RT_AMD_DEVICE_BVH_MIN")) b.bvh.device_min_items = (size_t)std::max(2, atoi(e));
  }
  b.gate_children.resize(d->n_gates);
  b.gate_leaves.resize(d->n_gates);
  b.gates.reserve((size_t)d->n_gates + (size_t)d->n_csg);
  b.gates.resize(d->n_gates);
  b.gate_csg.resize(d->n_gates);
  for (int i = 0; i < d->n_gates; i++) {
    const rt_gate& G = d->gates[i];
    if (G.csg >= 0) {
      // a group below a CSG operator: only its gate record is needed (eval_csg tests it per leaf)
      Gate& g = b.gates[i];
      copy3x4(g.m, G.to_local);
      b.gate_chain(i, g);
      for (int a = 0; a < 3; a++) {
        g.bmin[a] = G.bmin[a];
        g.bmax[a] = G.bmax[a];
      }
      g.root = REF_NONE;
      g.parent = G.parent;
      continue;
    }
    if (G.parent < 0) b.top_gates.push_back(i);
    else b.gate_children[G.parent].push_back(i);
  }
  for (int i = 0; i < d->n_leaves; i++) {
    if (d->leaves[i].csg >= 0) continue;  // reached through its CSG root
    if (d->leaves[i].gate < 0) b.top_leaves.push_back(i);
    else b.gate_leaves[d->leaves[i].gate].push_back(i);
  }
  for (int i = 0; i < d->n_csg; i++) {
    if (d->csg[i].parent >= 0) continue;
    if (d->csg[i].gate < 0) b.top_csg.push_back(i);
    else b.gate_csg[d->csg[i].gate].push_back(i);
  }
  std::vector<BuildItem> items;
  for (int gi : b.top_gates) {
    BuildItem it{};
    bool ok = false;
    b.build_gate(gi, &it, &ok);
    if (ok) items.push_back(it);
  }
  b.leaf_items(b.top_leaves, items);
  b.csg_items(b.top_csg, items);
  double lo[3], hi[3];
  uint32_t root = b.bvh.build(items, lo, hi);
  s->stack_need = b.bvh.stack_need + 2 + 3;  // + root and gate hand-over, + 3 scratch slots of push_if
  if (s->stack_need > TRAVERSAL_STACK)
    return fail(RT_ERR_UNSUPPORTED, "scene hierarchy is deeper than the traversal stack");
  if (getenv("RT_AMD_DEBUG")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_generation<STACK_SMALL, false, 0, false, false>, BLOCK, 0);
    fprintf(stderr, "[rt_amd] k_generation<%d> blocks of %d per CU (occupancy API): %d\n", STACK_SMALL, BLOCK, occ);
    fprintf(stderr, "[rt_amd] leaves=%d gates=%d nodes=%zu tris=%zu quads=%zu unbounded=%zu bands=%zu stack_need=%d\n", d->n_leaves,
            d->n_gates, b.bvh.nodes.size(), b.bvh.tri_order.size(), b.bvh.quad_order.size(), b.unbounded.size(), b.bands.size(),
            s->stack_need);
  }

  std::vector<TriIsect> tris(b.bvh.tri_order.size());
  for (size_t i = 0; i < tris.size(); i++) tris[i] = b.tri_src[b.bvh.tri_order[i]];
  std::vector<QuadPrim> quads(b.bvh.quad_order.size());
  for (size_t i = 0; i < quads.size(); i++) quads[i] = b.quad_src[b.bvh.quad_order[i]];
  // the shapes below CSG operators follow the BVH-ordered records
  for (CsgLeaf& cl : b.csg_leaves) cl.ref += (uint32_t)((cl.ref >> 30) == KIND_TRI ? tris.size() : quads.size());
  tris.insert(tris.end(), b.csg_tris.begin(), b.csg_tris.end());
  quads.insert(quads.end(), b.csg_quads.begin(), b.csg_quads.end());
  if (tris.size() > MAX_RUN_FIRST || quads.size() > MAX_RUN_FIRST) return fail(RT_ERR_UNSUPPORTED, "too many primitives");
  s->has_csg = !b.csg_roots.empty() || any_chain;  // either needs the general kernels
  s->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  s->device_build_seconds = b.bvh.device_seconds;
  s->device_builds = b.bvh.device_builds;

  // ---- shading tables
  ShadingTables sh;
  sh.build(d);
  s->max_children = sh.max_children;
  std::vector<LeafShade>& leaves = sh.leaves;
  std::vector<XformDev>& xforms = sh.xforms;
  std::vector<TriShade>& tri_shade = sh.tri_shade;
  std::vector<MaterialDev>& mats = sh.mats;
  std::vector<PatternDev>& pats = sh.pats;

  // ---- upload
  int rc;
  if ((rc = upload(s, b.bvh.nodes, &s->dev.nodes))) return rc;
  if ((rc = upload(s, tris, &s->dev.tris))) return rc;
  if ((rc = upload(s, quads, &s->dev.quads))) return rc;
  if ((rc = upload(s, b.gates, &s->dev.gates))) return rc;
  // top-level planes without a transform chain go into the kernel arguments (the order among the
  // shapes tested for every ray does not matter: equal distances resolve by leaf index)
  s->dev.n_inline_planes = 0;
  for (size_t i = 0; i < b.unbounded.size();) {
    const QuadPrim& q = b.unbounded[i];
    if (q.type == RT_PLANE && q.gate < 0 && s->dev.n_inline_planes < MAX_INLINE_PLANES) {
      InlinePlane& P = s->dev.inline_planes[s->dev.n_inline_planes++];
      for (int k = 0; k < 4; k++) P.row[k] = q.m[4 + k];
      P.leaf = q.leaf;
      P.pad = 0;
      b.unbounded.erase(b.unbounded.begin() + (long)i);
    } else {
      i++;
    }
  }
  if ((rc = upload(s, b.unbounded, &s->dev.unbounded))) return rc;
  if ((rc = upload(s, b.csg_roots, &s->dev.csg_roots))) return rc;
  if ((rc = upload(s, b.csg_nodes, &s->dev.csg_nodes))) return rc;
  if ((rc = upload(s, b.csg_leaves, &s->dev.csg_leaves))) return rc;
  if ((rc = upload(s, b.csg_unbounded, &s->dev.csg_unbounded))) return rc;
  s->dev.n_csg_unbounded = (int32_t)b.csg_unbounded.size();
  s->dev.csg_hits = b.csg_hits;
  if ((rc = upload(s, b.bands, &s->dev.bands))) return rc;
  {
    std::vector<ConeBandF32> bands32 = cone_bands_f32(b.bands);
    if ((rc = upload(s, bands32, &s->dev.bands32))) return rc;
  }
  if ((rc = upload(s, b.band_prims, &s->dev.band_prims))) return rc;
  s->dev.n_bands = (int32_t)b.bands.size();
#ifdef RT_DIAG_NO_BANDS  // diagnostic builds only (timing of the band list): drops the cones' parallel-ray hits
  if (getenv("RT_AMD_TIMING_NO_BANDS")) s->dev.n_bands = 0;
  s->dev.band_diag_passes = getenv("RT_AMD_TIMING_BAND_PASSES") ? atoi(getenv("RT_AMD_TIMING_BAND_PASSES")) : 3;
#endif
  if (getenv("RT_AMD_BAND_GRID_HOST")) {  // the host builder (what the device builder is checked against)
    BandGrid grid = build_band_grid(b.bands, (size_t)1 << 29);  // at most 2 GiB of entries
    s->dev.band_grid_n = grid.n;
    if ((rc = upload(s, grid.start, &s->dev.band_start))) return rc;
    if ((rc = upload(s, grid.items, &s->dev.band_items))) return rc;
    grid.strip.resize(grid.strip.size() + 4, 0u);  // a ray reads four words at a time
    if ((rc = upload(s, grid.strip, &s->dev.band_strip))) return rc;
    if (getenv("RT_AMD_DEBUG"))
      fprintf(stderr, "[rt_amd] band grid n=%d cells=%zu entries=%zu\n", grid.n, grid.start.size(), grid.items.size());
  } else {
    // built on this GPU from the bands just uploaded: milliseconds where the host builder takes seconds
    if ((rc = attach_band_grid(s, b.bands.size()))) return rc;
  }
  if ((rc = upload(s, leaves, &s->dev.leaves))) return rc;
  if ((rc = upload(s, xforms, &s->dev.xforms))) return rc;
  if (any_chain) {
    std::vector<double> chain_mats((size_t)d->n_chain_mats * 12);
    for (int i = 0; i < d->n_chain_mats; i++) copy3x4(&chain_mats[(size_t)i * 12], d->chain_mats + 16 * (size_t)i);
    std::vector<uint32_t> xform_chain((size_t)d->n_xforms * 2, 0u);
    for (int i = 0; d->xform_chain && i < d->n_xforms; i++) {
      xform_chain[2 * (size_t)i] = (uint32_t)d->xform_chain[i].first;
      xform_chain[2 * (size_t)i + 1] = (uint32_t)d->xform_chain[i].count;
    }
    if ((rc = upload(s, chain_mats, &s->dev.chain_mats))) return rc;
    if ((rc = upload(s, xform_chain, &s->dev.xform_chain))) return rc;
  }
  if ((rc = upload(s, tri_shade, &s->dev.tri_shade))) return rc;
  if ((rc = upload(s, mats, &s->dev.materials))) return rc;
  if ((rc = upload(s, pats, &s->dev.patterns))) return rc;
  s->dev.root = root;
  s->dev.root_alt = b.bvh.alt(root);
  {  // the host copy rt_scene_update works from
    rt_scene::Host& H = s->host;
    H.tris = tris;
    H.quads = quads;
    H.unbounded = b.unbounded;
    H.band_prims = b.band_prims;
    H.gates = b.gates;
    H.leaves.assign(d->leaves, d->leaves + d->n_leaves);
    H.gate_parent.resize(d->n_gates);
    for (int i = 0; i < d->n_gates; i++) H.gate_parent[i] = d->gates[i].parent;
    if (d->xform_chain) H.xform_chain.assign(d->xform_chain, d->xform_chain + d->n_xforms);
    if (d->gate_chain) H.gate_chain.assign(d->gate_chain, d->gate_chain + d->n_gates);
    H.boxed.assign((size_t)d->n_leaves, 0);
    for (size_t i = 0; i < b.bvh.tri_order.size(); i++) H.boxed[(size_t)tris[i].leaf] = 1;
    for (size_t i = 0; i < b.bvh.quad_order.size(); i++) H.boxed[(size_t)quads[i].leaf] = 1;
    for (int i = 0; i < s->dev.n_inline_planes; i++) H.inline_plane_leaf.push_back(s->dev.inline_planes[i].leaf);
    H.n_xforms = d->n_xforms; H.n_triangles = d->n_triangles; H.n_materials = d->n_materials;
    H.n_patterns = d->n_patterns; H.n_chain_mats = d->n_chain_mats; H.n_csg = d->n_csg;
    // who delivers the box of every node: its parent slot, through a gate when the node is a group's root
    const std::vector<BvhNode>& nodes = b.bvh.nodes;
    H.n_nodes = (uint32_t)nodes.size();
    H.parent.assign(nodes.size(), REF_NONE);
    H.waits.assign(nodes.size(), 0u);
    for (size_t n = 0; n < nodes.size(); n++)
      for (int c = 0; c < 4; c++) {
        uint32_t ref = nodes[n].child[c];
        if (ref != REF_NONE && (ref >> 30) == KIND_GATE) {
          const Gate& g = b.gates[ref & 0x3FFFFFFFu];
          ref = g.csg ? REF_NONE : g.root;
        }
        if (ref != REF_NONE && (ref >> 30) == KIND_INNER) {
          H.parent[ref & 0x3FFFFFFFu] = (uint32_t)(n << 2) | (uint32_t)c;
          H.waits[n]++;
        }
      }
  }
  // Measured: the scalar path for wave-uniform nodes pays off for hierarchies far beyond the 32 KB L1
  // of a CU (1M-triangle mesh: -6 % frame time) and costs a few percent on small ones (12k triangles:
  // +5 %), whose upper levels stay in L1 anyway.
  s->uniform_nodes = b.bvh.nodes.size() >= 8192;
  if (const char* e = getenv("RT_AMD_UNIFORM_NODES")) s->uniform_nodes = atoi(e) != 0;
  // Measured: in a run of triangles, fetching the next record while one is tested takes a tenth off the
  // frame of a 1M-triangle mesh (2.72 -> 2.44 ms at 4K: ten triangle tests per ray, records far beyond a
  // CU's L1) and costs the 12k-triangle teapot 6 % (the twenty registers held across a test are spilled,
  // and its one or two tests per ray find their records in L1).  Scenes that spawn rays keep the plain
  // loop (their kernels spill more than they gain).
  s->tri_prefetch = s->uniform_nodes && b.tri_src.size() * sizeof(TriIsect) >= (4u << 20);
  if (const char* e = getenv("RT_AMD_TRI_PREFETCH")) s->tri_prefetch = atoi(e) != 0;
  s->dev.n_unbounded = (int32_t)b.unbounded.size();
  s->dev.n_leaves = d->n_leaves;
  for (int a = 0; a < 3; a++) {
    s->dev.light_intensity[a] = d->light.intensity[a];
    s->dev.light_position[a] = d->light.position[a];
  }
  round_shape_noise(d, s->dev.box_noise);
  // Not in scenes with bounded cones: their parallel-ray branch (the band list) puts hits up to 1e8 units
  // out on the cones' extensions, and a ray that starts there would, widened, meet every box of the scene
  // one after the other in a single lane (measured: 12 -> 200 ms per frame with 25 000 cones, for rays
  // whose reference results are rounding noise).  Far cameras and lights are the case this is for.
  if (s->dev.n_bands > 0) s->dev.box_noise[0] = s->dev.box_noise[1] = 0.0;

  return RT_OK;
}

extern "C" int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
  return rt_scene_create_with(desc, device, 0u, out);
}

extern "C" int rt_scene_create_with(const rt_scene_desc* desc, int device, uint32_t build_flags, rt_scene** out) {
  if (!desc || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  rt_scene* s = new (std::nothrow) rt_scene();
  if (!s) return fail(RT_ERR_NOMEM, "out of host memory");
  int rc;
  try {
    rc = scene_create_impl(desc, device, build_flags, s);
  } catch (const std::bad_alloc&) {
    rc = fail(RT_ERR_NOMEM, "out of host memory while building the scene");
  } catch (const std::length_error& e) {
    rc = fail(RT_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) {
    rc = fail(RT_ERR_INVALID, e.what());
  }
  if (rc != RT_OK) {
    std::string keep = g_last_error;
    rt_scene_destroy(s);
    g_last_error = keep;
    return rc;
  }
  *out = s;
  return RT_OK;
}

// Builds the cones' direction index on the current device from the bands already uploaded and hangs it
// on the scene (16-bit entries when the band numbers fit).
static int attach_band_grid(rt_scene* s, size_t n_bands) {
  s->dev.band_grid_n = 0;
  s->dev.band_start = s->dev.band_items = s->dev.band_strip = nullptr;
  s->dev.band_items16 = 0;
  int n = 0;
  uint32_t *d_start = nullptr, *d_items = nullptr;
  size_t n_items = 0;
  int rc = build_band_grid_device(s->dev.bands, n_bands, (size_t)1 << 30, &n, &d_start, &d_items, &n_items);
  if (rc) return rc;
  size_t item_bytes = (n_items + 1) * 4;
  uint32_t* d_strip = nullptr;
  if (d_items && (rc = build_band_strips_device(s->dev.bands, n, d_start, d_items, n_items, &d_strip))) {
    (void)hipFree(d_items);
    if (d_start) (void)hipFree(d_start);
    return rc;
  }
  if (d_items && n_bands < 65536) {
    uint16_t* small = nullptr;
    if ((rc = compress_band_items_u16(d_items, n_items, &small))) {
      (void)hipFree(d_items);
      if (d_start) (void)hipFree(d_start);
      if (d_strip) (void)hipFree(d_strip);
      return rc;
    }
    d_items = reinterpret_cast<uint32_t*>(small);
    item_bytes = (n_items + 2) * 2;
    s->dev.band_items16 = 1;
  }
  s->dev.band_grid_n = n;
  s->dev.band_start = d_start;
  s->dev.band_items = d_items;
  if (d_start) {
    s->allocs.push_back(rt_scene::Alloc{d_start, ((size_t)3 * n * n + 1) * 4, offsetof(SceneDev, band_start)});
    s->scene_bytes += ((size_t)3 * n * n + 1) * 4;
  }
  if (d_items) {
    s->allocs.push_back(rt_scene::Alloc{d_items, item_bytes, offsetof(SceneDev, band_items)});
    s->scene_bytes += item_bytes;
  }
  s->dev.band_strip = d_strip;
  if (d_strip) {
    s->allocs.push_back(rt_scene::Alloc{d_strip, (n_items + 4) * 4, offsetof(SceneDev, band_strip)});
    s->scene_bytes += (n_items + 4) * 4;
  }
  if (getenv("RT_AMD_DEBUG"))
    fprintf(stderr, "[rt_amd] band grid n=%d entries=%zu (%d-bit) built on the device\n", n, n_items, s->dev.band_items16 ? 16 : 32);
  return RT_OK;
}

namespace {
int collect_all(rt_scene* s);  // waits for the frames in flight (defined with the pipeline below)
}

// Finds the device array a pointer of `dev` owns.
static const rt_scene::Alloc* alloc_of(const rt_scene* s, size_t field) {
  for (const rt_scene::Alloc& a : s->allocs)
    if (a.field == field) return &a;
  return nullptr;
}
template <class T>
static int reupload(rt_scene* s, size_t field, const std::vector<T>& v) {
  if (v.empty()) return RT_OK;
  const rt_scene::Alloc* a = alloc_of(s, field);
  if (!a || a->bytes != v.size() * sizeof(T)) return fail(RT_ERR_INVALID, "rt_scene_update: a table changed size");
  HIP_TRY(hipMemcpy(a->p, v.data(), a->bytes, hipMemcpyHostToDevice));
  return RT_OK;
}
#define RT_FIELD(f) offsetof(SceneDev, f)

static void box_to_f32(const double lo[3], const double hi[3], float* out6) {
  for (int a = 0; a < 3; a++) {
    float l = (float)lo[a], h = (float)hi[a];
    out6[a] = (double)l > lo[a] ? std::nextafterf(l, -INFINITY) : l;
    out6[3 + a] = (double)h < hi[a] ? std::nextafterf(h, INFINITY) : h;
  }
}

static int scene_update_impl(rt_scene* s, const rt_scene_desc* d) {
  rt_scene::Host& H = s->host;
  // ---- the shape tree must be the one the scene was built from
  if (d->n_leaves != (int32_t)H.leaves.size() || d->n_gates != (int32_t)H.gate_parent.size() || d->n_xforms != H.n_xforms ||
      d->n_triangles != H.n_triangles || d->n_materials != H.n_materials || d->n_patterns != H.n_patterns ||
      d->n_chain_mats != H.n_chain_mats || d->n_csg != H.n_csg)
    return fail(RT_ERR_INVALID, "rt_scene_update: the description has other counts than the scene was built from");
  if (H.n_csg > 0) return fail(RT_ERR_UNSUPPORTED, "rt_scene_update: scenes with CSG operators are rebuilt, not updated");
  if ((d->n_leaves > 0 && !d->leaves) || (d->n_gates > 0 && !d->gates) || (d->n_xforms > 0 && !d->xforms) ||
      (d->n_triangles > 0 && !d->triangles) || (d->n_materials > 0 && !d->materials) || (d->n_patterns > 0 && !d->patterns) ||
      (d->n_chain_mats > 0 && !d->chain_mats))
    return fail(RT_ERR_INVALID, "null array with a positive count in scene description");
  for (int i = 0; i < d->n_leaves; i++) {
    const rt_leaf &A = d->leaves[i], &B = H.leaves[(size_t)i];
    if (A.type != B.type || A.gate != B.gate || A.csg != B.csg || A.xform != B.xform || A.tri != B.tri || A.closed != B.closed)
      return fail(RT_ERR_INVALID, "rt_scene_update: the shape tree differs from the one the scene was built from");
    if (A.material < 0 || A.material >= d->n_materials) return fail(RT_ERR_INVALID, "leaf material out of range");
  }
  for (int i = 0; i < d->n_gates; i++) {
    if (d->gates[i].parent != H.gate_parent[(size_t)i])
      return fail(RT_ERR_INVALID, "rt_scene_update: the group tree differs from the one the scene was built from");
    if (!affine(d->gates[i].to_local)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) group transform");
  }
  for (int i = 0; i < d->n_xforms; i++)
    if (!affine(d->xforms[i].ray_inv) || !affine(d->xforms[i].full_inv) || !affine(d->xforms[i].fwd))
      return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) shape transform: last row must be 0 0 0 1");
  for (int i = 0; i < d->n_chain_mats; i++)
    if (!affine(d->chain_mats + 16 * (size_t)i)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) matrix in a transform chain");
  for (int pass = 0; pass < 2; pass++) {  // the chains keep their places (their matrices may change)
    const rt_chain_ref* now = pass ? d->gate_chain : d->xform_chain;
    const std::vector<rt_chain_ref>& then = pass ? H.gate_chain : H.xform_chain;
    const int n = pass ? d->n_gates : d->n_xforms;
    for (int i = 0; i < n; i++) {
      const rt_chain_ref a = now ? now[i] : rt_chain_ref{0, 0}, b = then.empty() ? rt_chain_ref{0, 0} : then[(size_t)i];
      if (a.count != b.count || (a.count > 0 && a.first != b.first))
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_update: a transform chain changed its length (an identity level became a transform or the reverse): rebuild the scene");
    }
  }
  for (int i = 0; i < d->n_patterns; i++) {
    const rt_pattern& P = d->patterns[i];
    if (!affine(P.inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) pattern transform");
    if (P.type < RT_PAT_NULL || P.type > RT_PAT_BLEND) return fail(RT_ERR_INVALID, "unknown pattern type");
    if (P.type == RT_PAT_BLEND && (P.first < 0 || P.first >= d->n_patterns || P.second < 0 || P.second >= d->n_patterns))
      return fail(RT_ERR_INVALID, "blend pattern operand out of range");
  }
  for (int i = 0; i < d->n_materials; i++)
    if (d->materials[i].pattern < -1 || d->materials[i].pattern >= d->n_patterns)
      return fail(RT_ERR_INVALID, "material pattern index out of range");

  SceneBuild b;
  b.d = d;
  // ---- every shape stays where it is: in the hierarchy, or among the shapes tested for every ray
  std::vector<float> tri_box(H.tris.size() * 6, 0.0f), quad_box(H.quads.size() * 6, 0.0f);
  std::vector<TriIsect> tris(H.tris.size());
  std::vector<QuadPrim> quads(H.quads.size()), unbounded(H.unbounded.size()), band_prims(H.band_prims.size());
  std::vector<ConeBand> bands(H.band_prims.size());
  auto box_of_leaf = [&](int li, double lo[3], double hi[3]) {
    const rt_leaf& L = d->leaves[li];
    const bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
    return leaf_box(L, d->xforms[L.xform], tri ? &d->triangles[L.tri] : nullptr, lo, hi);
  };
  for (int li = 0; li < d->n_leaves; li++) {
    if (d->leaves[li].type == RT_EMPTY) continue;
    double lo[3], hi[3];
    if (box_of_leaf(li, lo, hi) != (H.boxed[(size_t)li] != 0))
      return fail(RT_ERR_UNSUPPORTED, "rt_scene_update: a shape crossed between the hierarchy and the shapes tested for every ray "
                                      "(it became unbounded, or bounded): rebuild the scene");
  }
  for (size_t i = 0; i < tris.size(); i++) {
    const int li = H.tris[i].leaf;
    tris[i] = b.make_tri(li);
    double lo[3], hi[3];
    box_of_leaf(li, lo, hi);
    box_to_f32(lo, hi, &tri_box[i * 6]);
  }
  for (size_t i = 0; i < quads.size(); i++) {
    const int li = H.quads[i].leaf;
    quads[i] = b.make_quad(li);
    quads[i].type |= H.quads[i].type & QUAD_SKIP_PARALLEL;
    double lo[3], hi[3];
    box_of_leaf(li, lo, hi);
    box_to_f32(lo, hi, &quad_box[i * 6]);
  }
  for (size_t i = 0; i < unbounded.size(); i++) unbounded[i] = b.make_quad(H.unbounded[i].leaf);
  for (size_t i = 0; i < band_prims.size(); i++) {
    band_prims[i] = b.make_quad(H.band_prims[i].leaf);
    bands[i] = make_cone_band(band_prims[i], (uint32_t)i);
    band_prims[i].type |= QUAD_ONLY_PARALLEL;
  }
  std::vector<Gate> gates = H.gates;
  for (int gi = 0; gi < d->n_gates; gi++) {
    Gate& g = gates[(size_t)gi];
    const rt_gate& G = d->gates[gi];
    copy3x4(g.m, G.to_local);
    b.gate_chain(gi, g);
    for (int a = 0; a < 3; a++) {
      g.bmin[a] = G.bmin[a];
      g.bmax[a] = G.bmax[a];
    }
    // Whether the exact test of a group may be skipped depends on its box containing everything below,
    // which only the refit knows: from the first update on every gate is tested.
    g.skippable = 0;
  }
  ShadingTables sh;
  sh.build(d);

  // ---- nothing may be in flight while the tables change
  int rc = collect_all(s);
  if (rc) return rc;
  for (hipStream_t st : {s->stream, s->frame_stream[0], s->frame_stream[1], s->frame_stream[2], s->copy_stream}) HIP_TRY(hipStreamSynchronize(st));
  if ((rc = reupload(s, RT_FIELD(tris), tris))) return rc;
  if ((rc = reupload(s, RT_FIELD(quads), quads))) return rc;
  if ((rc = reupload(s, RT_FIELD(gates), gates))) return rc;
  if ((rc = reupload(s, RT_FIELD(unbounded), unbounded))) return rc;
  if ((rc = reupload(s, RT_FIELD(band_prims), band_prims))) return rc;
  if ((rc = reupload(s, RT_FIELD(bands), bands))) return rc;
  if ((rc = reupload(s, RT_FIELD(bands32), cone_bands_f32(bands)))) return rc;
  if ((rc = reupload(s, RT_FIELD(leaves), sh.leaves))) return rc;
  if ((rc = reupload(s, RT_FIELD(xforms), sh.xforms))) return rc;
  if ((rc = reupload(s, RT_FIELD(tri_shade), sh.tri_shade))) return rc;
  if ((rc = reupload(s, RT_FIELD(materials), sh.mats))) return rc;
  if ((rc = reupload(s, RT_FIELD(patterns), sh.pats))) return rc;
  if (s->dev.chain_mats) {
    std::vector<double> chain_mats((size_t)d->n_chain_mats * 12);
    for (int i = 0; i < d->n_chain_mats; i++) copy3x4(&chain_mats[(size_t)i * 12], d->chain_mats + 16 * (size_t)i);
    if ((rc = reupload(s, RT_FIELD(chain_mats), chain_mats))) return rc;
  }
  for (int i = 0; i < s->dev.n_inline_planes; i++) {
    const QuadPrim q = b.make_quad(H.inline_plane_leaf[(size_t)i]);
    for (int k = 0; k < 4; k++) s->dev.inline_planes[i].row[k] = q.m[4 + k];
  }
  for (int a = 0; a < 3; a++) {
    s->dev.light_intensity[a] = d->light.intensity[a];
    s->dev.light_position[a] = d->light.position[a];
  }
  round_shape_noise(d, s->dev.box_noise);
  // Not in scenes with bounded cones: their parallel-ray branch (the band list) puts hits up to 1e8 units
  // out on the cones' extensions, and a ray that starts there would, widened, meet every box of the scene
  // one after the other in a single lane (measured: 12 -> 200 ms per frame with 25 000 cones, for rays
  // whose reference results are rounding noise).  Far cameras and lights are the case this is for.
  if (s->dev.n_bands > 0) s->dev.box_noise[0] = s->dev.box_noise[1] = 0.0;
  s->max_children = sh.max_children;
  s->hint_valid = false;
  // ---- the cones' direction index follows their new matrices
  if (s->dev.band_grid_n > 0) {
    for (size_t field : {RT_FIELD(band_start), RT_FIELD(band_items), RT_FIELD(band_strip)})
      for (size_t k = 0; k < s->allocs.size(); k++)
        if (s->allocs[k].field == field) {
          s->scene_bytes -= s->allocs[k].bytes;
          (void)hipFree(s->allocs[k].p);
          s->allocs.erase(s->allocs.begin() + (long)k);
          break;
        }
    if ((rc = attach_band_grid(s, bands.size()))) return rc;
  }
  // ---- the hierarchy keeps its shape and gets new boxes
  if (H.n_nodes > 0) {
    if (!H.d_parent) {
      HIP_TRY(hipMalloc((void**)&H.d_parent, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_waits, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_arrived, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_tri_box, std::max<size_t>(1, tri_box.size()) * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&H.d_quad_box, std::max<size_t>(1, quad_box.size()) * sizeof(float)));
      HIP_TRY(hipMemcpy(H.d_parent, H.parent.data(), H.n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(H.d_waits, H.waits.data(), H.n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!tri_box.empty()) HIP_TRY(hipMemcpy(H.d_tri_box, tri_box.data(), tri_box.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!quad_box.empty()) HIP_TRY(hipMemcpy(H.d_quad_box, quad_box.data(), quad_box.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(H.d_arrived, 0, H.n_nodes * sizeof(uint32_t), main_stream(s)));
    RefitView v{const_cast<BvhNode*>(s->dev.nodes), H.d_tri_box, H.d_quad_box, s->dev.gates, H.d_parent, H.d_waits, H.d_arrived, H.n_nodes};
    hipLaunchKernelGGL(k_refit, dim3((H.n_nodes + 255) / 256), dim3(256), 0, main_stream(s), v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
  }
  s->dev.root_alt = s->dev.root;  // with every gate tested, both kinds of ray walk the same references
  H.tris.swap(tris);
  H.quads.swap(quads);
  H.unbounded.swap(unbounded);
  H.band_prims.swap(band_prims);
  H.gates.swap(gates);
  H.updated = true;
  return RT_OK;
}
#undef RT_FIELD

extern "C" int rt_scene_update(rt_scene* s, const rt_scene_desc* desc) {
  if (!s || !desc) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  try {
    return scene_update_impl(s, desc);
  } catch (const std::bad_alloc&) {
    return fail(RT_ERR_NOMEM, "out of host memory while updating the scene");
  } catch (const std::exception& e) {
    return fail(RT_ERR_INVALID, e.what());
  }
}

// A second copy of a scene on another GPU (or the same one): the acceleration structure is built once
// and its arrays are copied device to device.
extern "C" int rt_scene_clone(const rt_scene* src, int device, rt_scene** out) {
  if (!src || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  rt_scene* s = new (std::nothrow) rt_scene();
  if (!s) return fail(RT_ERR_NOMEM, "out of host memory");
  auto work = [&]() -> int {
    int rc = scene_runtime_init(s, device);
    if (rc) return rc;
    s->dev = src->dev;
    for (const rt_scene::Alloc& a : src->allocs) *(const void**)((char*)&s->dev + a.field) = nullptr;
    for (const rt_scene::Alloc& a : src->allocs) {
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, a.bytes));
      s->allocs.push_back(rt_scene::Alloc{p, a.bytes, a.field});
      HIP_TRY(hipMemcpyPeer(p, device, a.p, src->device, a.bytes));
      *(const void**)((char*)&s->dev + a.field) = p;
    }
    HIP_TRY(hipDeviceSynchronize());
    s->scene_bytes = src->scene_bytes;
    s->build_seconds = 0;
    s->stack_need = src->stack_need;
    s->has_csg = src->has_csg;
    s->uniform_nodes = src->uniform_nodes;
    s->tri_prefetch = src->tri_prefetch;
    s->max_children = src->max_children;
    s->host = src->host;  // vectors copied; the refit's device arrays are made on the first update
    s->host.d_tri_box = s->host.d_quad_box = nullptr;
    s->host.d_parent = s->host.d_waits = s->host.d_arrived = nullptr;
    return RT_OK;
  };
  int rc = work();
  if (rc != RT_OK) {
    std::string keep = g_last_error;
    rt_scene_destroy(s);
    g_last_error = keep;
    return rc;
  }
  *out = s;
  return RT_OK;
}

// Host-only self-check of the cone direction index: n_bands random cone bands, n_dirs random unit
// directions (half of them placed inside a band); returns how many (direction, band) pairs that pass
// the band pre-filter are missing from the direction's grid cell (0 when the index is conservative),
// or a negative rt_status.  *grid_n receives the grid resolution, *entries its size.
extern "C" int rt_selftest_bvh_build(int n_items, uint64_t seed, int threads, uint64_t* hash, uint64_t* n_nodes,
                                     double* seconds) {
  if (n_items < 0 || threads < 0 || !hash) return fail(RT_ERR_INVALID, "rt_selftest_bvh_build: bad argument");
  uint64_t state = seed;
  auto uniform = [&state]() {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
  };
  std::vector<BuildItem> items((size_t)n_items);
  for (int i = 0; i < n_items; i++) {
    BuildItem& it = items[i];
    for (int a = 0; a < 3; a++) {
      double c = uniform() * 100.0, e = uniform() * 0.5;
      it.lo[a] = c - e;
      it.hi[a] = c + e;
    }
    it.kind = uniform() < 0.7 ? KIND_TRI : KIND_QUAD;
    it.index = (uint32_t)i;
  }
  BvhBuilder b;
  b.threads = threads;
  double lo[3], hi[3];
  auto t0 = std::chrono::steady_clock::now();
  uint32_t root = b.build(items, lo, hi);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t h = 0xCBF29CE484222325ull ^ root;  // FNV-1a over everything the build produced
  auto mix = [&h](const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 0x100000001B3ull;
  };
  if (!b.nodes.empty()) mix(b.nodes.data(), b.nodes.size() * sizeof(BvhNode));
  if (!b.tri_order.empty()) mix(b.tri_order.data(), b.tri_order.size() * sizeof(uint32_t));
  if (!b.quad_order.empty()) mix(b.quad_order.data(), b.quad_order.size() * sizeof(uint32_t));
  mix(lo, sizeof lo);
  mix(hi, sizeof hi);
  mix(&b.stack_need, sizeof b.stack_need);
  *hash = h;
  if (n_nodes) *n_nodes = b.nodes.size();
  return RT_OK;
}

extern "C" int rt_selftest_band_grid(int n_bands, int n_dirs, uint64_t seed, int* grid_n, uint64_t* entries) {
  if (n_bands < 0 || n_dirs < 0) return fail(RT_ERR_INVALID, "negative count");
  uint64_t state = seed;
  auto next = [&]() {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto uni = [&](double lo, double hi) { return lo + (hi - lo) * ((next() >> 11) * (1.0 / 9007199254740992.0)); };
  std::vector<ConeBand> bands(n_bands);
  std::vector<std::array<double, 9>> mats(n_bands);
  for (int i = 0; i < n_bands; i++) {
    // rows of a random rotation (from a random quaternion) divided by random scales: world -> object
    double qv[4], nq = 0;
    for (double& v : qv) { v = uni(-1, 1); nq += v * v; }
    nq = sqrt(nq);
    double w = qv[0] / nq, x = qv[1] / nq, y = qv[2] / nq, z = qv[3] / nq;
    double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                   2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    double sc[3] = {uni(0.2, 1.0), uni(0.2, 1.0), uni(0.2, 1.0)};
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) mats[i][r * 3 + c] = R[c * 3 + r] / sc[r];
    const double sg[3] = {1.0, -1.0, 1.0};
    double Q[3][3], sum = 0;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        Q[a][b] = 0;
        for (int r = 0; r < 3; r++) Q[a][b] += sg[r] * mats[i][r * 3 + a] * mats[i][r * 3 + b];
      }
    for (double v : mats[i]) sum += v * v;
    ConeBand& cb = bands[i];
    cb.q[0] = Q[0][0]; cb.q[1] = Q[1][1]; cb.q[2] = Q[2][2];
    cb.q[3] = 2 * Q[0][1]; cb.q[4] = 2 * Q[0][2]; cb.q[5] = 2 * Q[1][2];
    cb.slack = 1e-11 * sum;
    cb.prim = (uint32_t)i;
  }
  BandGrid grid = build_band_grid(bands, (size_t)1 << 28);
  if (grid_n) *grid_n = grid.n;
  if (entries) *entries = grid.items.size();
  if (grid.n == 0) return 0;
  int missing = 0;
  const int n = grid.n;
  for (int k = 0; k < n_dirs; k++) {
    double d[3];
    if ((k & 1) && n_bands > 0) {
      // a direction on the surface of cone i: object direction (cos p, +-1, sin p) mapped to world
      int i = (int)(next() % (uint64_t)n_bands);
      double p = uni(0, 6.283185307179586), ol[3] = {cos(p), (next() & 1) ? 1.0 : -1.0, sin(p)};
      // world = M^-1 object; M = diag(1/s) R^T, so M^-1 = R diag(s): solve by Cramer on the 3x3
      const double* m = mats[i].data();
      double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
      double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                       (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                       (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
      for (int r = 0; r < 3; r++) d[r] = inv[r * 3] * ol[0] + inv[r * 3 + 1] * ol[1] + inv[r * 3 + 2] * ol[2];
      if (next() & 1) {
        // pushed across the band along the gradient of a, to anywhere between its two edges
        const ConeBand& cb = bands[i];
        double l = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (double& v : d) v /= l;
        const double gr[3] = {2 * cb.q[0] * d[0] + cb.q[3] * d[1] + cb.q[4] * d[2], 2 * cb.q[1] * d[1] + cb.q[3] * d[0] + cb.q[5] * d[2],
                              2 * cb.q[2] * d[2] + cb.q[4] * d[0] + cb.q[5] * d[1]};
        const double g2 = gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2], target = uni(-0.999, 0.999) * FEQ_EPSILON;
        if (g2 > 1e-12)
          for (int r = 0; r < 3; r++) d[r] += target / g2 * gr[r];
      }
    } else {
      for (double& v : d) v = uni(-1, 1);
    }
    double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(len > 1e-3)) continue;
    for (double& v : d) v /= len;
    const double xx = d[0] * d[0], yy = d[1] * d[1], zz = d[2] * d[2], xy = d[0] * d[1], xz = d[0] * d[2], yz = d[1] * d[2];
    const double dd = xx + yy + zz;
    int m = zz > yy ? 2 : 1;
    m = xx >= (m == 2 ? zz : yy) ? 0 : m;
    const double dm = d[m], da = d[(m + 1) % 3], dc = d[(m + 2) % 3];
    const float inv = 1.0f / (float)dm, half_n = 0.5f * (float)n;
    int iu, iv;
    float ls, lt;  // the cell and the local coordinates as the traversal computes them
    band_strip_coords(((float)da * inv + 1.0f) * half_n, ((float)dc * inv + 1.0f) * half_n, n, iu, iv, ls, lt);
    const uint32_t cell = ((uint32_t)m * n + iv) * n + iu;
    for (int i = 0; i < n_bands; i++) {
      const ConeBand& cb = bands[i];
      double a = cb.q[0] * xx + cb.q[1] * yy + cb.q[2] * zz + cb.q[3] * xy + cb.q[4] * xz + cb.q[5] * yz;
      if (!(fabs(a) < FEQ_EPSILON * (1.0 + 1e-6) + cb.slack * dd)) continue;
      auto at = std::lower_bound(grid.items.begin() + grid.start[cell], grid.items.begin() + grid.start[cell + 1], (uint32_t)i);
      bool found = at != grid.items.begin() + grid.start[cell + 1] && *at == (uint32_t)i;
      // listed in the cell, and the entry's strip holds the direction
      if (!found || !band_strip_passes(grid.strip[(size_t)(at - grid.items.begin())], ls, lt, ls * 1024.0f, lt * 1024.0f)) missing++;
    }
  }
  return missing;
}

extern "C" int rt_device_alloc(int device, uint64_t bytes, void** out) {
  if (!out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  int previous = -1;
  HIP_TRY(hipGetDevice(&previous));
  HIP_TRY(hipSetDevice(device));
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e == hipSuccess) e = hipMemset(p, 0, bytes);
  (void)hipSetDevice(previous);
  if (e != hipSuccess) {
    if (p) (void)hipFree(p);
    return fail(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, std::string("rt_device_alloc: ") + hipGetErrorString(e));
  }
  *out = p;
  return RT_OK;
}

extern "C" void rt_device_free(int device, void* p) {
  if (!p) return;
  int previous = -1;
  (void)hipGetDevice(&previous);
  (void)hipSetDevice(device);
  (void)hipFree(p);
  if (previous >= 0) (void)hipSetDevice(previous);
}

extern "C" void* rt_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

extern "C" void rt_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int rt_host_register(void* p, uint64_t bytes) {
  if (!p || bytes == 0) return fail(RT_ERR_INVALID, "rt_host_register: null or empty range");
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable));
  return RT_OK;
}

extern "C" int rt_host_unregister(void* p) {
  if (!p) return RT_OK;
  HIP_TRY(hipHostUnregister(p));
  return RT_OK;
}

extern "C" int rt_scene_build_info(const rt_scene* s, uint64_t* device_builds, double* device_build_seconds) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  if (device_builds) *device_builds = s->device_builds;
  if (device_build_seconds) *device_build_seconds = s->device_build_seconds;
  return RT_OK;
}

extern "C" int rt_scene_footprint(const rt_scene* s, uint64_t* scene_bytes, double* build_seconds) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  if (scene_bytes) *scene_bytes = s->scene_bytes;
  if (build_seconds) *build_seconds = s->build_seconds;
  return RT_OK;
}

// ---------------------------------------------------------------------------
// host side: the generation pipeline
// ---------------------------------------------------------------------------
namespace {

int ensure_gen(rt_scene* s, int g, size_t cap) {
  GenBuf& b = s->gens[g];
  if (cap == 0) cap = 64;
  if (b.capacity >= cap) return RT_OK;
  if (b.slab) {
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    HIP_TRY(hipFree(b.slab));
    b.slab = nullptr;
    b.capacity = 0;
  }
  HIP_TRY(hipMalloc(&b.slab, gen_bytes(cap)));
  b.capacity = cap;
  carve(b);
  return RT_OK;
}

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }

struct GenLaunch {
  GenView cur, next;
  uint32_t n, next_segcap;  // threads to launch; record slots per segment of the next generation's queue
  int remaining;
  uint32_t* next_counts;    // QUEUE_SEGMENTS counters
  CameraDev cam;
  const uint32_t* tiles;
  uint32_t tiles_x;
  ImageOut img;
  unsigned long long* stats;
  hipStream_t stream;
  const uint32_t* queue_counts = nullptr;  // the segment counters this generation's rays were queued under
  uint32_t segcap = 0;                     // (null: the rays are dense, n of them)
};

inline unsigned csg_lds_bytes(const rt_scene* s) { return (unsigned)s->dev.csg_hits * 3u * BLOCK * (unsigned)sizeof(uint32_t); }

template <int STACK, bool COUNT, bool CSG, bool UNIFORM = false, bool TRIPF = false>
void launch_generation_mode(rt_scene* s, int mode, const GenLaunch& a) {
  dim3 grid(blocks_for(a.n)), block(BLOCK);
  static const unsigned env_lds = getenv("RT_AMD_EXTRA_LDS") ? (unsigned)atoi(getenv("RT_AMD_EXTRA_LDS")) : 0u;
  const unsigned extra_lds = (CSG ? csg_lds_bytes(s) : 0u) + env_lds;
#define RT_GEN_ARGS s->dev, a.cur, a.n, a.next, a.next_segcap, a.remaining, a.next_counts, a.stats, a.cam, a.tiles, a.tiles_x, a.img, a.queue_counts, a.segcap
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_generation<STACK, COUNT, 0, CSG, UNIFORM, TRIPF>), grid, block, extra_lds, a.stream, RT_GEN_ARGS); break;
    case 1: hipLaunchKernelGGL((k_generation<STACK, COUNT, 1, CSG, UNIFORM>), grid, block, extra_lds, a.stream, RT_GEN_ARGS); break;
    default: hipLaunchKernelGGL((k_generation<STACK, COUNT, 2, CSG, UNIFORM>), grid, block, extra_lds, a.stream, RT_GEN_ARGS); break;
  }
#undef RT_GEN_ARGS
}

void launch_generation(rt_scene* s, int mode, bool count, const GenLaunch& a) {
  if (s->has_csg) {  // scenes with CSG operators: the kernels that carry eval_csg (64-entry stack)
    if (count) launch_generation_mode<64, true, true>(s, mode, a);
    else launch_generation_mode<64, false, true>(s, mode, a);
  } else if (pick_stack(s->stack_need) == STACK_SMALL) {
    if (count) launch_generation_mode<STACK_SMALL, true, false>(s, mode, a);
    else if (s->uniform_nodes && s->tri_prefetch && mode == 0) launch_generation_mode<STACK_SMALL, false, false, true, true>(s, mode, a);
    else if (s->uniform_nodes) launch_generation_mode<STACK_SMALL, false, false, true>(s, mode, a);
    else launch_generation_mode<STACK_SMALL, false, false>(s, mode, a);
  } else {
    if (count) launch_generation_mode<64, true, false>(s, mode, a);
    else if (s->uniform_nodes && s->tri_prefetch && mode == 0) launch_generation_mode<64, false, false, true, true>(s, mode, a);
    else if (s->uniform_nodes) launch_generation_mode<64, false, false, true>(s, mode, a);
    else launch_generation_mode<64, false, false>(s, mode, a);
  }
}

CameraDev camera_dev(const rt_camera* c) {
  CameraDev cd;
  cd.hsize = c->hsize;
  cd.vsize = c->vsize;
  cd.half_width = c->half_width;
  cd.half_height = c->half_height;
  cd.pixel_size = c->pixel_size;
  copy3x4(cd.inv, c->inv);
  return cd;
}

struct PipelineIn {
  uint32_t n0 = 0;
  int remaining0 = RECURSION_DEPTH;
  uint32_t flags = 0;
  bool camera = false;  // generation 0 from the camera (else from gens[0] ray arrays)
  CameraDev cam{};
  const uint32_t* tiles = nullptr;
  uint32_t tiles_x = 0;
  ImageOut img{nullptr, nullptr, nullptr, 0};
};

void decode_stats(const unsigned long long* h, rt_stats* stats) {
  unsigned long long sums[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < STAT_SLOTS; j++) sums[k] += h[STAT_RAYS + k * STAT_SLOTS + j];
  stats->shadow = sums[0];
  stats->reflect = sums[1];
  stats->refract = sums[2];
  stats->node_visits = h[0];
  stats->tri_tests = h[1];
  stats->quadric_tests = h[2];
  stats->gate_tests = h[3];
  stats->shadow_node_visits = h[4];
  stats->shadow_tri_tests = h[5];
  stats->shadow_quadric_tests = h[6];
  stats->shadow_gate_tests = h[7];
}

// Waits for one asynchronous frame and adds its kernel time to the scene's running total.
int collect_frame(rt_scene* s, rt_scene::FrameSlot& f) {
  if (!f.pending) return RT_OK;
  HIP_TRY(hipEventSynchronize(f.end));
  f.pending = false;
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, f.begin, f.end));
  rt_stats& t = s->async_total;
  t.primary += f.primary;
  t.kernel_ms += ms;
  t.kernel_launches += 1;
  t.generations += 1;
  return RT_OK;
}

// Waits for every asynchronous frame, then reads and clears the counters they added to.
int collect_all(rt_scene* s) {
  for (int k = 0; k < rt_scene::FRAME_SLOTS; k++) {
    int rc = collect_frame(s, s->slots[(s->next_slot + k) % rt_scene::FRAME_SLOTS]);
    if (rc) return rc;
  }
  if (!s->async_counters_dirty) return RT_OK;
  HIP_TRY(hipMemcpy(s->h_async_stats, s->d_async_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(s->d_async_stats, 0, STAT_WORDS * sizeof(unsigned long long)));
  s->async_counters_dirty = false;
#ifdef RT_STAMPS
  {
    const unsigned long long* t = s->h_async_stats + STAT_RAYS + 3 * STAT_SLOTS;
    double sum = 0;
    for (int k = 0; k < 5; k++) sum += (double)t[k];
    fprintf(stderr, "[rt_amd] wave clock shares: raygen %.3f closest %.3f shade %.3f shadow %.3f output %.3f (total %.3g ticks)\n",
            t[0] / sum, t[1] / sum, t[2] / sum, t[3] / sum, t[4] / sum, sum);
    fprintf(stderr, "[rt_amd]   closest: setup %.3f nodes %.3f tris %.3f quads/gates %.3f | shadow: setup %.3f nodes %.3f tris %.3f quads/gates %.3f\n",
            t[8] / sum, t[9] / sum, t[10] / sum, t[11] / sum, t[16] / sum, t[17] / sum, t[18] / sum, t[19] / sum);
    fprintf(stderr, "[rt_amd]   wave-level loop iterations: closest nodes %llu tris %llu quads %llu rounds %llu | shadow nodes %llu tris %llu quads %llu rounds %llu\n",
            t[12], t[13], t[14], t[15], t[20], t[21], t[22], t[23]);
  }
#endif
  rt_stats one{};
  decode_stats(s->h_async_stats, &one);
  rt_stats& t = s->async_total;
  t.shadow += one.shadow; t.reflect += one.reflect; t.refract += one.refract;
  t.node_visits += one.node_visits; t.tri_tests += one.tri_tests;
  t.quadric_tests += one.quadric_tests; t.gate_tests += one.gate_tests;
  t.shadow_node_visits += one.shadow_node_visits; t.shadow_tri_tests += one.shadow_tri_tests;
  t.shadow_quadric_tests += one.shadow_quadric_tests; t.shadow_gate_tests += one.shadow_gate_tests;
  return RT_OK;
}

// One camera frame of a scene that spawns no rays: a single kernel, nothing to read back, so the
// frame is only enqueued.
int enqueue_direct_frame(rt_scene* s, const PipelineIn& in, uint64_t primary) {
  rt_scene::FrameSlot& f = s->slots[s->next_slot];
  int rc = collect_frame(s, f);  // the slot's previous frame, FRAME_SLOTS frames ago
  if (rc) return rc;
  // consecutive frames go to different streams: they do not depend on each other, and the tail of
  // one kernel (few waves left) overlaps the head of the next
  // a whole frame fills the GPU from two streams (a third only stretches every launch); a part of a frame does not
  const unsigned lanes = in.tiles ? (unsigned)s->n_frame_streams : std::min(2u, (unsigned)s->n_frame_streams);
  const unsigned lane = s->frame_seq++ % lanes;
  hipStream_t stream = lane ? s->frame_stream[lane - 1] : main_stream(s);
  if (lane) s->frame_stream_used[lane - 1] = true;
  s->last_stream = stream;
  s->next_slot = (s->next_slot + 1) % rt_scene::FRAME_SLOTS;
  GenLaunch a;
  a.cur = s->gens[0].view;
  a.next = s->gens[1].view;
  a.n = in.n0;
  a.next_segcap = 0;
  a.remaining = 0;
  a.next_counts = s->d_counts;
  a.cam = in.cam;
  a.tiles = in.tiles;
  a.tiles_x = in.tiles_x;
  a.img = in.img;
  a.stats = s->d_async_stats;
  s->async_counters_dirty = true;
  a.stream = stream;
  HIP_TRY(hipEventRecord(f.begin, stream));
  launch_generation(s, 0, (in.flags & RT_FLAG_COUNTERS) != 0, a);
  HIP_TRY(hipEventRecord(f.end, stream));
  HIP_TRY(hipGetLastError());
  f.primary = primary;
  f.pending = true;
  return RT_OK;
}

// Record slots per queue segment that can never overflow: the waves of a generation of n rays are dealt
// to the segments round-robin and a wave spawns at most 64 * max_children rays.
inline uint64_t safe_segcap(uint64_t n, uint32_t max_children) {
  const uint64_t waves = (n + 63) / 64;
  return (waves + QUEUE_SEGMENTS - 1) / QUEUE_SEGMENTS * 64 * max_children;
}

struct GenPlan {
  uint32_t threads[MAX_GENERATIONS + 2] = {0};  // grid size of generation g (for g > 0 an upper bound when speculative)
  uint32_t segcap[MAX_GENERATIONS + 2] = {0};   // record slots per queue segment of generation g (g > 0)
};

void fill_launch(rt_scene* s, const PipelineIn& in, const GenPlan& plan, int g, int max_gen, GenLaunch& a) {
  a.cur = s->gens[g].view;
  a.next = s->gens[g + 1].view;
  a.n = plan.threads[g];
  a.next_segcap = g < max_gen ? plan.segcap[g + 1] : 0;
  a.remaining = g < max_gen ? std::max(in.remaining0 - g, 0) : 0;
  a.next_counts = s->d_counts + g * QUEUE_SEGMENTS;
  a.cam = in.cam;
  a.tiles = in.tiles;
  a.tiles_x = in.tiles_x;
  a.img = in.img;
  a.stats = s->d_stats;
  a.stream = main_stream(s);
  a.queue_counts = g > 0 ? s->d_counts + (g - 1) * QUEUE_SEGMENTS : nullptr;
  a.segcap = plan.segcap[g];
}

void launch_resolve(rt_scene* s, const GenPlan& plan, int g) {
  hipLaunchKernelGGL(k_resolve, dim3(blocks_for(plan.threads[g])), dim3(BLOCK), 0, main_stream(s), s->gens[g].view,
                     plan.threads[g], s->gens[g + 1].view,
                     g > 0 ? s->d_counts + (g - 1) * QUEUE_SEGMENTS : (const uint32_t*)nullptr, plan.segcap[g]);
}

// Colours flow back from the last generation to the first; a camera frame ends in its pixels.
void launch_resolves(rt_scene* s, const PipelineIn& in, const GenPlan& plan, int last) {
  for (int g = last - 1; g >= (in.camera ? 1 : 0); g--) launch_resolve(s, plan, g);
  if (in.camera && last > 0)
    hipLaunchKernelGGL(k_resolve_pixels, dim3(blocks_for(in.n0)), dim3(BLOCK), 0, main_stream(s), s->gens[0].view, in.n0,
                       s->gens[1].view, in.cam, in.tiles, in.tiles_x, in.img);
}

// The frame launched at once on twice the queue lengths of the frame before (scene->hint; unused room
// costs waves that exit at once):
// no host round trip between the generations; the kernels find their rays through the segment counters
// on the device.  The counters come back in one copy at the end.  *fits = false when a segment
// received more rays than it had room for: the frame's output is then unusable.
// Sizes the generations from the hints and makes room for them; returns the last generation to launch.
int plan_speculative(rt_scene* s, const PipelineIn& in, int max_gen, GenPlan& plan, int* last_out) {
  plan.threads[0] = in.n0;
  int last = 0;
  for (int g = 1; g <= max_gen && s->hint[g] > 0; g++) {
    const uint64_t safe = safe_segcap(plan.threads[g - 1], s->max_children);
    const uint64_t want = 2 * (uint64_t)s->hint[g] + 256;
    plan.segcap[g] = (uint32_t)std::min(safe, want);
    plan.threads[g] = plan.segcap[g] * QUEUE_SEGMENTS;
    last = g;
  }
  int rc;
  for (int g = 0; g <= last + 1 && g <= MAX_GENERATIONS; g++)
    if ((rc = ensure_gen(s, g, g == 0 ? in.n0 : (size_t)plan.segcap[g] * QUEUE_SEGMENTS))) return rc;
  *last_out = last;
  return RT_OK;
}

int run_speculative(rt_scene* s, const PipelineIn& in, int max_gen, rt_stats* stats, bool* fits) {
  GenPlan plan;
  int last = 0, rc;
  if ((rc = plan_speculative(s, in, max_gen, plan, &last))) return rc;
  HIP_TRY(hipEventRecord(s->ev[0], main_stream(s)));
  for (int g = 0; g <= last; g++) {
    GenLaunch a;
    fill_launch(s, in, plan, g, max_gen, a);
    launch_generation(s, (g == 0 && in.camera) ? 1 : 2, false, a);
  }
  HIP_TRY(hipEventRecord(s->ev[1], main_stream(s)));
  launch_resolves(s, in, plan, last);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_counts, s->d_counts, COUNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                         main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  *fits = true;
  int true_last = 0;
  for (int g = 1; g <= max_gen && *fits; g++) {  // counters [g - 1]: the rays queued for generation g
    uint32_t fullest = 0;
    for (int k = 0; k < QUEUE_SEGMENTS; k++) fullest = std::max(fullest, s->h_counts[(g - 1) * QUEUE_SEGMENTS + k]);
    if (fullest > plan.segcap[g]) *fits = false;
    if (fullest == 0) break;
    true_last = g;
  }
  if (!*fits) return RT_OK;
  for (int g = 1; g <= MAX_GENERATIONS + 1; g++) {
    uint32_t fullest = 0;
    if (g <= true_last)
      for (int k = 0; k < QUEUE_SEGMENTS; k++) fullest = std::max(fullest, s->h_counts[(g - 1) * QUEUE_SEGMENTS + k]);
    s->hint[g] = fullest;
  }
  if (stats) {
    decode_stats(s->h_stats, stats);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    stats->kernel_ms = ms;
    stats->kernel_launches = (uint64_t)true_last + 1;
    stats->generations = (uint64_t)true_last + 1;
  }
  return RT_OK;
}

// Runs the generations, resolves colours back into generation 0 and, for camera
// rays, writes the image.  Fills the device-side fields of stats.
int run_pipeline(rt_scene* s, const PipelineIn& in, rt_stats* stats) {
  const bool count = (in.flags & RT_FLAG_COUNTERS) != 0;
  int max_gen = std::min(std::max(in.remaining0, 0), RECURSION_DEPTH);
  if (s->max_children == 0) max_gen = 0;
  const bool direct = in.camera && max_gen == 0;  // every pixel is final after generation 0
  static const bool no_speculation = getenv("RT_AMD_NO_SPECULATION") != nullptr;
  const bool hinted = max_gen > 0 && !count && !no_speculation && s->hint_valid && s->hint_key[0] == in.n0 &&
                      s->hint_key[1] == (uint32_t)in.remaining0;
  if (hinted && s->holdoff > 0) {
    s->holdoff--;
  } else if (hinted) {
    HIP_TRY(hipMemsetAsync(s->d_stats, 0, STAT_WORDS * sizeof(unsigned long long), main_stream(s)));
    HIP_TRY(hipMemsetAsync(s->d_counts, 0, COUNT_WORDS * sizeof(uint32_t), main_stream(s)));
    bool fits = false;
    int rc = run_speculative(s, in, max_gen, stats, &fits);
    if (rc) return rc;
    if (fits) {
      s->holdoff_next = 1;
      return RT_OK;
    }
    // start over, one generation at a time, and stay exact for a while: a frame that does not fit is
    // rendered twice, and a camera that moves this fast will do it again
    s->holdoff = s->holdoff_next;
    s->holdoff_next = std::min(2 * s->holdoff_next, 64u);
    static const bool debug = getenv("RT_AMD_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "[rt_amd] speculative frame did not fit; exact for the next %u frames\n", s->holdoff);
  }
  HIP_TRY(hipMemsetAsync(s->d_stats, 0, STAT_WORDS * sizeof(unsigned long long), main_stream(s)));
  if (max_gen > 0) HIP_TRY(hipMemsetAsync(s->d_counts, 0, COUNT_WORDS * sizeof(uint32_t), main_stream(s)));
  GenPlan plan;
  uint32_t fullest[MAX_GENERATIONS + 2] = {0};
  plan.threads[0] = in.n0;
  int last = 0;
  for (int g = 0; g <= max_gen && plan.threads[g] > 0; g++) {
    last = g;
    const uint64_t next_segcap = g < max_gen ? safe_segcap(plan.threads[g], s->max_children) : 0;
    if (next_segcap * QUEUE_SEGMENTS > 0xFFFFFF00ull)
      return fail(RT_ERR_UNSUPPORTED, "too many secondary rays for one generation");
    plan.segcap[g + 1] = (uint32_t)next_segcap;
    int rc;
    if (!direct && g == 0 && (rc = ensure_gen(s, 0, in.n0))) return rc;
    if ((rc = ensure_gen(s, g + 1, (size_t)next_segcap * QUEUE_SEGMENTS))) return rc;
    GenLaunch a;
    fill_launch(s, in, plan, g, max_gen, a);
    int mode = (g == 0 && in.camera) ? (direct ? 0 : 1) : 2;
    HIP_TRY(hipEventRecord(s->ev[g * 2 + 0], main_stream(s)));
    launch_generation(s, mode, count, a);
    HIP_TRY(hipEventRecord(s->ev[g * 2 + 1], main_stream(s)));
    if (g < max_gen) {
      uint32_t* h = s->h_counts + g * QUEUE_SEGMENTS;
      HIP_TRY(hipMemcpyAsync(h, s->d_counts + g * QUEUE_SEGMENTS, QUEUE_SEGMENTS * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             main_stream(s)));
      HIP_TRY(hipStreamSynchronize(main_stream(s)));
      uint64_t total = 0;
      for (int k = 0; k < QUEUE_SEGMENTS; k++) {
        total += std::min<uint32_t>(h[k], (uint32_t)next_segcap);
        fullest[g + 1] = std::max(fullest[g + 1], h[k]);
      }
      plan.threads[g + 1] = (uint32_t)total;
    }
#ifdef RT_STAMPS
    {
      static unsigned long long before[STAMP_WORDS];
      if (g == 0) memset(before, 0, sizeof before);
      HIP_TRY(hipStreamSynchronize(main_stream(s)));  // the generation just launched (the copy below does not wait for this stream)
      HIP_TRY(hipMemcpy(s->h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      const unsigned long long* t = s->h_stats + STAT_RAYS + 3 * STAT_SLOTS;
      double d[STAMP_WORDS], sum = 0;
      for (int k = 0; k < STAMP_WORDS; k++) { d[k] = (double)(t[k] - before[k]); before[k] = t[k]; }
      for (int k = 0; k < 5; k++) sum += d[k];
      const double waves = (double)blocks_for(plan.threads[g]) * (BLOCK / 64);
      fprintf(stderr, "[rt_amd] generation %d rays %u: ticks per wave %.0f | raygen %.3f closest %.3f shade %.3f shadow %.3f output %.3f\n",
              g, plan.threads[g], sum / waves, d[0] / sum, d[1] / sum, d[2] / sum, d[3] / sum, d[4] / sum);
      fprintf(stderr, "[rt_amd]   closest: setup %.3f nodes %.3f tris %.3f quads/gates %.3f | shadow: setup %.3f nodes %.3f tris %.3f quads/gates %.3f\n",
              d[8] / sum, d[9] / sum, d[10] / sum, d[11] / sum, d[16] / sum, d[17] / sum, d[18] / sum, d[19] / sum);
      fprintf(stderr, "[rt_amd]   wave-level loop iterations: closest nodes %.0f tris %.0f quads %.0f rounds %.0f | shadow nodes %.0f tris %.0f quads %.0f rounds %.0f (waves %.0f)\n",
              d[12], d[13], d[14], d[15], d[20], d[21], d[22], d[23], waves);
    }
#endif
  }
  launch_resolves(s, in, plan, last);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                         main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  if (max_gen > 0) {
    for (int g = 0; g <= MAX_GENERATIONS + 1; g++) s->hint[g] = fullest[g];
    s->hint_key[0] = in.n0;
    s->hint_key[1] = (uint32_t)in.remaining0;
    s->hint_valid = true;
    if (!count && !no_speculation) {  // room for the next frame's queues now, not inside that frame
      GenPlan next_plan;
      int next_last = 0, rc;
      if ((rc = plan_speculative(s, in, max_gen, next_plan, &next_last))) return rc;
    }
  }
  if (stats) {
    decode_stats(s->h_stats, stats);
    stats->kernel_ms = 0;
    stats->kernel_launches = 0;
    stats->generations = (uint64_t)last + 1;
    for (int g = 0; g <= last; g++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, s->ev[g * 2 + 0], s->ev[g * 2 + 1]));
      stats->kernel_ms += ms;
      stats->kernel_launches++;
    }
  }
  return RT_OK;
}

int stage_reserve(rt_scene* s, size_t bytes) {
  if (s->stage_bytes >= bytes) return RT_OK;
  if (s->stage) {
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    HIP_TRY(hipFree(s->stage));
    s->stage = nullptr;
    s->stage_bytes = 0;
  }
  HIP_TRY(hipMalloc(&s->stage, bytes));
  s->stage_bytes = bytes;
  return RT_OK;
}

}  // namespace

extern "C" int rt_render(rt_scene* s, const rt_camera* camera, int part, int n_parts, uint32_t flags,
                         uint8_t* rgb8_dev, double* rgb64_dev, int32_t* hit_leaf_dev, rt_stats* stats) {
  if (!s || !camera) return fail(RT_ERR_INVALID, "null argument");
  if (n_parts < 1 || part < 0 || part >= n_parts) return fail(RT_ERR_INVALID, "part must lie in [0, n_parts)");
  if (camera->hsize == 0 || camera->vsize == 0) return fail(RT_ERR_INVALID, "empty canvas");
  if (!affine(camera->inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) camera transform");
  auto w0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  if (stats) memset(stats, 0, sizeof(*stats));
  uint32_t tiles_x = (camera->hsize + TILE - 1) / TILE, tiles_y = (camera->vsize + TILE - 1) / TILE;
  const bool bands = (flags & RT_FLAG_BANDS) != 0;
  const uint32_t key[4] = {camera->hsize, camera->vsize, (uint32_t)part | (bands ? 0x80000000u : 0u), (uint32_t)n_parts};
  if (memcmp(key, s->tiles_key, sizeof(key)) != 0) {
    int rcw = collect_all(s);  // frames in flight still read the current tile table
    if (rcw) return rcw;
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    s->h_tiles.clear();
    for (uint32_t ty = 0; ty < tiles_y; ty++)
      for (uint32_t tx = 0; tx < tiles_x; tx++)
        if ((int)(((bands ? 0u : tx) + ty) % (uint32_t)n_parts) == part) s->h_tiles.push_back(ty * tiles_x + tx);
    if (s->tiles_capacity < s->h_tiles.size()) {
      if (s->d_tiles) {
        HIP_TRY(hipStreamSynchronize(main_stream(s)));
        HIP_TRY(hipFree(s->d_tiles));
        s->d_tiles = nullptr;
      }
      HIP_TRY(hipMalloc((void**)&s->d_tiles, s->h_tiles.size() * sizeof(uint32_t)));
      s->tiles_capacity = s->h_tiles.size();
    }
    if (!s->h_tiles.empty())
      HIP_TRY(hipMemcpy(s->d_tiles, s->h_tiles.data(), s->h_tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    memcpy(s->tiles_key, key, sizeof(key));
  }
  uint64_t n64 = (uint64_t)s->h_tiles.size() * TILE_PIXELS;
  if (n64 > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "canvas part exceeds 2^32 rays");
  if (n64 == 0) return RT_OK;
  PipelineIn in;
  in.n0 = (uint32_t)n64;
  in.flags = flags;
  in.camera = true;
  in.cam = camera_dev(camera);
  in.tiles = n_parts == 1 ? nullptr : s->d_tiles;  // the whole image: tile k is tile k, no look-up at the start of every wave
  in.tiles_x = tiles_x;
  if ((flags & RT_FLAG_COMPACT) && (rgb64_dev || hit_leaf_dev || !rgb8_dev))
    return fail(RT_ERR_INVALID, "the compact layout is for rgb8 only");
  in.img = ImageOut{rgb8_dev, rgb64_dev, hit_leaf_dev, (flags & RT_FLAG_COMPACT) ? 1u : 0u};
  uint64_t valid = 0;
  for (uint32_t t : s->h_tiles) {
    uint32_t x0 = (t % tiles_x) * TILE, y0 = (t / tiles_x) * TILE;
    uint32_t w = std::min<uint32_t>(TILE, camera->hsize - x0), h = std::min<uint32_t>(TILE, camera->vsize - y0);
    valid += (uint64_t)w * h;
  }
  if ((flags & RT_FLAG_ASYNC) && s->max_children == 0) return enqueue_direct_frame(s, in, valid);
  int rc = collect_all(s);  // frames still in flight finish first
  if (rc) return rc;
  s->last_stream = s->stream;
  rt_stats local{};
  const bool deferred = (flags & RT_FLAG_ASYNC) != 0;  // asked to enqueue, but this scene's frames need the host
  rc = run_pipeline(s, in, deferred ? &local : stats);
  if (rc) return rc;
  if (deferred) {  // reported by rt_scene_wait like the frames that were only enqueued
    rt_stats& t = s->async_total;
    t.primary += valid;
    t.shadow += local.shadow; t.reflect += local.reflect; t.refract += local.refract;
    t.kernel_ms += local.kernel_ms; t.kernel_launches += local.kernel_launches; t.generations += local.generations;
  } else if (stats) {
    stats->primary = valid;
    stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  return RT_OK;
}

extern "C" int rt_scene_wait(rt_scene* s, rt_stats* total) {
  if (!s) return fail(RT_ERR_INVALID, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  int rc = collect_all(s);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s->copy_stream));
  if (total) *total = s->async_total;
  s->async_total = rt_stats{};
  return RT_OK;
}

extern "C" int rt_scene_stream(rt_scene* s, void** hip_stream) {
  if (!s || !hip_stream) return fail(RT_ERR_INVALID, "null argument");
  *hip_stream = (void*)s->last_stream;
  return RT_OK;
}

extern "C" int rt_scene_copy_stream(rt_scene* s, void** hip_stream) {
  if (!s || !hip_stream) return fail(RT_ERR_INVALID, "null argument");
  *hip_stream = (void*)s->copy_stream;
  return RT_OK;
}

// The copy stream waits for everything enqueued so far on after_stream or, when none is given, on every render
// stream that has had work enqueued since the copy stream was last ordered after it.
static int order_copy_stream(rt_scene* s, void* after_stream) {
  if (after_stream) {
    HIP_TRY(hipEventRecord(s->copy_order, (hipStream_t)after_stream));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->copy_order, 0));
    return RT_OK;
  }
  if (s->stream_used) {  // nothing new on a stream since the last ordering otherwise
    HIP_TRY(hipEventRecord(s->copy_order, s->stream));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->copy_order, 0));
    s->stream_used = false;
  }
  for (int k = 0; k < rt_scene::FRAME_STREAMS - 1; k++) {
    if (!s->frame_stream_used[k]) continue;
    HIP_TRY(hipEventRecord(s->copy_order2[k], s->frame_stream[k]));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->copy_order2[k], 0));
    s->frame_stream_used[k] = false;
  }
  return RT_OK;
}

extern "C" int rt_download_2d(rt_scene* s, const void* src_dev, void* dst_host, uint64_t row_bytes, uint64_t rows,
                              uint64_t pitch, void* after_stream) {
  if (!s || (row_bytes && rows && (!src_dev || !dst_host))) return fail(RT_ERR_INVALID, "null argument");
  if (pitch < row_bytes) return fail(RT_ERR_INVALID, "pitch smaller than a row");
  HIP_TRY(hipSetDevice(s->device));
  if (row_bytes == 0 || rows == 0) return RT_OK;
  int rco = order_copy_stream(s, after_stream);
  if (rco) return rco;
  HIP_TRY(hipMemcpy2DAsync(dst_host, pitch, src_dev, pitch, row_bytes, rows, hipMemcpyDeviceToHost, s->copy_stream));
  return RT_OK;
}

extern "C" int rt_download(rt_scene* s, const void* src_dev, void* dst_host, uint64_t bytes, void* after_stream) {
  if (!s || (bytes && (!src_dev || !dst_host))) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (bytes == 0) return RT_OK;
  int rco = order_copy_stream(s, after_stream);
  if (rco) return rco;
  // a plain asynchronous copy: the runtime moves it with a DMA engine, which leaves the compute units
  // to the frame that renders meanwhile (a copy kernel of our own measured 40 % slower frames)
  HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s->copy_stream));
  return RT_OK;
}

extern "C" int rt_copy_fence(rt_scene* s, uint32_t slot) {
  if (!s || slot >= (uint32_t)rt_scene::COPY_FENCES) return fail(RT_ERR_INVALID, "rt_copy_fence: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventRecord(s->copy_fence[slot], s->copy_stream));
  s->copy_fence_set[slot] = true;
  return RT_OK;
}

extern "C" int rt_copy_wait(rt_scene* s, uint32_t slot) {
  if (!s || slot >= (uint32_t)rt_scene::COPY_FENCES) return fail(RT_ERR_INVALID, "rt_copy_wait: bad argument");
  if (!s->copy_fence_set[slot]) return RT_OK;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventSynchronize(s->copy_fence[slot]));
  return RT_OK;
}

extern "C" int rt_render_host(rt_scene* s, const rt_camera* camera, int part, int n_parts, uint32_t flags,
                              uint8_t* rgb8, double* rgb64, int32_t* hit_leaf, rt_stats* stats) {
  if (!s || !camera) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  size_t np = (size_t)camera->hsize * camera->vsize;
  size_t b8 = rgb8 ? np * 3 : 0, b64 = rgb64 ? np * 3 * sizeof(double) : 0, bl = hit_leaf ? np * sizeof(int32_t) : 0;
  size_t off64 = (b8 + 255) & ~(size_t)255, offl = (off64 + b64 + 255) & ~(size_t)255;
  int rc = stage_reserve(s, offl + bl + 256);
  if (rc) return rc;
  char* base = (char*)s->stage;
  HIP_TRY(hipMemsetAsync(base, 0, offl + bl, main_stream(s)));
  if (hit_leaf) HIP_TRY(hipMemsetAsync(base + offl, 0xFF, bl, main_stream(s)));
  rc = rt_render(s, camera, part, n_parts, flags & ~(uint32_t)(RT_FLAG_ASYNC | RT_FLAG_COMPACT), rgb8 ? (uint8_t*)base : nullptr,
                 rgb64 ? (double*)(base + off64) : nullptr, hit_leaf ? (int32_t*)(base + offl) : nullptr, stats);
  if (rc) return rc;
  if ((flags & RT_FLAG_BANDS) && n_parts > 1) {
    // only this part's bands of 32 rows go to the host images, so the parts of one frame can share them
    const size_t rows_total = camera->vsize, w = camera->hsize;
    for (size_t y0 = (size_t)part * TILE; y0 < rows_total; y0 += (size_t)n_parts * TILE) {
      const size_t rows = std::min<size_t>(TILE, rows_total - y0), px = y0 * w, n = rows * w;
      if (rgb8) HIP_TRY(hipMemcpyAsync(rgb8 + px * 3, base + px * 3, n * 3, hipMemcpyDeviceToHost, main_stream(s)));
      if (rgb64)
        HIP_TRY(hipMemcpyAsync(rgb64 + px * 3, base + off64 + px * 3 * sizeof(double), n * 3 * sizeof(double),
                               hipMemcpyDeviceToHost, main_stream(s)));
      if (hit_leaf)
        HIP_TRY(hipMemcpyAsync(hit_leaf + px, base + offl + px * sizeof(int32_t), n * sizeof(int32_t),
                               hipMemcpyDeviceToHost, main_stream(s)));
    }
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    return RT_OK;
  }
  if (rgb8) HIP_TRY(hipMemcpy(rgb8, base, b8, hipMemcpyDeviceToHost));
  if (rgb64) HIP_TRY(hipMemcpy(rgb64, base + off64, b64, hipMemcpyDeviceToHost));
  if (hit_leaf) HIP_TRY(hipMemcpy(hit_leaf, base + offl, bl, hipMemcpyDeviceToHost));
  return RT_OK;
}

extern "C" int rt_ray_for_pixel(rt_scene* s, const rt_camera* camera, uint64_t n, const uint32_t* xy,
                                double* rays6) {
  if (!s || !camera || (n && (!xy || !rays6))) return fail(RT_ERR_INVALID, "null argument");
  if (!affine(camera->inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) camera transform");
  HIP_TRY(hipSetDevice(s->device));
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "too many pixels in one call");
  int rc = stage_reserve(s, n * 6 * sizeof(double) + n * 2 * sizeof(uint32_t));
  if (rc) return rc;
  double* d_rays = (double*)s->stage;
  uint32_t* d_xy = (uint32_t*)(d_rays + n * 6);
  HIP_TRY(hipMemcpyAsync(d_xy, xy, n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, main_stream(s)));
  hipLaunchKernelGGL(k_pixel_rays, dim3(blocks_for(n)), dim3(BLOCK), 0, main_stream(s), camera_dev(camera), (uint32_t)n,
                     d_xy, d_rays);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rays6, d_rays, n * 6 * sizeof(double), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return RT_OK;
}

static int load_user_rays(rt_scene* s, uint64_t n, const double* rays6) {
  if (n > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "too many rays in one call");
  int rc = ensure_gen(s, 0, (size_t)n);
  if (rc) return rc;
  rc = stage_reserve(s, n * 6 * sizeof(double) + n * 16);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(s->stage, rays6, n * 6 * sizeof(double), hipMemcpyHostToDevice, main_stream(s)));
  hipLaunchKernelGGL(k_load_rays, dim3(blocks_for(n)), dim3(BLOCK), 0, main_stream(s), s->gens[0].view, (uint32_t)n,
                     (const double*)s->stage);
  return RT_OK;
}

extern "C" int rt_color_at(rt_scene* s, uint64_t n, const double* rays6, int remaining, uint32_t flags, double* rgb,
                           int32_t* leaf, rt_stats* stats) {
  if (!s || (n && (!rays6 || !rgb))) return fail(RT_ERR_INVALID, "null argument");
  if (remaining < 0) return fail(RT_ERR_INVALID, "remaining must be >= 0");
  if (remaining > RECURSION_DEPTH) return fail(RT_ERR_UNSUPPORTED, "remaining above the reference's depth of 5");
  HIP_TRY(hipSetDevice(s->device));
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n == 0) return RT_OK;
  auto w0 = std::chrono::steady_clock::now();
  int rc = load_user_rays(s, n, rays6);
  if (rc) return rc;
  PipelineIn in;
  in.n0 = (uint32_t)n;
  in.remaining0 = remaining;
  in.flags = flags;
  rc = run_pipeline(s, in, stats);
  if (rc) return rc;
  // the staged rays are consumed: reuse the buffer for the results
  double* d_rgb = (double*)s->stage;
  int32_t* d_leaf = (int32_t*)(d_rgb + n * 3);
  hipLaunchKernelGGL(k_output_rays, dim3(blocks_for(n)), dim3(BLOCK), 0, main_stream(s), s->gens[0].view, (uint32_t)n,
                     d_rgb, d_leaf, (double*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb, d_rgb, n * 3 * sizeof(double), hipMemcpyDeviceToHost, main_stream(s)));
  if (leaf) HIP_TRY(hipMemcpyAsync(leaf, d_leaf, n * sizeof(int32_t), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  if (stats) {
    stats->primary = n;
    stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
  return RT_OK;
}

extern "C" int rt_intersect_hit(rt_scene* s, uint64_t n, const double* rays6, double* t, int32_t* leaf) {
  if (!s || (n && (!rays6 || !t || !leaf))) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (n == 0) return RT_OK;
  int rc = load_user_rays(s, n, rays6);
  if (rc) return rc;
  dim3 grid(blocks_for(n)), block(BLOCK);
  if (s->has_csg)
    hipLaunchKernelGGL((k_intersect<64, true>), grid, block, csg_lds_bytes(s), main_stream(s), s->dev, s->gens[0].view, (uint32_t)n);
  else if (pick_stack(s->stack_need) == STACK_SMALL)
    hipLaunchKernelGGL((k_intersect<STACK_SMALL, false>), grid, block, 0, main_stream(s), s->dev, s->gens[0].view, (uint32_t)n);
  else
    hipLaunchKernelGGL((k_intersect<64, false>), grid, block, 0, main_stream(s), s->dev, s->gens[0].view, (uint32_t)n);
  double* d_t = (double*)s->stage;
  int32_t* d_leaf = (int32_t*)(d_t + n);
  hipLaunchKernelGGL(k_output_rays, grid, block, 0, main_stream(s), s->gens[0].view, (uint32_t)n, (double*)nullptr, d_leaf,
                     d_t);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipMemcpyAsync(leaf, d_leaf, n * sizeof(int32_t), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return RT_OK;
}

extern "C" int rt_is_shadowed(rt_scene* s, uint64_t n, const double* points3, uint8_t* shadowed) {
  if (!s || (n && (!points3 || !shadowed))) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "too many points in one call");
  int rc = stage_reserve(s, n * 3 * sizeof(double) + n + 256);
  if (rc) return rc;
  double* d_pts = (double*)s->stage;
  uint8_t* d_out = (uint8_t*)(d_pts + n * 3);
  HIP_TRY(hipMemcpyAsync(d_pts, points3, n * 3 * sizeof(double), hipMemcpyHostToDevice, main_stream(s)));
  dim3 grid(blocks_for(n)), block(BLOCK);
  if (s->has_csg)
    hipLaunchKernelGGL((k_points_shadowed<64, true>), grid, block, csg_lds_bytes(s), main_stream(s), s->dev, (uint32_t)n, d_pts, d_out);
  else if (pick_stack(s->stack_need) == STACK_SMALL)
    hipLaunchKernelGGL((k_points_shadowed<STACK_SMALL, false>), grid, block, 0, main_stream(s), s->dev, (uint32_t)n, d_pts, d_out);
  else
    hipLaunchKernelGGL((k_points_shadowed<64, false>), grid, block, 0, main_stream(s), s->dev, (uint32_t)n, d_pts, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(shadowed, d_out, n, hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return RT_OK;
}
