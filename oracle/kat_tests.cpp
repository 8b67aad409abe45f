// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp.  Known-answer tests that pin
// the CPU oracle against the reference's own #[test] vectors for the render
// path.  Each case names the reference test it restates (file:line).  Equality
// on tuples / colours / matrices is the reference's approximate one (1e-4);
// assert_eq! on raw f64 there is exact here too.
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "rt_oracle.hpp"

using namespace orc;

static int g_fail = 0, g_checks = 0;
static const char* g_case = "";
#define CHECK(cond)                                                              \
  do {                                                                           \
    g_checks++;                                                                  \
    if (!(cond)) {                                                               \
      g_fail++;                                                                  \
      printf("FAIL %s: %s (line %d)\n", g_case, #cond, __LINE__);                \
    }                                                                            \
  } while (0)

struct Case {
  const char* name;
  std::function<void()> fn;
};
static std::vector<Case>& cases() {
  static std::vector<Case> c;
  return c;
}
struct Reg {
  Reg(const char* n, std::function<void()> f) { cases().push_back({n, f}); }
};
#define KAT(id, cite) \
  static void id();   \
  static Reg reg_##id(#id " [" cite "]", id); \
  static void id()

static const double SQ2 = std::sqrt(2.0), SQ3 = std::sqrt(3.0);
static Intersection mk(double t, const Shape* s) { Intersection i; i.t = t; i.what = s; return i; }
static Intersection mkuv(double t, const Shape* s, double u, double v) {
  Intersection i; i.t = t; i.what = s; i.has_uv = true; i.u = u; i.v = v; return i;
}
static Ray R(double ox, double oy, double oz, double dx, double dy, double dz) {
  return ray_new(point(ox, oy, oz), vector(dx, dy, dz));
}

// ------------------------------------------------------------------ tuple.rs
KAT(tuple_ops, "tuple.rs:278-390") {
  CHECK(tuple_eq(Tuple{3, -2, 5, 1} + Tuple{-2, 3, 1, 0}, Tuple{1, 1, 6, 1}));
  CHECK(tuple_eq(point(3, 2, 1) - point(5, 6, 7), vector(-2, -4, -6)));
  CHECK(tuple_eq(point(3, 2, 1) - vector(5, 6, 7), point(-2, -4, -6)));
  CHECK(tuple_eq(-Tuple{1, -2, 3, -4}, Tuple{-1, 2, -3, 4}));
  CHECK(tuple_eq(Tuple{1, -2, 3, -4} * 3.5, Tuple{3.5, -7, 10.5, -14}));
  CHECK(magnitude(vector(1, 2, 3)) == std::sqrt(14.0));
  CHECK(magnitude(vector(-1, -2, -3)) == std::sqrt(14.0));
  CHECK(tuple_eq(normalize(vector(4, 0, 0)), vector(1, 0, 0)));
  CHECK(tuple_eq(normalize(vector(1, 2, 3)),
                 vector(1 / std::sqrt(14.0), 2 / std::sqrt(14.0), 3 / std::sqrt(14.0))));
  CHECK(dot(vector(1, 2, 3), vector(2, 3, 4)) == 20.0);
  CHECK(tuple_eq(cross(vector(1, 2, 3), vector(2, 3, 4)), vector(-1, 2, -1)));
  CHECK(tuple_eq(cross(vector(2, 3, 4), vector(1, 2, 3)), vector(1, -2, 1)));
  CHECK(tuple_eq(reflect(vector(1, -1, 0), vector(0, 1, 0)), vector(1, 1, 0)));
}

// ----------------------------------------------------------------- matrix.rs
KAT(mat4_determinant, "matrix.rs:767-778") {
  Matrix a{{-2, -8, 3, 5, -3, 1, 7, 3, 1, 2, -9, 6, -6, 7, 7, -9}};
  CHECK(mat_cofactor(a, 0, 0) == 690.0);
  CHECK(mat_cofactor(a, 0, 1) == 447.0);
  CHECK(mat_cofactor(a, 0, 2) == 210.0);
  CHECK(mat_cofactor(a, 0, 3) == 51.0);
  CHECK(mat_determinant(a) == -4071.0);
}
KAT(mat4_inverse, "matrix.rs:781-794") {
  Matrix a{{8, -5, 9, 2, 7, 5, 6, 1, -6, 0, 9, 6, -3, 0, -9, -4}};
  Matrix i{{-0.15385, -0.15385, -0.28205, -0.53846, -0.07692, 0.12308, 0.02564, 0.03077,
            0.35897, 0.35897, 0.43590, 0.92308, -0.69231, -0.69231, -0.76923, -1.92308}};
  Matrix inv;
  CHECK(mat_inverse(a, &inv));
  CHECK(mat_eq(inv, i));
}
KAT(mat4_inverse_mult, "matrix.rs:797-812") {
  Matrix a{{3, -9, 7, 3, 3, 8, 2, -9, -4, 4, 4, 1, -6, 5, -1, 1}};
  Matrix b{{8, 2, 2, 2, 3, -1, 7, 0, 7, 0, 5, 4, 6, -2, 0, 5}};
  Matrix c = a * b, binv;
  CHECK(mat_inverse(b, &binv));
  CHECK(mat_eq(a, c * binv));
}
KAT(mat4_transforms, "matrix.rs:815-956") {
  CHECK(tuple_eq(mat_translation(5, -3, 2) * point(-3, 4, 5), point(2, 1, 7)));
  Matrix ti;
  CHECK(mat_inverse(mat_translation(5, -3, 2), &ti));
  CHECK(tuple_eq(ti * point(-3, 4, 5), point(-8, 7, 3)));
  CHECK(tuple_eq(mat_translation(5, -3, 2) * vector(-3, 4, 5), vector(-3, 4, 5)));
  CHECK(tuple_eq(mat_scaling(2, 3, 4) * vector(-4, 6, 8), vector(-8, 18, 32)));
  Matrix si;
  CHECK(mat_inverse(mat_scaling(2, 3, 4), &si));
  CHECK(tuple_eq(si * vector(-4, 6, 8), vector(-2, 2, 2)));
  CHECK(tuple_eq(mat_rotation_x(M_PI / 2) * point(0, 1, 0), point(0, 0, 1)));
  CHECK(tuple_eq(mat_rotation_x(M_PI / 4) * point(0, 1, 0), point(0, SQ2 / 2, SQ2 / 2)));
  CHECK(tuple_eq(mat_rotation_y(M_PI / 2) * point(0, 0, 1), point(1, 0, 0)));
  CHECK(tuple_eq(mat_rotation_y(M_PI / 4) * point(0, 0, 1), point(SQ2 / 2, 0, SQ2 / 2)));
  CHECK(tuple_eq(mat_rotation_z(M_PI / 2) * point(0, 1, 0), point(-1, 0, 0)));
  CHECK(tuple_eq(mat_rotation_z(M_PI / 4) * point(0, 1, 0), point(-SQ2 / 2, SQ2 / 2, 0)));
  CHECK(tuple_eq(mat_shearing(1, 0, 0, 0, 0, 0) * point(2, 3, 4), point(5, 3, 4)));
  CHECK(tuple_eq(mat_shearing(0, 0, 0, 0, 0, 1) * point(2, 3, 4), point(2, 3, 7)));
  Matrix t = mat_translation(10, 5, 7) * mat_scaling(5, 5, 5) * mat_rotation_x(M_PI / 2);
  CHECK(tuple_eq(t * point(1, 0, 1), point(15, 0, 7)));
}
KAT(mat4_transpose, "matrix.rs:686-699") {
  Matrix a{{0, 9, 3, 0, 9, 8, 0, 8, 1, 8, 5, 3, 0, 0, 5, 8}};
  Matrix t{{0, 9, 1, 0, 9, 8, 8, 0, 3, 0, 5, 5, 0, 8, 3, 8}};
  CHECK(mat_eq(t, mat_transposition(a)));
  CHECK(mat_eq(mat_transposition(t), a));
}
KAT(view_transforms, "matrix.rs:959-999") {
  CHECK(mat_eq(mat_identity(), mat_view_transform(point(0, 0, 0), point(0, 0, -1), vector(0, 1, 0))));
  CHECK(mat_eq(mat_view_transform(point(0, 0, 0), point(0, 0, 1), vector(0, 1, 0)), mat_scaling(-1, 1, -1)));
  CHECK(mat_eq(mat_view_transform(point(0, 0, 8), point(0, 0, 0), vector(0, 1, 0)), mat_translation(0, 0, -8)));
  Matrix a{{-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843,
            -0.35857, 0.59761, -0.71714, 0.0, 0.0, 0.0, 0.0, 1.0}};
  CHECK(mat_eq(mat_view_transform(point(1, 3, 2), point(4, -2, 8), vector(1, 1, 0)), a));
}

// -------------------------------------------------------------------- ray.rs
KAT(ray_ops, "ray.rs:59-95") {
  Ray r = R(2, 3, 4, 1, 0, 0);
  CHECK(tuple_eq(ray_position(r, 0.0), point(2, 3, 4)));
  CHECK(tuple_eq(ray_position(r, 1.0), point(3, 3, 4)));
  CHECK(tuple_eq(ray_position(r, -1.0), point(1, 3, 4)));
  CHECK(tuple_eq(ray_position(r, 2.5), point(4.5, 3, 4)));
  Ray r2 = R(1, 2, 3, 0, 1, 0);
  Ray t = ray_transform(r2, mat_translation(3, 4, 5));
  CHECK(tuple_eq(t.origin, point(4, 6, 8)) && tuple_eq(t.direction, vector(0, 1, 0)));
  Ray s = ray_transform(r2, mat_scaling(2, 3, 4));
  CHECK(tuple_eq(s.origin, point(2, 6, 12)) && tuple_eq(s.direction, vector(0, 3, 0)));
}

// ----------------------------------------------------------------- camera.rs
KAT(camera_center_ray, "camera.rs:120-127") {
  Camera c = Camera::make(201, 101, M_PI / 2.0, mat_identity());
  Ray r = c.ray_for_pixel(100, 50);
  CHECK(tuple_eq(r.origin, point(0, 0, 0)));
  CHECK(tuple_eq(r.direction, vector(0, 0, -1)));
}
KAT(camera_corner_ray, "camera.rs:130-137") {
  Camera c = Camera::make(201, 101, M_PI / 2.0, mat_identity());
  Ray r = c.ray_for_pixel(0, 0);
  CHECK(tuple_eq(r.origin, point(0, 0, 0)));
  CHECK(tuple_eq(r.direction, vector(0.66519, 0.33259, -0.66851)));
}
KAT(camera_transformed, "camera.rs:140-149") {
  Camera c = Camera::make(201, 101, M_PI / 2.0, mat_rotation_y(M_PI / 4.0) * mat_translation(0, -2, 5));
  Ray r = c.ray_for_pixel(100, 50);
  CHECK(tuple_eq(r.origin, point(0, 2, -5)));
  CHECK(tuple_eq(r.direction, vector(SQ2 / 2, 0, -SQ2 / 2)));
}
KAT(camera_render_world, "camera.rs:152-168") {
  World w = World::default_world();
  w.finalize();
  Camera c = Camera::make(11, 11, M_PI / 2.0,
                          mat_view_transform(point(0, 0, -5), point(0, 0, 0), vector(0, 1, 0)));
  RenderOut out;
  parallel_render(w, c, 2, 0, 1, 0, 1, &out);
  size_t p = 5 * 11 + 5;
  CHECK(color_eq(Color{out.rgb[p * 3], out.rgb[p * 3 + 1], out.rgb[p * 3 + 2]}, Color{0.38066, 0.47583, 0.2855}));
}

// ------------------------------------------------------------------ light.rs
static Color light_case(Tuple eyev, Tuple lightpos, bool shadow = false) {
  Shape s = Shape::sphere();
  PointLight l = point_light_new({1, 1, 1}, lightpos);
  return lighting(s.material, s, l, point(0, 0, 0), eyev, vector(0, 0, -1), shadow);
}
KAT(lighting_angles, "light.rs:177-284") {
  CHECK(color_eq(light_case(vector(0, 0, -1), point(0, 0, -10)), Color{1.9, 1.9, 1.9}));
  CHECK(color_eq(light_case(vector(0, SQ2 / 2, SQ2 / 2), point(0, 0, -10)), Color{1.0, 1.0, 1.0}));
  CHECK(color_eq(light_case(vector(0, 0, -1), point(0, 10, -10)), Color{0.7364, 0.7364, 0.7364}));
  CHECK(color_eq(light_case(vector(0, -SQ2 / 2, -SQ2 / 2), point(0, 10, -10)), Color{1.6364, 1.6364, 1.6364}));
  CHECK(color_eq(light_case(vector(0, 0, -1), point(0, 0, 10)), Color{0.1, 0.1, 0.1}));
}
KAT(lighting_stripe, "light.rs:287-327") {
  Shape s = Shape::sphere();
  s.material.color = {0.5, 0.5, 0.5};
  s.material.pattern = pattern_make(PatternType::Stripe, {1, 1, 1}, {0, 0, 0});
  s.material.ambient = 1.0; s.material.diffuse = 0.0; s.material.specular = 0.0;
  s.material.shininess = 0.0; s.material.reflective = 0.0;
  PointLight l = point_light_new({1, 1, 1}, point(0, 0, -10));
  CHECK(color_eq(lighting(s.material, s, l, point(0.9, 0, 0), vector(0, 0, -1), vector(0, 0, -1), false), Color{1, 1, 1}));
  CHECK(color_eq(lighting(s.material, s, l, point(1.1, 0, 0), vector(0, 0, -1), vector(0, 0, -1), false), Color{0, 0, 0}));
}

// ---------------------------------------------------------------- pattern.rs
KAT(pattern_stripe, "pattern.rs:212-251") {
  Pattern p = pattern_make(PatternType::Stripe, {1, 1, 1}, {0, 0, 0});
  Color W{1, 1, 1}, B{0, 0, 0};
  CHECK(color_eq(pattern_at(p, point(0, 0, 0)), W));
  CHECK(color_eq(pattern_at(p, point(0, 1, 0)), W));
  CHECK(color_eq(pattern_at(p, point(0, 2, 0)), W));
  CHECK(color_eq(pattern_at(p, point(0, 0, 1)), W));
  CHECK(color_eq(pattern_at(p, point(0, 0, 2)), W));
  CHECK(color_eq(pattern_at(p, point(0.9, 0, 0)), W));
  CHECK(color_eq(pattern_at(p, point(1.0, 0, 0)), B));
  CHECK(color_eq(pattern_at(p, point(-0.1, 0, 0)), B));
  CHECK(color_eq(pattern_at(p, point(-1.0, 0, 0)), B));
  CHECK(color_eq(pattern_at(p, point(-1.1, 0, 0)), W));
}
static Color pat_obj(const Pattern& p, const Shape& s, Tuple pt) {
  // lighting() with ambient 1 and a white light returns the pattern colour
  Shape o = s;
  o.material = Material{};
  o.material.pattern = p;
  o.material.ambient = 1.0; o.material.diffuse = 0.0; o.material.specular = 0.0;
  PointLight l = point_light_new({1, 1, 1}, point(0, 0, -10));
  return lighting(o.material, o, l, pt, vector(0, 0, -1), vector(0, 0, -1), true);
}
KAT(pattern_on_objects, "pattern.rs:254-330") {
  Color W{1, 1, 1};
  Shape o = Shape::sphere();
  o.set_transform(mat_scaling(2, 2, 2));
  Pattern st = pattern_make(PatternType::Stripe, {1, 1, 1}, {0, 0, 0});
  CHECK(color_eq(pat_obj(st, o, point(1.5, 0, 0)), W));
  Shape plain = Shape::sphere();
  Pattern st2 = st;
  st2.transform = mat_scaling(2, 2, 2);
  CHECK(color_eq(pat_obj(st2, plain, point(1.5, 0, 0)), W));
  Pattern st3 = st;
  st3.transform = mat_translation(0.5, 0, 0);
  CHECK(color_eq(pat_obj(st3, o, point(2.5, 0, 0)), W));
  Pattern pp = pattern_make(PatternType::Point);
  CHECK(color_eq(pat_obj(pp, o, point(2, 3, 4)), Color{1.0, 1.5, 2.0}));
  Pattern pp2 = pp;
  pp2.transform = mat_scaling(2, 2, 2);
  CHECK(color_eq(pat_obj(pp2, plain, point(2, 3, 4)), Color{1.0, 1.5, 2.0}));
  Pattern pp3 = pp;
  pp3.transform = mat_translation(0.5, 1.0, 1.5);
  CHECK(color_eq(pat_obj(pp3, o, point(2.5, 3.0, 3.5)), Color{0.75, 0.5, 0.25}));
}
KAT(pattern_gradient_ring_checker, "pattern.rs:333-380") {
  Color W{1, 1, 1}, B{0, 0, 0};
  Pattern g = pattern_make(PatternType::Gradient, W, B);
  CHECK(color_eq(pattern_at(g, point(0, 0, 0)), W));
  CHECK(color_eq(pattern_at(g, point(0.25, 0, 0)), Color{0.75, 0.75, 0.75}));
  CHECK(color_eq(pattern_at(g, point(0.5, 0, 0)), Color{0.5, 0.5, 0.5}));
  CHECK(color_eq(pattern_at(g, point(0.75, 0, 0)), Color{0.25, 0.25, 0.25}));
  Pattern r = pattern_make(PatternType::Ring, W, B);
  CHECK(color_eq(pattern_at(r, point(0, 0, 0)), W));
  CHECK(color_eq(pattern_at(r, point(1, 0, 0)), B));
  CHECK(color_eq(pattern_at(r, point(0, 0, 1)), B));
  CHECK(color_eq(pattern_at(r, point(0.708, 0, 0.708)), B));
  Pattern c = pattern_make(PatternType::Checker, W, B);
  CHECK(color_eq(pattern_at(c, point(0, 0, 0)), W));
  CHECK(color_eq(pattern_at(c, point(0.99, 0, 0)), W));
  CHECK(color_eq(pattern_at(c, point(1.01, 0, 0)), B));
  CHECK(color_eq(pattern_at(c, point(0, 0.99, 0)), W));
  CHECK(color_eq(pattern_at(c, point(0, 1.01, 0)), B));
  CHECK(color_eq(pattern_at(c, point(0, 0, 0.99)), W));
  CHECK(color_eq(pattern_at(c, point(0, 0, 1.01)), B));
}

// ------------------------------------------------------------------ shape.rs
KAT(sphere_normals, "shape.rs:1487-1509,1645-1719") {
  Shape s = Shape::sphere();
  s.set_transform(mat_translation(0, 1, 0));
  CHECK(tuple_eq(normal_at(s, point(0, 1.70711, -0.70711), mk(0, &s)), vector(0, 0.70711, -0.70711)));
  Shape s2 = Shape::sphere();
  s2.set_transform(mat_scaling(1, 0.5, 1) * mat_rotation_z(M_PI / 5.0));
  CHECK(tuple_eq(normal_at(s2, point(0, SQ2 / 2, -SQ2 / 2), mk(0, &s2)), vector(0, 0.97014, -0.24254)));
  Shape u = Shape::sphere();
  CHECK(tuple_eq(normal_at(u, point(1, 0, 0), mk(0, &u)), vector(1, 0, 0)));
  CHECK(tuple_eq(normal_at(u, point(0, 1, 0), mk(0, &u)), vector(0, 1, 0)));
  CHECK(tuple_eq(normal_at(u, point(0, 0, 1), mk(0, &u)), vector(0, 0, 1)));
  CHECK(tuple_eq(normal_at(u, point(SQ3 / 3, SQ3 / 3, SQ3 / 3), mk(0, &u)), vector(SQ3 / 3, SQ3 / 3, SQ3 / 3)));
}
KAT(plane_cases, "shape.rs:1512-1559") {
  Shape p = Shape::plane();
  CHECK(tuple_eq(p.local_normal_at(point(0, 0, 0), mk(0, &p)), vector(0, 1, 0)));
  CHECK(tuple_eq(p.local_normal_at(point(10, 0, -10), mk(0, &p)), vector(0, 1, 0)));
  CHECK(tuple_eq(p.local_normal_at(point(-5, 0, 150), mk(0, &p)), vector(0, 1, 0)));
  Intersections a, b;
  p.local_intersect(R(0, 1, 0, 0, -1, 0), a);
  CHECK(a.xs.size() == 1 && a.xs[0].t == 1.0);
  p.local_intersect(R(0, -1, 0, 0, 1, 0), b);
  CHECK(b.xs.size() == 1 && b.xs[0].t == 1.0);
}
KAT(sphere_intersections, "shape.rs:1562-1600") {
  Shape s = Shape::sphere();
  Intersections a, b, c;
  intersect(s, R(0, 1, -5, 0, 0, 1), a);
  CHECK(a.xs.size() == 2 && a.xs[0].t == 5.0 && a.xs[1].t == 5.0);
  s.local_intersect(R(0, 0, 0, 0, 0, 1), b);
  CHECK(b.xs.size() == 2 && b.xs[0].t == -1.0 && b.xs[1].t == 1.0);
  intersect(s, R(0, 0, 5, 0, 0, 1), c);
  CHECK(c.xs.size() == 2 && c.xs[0].t == -6.0 && c.xs[1].t == -4.0);
}
KAT(hit_selection, "shape.rs:1603-1642") {
  Shape s = Shape::sphere();
  Intersection h;
  Intersections a; a.xs = {mk(1, &s), mk(2, &s)};
  CHECK(a.hit(&h) && h.t == 1.0);
  Intersections b; b.xs = {mk(-1, &s), mk(1, &s)};
  CHECK(b.hit(&h) && h.t == 1.0);
  Intersections c; c.xs = {mk(-2, &s), mk(-1, &s)};
  CHECK(!c.hit(&h));
  Intersections d; d.xs = {mk(5, &s), mk(7, &s), mk(-3, &s), mk(2, &s)};
  CHECK(d.hit(&h) && h.t == 2.0);
}
KAT(precompute_state, "shape.rs:1722-1797") {
  Shape shape = Shape::sphere();
  Ray r = R(0, 0, -5, 0, 0, 1);
  Intersection i = mk(4.0, &shape);
  Computation c = computation_new(r, i, nullptr);
  CHECK(c.obj == i.what && c.t == i.t);
  CHECK(tuple_eq(c.point, point(0, 0, -1)));
  CHECK(tuple_eq(c.eyev, vector(0, 0, -1)));
  CHECK(tuple_eq(c.normalv, vector(0, 0, -1)));
  CHECK(!c.inside);
  Computation ci = computation_new(R(0, 0, 0, 0, 0, 1), mk(1.0, &shape), nullptr);
  CHECK(ci.inside);
  CHECK(tuple_eq(ci.point, point(0, 0, 1)));
  CHECK(tuple_eq(ci.eyev, vector(0, 0, -1)));
  CHECK(tuple_eq(ci.normalv, vector(0, 0, -1)));
  Shape moved = Shape::sphere();
  moved.set_transform(mat_translation(0, 0, 1));
  Computation co = computation_new(r, mk(5.0, &moved), nullptr);
  CHECK(co.over_point.z < -FEQ_EPSILON / 2.0);
  CHECK(co.point.z > co.over_point.z);
}
KAT(group_intersections, "shape.rs:1831-1886") {
  Shape g = Shape::group();
  Intersections e;
  g.local_intersect(R(0, 0, 0, 0, 0, 1), e);
  CHECK(e.xs.empty());

  Shape g2 = Shape::group();
  Shape s1 = Shape::sphere();
  Shape s2 = Shape::sphere();
  s2.set_transform(mat_translation(0, 0, -3));
  Shape s3 = Shape::sphere();
  s3.set_transform(mat_translation(5, 0, 0));
  g2.add_child(s1);
  g2.add_child(s2);
  g2.add_child(s3);
  Intersections xs;
  g2.local_intersect(R(0, 0, -5, 0, 0, 1), xs);
  CHECK(xs.xs.size() == 4);
  CHECK(xs.xs[0].what == &g2.children[1] && xs.xs[1].what == &g2.children[1]);
  CHECK(xs.xs[2].what == &g2.children[0] && xs.xs[3].what == &g2.children[0]);

  // shape.rs:1870-1886: transform fields written without set_transform, so the
  // cached inverses stay identity and the gate box is the child's unit box
  // moved by the group's own scaling -> no intersections
  Shape g3 = Shape::group();
  g3.transform = mat_scaling(2, 2, 2);
  Shape c = Shape::sphere();
  c.transform = mat_translation(5, 0, 0);
  g3.add_child(c);
  Intersections none;
  intersect(g3, R(10, 0, -10, 0, 0, 1), none);
  CHECK(none.xs.size() == 0);
}
static Shape nested_groups(Matrix m2) {
  Shape g1 = Shape::group();
  g1.transform = mat_rotation_y(M_PI / 2.0);
  Shape g2 = Shape::group();
  g2.transform = m2;
  Shape s = Shape::sphere();
  s.transform = mat_translation(5, 0, 0);
  g2.add_child(s);
  g1.add_child(g2);
  return g1;
}
KAT(group_spaces, "shape.rs:1889-1963") {
  Shape a = nested_groups(mat_scaling(2, 2, 2));
  CHECK(tuple_eq(a.children[0].children[0].world_to_object(point(-2, 0, -10)), point(0, 0, -1)));
  Shape b = nested_groups(mat_scaling(1, 2, 3));
  const Shape& leaf = b.children[0].children[0];
  CHECK(tuple_eq(leaf.normal_to_world(vector(SQ3 / 3, SQ3 / 3, SQ3 / 3)), vector(0.2857, 0.4286, -0.8571)));
  Shape dummy = Shape::sphere();
  CHECK(tuple_eq(normal_at(leaf, point(1.7321, 1.1547, -5.5774), mk(0, &dummy)), vector(0.2857, 0.4286, -0.8571)));
}
static Shape tri0() { return Shape::triangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0)); }
KAT(triangles, "shape.rs:1966-2085") {
  Shape t = tri0();
  CHECK(tuple_eq(t.tri.e1, vector(-1, -1, 0)));
  CHECK(tuple_eq(t.tri.e2, vector(1, -1, 0)));
  CHECK(tuple_eq(t.tri.normal, vector(0, 0, -1)));
  Intersection i = mk(0, &t);
  CHECK(tuple_eq(t.tri.normal, normal_at(t, point(0, 0.5, 0), i)));
  CHECK(tuple_eq(t.tri.normal, normal_at(t, point(-0.5, 0.75, 0), i)));
  CHECK(tuple_eq(t.tri.normal, normal_at(t, point(0.5, 0.25, 0), i)));
  Intersections a, b, c, d, e;
  t.local_intersect(R(0, -1, -2, 0, 1, 0), a);
  CHECK(a.xs.empty());
  t.local_intersect(R(1, 1, -2, 0, 0, 1), b);
  CHECK(b.xs.empty());
  t.local_intersect(R(-1, 1, -2, 0, 0, 1), c);
  CHECK(c.xs.empty());
  t.local_intersect(R(0, -1, -2, 0, 0, 1), d);
  CHECK(d.xs.empty());
  t.local_intersect(R(0, 0.5, -2, 0, 0, 1), e);
  CHECK(e.xs.size() == 1 && feq(e.xs[0].t, 2.0));
}
KAT(smooth_triangles, "shape.rs:2088-2170") {
  Shape s = Shape::smooth_triangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0), vector(1, 0, 0),
                                   vector(-1, 0, 0), vector(1, 0, 0));
  Intersections xs;
  s.local_intersect(R(-0.2, 0.3, -2, 0, 0, 1), xs);
  CHECK(xs.xs.size() == 1 && xs.xs[0].has_uv && feq(xs.xs[0].u, 0.45) && feq(xs.xs[0].v, 0.25));
  Shape s2 = Shape::smooth_triangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0), vector(0, 1, 0),
                                    vector(-1, 0, 0), vector(1, 0, 0));
  Intersection i = mkuv(1.0, &s2, 0.45, 0.25);
  CHECK(tuple_eq(normal_at(s2, point(0, 0, 0), i), vector(-0.5547, 0.83205, 0.0)));
}
KAT(csg, "shape.rs:2173-2223; intersect.rs:331-403") {
  bool expect[24] = {false, true, false, true, false, false, true, true, true, false, true, false,
                     true, true, false, false, false, true, false, true, true, true, false, false};
  ShapeType ops[3] = {ShapeType::Union, ShapeType::Intersection, ShapeType::Difference};
  int k = 0;
  for (int o = 0; o < 3; o++)
    for (int lhit = 1; lhit >= 0; lhit--)
      for (int inl = 1; inl >= 0; inl--)
        for (int inr = 1; inr >= 0; inr--) CHECK(intersection_allowed(ops[o], lhit, inl, inr) == expect[k++]);
  Shape miss = Shape::csg(ShapeType::Union, Shape::sphere(), Shape::cube());
  Intersections a;
  miss.local_intersect(R(0, 2, -5, 0, 0, 1), a);
  CHECK(a.xs.empty());
  Shape s1 = Shape::sphere(), s2 = Shape::sphere();
  s2.set_transform(mat_translation(0, 0, 0.5));
  Shape c = Shape::csg(ShapeType::Union, s1, s2);
  Intersections b;
  c.local_intersect(R(0, 0, -5, 0, 0, 1), b);
  CHECK(b.xs.size() == 2);
  if (b.xs.size() == 2) {
    CHECK(feq(b.xs[0].t, 4.0) && shape_eq(*b.xs[0].what, s1));
    CHECK(feq(b.xs[1].t, 6.5) && shape_eq(*b.xs[1].what, s2));
  }
}

// ------------------------------------------------------------------ world.rs
KAT(world_intersect, "world.rs:265-279") {
  World w = World::default_world();
  Intersections xs;
  w.intersect(R(0, 0, -5, 0, 0, 1), xs);
  CHECK(xs.xs.size() == 4);
  CHECK(xs.xs[0].t == 4.0 && xs.xs[1].t == 4.5 && xs.xs[2].t == 5.5 && xs.xs[3].t == 6.0);
}
KAT(world_shade_outside_inside_shadow, "world.rs:282-355") {
  World w = World::default_world();
  Ray r = R(0, 0, -5, 0, 0, 1);
  Computation c = computation_new(r, mk(4.0, &w.objects[0]), nullptr);
  CHECK(color_eq(w.shade_hit(c, 5, nullptr), Color{0.38066, 0.47583, 0.2855}));

  World w2 = World::default_world();
  w2.light = point_light_new({1, 1, 1}, point(0, 0.25, 0));
  Computation c2 = computation_new(R(0, 0, 0, 0, 0, 1), mk(0.5, &w2.objects[1]), nullptr);
  CHECK(color_eq(w2.shade_hit(c2, 5, nullptr), Color{0.90498, 0.90498, 0.90498}));

  World w3 = World::empty_world();
  w3.light = point_light_new({1, 1, 1}, point(0, 0, -10));
  w3.objects.push_back(Shape::sphere());
  Shape s2 = Shape::sphere();
  s2.set_transform(mat_translation(0, 0, 10));
  w3.objects.push_back(s2);
  Computation c3 = computation_new(R(0, 0, 5, 0, 0, 1), mk(4.0, &w3.objects[1]), nullptr);
  CHECK(color_eq(w3.shade_hit(c3, 5, nullptr), Color{0.1, 0.1, 0.1}));
}
KAT(world_color_at, "world.rs:358-411") {
  World w = World::default_world();
  CHECK(color_eq(w.color_at(R(0, 0, -5, 0, 1, 0), 5, nullptr), Color{0, 0, 0}));
  CHECK(color_eq(w.color_at(R(0, 0, -5, 0, 0, 1), 5, nullptr), Color{0.38066, 0.47583, 0.2855}));
  World w2 = World::default_world();
  w2.objects[0].material.ambient = 1.0;
  w2.objects[1].material.ambient = 1.0;
  CHECK(color_eq(w2.color_at(R(0, 0, 0.75, 0, 0, -1), 5, nullptr), w2.objects[1].material.color));
}
KAT(world_shadows, "world.rs:414-443") {
  World w = World::default_world();
  CHECK(!w.is_shadowed(point(0, 10, 0), nullptr));
  CHECK(w.is_shadowed(point(10, -10, 10), nullptr));
  CHECK(!w.is_shadowed(point(-20, 20, -20), nullptr));
  CHECK(!w.is_shadowed(point(-2, 2, -2), nullptr));
}
static World reflective_floor_world() {
  World w = World::default_world();
  Shape s = Shape::plane();
  s.material.reflective = 0.5;
  s.set_transform(mat_translation(0, -1, 0));
  w.objects.push_back(s);
  return w;
}
KAT(world_reflection, "world.rs:446-524") {
  World w0 = World::default_world();
  w0.objects[1].material.ambient = 1.0;
  Computation c0 = computation_new(R(0, 0, 0, 0, 0, 1), mk(1.0, &w0.objects[1]), nullptr);
  CHECK(color_eq(w0.reflected_color(c0, 5, nullptr), Color{0, 0, 0}));
  World w = reflective_floor_world();
  Ray r = R(0, 0, -3, 0, -SQ2 / 2, SQ2 / 2);
  Computation c = computation_new(r, mk(SQ2, &w.objects[2]), nullptr);
  CHECK(color_eq(w.reflected_color(c, 5, nullptr), Color{0.19032, 0.2379, 0.14274}));
  CHECK(color_eq(w.shade_hit(c, 5, nullptr), Color{0.87677, 0.92436, 0.82918}));
}
KAT(world_refraction, "world.rs:527-667") {
  World w = World::default_world();
  Intersections xs;
  xs.xs = {mk(4.0, &w.objects[0]), mk(6.0, &w.objects[0])};
  Ray r = R(0, 0, -5, 0, 0, 1);
  Computation c = computation_new(r, xs.xs[0], &xs);
  CHECK(color_eq(w.refracted_color(c, 5, nullptr), Color{0, 0, 0}));  // opaque

  World w2 = World::default_world();
  w2.objects[0].material.transparency = 1.0;
  w2.objects[0].material.refractive_index = 1.5;
  Intersections xs2;
  xs2.xs = {mk(4.0, &w2.objects[0]), mk(6.0, &w2.objects[0])};
  Computation c2 = computation_new(r, xs2.xs[0], &xs2);
  CHECK(color_eq(w2.refracted_color(c2, 0, nullptr), Color{0, 0, 0}));  // depth exhausted

  Intersections xs3;
  xs3.xs = {mk(-SQ2 / 2, &w2.objects[0]), mk(SQ2 / 2, &w2.objects[0])};
  Computation c3 = computation_new(R(0, 0, SQ2 / 2, 0, 1, 0), xs3.xs[1], &xs3);
  CHECK(color_eq(w2.refracted_color(c3, 5, nullptr), Color{0, 0, 0}));  // total internal reflection

  World w4 = World::default_world();
  w4.objects[0].material.ambient = 1.0;
  w4.objects[0].material.pattern = pattern_make(PatternType::Point);
  w4.objects[1].material.transparency = 1.0;
  w4.objects[1].material.refractive_index = 1.5;
  Intersections xs4;
  xs4.xs = {mk(-0.9899, &w4.objects[0]), mk(-0.4899, &w4.objects[1]), mk(0.4899, &w4.objects[1]),
            mk(0.9899, &w4.objects[0])};
  Computation c4 = computation_new(R(0, 0, 0.1, 0, 1, 0), xs4.xs[2], &xs4);
  CHECK(color_eq(w4.refracted_color(c4, 5, nullptr), Color{0.0, 0.99888, 0.04725}));
}
static World glass_floor_world(double reflective) {
  World w = World::default_world();
  Shape floor = Shape::plane();
  floor.set_transform(mat_translation(0, -1, 0));
  floor.material.reflective = reflective;
  floor.material.transparency = 0.5;
  floor.material.refractive_index = 1.5;
  Shape ball = Shape::sphere();
  ball.material.color = {1, 0, 0};
  ball.material.ambient = 0.5;
  ball.set_transform(mat_translation(0, -3.5, -0.5));
  w.objects.push_back(floor);
  w.objects.push_back(ball);
  return w;
}
KAT(world_transparent_floor, "world.rs:670-708") {
  World w = glass_floor_world(0.0);
  Intersections xs;
  xs.xs = {mk(SQ2, &w.objects[2])};
  Computation c = computation_new(R(0, 0, -3, 0, -SQ2 / 2, SQ2 / 2), xs.xs[0], &xs);
  CHECK(color_eq(w.shade_hit(c, 5, nullptr), Color{0.93642, 0.68642, 0.68642}));
}
KAT(world_schlick_floor, "world.rs:711-749") {
  World w = glass_floor_world(0.5);
  Intersections xs;
  xs.xs = {mk(SQ2, &w.objects[2])};
  Computation c = computation_new(R(0, 0, -3, 0, -SQ2 / 2, SQ2 / 2), xs.xs[0], &xs);
  CHECK(color_eq(w.shade_hit(c, 5, nullptr), Color{0.93391, 0.69643, 0.69243}));
}

// ------------------------------------------ canvas.rs (no reference test; source text only)
KAT(quantise_and_ppm, "canvas.rs:47-94 (unpinned)") {
  CHECK(quantise(0.0) == 0 && quantise(1.0) == 255 && quantise(2.0) == 255 && quantise(-1.0) == 0);
  CHECK(quantise(0.5) == 128);     // ceil(127.5)
  CHECK(quantise(1e-9) == 1);      // ceil, not round
  CHECK(quantise(std::nan("")) == 0);
  double px[6] = {1.0, 0.0, 0.5, 0.0, 0.0, 0.0};
  CHECK(ppm_encode(2, 1, px) == "P3\n2 1\n255\n255 0 128 0 0 0\n");
  // Hand trace of the column state machine (canvas.rs:57-86): its counter skips
  // part of the separators, so six white pixels (71 characters) stay on one
  // line and the seventh wraps.
  std::vector<double> wide(7 * 3, 1.0);
  std::string six;
  for (int i = 0; i < 6; i++) six += std::string(i ? " " : "") + "255 255 255";
  CHECK(ppm_encode(7, 1, wide.data()) == "P3\n7 1\n255\n" + six + "\n255 255 255\n");
}

// ------------------------------------------ scene.rs / obj.rs (no usable reference test)
KAT(obj_and_scene_loaders, "obj.rs:95-276; scene.rs:25-207 (unpinned)") {
  const char* text =
      "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\n\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\n"
      "g\nf 1 2 3 4 5\nvn 0 0 1\nf 1//1 2//1 3//1\n";
  ObjResult o = obj_parse_text(text);
  CHECK(o.vertices.size() == 5 && o.normals.size() == 1);
  CHECK(o.groups.count("") && o.groups.count("FirstGroup") && o.groups.count("SecondGroup"));
  CHECK(o.groups[""].children.size() == 0);
  CHECK(o.groups["FirstGroup"].children.size() == 1);
  // "g" alone is ignored, so the polygon and the smooth face land in SecondGroup
  CHECK(o.groups["SecondGroup"].children.size() == 1 + 3 + 1);
  const Shape& t2 = o.groups["SecondGroup"].children[2];
  CHECK(tuple_eq(t2.tri.p1, o.vertices[0]) && tuple_eq(t2.tri.p2, o.vertices[2]) && tuple_eq(t2.tri.p3, o.vertices[3]));
  CHECK(o.groups["SecondGroup"].children[4].ty == ShapeType::SmoothTriangle);
  CHECK(o.groups["SecondGroup"].children[4].has_full);  // add_child propagated an identity parent

  const char* js =
      "{\"canvas_width\": 4, \"canvas_height\": 2, \"field_of_view\": 1.0472,"
      "\"camera_from\": [0, 1.5, -10], \"camera_to\": [0, 1, 0], \"camera_up\": [0, 1, 0],"
      "\"light\": {\"intensity\": [0.85, 0.8, 0.65], \"position\": [-10, 10, -10, 1]},"
      "\"shapes\": [{\"ty\": \"group\", \"transform\": [2,0,0,0, 0,2,0,0, 0,0,2,0, 0,0,0,1],"
      "  \"children\": [{\"ty\": \"sphere\", \"transform\": [1,0,0,5, 0,1,0,0, 0,0,1,0, 0,0,0,1]}]},"
      " {\"ty\": \"plane\", \"material\": {\"color\": [1, 0.5], \"reflective\": 0.25}}]}";
  Scene s = scene_from_json(js, "");
  CHECK(s.camera.hsize == 4 && s.camera.vsize == 2);
  CHECK(s.world.objects.size() == 2 && s.world.n_leaves == 2);
  const Shape& leaf = s.world.objects[0].children[0];
  CHECK(leaf.has_parent && mat_eq(leaf.parent_transform, mat_scaling(2, 2, 2)));
  CHECK(tuple_eq(leaf.world_to_object(point(12, 0, 0)), point(1, 0, 0)));
  CHECK(s.world.objects[1].material.reflective == 0.25);
  CHECK(s.world.objects[1].material.color.b == 0.0);  // short colour arrays are zero-filled
  bool threw = false;
  try { scene_from_json("{\"canvas_width\": 4}", ""); } catch (const std::exception&) { threw = true; }
  CHECK(threw);
}

int main(int argc, char** argv) {
  int ran = 0;
  for (auto& c : cases()) {
    if (argc > 1 && !strstr(c.name, argv[1])) continue;
    g_case = c.name;
    int before = g_fail;
    try {
      c.fn();
    } catch (const std::exception& e) {
      g_fail++;
      printf("FAIL %s: exception %s\n", c.name, e.what());
    }
    printf("%s %s\n", g_fail == before ? "ok  " : "FAIL", c.name);
    ran++;
  }
  printf("%d cases, %d checks, %d failures\n", ran, g_checks, g_fail);
  return g_fail ? 1 : 0;
}
