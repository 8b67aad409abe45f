// TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-pixel render path.
//
// This is a plain C++17 / IEEE f64 restatement of the reference ray tracer's
// algorithm (camera ray -> world intersection -> Phong shading with shadow,
// reflection and refraction rays).  It exists to check the HIP path and to
// serve as the timed CPU baseline ("port").  Nothing in the shipped library may
// include, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it.
//
// Every function cites the reference file:line it follows (paths relative to
// the reference crate's src/).  Arithmetic keeps the reference's operation
// order; build with -ffp-contract=off so no mul+add is fused (Rust never fuses).
//
// Parity pin: the known-answer tests in kat_tests.cpp restate the reference's
// own #[test] vectors for this path.  Cube, cylinder, cone, group bounds,
// Schlick-as-unit, the PPM writer and the scene/OBJ loaders have no reference
// test: for those the restatement follows the source text only ("parity
// unpinned").
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

// consts.rs:7-11
constexpr double FEQ_EPSILON = 0.0001;
constexpr int RECURSION_DEPTH = 5;

// lib.rs:217-219
inline bool feq(double l, double r) { return std::fabs(l - r) < FEQ_EPSILON; }

// ---------------------------------------------------------------- tuple.rs
struct Tuple {
  double x = 0, y = 0, z = 0, w = 0;
};
inline Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
inline Tuple vector(double x, double y, double z) { return {x, y, z, 0.0}; }
Tuple operator+(Tuple a, Tuple b);
Tuple operator-(Tuple a, Tuple b);
Tuple operator-(Tuple a);
Tuple operator*(Tuple a, double s);
Tuple operator*(double s, Tuple a);
double dot(Tuple a, Tuple b);
double magnitude(Tuple a);
Tuple normalize(Tuple a);
Tuple cross(Tuple a, Tuple b);
Tuple reflect(Tuple v, Tuple n);
bool tuple_eq(Tuple a, Tuple b);  // feq per component (tuple.rs:45-52)
Tuple tuple_from_vec(const std::vector<double>& v);  // tuple.rs:62-72

// ---------------------------------------------------------------- color.rs
struct Color {
  double r = 0, g = 0, b = 0;
};
Color operator+(Color a, Color b);
Color operator-(Color a, Color b);
Color operator*(Color a, double s);
Color operator*(Color a, Color b);  // hadamard
Color color_average(Color a, Color b);
bool color_eq(Color a, Color b);
Color color_from_vec(const std::vector<double>& v);  // color.rs:59-68

// --------------------------------------------------------------- matrix.rs
struct Matrix {
  double d[16] = {0};
  double& at(int r, int c) { return d[r * 4 + c]; }
  double at(int r, int c) const { return d[r * 4 + c]; }
};
Matrix mat_identity();
Matrix mat_translation(double x, double y, double z);
Matrix mat_scaling(double x, double y, double z);
Matrix mat_rotation_x(double r);
Matrix mat_rotation_y(double r);
Matrix mat_rotation_z(double r);
Matrix mat_shearing(double xy, double xz, double yx, double yz, double zx, double zy);
Matrix mat_view_transform(Tuple from, Tuple to, Tuple up);
Matrix mat_transposition(const Matrix& m);
double mat_cofactor(const Matrix& m, int row, int col);
double mat_determinant(const Matrix& m);
bool mat_inverse(const Matrix& m, Matrix* out);  // false when det == 0.0
Matrix mat_inverse_or_throw(const Matrix& m, const char* what);
Matrix operator*(const Matrix& a, const Matrix& b);
Tuple operator*(const Matrix& m, Tuple t);
bool mat_eq(const Matrix& a, const Matrix& b);

// ------------------------------------------------------------------ ray.rs
struct Ray {
  Tuple origin, direction;
};
Ray ray_new(Tuple origin, Tuple direction);
Tuple ray_position(const Ray& r, double t);
Ray ray_transform(const Ray& r, const Matrix& m);

// -------------------------------------------------------------- pattern.rs
enum class PatternType { Null, Point, Identity, Stripe, Ring, Checker, Gradient, Blend };
struct Pattern {
  PatternType ty = PatternType::Null;
  Color a, b;  // primary / secondary (Identity uses a)
  std::shared_ptr<Pattern> first, second;  // Blend
  Matrix transform = mat_identity();
};
Pattern pattern_make(PatternType ty, Color a = {1, 1, 1}, Color b = {0, 0, 0});
Pattern pattern_blend(const Pattern& f, const Pattern& s);
Color pattern_at(const Pattern& p, Tuple pt);
bool pattern_eq(const Pattern& a, const Pattern& b);

// ---------------------------------------------------------------- light.rs
struct PointLight {
  Color intensity;
  Tuple position;
};
PointLight point_light_new(Color intensity, Tuple position);
struct Material {
  Color color{1.0, 1.0, 1.0};
  std::optional<Pattern> pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflective = 0.0, refractive_index = 1.0, transparency = 0.0;
};
bool material_eq(const Material& a, const Material& b);

// ------------------------------------------------------------- geometry.rs
struct TriangleInfo {
  Tuple p1, p2, p3, e1, e2, normal;
};
TriangleInfo triangle_info_new(Tuple p1, Tuple p2, Tuple p3);
struct Bounds {
  Tuple minimum, maximum;
};
Bounds bounds_new(double minx, double miny, double minz, double maxx, double maxy, double maxz);
Bounds bounds_transform(const Bounds& b, const Matrix& m);
void bounds_check_axis(double mn, double mx, double origin, double direction, double* tmin,
                       double* tmax);

// ---------------------------------------------------------------- shape.rs
enum class ShapeType {
  Empty, Sphere, Cube, Cone, Cylinder, Plane, Triangle, SmoothTriangle, Group,
  Union, Intersection, Difference
};
struct Intersection;
struct Intersections;
struct Shape {
  ShapeType ty = ShapeType::Empty;
  double minimum = 0, maximum = 0;  // Cone / Cylinder
  bool closed = false;              // Cone / Cylinder
  Tuple plane_normal;               // Plane
  TriangleInfo tri;                 // Triangle / SmoothTriangle
  Tuple n1, n2, n3;                 // SmoothTriangle
  std::vector<Shape> children;      // Group: all children; CSG: [left, right]
  Material material;

  Matrix transform = mat_identity();
  bool has_parent = false;
  Matrix parent_transform;
  Matrix transform_inverse = mat_identity();
  bool has_full = false;
  Matrix full_transform_inverse;
  mutable bool has_saved_bounds = false;
  mutable Bounds saved_bounds;

  int leaf_id = -1;  // depth-first index among non-container shapes (set by World::finalize)

  static Shape empty();
  static Shape sphere();
  static Shape plane();
  static Shape cube();
  static Shape cylinder();
  static Shape bounded_cylinder(double mn, double mx);
  static Shape capped_cylinder(double mn, double mx);
  static Shape bounded_cone(double mn, double mx);
  static Shape capped_cone(double mn, double mx);
  static Shape triangle(Tuple p1, Tuple p2, Tuple p3);
  static Shape smooth_triangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3);
  static Shape group();
  static Shape csg(ShapeType op, Shape l, Shape r);

  bool is_csg() const {
    return ty == ShapeType::Union || ty == ShapeType::Intersection || ty == ShapeType::Difference;
  }
  void add_child(Shape child);
  void set_transform(const Matrix& m);
  void propagate_parent_transform(const Matrix& parent);
  bool includes(const Shape& other) const;
  Bounds bounds() const;
  Tuple world_to_object(Tuple p) const;
  Tuple normal_to_world(Tuple n) const;
  void local_intersect(const Ray& r, Intersections& out) const;
  Tuple local_normal_at(Tuple at, const Intersection& hit) const;
};
bool shape_eq(const Shape& a, const Shape& b);
void intersect(const Shape& s, const Ray& r, Intersections& out);  // shape.rs:1467
Tuple normal_at(const Shape& s, Tuple world_point, const Intersection& hit);

// ------------------------------------------------------------ intersect.rs
struct Intersection {
  double t = 0;
  const Shape* what = nullptr;
  bool has_uv = false;
  double u = 0, v = 0;
};
struct Intersections {
  std::vector<Intersection> xs;
  void sort();
  bool hit(Intersection* out);  // mutates: drops non-finite, sorts
};
struct Computation {
  double t;
  const Shape* obj;
  Tuple point, over_point, under_point, eyev, normalv, reflectv;
  bool inside;
  double n1, n2;
};
Computation computation_new(const Ray& r, const Intersection& hit, const Intersections* xs);
double schlick(const Computation& c);
bool intersection_allowed(ShapeType op, bool which_hit, bool inl, bool inr);

// ---------------------------------------------------------------- world.rs
// One record per World::color_at call when tracing (debugging aid of the tests): the ray, what it
// hit, the refraction indices and the colour the call returned.
struct TraceRecord {
  double ray[6];
  int remaining, leaf;
  double t, n1, n2;
  double color[3];
};
struct RayCounts {
  uint64_t primary = 0, shadow = 0, reflect = 0, refract = 0;
  std::vector<TraceRecord>* trace = nullptr;
};
struct World {
  std::vector<Shape> objects;
  PointLight light;
  int n_leaves = 0;
  static World default_world();
  static World empty_world();
  void finalize();  // number leaves, warm bounds caches (single-threaded)
  void intersect(const Ray& r, Intersections& out) const;
  bool is_shadowed(Tuple p, RayCounts* rc) const;
  Color reflected_color(const Computation& c, int remaining, RayCounts* rc) const;
  Color refracted_color(const Computation& c, int remaining, RayCounts* rc) const;
  Color shade_hit(const Computation& c, int remaining, RayCounts* rc) const;
  Color color_at(const Ray& r, int remaining, RayCounts* rc, int* hit_leaf = nullptr) const;
};
Color lighting(const Material& m, const Shape& obj, const PointLight& light, Tuple pt, Tuple eyev,
               Tuple normalv, bool in_shadow);

// --------------------------------------------------------------- camera.rs
struct Camera {
  size_t hsize = 0, vsize = 0;
  double half_width = 0, half_height = 0, pixel_size = 0, field_of_view = 0;
  Matrix transform = mat_identity();
  static Camera make(size_t hsize, size_t vsize, double fov, const Matrix& transform);
  Ray ray_for_pixel(size_t px, size_t py) const;
};

// --------------------------------------------------------------- canvas.rs
uint32_t quantise(double c);                       // canvas.rs:60-62
std::string ppm_encode(size_t w, size_t h, const double* rgb);  // canvas.rs:47-94

// ------------------------------------------------------- scene.rs / obj.rs
struct Scene {
  World world;
  Camera camera;
};
// Panics in the reference surface as std::runtime_error here.
Scene scene_from_json(const std::string& text, const std::string& base_dir);
struct ObjResult {
  std::map<std::string, Shape> groups;
  std::vector<Tuple> vertices, normals;
  size_t ignored_lines = 0;
};
ObjResult obj_parse(const std::string& path);
ObjResult obj_parse_text(const std::string& text);

// -------------------------------------------------------------- parallel.rs
struct RenderOut {
  std::vector<double> rgb;       // W*H*3, row-major y*W+x (canvas.rs:120)
  std::vector<int32_t> hit_leaf; // W*H, primary hit leaf id or -1
  RayCounts counts;
};
// Renders pixels (x0 + i*xstep, y0 + j*ystep); steps of 1 and offsets of 0
// give the full image.  Untouched pixels stay black / -1.
void parallel_render(const World& w, const Camera& c, int jobs, size_t x0, size_t xstep, size_t y0,
                     size_t ystep, RenderOut* out);

}  // namespace orc
