// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp.  CPU restatement of the
// reference ray tracer's render path; citations are reference file:line.
#include "rt_oracle.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <thread>

namespace orc {

static const double INF = std::numeric_limits<double>::infinity();

// Rust f64::max / f64::min ignore a NaN operand; C fmax / fmin do the same.
static inline double rmax(double a, double b) { return std::fmax(a, b); }
static inline double rmin(double a, double b) { return std::fmin(a, b); }

// ======================================================================
// tuple.rs
// ======================================================================
Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }  // 179-190
Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }  // 196-207
Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }                               // 212-223
Tuple operator*(Tuple a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }         // 238-249
Tuple operator*(double s, Tuple a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }         // 264-275

// tuple.rs:145-150 -- w takes part, summed left to right
double dot(Tuple a, Tuple b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// tuple.rs:111-118 (powi(2) == x*x)
double magnitude(Tuple a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }

// tuple.rs:126-135 -- multiply by the reciprocal, not divide
Tuple normalize(Tuple a) {
  double mag = magnitude(a);
  return {a.x * (1.0 / mag), a.y * (1.0 / mag), a.z * (1.0 / mag), a.w * (1.0 / mag)};
}

// tuple.rs:160-167
Tuple cross(Tuple a, Tuple b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0};
}

// tuple.rs:170-172: *self - (*normal * 2.0 * self.dot(normal))
Tuple reflect(Tuple v, Tuple n) { return v - ((n * 2.0) * dot(v, n)); }

bool tuple_eq(Tuple a, Tuple b) {
  return feq(a.x, b.x) && feq(a.y, b.y) && feq(a.z, b.z) && feq(a.w, b.w);
}

Tuple tuple_from_vec(const std::vector<double>& v) {
  Tuple t;
  if (v.size() > 0) t.x = v[0];
  if (v.size() > 1) t.y = v[1];
  if (v.size() > 2) t.z = v[2];
  if (v.size() > 3) t.w = v[3];
  return t;
}

// ======================================================================
// color.rs
// ======================================================================
Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
Color operator*(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }
Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // 141-147
Color color_average(Color a, Color b) {                                          // 169-175
  return {(a.r + b.r) / 2.0, (a.g + b.g) / 2.0, (a.b + b.b) / 2.0};
}
bool color_eq(Color a, Color b) { return feq(a.r, b.r) && feq(a.g, b.g) && feq(a.b, b.b); }
Color color_from_vec(const std::vector<double>& v) {
  Color c;
  if (v.size() > 0) c.r = v[0];
  if (v.size() > 1) c.g = v[1];
  if (v.size() > 2) c.b = v[2];
  return c;
}

// ======================================================================
// matrix.rs
// ======================================================================
Matrix mat_identity() {
  Matrix m;
  m.d[0] = m.d[5] = m.d[10] = m.d[15] = 1.0;
  return m;
}
Matrix mat_translation(double x, double y, double z) {
  Matrix m = mat_identity();
  m.at(0, 3) = x; m.at(1, 3) = y; m.at(2, 3) = z;
  return m;
}
Matrix mat_scaling(double x, double y, double z) {
  Matrix m = mat_identity();
  m.at(0, 0) = x; m.at(1, 1) = y; m.at(2, 2) = z;
  return m;
}
Matrix mat_rotation_x(double r) {  // 278-286
  Matrix m = mat_identity();
  m.at(1, 1) = std::cos(r); m.at(1, 2) = -std::sin(r);
  m.at(2, 1) = std::sin(r); m.at(2, 2) = std::cos(r);
  return m;
}
Matrix mat_rotation_y(double r) {  // 304-312
  Matrix m = mat_identity();
  m.at(0, 0) = std::cos(r); m.at(0, 2) = std::sin(r);
  m.at(2, 0) = -std::sin(r); m.at(2, 2) = std::cos(r);
  return m;
}
Matrix mat_rotation_z(double r) {  // 330-338
  Matrix m = mat_identity();
  m.at(0, 0) = std::cos(r); m.at(0, 1) = -std::sin(r);
  m.at(1, 0) = std::sin(r); m.at(1, 1) = std::cos(r);
  return m;
}
Matrix mat_shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
  Matrix m = mat_identity();
  m.at(0, 1) = xy; m.at(0, 2) = xz; m.at(1, 0) = yx;
  m.at(1, 2) = yz; m.at(2, 0) = zx; m.at(2, 1) = zy;
  return m;
}

// matrix.rs:604-621
Matrix operator*(const Matrix& a, const Matrix& b) {
  Matrix r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r.at(i, j) = a.at(i, 0) * b.at(0, j) + a.at(i, 1) * b.at(1, j) + a.at(i, 2) * b.at(2, j) +
                   a.at(i, 3) * b.at(3, j);
  return r;
}

// matrix.rs:638-653 -- all four rows, w included
Tuple operator*(const Matrix& m, Tuple t) {
  double b[4];
  for (int r = 0; r < 4; r++)
    b[r] = m.at(r, 0) * t.x + m.at(r, 1) * t.y + m.at(r, 2) * t.z + m.at(r, 3) * t.w;
  return {b[0], b[1], b[2], b[3]};
}

bool mat_eq(const Matrix& a, const Matrix& b) {
  for (int i = 0; i < 16; i++)
    if (!feq(a.d[i], b.d[i])) return false;
  return true;
}

// matrix.rs:415-434
Matrix mat_view_transform(Tuple from, Tuple to, Tuple up) {
  Tuple forward = normalize(to - from);
  Tuple left = cross(forward, normalize(up));
  Tuple true_up = cross(left, forward);
  Matrix o = mat_identity();
  o.at(0, 0) = left.x; o.at(0, 1) = left.y; o.at(0, 2) = left.z;
  o.at(1, 0) = true_up.x; o.at(1, 1) = true_up.y; o.at(1, 2) = true_up.z;
  o.at(2, 0) = -forward.x; o.at(2, 1) = -forward.y; o.at(2, 2) = -forward.z;
  return o * mat_translation(-from.x, -from.y, -from.z);
}

// matrix.rs:461-473
Matrix mat_transposition(const Matrix& m) {
  Matrix b = m;
  for (int r = 0; r < 4; r++)
    for (int c = r + 1; c < 4; c++) std::swap(b.at(r, c), b.at(c, r));
  return b;
}

// matrix.rs:122-124
static double det2(const double* m) { return m[0] * m[3] - m[1] * m[2]; }

// matrix.rs:168-210 -- 3x3 determinant by cofactor expansion along row 0
static double det3(const double* m) {
  double sum = 0.0;
  for (int c = 0; c < 3; c++) {
    double sub[4];
    int n = 0;
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++)
        if (!(r == 0 || cc == c)) sub[n++] = m[r * 3 + cc];
    double minor = det2(sub);
    double cof = minor * (((0 + c) % 2 == 0) ? 1.0 : -1.0);
    sum += m[c] * cof;
  }
  return sum;
}

// matrix.rs:502-534
double mat_cofactor(const Matrix& m, int row, int col) {
  double sub[9];
  int n = 0;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      if (!(r == row || c == col)) sub[n++] = m.at(r, c);
  double minor = det3(sub);
  return minor * (((row + col) % 2 == 0) ? 1.0 : -1.0);
}

// matrix.rs:537-544
double mat_determinant(const Matrix& m) {
  double sum = 0.0;
  for (int c = 0; c < 4; c++) sum += m.at(0, c) * mat_cofactor(m, 0, c);
  return sum;
}

// matrix.rs:550-565
bool mat_inverse(const Matrix& m, Matrix* out) {
  double det = mat_determinant(m);
  if (det == 0.0) return false;
  Matrix inv;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) inv.at(c, r) = mat_cofactor(m, r, c) / det;
  *out = inv;
  return true;
}

Matrix mat_inverse_or_throw(const Matrix& m, const char* what) {
  Matrix inv;
  if (!mat_inverse(m, &inv)) throw std::runtime_error(what);
  return inv;
}

// ======================================================================
// ray.rs
// ======================================================================
Ray ray_new(Tuple origin, Tuple direction) {  // 20-30
  if (!feq(origin.w, 1.0)) origin.w = 1.0;
  if (!feq(direction.w, 0.0)) direction.w = 0.0;
  return {origin, direction};
}
Tuple ray_position(const Ray& r, double t) { return r.origin + (t * r.direction); }  // 42-44
Ray ray_transform(const Ray& r, const Matrix& m) { return {m * r.origin, m * r.direction}; }  // 50-55

// ======================================================================
// pattern.rs
// ======================================================================
Pattern pattern_make(PatternType ty, Color a, Color b) {
  Pattern p;
  p.ty = ty; p.a = a; p.b = b;
  return p;
}
Pattern pattern_blend(const Pattern& f, const Pattern& s) {
  Pattern p;
  p.ty = PatternType::Blend;
  p.first = std::make_shared<Pattern>(f);
  p.second = std::make_shared<Pattern>(s);
  return p;
}

// Rust f64::rem_euclid
static double rem_euclid(double x, double rhs) {
  double r = std::fmod(x, rhs);
  return r < 0.0 ? r + std::fabs(rhs) : r;
}

// pattern.rs:97-189
Color pattern_at(const Pattern& p, Tuple pt) {
  switch (p.ty) {
    case PatternType::Null: return {0, 0, 0};
    case PatternType::Point: return {pt.x, pt.y, pt.z};
    case PatternType::Identity: return p.a;
    case PatternType::Stripe:
      return feq(rem_euclid(std::floor(pt.x), 2.0), 0.0) ? p.a : p.b;
    case PatternType::Ring: {
      // powf(2.0) of a finite double equals x*x after rounding in every libm in use
      double pyth = std::sqrt(std::pow(pt.x, 2.0) + std::pow(pt.z, 2.0));
      return feq(rem_euclid(std::floor(pyth), 2.0), 0.0) ? p.a : p.b;
    }
    case PatternType::Checker: {
      double checker = std::floor(pt.x) + std::floor(pt.y) + std::floor(pt.z);
      return feq(rem_euclid(checker, 2.0), 0.0) ? p.a : p.b;
    }
    case PatternType::Gradient: {
      Color distance = p.b - p.a;
      double fraction = pt.x - std::floor(pt.x);
      return p.a + distance * fraction;
    }
    case PatternType::Blend:
      return color_average(pattern_at(*p.first, pt), pattern_at(*p.second, pt));
  }
  return {0, 0, 0};
}

bool pattern_eq(const Pattern& a, const Pattern& b) {
  if (a.ty != b.ty || !mat_eq(a.transform, b.transform)) return false;
  switch (a.ty) {
    case PatternType::Null:
    case PatternType::Point: return true;
    case PatternType::Identity: return color_eq(a.a, b.a);
    case PatternType::Blend: return pattern_eq(*a.first, *b.first) && pattern_eq(*a.second, *b.second);
    default: return color_eq(a.a, b.a) && color_eq(a.b, b.b);
  }
}

// pattern.rs:197-208 (the inverse is recomputed per call there; same value)
static Color pattern_at_object(const Pattern& p, const Shape& obj, Tuple world_point) {
  Matrix pti = mat_inverse_or_throw(p.transform, "Pattern transform should be invertible.");
  Tuple object_point = obj.world_to_object(world_point);
  Tuple pattern_point = pti * object_point;
  return pattern_at(p, pattern_point);
}

// ======================================================================
// light.rs
// ======================================================================
PointLight point_light_new(Color intensity, Tuple position) {  // 20-26
  if (!feq(position.w, 1.0)) position.w = 1.0;
  return {intensity, position};
}

bool material_eq(const Material& a, const Material& b) {
  if (!color_eq(a.color, b.color)) return false;
  if (a.pattern.has_value() != b.pattern.has_value()) return false;
  if (a.pattern && !pattern_eq(*a.pattern, *b.pattern)) return false;
  return a.ambient == b.ambient && a.diffuse == b.diffuse && a.specular == b.specular &&
         a.shininess == b.shininess && a.reflective == b.reflective &&
         a.refractive_index == b.refractive_index && a.transparency == b.transparency;
}

// light.rs:125-174
Color lighting(const Material& m, const Shape& obj, const PointLight& light, Tuple pt, Tuple eyev,
               Tuple normalv, bool in_shadow) {
  Color color = m.pattern ? pattern_at_object(*m.pattern, obj, pt) : obj.material.color;
  Color effective_color = color * light.intensity;
  Tuple lightv = normalize(light.position - pt);
  Color ambient = effective_color * m.ambient;
  if (in_shadow) return ambient;

  Color diffuse, specular;
  double light_dot_normal = dot(lightv, normalv);
  if (light_dot_normal < 0.0) {
    diffuse = {0, 0, 0};
    specular = {0, 0, 0};
  } else {
    diffuse = effective_color * m.diffuse * light_dot_normal;
    Tuple reflectv = reflect(-lightv, normalv);
    double reflect_dot_eye = dot(reflectv, eyev);
    if (reflect_dot_eye <= 0.0) {
      specular = {0, 0, 0};
    } else {
      double factor = std::pow(reflect_dot_eye, m.shininess);
      specular = light.intensity * m.specular * factor;
    }
  }
  return ambient + diffuse + specular;
}

// ======================================================================
// geometry.rs
// ======================================================================
TriangleInfo triangle_info_new(Tuple p1, Tuple p2, Tuple p3) {  // 28-33
  TriangleInfo t;
  t.p1 = p1; t.p2 = p2; t.p3 = p3;
  t.e1 = p2 - p1;
  t.e2 = p3 - p1;
  t.normal = normalize(cross(t.e2, t.e1));
  return t;
}

Bounds bounds_new(double minx, double miny, double minz, double maxx, double maxy, double maxz) {
  return {point(minx, miny, minz), point(maxx, maxy, maxz)};
}

// geometry.rs:118-150 -- eight corners, NaN-ignoring min/max
Bounds bounds_transform(const Bounds& b, const Matrix& m) {
  Tuple corners[8] = {
      m * b.minimum,
      m * point(b.minimum.x, b.minimum.y, b.maximum.z),
      m * point(b.minimum.x, b.maximum.y, b.minimum.z),
      m * point(b.minimum.x, b.maximum.y, b.maximum.z),
      m * point(b.maximum.x, b.minimum.y, b.minimum.z),
      m * point(b.maximum.x, b.minimum.y, b.maximum.z),
      m * point(b.maximum.x, b.maximum.y, b.minimum.z),
      m * b.maximum,
  };
  double mnx = INF, mny = INF, mnz = INF, mxx = -1.0 * INF, mxy = -1.0 * INF, mxz = -1.0 * INF;
  for (const Tuple& c : corners) {
    mnx = rmin(mnx, c.x); mny = rmin(mny, c.y); mnz = rmin(mnz, c.z);
    mxx = rmax(mxx, c.x); mxy = rmax(mxy, c.y); mxz = rmax(mxz, c.z);
  }
  return bounds_new(mnx, mny, mnz, mxx, mxy, mxz);
}

// geometry.rs:160-179
void bounds_check_axis(double mn, double mx, double origin, double direction, double* tmin_out,
                       double* tmax_out) {
  double tmin_numerator = mn - origin;
  double tmax_numerator = mx - origin;
  double tmin, tmax;
  if (std::fabs(direction) >= FEQ_EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INF;
    tmax = tmax_numerator * INF;
  }
  if (tmin > tmax) std::swap(tmin, tmax);
  *tmin_out = tmin;
  *tmax_out = tmax;
}

// ======================================================================
// intersect.rs
// ======================================================================
// intersect.rs:85-89 -- stable sort by t, incomparable (NaN) pairs count as equal
static void stable_sort_range(std::vector<Intersection>& v, size_t lo, size_t hi) {
  size_t n = hi - lo;
  if (n < 2) return;
  if (n <= 24) {  // stable insertion sort; same result as a stable merge sort
    for (size_t i = lo + 1; i < hi; i++) {
      Intersection key = v[i];
      size_t j = i;
      while (j > lo && key.t < v[j - 1].t) {
        v[j] = v[j - 1];
        j--;
      }
      v[j] = key;
    }
    return;
  }
  std::stable_sort(v.begin() + lo, v.begin() + hi,
                   [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
}
void Intersections::sort() { stable_sort_range(xs, 0, xs.size()); }

// intersect.rs:71-82
bool Intersections::hit(Intersection* out) {
  xs.erase(std::remove_if(xs.begin(), xs.end(),
                          [](const Intersection& i) { return !std::isfinite(i.t); }),
           xs.end());
  sort();
  for (const Intersection& i : xs)
    if (i.t >= 0.0) {
      *out = i;
      return true;
    }
  return false;
}

// intersect.rs:40-44
static bool isect_eq(const Intersection& a, const Intersection& b) {
  return a.t == b.t && a.what == b.what;
}

// intersect.rs:191-234
static void refraction_indices(const Intersection& hit, const Intersections& xs, double* n1o,
                               double* n2o) {
  double n1 = 1.0, n2 = 1.0;
  std::vector<const Shape*> containers;
  for (const Intersection& i : xs.xs) {
    if (isect_eq(i, hit)) {
      n1 = containers.empty() ? 1.0 : containers.back()->material.refractive_index;
    }
    auto it = std::find(containers.begin(), containers.end(), i.what);
    if (it != containers.end())
      containers.erase(it);
    else
      containers.push_back(i.what);
    if (isect_eq(i, hit)) {
      n2 = containers.empty() ? 1.0 : containers.back()->material.refractive_index;
      break;
    }
  }
  *n1o = n1;
  *n2o = n2;
}

// intersect.rs:152-184
Computation computation_new(const Ray& r, const Intersection& hit, const Intersections* xs) {
  Computation c;
  c.t = hit.t;
  c.obj = hit.what;
  c.point = ray_position(r, c.t);
  c.eyev = -r.direction;
  c.normalv = normal_at(*c.obj, c.point, hit);
  if (dot(c.normalv, c.eyev) < 0.0) {
    c.normalv = -c.normalv;
    c.inside = true;
  } else {
    c.inside = false;
  }
  c.over_point = c.point + c.normalv * FEQ_EPSILON;
  c.under_point = c.point - c.normalv * FEQ_EPSILON;
  c.reflectv = reflect(r.direction, c.normalv);
  if (xs) {
    refraction_indices(hit, *xs, &c.n1, &c.n2);
  } else {
    c.n1 = 1.0;
    c.n2 = 1.0;
  }
  return c;
}

// x.powi(5): x * ((x*x) * (x*x)) in both LLVM's expansion and compiler-rt
static inline double powi5(double x) {
  double x2 = x * x;
  return x * (x2 * x2);
}

// intersect.rs:240-260
double schlick(const Computation& c) {
  double cosv = dot(c.eyev, c.normalv);
  if (c.n1 > c.n2) {
    double n = c.n1 / c.n2;
    double sin2_t = (n * n) * (1.0 - cosv * cosv);
    if (sin2_t > 1.0) return 1.0;
    cosv = std::sqrt(1.0 - sin2_t);
  }
  double q = (c.n1 - c.n2) / (c.n1 + c.n2);
  double r0 = q * q;
  return r0 + (1.0 - r0) * powi5(1.0 - cosv);
}

// intersect.rs:278-290
bool intersection_allowed(ShapeType op, bool which_hit, bool inl, bool inr) {
  switch (op) {
    case ShapeType::Union: return (which_hit && !inr) || (!which_hit && !inl);
    case ShapeType::Intersection: return (which_hit && inr) || (!which_hit && inl);
    case ShapeType::Difference: return (which_hit && !inr) || (!which_hit && inl);
    default: return false;
  }
}

// ======================================================================
// shape.rs
// ======================================================================
Shape Shape::empty() { return Shape(); }
Shape Shape::sphere() { Shape s; s.ty = ShapeType::Sphere; return s; }
Shape Shape::plane() { Shape s; s.ty = ShapeType::Plane; s.plane_normal = vector(0.0, 1.0, 0.0); return s; }
Shape Shape::cube() { Shape s; s.ty = ShapeType::Cube; return s; }
Shape Shape::cylinder() {  // 245-254
  Shape s; s.ty = ShapeType::Cylinder; s.minimum = -1.0 * INF; s.maximum = INF; s.closed = false; return s;
}
Shape Shape::bounded_cylinder(double mn, double mx) {
  Shape s; s.ty = ShapeType::Cylinder; s.minimum = mn; s.maximum = mx; s.closed = false; return s;
}
Shape Shape::capped_cylinder(double mn, double mx) {
  Shape s; s.ty = ShapeType::Cylinder; s.minimum = mn; s.maximum = mx; s.closed = true; return s;
}
Shape Shape::bounded_cone(double mn, double mx) {
  Shape s; s.ty = ShapeType::Cone; s.minimum = mn; s.maximum = mx; s.closed = false; return s;
}
Shape Shape::capped_cone(double mn, double mx) {
  Shape s; s.ty = ShapeType::Cone; s.minimum = mn; s.maximum = mx; s.closed = true; return s;
}
Shape Shape::triangle(Tuple p1, Tuple p2, Tuple p3) {
  Shape s; s.ty = ShapeType::Triangle; s.tri = triangle_info_new(p1, p2, p3); return s;
}
Shape Shape::smooth_triangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3) {
  Shape s; s.ty = ShapeType::SmoothTriangle; s.tri = triangle_info_new(p1, p2, p3);
  s.n1 = n1; s.n2 = n2; s.n3 = n3; return s;
}
Shape Shape::group() { Shape s; s.ty = ShapeType::Group; return s; }
Shape Shape::csg(ShapeType op, Shape l, Shape r) {  // 319-364: operands stored as they are
  Shape s; s.ty = op; s.children.push_back(std::move(l)); s.children.push_back(std::move(r)); return s;
}

// shape.rs:69-102, 203-209
static bool shape_type_eq(const Shape& a, const Shape& b) {
  if (a.ty != b.ty) return false;
  switch (a.ty) {
    case ShapeType::Empty: case ShapeType::Sphere: case ShapeType::Cube: return true;
    case ShapeType::Cone: case ShapeType::Cylinder:
      return a.minimum == b.minimum && a.maximum == b.maximum && a.closed == b.closed;
    case ShapeType::Plane: return tuple_eq(a.plane_normal, b.plane_normal);
    case ShapeType::Triangle:
    case ShapeType::SmoothTriangle: {
      bool t = tuple_eq(a.tri.p1, b.tri.p1) && tuple_eq(a.tri.p2, b.tri.p2) &&
               tuple_eq(a.tri.p3, b.tri.p3) && tuple_eq(a.tri.e1, b.tri.e1) &&
               tuple_eq(a.tri.e2, b.tri.e2) && tuple_eq(a.tri.normal, b.tri.normal);
      if (a.ty == ShapeType::Triangle) return t;
      return t && tuple_eq(a.n1, b.n1) && tuple_eq(a.n2, b.n2) && tuple_eq(a.n3, b.n3);
    }
    case ShapeType::Group:
      if (a.children.size() != b.children.size()) return false;
      for (size_t i = 0; i < a.children.size(); i++)
        if (!shape_eq(a.children[i], b.children[i])) return false;
      return true;
    default: return false;  // CSG operators never compare equal (shape.rs:98-99)
  }
}
bool shape_eq(const Shape& a, const Shape& b) {
  return shape_type_eq(a, b) && mat_eq(a.transform, b.transform) && material_eq(a.material, b.material);
}

// shape.rs:369-386
void Shape::add_child(Shape child) {
  if (ty != ShapeType::Group) throw std::runtime_error("Cannot add child to non-group shape.");
  Matrix to_propagate = has_parent ? parent_transform * transform : transform;
  child.propagate_parent_transform(to_propagate);
  children.push_back(std::move(child));
}

// shape.rs:592-656
void Shape::set_transform(const Matrix& m) {
  transform = m;
  transform_inverse = mat_inverse_or_throw(transform, "Shape transform should have an inverse.");
  if (ty == ShapeType::Group || is_csg()) {
    for (Shape& child : children) {
      Matrix to_propagate = has_parent ? parent_transform * transform : transform;
      child.propagate_parent_transform(to_propagate);
    }
  }
}

// shape.rs:681-736
void Shape::propagate_parent_transform(const Matrix& parent) {
  if (ty == ShapeType::Group || is_csg()) {
    for (Shape& child : children) child.propagate_parent_transform(parent * transform);
  }
  parent_transform = parent;
  has_parent = true;
  full_transform_inverse = mat_inverse_or_throw(
      parent * transform, "Shape parent transform and transform should be invertible.");
  has_full = true;
}

// shape.rs:450-467
bool Shape::includes(const Shape& other) const {
  if (ty == ShapeType::Group) {
    for (const Shape& c : children)
      if (c.includes(other)) return true;
    return false;
  }
  if (is_csg()) return children[0].includes(other) || children[1].includes(other);
  return shape_eq(*this, other);
}

// shape.rs:470-544
Bounds Shape::bounds() const {
  if (has_saved_bounds) return saved_bounds;
  Bounds b;
  switch (ty) {
    case ShapeType::Empty: b = bounds_new(0, 0, 0, 0, 0, 0); break;
    case ShapeType::Sphere:
    case ShapeType::Cube: b = bounds_new(-1.0, -1.0, -1.0, 1.0, 1.0, 1.0); break;
    case ShapeType::Cone:
    case ShapeType::Cylinder: b = bounds_new(minimum, minimum, minimum, maximum, maximum, maximum); break;
    case ShapeType::Triangle:
    case ShapeType::SmoothTriangle: {
      const Tuple &p1 = tri.p1, &p2 = tri.p2, &p3 = tri.p3;
      b = bounds_new(rmin(p1.x, rmin(p2.x, p3.x)), rmin(p1.y, rmin(p2.y, p3.y)),
                     rmin(p1.z, rmin(p2.z, p3.z)), rmax(p1.x, rmax(p2.x, p3.x)),
                     rmax(p1.y, rmax(p2.y, p3.y)), rmax(p1.z, rmax(p2.z, p3.z)));
      break;
    }
    case ShapeType::Group: {
      double mnx = INF, mny = INF, mnz = INF, mxx = -1.0 * INF, mxy = -1.0 * INF, mxz = -1.0 * INF;
      for (const Shape& child : children) {
        // child box moved by the GROUP's own transform (shape.rs:513)
        Bounds cb = bounds_transform(child.bounds(), transform);
        mnx = rmin(mnx, cb.minimum.x); mny = rmin(mny, cb.minimum.y); mnz = rmin(mnz, cb.minimum.z);
        mxx = rmax(mxx, cb.maximum.x); mxy = rmax(mxy, cb.maximum.y); mxz = rmax(mxz, cb.maximum.z);
      }
      b = bounds_new(mnx, mny, mnz, mxx, mxy, mxz);
      break;
    }
    default:  // planes and CSG: everything (geometry.rs:183-193)
      b = bounds_new(-1.0 * INF, -1.0 * INF, -1.0 * INF, INF, INF, INF);
  }
  saved_bounds = b;
  has_saved_bounds = true;
  return b;
}

// shape.rs:757-765
Tuple Shape::world_to_object(Tuple p) const {
  return (has_full ? full_transform_inverse : transform_inverse) * p;
}

// shape.rs:775-784
Tuple Shape::normal_to_world(Tuple n) const {
  const Matrix& fti = has_full ? full_transform_inverse : transform_inverse;
  n = mat_transposition(fti) * n;
  n.w = 0.0;
  return normalize(n);
}

static inline void push_isect(Intersections& out, double t, const Shape* s) {
  Intersection i;
  i.t = t;
  i.what = s;
  out.xs.push_back(i);
}

// shape.rs:1367-1385
static void check_cube_axis(double origin, double direction, double* tmin_out, double* tmax_out) {
  double tmin_numerator = -1.0 - origin;
  double tmax_numerator = 1.0 - origin;
  double tmin, tmax;
  if (std::fabs(direction) >= FEQ_EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INF;
    tmax = tmax_numerator * INF;
  }
  if (tmin > tmax) std::swap(tmin, tmax);
  *tmin_out = tmin;
  *tmax_out = tmax;
}

// shape.rs:1415-1420
static bool check_cylinder_cap(const Ray& ray, double t) {
  double x = ray.origin.x + t * ray.direction.x;
  double z = ray.origin.z + t * ray.direction.z;
  return (x * x + z * z) <= 1.0;
}

// shape.rs:1388-1411
static void intersect_cylinder_caps(const Shape& s, const Ray& ray, Intersections& out) {
  if (!s.closed || std::fabs(ray.direction.y) < FEQ_EPSILON) return;
  double tl = (s.minimum - ray.origin.y) / ray.direction.y;
  if (check_cylinder_cap(ray, tl)) push_isect(out, tl, &s);
  double tu = (s.maximum - ray.origin.y) / ray.direction.y;
  if (check_cylinder_cap(ray, tu)) push_isect(out, tu, &s);
}

// shape.rs:1450-1455
static bool check_cone_cap(const Ray& ray, double t, double y) {
  double x = ray.origin.x + t * ray.direction.x;
  double z = ray.origin.z + t * ray.direction.z;
  return x * x + z * z <= y * y;
}

// shape.rs:1423-1446 -- returns early when the cone IS closed (reference behaviour)
static void intersect_cone_caps(const Shape& s, const Ray& ray, Intersections& out) {
  if (s.closed || std::fabs(ray.direction.y) < FEQ_EPSILON) return;
  double tl = (s.minimum - ray.origin.y) / ray.direction.y;
  if (check_cone_cap(ray, tl, s.minimum)) push_isect(out, tl, &s);
  double tu = (s.maximum - ray.origin.y) / ray.direction.y;
  if (check_cone_cap(ray, tu, s.maximum)) push_isect(out, tu, &s);
}

static void filter_intersections(const Shape& op, const Intersections& all, Intersections& out);

// shape.rs:787-802 with the per-type routines inlined
void Shape::local_intersect(const Ray& ray, Intersections& out) const {
  switch (ty) {
    case ShapeType::Empty: return;
    case ShapeType::Sphere: {  // 839-866
      Tuple sphere_to_ray = ray.origin - point(0.0, 0.0, 0.0);
      double a = dot(ray.direction, ray.direction);
      double b = 2.0 * dot(ray.direction, sphere_to_ray);
      double c = dot(sphere_to_ray, sphere_to_ray) - 1.0;
      double discriminant = b * b - (4.0 * a * c);
      if (discriminant < 0.0) return;
      double t1 = (-b - std::sqrt(discriminant)) / (2.0 * a);
      double t2 = (-b + std::sqrt(discriminant)) / (2.0 * a);
      push_isect(out, t1, this);
      push_isect(out, t2, this);
      return;
    }
    case ShapeType::Plane: {  // 891-908
      if (std::fabs(ray.direction.y) <= FEQ_EPSILON) return;
      double t = -ray.origin.y / ray.direction.y;
      push_isect(out, t, this);
      return;
    }
    case ShapeType::Cube: {  // 925-953
      double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
      check_cube_axis(ray.origin.x, ray.direction.x, &xtmin, &xtmax);
      check_cube_axis(ray.origin.y, ray.direction.y, &ytmin, &ytmax);
      check_cube_axis(ray.origin.z, ray.direction.z, &ztmin, &ztmax);
      double tmin = rmax(rmax(xtmin, ytmin), ztmin);
      double tmax = rmin(rmin(xtmax, ytmax), ztmax);
      if (tmin > tmax) return;
      push_isect(out, tmin, this);
      push_isect(out, tmax, this);
      return;
    }
    case ShapeType::Cylinder: {  // 981-1040
      double a = ray.direction.x * ray.direction.x + ray.direction.z * ray.direction.z;
      if (a < FEQ_EPSILON) {
        intersect_cylinder_caps(*this, ray, out);
        return;
      }
      double b = 2.0 * ray.origin.x * ray.direction.x + 2.0 * ray.origin.z * ray.direction.z;
      double c = ray.origin.x * ray.origin.x + ray.origin.z * ray.origin.z - 1.0;
      double disc = b * b - 4.0 * a * c;
      if (disc < 0.0) return;
      double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
      double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
      if (t0 > t1) std::swap(t0, t1);
      double y0 = ray.origin.y + t0 * ray.direction.y;
      if (minimum < y0 && y0 < maximum) push_isect(out, t0, this);
      double y1 = ray.origin.y + t1 * ray.direction.y;
      if (minimum < y1 && y1 < maximum) push_isect(out, t1, this);
      intersect_cylinder_caps(*this, ray, out);
      return;
    }
    case ShapeType::Cone: {  // 1071-1144
      double a = ray.direction.x * ray.direction.x - ray.direction.y * ray.direction.y +
                 ray.direction.z * ray.direction.z;
      double b = 2.0 * ray.origin.x * ray.direction.x - 2.0 * ray.origin.y * ray.direction.y +
                 2.0 * ray.origin.z * ray.direction.z;
      double c = ray.origin.x * ray.origin.x - ray.origin.y * ray.origin.y +
                 ray.origin.z * ray.origin.z;
      if (std::fabs(a) < FEQ_EPSILON) {
        intersect_cone_caps(*this, ray, out);
        if (std::fabs(b) < FEQ_EPSILON) return;
        double t = -c / (2.0 * b);
        push_isect(out, t, this);
        return;
      }
      double disc = b * b - 4.0 * a * c;
      if (disc < 0.0) return;
      double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
      double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
      if (t0 > t1) std::swap(t0, t1);
      double y0 = ray.origin.y + t0 * ray.direction.y;
      if (minimum < y0 && y0 < maximum) push_isect(out, t0, this);
      double y1 = ray.origin.y + t1 * ray.direction.y;
      if (minimum < y1 && y1 < maximum) push_isect(out, t1, this);
      intersect_cone_caps(*this, ray, out);
      return;
    }
    case ShapeType::Triangle:
    case ShapeType::SmoothTriangle: {  // 1225-1260, 1273-1315
      Tuple dir_cross_e2 = cross(ray.direction, tri.e2);
      double determinant = dot(tri.e1, dir_cross_e2);
      if (std::fabs(determinant) < FEQ_EPSILON) return;
      double f = 1.0 / determinant;
      Tuple p1_to_origin = ray.origin - tri.p1;
      double u = f * dot(p1_to_origin, dir_cross_e2);
      if (u < 0.0) return;
      Tuple origin_cross_e1 = cross(p1_to_origin, tri.e1);
      double v = f * dot(ray.direction, origin_cross_e1);
      if (v < 0.0 || u + v > 1.0) return;
      double t = f * dot(tri.e2, origin_cross_e1);
      Intersection i;
      i.t = t;
      i.what = this;
      if (ty == ShapeType::SmoothTriangle) {
        i.has_uv = true;
        i.u = u;
        i.v = v;
      }
      out.xs.push_back(i);
      return;
    }
    case ShapeType::Group: {  // 1179-1222
      if (children.empty()) return;
      Bounds b = bounds();
      double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
      bounds_check_axis(b.minimum.x, b.maximum.x, ray.origin.x, ray.direction.x, &xtmin, &xtmax);
      bounds_check_axis(b.minimum.y, b.maximum.y, ray.origin.y, ray.direction.y, &ytmin, &ytmax);
      bounds_check_axis(b.minimum.z, b.maximum.z, ray.origin.z, ray.direction.z, &ztmin, &ztmax);
      double tmin = rmax(rmax(xtmin, ytmin), ztmin);
      double tmax = rmin(rmin(xtmax, ytmax), ztmax);
      if (tmin > tmax) return;
      size_t start = out.xs.size();
      for (const Shape& child : children) intersect(child, ray, out);
      stable_sort_range(out.xs, start, out.xs.size());  // Intersections::aggregate
      return;
    }
    case ShapeType::Union:
    case ShapeType::Intersection:
    case ShapeType::Difference: {  // 1339-1354
      Intersections all;
      intersect(children[0], ray, all);
      intersect(children[1], ray, all);
      all.sort();
      filter_intersections(*this, all, out);
      return;
    }
  }
}

// intersect.rs:300-328
static void filter_intersections(const Shape& op, const Intersections& all, Intersections& out) {
  bool inside_left = false, inside_right = false;
  for (const Intersection& i : all.xs) {
    bool which_hit = op.children[0].includes(*i.what);
    if (intersection_allowed(op.ty, which_hit, inside_left, inside_right)) out.xs.push_back(i);
    if (which_hit)
      inside_left = !inside_left;
    else
      inside_right = !inside_right;
  }
}

// shape.rs:805-829 with the per-type routines inlined
Tuple Shape::local_normal_at(Tuple at, const Intersection& hit) const {
  switch (ty) {
    case ShapeType::Empty:
    case ShapeType::Sphere: return {at.x, at.y, at.z, 0.0};  // 808, 880-888
    case ShapeType::Plane: return plane_normal;              // 913-919
    case ShapeType::Cube: {                                  // 959-978
      double xa = std::fabs(at.x), ya = std::fabs(at.y), za = std::fabs(at.z);
      double max_component = rmax(rmax(xa, ya), za);
      if (max_component == xa) return vector(at.x, 0.0, 0.0);
      if (max_component == ya) return vector(0.0, at.y, 0.0);
      return vector(0.0, 0.0, at.z);
    }
    case ShapeType::Cylinder: {  // 1047-1068
      double dist = at.x * at.x + at.z * at.z;
      if (dist < 1.0 && at.y >= maximum - FEQ_EPSILON) return vector(0.0, 1.0, 0.0);
      if (dist < 1.0 && at.y <= minimum + FEQ_EPSILON) return vector(0.0, -1.0, 0.0);
      return vector(at.x, 0.0, at.z);
    }
    case ShapeType::Cone: {  // 1147-1173
      double dist = at.x * at.x + at.z * at.z;
      if (dist < 1.0 && at.y >= maximum - FEQ_EPSILON) return vector(0.0, 1.0, 0.0);
      if (dist < 1.0 && at.y <= minimum + FEQ_EPSILON) return vector(0.0, -1.0, 0.0);
      double y = std::sqrt(dist);
      if (at.y > 0.0) y = -y;
      return vector(at.x, y, at.z);
    }
    case ShapeType::Triangle: return tri.normal;  // 1263-1270
    case ShapeType::SmoothTriangle:                // 1318-1332
      if (hit.has_uv) return n2 * hit.u + n3 * hit.v + n1 * (1.0 - hit.u - hit.v);
      return tri.normal;
    default: throw std::runtime_error("Local normal calculations should never occur on groups or CSG ops.");
  }
}

// shape.rs:1467-1470 -- the ray goes through the LOCAL inverse only
void intersect(const Shape& s, const Ray& r, Intersections& out) {
  Ray transformed = ray_transform(r, s.transform_inverse);
  s.local_intersect(transformed, out);
}

// shape.rs:1478-1484
Tuple normal_at(const Shape& s, Tuple world_point, const Intersection& hit) {
  Tuple local_point = s.world_to_object(world_point);
  Tuple local_normal = s.local_normal_at(local_point, hit);
  return s.normal_to_world(local_normal);
}

// ======================================================================
// world.rs
// ======================================================================
World World::empty_world() {  // 62-64
  World w;
  w.light = PointLight{};  // Default: black light at the zero tuple
  return w;
}

World World::default_world() {  // 31-53
  World w;
  w.light = point_light_new({1.0, 1.0, 1.0}, point(-10.0, 10.0, -10.0));
  Shape s1 = Shape::sphere();
  s1.material.color = {0.8, 1.0, 0.6};
  s1.material.diffuse = 0.7;
  s1.material.specular = 0.2;
  Shape s2 = Shape::sphere();
  s2.set_transform(mat_scaling(0.5, 0.5, 0.5));
  w.objects.push_back(std::move(s1));
  w.objects.push_back(std::move(s2));
  return w;
}

static void number_leaves(Shape& s, int* next) {
  if (s.ty == ShapeType::Group || s.is_csg()) {
    for (Shape& c : s.children) number_leaves(c, next);
    if (s.ty == ShapeType::Group) (void)s.bounds();
  } else {
    s.leaf_id = (*next)++;
  }
}

void World::finalize() {
  int next = 0;
  for (Shape& s : objects) number_leaves(s, &next);
  n_leaves = next;
}

// world.rs:107-117
void World::intersect(const Ray& r, Intersections& out) const {
  for (const Shape& obj : objects) orc::intersect(obj, r, out);
  out.sort();
}

// world.rs:123-137
bool World::is_shadowed(Tuple p, RayCounts* rc) const {
  Tuple v = light.position - p;
  double distance = magnitude(v);
  Tuple direction = normalize(v);
  Ray r = ray_new(p, direction);
  if (rc) rc->shadow++;
  Intersections xs;
  intersect(r, xs);
  Intersection h;
  if (xs.hit(&h)) return h.t < distance;
  return false;
}

// world.rs:142-159
Color World::reflected_color(const Computation& c, int remaining, RayCounts* rc) const {
  if (remaining == 0) return {0, 0, 0};
  if (feq(c.obj->material.reflective, 0.0)) return {0, 0, 0};
  Ray reflect_ray = ray_new(c.over_point, c.reflectv);
  if (rc) rc->reflect++;
  Color color = color_at(reflect_ray, remaining - 1, rc);
  return color * c.obj->material.reflective;
}

// world.rs:181-215
Color World::refracted_color(const Computation& c, int remaining, RayCounts* rc) const {
  if (remaining == 0) return {0, 0, 0};
  if (feq(c.obj->material.transparency, 0.0)) return {0, 0, 0};
  double n_ratio = c.n1 / c.n2;
  double cos_i = dot(c.eyev, c.normalv);
  double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
  if (sin2_t > 1.0) return {0, 0, 0};
  double cos_t = std::sqrt(1.0 - sin2_t);
  Tuple direction = c.normalv * (n_ratio * cos_i - cos_t) - c.eyev * n_ratio;
  Ray refracted_ray = ray_new(c.under_point, direction);
  if (rc) rc->refract++;
  return color_at(refracted_ray, remaining - 1, rc) * c.obj->material.transparency;
}

// world.rs:220-246
Color World::shade_hit(const Computation& c, int remaining, RayCounts* rc) const {
  Color surface = lighting(c.obj->material, *c.obj, light, c.over_point, c.eyev, c.normalv,
                           is_shadowed(c.over_point, rc));
  Color reflected = reflected_color(c, remaining, rc);
  Color refracted = refracted_color(c, remaining, rc);
  const Material& m = c.obj->material;
  if (m.reflective > 0.0 && m.transparency > 0.0) {
    double reflectance = schlick(c);
    return surface + reflected * reflectance + refracted * (1.0 - reflectance);
  }
  return surface + reflected + refracted;
}

// world.rs:249-261
Color World::color_at(const Ray& r, int remaining, RayCounts* rc, int* hit_leaf) const {
  Intersections xs;
  intersect(r, xs);
  Intersection h;
  if (!xs.hit(&h)) {
    if (hit_leaf) *hit_leaf = -1;
    return {0, 0, 0};
  }
  if (hit_leaf) *hit_leaf = h.what->leaf_id;
  Computation comps = computation_new(r, h, &xs);
  if (rc && rc->trace) {
    size_t slot = rc->trace->size();
    rc->trace->push_back(TraceRecord{{r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z},
                                     remaining, h.what->leaf_id, h.t, comps.n1, comps.n2, {0, 0, 0}});
    Color c = shade_hit(comps, remaining, rc);
    TraceRecord& rec = (*rc->trace)[slot];
    rec.color[0] = c.r; rec.color[1] = c.g; rec.color[2] = c.b;
    return c;
  }
  return shade_hit(comps, remaining, rc);
}

// ======================================================================
// camera.rs
// ======================================================================
Camera Camera::make(size_t hsize, size_t vsize, double fov, const Matrix& transform) {  // 43-69
  Camera c;
  double half_view = std::tan(fov / 2.0);
  double aspect = (double)hsize / (double)vsize;
  if (aspect >= 1.0) {
    c.half_width = half_view;
    c.half_height = half_view / aspect;
  } else {
    c.half_width = half_view * aspect;
    c.half_height = half_view;
  }
  c.pixel_size = c.half_width * 2.0 / (double)hsize;
  c.hsize = hsize;
  c.vsize = vsize;
  c.field_of_view = fov;
  c.transform = transform;
  return c;
}

Ray Camera::ray_for_pixel(size_t px, size_t py) const {  // 76-95
  double xoffset = ((double)px + 0.5) * pixel_size;
  double yoffset = ((double)py + 0.5) * pixel_size;
  double world_x = half_width - xoffset;
  double world_y = half_height - yoffset;
  Matrix tr_inv = mat_inverse_or_throw(transform, "Camera matrix should have an inverse.");
  Tuple pixel = tr_inv * point(world_x, world_y, -1.0);
  Tuple origin = tr_inv * point(0.0, 0.0, 0.0);
  Tuple direction = normalize(pixel - origin);
  return ray_new(origin, direction);
}

// ======================================================================
// canvas.rs
// ======================================================================
uint32_t quantise(double c) {  // 60-62: (c*255).clamp(0,255).ceil() as usize
  double v = c * 255.0;
  if (v < 0.0) v = 0.0; else if (v > 255.0) v = 255.0;  // NaN passes through
  v = std::ceil(v);
  if (!(v == v)) return 0;  // NaN -> 0 under Rust's saturating cast
  return (uint32_t)v;
}

std::string ppm_encode(size_t w, size_t h, const double* rgb) {  // 47-94
  std::string out;
  out.reserve(w * h * 12 + 32);
  out += "P3\n";
  out += std::to_string(w) + " " + std::to_string(h) + "\n";
  out += "255\n";
  size_t col = 1;
  for (size_t i = 0; i < w * h; i++) {
    std::string r = std::to_string(quantise(rgb[i * 3 + 0]));
    std::string g = std::to_string(quantise(rgb[i * 3 + 1]));
    std::string b = std::to_string(quantise(rgb[i * 3 + 2]));
    if (col + r.size() > 70) {
      out += "\n" + r + " " + g + " " + b;
      col = r.size() + g.size() + b.size() + 3;
    } else if (col + r.size() + g.size() > 70) {
      out += " " + r + "\n" + g + " " + b;
      col = g.size() + b.size() + 2;
    } else if (col + r.size() + g.size() + b.size() > 70) {
      out += " " + r + " " + g + "\n" + b;
      col = b.size() + 1;
    } else {
      if (col != 1) {
        out += " ";
        col += 1;
      }
      out += r + " " + g + " " + b;
      col += r.size() + g.size() + b.size() + 2;
    }
  }
  out += "\n";
  return out;
}

// ======================================================================
// parallel.rs -- per-pixel dynamic scheduling over `jobs` threads
// ======================================================================
void parallel_render(const World& w, const Camera& c, int jobs, size_t x0, size_t xstep, size_t y0,
                     size_t ystep, RenderOut* out) {
  if (jobs <= 0) throw std::runtime_error("assertion failed: size > 0");  // parallel.rs:94
  size_t W = c.hsize, H = c.vsize;
  out->rgb.assign(W * H * 3, 0.0);
  out->hit_leaf.assign(W * H, -1);
  size_t nx = x0 < W ? (W - x0 + xstep - 1) / xstep : 0;
  size_t ny = y0 < H ? (H - y0 + ystep - 1) / ystep : 0;
  size_t total = nx * ny;
  std::atomic<size_t> next{0};
  std::vector<RayCounts> counts(jobs);
  auto worker = [&](int tid) {
    RayCounts rc;
    for (;;) {
      size_t i = next.fetch_add(1, std::memory_order_relaxed);  // one message per pixel (194-198)
      if (i >= total) break;
      size_t x = x0 + (i % nx) * xstep, y = y0 + (i / nx) * ystep;
      Ray ray = c.ray_for_pixel(x, y);
      rc.primary++;
      int leaf = -1;
      Color col = w.color_at(ray, RECURSION_DEPTH, &rc, &leaf);
      size_t p = y * W + x;
      out->rgb[p * 3 + 0] = col.r;
      out->rgb[p * 3 + 1] = col.g;
      out->rgb[p * 3 + 2] = col.b;
      out->hit_leaf[p] = leaf;
    }
    counts[tid] = rc;
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < jobs; t++) threads.emplace_back(worker, t);
  worker(0);
  for (auto& t : threads) t.join();
  out->counts = RayCounts{};
  for (const RayCounts& rc : counts) {
    out->counts.primary += rc.primary;
    out->counts.shadow += rc.shadow;
    out->counts.reflect += rc.reflect;
    out->counts.refract += rc.refract;
  }
}

}  // namespace orc
