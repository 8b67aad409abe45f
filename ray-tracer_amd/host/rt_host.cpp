// Host side of the MI355X render path; see rt_host.hpp.  Citations are
// reference file:line (paths relative to the reference crate).
#include "rt_host.hpp"

#include <condition_variable>
#include <mutex>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <sstream>
#include <chrono>
#include <memory>
#include <thread>

namespace rthost {

static const double INF = std::numeric_limits<double>::infinity();
static inline bool feq(double a, double b) { return std::fabs(a - b) < 0.0001; }  // src/lib.rs:217-219

// ===========================================================================
// Matrix4D (src/matrix.rs)
// ===========================================================================
Mat4 Mat4::identity() {
  Mat4 r{};
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
  return r;
}

Mat4 Mat4::operator*(const Mat4& o) const {
  Mat4 r{};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r.m[i * 4 + j] = m[i * 4 + 0] * o.m[0 * 4 + j] + m[i * 4 + 1] * o.m[1 * 4 + j] +
                       m[i * 4 + 2] * o.m[2 * 4 + j] + m[i * 4 + 3] * o.m[3 * 4 + j];
  return r;
}

namespace {

// Determinant of the 3x3 matrix left after deleting one row and one column,
// expanded along its first row from 2x2 determinants (src/matrix.rs:122-124,
// 168-210), accumulating from 0.0 in column order.
double minor4(const double* a, int skip_row, int skip_col) {
  double s[9];
  int k = 0;
  for (int r = 0; r < 4; r++) {
    if (r == skip_row) continue;
    for (int c = 0; c < 4; c++) {
      if (c == skip_col) continue;
      s[k++] = a[r * 4 + c];
    }
  }
  const int others[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  double det = 0.0;
  for (int c = 0; c < 3; c++) {
    int c0 = others[c][0], c1 = others[c][1];
    double minor2 = s[3 + c0] * s[6 + c1] - s[3 + c1] * s[6 + c0];
    double cof = minor2 * ((c % 2 == 0) ? 1.0 : -1.0);
    det += s[c] * cof;
  }
  return det;
}

double cofactor4(const double* a, int r, int c) {  // src/matrix.rs:531-534
  return minor4(a, r, c) * (((r + c) % 2 == 0) ? 1.0 : -1.0);
}

}  // namespace

bool Mat4::inverse(Mat4* out) const {
  double det = 0.0;
  for (int c = 0; c < 4; c++) det += m[c] * cofactor4(m, 0, c);  // src/matrix.rs:537-544
  if (det == 0.0) return false;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) out->m[c * 4 + r] = cofactor4(m, r, c) / det;  // src/matrix.rs:556-561
  return true;
}

Mat4 Mat4::inverse_or_panic(const char* msg) const {
  Mat4 r;
  if (!inverse(&r)) throw Panic(msg);
  return r;
}

namespace {
struct Vec4 {
  double x, y, z, w;
};
Vec4 sub(Vec4 a, Vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
Vec4 normalize4(Vec4 a) {  // src/tuple.rs:111-135: w takes part; multiply by the reciprocal
  double mag = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  return {a.x * (1.0 / mag), a.y * (1.0 / mag), a.z * (1.0 / mag), a.w * (1.0 / mag)};
}
Vec4 cross3(Vec4 a, Vec4 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0};
}
Vec4 apply(const Mat4& t, Vec4 p) {  // src/matrix.rs:638-653
  double b[4];
  for (int r = 0; r < 4; r++) b[r] = t.m[r * 4] * p.x + t.m[r * 4 + 1] * p.y + t.m[r * 4 + 2] * p.z + t.m[r * 4 + 3] * p.w;
  return {b[0], b[1], b[2], b[3]};
}
}  // namespace

Mat4 view_transform(const double from[3], const double to[3], const double up[4]) {
  Vec4 f{from[0], from[1], from[2], 1.0}, t{to[0], to[1], to[2], 1.0}, u{up[0], up[1], up[2], up[3]};
  Vec4 forward = normalize4(sub(t, f));
  Vec4 left = cross3(forward, normalize4(u));
  Vec4 true_up = cross3(left, forward);
  Mat4 o = Mat4::identity();
  o.m[0] = left.x; o.m[1] = left.y; o.m[2] = left.z;
  o.m[4] = true_up.x; o.m[5] = true_up.y; o.m[6] = true_up.z;
  o.m[8] = -forward.x; o.m[9] = -forward.y; o.m[10] = -forward.z;
  Mat4 tr = Mat4::identity();
  tr.m[3] = -f.x; tr.m[7] = -f.y; tr.m[11] = -f.z;
  return o * tr;
}

// ===========================================================================
// Camera (src/camera.rs)
// ===========================================================================
Camera Camera::make(size_t hsize, size_t vsize, double fov, const Mat4& transform) {
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

rt_camera Camera::describe() const {
  rt_camera c{};
  c.hsize = (uint32_t)hsize;
  c.vsize = (uint32_t)vsize;
  c.half_width = half_width;
  c.half_height = half_height;
  c.pixel_size = pixel_size;
  Mat4 inv = transform.inverse_or_panic("Camera matrix should have an inverse.");
  memcpy(c.inv, inv.m, sizeof(c.inv));
  return c;
}

// ===========================================================================
// Scene JSON (src/scene.rs) -- a small JSON reader, then the conversions
// ===========================================================================
namespace {

struct Json {
  enum T { Nul, Boolean, Number, String, Array, Object } t = Nul;
  double num = 0;
  bool integral = false;
  bool flag = false;
  std::string str;
  std::vector<Json> items;
  std::vector<std::string> keys;  // parallel to items for objects

  const Json* field(const std::string& k) const {
    for (size_t i = keys.size(); i-- > 0;)
      if (keys[i] == k) return &items[i];
    return nullptr;
  }
};

class JsonReader {
 public:
  explicit JsonReader(const std::string& s) : s_(s) {}
  Json parse() {
    Json v = value();
    skip();
    if (i_ != s_.size()) err("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
  [[noreturn]] void err(const std::string& what) const {
    throw Panic("Failed to parse scene JSON: " + what + " at byte " + std::to_string(i_));
  }
  void skip() {
    while (i_ < s_.size() && strchr(" \t\r\n", s_[i_])) i_++;
  }
  bool eat(char c) {
    skip();
    if (i_ < s_.size() && s_[i_] == c) {
      i_++;
      return true;
    }
    return false;
  }
  bool literal(const char* w) {
    size_t n = strlen(w);
    if (s_.compare(i_, n, w) == 0) {
      i_ += n;
      return true;
    }
    return false;
  }
  std::string text() {
    std::string out;
    i_++;
    for (;;) {
      if (i_ >= s_.size()) err("unterminated string");
      char c = s_[i_++];
      if (c == '"') break;
      if (c != '\\') {
        out.push_back(c);
        continue;
      }
      if (i_ >= s_.size()) err("bad escape");
      char e = s_[i_++];
      switch (e) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          if (i_ + 4 > s_.size()) err("bad unicode escape");
          unsigned cp = (unsigned)std::stoul(s_.substr(i_, 4), nullptr, 16);
          i_ += 4;
          if (cp < 0x80) out.push_back((char)cp);
          else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 63)));
          } else {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 63)));
            out.push_back((char)(0x80 | (cp & 63)));
          }
          break;
        }
        default: out.push_back(e);  // \" \\ \/
      }
    }
    return out;
  }
  Json value() {
    skip();
    if (i_ >= s_.size()) err("unexpected end of input");
    Json v;
    char c = s_[i_];
    if (c == '{') {
      i_++;
      v.t = Json::Object;
      if (eat('}')) return v;
      do {
        skip();
        if (i_ >= s_.size() || s_[i_] != '"') err("expected an object key");
        v.keys.push_back(text());
        if (!eat(':')) err("expected ':'");
        v.items.push_back(value());
      } while (eat(','));
      if (!eat('}')) err("expected '}'");
    } else if (c == '[') {
      i_++;
      v.t = Json::Array;
      if (eat(']')) return v;
      do v.items.push_back(value());
      while (eat(','));
      if (!eat(']')) err("expected ']'");
    } else if (c == '"') {
      v.t = Json::String;
      v.str = text();
    } else if (literal("true")) {
      v.t = Json::Boolean;
      v.flag = true;
    } else if (literal("false")) {
      v.t = Json::Boolean;
    } else if (literal("null")) {
      v.t = Json::Nul;
    } else {
      size_t b = i_;
      while (i_ < s_.size() && (isdigit((unsigned char)s_[i_]) || strchr("+-.eE", s_[i_]))) i_++;
      if (b == i_) err("unexpected character");
      std::string lit = s_.substr(b, i_ - b);
      char* endp = nullptr;
      v.t = Json::Number;
      v.num = strtod(lit.c_str(), &endp);  // correctly rounded, as serde_json is on short decimals
      if (*endp) err("malformed number");
      v.integral = lit.find_first_of(".eE") == std::string::npos;
    }
    return v;
  }
};

[[noreturn]] void jerr(const std::string& what) { throw Panic("Failed to parse scene JSON: " + what); }
const Json& required(const Json& o, const char* k) {
  if (o.t != Json::Object) jerr("expected an object");
  const Json* f = o.field(k);
  if (!f) jerr(std::string("missing field `") + k + "`");
  return *f;
}
const Json* optional(const Json& o, const char* k) {
  const Json* f = o.field(k);
  return f && f->t != Json::Nul ? f : nullptr;
}
double number(const Json& j) {
  if (j.t != Json::Number) jerr("expected a number");
  return j.num;
}
size_t unsigned_int(const Json& j) {
  if (j.t != Json::Number || !j.integral || j.num < 0) jerr("expected an unsigned integer");
  return (size_t)j.num;
}
std::vector<double> numbers(const Json& j) {
  if (j.t != Json::Array) jerr("expected an array of numbers");
  std::vector<double> v;
  for (const Json& e : j.items) v.push_back(number(e));
  return v;
}
// `From<&Vec<f64>>` for Tuple4D / Color: missing trailing entries are zero (src/tuple.rs:62-72)
void fill(double* dst, int n, const std::vector<double>& v) {
  for (int i = 0; i < n; i++) dst[i] = i < (int)v.size() ? v[i] : 0.0;
}
Mat4 matrix_of(const std::vector<double>& v) {  // src/scene.rs:190-195
  if (v.size() > 16) throw Panic("index out of bounds: the len is 16 but a transform lists more elements");
  Mat4 m{};
  for (size_t i = 0; i < v.size(); i++) m.m[i] = v[i];
  return m;
}

Pattern pattern_of(const Json& j) {  // src/scene.rs:275-366
  const Json& ty = required(j, "ty");
  if (ty.t != Json::String) jerr("pattern ty must be a string");
  Pattern p;
  static const std::pair<const char*, int> names[] = {
      {"null", RT_PAT_NULL},     {"point", RT_PAT_POINT},     {"identity", RT_PAT_IDENTITY}, {"stripe", RT_PAT_STRIPE},
      {"ring", RT_PAT_RING},     {"checker", RT_PAT_CHECKER}, {"gradient", RT_PAT_GRADIENT}};
  bool known = false;
  for (auto& n : names)
    if (ty.str == n.first) {
      p.type = n.second;
      known = true;
    }
  const Json* pri = optional(j, "primary_color");
  const Json* sec = optional(j, "secondary_color");
  if (pri) fill(p.a, 3, numbers(*pri));
  if (sec) fill(p.b, 3, numbers(*sec));
  if (!known) throw Panic("Unrecognized pattern in scene description JSON.");
  if (const Json* t = optional(j, "transform")) p.transform = matrix_of(numbers(*t));
  return p;
}

Material material_of(const Json& j) {  // src/scene.rs:224-265
  Material m;
  fill(m.color, 3, numbers(required(j, "color")));
  if (const Json* p = optional(j, "pattern")) m.pattern = pattern_of(*p);
  struct { const char* key; double* dst; } scalars[] = {
      {"ambient", &m.ambient}, {"diffuse", &m.diffuse}, {"specular", &m.specular}, {"shininess", &m.shininess},
      {"reflective", &m.reflective}, {"refractive_index", &m.refractive_index}, {"transparency", &m.transparency}};
  for (auto& s : scalars)
    if (const Json* v = optional(j, s.key)) *s.dst = number(*v);
  return m;
}

Shape leaf_shape(int type, double mn = 0, double mx = 0, int closed = 0) {
  Shape s;
  s.type = type;
  s.minimum = mn;
  s.maximum = mx;
  s.closed = closed;
  return s;
}

std::string resolve(const std::string& base, const std::string& rel) {
  if (base.empty() || (!rel.empty() && rel[0] == '/')) return rel;
  return base + "/" + rel;
}

Shape shape_of(const Json& j, const std::string& base_dir) {  // src/scene.rs:95-207
  const Json& tyj = required(j, "ty");
  if (tyj.t != Json::String) jerr("shape ty must be a string");
  const std::string& ty = tyj.str;
  const Json* kids = optional(j, "children");
  if (kids && kids->t != Json::Array) jerr("children must be an array");
  Shape s;
  if (ty == "empty") s = leaf_shape(RT_EMPTY);
  else if (ty == "sphere") s = leaf_shape(RT_SPHERE);
  else if (ty == "plane") s = leaf_shape(RT_PLANE);
  else if (ty == "cube") s = leaf_shape(RT_CUBE);
  else if (ty == "cylinder") s = leaf_shape(RT_CYLINDER, -1.0 * INF, INF, 0);   // src/shape.rs:245-254
  else if (ty == "bounded_cylinder") s = leaf_shape(RT_CYLINDER, -1.0, 1.0, 0);
  else if (ty == "capped_cylinder") s = leaf_shape(RT_CYLINDER, -1.0, 1.0, 1);
  else if (ty == "bounded_cone") s = leaf_shape(RT_CONE, 0.0, 1.0, 0);
  else if (ty == "bounded_dn_cone") s = leaf_shape(RT_CONE, -1.0, 1.0, 0);
  else if (ty == "capped_cone") s = leaf_shape(RT_CONE, 0.0, 1.0, 1);
  else if (ty == "capped_dn_cone") s = leaf_shape(RT_CONE, -1.0, 1.0, 1);
  else if (ty == "group") {
    s.kind = Shape::Group;
    if (kids)
      for (const Json& k : kids->items) s.children.push_back(shape_of(k, base_dir));
  } else if (ty == "union" || ty == "intersection" || ty == "difference") {
    if (kids) {
      if (kids->items.size() < 2) throw Panic("CSG union must have at least two operands.");
      s.kind = Shape::Csg;
      s.type = ty == "union" ? RT_CSG_UNION : (ty == "intersection" ? RT_CSG_INTERSECTION : RT_CSG_DIFFERENCE);
      s.children.push_back(shape_of(kids->items[0], base_dir));
      s.children.push_back(shape_of(kids->items[1], base_dir));
    } else {
      s = leaf_shape(RT_EMPTY);  // src/scene.rs:124,138,152: no children leaves Shape::empty()
    }
  } else if (ty == "model") {
    const Json* path = optional(j, "path");
    if (!path) throw Panic("Model requires a path in scene description JSON.");
    if (path->t != Json::String) jerr("path must be a string");
    std::string full = resolve(base_dir, path->str);
    std::ifstream f(full, std::ios::binary);
    if (!f) throw Panic("Failed to open file \"" + full + "\".");
    std::stringstream ss;
    ss << f.rdbuf();
    std::map<std::string, Shape> groups = obj_groups_from_text(ss.str(), nullptr);
    s.kind = Shape::Group;
    for (auto& kv : groups) s.children.push_back(std::move(kv.second));  // BTreeMap order (src/scene.rs:175)
  } else {
    throw Panic("Unrecognized shape type in scene description JSON.");
  }
  if (const Json* t = optional(j, "transform")) s.transform = matrix_of(numbers(*t));
  else s.transform = Mat4::identity();
  // set_transform inverts at load time (src/shape.rs:597-599)
  (void)s.transform.inverse_or_panic("Shape transform should have an inverse.");
  if (const Json* m = optional(j, "material")) s.material = material_of(*m);
  return s;
}

// ---- OBJ (src/obj.rs)
std::vector<std::string> split_on(const std::string& s, char sep) {
  std::vector<std::string> parts;
  size_t b = 0;
  for (size_t i = 0; i <= s.size(); i++)
    if (i == s.size() || s[i] == sep) {
      parts.push_back(s.substr(b, i - b));
      b = i + 1;
    }
  return parts;
}
bool to_f64(const std::string& s, double* v) {
  if (s.empty() || isspace((unsigned char)s[0]) || s.find_first_of("xX") != std::string::npos) return false;
  char* e = nullptr;
  *v = strtod(s.c_str(), &e);
  return e != s.c_str() && !*e;
}
bool to_index(const std::string& s, size_t* v) {
  size_t i = (!s.empty() && s[0] == '+') ? 1 : 0;
  if (i >= s.size()) return false;
  size_t acc = 0;
  for (; i < s.size(); i++) {
    if (!isdigit((unsigned char)s[i])) return false;
    acc = acc * 10 + (size_t)(s[i] - '0');
  }
  *v = acc;
  return true;
}

struct Corner {
  size_t v = 0, n = 0;
  bool has_n = false;
};

rt_triangle triangle_record(const double* p1, const double* p2, const double* p3) {  // src/geometry.rs:28-33
  rt_triangle t{};
  for (int a = 0; a < 3; a++) {
    t.p1[a] = p1[a];
    t.e1[a] = p2[a] - p1[a];
    t.e2[a] = p3[a] - p1[a];
  }
  Vec4 e1{t.e1[0], t.e1[1], t.e1[2], 0.0}, e2{t.e2[0], t.e2[1], t.e2[2], 0.0};
  Vec4 n = normalize4(cross3(e2, e1));
  t.normal[0] = n.x; t.normal[1] = n.y; t.normal[2] = n.z;
  return t;
}

}  // namespace

std::map<std::string, Shape> obj_groups_from_text(const std::string& text, size_t* ignored_out) {
  std::map<std::string, Shape> groups;  // std::map iterates keys in byte order like BTreeMap<String, _>
  auto new_group = [] {
    Shape g;
    g.kind = Shape::Group;
    return g;
  };
  groups.emplace("", new_group());  // src/obj.rs:27-28
  std::vector<std::array<double, 3>> vertices, normals;
  std::string current;
  size_t ignored = 0;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t nl = text.find('\n', pos);
    std::string line = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
    pos = nl == std::string::npos ? text.size() : nl + 1;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    std::vector<std::string> params = split_on(line, ' ');  // single spaces only (src/obj.rs:96)
    auto coord = [&](const std::string& s) {
      double v;
      if (!to_f64(s, &v)) throw Panic("called `Result::unwrap()` on an `Err` value: ParseFloatError (OBJ line: " + line + ")");
      return v;
    };
    if (params[0] == "v" || params[0] == "vn") {
      if (params.size() < 4) {
        ignored++;
        continue;
      }
      std::array<double, 3> p = {coord(params[1]), coord(params[2]), coord(params[3])};
      (params[0] == "v" ? vertices : normals).push_back(p);
    } else if (params[0] == "f") {
      std::vector<Corner> face;
      bool skip_line = false;
      for (size_t k = 1; k < params.size() && !skip_line; k++) {
        Corner c;
        if (to_index(params[k], &c.v)) {
          face.push_back(c);
          continue;
        }
        std::vector<std::string> attrs = split_on(params[k], '/');
        if (attrs.size() > 3) {
          skip_line = true;
          break;
        }
        if (!to_index(attrs[0], &c.v)) throw Panic("called `Result::unwrap()` on an `Err` value: ParseIntError (OBJ line: " + line + ")");
        if (attrs.size() < 3) throw Panic("index out of bounds: OBJ face vertex must be `v`, `v/t/n` or `v//n`");
        size_t unused;
        (void)to_index(attrs[1], &unused);
        c.has_n = to_index(attrs[2], &c.n);
        face.push_back(c);
      }
      if (!skip_line)
        for (const Corner& c : face)
          if (c.v > vertices.size() || (c.has_n ? c.n : 0) > normals.size()) skip_line = true;  // src/obj.rs:157-161
      if (skip_line) {
        ignored++;
        continue;
      }
      if (face.empty()) throw Panic("attempt to subtract with overflow (OBJ face without vertices)");
      Shape& group = groups.at(current);
      for (size_t i = 1; i + 1 < face.size(); i++) {  // fan triangulation (src/obj.rs:236-276)
        const Corner* c[3] = {&face[0], &face[i], &face[i + 1]};
        for (auto* cc : c)
          if (cc->v == 0) throw Panic("index out of bounds: OBJ vertex index 0");
        Shape t;
        t.tri = triangle_record(vertices[c[0]->v - 1].data(), vertices[c[1]->v - 1].data(), vertices[c[2]->v - 1].data());
        for (int a = 0; a < 3; a++) {
          t.p2[a] = vertices[c[1]->v - 1][a];
          t.p3[a] = vertices[c[2]->v - 1][a];
        }
        if (c[0]->has_n && c[1]->has_n && c[2]->has_n) {
          for (auto* cc : c)
            if (cc->n == 0) throw Panic("index out of bounds: OBJ normal index 0");
          t.type = RT_SMOOTH_TRIANGLE;
          for (int a = 0; a < 3; a++) {
            t.tri.n1[a] = normals[c[0]->n - 1][a];
            t.tri.n2[a] = normals[c[1]->n - 1][a];
            t.tri.n3[a] = normals[c[2]->n - 1][a];
          }
        } else {
          t.type = RT_TRIANGLE;
        }
        group.children.push_back(std::move(t));
      }
    } else if (line[0] == 'g') {
      if (params.size() < 2) {
        ignored++;
        continue;
      }
      if (!groups.count(params[1])) groups.emplace(params[1], new_group());
      current = params[1];
    }
    ignored++;  // every handled line falls through to the counter (src/obj.rs:192-193)
  }
  if (ignored_out) *ignored_out = ignored;
  return groups;
}

Scene scene_from_json(const std::string& text, const std::string& base_dir) {  // src/scene.rs:25-50
  Json root = JsonReader(text).parse();
  size_t width = unsigned_int(required(root, "canvas_width"));
  size_t height = unsigned_int(required(root, "canvas_height"));
  double fov = number(required(root, "field_of_view"));
  double from[4], to[4], up[4];
  fill(from, 4, numbers(required(root, "camera_from")));
  fill(to, 4, numbers(required(root, "camera_to")));
  fill(up, 4, numbers(required(root, "camera_up")));
  const Json& light = required(root, "light");
  Scene sc;
  fill(sc.world.light_source.intensity, 3, numbers(required(light, "intensity")));
  fill(sc.world.light_source.position, 3, numbers(required(light, "position")));  // w forced to 1 (src/scene.rs:81)
  const Json& shapes = required(root, "shapes");
  if (shapes.t != Json::Array) jerr("shapes must be an array");
  sc.camera = Camera::make(width, height, fov, view_transform(from, to, up));
  for (const Json& s : shapes.items) sc.world.objects.push_back(shape_of(s, base_dir));
  return sc;
}

// ===========================================================================
// Flattening: Shape tree -> rt_scene_desc arrays
// ===========================================================================
namespace {

struct Box {
  double lo[3], hi[3];
};

// Bounds::transform (src/geometry.rs:118-150): eight corners through the full
// 4x4 product, then NaN-ignoring min / max (f64::min / f64::max == fmin / fmax).
Box transform_box(const Box& b, const Mat4& t) {
  Box r;
  for (int a = 0; a < 3; a++) {
    r.lo[a] = INF;
    r.hi[a] = -1.0 * INF;
  }
  const int order[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
  for (auto& o : order) {
    Vec4 c = apply(t, Vec4{o[0] ? b.hi[0] : b.lo[0], o[1] ? b.hi[1] : b.lo[1], o[2] ? b.hi[2] : b.lo[2], 1.0});
    double v[3] = {c.x, c.y, c.z};
    for (int a = 0; a < 3; a++) {
      r.lo[a] = std::fmin(r.lo[a], v[a]);
      r.hi[a] = std::fmax(r.hi[a], v[a]);
    }
  }
  return r;
}

// Shape::bounds (src/shape.rs:470-544), including its quirks: a child's box is
// moved by the GROUP's transform, cones and cylinders use (min,min,min)-(max,max,max).
Box reference_bounds(const Shape& s) {
  Box b;
  auto set = [&](double l0, double l1, double l2, double h0, double h1, double h2) {
    b.lo[0] = l0; b.lo[1] = l1; b.lo[2] = l2;
    b.hi[0] = h0; b.hi[1] = h1; b.hi[2] = h2;
  };
  if (s.kind == Shape::Group) {
    set(INF, INF, INF, -1.0 * INF, -1.0 * INF, -1.0 * INF);
    for (const Shape& c : s.children) {
      Box cb = transform_box(reference_bounds(c), s.transform);
      for (int a = 0; a < 3; a++) {
        b.lo[a] = std::fmin(b.lo[a], cb.lo[a]);
        b.hi[a] = std::fmax(b.hi[a], cb.hi[a]);
      }
    }
    return b;
  }
  if (s.kind == Shape::Csg) {
    set(-1.0 * INF, -1.0 * INF, -1.0 * INF, INF, INF, INF);
    return b;
  }
  switch (s.type) {
    case RT_EMPTY: set(0, 0, 0, 0, 0, 0); break;
    case RT_SPHERE:
    case RT_CUBE: set(-1.0, -1.0, -1.0, 1.0, 1.0, 1.0); break;
    case RT_CONE:
    case RT_CYLINDER: set(s.minimum, s.minimum, s.minimum, s.maximum, s.maximum, s.maximum); break;
    case RT_TRIANGLE:
    case RT_SMOOTH_TRIANGLE:
      for (int a = 0; a < 3; a++) {  // the stored corner points themselves (src/shape.rs:485-494)
        double p1 = s.tri.p1[a], p2 = s.p2[a], p3 = s.p3[a];
        b.lo[a] = std::fmin(p1, std::fmin(p2, p3));
        b.hi[a] = std::fmax(p1, std::fmax(p2, p3));
      }
      break;
    default: set(-1.0 * INF, -1.0 * INF, -1.0 * INF, INF, INF, INF);  // plane
  }
  return b;
}

struct Flattener {
  FlatWorld out;

  int add_pattern(const Pattern& p) {
    rt_pattern r{};
    r.type = p.type;
    r.first = r.second = -1;
    if (p.type == RT_PAT_BLEND) {
      if (!p.first || !p.second) throw Panic("blend pattern without operands");
      r.first = add_pattern(*p.first);
      r.second = add_pattern(*p.second);
    }
    Mat4 inv = p.transform.inverse_or_panic("Pattern transform should be invertible.");  // src/pattern.rs:199
    memcpy(r.inv, inv.m, sizeof(r.inv));
    memcpy(r.a, p.a, sizeof(r.a));
    memcpy(r.b, p.b, sizeof(r.b));
    out.patterns.push_back(r);
    return (int)out.patterns.size() - 1;
  }

  int add_material(const Material& m) {
    rt_material r{};
    memcpy(r.color, m.color, sizeof(r.color));
    r.ambient = m.ambient; r.diffuse = m.diffuse; r.specular = m.specular; r.shininess = m.shininess;
    r.reflective = m.reflective; r.refractive_index = m.refractive_index; r.transparency = m.transparency;
    r.pattern = m.pattern ? add_pattern(*m.pattern) : -1;
    out.materials.push_back(r);
    return (int)out.materials.size() - 1;
  }

  // ray_acc: product of the local inverses of the enclosing groups, outermost last applied first;
  // parent: product of the enclosing groups' transforms (src/shape.rs:681-736), when has_parent.
  struct Share {  // records reused by the plain triangles of one group (OBJ meshes)
    int material = -1, xform = -1;
  };

  // Left operands of the enclosing CSG operators, innermost last (for Shape::includes).
  std::vector<const Shape*> left_operands;

  // PartialEq of shapes as `includes` uses it (src/shape.rs:69-102, 203-209, 450-467): kind and
  // parameters, the local transform and the material, compared with the reference's tolerances.
  static bool near(double a, double b) { return std::fabs(a - b) < 0.0001; }
  static bool pattern_eq(const Pattern& a, const Pattern& b) {
    if (a.type != b.type) return false;
    for (int i = 0; i < 16; i++)
      if (!near(a.transform.m[i], b.transform.m[i])) return false;
    auto col = [](const double* x, const double* y) { return near(x[0], y[0]) && near(x[1], y[1]) && near(x[2], y[2]); };
    switch (a.type) {
      case RT_PAT_NULL:
      case RT_PAT_POINT: return true;
      case RT_PAT_IDENTITY: return col(a.a, b.a);
      case RT_PAT_BLEND: return a.first && b.first && a.second && b.second && pattern_eq(*a.first, *b.first) && pattern_eq(*a.second, *b.second);
      default: return col(a.a, b.a) && col(a.b, b.b);
    }
  }
  static bool leaf_eq(const Shape& a, const Shape& b) {
    if (a.kind != Shape::Leaf || b.kind != Shape::Leaf || a.type != b.type) return false;
    auto v3 = [](const double* x, const double* y) { return near(x[0], y[0]) && near(x[1], y[1]) && near(x[2], y[2]); };
    switch (a.type) {
      case RT_CONE:
      case RT_CYLINDER:
        if (!(a.minimum == b.minimum && a.maximum == b.maximum && a.closed == b.closed)) return false;
        break;
      case RT_PLANE:
        if (!v3(a.plane_normal, b.plane_normal)) return false;
        break;
      case RT_TRIANGLE:
      case RT_SMOOTH_TRIANGLE:
        if (!(v3(a.tri.p1, b.tri.p1) && v3(a.p2, b.p2) && v3(a.p3, b.p3) && v3(a.tri.e1, b.tri.e1) && v3(a.tri.e2, b.tri.e2) &&
              v3(a.tri.normal, b.tri.normal)))
          return false;
        if (a.type == RT_SMOOTH_TRIANGLE && !(v3(a.tri.n1, b.tri.n1) && v3(a.tri.n2, b.tri.n2) && v3(a.tri.n3, b.tri.n3))) return false;
        break;
      default: break;
    }
    for (int i = 0; i < 16; i++)
      if (!near(a.transform.m[i], b.transform.m[i])) return false;
    const Material &m = a.material, &n = b.material;
    if (!(near(m.color[0], n.color[0]) && near(m.color[1], n.color[1]) && near(m.color[2], n.color[2]))) return false;
    if (m.pattern.has_value() != n.pattern.has_value()) return false;
    if (m.pattern && !pattern_eq(*m.pattern, *n.pattern)) return false;
    return m.ambient == n.ambient && m.diffuse == n.diffuse && m.specular == n.specular && m.shininess == n.shininess &&
           m.reflective == n.reflective && m.refractive_index == n.refractive_index && m.transparency == n.transparency;
  }
  static bool includes(const Shape& tree, const Shape& leaf) {  // src/shape.rs:450-467
    if (tree.kind == Shape::Leaf) return leaf_eq(tree, leaf);
    for (const Shape& c : tree.children)
      if (includes(c, leaf)) return true;
    return false;
  }

  // The non-identity local inverses from the top level down to the shape being walked, outermost
  // first: the matrices shape::intersect applies to a ray one after the other (src/shape.rs:1467-1470).
  // Multiplying by the identity changes nothing, so with fewer than two of them the product ray_inv
  // applied once gives the same ray bit for bit; with two or more the device has to follow the chain.
  std::vector<Mat4> chain;
  rt_chain_ref emit_chain() {
    rt_chain_ref r{0, 0};
    if (chain.size() < 2) return r;
    r.first = (int32_t)(out.chain_mats.size() / 16);
    r.count = (int32_t)chain.size();
    for (const Mat4& m : chain) out.chain_mats.insert(out.chain_mats.end(), m.m, m.m + 16);
    return r;
  }
  struct ChainLevel {  // pushes a level's local inverse for the lifetime of the walk below it
    std::vector<Mat4>& chain;
    bool pushed;
    static bool same_as_identity(const Mat4& m) {  // by value: the cofactor inverse of the identity holds -0.0
      for (int i = 0; i < 16; i++)
        if (m.m[i] != (i % 5 == 0 ? 1.0 : 0.0)) return false;
      return true;
    }
    ChainLevel(std::vector<Mat4>& c, const Mat4& local_inv) : chain(c), pushed(!same_as_identity(local_inv)) {
      if (pushed) chain.push_back(local_inv);
    }
    ~ChainLevel() {
      if (pushed) chain.pop_back();
    }
  };

  void walk(const Shape& s, const Mat4& ray_acc, bool has_parent, const Mat4& parent, int gate, Share* share, int csg = -1) {
    Mat4 local_inv = s.transform.inverse_or_panic("Shape transform should have an inverse.");
    Mat4 ray_inv = has_parent ? local_inv * ray_acc : local_inv;
    ChainLevel level(chain, local_inv);
    if (s.kind == Shape::Csg) {
      // a transform level like a group, without the bounding-box gate (src/shape.rs:1339-1354)
      if (left_operands.size() >= 32) throw std::invalid_argument("CSG operators nested deeper than 32 levels");
      rt_csg c{};
      c.op = s.type;
      c.parent = csg;
      c.gate = gate;
      c.first_leaf = (int32_t)out.leaves.size();
      out.csg.push_back(c);
      int ci = (int)out.csg.size() - 1;
      Mat4 child_parent = has_parent ? parent * s.transform : s.transform;
      left_operands.push_back(&s.children[0]);
      for (const Shape& ch : s.children) walk(ch, ray_inv, true, child_parent, gate, nullptr, ci);
      left_operands.pop_back();
      out.csg[ci].n_leaves = (int32_t)out.leaves.size() - out.csg[ci].first_leaf;
      return;
    }
    if (s.kind == Shape::Group) {
      if (s.children.empty()) return;  // src/shape.rs:1186-1188
      rt_gate g{};
      g.parent = gate;
      g.csg = csg;
      memcpy(g.to_local, ray_inv.m, sizeof(g.to_local));
      Box b = reference_bounds(s);
      memcpy(g.bmin, b.lo, sizeof(g.bmin));
      memcpy(g.bmax, b.hi, sizeof(g.bmax));
      out.gates.push_back(g);
      out.gate_chain.push_back(emit_chain());
      int gi = (int)out.gates.size() - 1;
      Mat4 child_parent = has_parent ? parent * s.transform : s.transform;
      Share group_share;
      for (const Shape& c : s.children) walk(c, ray_inv, true, child_parent, gi, &group_share, csg);
      return;
    }
    rt_leaf L{};
    L.type = s.type;
    L.closed = s.closed;
    L.gate = gate;
    L.minimum = s.minimum;
    L.maximum = s.maximum;
    memcpy(L.plane_normal, s.plane_normal, sizeof(L.plane_normal));
    L.tri = -1;
    L.csg = csg;
    L.csg_left = 0;
    for (size_t k = 0; k < left_operands.size(); k++)  // bit 0 = innermost operator
      if (includes(*left_operands[left_operands.size() - 1 - k], s)) L.csg_left |= 1u << k;
    bool is_tri = s.type == RT_TRIANGLE || s.type == RT_SMOOTH_TRIANGLE;
    bool plain_tri = is_tri && is_default_material(s.material) && is_identity(s.transform);
    if (plain_tri && share && share->material >= 0) {
      L.material = share->material;
      L.xform = share->xform;
    } else {
      L.material = add_material(s.material);
      rt_xform X{};
      Mat4 fwd = has_parent ? parent * s.transform : s.transform;
      // full_transform_inverse = (parent * transform)^-1 (src/shape.rs:730-735); a top-level shape
      // falls back to its own transform_inverse (src/shape.rs:758-761)
      Mat4 full = has_parent ? fwd.inverse_or_panic("Shape parent transform and transform should be invertible.") : local_inv;
      memcpy(X.ray_inv, ray_inv.m, sizeof(X.ray_inv));
      memcpy(X.full_inv, full.m, sizeof(X.full_inv));
      memcpy(X.fwd, fwd.m, sizeof(X.fwd));
      out.xforms.push_back(X);
      out.xform_chain.push_back(emit_chain());
      L.xform = (int)out.xforms.size() - 1;
      if (plain_tri && share) {
        share->material = L.material;
        share->xform = L.xform;
      }
    }
    if (is_tri) {
      out.triangles.push_back(s.tri);
      L.tri = (int)out.triangles.size() - 1;
    }
    out.leaves.push_back(L);
  }

  static bool is_identity(const Mat4& m) {
    Mat4 i = Mat4::identity();
    return memcmp(m.m, i.m, sizeof(m.m)) == 0;
  }
  static bool is_default_material(const Material& m) {
    Material d;
    return !m.pattern && m.color[0] == d.color[0] && m.color[1] == d.color[1] && m.color[2] == d.color[2] &&
           m.ambient == d.ambient && m.diffuse == d.diffuse && m.specular == d.specular && m.shininess == d.shininess &&
           m.reflective == d.reflective && m.refractive_index == d.refractive_index && m.transparency == d.transparency;
  }
};

}  // namespace

rt_scene_desc FlatWorld::desc() const {
  rt_scene_desc d{};
  d.light = light;
  d.n_leaves = (int32_t)leaves.size();
  d.n_gates = (int32_t)gates.size();
  d.n_xforms = (int32_t)xforms.size();
  d.n_triangles = (int32_t)triangles.size();
  d.n_materials = (int32_t)materials.size();
  d.n_patterns = (int32_t)patterns.size();
  d.n_csg = (int32_t)csg.size();
  d.n_chain_mats = (int32_t)(chain_mats.size() / 16);
  d.chain_mats = chain_mats.data();
  d.xform_chain = xform_chain.data();
  d.gate_chain = gate_chain.data();
  d.csg = csg.data();
  d.leaves = leaves.data();
  d.gates = gates.data();
  d.xforms = xforms.data();
  d.triangles = triangles.data();
  d.materials = materials.data();
  d.patterns = patterns.data();
  return d;
}

FlatWorld flatten(const World& w) {
  Flattener f;
  Mat4 id = Mat4::identity();
  for (const Shape& s : w.objects) f.walk(s, id, false, id, -1, nullptr);
  memcpy(f.out.light.intensity, w.light_source.intensity, sizeof(f.out.light.intensity));
  memcpy(f.out.light.position, w.light_source.position, sizeof(f.out.light.position));
  return std::move(f.out);
}

// ===========================================================================
// Canvas::save (src/canvas.rs:47-94)
// ===========================================================================
std::string ppm_encode(size_t width, size_t height, const uint8_t* rgb8) {
  // decimal text of every byte value, once: formatting dominates the writer otherwise
  struct Digits {
    char text[256][4];
    uint8_t len[256];
    Digits() {
      for (int v = 0; v < 256; v++) len[v] = (uint8_t)snprintf(text[v], sizeof(text[v]), "%d", v);
    }
  };
  static const Digits D;
  std::string head = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
  std::string out;
  out.resize(head.size() + width * height * 12 + 2);  // at most "255 255 255" plus a separator per pixel
  memcpy(&out[0], head.data(), head.size());
  char* w = &out[head.size()];
  auto put = [&](unsigned v) {
    memcpy(w, D.text[v], 4);  // copies the terminator too; the next write overwrites it
    w += D.len[v];
  };
  size_t col = 1;
  for (size_t p = 0; p < width * height; p++) {
    const unsigned a = rgb8[p * 3], b = rgb8[p * 3 + 1], c = rgb8[p * 3 + 2];
    const size_t la = D.len[a], lb = D.len[b], lc = D.len[c];
    if (col + la > 70) {
      *w++ = '\n';
      put(a); *w++ = ' '; put(b); *w++ = ' '; put(c);
      col = la + lb + lc + 3;
    } else if (col + la + lb > 70) {
      *w++ = ' ';
      put(a); *w++ = '\n'; put(b); *w++ = ' '; put(c);
      col = lb + lc + 2;
    } else if (col + la + lb + lc > 70) {
      *w++ = ' ';
      put(a); *w++ = ' '; put(b); *w++ = '\n'; put(c);
      col = lc + 1;
    } else {
      if (col != 1) {
        *w++ = ' ';
        col += 1;
      }
      put(a); *w++ = ' '; put(b); *w++ = ' '; put(c);
      col += la + lb + lc + 2;
    }
  }
  *w++ = '\n';
  out.resize((size_t)(w - out.data()));
  return out;
}

// ===========================================================================
// Canvas (src/canvas.rs:22-140)
// ===========================================================================
Canvas::Canvas(size_t w, size_t h) : width(w), height(h), pixels(w * h) {}

void Canvas::write_pixel(size_t x, size_t y, const Color& pixel) {
  if (x >= width || y >= height) return;
  pixels[y * width + x] = pixel;
}

std::optional<Color> Canvas::read_pixel(size_t x, size_t y) const {
  if (x >= width || y >= height) return std::nullopt;
  return pixels[y * width + x];
}

std::vector<uint8_t> Canvas::quantised() const {
  auto q = [](double c) -> uint8_t {
    double v = c * 255.0;  // f64::clamp keeps NaN, `as usize` turns it into 0
    if (v < 0.0) v = 0.0;
    else if (v > 255.0) v = 255.0;
    v = std::ceil(v);
    return v == v ? (uint8_t)v : (uint8_t)0;
  };
  std::vector<uint8_t> out(pixels.size() * 3);
  for (size_t p = 0; p < pixels.size(); p++) {
    out[p * 3 + 0] = q(pixels[p].r);
    out[p * 3 + 1] = q(pixels[p].g);
    out[p * 3 + 2] = q(pixels[p].b);
  }
  return out;
}

void Canvas::save(const std::string& path) const {
  std::vector<uint8_t> rgb8 = quantised();
  const uint64_t cap = rt_ppm_max_bytes(width, height);
  std::unique_ptr<char, void (*)(void*)> text((char*)malloc(cap), free);
  if (!text) throw std::bad_alloc();
  uint64_t length = 0;
  if (rt_ppm_encode_host(0, rgb8.data(), width, height, text.get(), cap, &length) != RT_OK)
    throw std::runtime_error(rt_last_error());
  std::ofstream of(path, std::ios::binary);
  if (!of) throw Panic("could not create output file " + path);
  of.write(text.get(), (std::streamsize)length);
}

namespace {

// All workers leave together; reusable.
class Rendezvous {
 public:
  explicit Rendezvous(size_t n) : n_(n) {}
  void arrive_and_wait() {
    std::unique_lock<std::mutex> lock(m_);
    const size_t round = round_;
    if (++here_ == n_) {
      here_ = 0;
      round_++;
      cv_.notify_all();
    } else {
      cv_.wait(lock, [&] { return round_ != round; });
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  size_t n_, here_ = 0, round_ = 0;
};

void check_rt(int rc) {
  if (rc != RT_OK) throw std::runtime_error(rt_last_error());
}

// Workers of a render: one per GPU, at most as many as asked for (`jobs` counts worker threads in the
// reference; here one worker drives one GPU).  RT_AMD_REHEARSE_JOBS=1 keeps up to 16 workers on fewer GPUs,
// dealt round-robin, so that the several-GPU path can be run on a machine with one.
size_t workers_for(size_t jobs, int gpus) {
  const char* e = getenv("RT_AMD_REHEARSE_JOBS");
  if (e && atoi(e) != 0) return std::min<size_t>(jobs, 16);
  return std::min(jobs, (size_t)gpus);
}

// One acceleration structure for all GPUs (src/parallel.rs:183-204 shares one World between its threads):
// the first GPU builds the scene, the others receive device-to-device copies of it (rt_scene_clone), made
// by their own worker threads before any of them renders; then worker k renders part k of `jobs` on GPU k
// (k modulo `gpus` when workers_for() let a rehearsal have more workers than GPUs).
void render_on_gpus(const rt_scene_desc& desc, size_t jobs, int gpus, const std::function<int(rt_scene*, size_t)>& render) {
  rt_scene* first = nullptr;
  check_rt(rt_scene_create(&desc, 0, &first));
  std::vector<std::string> errors(jobs);
  Rendezvous cloned(jobs);
  auto worker = [&](size_t k) {
    rt_scene* sc = first;
    if (k > 0 && rt_scene_clone(first, (int)(k % (size_t)gpus), &sc) != RT_OK) {
      errors[k] = rt_last_error();
      sc = nullptr;
    }
    cloned.arrive_and_wait();  // the source scene is read by the copies: nobody renders with it before they are made
    if (sc && render(sc, k) != RT_OK) errors[k] = rt_last_error();
    if (k > 0) rt_scene_destroy(sc);
  };
  std::vector<std::thread> threads;
  for (size_t k = 1; k < jobs; k++) threads.emplace_back(worker, k);
  worker(0);
  for (auto& t : threads) t.join();
  rt_scene_destroy(first);
  for (const std::string& e : errors)
    if (!e.empty()) throw std::runtime_error(e);
}

}  // namespace

Canvas render_canvas(const World& world, const Camera& camera, size_t jobs) {
  if (jobs == 0) throw Panic("assertion failed: size > 0");  // src/parallel.rs:94
  int have = rt_device_count();
  if (have < 0) throw std::runtime_error(rt_last_error());
  if (have == 0) throw std::runtime_error("no GPU visible");
  jobs = workers_for(jobs, have);
  FlatWorld flat = flatten(world);
  rt_scene_desc desc = flat.desc();
  rt_camera cam = camera.describe();
  Canvas canvas(camera.hsize, camera.vsize);
  static_assert(sizeof(Color) == 3 * sizeof(double), "Color is three packed doubles");
  double* rgb64 = canvas.pixels.empty() ? nullptr : &canvas.pixels[0].r;
  // every GPU writes its bands of 32 rows straight into the canvas
  render_on_gpus(desc, jobs, have, [&](rt_scene* sc, size_t k) {
    return rt_render_host(sc, &cam, (int)k, (int)jobs, RT_FLAG_BANDS, nullptr, rgb64, nullptr, nullptr);
  });
  return canvas;
}

// ===========================================================================
// parallel_render (src/parallel.rs:183-204): image tiles over `jobs` GPUs
// ===========================================================================
void parallel_render(const World& world, const Camera& camera, size_t jobs, const std::string& out_path) {
  if (jobs == 0) throw Panic("assertion failed: size > 0");  // src/parallel.rs:94
  int have = rt_device_count();
  if (have < 0) throw std::runtime_error(rt_last_error());
  if (have == 0) throw std::runtime_error("no GPU visible");
  printf("Rendering using %zu threads...\n", jobs);  // the three lines of src/parallel.rs:188,202-203
  fflush(stdout);
  // `jobs` counts worker threads in the reference; here one worker drives one GPU, so a request for
  // more workers than GPUs uses every GPU there is
  jobs = workers_for(jobs, have);
  // RT_AMD_TIMINGS=1: where the wall time of one call goes, on stderr
  const bool timings = getenv("RT_AMD_TIMINGS") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t) {
    auto t1 = now();
    if (timings) fprintf(stderr, "[rt_host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t).count());
    t = t1;
  };
  auto t = now();
  FlatWorld flat = flatten(world);
  rt_scene_desc desc = flat.desc();
  rt_camera cam = camera.describe();
  size_t np = camera.hsize * camera.vsize;
  lap("flatten", t);
  // One frame for all GPUs: part k is the bands of 32 pixel rows k, k + jobs, ... and every worker
  // copies exactly its bands into the frame, so the pieces need no merging afterwards.
  struct Frame {
    uint8_t* p = nullptr;
    bool pinned = false;
    ~Frame() {
      if (pinned) rt_host_free(p);
      else free(p);
    }
  } frame;
  frame.p = (uint8_t*)rt_host_alloc(np * 3);
  frame.pinned = frame.p != nullptr;
  lap("device start, frame memory", t);
  if (!frame.p) frame.p = (uint8_t*)malloc(std::max<size_t>(np * 3, 1));
  if (!frame.p) throw std::bad_alloc();
  render_on_gpus(desc, jobs, have, [&](rt_scene* sc, size_t k) {
    return rt_render_host(sc, &cam, (int)k, (int)jobs, RT_FLAG_BANDS, frame.p, nullptr, nullptr, nullptr);
  });
  lap("scene upload, BVH, render", t);
  // Canvas::save (src/canvas.rs:47-94): the text is formatted on the first GPU (the frame goes up,
  // the file comes down), about ten times faster than this host's one-core writer at 4K
  // (pageable memory for the text: page-locking 12 bytes per pixel for one copy costs more than it saves)
  const uint64_t cap = rt_ppm_max_bytes(camera.hsize, camera.vsize);
  std::unique_ptr<char, void (*)(void*)> text((char*)malloc(cap), free);
  if (!text) throw std::bad_alloc();
  uint64_t length = 0;
  if (rt_ppm_encode_host(0, frame.p, camera.hsize, camera.vsize, text.get(), cap, &length) != RT_OK)
    throw std::runtime_error(rt_last_error());
  lap("PPM text (device)", t);
  std::ofstream of(out_path, std::ios::binary);
  if (!of) throw Panic("could not create output file " + out_path);
  of.write(text.get(), (std::streamsize)length);
  of.close();
  lap("file write", t);
  printf("...done.\n\n");
  printf("Saved render to \"%s\".\n", out_path.c_str());
}

// ===========================================================================
// FrameLoop: parallel_render's workers over many frames
// ===========================================================================
FrameLoop::FrameLoop(const World& world, const Camera& camera, const std::vector<int>& devices, int depth, const Share& share)
    : devices_(devices), depth_(depth) {
  if (devices.empty()) throw Panic("assertion failed: size > 0");  // src/parallel.rs:94
  if (depth < 1 || depth > 16) throw std::invalid_argument("FrameLoop: depth must lie in 1..16");
  first_part_ = share.first_part;
  total_parts_ = share.total_parts ? share.total_parts : devices.size();
  if (first_part_ + devices.size() > total_parts_) throw std::invalid_argument("FrameLoop: more parts than the frame has");
  own_ring_ = share.ring == nullptr;
  if (const char* e = getenv("RT_AMD_BAND_ROWS")) {
    const int rows = atoi(e);
    if (rows != 8 && rows != 16 && rows != 32) throw std::invalid_argument("RT_AMD_BAND_ROWS must be 8, 16 or 32");
    band_rows_ = (size_t)rows;
  }
  cam_ = camera.describe();
  frame_bytes_ = (size_t)cam_.hsize * cam_.vsize * 3;
  if (frame_bytes_ == 0) throw std::invalid_argument("FrameLoop: empty canvas");
  try {
    FlatWorld flat = flatten(world);
    rt_scene_desc desc = flat.desc();
    for (const rt_material& m : flat.materials) spawns_ = spawns_ || m.reflective != 0.0 || m.transparency != 0.0;
    scenes_.assign(devices.size(), nullptr);
    check_rt(rt_scene_create(&desc, devices[0], &scenes_[0]));
    check_rt(rt_scene_footprint(scenes_[0], nullptr, &build_seconds_));
    for (size_t k = 1; k < devices.size(); k++) check_rt(rt_scene_clone(scenes_[0], devices[k], &scenes_[k]));
    frame_stride_ = frame_bytes_;
    if (own_ring_ && total_parts_ > 1) {
      const size_t round_bytes = total_parts_ * band_rows_ * cam_.hsize * 3;  // one band of every part
      frame_stride_ = (frame_bytes_ + round_bytes - 1) / round_bytes * round_bytes;
    }
    if (own_ring_) {
      host_block_ = (uint8_t*)rt_host_alloc((size_t)depth * frame_stride_);
      if (!host_block_) throw std::runtime_error("FrameLoop: no page-locked host memory for the frame ring");
    }
    for (int k = 0; k < depth; k++) host_.push_back((own_ring_ ? host_block_ : share.ring) + (size_t)k * frame_stride_);
    dev_frames_.resize(devices.size());
    for (size_t k = 0; k < devices.size(); k++) {
      void* p = nullptr;
      check_rt(rt_device_alloc(devices[k], (size_t)depth * frame_stride_, &p));
      for (int j = 0; j < depth; j++) dev_frames_[k].push_back((uint8_t*)p + (size_t)j * frame_stride_);
    }
  } catch (...) {
    release();
    throw;
  }
}

void FrameLoop::release() {
  for (size_t k = 0; k < dev_frames_.size(); k++)
    if (!dev_frames_[k].empty()) rt_device_free(devices_[k], dev_frames_[k][0]);
  dev_frames_.clear();
  for (rt_scene* s : scenes_) rt_scene_destroy(s);
  scenes_.clear();
  if (host_block_) rt_host_free(host_block_);
  host_block_ = nullptr;
  host_.clear();
}

FrameLoop::~FrameLoop() { release(); }

uint64_t FrameLoop::scene_bytes() const {
  uint64_t b = 0;
  if (!scenes_.empty()) (void)rt_scene_footprint(scenes_[0], &b, nullptr);
  return b;
}

uint32_t FrameLoop::band_flags() const {
  return RT_FLAG_BANDS | (band_rows_ == 8 ? RT_FLAG_BANDS_8 : (band_rows_ == 16 ? RT_FLAG_BANDS_16 : 0u));
}

FrameLoop::Result FrameLoop::run(size_t warmup, size_t frames, bool to_host) {
  const size_t workers = scenes_.size(), n = total_parts_;  // n: parts of the frame, of which this loop renders `workers`
  Result result;
  result.part.assign(workers, rt_stats{});
  std::vector<std::string> errors(workers);
  Rendezvous meet(workers);
  std::chrono::steady_clock::time_point t0, t1;
  const size_t w = cam_.hsize, rows_total = cam_.vsize, band = band_rows_, band_bytes = band * w * 3;
  const uint32_t part_flags = band_flags();
  // A fence on the copy stream after every copy costs about 5 us of the copy engine's time (the next
  // copy starts only when the fence's marker has gone through the stream's queue).  Copies on one stream
  // finish in order, so a fence after every `stride`-th copy tells as much: a slot is free once a fence
  // recorded at or after its copy has passed.  stride <= depth - 1 keeps the fence a frame waits for
  // recorded before that frame is enqueued, with at least one other slot to render into meanwhile.
  size_t stride = depth_ >= 3 ? 2 : 1;
  if (const char* e = getenv("RT_AMD_COPIES_PER_FENCE")) stride = (size_t)std::max(1, atoi(e));
  stride = std::min(stride, (size_t)std::max(1, depth_ - 1));
  // Frames per launch.  A part of a frame alone is a short launch that ends with its slowest wave while most of the
  // GPU idles; several frames of the part in one launch (rt_render_frames) fill it as a whole frame does, and one
  // copy for them all pays the fixed cost of a copy once (whole frames gain by that alone).  Three launches or more
  // are to be in flight (one renders, one is being copied, one waits for its slots), so a launch takes at most a
  // third of the ring, and a count that divides the ring keeps a launch's slots in one piece (one GPU rendering an
  // eighth of the teapot frame at 1080p, ms per frame on the host: 0.034 one at a time in a ring of 8, 0.024 two at
  // a time, 0.019-0.021 four at a time in a ring of 12 or 16; four at a time in a ring of 8: 0.028).  More than four
  // launches in flight are no help (whole frames one at a time: 0.128 ms per frame in a ring of 4 and in a ring of 8).
  size_t group = spawns_ ? 1 : 4;
  const bool asked = getenv("RT_AMD_FRAMES_PER_LAUNCH") != nullptr;
  if (asked) group = (size_t)std::max(1, atoi(getenv("RT_AMD_FRAMES_PER_LAUNCH")));
  group = spawns_ ? 1 : std::max<size_t>(1, std::min({group, (size_t)depth_ / (asked ? 2 : 3), (size_t)RT_MAX_FRAMES_PER_CALL}));
  while ((size_t)depth_ % group) group--;
  frames_per_launch_ = group;
  auto worker = [&](size_t k) {
    rt_scene* sc = scenes_[k];
    const size_t part = first_part_ + k;
    bool ok = true;
    std::vector<int> fenced_by((size_t)depth_, -1);  // per slot: the slot whose fence covers this slot's last copy
    std::vector<int> unfenced;                        // slots copied since the last fence
    // At most three fenced copies (or pairs of copies) wait in the copy stream: the worker holds back a fourth until
    // the oldest has gone through.  With more of them queued the same loop alternates between two rates from one run
    // to the next -- whole frames four per launch in a ring of 16: 0.116 or 0.118-0.125 ms per frame, two per launch
    // in a ring of 12: 0.120 or 0.22-0.24 -- and with the limit every ring and launch size repeats to 0.0001 ms.  Two
    // are too few for launches of four frames (0.127-0.129: the host's turn-around shows), one for any (0.142).
    std::vector<int> fences_out;  // slots whose fence was recorded and not yet waited for, oldest first
    auto fence_out = [&](int slot) {
      // a slot is fenced again only after its last fence was waited for (the ring): an older entry of it is done
      fences_out.erase(std::remove(fences_out.begin(), fences_out.end(), slot), fences_out.end());
      fences_out.push_back(slot);
    };
    auto make_room = [&]() {
      while (ok && fences_out.size() >= 3) {
        if (rt_copy_wait(sc, (uint32_t)fences_out.front()) != RT_OK) ok = false;
        fences_out.erase(fences_out.begin());
      }
    };
    size_t last_frame = 0;                            // of the stretch of frames being enqueued
    auto render_part = [&](size_t f, size_t count) {  // frames f .. f + count - 1 in one launch
      if (!ok) return;
      rt_camera cams[RT_MAX_FRAMES_PER_CALL];
      uint8_t* devs[RT_MAX_FRAMES_PER_CALL];
      for (size_t j = 0; j < count && ok; j++) {
        const int slot = (int)((f + j) % (size_t)depth_);
        cams[j] = cam_;
        devs[j] = dev_frames_[k][(size_t)slot];
        // the copy that last read this device frame (depth frames ago) must be through before it is rendered over
        if (to_host && fenced_by[(size_t)slot] >= 0 && rt_copy_wait(sc, (uint32_t)fenced_by[(size_t)slot]) != RT_OK) ok = false;
      }
      if (ok && count == 1 && rt_render(sc, &cam_, (int)part, (int)n, RT_FLAG_ASYNC | part_flags, devs[0], nullptr, nullptr, nullptr) != RT_OK)
        ok = false;
      if (ok && count > 1 && rt_render_frames(sc, cams, (int)count, (int)part, (int)n, RT_FLAG_ASYNC | part_flags, devs) != RT_OK) ok = false;
      if (!ok) errors[k] = rt_last_error();
    };
    // The frames of one launch, in consecutive slots of a padded ring: every band of this part in all of them with one
    // strided copy (whole bands: what lies beyond the image's last row is padding on both sides).
    // (Whole frames lie back to back in any ring: one plain copy.)
    const bool one_copy_per_launch = n == 1 || frame_stride_ % (n * band_bytes) == 0;
    auto copy_launch = [&](size_t f, size_t count) {
      if (!ok || !to_host) return;
      const int slot = (int)(f % (size_t)depth_);
      const size_t rounds = frame_stride_ / (n * band_bytes), off = part * band_bytes;
      make_room();
      if (n == 1) {
        if (rt_download(sc, dev_frames_[k][(size_t)slot], host_[(size_t)slot], count * frame_bytes_, nullptr) != RT_OK) ok = false;
      } else if (rt_download_2d(sc, dev_frames_[k][(size_t)slot] + off, host_[(size_t)slot] + off, band_bytes, count * rounds, n * band_bytes,
                                nullptr) != RT_OK) {
        ok = false;
      }
      const int last = slot + (int)count - 1;
      if (ok && rt_copy_fence(sc, (uint32_t)last) != RT_OK) ok = false;
      fence_out(last);
      for (int s : unfenced) fenced_by[(size_t)s] = last;
      unfenced.clear();
      for (size_t j = 0; j < count; j++) fenced_by[(size_t)slot + j] = last;
      if (!ok) errors[k] = rt_last_error();
    };
    auto copy_part = [&](size_t f, bool last_of_launch) {
      if (!ok || !to_host) return;
      const int slot = (int)(f % (size_t)depth_);
      uint8_t* dev = dev_frames_[k][(size_t)slot];
      auto fence = [&]() {  // after this slot's copy has been enqueued
        unfenced.push_back(slot);
        if (unfenced.size() < stride && f != last_frame && !(last_of_launch && group > 1)) return;
        if (rt_copy_fence(sc, (uint32_t)slot) != RT_OK) ok = false;
        fence_out(slot);
        for (int s : unfenced) fenced_by[(size_t)s] = slot;
        unfenced.clear();
      };
      if (unfenced.empty()) make_room();  // before the first copy of a new fenced group
      if (n == 1) {  // the whole frame is this part's: one contiguous copy
        if (rt_download(sc, dev, host_[(size_t)slot], frame_bytes_, nullptr) != RT_OK) ok = false;
        if (ok) fence();
      } else {
        // this part's bands: whole bands with one strided copy, a last partial band with a plain one
        const size_t first = part * band, n_bands_mine = first < rows_total ? (rows_total - first + n * band - 1) / (n * band) : 0;
        size_t whole = n_bands_mine;
        if (n_bands_mine && first + (n_bands_mine - 1) * n * band + band > rows_total) whole--;
        uint8_t* host = host_[(size_t)slot];
        const size_t off = first * w * 3;
        if (whole && rt_download_2d(sc, dev + off, host + off, band_bytes, whole, n * band_bytes, nullptr) != RT_OK) ok = false;
        if (ok && whole != n_bands_mine) {
          const size_t y0 = first + whole * n * band, o = y0 * w * 3;
          if (rt_download(sc, dev + o, host + o, (rows_total - y0) * w * 3, nullptr) != RT_OK) ok = false;
        }
        if (ok) fence();
      }
      if (!ok) errors[k] = rt_last_error();
    };
    // A frame of a scene that spawns rays is enqueued unverified (whether its ray queues had room is known when
    // its counters are back), and a download of its image waits for that.  Its copy is therefore asked for after
    // the next frame has been enqueued, so the GPU has work while the host looks at the counters; the frames of
    // other scenes need no such look and are copied as soon as they are enqueued.
    const bool copy_one_late = spawns_ && to_host && depth_ >= 2;
    auto stretch = [&](size_t first_f, size_t count) {
      // a stretch starts and ends with small launches: the first copy need not wait for four frames, and the last
      // copy, which nothing overlaps, is a short one
      for (size_t f = first_f, g = 1; f < first_f + count; f += g) {
        const size_t left = first_f + count - f;
        g = std::min({group, f - first_f + 1, (left + 1) / 2});
        render_part(f, g);
        if (copy_one_late) {  // group == 1
          if (f > first_f) copy_part(f - 1, false);
        } else if (one_copy_per_launch && f % (size_t)depth_ + g <= (size_t)depth_) {
          copy_launch(f, g);
        } else {
          for (size_t j = 0; j < g; j++) copy_part(f + j, j + 1 == g);
        }
      }
      if (copy_one_late && count > 0) copy_part(first_f + count - 1, false);
    };
    last_frame = warmup ? warmup - 1 : 0;
    stretch(0, warmup);
    rt_stats st{};
    if (ok && rt_scene_wait(sc, &st) != RT_OK) {
      ok = false;
      errors[k] = rt_last_error();
    }
    meet.arrive_and_wait();  // every GPU idle, every worker ready
    if (k == 0) t0 = std::chrono::steady_clock::now();
    last_frame = warmup + frames - 1;
    stretch(warmup, frames);
    if (ok && rt_scene_wait(sc, &result.part[k]) != RT_OK) {  // rendered and, with to_host, on the host
      ok = false;
      errors[k] = rt_last_error();
    }
    meet.arrive_and_wait();
    if (k == 0) t1 = std::chrono::steady_clock::now();
  };
  std::vector<std::thread> threads;
  for (size_t k = 1; k < workers; k++) threads.emplace_back(worker, k);
  worker(0);
  for (auto& t : threads) t.join();
  for (const std::string& e : errors)
    if (!e.empty()) throw std::runtime_error(e);
  result.seconds = std::chrono::duration<double>(t1 - t0).count();
  return result;
}

std::vector<rt_stats> FrameLoop::blocking_frames(size_t frames, bool counters) {
  std::vector<rt_stats> out(scenes_.size(), rt_stats{});
  for (size_t f = 0; f < frames; f++)
    for (size_t k = 0; k < scenes_.size(); k++) {
      rt_stats st{};
      check_rt(rt_render(scenes_[k], &cam_, (int)(first_part_ + k), (int)total_parts_, (counters ? RT_FLAG_COUNTERS : 0u) | band_flags(),
                         dev_frames_[k][0], nullptr, nullptr, &st));
      rt_stats& o = out[k];
      o.primary += st.primary; o.shadow += st.shadow; o.reflect += st.reflect; o.refract += st.refract;
      o.render_ms += st.render_ms; o.kernel_ms += st.kernel_ms;
      o.kernel_launches += st.kernel_launches; o.generations += st.generations;
      o.node_visits += st.node_visits; o.tri_tests += st.tri_tests; o.quadric_tests += st.quadric_tests; o.gate_tests += st.gate_tests;
      o.shadow_node_visits += st.shadow_node_visits; o.shadow_tri_tests += st.shadow_tri_tests;
      o.shadow_quadric_tests += st.shadow_quadric_tests; o.shadow_gate_tests += st.shadow_gate_tests;
    }
  return out;
}

}  // namespace rthost
