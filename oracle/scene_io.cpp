// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp.  Scene-JSON and OBJ loaders
// of the CPU oracle, following the reference's scene.rs and obj.rs.  Where the
// reference panics this code throws std::runtime_error with the same message.
// No reference test covers scene.rs; obj.rs tests need model files the
// reference does not ship: parity unpinned, source text only.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_oracle.hpp"

namespace orc {
namespace {

// ---------------------------------------------------------------- mini JSON
struct JVal {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  bool is_integer = false;  // literal had no '.', 'e' or 'E'
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const char* key) const {
    const JVal* found = nullptr;
    for (auto& kv : obj)
      if (kv.first == key) found = &kv.second;  // serde keeps the last duplicate
    return found;
  }
};

struct JParser {
  const char* p;
  const char* end;
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("Failed to parse scene JSON: ") + what); }
  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  JVal value() {
    ws();
    if (p >= end) fail("unexpected end");
    JVal v;
    char c = *p;
    if (c == '{') {
      v.kind = JVal::Obj;
      p++;
      ws();
      if (p < end && *p == '}') { p++; return v; }
      for (;;) {
        ws();
        if (p >= end || *p != '"') fail("expected key");
        std::string k = string();
        ws();
        if (p >= end || *p != ':') fail("expected ':'");
        p++;
        JVal child = value();
        v.obj.emplace_back(std::move(k), std::move(child));
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = JVal::Arr;
      p++;
      ws();
      if (p < end && *p == ']') { p++; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.kind = JVal::Str;
      v.str = string();
    } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) {
      v.kind = JVal::Bool; v.b = true; p += 4;
    } else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) {
      v.kind = JVal::Bool; v.b = false; p += 5;
    } else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) {
      v.kind = JVal::Null; p += 4;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
      const char* s = p;
      if (*p == '-') p++;
      while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) p++;
      std::string lit(s, p);
      char* e = nullptr;
      v.kind = JVal::Num;
      v.num = strtod(lit.c_str(), &e);  // correctly rounded decimal -> f64
      if (e == lit.c_str() || *e) fail("bad number");
      v.is_integer = lit.find_first_of(".eE") == std::string::npos;
    } else {
      fail("unexpected character");
    }
    return v;
  }
  std::string string() {
    std::string s;
    p++;  // opening quote
    while (p < end && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= end) fail("bad escape");
        switch (*p) {
          case '"': s += '"'; break;
          case '\\': s += '\\'; break;
          case '/': s += '/'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'n': s += '\n'; break;
          case 'r': s += '\r'; break;
          case 't': s += '\t'; break;
          case 'u': {
            if (end - p < 5) fail("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) s += (char)cp;
            else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
            else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: fail("bad escape");
        }
        p++;
      } else {
        s += *p++;
      }
    }
    if (p >= end) fail("unterminated string");
    p++;
    return s;
  }
};

[[noreturn]] void bad(const std::string& what) { throw std::runtime_error("Failed to parse scene JSON: " + what); }

const JVal& need(const JVal& o, const char* key) {
  if (o.kind != JVal::Obj) bad("expected an object");
  const JVal* v = o.get(key);
  if (!v) bad(std::string("missing field `") + key + "`");
  return *v;
}
const JVal* opt(const JVal& o, const char* key) {
  const JVal* v = o.get(key);
  return (v && v->kind != JVal::Null) ? v : nullptr;
}
double as_f64(const JVal& v) {
  if (v.kind != JVal::Num) bad("expected a number");
  return v.num;
}
size_t as_usize(const JVal& v) {
  if (v.kind != JVal::Num || !v.is_integer || v.num < 0) bad("expected an unsigned integer");
  return (size_t)v.num;
}
std::vector<double> as_vec(const JVal& v) {
  if (v.kind != JVal::Arr) bad("expected an array of numbers");
  std::vector<double> out;
  for (auto& e : v.arr) out.push_back(as_f64(e));
  return out;
}
const std::string& as_str(const JVal& v) {
  if (v.kind != JVal::Str) bad("expected a string");
  return v.str;
}

// scene.rs:190-195 / 354-359: up to 16 numbers, zero-filled; more than 16 is an
// out-of-bounds index in the reference (panic)
Matrix matrix_from_vec(const std::vector<double>& v) {
  if (v.size() > 16) throw std::runtime_error("index out of bounds: transform has more than 16 elements");
  Matrix m;
  for (size_t i = 0; i < v.size(); i++) m.d[i] = v[i];
  return m;
}

// scene.rs:275-366
Pattern pattern_from_json(const JVal& j) {
  const std::string& ty = as_str(need(j, "ty"));
  auto primary = [&]() -> Color {
    const JVal* v = opt(j, "primary_color");
    return v ? color_from_vec(as_vec(*v)) : Color{1.0, 1.0, 1.0};
  };
  auto secondary = [&]() -> Color {
    const JVal* v = opt(j, "secondary_color");
    return v ? color_from_vec(as_vec(*v)) : Color{0.0, 0.0, 0.0};
  };
  // type-check the optional fields even when unused, as serde does
  if (const JVal* v = opt(j, "primary_color")) (void)as_vec(*v);
  if (const JVal* v = opt(j, "secondary_color")) (void)as_vec(*v);
  Pattern p;
  if (ty == "null") p = pattern_make(PatternType::Null);
  else if (ty == "point") p = pattern_make(PatternType::Point);
  else if (ty == "identity") p = pattern_make(PatternType::Identity, primary());
  else if (ty == "stripe") p = pattern_make(PatternType::Stripe, primary(), secondary());
  else if (ty == "ring") p = pattern_make(PatternType::Ring, primary(), secondary());
  else if (ty == "checker") p = pattern_make(PatternType::Checker, primary(), secondary());
  else if (ty == "gradient") p = pattern_make(PatternType::Gradient, primary(), secondary());
  else throw std::runtime_error("Unrecognized pattern in scene description JSON.");
  if (const JVal* t = opt(j, "transform")) p.transform = matrix_from_vec(as_vec(*t));
  else p.transform = mat_identity();
  return p;
}

// scene.rs:224-265
Material material_from_json(const JVal& j) {
  Material m;
  m.color = color_from_vec(as_vec(need(j, "color")));
  if (const JVal* v = opt(j, "pattern")) m.pattern = pattern_from_json(*v);
  if (const JVal* v = opt(j, "ambient")) m.ambient = as_f64(*v);
  if (const JVal* v = opt(j, "diffuse")) m.diffuse = as_f64(*v);
  if (const JVal* v = opt(j, "specular")) m.specular = as_f64(*v);
  if (const JVal* v = opt(j, "shininess")) m.shininess = as_f64(*v);
  if (const JVal* v = opt(j, "reflective")) m.reflective = as_f64(*v);
  if (const JVal* v = opt(j, "refractive_index")) m.refractive_index = as_f64(*v);
  if (const JVal* v = opt(j, "transparency")) m.transparency = as_f64(*v);
  return m;
}

std::string join_path(const std::string& base, const std::string& rel) {
  if (base.empty() || (!rel.empty() && rel[0] == '/')) return rel;
  return base + "/" + rel;
}

// scene.rs:95-207
Shape shape_from_json(const JVal& j, const std::string& base_dir) {
  const std::string& ty = as_str(need(j, "ty"));
  const JVal* children = opt(j, "children");
  if (children && children->kind != JVal::Arr) bad("children must be an array");
  Shape shape;
  if (ty == "empty") shape = Shape::empty();
  else if (ty == "sphere") shape = Shape::sphere();
  else if (ty == "plane") shape = Shape::plane();
  else if (ty == "cube") shape = Shape::cube();
  else if (ty == "cylinder") shape = Shape::cylinder();
  else if (ty == "bounded_cylinder") shape = Shape::bounded_cylinder(-1.0, 1.0);
  else if (ty == "capped_cylinder") shape = Shape::capped_cylinder(-1.0, 1.0);
  else if (ty == "bounded_cone") shape = Shape::bounded_cone(0.0, 1.0);
  else if (ty == "bounded_dn_cone") shape = Shape::bounded_cone(-1.0, 1.0);
  else if (ty == "capped_cone") shape = Shape::capped_cone(0.0, 1.0);
  else if (ty == "capped_dn_cone") shape = Shape::capped_cone(-1.0, 1.0);
  else if (ty == "group") {
    shape = Shape::group();
    // children are stored as converted, without add_child (scene.rs:114-118);
    // the set_transform below pushes the parent transform down
    if (children)
      for (auto& c : children->arr) shape.children.push_back(shape_from_json(c, base_dir));
  } else if (ty == "union" || ty == "intersection" || ty == "difference") {
    shape = Shape::empty();
    if (children) {
      // serde converts every listed child before the length check runs
      if (children->arr.size() < 2) throw std::runtime_error("CSG union must have at least two operands.");
      ShapeType op = ty == "union" ? ShapeType::Union
                     : ty == "intersection" ? ShapeType::Intersection : ShapeType::Difference;
      shape = Shape::csg(op, shape_from_json(children->arr[0], base_dir),
                         shape_from_json(children->arr[1], base_dir));
    }
  } else if (ty == "model") {
    const JVal* path = opt(j, "path");
    if (!path) throw std::runtime_error("Model requires a path in scene description JSON.");
    ObjResult obj = obj_parse(join_path(base_dir, as_str(*path)));
    shape = Shape::group();
    for (auto& kv : obj.groups) shape.children.push_back(kv.second);  // BTreeMap key order
  } else {
    throw std::runtime_error("Unrecognized shape type in scene description JSON.");
  }
  if (const JVal* t = opt(j, "transform")) shape.set_transform(matrix_from_vec(as_vec(*t)));
  else shape.set_transform(mat_identity());
  if (const JVal* m = opt(j, "material")) shape.material = material_from_json(*m);
  return shape;
}

bool parse_f64_token(const std::string& s, double* out) {
  if (s.empty() || isspace((unsigned char)s[0])) return false;
  char* e = nullptr;
  double v = strtod(s.c_str(), &e);
  if (e == s.c_str() || *e) return false;
  // Rust's f64 parser takes no hex floats
  if (s.find_first_of("xX") != std::string::npos) return false;
  *out = v;
  return true;
}
bool parse_usize_token(const std::string& s, size_t* out) {
  size_t i = 0;
  if (i < s.size() && s[i] == '+') i++;
  if (i >= s.size()) return false;
  size_t v = 0;
  for (; i < s.size(); i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (size_t)(s[i] - '0');
  }
  *out = v;
  return true;
}
std::vector<std::string> split_char(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t start = 0;
  for (;;) {
    size_t pos = s.find(sep, start);
    if (pos == std::string::npos) { out.push_back(s.substr(start)); break; }
    out.push_back(s.substr(start, pos - start));
    start = pos + 1;
  }
  return out;
}

struct FaceVertex { size_t v; bool has_t; size_t t; bool has_n; size_t n; };

// obj.rs:236-276
std::vector<Shape> fan_triangulation(const ObjResult& o, const std::vector<FaceVertex>& face) {
  std::vector<Shape> tris;
  if (face.empty()) throw std::runtime_error("attempt to subtract with overflow (empty face)");
  auto vert = [&](size_t idx) -> Tuple {
    if (idx == 0 || idx > o.vertices.size()) throw std::runtime_error("index out of bounds: vertex");
    return o.vertices[idx - 1];
  };
  auto norm = [&](size_t idx) -> Tuple {
    if (idx == 0 || idx > o.normals.size()) throw std::runtime_error("index out of bounds: normal");
    return o.normals[idx - 1];
  };
  for (size_t i = 1; i + 1 < face.size(); i++) {
    if (face[0].has_n && face[i].has_n && face[i + 1].has_n) {
      tris.push_back(Shape::smooth_triangle(vert(face[0].v), vert(face[i].v), vert(face[i + 1].v),
                                            norm(face[0].n), norm(face[i].n), norm(face[i + 1].n)));
    } else {
      tris.push_back(Shape::triangle(vert(face[0].v), vert(face[i].v), vert(face[i + 1].v)));
    }
  }
  return tris;
}

// obj.rs:95-194
void handle_command(ObjResult& o, const std::string& line, std::string& current_group) {
  std::vector<std::string> params = split_char(line, ' ');
  auto f64_or_panic = [&](const std::string& s) {
    double v;
    if (!parse_f64_token(s, &v)) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: ParseFloatError");
    return v;
  };
  if (params[0] == "v") {
    if (params.size() < 4) { o.ignored_lines++; return; }
    double x = f64_or_panic(params[1]), y = f64_or_panic(params[2]), z = f64_or_panic(params[3]);
    o.vertices.push_back(point(x, y, z));
  } else if (params[0] == "vn") {
    if (params.size() < 4) { o.ignored_lines++; return; }
    double x = f64_or_panic(params[1]), y = f64_or_panic(params[2]), z = f64_or_panic(params[3]);
    o.normals.push_back(vector(x, y, z));
  } else if (params[0] == "f") {
    std::vector<FaceVertex> face;
    for (size_t k = 1; k < params.size(); k++) {
      const std::string& vertex = params[k];
      size_t v;
      if (parse_usize_token(vertex, &v)) {
        face.push_back({v, false, 0, false, 0});
      } else {
        std::vector<std::string> attrs = split_char(vertex, '/');
        if (attrs.size() > 3) { o.ignored_lines++; return; }
        FaceVertex fv{0, false, 0, false, 0};
        if (!parse_usize_token(attrs[0], &fv.v)) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: ParseIntError");
        if (attrs.size() < 3) throw std::runtime_error("index out of bounds: face vertex needs v/t/n");
        fv.has_t = parse_usize_token(attrs[1], &fv.t);
        fv.has_n = parse_usize_token(attrs[2], &fv.n);
        face.push_back(fv);
      }
    }
    for (const FaceVertex& fv : face)
      if (fv.v > o.vertices.size() || (fv.has_n ? fv.n : 0) > o.normals.size()) { o.ignored_lines++; return; }
    std::vector<Shape> tris = fan_triangulation(o, face);
    auto it = o.groups.find(current_group);
    if (it == o.groups.end()) throw std::runtime_error("OBJ group should be instantiated.");
    for (Shape& t : tris) it->second.add_child(std::move(t));
  } else if (!line.empty() && line[0] == 'g') {
    if (params.size() < 2) { o.ignored_lines++; return; }
    if (!o.groups.count(params[1])) o.groups.emplace(params[1], Shape::group());
    current_group = params[1];
  }
  o.ignored_lines++;
}

}  // namespace

ObjResult obj_parse_text(const std::string& text) {  // obj.rs:26-68
  ObjResult o;
  o.groups.emplace("", Shape::group());
  std::string current_group;
  size_t start = 0;
  while (start < text.size()) {
    size_t nl = text.find('\n', start);
    std::string line = text.substr(start, nl == std::string::npos ? std::string::npos : nl - start);
    start = nl == std::string::npos ? text.size() : nl + 1;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    handle_command(o, line, current_group);
  }
  return o;
}

ObjResult obj_parse(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Failed to open file \"" + path + "\".");
  std::stringstream ss;
  ss << f.rdbuf();
  return obj_parse_text(ss.str());
}

// scene.rs:25-50, 77-84
Scene scene_from_json(const std::string& text, const std::string& base_dir) {
  JParser jp{text.data(), text.data() + text.size()};
  JVal root = jp.value();
  jp.ws();
  if (jp.p != jp.end) bad("trailing characters");

  size_t width = as_usize(need(root, "canvas_width"));
  size_t height = as_usize(need(root, "canvas_height"));
  double fov = as_f64(need(root, "field_of_view"));
  Tuple from = tuple_from_vec(as_vec(need(root, "camera_from")));
  Tuple to = tuple_from_vec(as_vec(need(root, "camera_to")));
  Tuple up = tuple_from_vec(as_vec(need(root, "camera_up")));
  const JVal& light = need(root, "light");
  Color intensity = color_from_vec(as_vec(need(light, "intensity")));
  Tuple position = tuple_from_vec(as_vec(need(light, "position")));
  const JVal& shapes = need(root, "shapes");
  if (shapes.kind != JVal::Arr) bad("shapes must be an array");

  from.w = 1.0;
  to.w = 1.0;
  Scene s;
  s.camera = Camera::make(width, height, fov, mat_view_transform(from, to, up));
  position.w = 1.0;
  s.world = World::empty_world();
  s.world.light = PointLight{intensity, position};
  for (auto& sj : shapes.arr) s.world.objects.push_back(shape_from_json(sj, base_dir));
  s.world.finalize();
  return s;
}

}  // namespace orc
