// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp.  C entry points over the CPU
// oracle so that tests (ctypes) and bench.py's cpu_baseline leg can call it.
#include <cstring>

#include "rt_oracle.hpp"

using namespace orc;

namespace {
void set_err(char* err, int errlen, const char* msg) {
  if (err && errlen > 0) {
    strncpy(err, msg, (size_t)errlen - 1);
    err[errlen - 1] = 0;
  }
}
}  // namespace

extern "C" {

void* orc_scene_from_json(const char* text, const char* base_dir, char* err, int errlen) {
  try {
    Scene* s = new Scene(scene_from_json(text, base_dir ? base_dir : ""));
    return s;
  } catch (const std::exception& e) {
    set_err(err, errlen, e.what());
    return nullptr;
  }
}

void orc_scene_free(void* h) { delete (Scene*)h; }

// out[0..3] = width, height, leaf count, top-level object count
void orc_scene_info(void* h, uint64_t* out) {
  Scene* s = (Scene*)h;
  out[0] = s->camera.hsize;
  out[1] = s->camera.vsize;
  out[2] = (uint64_t)s->world.n_leaves;
  out[3] = s->world.objects.size();
}

// Same view and field of view, different canvas size (Camera::new again).
void orc_scene_set_canvas(void* h, uint64_t w, uint64_t hh) {
  Scene* s = (Scene*)h;
  s->camera = Camera::make(w, hh, s->camera.field_of_view, s->camera.transform);
}

// from/to: 3 doubles (w forced to 1); up: 4 doubles (scene.rs:28-32)
void orc_scene_set_camera(void* h, uint64_t w, uint64_t hh, double fov, const double* from,
                          const double* to, const double* up) {
  Scene* s = (Scene*)h;
  Tuple f = point(from[0], from[1], from[2]), t = point(to[0], to[1], to[2]);
  Tuple u{up[0], up[1], up[2], up[3]};
  s->camera = Camera::make(w, hh, fov, mat_view_transform(f, t, u));
}

int orc_render(void* h, int jobs, uint64_t x0, uint64_t xstep, uint64_t y0, uint64_t ystep,
               double* rgb, int32_t* hit_leaf, uint64_t* counts4, char* err, int errlen) {
  Scene* s = (Scene*)h;
  try {
    RenderOut out;
    parallel_render(s->world, s->camera, jobs, x0, xstep, y0, ystep, &out);
    if (rgb) memcpy(rgb, out.rgb.data(), out.rgb.size() * sizeof(double));
    if (hit_leaf) memcpy(hit_leaf, out.hit_leaf.data(), out.hit_leaf.size() * sizeof(int32_t));
    if (counts4) {
      counts4[0] = out.counts.primary;
      counts4[1] = out.counts.shadow;
      counts4[2] = out.counts.reflect;
      counts4[3] = out.counts.refract;
    }
    return 0;
  } catch (const std::exception& e) {
    set_err(err, errlen, e.what());
    return -1;
  }
}

// rays6: n x (ox,oy,oz,dx,dy,dz).  World::color_at on Ray4D::new(origin, direction).
int orc_color_at(void* h, uint64_t n, const double* rays6, int remaining, double* rgb,
                 int32_t* leaf, uint64_t* counts4) {
  Scene* s = (Scene*)h;
  RayCounts rc;
  for (uint64_t i = 0; i < n; i++) {
    const double* r = rays6 + i * 6;
    Ray ray = ray_new(point(r[0], r[1], r[2]), vector(r[3], r[4], r[5]));
    int lf = -1;
    rc.primary++;
    Color c = s->world.color_at(ray, remaining, &rc, &lf);
    rgb[i * 3 + 0] = c.r;
    rgb[i * 3 + 1] = c.g;
    rgb[i * 3 + 2] = c.b;
    if (leaf) leaf[i] = lf;
  }
  if (counts4) {
    counts4[0] = rc.primary; counts4[1] = rc.shadow; counts4[2] = rc.reflect; counts4[3] = rc.refract;
  }
  return 0;
}

// Debugging aid: every World::color_at call that hit something while tracing one ray, in call order.
// Each record is 14 doubles: ray (6), remaining, leaf, t, n1, n2, returned colour (3).  Returns the count.
uint64_t orc_trace(void* h, const double* ray6, int remaining, double* out14, uint64_t max_records) {
  Scene* s = (Scene*)h;
  std::vector<TraceRecord> trace;
  RayCounts rc;
  rc.trace = &trace;
  s->world.color_at(ray_new(point(ray6[0], ray6[1], ray6[2]), vector(ray6[3], ray6[4], ray6[5])), remaining, &rc);
  uint64_t n = std::min<uint64_t>(trace.size(), max_records);
  for (uint64_t i = 0; i < n; i++) {
    const TraceRecord& t = trace[i];
    double* o = out14 + i * 14;
    for (int k = 0; k < 6; k++) o[k] = t.ray[k];
    o[6] = t.remaining; o[7] = t.leaf; o[8] = t.t; o[9] = t.n1; o[10] = t.n2;
    o[11] = t.color[0]; o[12] = t.color[1]; o[13] = t.color[2];
  }
  return n;
}

// World::intersect followed by Intersections::hit: t (NaN when none) and leaf id (-1 when none).
int orc_intersect_hit(void* h, uint64_t n, const double* rays6, double* t_out, int32_t* leaf_out) {
  Scene* s = (Scene*)h;
  for (uint64_t i = 0; i < n; i++) {
    const double* r = rays6 + i * 6;
    Ray ray = ray_new(point(r[0], r[1], r[2]), vector(r[3], r[4], r[5]));
    Intersections xs;
    s->world.intersect(ray, xs);
    Intersection hit;
    if (xs.hit(&hit)) {
      t_out[i] = hit.t;
      leaf_out[i] = hit.what->leaf_id;
    } else {
      t_out[i] = std::nan("");
      leaf_out[i] = -1;
    }
  }
  return 0;
}

int orc_is_shadowed(void* h, uint64_t n, const double* pts3, uint8_t* out) {
  Scene* s = (Scene*)h;
  for (uint64_t i = 0; i < n; i++)
    out[i] = s->world.is_shadowed(point(pts3[i * 3], pts3[i * 3 + 1], pts3[i * 3 + 2]), nullptr) ? 1 : 0;
  return 0;
}

int orc_ray_for_pixel(void* h, uint64_t n, const uint32_t* xy, double* rays6) {
  Scene* s = (Scene*)h;
  for (uint64_t i = 0; i < n; i++) {
    Ray r = s->camera.ray_for_pixel(xy[i * 2], xy[i * 2 + 1]);
    double* o = rays6 + i * 6;
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
  }
  return 0;
}

// shape::normal_at (shape.rs:1478-1484) of the depth-first leaf `leaf` at a world-space point, with the hit's
// (u, v) for smooth triangles.  Returns 0, or -1 when there is no such leaf.
static const Shape* find_leaf(const Shape& s, int leaf) {
  if (s.ty == ShapeType::Group || s.is_csg()) {
    for (const Shape& c : s.children)
      if (const Shape* f = find_leaf(c, leaf)) return f;
    return nullptr;
  }
  return s.leaf_id == leaf ? &s : nullptr;
}
int orc_leaf_normal_at(void* h, int leaf, const double* point3, double u, double v, double* out3, int* type_out) {
  Scene* s = (Scene*)h;
  for (const Shape& o : s->world.objects)
    if (const Shape* f = find_leaf(o, leaf)) {
      Intersection hit{};
      hit.t = 1.0;
      hit.what = f;
      hit.has_uv = true;
      hit.u = u;
      hit.v = v;
      Tuple n = normal_at(*f, point(point3[0], point3[1], point3[2]), hit);
      out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
      if (type_out) *type_out = (int)f->ty;
      return 0;
    }
  return -1;
}

void orc_quantise(uint64_t n, const double* c, uint8_t* out) {
  for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)quantise(c[i]);
}

// Returns the encoded size; writes at most cap bytes.
uint64_t orc_ppm_encode(uint64_t w, uint64_t h, const double* rgb, char* buf, uint64_t cap) {
  std::string s = ppm_encode(w, h, rgb);
  if (buf && cap) memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
  return s.size();
}

}  // extern "C"
