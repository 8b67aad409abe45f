// C entry points over the host side (rt_host.hpp) for language bindings: load a
// scene description, flatten it, create the device scene, render.  Every call
// returns 0 or a negative rt_status; rth_last_error() holds the message.
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_host.hpp"

using namespace rthost;

struct rth_scene {
  Scene scene;
  FlatWorld flat;
  rt_scene* dev = nullptr;
};

static thread_local std::string g_err;

template <class F>
static int guarded(F&& f) {
  try {
    return f();
  } catch (const Panic& p) {
    g_err = p.what();
    return RT_ERR_INVALID;
  } catch (const std::invalid_argument& e) {
    g_err = e.what();
    return RT_ERR_UNSUPPORTED;
  } catch (const std::exception& e) {
    g_err = e.what();
    return RT_ERR_INVALID;
  }
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// device < 0: parse and flatten only (no GPU needed).
int rth_scene_from_json(const char* text, const char* base_dir, int device, rth_scene** out) {
  *out = nullptr;
  return guarded([&]() -> int {
    std::unique_ptr<rth_scene> h(new rth_scene());
    h->scene = scene_from_json(text, base_dir ? base_dir : "");
    h->flat = flatten(h->scene.world);
    if (device >= 0) {
      rt_scene_desc d = h->flat.desc();
      int rc = rt_scene_create(&d, device, &h->dev);
      if (rc != RT_OK) {
        g_err = rt_last_error();
        return rc;
      }
    }
    *out = h.release();
    return RT_OK;
  });
}

void rth_scene_free(rth_scene* h) {
  if (!h) return;
  rt_scene_destroy(h->dev);
  delete h;
}

rt_scene* rth_scene_device(rth_scene* h) { return h->dev; }

// out[0..7] = width, height, leaves, gates, xforms, triangles, materials, patterns
void rth_scene_info(rth_scene* h, uint64_t* out) {
  out[0] = h->scene.camera.hsize;
  out[1] = h->scene.camera.vsize;
  out[2] = h->flat.leaves.size();
  out[3] = h->flat.gates.size();
  out[4] = h->flat.xforms.size();
  out[5] = h->flat.triangles.size();
  out[6] = h->flat.materials.size();
  out[7] = h->flat.patterns.size();
}

const rt_scene_desc* rth_scene_desc(rth_scene* h) {
  static thread_local rt_scene_desc d;
  d = h->flat.desc();
  return &d;
}

int rth_scene_camera(rth_scene* h, rt_camera* out) {
  return guarded([&]() -> int {
    *out = h->scene.camera.describe();
    return RT_OK;
  });
}

// Camera::new again with the same view and field of view.
void rth_scene_set_canvas(rth_scene* h, uint64_t w, uint64_t hh) {
  h->scene.camera = Camera::make(w, hh, h->scene.camera.field_of_view, h->scene.camera.transform);
}

void rth_scene_set_camera(rth_scene* h, uint64_t w, uint64_t hh, double fov, const double* from, const double* to,
                          const double* up4) {
  h->scene.camera = Camera::make(w, hh, fov, view_transform(from, to, up4));
}

// Shape::set_transform (src/shape.rs:592-656) on the shape reached from world.objects by `path`
// (path[0] indexes the objects, every further entry the children of a group or CSG operator).
int rth_scene_set_transform(rth_scene* h, const uint64_t* path, uint64_t depth, const double* m16) {
  return guarded([&]() -> int {
    if (depth == 0) throw Panic("empty shape path");
    std::vector<Shape>* level = &h->scene.world.objects;
    Shape* shape = nullptr;
    for (uint64_t k = 0; k < depth; k++) {
      if (path[k] >= level->size()) throw Panic("shape path leaves the tree");
      shape = &(*level)[path[k]];
      level = &shape->children;
    }
    memcpy(shape->transform.m, m16, sizeof(shape->transform.m));
    return RT_OK;
  });
}

// The world as it is now, on the device: flattened again, then rt_scene_update (new boxes for the same
// hierarchy, no rebuild).  RT_ERR_UNSUPPORTED when the change needs a rebuild.
int rth_scene_update(rth_scene* h) {
  return guarded([&]() -> int {
    if (!h->dev) throw Panic("scene was loaded without a device");
    h->flat = flatten(h->scene.world);
    rt_scene_desc d = h->flat.desc();
    int rc = rt_scene_update(h->dev, &d);
    if (rc != RT_OK) g_err = rt_last_error();
    return rc;
  });
}

// parallel_render: jobs GPUs, writes the PPM.
int rth_parallel_render(rth_scene* h, uint64_t jobs, const char* out_path) {
  return guarded([&]() -> int {
    parallel_render(h->scene.world, h->scene.camera, jobs, out_path);
    return RT_OK;
  });
}

// ---- Canvas (src/canvas.rs:22-140)
struct rth_canvas {
  Canvas canvas;
};

int rth_canvas_new(uint64_t w, uint64_t hh, rth_canvas** out) {
  *out = nullptr;
  return guarded([&]() -> int {
    *out = new rth_canvas{Canvas(w, hh)};
    return RT_OK;
  });
}

// The frame parallel_render would save, as a canvas.
int rth_render_canvas(rth_scene* h, uint64_t jobs, rth_canvas** out) {
  *out = nullptr;
  return guarded([&]() -> int {
    *out = new rth_canvas{render_canvas(h->scene.world, h->scene.camera, jobs)};
    return RT_OK;
  });
}

void rth_canvas_free(rth_canvas* c) { delete c; }

void rth_canvas_size(const rth_canvas* c, uint64_t* w, uint64_t* hh) {
  *w = c->canvas.width;
  *hh = c->canvas.height;
}

void rth_canvas_write_pixel(rth_canvas* c, uint64_t x, uint64_t y, const double* rgb) {
  c->canvas.write_pixel(x, y, Color{rgb[0], rgb[1], rgb[2]});
}

// 1 and the colour, or 0 for a pixel outside the canvas (the reference returns None).
int rth_canvas_read_pixel(const rth_canvas* c, uint64_t x, uint64_t y, double* rgb) {
  std::optional<Color> p = c->canvas.read_pixel(x, y);
  if (!p) return 0;
  rgb[0] = p->r; rgb[1] = p->g; rgb[2] = p->b;
  return 1;
}

// All pixels, row-major, three doubles each.
const double* rth_canvas_pixels(const rth_canvas* c) { return c->canvas.pixels.empty() ? nullptr : &c->canvas.pixels[0].r; }

int rth_canvas_save(const rth_canvas* c, const char* path) {
  return guarded([&]() -> int {
    c->canvas.save(path);
    return RT_OK;
  });
}

// ---- FrameLoop: one worker and one copy of the scene per part, frames gathered in a shared host ring
struct rth_frame_loop {
  std::unique_ptr<FrameLoop> loop;
};

int rth_frame_loop_new(rth_scene* h, uint64_t parts, const int32_t* devices, int32_t depth, rth_frame_loop** out) {
  *out = nullptr;
  return guarded([&]() -> int {
    std::vector<int> dev(devices, devices + parts);
    std::unique_ptr<rth_frame_loop> l(new rth_frame_loop());
    l->loop.reset(new FrameLoop(h->scene.world, h->scene.camera, dev, depth));
    *out = l.release();
    return RT_OK;
  });
}

// One process per GPU: this process renders parts first_part .. first_part + parts - 1 of total_parts into
// `ring` (depth frames back to back, registered by the caller with rt_host_register, shared by the processes).
int rth_frame_loop_new_shared(rth_scene* h, uint64_t parts, const int32_t* devices, int32_t depth, uint64_t first_part,
                              uint64_t total_parts, uint8_t* ring, rth_frame_loop** out) {
  *out = nullptr;
  return guarded([&]() -> int {
    std::vector<int> dev(devices, devices + parts);
    FrameLoop::Share share;
    share.first_part = first_part;
    share.total_parts = total_parts;
    share.ring = ring;
    std::unique_ptr<rth_frame_loop> l(new rth_frame_loop());
    l->loop.reset(new FrameLoop(h->scene.world, h->scene.camera, dev, depth, share));
    *out = l.release();
    return RT_OK;
  });
}

void rth_frame_loop_free(rth_frame_loop* l) { delete l; }

// stats: one rt_stats per part.  *seconds: wall time of the timed frames.
int rth_frame_loop_run(rth_frame_loop* l, uint64_t warmup, uint64_t frames, int32_t to_host, double* seconds, rt_stats* stats) {
  return guarded([&]() -> int {
    FrameLoop::Result r = l->loop->run(warmup, frames, to_host != 0);
    if (seconds) *seconds = r.seconds;
    if (stats)
      for (size_t k = 0; k < r.part.size(); k++) stats[k] = r.part[k];
    return RT_OK;
  });
}

int rth_frame_loop_blocking(rth_frame_loop* l, uint64_t frames, int32_t counters, rt_stats* stats) {
  return guarded([&]() -> int {
    std::vector<rt_stats> r = l->loop->blocking_frames(frames, counters != 0);
    for (size_t k = 0; k < r.size(); k++) stats[k] = r[k];
    return RT_OK;
  });
}

const uint8_t* rth_frame_loop_host_frame(rth_frame_loop* l, int32_t slot) { return l->loop->host_frame(slot); }

// out[0] = scene bytes per GPU, out[1] = frame bytes; *build_seconds: the one build of the acceleration structure
void rth_frame_loop_info(rth_frame_loop* l, uint64_t* out, double* build_seconds) {
  out[0] = l->loop->scene_bytes();
  out[1] = l->loop->frame_bytes();
  if (build_seconds) *build_seconds = l->loop->build_seconds();
}

// rows of a band of this loop's parts (32 unless RT_AMD_BAND_ROWS asked for 16 or 8 when the loop was made)
uint64_t rth_frame_loop_band_rows(rth_frame_loop* l) { return l->loop->band_rows(); }
uint64_t rth_frame_loop_frames_per_launch(rth_frame_loop* l) { return l->loop->frames_per_launch(); }

uint64_t rth_ppm_encode(uint64_t w, uint64_t hh, const uint8_t* rgb8, char* buf, uint64_t cap) {
  std::string s = ppm_encode(w, hh, rgb8);
  if (buf && cap) memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
  return s.size();
}

}  // extern "C"
