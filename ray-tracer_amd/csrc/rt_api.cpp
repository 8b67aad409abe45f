// The C ABI of include/rt_amd.h: rendering, downloads, the per-ray entry points, memory helpers and
// error text.  Scene construction is in rt_scene_build.cpp, the frame pipeline in rt_pipeline.cpp.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "rt_error.h"
#include "rt_scene.h"

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
// error text
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;
std::string& rt::last_error_text() { return g_last_error; }
int rt::fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int rt_fail_message(int code, const char* msg) { return fail(code, msg ? msg : ""); }

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }

extern "C" int rt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
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
  // the fill returns before it is done, and streams that do not wait for the null stream could write the memory first
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
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

// rows of a band under RT_FLAG_BANDS: 32, or 16 / 8 when the caller asks for finer bands
static uint32_t band_rows(uint32_t flags) { return (flags & RT_FLAG_BANDS_8) ? 8u : ((flags & RT_FLAG_BANDS_16) ? 16u : 32u); }

// What rt_render and rt_render_frames share: the arguments checked, the part's tile table on the device, the
// pipeline's input for this camera, and how many of the part's pixels lie inside the image.
static int prepare_camera_frame(rt_scene* s, const rt_camera* camera, int part, int n_parts, uint32_t flags, uint8_t* rgb8_dev,
                                double* rgb64_dev, int32_t* hit_leaf_dev, PipelineIn& in, uint64_t& valid) {
  if (n_parts < 1 || part < 0 || part >= n_parts) return fail(RT_ERR_INVALID, "part must lie in [0, n_parts)");
  if (camera->hsize == 0 || camera->vsize == 0) return fail(RT_ERR_INVALID, "empty canvas");
  if (!affine(camera->inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) camera transform");
  RT_REQUIRE_USABLE(s);
  HIP_TRY(hipSetDevice(s->device));
  const bool bands = (flags & RT_FLAG_BANDS) != 0;
  const uint32_t tile_h = bands ? band_rows(flags) : (uint32_t)TILE;
  uint32_t tiles_x = (camera->hsize + TILE - 1) / TILE, tiles_y = (camera->vsize + tile_h - 1) / tile_h;
  const uint32_t key[4] = {camera->hsize, camera->vsize, (uint32_t)part | (bands ? 0x80000000u : 0u) | (tile_h << 20), (uint32_t)n_parts};
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
  uint64_t n64 = (uint64_t)s->h_tiles.size() * TILE * tile_h;
  if (n64 > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "canvas part exceeds 2^32 rays");
  in.n0 = (uint32_t)n64;
  in.flags = flags;
  in.camera = true;
  in.cam = camera_dev(camera);
  in.cam.tile_h = tile_h;
  in.cam.tile_shift = tile_h == 8 ? 8u : (tile_h == 16 ? 9u : 10u);
  in.tiles = n_parts == 1 ? nullptr : s->d_tiles;  // the whole image: tile k is tile k, no look-up at the start of every wave
  in.tiles_x = tiles_x;
  if ((flags & RT_FLAG_COMPACT) && (rgb64_dev || hit_leaf_dev || !rgb8_dev))
    return fail(RT_ERR_INVALID, "the compact layout is for rgb8 only");
  in.img = ImageOut{rgb8_dev, rgb64_dev, hit_leaf_dev, (flags & RT_FLAG_COMPACT) ? 1u : 0u};
  valid = 0;
  for (uint32_t t : s->h_tiles) {
    uint32_t x0 = (t % tiles_x) * TILE, y0 = (t / tiles_x) * tile_h;
    uint32_t w = std::min<uint32_t>(TILE, camera->hsize - x0), h = std::min<uint32_t>(tile_h, camera->vsize - y0);
    valid += (uint64_t)w * h;
  }
  return RT_OK;
}

extern "C" int rt_render(rt_scene* s, const rt_camera* camera, int part, int n_parts, uint32_t flags,
                         uint8_t* rgb8_dev, double* rgb64_dev, int32_t* hit_leaf_dev, rt_stats* stats) {
  if (!s || !camera) return fail(RT_ERR_INVALID, "null argument");
  auto w0 = std::chrono::steady_clock::now();
  if (stats) memset(stats, 0, sizeof(*stats));
  PipelineIn in;
  uint64_t valid = 0;
  int rc = prepare_camera_frame(s, camera, part, n_parts, flags, rgb8_dev, rgb64_dev, hit_leaf_dev, in, valid);
  if (rc) return rc;
  if (in.n0 == 0) return RT_OK;
  if ((flags & RT_FLAG_ASYNC) && s->max_children == 0) return enqueue_direct_frame(s, in, valid);
  if (flags & RT_FLAG_ASYNC) {  // a scene that spawns rays: every generation at once, verified later, when it can be
    bool enqueued = false;
    int rce = enqueue_speculative_frame(s, in, valid, &enqueued);
    if (rce) return rce;
    if (enqueued) {
      s->last_stream = s->stream;
      return RT_OK;
    }
  }
  rc = collect_all(s);  // frames still in flight finish first
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

extern "C" int rt_render_frames(rt_scene* s, const rt_camera* cameras, int n_frames, int part, int n_parts, uint32_t flags,
                                uint8_t* const* rgb8_dev) {
  if (!s || !cameras || !rgb8_dev) return fail(RT_ERR_INVALID, "null argument");
  if (n_frames < 1 || n_frames > RT_MAX_FRAMES_PER_CALL) return fail(RT_ERR_INVALID, "n_frames must lie in 1..RT_MAX_FRAMES_PER_CALL");
  if (!(flags & RT_FLAG_ASYNC)) return fail(RT_ERR_INVALID, "rt_render_frames enqueues its frames: pass RT_FLAG_ASYNC");
  for (int k = 0; k < n_frames; k++) {
    if (!rgb8_dev[k]) return fail(RT_ERR_INVALID, "null frame");
    if (cameras[k].hsize != cameras[0].hsize || cameras[k].vsize != cameras[0].vsize)
      return fail(RT_ERR_INVALID, "the frames of one call share one canvas size");
    if (!affine(cameras[k].inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) camera transform");
  }
  static_assert(RT_MAX_FRAMES_PER_CALL == MAX_LAUNCH_FRAMES, "the header's limit is the kernel's");
  // One launch where a frame is one kernel: a scene that spawns no rays, whose CSG hit lists (if any) live in LDS.
  // Every other scene gets its frames one after the other, as from rt_render.
  const bool one_launch = n_frames > 1 && s->max_children == 0 && !s->dev.csg_spill;
  if (!one_launch) {
    for (int k = 0; k < n_frames; k++) {
      int rc = rt_render(s, &cameras[k], part, n_parts, flags, rgb8_dev[k], nullptr, nullptr, nullptr);
      if (rc) return rc;
    }
    return RT_OK;
  }
  PipelineIn in;
  uint64_t valid = 0;
  int rc = prepare_camera_frame(s, &cameras[0], part, n_parts, flags, rgb8_dev[0], nullptr, nullptr, in, valid);
  if (rc) return rc;
  if (in.n0 == 0) return RT_OK;
  CameraDev more[MAX_LAUNCH_FRAMES - 1];
  for (int k = 1; k < n_frames; k++) {
    more[k - 1] = camera_dev(&cameras[k]);
    more[k - 1].tile_h = in.cam.tile_h;
    more[k - 1].tile_shift = in.cam.tile_shift;
  }
  return enqueue_direct_frames(s, in, valid, n_frames, more, rgb8_dev + 1);
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

// The first device-to-host copies of a process are slow in a way that does not show at once: the runtime sets up
// every DMA engine it hands copies to on that engine's first use (measured here: about 5.6 ms each, three of them),
// and it picks a second and a third engine only when a copy finds the first ones busy -- which in a frame loop
// happens some time within the first hundred frames (tools/probes/stall_probe.py: the loop without copies shows no
// such stall).  Before a device's first download the engines are therefore made to start: several copies on streams
// of their own, in flight together.
static int prime_copy_engines(int device) {
  static std::mutex guard;
  static bool done[64] = {};
  std::lock_guard<std::mutex> lock(guard);
  if (device < 0 || device >= 64 || done[device]) return RT_OK;
  done[device] = true;
  static const bool off = getenv("RT_AMD_NO_COPY_PRIMING") != nullptr;
  if (off) return RT_OK;
  constexpr int COPIES = 8, ROUNDS = 3;
  constexpr size_t BYTES = 4u << 20;
  void* dev = nullptr;
  char* host = nullptr;
  hipStream_t streams[COPIES] = {};
  int rc = RT_OK;
  auto check = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == RT_OK) {
      (void)hipGetLastError();
      rc = fail(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    }
  };
  check(hipMalloc(&dev, BYTES), "hipMalloc");
  check(hipHostMalloc((void**)&host, COPIES * BYTES, hipHostMallocDefault), "hipHostMalloc");
  for (hipStream_t& st : streams)
    if (rc == RT_OK) check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  for (int r = 0; r < ROUNDS && rc == RT_OK; r++) {
    for (int k = 0; k < COPIES && rc == RT_OK; k++)
      check(hipMemcpyAsync(host + (size_t)k * BYTES, dev, BYTES, hipMemcpyDeviceToHost, streams[k]), "hipMemcpyAsync");
    for (int k = 0; k < COPIES && rc == RT_OK; k++) check(hipStreamSynchronize(streams[k]), "hipStreamSynchronize");
  }
  for (hipStream_t st : streams)
    if (st) (void)hipStreamDestroy(st);
  if (host) (void)hipHostFree(host);
  if (dev) (void)hipFree(dev);
  return rc;
}

extern "C" int rt_download_2d(rt_scene* s, const void* src_dev, void* dst_host, uint64_t row_bytes, uint64_t rows,
                              uint64_t pitch, void* after_stream) {
  if (!s || (row_bytes && rows && (!src_dev || !dst_host))) return fail(RT_ERR_INVALID, "null argument");
  if (pitch < row_bytes) return fail(RT_ERR_INVALID, "pitch smaller than a row");
  HIP_TRY(hipSetDevice(s->device));
  if (row_bytes == 0 || rows == 0) return RT_OK;
  int rcp = prime_copy_engines(s->device);
  if (rcp) return rcp;
  bool finished = false;  // the source is the image of a frame whose end the host has just seen: nothing to order after
  int rcv = verify_frames(s, src_dev, &finished);
  if (rcv) return rcv;
  int rco = finished ? RT_OK : order_copy_stream(s, after_stream);
  if (rco) return rco;
  HIP_TRY(hipMemcpy2DAsync(dst_host, pitch, src_dev, pitch, row_bytes, rows, hipMemcpyDeviceToHost, s->copy_stream));
  return RT_OK;
}

extern "C" int rt_download(rt_scene* s, const void* src_dev, void* dst_host, uint64_t bytes, void* after_stream) {
  if (!s || (bytes && (!src_dev || !dst_host))) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (bytes == 0) return RT_OK;
  int rcp = prime_copy_engines(s->device);
  if (rcp) return rcp;
  bool finished = false;
  int rcv = verify_frames(s, src_dev, &finished);
  if (rcv) return rcv;
  int rco = finished ? RT_OK : order_copy_stream(s, after_stream);
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
    // only this part's bands go to the host images, so the parts of one frame can share them
    const size_t rows_total = camera->vsize, w = camera->hsize, band = band_rows(flags);
    for (size_t y0 = (size_t)part * band; y0 < rows_total; y0 += (size_t)n_parts * band) {
      const size_t rows = std::min<size_t>(band, rows_total - y0), px = y0 * w, n = rows * w;
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
  launch_pixel_rays(main_stream(s), camera_dev(camera), (uint32_t)n, d_xy, d_rays);
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
  launch_load_rays(main_stream(s), s->gens[0].view, (uint32_t)n, (const double*)s->stage);
  return RT_OK;
}

extern "C" int rt_color_at(rt_scene* s, uint64_t n, const double* rays6, int remaining, uint32_t flags, double* rgb,
                           int32_t* leaf, rt_stats* stats) {
  if (!s || (n && (!rays6 || !rgb))) return fail(RT_ERR_INVALID, "null argument");
  if (remaining < 0) return fail(RT_ERR_INVALID, "remaining must be >= 0");
  // any depth, as World::color_at takes any usize (src/world.rs:249); every level keeps its rays' records until the
  // colours have flowed back, so the depth is bounded by what the host will hold tables for
  if (remaining > 4096) return fail(RT_ERR_UNSUPPORTED, "remaining above 4096");
  RT_REQUIRE_USABLE(s);
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
  launch_output_rays(main_stream(s), s->gens[0].view, (uint32_t)n, d_rgb, d_leaf, (double*)nullptr);
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
  RT_REQUIRE_USABLE(s);
  HIP_TRY(hipSetDevice(s->device));
  if (n == 0) return RT_OK;
  int rc = load_user_rays(s, n, rays6);
  if (rc) return rc;
  launch_intersect(main_stream(s), s->dev, s->has_csg, traversal_lds_bytes(s), s->gens[0].view, (uint32_t)n);
  double* d_t = (double*)s->stage;
  int32_t* d_leaf = (int32_t*)(d_t + n);
  launch_output_rays(main_stream(s), s->gens[0].view, (uint32_t)n, (double*)nullptr, d_leaf, d_t);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipMemcpyAsync(leaf, d_leaf, n * sizeof(int32_t), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return RT_OK;
}

extern "C" int rt_is_shadowed(rt_scene* s, uint64_t n, const double* points3, uint8_t* shadowed) {
  if (!s || (n && (!points3 || !shadowed))) return fail(RT_ERR_INVALID, "null argument");
  RT_REQUIRE_USABLE(s);
  HIP_TRY(hipSetDevice(s->device));
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFF00ull) return fail(RT_ERR_UNSUPPORTED, "too many points in one call");
  int rc = stage_reserve(s, n * 3 * sizeof(double) + n + 256);
  if (rc) return rc;
  double* d_pts = (double*)s->stage;
  uint8_t* d_out = (uint8_t*)(d_pts + n * 3);
  HIP_TRY(hipMemcpyAsync(d_pts, points3, n * 3 * sizeof(double), hipMemcpyHostToDevice, main_stream(s)));
  launch_points_shadowed(main_stream(s), s->dev, s->has_csg, traversal_lds_bytes(s), (uint32_t)n, d_pts, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(shadowed, d_out, n, hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return RT_OK;
}
