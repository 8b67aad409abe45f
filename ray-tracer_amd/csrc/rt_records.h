// Ray records (one generation = struct of arrays in HBM), camera, image and statistics layout shared
// by the kernels and the host pipeline.
#pragma once
#include "rt_math.h"

namespace rt {

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
  double origin[3];  // inv * point(0, 0, 0), the same for every pixel (src/camera.rs:91)
  // A part of a frame is made of tiles 32 pixels wide and tile_h = 8, 16 or 32 high (2^tile_shift pixels each)
  uint32_t tile_h, tile_shift;
};

// The frames of one launch of the camera-ray kernel beyond its first (rt_render_frames): frame k > 0 is seen through
// cam[k - 1] and written to rgb8[k - 1]; the canvas, the part and its tiles are those of the first.  A part of a
// frame is a short launch that ends with its slowest wave; several frames' worth of it fill the GPU as a whole
// frame does.
constexpr int MAX_LAUNCH_FRAMES = 8;
struct FrameBatch {
  uint32_t frames;           // 0 or 1: one frame
  uint32_t waves_per_frame;  // the grid is frames x this many workgroups
  uint8_t* rgb8[MAX_LAUNCH_FRAMES - 1];
  CameraDev cam[MAX_LAUNCH_FRAMES - 1];
};

struct ImageOut {
  uint8_t* rgb8;
  double* rgb64;
  int32_t* hit_leaf;
  uint32_t compact;  // rgb8 holds this part's pixels in ray order (3 bytes at 3 i) instead of at y * width + x
};

// Shadow queries a generation kernel sets aside for a kernel of their own (k_shadow): World::is_shadowed
// (src/world.rs:123-137) needs the point, and its answer picks between two colours lighting() can return
// (src/light.rs:135-173); the generation kernel stored the lit one, the other travels with the query.
struct ShadowView {
  double *px, *py, *pz;  // over_point
  double *ar, *ag, *ab;  // lighting()'s result in shadow: the ambient term
  uint32_t* target;      // the ray's record (camera rays: the ray index, which names the pixel)
  uint32_t* counts;      // QUEUE_SEGMENTS counters, zero before the generation runs
  uint32_t segcap;       // entries per segment (0: no such queue)
};

// Arguments of one generation launch (host side).
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
  // where the shadow queries of hits on a cone's extension go (segcap 0: nowhere, they are walked in place)
  ShadowView aside{};
  FrameBatch batch{};  // mode 0 only
};

constexpr int TILE = 32;  // multi-GPU partition granule: pixels per side of a tile, rows of a band unless the caller asks for 16 or 8
constexpr int BLOCK = 64;  // threads per workgroup of every kernel: one wave
// Thread index within the workgroup: the lane id, which costs two instructions to recompute wherever
// it is needed instead of occupying a register for the whole kernel.
DEV uint32_t thread_in_block() { return (uint32_t)__lane_id(); }

// Device statistics block (unsigned long long): [0..3] closest-hit traversal work,
// [4..7] shadow traversal work, then three ray counters spread over 64 slots each
// (shaded hits = shadow rays, reflection rays, refraction rays) so that the per-wave
// adds do not all land on one address.
constexpr int STAT_SLOTS = 64;
constexpr int QUEUE_SEGMENTS = 64;
// Queue counters of one frame for a recursion `depth` deep (generations 0 .. depth, one block of QUEUE_SEGMENTS
// words per generation and one to spare): the ray queues', then as many for the shadow queries set aside.
constexpr size_t count_words(int depth) { return (size_t)(depth + 2) * QUEUE_SEGMENTS; }
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
  uint32_t tile = tiles ? tiles[i >> cam.tile_shift] : i >> cam.tile_shift;  // no table: the part owns every tile, in order
  uint32_t in_tile = i & ((1u << cam.tile_shift) - 1u);
  uint32_t wave = in_tile >> 6, lane = in_tile & 63u;
  x = (tile % tiles_x) * TILE + (wave & 3u) * 8u + (lane & 7u);
  y = (tile / tiles_x) * cam.tile_h + (wave >> 2) * 8u + (lane >> 3);
  return x < cam.hsize && y < cam.vsize;
}

// Camera::ray_for_pixel (src/camera.rs:76-95) with the inverse precomputed.
DEV Ray camera_ray(const CameraDev& cam, uint32_t px, uint32_t py) {
  double xoffset = ((double)px + 0.5) * cam.pixel_size;
  double yoffset = ((double)py + 0.5) * cam.pixel_size;
  double world_x = cam.half_width - xoffset;
  double world_y = cam.half_height - yoffset;
  V3 pixel = mul_point(cam.inv, v3(world_x, world_y, -1.0));
  V3 origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
  V3 direction = normalize(pixel - origin);
  return Ray{origin, direction};
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

}  // namespace rt
