// The scene object behind the C ABI (struct rt_scene of include/rt_amd.h) and what the host-side
// translation units of librt_amd.so share.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_launch.h"
#include "rt_types.h"

namespace rt {

struct GenBuf {
  void* slab = nullptr;
  size_t capacity = 0;
  GenView view;
};

constexpr int N_DOUBLE_ARRAYS = 13;
constexpr int N_U32_ARRAYS = 4;

inline size_t gen_bytes(size_t cap) {
  size_t c = (cap + 63) & ~(size_t)63;
  return c * (N_DOUBLE_ARRAYS * sizeof(double) + N_U32_ARRAYS * sizeof(uint32_t));
}

inline void carve(GenBuf& b) {
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

// Room for the shadow queries of one generation (ShadowView): struct of arrays in one slab.
struct ShadowBuf {
  void* slab = nullptr;
  size_t capacity = 0;  // entries
  ShadowView view{};
};
inline size_t shadow_bytes(size_t cap) {
  const size_t c = (cap + 63) & ~(size_t)63;
  return c * (6 * sizeof(double) + sizeof(uint32_t));
}
inline void carve(ShadowBuf& b) {
  const size_t c = (b.capacity + 63) & ~(size_t)63;
  double* d = (double*)b.slab;
  ShadowView& v = b.view;
  v.px = d; v.py = d + c; v.pz = d + 2 * c;
  v.ar = d + 3 * c; v.ag = d + 4 * c; v.ab = d + 5 * c;
  v.target = (uint32_t*)(d + 6 * c);
}

// What one frame (or one batch of caller rays) asks of the pipeline.
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

}  // namespace rt

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
  static constexpr int COPY_FENCES = 16;
  hipEvent_t copy_fence[COPY_FENCES] = {};  // rt_copy_fence / rt_copy_wait
  bool copy_fence_set[COPY_FENCES] = {};
  rt::SceneDev dev{};
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
  uint32_t build_flags = 0;  // of rt_scene_create_with: an update that has to build again builds the same way
  uint64_t rebuilds = 0;     // updates that built the acceleration structure again instead of refitting it
  // A failed rt_scene_update may have replaced some of the device tables and not others: such a scene refuses to
  // render until an update succeeds.
  bool broken = false;
  // queue lengths per generation of the last multi-generation frame: the next frame is launched on them
  // without asking the device after every generation
  std::vector<uint32_t> hint;                // [g]: the fullest queue segment of generation g (depth + 3 entries)
  uint32_t hint_key[2] = {0, 0};             // n0 and remaining0 the hints belong to
  bool hint_valid = false;
  uint32_t holdoff = 0, holdoff_next = 1;  // frames to render exactly after a speculative frame did not fit
  bool uniform_nodes = false;  // large hierarchy: kernels with the scalar path for nodes a whole wave visits
  bool tri_prefetch = false;   // large mesh: kernels that fetch the next triangle of a run while one is tested
  int max_children = 0;  // rays a hit can spawn: reflective? + transparent?
  // What depends on the recursion depth (rt_color_at takes any `remaining`, src/world.rs:249): the generations'
  // record buffers (depth + 2), their queue counters, the events around their kernels.  Made for the reference's
  // depth of 5 (src/consts.rs:10) and grown by ensure_depth() when a caller asks for more.
  int depth = 0;
  std::vector<rt::GenBuf> gens;
  // Shadow queries of hits on a cone's extension, walked by a kernel of their own (scenes with bounded cones)
  bool set_aside = false;
  rt::ShadowBuf aside_buf;
  void* csg_spill_buf = nullptr;  // SceneDev::csg_spill: CSG hit lists too long for the LDS
  uint32_t* d_counts = nullptr;           // [gen][QUEUE_SEGMENTS]: rays queued for the next generation, per segment
  unsigned long long* d_stats = nullptr;  // STAT_WORDS counters
  uint32_t* h_counts = nullptr;           // pinned mirrors
  unsigned long long* h_stats = nullptr;
  uint32_t* d_tiles = nullptr;
  size_t tiles_capacity = 0;
  std::vector<uint32_t> h_tiles;
  uint32_t tiles_key[4] = {0, 0, 0, 0};  // hsize, vsize, part, n_parts of the uploaded tile table
  std::vector<hipEvent_t> ev;  // two per generation
  // frames enqueued under RT_FLAG_ASYNC and not collected yet: a small ring, each slot with its own
  // statistics block so consecutive frames do not share counters
  struct FrameSlot {
    hipEvent_t begin = nullptr, end = nullptr;
    bool pending = false;
    uint64_t primary = 0;
    uint32_t frames = 1;  // frames the launch rendered (rt_render_frames)
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
  // Frames of a scene that spawns rays, enqueued under RT_FLAG_ASYNC with every generation launched at once on the
  // queue lengths of the last verified frame.  Whether such a frame fitted its queues is known when its counters
  // are back; until then it is unverified, and one that did not fit is rendered again, exactly, into the same
  // buffers before anything reads them (rt::verify_frames: downloads of its images, rt_scene_wait, the next
  // blocking call).  A short queue, oldest first.
  struct SpecFrame {
    rt::PipelineIn in;
    int max_gen = 0;
    std::vector<uint32_t> segcap;  // record slots per queue segment the generations were launched with
    uint64_t primary = 0;
    hipEvent_t begin = nullptr, end = nullptr, done = nullptr;  // around the generation kernels; after the counters' copies
    uint32_t* h_counts = nullptr;  // pinned: this frame's queue counters and statistics
    unsigned long long* h_stats = nullptr;
    int counts_depth = -1;         // recursion depth h_counts was sized for
  };
  static constexpr int SPEC_SLOTS = 4;
  SpecFrame spec[SPEC_SLOTS];
  int spec_first = 0, spec_count = 0;
  // Such frames alternate between two sets of what a frame in flight owns on the device -- ray records, queue
  // counters, statistics, the stream -- so that the small late generations of one frame (a wave's life each, the
  // GPU far from full) run beside the first generation of the next.  The set not in use waits here; the fields
  // above (stream, gens, aside_buf, d_counts, d_stats) are the set in use (rt::swap_lane).  Made on first use.
  struct Lane {
    hipStream_t stream = nullptr;
    std::vector<rt::GenBuf> gens;
    rt::ShadowBuf aside_buf;
    uint32_t* d_counts = nullptr;
    unsigned long long* d_stats = nullptr;
    int depth = -1;  // recursion depth d_counts and gens were sized for
  };
  static constexpr int MAX_SPARE_LANES = 3;
  Lane spare_lane[MAX_SPARE_LANES];
  int next_spare = 0;  // the spare set the next enqueued frame takes
  // staging for the host-buffer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // What rt_scene_update needs again: the records in device order, the structure of the description
  // they were made from, and the device arrays of the refit.
  struct Host {
    std::vector<rt::TriIsect> tris;
    std::vector<rt::QuadPrim> quads, unbounded, band_prims;
    std::vector<rt::Gate> gates;
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

namespace rt {

// The scene's main stream for an enqueue; the copy stream is ordered after it at the next download.
inline hipStream_t main_stream(rt_scene* s) {
  s->stream_used = true;
  return s->stream;
}

// Stores msg as the calling thread's rt_last_error() text and returns code.
int fail(int code, const std::string& msg);
std::string& last_error_text();

// every entry point that traces rays through a scene starts with this
#define RT_REQUIRE_USABLE(s) do { if ((s)->broken) return rt::fail(RT_ERR_INVALID, "the scene is unusable: an rt_scene_update failed half way; update it again or destroy it"); } while (0)

// rt_scene_build.cpp
int attach_band_grid(rt_scene* s, size_t n_bands);
void world_normal(const double* full_inv, const double* local_normal, double* out);  // of a plane or flat triangle
// rt_pipeline.cpp
int collect_all(rt_scene* s);  // waits for the frames in flight (verifying those that need it), then reads and clears the counters they added to
// Verifies the unverified frames, oldest first: all of them, or (touching != nullptr) those up to the last one whose
// image buffers hold that device address; *was_frame tells whether one did.
int verify_frames(rt_scene* s, const void* touching, bool* was_frame);
int ensure_gen(rt_scene* s, int g, size_t cap);
// `want` new streams of the current device, chosen so that they share a hardware queue neither with one another nor
// with the `avoid` streams (which must be idle), as far as the process has queues for that (rt_scene_build.cpp).
int pick_streams(const hipStream_t* avoid, int n_avoid, int want, hipStream_t* out);
int ensure_depth(rt_scene* s, int depth);  // buffers, counters and events for a recursion that deep
int ensure_aside(rt_scene* s, uint32_t rays);  // room for the shadow queries a generation of that many rays may set aside
int stage_reserve(rt_scene* s, size_t bytes);
bool affine(const double* m16);  // last row 0 0 0 1
inline void copy3x4(double* dst, const double* src16) { memcpy(dst, src16, 12 * sizeof(double)); }
// dynamic LDS of a traversal kernel's workgroup: the per-lane stack, then (scenes with CSG operators) the hit list
inline unsigned traversal_lds_bytes(const rt_scene* s) {
  return (unsigned)s->dev.stack_entries * BLOCK * (unsigned)sizeof(uint32_t) +
         (s->has_csg && !s->dev.csg_spill ? (unsigned)s->dev.csg_hits * 3u * BLOCK * (unsigned)sizeof(uint32_t) : 0u);
}

CameraDev camera_dev(const rt_camera* c);
// Runs the generations, resolves colours back into generation 0 and, for camera rays, writes the image.
int run_pipeline(rt_scene* s, const PipelineIn& in, rt_stats* stats);
// One camera frame of a scene that spawns no rays: a single kernel, only enqueued.
int enqueue_direct_frame(rt_scene* s, const PipelineIn& in, uint64_t primary);
int enqueue_direct_frames(rt_scene* s, const PipelineIn& in, uint64_t primary, int n_frames, const CameraDev* more_cams,
                          uint8_t* const* more_rgb8);
// One camera frame of a scene that spawns rays, only enqueued: possible when the last frame's queue lengths are
// known for this job (*enqueued; otherwise the caller renders it with run_pipeline).
int enqueue_speculative_frame(rt_scene* s, const PipelineIn& in, uint64_t primary, bool* enqueued);

}  // namespace rt

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* reported here: a later launch check must not see it again */ \
      return rt::fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    }                                                                                       \
  } while (0)
