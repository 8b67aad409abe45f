// Host side of a frame: the generation kernels launched one after the other (or all at once on the last
// frame's queue lengths), the colours resolved back through the generations, asynchronous frames.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_scene.h"

using namespace rt;

// Releases what a set that is not in use holds on the device (its stream stays).
static int release_lane(rt_scene::Lane& o) {
  if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
  for (GenBuf& g : o.gens)
    if (g.slab) HIP_TRY(hipFree(g.slab));
  o.gens.clear();
  if (o.aside_buf.slab) HIP_TRY(hipFree(o.aside_buf.slab));
  o.aside_buf = ShadowBuf{};
  if (o.d_counts) HIP_TRY(hipFree(o.d_counts));
  if (o.d_stats) HIP_TRY(hipFree(o.d_stats));
  o.d_counts = nullptr;
  o.d_stats = nullptr;
  o.depth = -1;
  return RT_OK;
}

int rt::ensure_depth(rt_scene* s, int depth) {
  if (depth <= s->depth && !s->gens.empty()) return RT_OK;
  {
    for (rt_scene::Lane& o : s->spare_lane) {  // sized for the old depth; made again when frames are enqueued
      int rc = release_lane(o);
      if (rc) return rc;
    }
  }
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  if (s->d_counts) HIP_TRY(hipFree(s->d_counts));
  s->d_counts = nullptr;
  if (s->h_counts) HIP_TRY(hipHostFree(s->h_counts));
  s->h_counts = nullptr;
  HIP_TRY(hipMalloc((void**)&s->d_counts, 2 * count_words(depth) * sizeof(uint32_t)));
  HIP_TRY(hipHostMalloc((void**)&s->h_counts, count_words(depth) * sizeof(uint32_t), hipHostMallocDefault));
  s->gens.resize((size_t)depth + 2);
  s->hint.assign((size_t)depth + 3, 0u);
  s->hint_valid = false;
  const size_t had = s->ev.size();
  s->ev.resize(2 * ((size_t)depth + 1), nullptr);
  for (size_t k = had; k < s->ev.size(); k++) HIP_TRY(hipEventCreate(&s->ev[k]));
  s->depth = depth;
  return RT_OK;
}

int rt::ensure_gen(rt_scene* s, int g, size_t cap) {
  GenBuf& b = s->gens[g];
  if (cap == 0) cap = 64;
  if (b.capacity >= cap) return RT_OK;
  if (b.slab) {
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    HIP_TRY(hipFree(b.slab));
    b.slab = nullptr;
    b.capacity = 0;
  }
  HIP_TRY(hipMalloc(&b.slab, gen_bytes(cap)));
  b.capacity = cap;
  carve(b);
  return RT_OK;
}

// Entries per segment of the queue of shadow queries set aside by a generation of `rays` rays: an eighth of
// what its waves could queue (they are dealt to the segments round-robin, at most 64 queries each); a query
// that finds its segment full is walked in place.
static inline uint32_t aside_segcap(uint32_t rays) {
  const uint64_t waves = ((uint64_t)rays + BLOCK - 1) / BLOCK;
  const uint64_t all = (waves + QUEUE_SEGMENTS - 1) / QUEUE_SEGMENTS * BLOCK;
  return (uint32_t)((all / 8 + 63) & ~(uint64_t)63);
}

int rt::ensure_aside(rt_scene* s, uint32_t rays) {
  ShadowBuf& b = s->aside_buf;
  const size_t cap = (size_t)aside_segcap(rays) * QUEUE_SEGMENTS;
  if (b.capacity >= cap) return RT_OK;
  if (b.slab) {
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    HIP_TRY(hipFree(b.slab));
    b.slab = nullptr;
    b.capacity = 0;
  }
  HIP_TRY(hipMalloc(&b.slab, shadow_bytes(cap)));
  b.capacity = cap;
  carve(b);
  return RT_OK;
}

namespace {

// The kernel for this scene and generation: the instantiations live in rt_gen_*.hip.
void launch_generation(rt_scene* s, int mode, bool count, const GenLaunch& a) {
  static const unsigned env_lds = getenv("RT_AMD_EXTRA_LDS") ? (unsigned)atoi(getenv("RT_AMD_EXTRA_LDS")) : 0u;
  const unsigned lds = traversal_lds_bytes(s) + env_lds;
  const bool tripf = s->uniform_nodes && s->tri_prefetch && mode == 0;
  if (s->has_csg) {  // scenes with CSG operators or transform chains: the general kernels
    if (count) launch_generation_variant<true, true, false, false>(s->dev, lds, mode, a);
    else launch_generation_variant<false, true, false, false>(s->dev, lds, mode, a);
  } else if (count) launch_generation_variant<true, false, false, false>(s->dev, lds, mode, a);
  else if (tripf) launch_generation_variant<false, false, true, true>(s->dev, lds, mode, a);
  else if (s->uniform_nodes) launch_generation_variant<false, false, true, false>(s->dev, lds, mode, a);
  else launch_generation_variant<false, false, false, false>(s->dev, lds, mode, a);
}

}  // namespace

CameraDev rt::camera_dev(const rt_camera* c) {
  CameraDev cd;
  cd.hsize = c->hsize;
  cd.vsize = c->vsize;
  cd.half_width = c->half_width;
  cd.half_height = c->half_height;
  cd.pixel_size = c->pixel_size;
  copy3x4(cd.inv, c->inv);
  cd.tile_h = TILE;
  cd.tile_shift = 10;
  // Matrix4D * point(0, 0, 0) as the device would form it (src/matrix.rs:638-653): products with zero stay in
  for (int r = 0; r < 3; r++) {
    const double* m = cd.inv + 4 * r;
    cd.origin[r] = m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3];
  }
  return cd;
}

namespace rt {

static void decode_stats(const unsigned long long* h, rt_stats* stats) {
  unsigned long long sums[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < STAT_SLOTS; j++) sums[k] += h[STAT_RAYS + k * STAT_SLOTS + j];
  stats->shadow = sums[0];
  stats->reflect = sums[1];
  stats->refract = sums[2];
  stats->node_visits = h[0];
  stats->tri_tests = h[1];
  stats->quadric_tests = h[2];
  stats->gate_tests = h[3];
  stats->shadow_node_visits = h[4];
  stats->shadow_tri_tests = h[5];
  stats->shadow_quadric_tests = h[6];
  stats->shadow_gate_tests = h[7];
}

// Waits for one asynchronous frame and adds its kernel time to the scene's running total.
static int collect_frame(rt_scene* s, rt_scene::FrameSlot& f) {
  if (!f.pending) return RT_OK;
  HIP_TRY(hipEventSynchronize(f.end));
  f.pending = false;
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, f.begin, f.end));
  rt_stats& t = s->async_total;
  t.primary += f.primary;
  t.kernel_ms += ms;
  t.kernel_launches += 1;
  t.generations += f.frames;
  return RT_OK;
}

// Waits for every asynchronous frame, then reads and clears the counters they added to.
int collect_all(rt_scene* s) {
  {
    int rc = verify_frames(s, nullptr, nullptr);
    if (rc) return rc;
  }
  for (int k = 0; k < rt_scene::FRAME_SLOTS; k++) {
    int rc = collect_frame(s, s->slots[(s->next_slot + k) % rt_scene::FRAME_SLOTS]);
    if (rc) return rc;
  }
  if (!s->async_counters_dirty) return RT_OK;
  HIP_TRY(hipMemcpy(s->h_async_stats, s->d_async_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  // hipMemset of device memory returns before it is done, and the null stream it runs on does not hold back the
  // scene's streams: without the wait the fill could land after the next frame's kernel has added its counts
  HIP_TRY(hipMemset(s->d_async_stats, 0, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipStreamSynchronize(nullptr));
  s->async_counters_dirty = false;
#ifdef RT_STAMPS
  {
    const unsigned long long* t = s->h_async_stats + STAT_RAYS + 3 * STAT_SLOTS;
    double sum = 0;
    for (int k = 0; k < 5; k++) sum += (double)t[k];
    fprintf(stderr, "[rt_amd] wave clock shares: raygen %.3f closest %.3f shade %.3f shadow %.3f output %.3f (total %.3g ticks)\n",
            t[0] / sum, t[1] / sum, t[2] / sum, t[3] / sum, t[4] / sum, sum);
    fprintf(stderr, "[rt_amd]   closest: setup %.3f nodes %.3f tris %.3f quads/gates %.3f | shadow: setup %.3f nodes %.3f tris %.3f quads/gates %.3f\n",
            t[8] / sum, t[9] / sum, t[10] / sum, t[11] / sum, t[16] / sum, t[17] / sum, t[18] / sum, t[19] / sum);
    fprintf(stderr, "[rt_amd]   wave-level loop iterations: closest nodes %llu tris %llu quads %llu rounds %llu | shadow nodes %llu tris %llu quads %llu rounds %llu\n",
            t[12], t[13], t[14], t[15], t[20], t[21], t[22], t[23]);
  }
#endif
  rt_stats one{};
  decode_stats(s->h_async_stats, &one);
  rt_stats& t = s->async_total;
  t.shadow += one.shadow; t.reflect += one.reflect; t.refract += one.refract;
  t.node_visits += one.node_visits; t.tri_tests += one.tri_tests;
  t.quadric_tests += one.quadric_tests; t.gate_tests += one.gate_tests;
  t.shadow_node_visits += one.shadow_node_visits; t.shadow_tri_tests += one.shadow_tri_tests;
  t.shadow_quadric_tests += one.shadow_quadric_tests; t.shadow_gate_tests += one.shadow_gate_tests;
  return RT_OK;
}

// One camera frame of a scene that spawns no rays: a single kernel, nothing to read back, so the
// frame is only enqueued.
int enqueue_direct_frame(rt_scene* s, const PipelineIn& in, uint64_t primary) {
  return enqueue_direct_frames(s, in, primary, 1, nullptr, nullptr);
}

// The same for n_frames frames in one launch (1..MAX_LAUNCH_FRAMES): frame 0 is `in`, frame k > 0 is seen through
// more_cams[k - 1] and written to more_rgb8[k - 1].
int enqueue_direct_frames(rt_scene* s, const PipelineIn& in, uint64_t primary, int n_frames, const CameraDev* more_cams,
                          uint8_t* const* more_rgb8) {
  rt_scene::FrameSlot& f = s->slots[s->next_slot];
  int rc = collect_frame(s, f);  // the slot's previous frame, FRAME_SLOTS frames ago
  if (rc) return rc;
  // consecutive frames go to different streams: they do not depend on each other, and the tail of
  // one kernel (few waves left) overlaps the head of the next
  // a whole frame fills the GPU from two streams (a third only stretches every launch); a part of a frame does not
  const unsigned lanes = in.tiles ? (unsigned)s->n_frame_streams : std::min(2u, (unsigned)s->n_frame_streams);
  const unsigned lane = s->frame_seq++ % lanes;
  hipStream_t stream = lane ? s->frame_stream[lane - 1] : main_stream(s);
  if (lane) s->frame_stream_used[lane - 1] = true;
  s->last_stream = stream;
  s->next_slot = (s->next_slot + 1) % rt_scene::FRAME_SLOTS;
  GenLaunch a;
  a.cur = s->gens[0].view;
  a.next = s->gens[1].view;
  a.n = in.n0;
  a.next_segcap = 0;
  a.remaining = 0;
  a.next_counts = s->d_counts;
  a.cam = in.cam;
  a.tiles = in.tiles;
  a.tiles_x = in.tiles_x;
  a.img = in.img;
  a.stats = s->d_async_stats;
  if (n_frames > 1) {
    a.batch.frames = (uint32_t)n_frames;
    a.batch.waves_per_frame = (in.n0 + BLOCK - 1) / BLOCK;
    for (int k = 1; k < n_frames; k++) {
      a.batch.cam[k - 1] = more_cams[k - 1];
      a.batch.rgb8[k - 1] = more_rgb8[k - 1];
    }
  }
  s->async_counters_dirty = true;
  a.stream = stream;
  HIP_TRY(hipEventRecord(f.begin, stream));
  launch_generation(s, 0, (in.flags & RT_FLAG_COUNTERS) != 0, a);
  HIP_TRY(hipEventRecord(f.end, stream));
  HIP_TRY(hipGetLastError());
  f.primary = primary * (uint64_t)n_frames;
  f.frames = (uint32_t)n_frames;
  f.pending = true;
  return RT_OK;
}

// Record slots per queue segment that can never overflow: the waves of a generation of n rays are dealt
// to the segments round-robin and a wave spawns at most 64 * max_children rays.
static inline uint64_t safe_segcap(uint64_t n, uint32_t max_children) {
  const uint64_t waves = (n + 63) / 64;
  return (waves + QUEUE_SEGMENTS - 1) / QUEUE_SEGMENTS * 64 * max_children;
}

struct GenPlan {
  std::vector<uint32_t> threads;  // grid size of generation g (for g > 0 an upper bound when speculative)
  std::vector<uint32_t> segcap;   // record slots per queue segment of generation g (g > 0)
  explicit GenPlan(int depth) : threads((size_t)depth + 3, 0u), segcap((size_t)depth + 3, 0u) {}
};

static void fill_launch(rt_scene* s, const PipelineIn& in, const GenPlan& plan, int g, int max_gen, GenLaunch& a) {
  a.cur = s->gens[g].view;
  a.next = s->gens[g + 1].view;
  a.n = plan.threads[g];
  a.next_segcap = g < max_gen ? plan.segcap[g + 1] : 0;
  a.remaining = g < max_gen ? std::max(in.remaining0 - g, 0) : 0;
  a.next_counts = s->d_counts + g * QUEUE_SEGMENTS;
  a.cam = in.cam;
  a.tiles = in.tiles;
  a.tiles_x = in.tiles_x;
  a.img = in.img;
  a.stats = s->d_stats;
  a.stream = main_stream(s);
  a.queue_counts = g > 0 ? s->d_counts + (g - 1) * QUEUE_SEGMENTS : nullptr;
  a.segcap = plan.segcap[g];
  if (s->set_aside) {  // room was made by ensure_aside(plan.threads[g])
    a.aside = s->aside_buf.view;
    a.aside.counts = s->d_counts + count_words(s->depth) + g * QUEUE_SEGMENTS;
    a.aside.segcap = aside_segcap(plan.threads[g]);
  }
}

static void launch_resolve_generation(rt_scene* s, const GenPlan& plan, int g) {
  launch_resolve(main_stream(s), s->gens[g].view, plan.threads[g], s->gens[g + 1].view,
                 g > 0 ? s->d_counts + (g - 1) * QUEUE_SEGMENTS : (const uint32_t*)nullptr, plan.segcap[g]);
}

// Colours flow back from the last generation to the first; a camera frame ends in its pixels.
static void launch_resolves(rt_scene* s, const PipelineIn& in, const GenPlan& plan, int last) {
  for (int g = last - 1; g >= (in.camera ? 1 : 0); g--) launch_resolve_generation(s, plan, g);
  if (in.camera && last > 0)
    launch_resolve_pixels(main_stream(s), s->gens[0].view, in.n0, s->gens[1].view, in.cam, in.tiles, in.tiles_x, in.img);
}

// The frame launched at once on twice the queue lengths of the frame before (scene->hint; unused room
// costs waves that exit at once):
// no host round trip between the generations; the kernels find their rays through the segment counters
// on the device.  The counters come back in one copy at the end.  *fits = false when a segment
// received more rays than it had room for: the frame's output is then unusable.
// Sizes the generations from the hints and makes room for them; returns the last generation to launch.
static int plan_speculative(rt_scene* s, const PipelineIn& in, int max_gen, GenPlan& plan, int* last_out) {
  plan.threads[0] = in.n0;
  int last = 0;
  for (int g = 1; g <= max_gen && s->hint[g] > 0; g++) {
    const uint64_t safe = safe_segcap(plan.threads[g - 1], s->max_children);
    const uint64_t want = 2 * (uint64_t)s->hint[g] + 256;
    plan.segcap[g] = (uint32_t)std::min(safe, want);
    plan.threads[g] = plan.segcap[g] * QUEUE_SEGMENTS;
    last = g;
  }
  int rc;
  for (int g = 0; g <= last + 1 && g <= s->depth + 1; g++)
    if ((rc = ensure_gen(s, g, g == 0 ? in.n0 : (size_t)plan.segcap[g] * QUEUE_SEGMENTS))) return rc;
  if (s->set_aside) {
    uint32_t most = 0;
    for (int g = 0; g <= last; g++) most = std::max(most, plan.threads[g]);
    if ((rc = ensure_aside(s, most))) return rc;
  }
  *last_out = last;
  return RT_OK;
}

// Enqueues every generation of the frame, the resolves and the copies of the counters into h_counts / h_stats
// (pinned); begin / end are recorded around the generation kernels.
static int launch_speculative(rt_scene* s, const PipelineIn& in, int max_gen, GenPlan& plan, hipEvent_t begin, hipEvent_t end,
                              uint32_t* h_counts, unsigned long long* h_stats) {
  HIP_TRY(hipMemsetAsync(s->d_stats, 0, STAT_WORDS * sizeof(unsigned long long), main_stream(s)));
  HIP_TRY(hipMemsetAsync(s->d_counts, 0, 2 * count_words(s->depth) * sizeof(uint32_t), main_stream(s)));
  int last = 0, rc;
  if ((rc = plan_speculative(s, in, max_gen, plan, &last))) return rc;
  HIP_TRY(hipEventRecord(begin, main_stream(s)));
  for (int g = 0; g <= last; g++) {
    GenLaunch a;
    fill_launch(s, in, plan, g, max_gen, a);
    launch_generation(s, (g == 0 && in.camera) ? 1 : 2, false, a);
  }
  HIP_TRY(hipEventRecord(end, main_stream(s)));
  launch_resolves(s, in, plan, last);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h_counts, s->d_counts, count_words(s->depth) * sizeof(uint32_t), hipMemcpyDeviceToHost, main_stream(s)));
  HIP_TRY(hipMemcpyAsync(h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, main_stream(s)));
  return RT_OK;
}

// With the counters back: did every queue segment have room for the rays it received?  If so the counts become
// the next frame's hints and the statistics are filled in.
static int check_speculative(rt_scene* s, int max_gen, const uint32_t* segcap, const uint32_t* h_counts,
                             const unsigned long long* h_stats, hipEvent_t begin, hipEvent_t end, rt_stats* stats, bool* fits) {
  *fits = true;
  int true_last = 0;
  for (int g = 1; g <= max_gen && *fits; g++) {  // counters [g - 1]: the rays queued for generation g
    uint32_t fullest = 0;
    for (int k = 0; k < QUEUE_SEGMENTS; k++) fullest = std::max(fullest, h_counts[(g - 1) * QUEUE_SEGMENTS + k]);
    if (fullest > segcap[g]) *fits = false;
    if (fullest == 0) break;
    true_last = g;
  }
  if (!*fits) return RT_OK;
  for (int g = 1; g <= s->depth + 2; g++) {
    uint32_t fullest = 0;
    if (g <= true_last)
      for (int k = 0; k < QUEUE_SEGMENTS; k++) fullest = std::max(fullest, h_counts[(g - 1) * QUEUE_SEGMENTS + k]);
    s->hint[g] = fullest;
  }
  if (stats) {
    decode_stats(h_stats, stats);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, begin, end));
    stats->kernel_ms = ms;
    stats->kernel_launches = (uint64_t)true_last + 1;
    stats->generations = (uint64_t)true_last + 1;
  }
  return RT_OK;
}

static int run_speculative(rt_scene* s, const PipelineIn& in, int max_gen, rt_stats* stats, bool* fits) {
  GenPlan plan(s->depth);
  int rc;
  if ((rc = launch_speculative(s, in, max_gen, plan, s->ev[0], s->ev[1], s->h_counts, s->h_stats))) return rc;
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  return check_speculative(s, max_gen, plan.segcap.data(), s->h_counts, s->h_stats, s->ev[0], s->ev[1], stats, fits);
}

static bool speculation_off() {
  static const bool off = getenv("RT_AMD_NO_SPECULATION") != nullptr;
  return off;
}
static bool hints_match(const rt_scene* s, const PipelineIn& in) {
  return s->hint_valid && s->hint_key[0] == in.n0 && s->hint_key[1] == (uint32_t)in.remaining0;
}
// a frame that did not fit is rendered twice, and a camera that moves this fast will do it again: exact for a while
static void hold_off(rt_scene* s) {
  s->holdoff = s->holdoff_next;
  s->holdoff_next = std::min(2 * s->holdoff_next, 64u);
  static const bool debug = getenv("RT_AMD_DEBUG") != nullptr;
  if (debug) fprintf(stderr, "[rt_amd] speculative frame did not fit; exact for the next %u frames\n", s->holdoff);
}

static int run_pipeline_impl(rt_scene* s, const PipelineIn& in, rt_stats* stats, bool exact);
static int verify_oldest(rt_scene* s, int upto);

// Makes the set that waits the set in use.  Frames enqueued before keep running on the set they were given.
static int swap_lane(rt_scene* s, int spares) {
  rt_scene::Lane& o = s->spare_lane[s->next_spare % spares];
  s->next_spare = (s->next_spare + 1) % spares;
  if (!o.stream) {  // not on the queue of the copy stream, of the set in use or of another set that waits
    std::vector<hipStream_t> avoid = {s->copy_stream, s->stream};
    for (const rt_scene::Lane& other : s->spare_lane)
      if (other.stream) avoid.push_back(other.stream);
    for (hipStream_t st : avoid) HIP_TRY(hipStreamSynchronize(st));  // once per set: the measurement wants them idle
    int rc = pick_streams(avoid.data(), (int)avoid.size(), 1, &o.stream);
    if (rc) return rc;
  }
  if (o.depth != s->depth) {
    int rc = release_lane(o);
    if (rc) return rc;
    HIP_TRY(hipMalloc((void**)&o.d_counts, 2 * count_words(s->depth) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&o.d_stats, STAT_WORDS * sizeof(unsigned long long)));
    o.gens.resize((size_t)s->depth + 2);
    o.depth = s->depth;
  }
  std::swap(s->stream, o.stream);
  std::swap(s->gens, o.gens);
  std::swap(s->aside_buf, o.aside_buf);
  std::swap(s->d_counts, o.d_counts);
  std::swap(s->d_stats, o.d_stats);
  return RT_OK;
}

int enqueue_speculative_frame(rt_scene* s, const PipelineIn& in, uint64_t primary, bool* enqueued) {
  *enqueued = false;
  const int max_gen = std::max(in.remaining0, 0);
  static const bool off = getenv("RT_AMD_NO_ASYNC_GENERATIONS") != nullptr;
  if (off || speculation_off() || s->max_children == 0 || max_gen == 0 || (in.flags & RT_FLAG_COUNTERS) || !in.camera) return RT_OK;
  if (std::max(max_gen, RECURSION_DEPTH) > s->depth || !hints_match(s, in) || s->holdoff > 0) return RT_OK;
  // Unverified frames in flight, this one included (RT_AMD_FRAMES_IN_FLIGHT: 1..4).  Two sets of ray records want
  // two; a third keeps a set's stream fed while the host looks at the frame before, a fourth measured slower on the
  // 100k-primitive sweep (10.2-10.4 ms per frame against 10.0 with two or three; glass 4K the same either way).
  static const int in_flight = getenv("RT_AMD_FRAMES_IN_FLIGHT")
                                   ? std::min(std::max(atoi(getenv("RT_AMD_FRAMES_IN_FLIGHT")), 1), (int)rt_scene::SPEC_SLOTS)
                                   : 3;
  if (s->spec_count >= in_flight) {  // the oldest make room; they may turn out not to have fitted
    int rc = verify_oldest(s, s->spec_count - in_flight + 1);
    if (rc) return rc;
    if (s->spec_count >= in_flight || !hints_match(s, in) || s->holdoff > 0) return RT_OK;
  }
  rt_scene::SpecFrame& f = s->spec[(s->spec_first + s->spec_count) % rt_scene::SPEC_SLOTS];
  if (!f.begin) {
    HIP_TRY(hipEventCreate(&f.begin));
    HIP_TRY(hipEventCreate(&f.end));
    HIP_TRY(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&f.h_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  }
  if (f.counts_depth != s->depth) {
    if (f.h_counts) HIP_TRY(hipHostFree(f.h_counts));
    f.h_counts = nullptr;
    HIP_TRY(hipHostMalloc((void**)&f.h_counts, count_words(s->depth) * sizeof(uint32_t), hipHostMallocDefault));
    f.counts_depth = s->depth;
  }
  // sets of ray records the enqueued frames take in turn (RT_AMD_LANES=1: all frames on one set and one stream)
  static const int lanes = getenv("RT_AMD_LANES") ? std::min(std::max(atoi(getenv("RT_AMD_LANES")), 1), rt_scene::MAX_SPARE_LANES + 1) : 2;
  if (lanes > 1) {  // the frames before this one keep their sets; this one takes the set that has waited longest
    int rcl = swap_lane(s, lanes - 1);
    if (rcl) return rcl;
  }
  GenPlan plan(s->depth);
  int rc = launch_speculative(s, in, max_gen, plan, f.begin, f.end, f.h_counts, f.h_stats);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(f.done, main_stream(s)));
  f.in = in;
  f.max_gen = max_gen;
  f.segcap = plan.segcap;
  f.primary = primary;
  s->spec_count++;
  *enqueued = true;
  return RT_OK;
}

// Does the frame write to this device address?
static bool frame_holds(const rt_scene::SpecFrame& f, const void* p) {
  const size_t pixels = (size_t)f.in.cam.hsize * f.in.cam.vsize;
  auto within = [&](const void* base, size_t bytes) {
    return base && (const char*)p >= (const char*)base && (const char*)p < (const char*)base + bytes;
  };
  return within(f.in.img.rgb8, pixels * 3) || within(f.in.img.rgb64, pixels * 3 * sizeof(double)) ||
         within(f.in.img.hit_leaf, pixels * sizeof(int32_t));
}

// Verifies the `upto` oldest unverified frames.
static int verify_oldest(rt_scene* s, int upto) {
  for (int k = 0; k < upto && s->spec_count > 0; k++) {
    rt_scene::SpecFrame& f = s->spec[s->spec_first];
    s->spec_first = (s->spec_first + 1) % rt_scene::SPEC_SLOTS;
    s->spec_count--;
    HIP_TRY(hipEventSynchronize(f.done));
    rt_stats one{};
    bool fits = false;
    int rc = check_speculative(s, f.max_gen, f.segcap.data(), f.h_counts, f.h_stats, f.begin, f.end, &one, &fits);
    if (rc) return rc;
    if (fits) {
      s->holdoff_next = 1;
    } else {  // again, one generation at a time, into the same buffers (behind whatever was enqueued after it)
      hold_off(s);
      if ((rc = run_pipeline_impl(s, f.in, &one, true))) return rc;
    }
    rt_stats& t = s->async_total;
    t.primary += f.primary;
    t.shadow += one.shadow; t.reflect += one.reflect; t.refract += one.refract;
    t.kernel_ms += one.kernel_ms; t.kernel_launches += one.kernel_launches; t.generations += one.generations;
  }
  return RT_OK;
}

int verify_frames(rt_scene* s, const void* touching, bool* was_frame) {
  if (was_frame) *was_frame = false;
  int upto = s->spec_count;
  if (touching) {
    upto = 0;
    for (int k = 0; k < s->spec_count; k++)
      if (frame_holds(s->spec[(s->spec_first + k) % rt_scene::SPEC_SLOTS], touching)) upto = k + 1;
    if (upto > 0 && was_frame) *was_frame = true;
  }
  return verify_oldest(s, upto);
}

int run_pipeline(rt_scene* s, const PipelineIn& in, rt_stats* stats) { return run_pipeline_impl(s, in, stats, false); }

static int run_pipeline_impl(rt_scene* s, const PipelineIn& in, rt_stats* stats, bool exact) {
  const bool count = (in.flags & RT_FLAG_COUNTERS) != 0;
  int max_gen = std::max(in.remaining0, 0);  // World::color_at recurses `remaining` levels (src/world.rs:146, 185, 249)
  if (s->max_children == 0) max_gen = 0;
  {
    int rcd = ensure_depth(s, std::max(max_gen, RECURSION_DEPTH));
    if (rcd) return rcd;
  }
  const bool direct = in.camera && max_gen == 0;  // every pixel is final after generation 0
  const bool no_speculation = speculation_off();
  const bool hinted = max_gen > 0 && !count && !no_speculation && hints_match(s, in);
  if (hinted && exact) {
    if (s->holdoff > 0) s->holdoff--;
  } else if (hinted && s->holdoff > 0) {
    s->holdoff--;
  } else if (hinted) {
    bool fits = false;
    int rc = run_speculative(s, in, max_gen, stats, &fits);
    if (rc) return rc;
    if (fits) {
      s->holdoff_next = 1;
      return RT_OK;
    }
    hold_off(s);  // start over, one generation at a time, and stay exact for a while
  }
  HIP_TRY(hipMemsetAsync(s->d_stats, 0, STAT_WORDS * sizeof(unsigned long long), main_stream(s)));
  if (max_gen > 0 || s->set_aside) HIP_TRY(hipMemsetAsync(s->d_counts, 0, 2 * count_words(s->depth) * sizeof(uint32_t), main_stream(s)));
  GenPlan plan(s->depth);
  std::vector<uint32_t> fullest((size_t)s->depth + 3, 0u);
  plan.threads[0] = in.n0;
  int last = 0;
  for (int g = 0; g <= max_gen && plan.threads[g] > 0; g++) {
    last = g;
    const uint64_t next_segcap = g < max_gen ? safe_segcap(plan.threads[g], s->max_children) : 0;
    if (next_segcap * QUEUE_SEGMENTS > 0xFFFFFF00ull)
      return fail(RT_ERR_UNSUPPORTED, "too many secondary rays for one generation");
    plan.segcap[g + 1] = (uint32_t)next_segcap;
    int rc;
    if (!direct && g == 0 && (rc = ensure_gen(s, 0, in.n0))) return rc;
    if ((rc = ensure_gen(s, g + 1, (size_t)next_segcap * QUEUE_SEGMENTS))) return rc;
    if (s->set_aside && (rc = ensure_aside(s, plan.threads[g]))) return rc;
    GenLaunch a;
    fill_launch(s, in, plan, g, max_gen, a);
    int mode = (g == 0 && in.camera) ? (direct ? 0 : 1) : 2;
    HIP_TRY(hipEventRecord(s->ev[g * 2 + 0], main_stream(s)));
    launch_generation(s, mode, count, a);
    HIP_TRY(hipEventRecord(s->ev[g * 2 + 1], main_stream(s)));
    if (g < max_gen) {
      uint32_t* h = s->h_counts + g * QUEUE_SEGMENTS;
      HIP_TRY(hipMemcpyAsync(h, s->d_counts + g * QUEUE_SEGMENTS, QUEUE_SEGMENTS * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             main_stream(s)));
      HIP_TRY(hipStreamSynchronize(main_stream(s)));
      uint64_t total = 0;
      for (int k = 0; k < QUEUE_SEGMENTS; k++) {
        total += std::min<uint32_t>(h[k], (uint32_t)next_segcap);
        fullest[g + 1] = std::max(fullest[g + 1], h[k]);
      }
      plan.threads[g + 1] = (uint32_t)total;
    }
#ifdef RT_STAMPS
    {
      static unsigned long long before[STAMP_WORDS];
      if (g == 0) memset(before, 0, sizeof before);
      HIP_TRY(hipStreamSynchronize(main_stream(s)));  // the generation just launched (the copy below does not wait for this stream)
      HIP_TRY(hipMemcpy(s->h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      const unsigned long long* t = s->h_stats + STAT_RAYS + 3 * STAT_SLOTS;
      double d[STAMP_WORDS], sum = 0;
      for (int k = 0; k < STAMP_WORDS; k++) { d[k] = (double)(t[k] - before[k]); before[k] = t[k]; }
      for (int k = 0; k < 5; k++) sum += d[k];
      const double waves = (double)((plan.threads[g] + BLOCK - 1) / BLOCK);
      fprintf(stderr, "[rt_amd] generation %d rays %u: ticks per wave %.0f | raygen %.3f closest %.3f shade %.3f shadow %.3f output %.3f\n",
              g, plan.threads[g], sum / waves, d[0] / sum, d[1] / sum, d[2] / sum, d[3] / sum, d[4] / sum);
      fprintf(stderr, "[rt_amd]   closest: setup %.3f nodes %.3f tris %.3f quads/gates %.3f | shadow: setup %.3f nodes %.3f tris %.3f quads/gates %.3f\n",
              d[8] / sum, d[9] / sum, d[10] / sum, d[11] / sum, d[16] / sum, d[17] / sum, d[18] / sum, d[19] / sum);
      fprintf(stderr, "[rt_amd]   wave-level loop iterations: closest nodes %.0f tris %.0f quads %.0f rounds %.0f | shadow nodes %.0f tris %.0f quads %.0f rounds %.0f (waves %.0f)\n",
              d[12], d[13], d[14], d[15], d[20], d[21], d[22], d[23], waves);
    }
#endif
  }
  launch_resolves(s, in, plan, last);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                         main_stream(s)));
  HIP_TRY(hipStreamSynchronize(main_stream(s)));
  if (max_gen > 0) {
    for (int g = 0; g <= s->depth + 2; g++) s->hint[(size_t)g] = fullest[(size_t)g];
    s->hint_key[0] = in.n0;
    s->hint_key[1] = (uint32_t)in.remaining0;
    s->hint_valid = true;
    if (!count && !no_speculation) {  // room for the next frame's queues now, not inside that frame
      GenPlan next_plan(s->depth);
      int next_last = 0, rc;
      if ((rc = plan_speculative(s, in, max_gen, next_plan, &next_last))) return rc;
    }
  }
  if (stats) {
    decode_stats(s->h_stats, stats);
    stats->kernel_ms = 0;
    stats->kernel_launches = 0;
    stats->generations = (uint64_t)last + 1;
    for (int g = 0; g <= last; g++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, s->ev[g * 2 + 0], s->ev[g * 2 + 1]));
      stats->kernel_ms += ms;
      stats->kernel_launches++;
    }
  }
  return RT_OK;
}

int stage_reserve(rt_scene* s, size_t bytes) {
  if (s->stage_bytes >= bytes) return RT_OK;
  if (s->stage) {
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
    HIP_TRY(hipFree(s->stage));
    s->stage = nullptr;
    s->stage_bytes = 0;
  }
  HIP_TRY(hipMalloc(&s->stage, bytes));
  s->stage_bytes = bytes;
  return RT_OK;
}

}  // namespace rt
