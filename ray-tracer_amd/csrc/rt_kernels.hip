// The small kernels around the generation kernel: loading caller rays, resolving colours back through
// the generations, the per-ray entry points of the C ABI, and the refit of the hierarchy.
#include <algorithm>

#include "rt_launch.h"
#include "rt_shading.h"
#include "rt_traverse.h"

namespace rt {

// Rays supplied by the caller (rt_color_at and friends): Ray4D::new keeps the
// three components as given.
__global__ void __launch_bounds__(BLOCK) k_load_rays(GenView g, uint32_t n, const double* rays6) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double* r = rays6 + (size_t)i * 6;
  g.ox[i] = r[0]; g.oy[i] = r[1]; g.oz[i] = r[2];
  g.dx[i] = r[3]; g.dy[i] = r[4]; g.dz[i] = r[5];
}

// shade_hit's final expression (src/world.rs:236-245) for record i, its children resolved already.
// A record without children keeps its surface colour (returns false, nothing written).
DEV bool resolve_record(const GenView& g, const GenView& next, uint32_t i, double* out_r = nullptr,
                        double* out_g = nullptr, double* out_b = nullptr) {
  const uint32_t flags = g.flags[i];
  if (!(flags & 4u)) return false;
  uint32_t cr = g.child_reflect[i], cf = g.child_refract[i];
  double rr = 0.0, rg = 0.0, rb = 0.0, fr = 0.0, fg = 0.0, fb = 0.0;
  if (cr != REF_NONE) {
    double k = g.reflective[i];
    rr = next.cr[cr] * k; rg = next.cg[cr] * k; rb = next.cb[cr] * k;
  }
  if (cf != REF_NONE) {
    double k = g.transparency[i];
    fr = next.cr[cf] * k; fg = next.cg[cf] * k; fb = next.cb[cf] * k;
  }
  double r = g.cr[i], gg = g.cg[i], b = g.cb[i];
  if (flags & 2u) {
    double R = g.reflectance[i];
    r = r + rr * R + fr * (1.0 - R);
    gg = gg + rg * R + fg * (1.0 - R);
    b = b + rb * R + fb * (1.0 - R);
  } else {
    r = r + rr + fr;
    gg = gg + rg + fg;
    b = b + rb + fb;
  }
  if (out_r) {
    *out_r = r; *out_g = gg; *out_b = b;
  } else {
    g.cr[i] = r; g.cg[i] = gg; g.cb[i] = b;
  }
  return true;
}

// Children first: generation g is resolved after generation g + 1.
__global__ void __launch_bounds__(BLOCK)
    k_resolve(GenView g, uint32_t n, GenView next, const uint32_t* queue_counts, uint32_t segcap) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  bool active = i < n;
  if (queue_counts) {  // spawned rays: records found through the segment counters
    uint32_t rec;
    active = queue_slot(queue_counts, segcap, i, rec) && active;
    i = rec;
  }
  if (!active) return;
  resolve_record(g, next, i);
}

// The last step of a camera frame with secondary rays: the pixels whose primary ray had children get
// their final colour (the kernel of generation 0 wrote every pixel with the surface colour alone).
__global__ void __launch_bounds__(BLOCK)
    k_resolve_pixels(GenView g, uint32_t n, GenView next, CameraDev cam, const uint32_t* tiles, uint32_t tiles_x,
                     ImageOut img) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  double r, gg, b;
  if (!resolve_record(g, next, i, &r, &gg, &b)) return;
  uint32_t x = 0, y = 0;
  if (!ray_pixel(tiles, tiles_x, cam, i, x, y)) return;
  const size_t p = (size_t)y * cam.hsize + x;
  if (img.rgb64) {
    img.rgb64[p * 3 + 0] = r; img.rgb64[p * 3 + 1] = gg; img.rgb64[p * 3 + 2] = b;
  }
  if (img.rgb8) {
    uint8_t* dst = img.rgb8 + (img.compact ? (size_t)i : p) * 3;
    dst[0] = quantise(r); dst[1] = quantise(gg); dst[2] = quantise(b);
  }
}

__global__ void __launch_bounds__(BLOCK)
    k_output_rays(GenView g, uint32_t n, double* rgb, int32_t* leaf, double* t) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (rgb) {
    rgb[(size_t)i * 3 + 0] = g.cr[i]; rgb[(size_t)i * 3 + 1] = g.cg[i]; rgb[(size_t)i * 3 + 2] = g.cb[i];
  }
  int l = g.leaf[i];
  if (leaf) leaf[i] = l;
  if (t) t[i] = l >= 0 ? g.t[i] : NAN;
}

// World::intersect + Intersections::hit only (rt_intersect_hit).
template <bool CSG>
__global__ void __launch_bounds__(BLOCK) k_intersect(SceneDev S, GenView g, uint32_t n) {
  extern __shared__ uint32_t dyn_lds[];  // traversal stack, then the CSG hit list (see k_generation)
  uint32_t i = (blockIdx.x + S.first_block) * BLOCK + threadIdx.x;
  if (i >= n) return;
  LdsStack<BLOCK, CSG> st;
  st.init(dyn_lds, S.stack_entries);
  Counters cnt{};
  ClosestQuery Q;
  Q.init();
  traverse<ClosestQuery, LdsStack<BLOCK, CSG>, false>(S, load_ray(g, i), Q, st, cnt);
  bool hit = Q.best_leaf != 0x7FFFFFFF;
  g.leaf[i] = hit ? Q.best_leaf : -1;
  if (hit) g.t[i] = Q.best_t;
}

// World::is_shadowed for caller-supplied points.
template <bool CSG>
__global__ void __launch_bounds__(BLOCK) k_points_shadowed(SceneDev S, uint32_t n, const double* pts, uint8_t* out) {
  extern __shared__ uint32_t dyn_lds[];  // traversal stack, then the CSG hit list (see k_generation)
  uint32_t i = (blockIdx.x + S.first_block) * BLOCK + threadIdx.x;
  if (i >= n) return;
  LdsStack<BLOCK, CSG> st;
  st.init(dyn_lds, S.stack_entries);
  Counters cnt{};
  V3 p = v3(pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]);
  V3 v = v3(S.light_position[0], S.light_position[1], S.light_position[2]) - p;
  ShadowQuery Q;
  Q.distance = magnitude(v);
  Q.found = false;
  Ray ray{p, normalize(v)};
  traverse<ShadowQuery, LdsStack<BLOCK, CSG>, false>(S, ray, Q, st, cnt);
  out[i] = Q.found ? 1 : 0;
}

__global__ void __launch_bounds__(BLOCK) k_pixel_rays(CameraDev cam, uint32_t n, const uint32_t* xy, double* rays6) {
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  Ray r = camera_ray(cam, xy[i * 2], xy[i * 2 + 1]);
  double* o = rays6 + (size_t)i * 6;
  o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
  o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

// ---------------------------------------------------------------------------
// refit: new boxes for a hierarchy whose primitives moved (rt_scene_update)
// ---------------------------------------------------------------------------
// The reference lets any shape move between renders (Shape::set_transform re-propagates to the
// children, src/shape.rs:592-656).  The topology of the hierarchy stays; every box is recomputed bottom
// up from the boxes of the primitive slots.  One thread starts at every node none of whose children is
// another node (directly or behind a gate); a thread that finishes a node writes the node's box into
// its slot of the parent and goes on with the parent if it was the last child to arrive.
// A plane another thread (possibly on another compute unit) wrote during this launch: read past the
// vector L1, which is not coherent between compute units.
DEV float refit_read(const float* p) {
  return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

DEV void refit_run_box(const RefitView& v, uint32_t ref, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; a++) {
    lo[a] = INFINITY;
    hi[a] = -INFINITY;
  }
  if (ref == REF_NONE) return;
  const uint32_t kind = ref >> 30;
  if (kind != KIND_TRI && kind != KIND_QUAD) return;
  const float* boxes = kind == KIND_TRI ? v.tri_box : v.quad_box;
  const uint32_t first = ref & MAX_RUN_FIRST, count = ((ref >> 27) & 7u) + 1u;
  for (uint32_t i = 0; i < count; i++)
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], boxes[(size_t)(first + i) * 6 + a]);
      hi[a] = fmaxf(hi[a], boxes[(size_t)(first + i) * 6 + 3 + a]);
    }
}

__global__ void __launch_bounds__(256) k_refit(RefitView v) {
  uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= v.n_nodes || v.waits[n] != 0) return;
  for (;;) {
    BvhNode& node = v.nodes[n];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 4; c++) {
      uint32_t ref = node.child[c];
      if (ref != REF_NONE && (ref >> 30) == KIND_GATE) ref = v.gates[ref & 0x3FFFFFFFu].root;  // a run right behind a gate
      if (ref != REF_NONE && ((ref >> 30) == KIND_TRI || (ref >> 30) == KIND_QUAD)) {
        float l[3], h[3];
        refit_run_box(v, ref, l, h);
        for (int a = 0; a < 3; a++) {
          node.plane[a][0][c] = l[a];
          node.plane[a][1][c] = h[a];
        }
      }
      if (node.child[c] != REF_NONE)  // slots of other nodes were written by the threads that finished them
        for (int a = 0; a < 3; a++) {
          lo[a] = fminf(lo[a], refit_read(&node.plane[a][0][c]));
          hi[a] = fmaxf(hi[a], refit_read(&node.plane[a][1][c]));
        }
      node.alt[c] = node.child[c];  // after an update every gate is tested (see rt_scene_update)
    }
    const uint32_t up = v.parent[n];
    if (up == REF_NONE) return;
    BvhNode& p = v.nodes[up >> 2];
    const uint32_t slot = up & 3u;
    for (int a = 0; a < 3; a++) {
      p.plane[a][0][slot] = lo[a];
      p.plane[a][1][slot] = hi[a];
    }
    __threadfence();
    const uint32_t before = atomicAdd(&v.arrived[up >> 2], 1u);
    if (before + 1u != v.waits[up >> 2]) return;  // a sibling subtree is still on its way: its thread goes on
    __threadfence();
    n = up >> 2;
  }
}

// ---------------------------------------------------------------------------
// host-side launchers (declared in rt_launch.h)
// ---------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + BLOCK - 1) / BLOCK)); }

void launch_load_rays(hipStream_t stream, GenView g, uint32_t n, const double* rays6) {
  hipLaunchKernelGGL(k_load_rays, grid_for(n), dim3(BLOCK), 0, stream, g, n, rays6);
}
void launch_resolve(hipStream_t stream, GenView g, uint32_t n, GenView next, const uint32_t* queue_counts, uint32_t segcap) {
  hipLaunchKernelGGL(k_resolve, grid_for(n), dim3(BLOCK), 0, stream, g, n, next, queue_counts, segcap);
}
void launch_resolve_pixels(hipStream_t stream, GenView g, uint32_t n, GenView next, const CameraDev& cam, const uint32_t* tiles,
                           uint32_t tiles_x, const ImageOut& img) {
  hipLaunchKernelGGL(k_resolve_pixels, grid_for(n), dim3(BLOCK), 0, stream, g, n, next, cam, tiles, tiles_x, img);
}
void launch_output_rays(hipStream_t stream, GenView g, uint32_t n, double* rgb, int32_t* leaf, double* t) {
  hipLaunchKernelGGL(k_output_rays, grid_for(n), dim3(BLOCK), 0, stream, g, n, rgb, leaf, t);
}
// a scene whose CSG hit lists live in device memory owns csg_spill_slots of them: its jobs go in pieces
template <class Launch>
static void in_pieces(const SceneDev& scene, uint64_t n, Launch&& launch) {
  const uint32_t blocks = grid_for(n).x;
  if (!scene.csg_spill) {
    launch(scene, dim3(blocks));
    return;
  }
  for (uint32_t first = 0; first < blocks; first += scene.csg_spill_slots) {
    SceneDev piece = scene;
    piece.first_block = first;
    launch(piece, dim3(std::min(scene.csg_spill_slots, blocks - first)));
  }
}
void launch_intersect(hipStream_t stream, const SceneDev& scene, bool general, unsigned lds_bytes, GenView g, uint32_t n) {
  in_pieces(scene, n, [&](const SceneDev& dev, dim3 grid) {
    if (general) hipLaunchKernelGGL((k_intersect<true>), grid, dim3(BLOCK), lds_bytes, stream, dev, g, n);
    else hipLaunchKernelGGL((k_intersect<false>), grid, dim3(BLOCK), lds_bytes, stream, dev, g, n);
  });
}
void launch_points_shadowed(hipStream_t stream, const SceneDev& scene, bool general, unsigned lds_bytes, uint32_t n,
                            const double* pts, uint8_t* out) {
  in_pieces(scene, n, [&](const SceneDev& dev, dim3 grid) {
    if (general) hipLaunchKernelGGL((k_points_shadowed<true>), grid, dim3(BLOCK), lds_bytes, stream, dev, n, pts, out);
    else hipLaunchKernelGGL((k_points_shadowed<false>), grid, dim3(BLOCK), lds_bytes, stream, dev, n, pts, out);
  });
}
void launch_pixel_rays(hipStream_t stream, const CameraDev& cam, uint32_t n, const uint32_t* xy, double* rays6) {
  hipLaunchKernelGGL(k_pixel_rays, grid_for(n), dim3(BLOCK), 0, stream, cam, n, xy, rays6);
}
void launch_refit(hipStream_t stream, const RefitView& v) {
  hipLaunchKernelGGL(k_refit, dim3((v.n_nodes + 255) / 256), dim3(256), 0, stream, v);
}

// One wave that does nothing for `ticks` of the constant-rate counter (100 MHz): two of these on two streams take
// once that time when the streams have hardware queues of their own and twice when they share one.
__global__ void __launch_bounds__(64) k_linger(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long now = t0;
  for (unsigned guard = 0; guard < (1u << 22) && now - t0 < ticks; guard++) {  // the guard ends it whatever the clock does
    __builtin_amdgcn_s_sleep(8);
    now = __builtin_amdgcn_s_memtime();
  }
  if (sink && threadIdx.x == 0) *sink = now - t0;  // a vector store
}
void launch_linger(hipStream_t stream, unsigned long long ticks, unsigned long long* sink) {
  hipLaunchKernelGGL(k_linger, dim3(1), dim3(64), 0, stream, ticks, sink);
}

}  // namespace rt
