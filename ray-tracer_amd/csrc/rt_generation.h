// The generation kernel: one ray per lane from camera or queue to its surface colour, and the
// launcher of its instantiations.
#pragma once
#include <algorithm>

#include "rt_launch.h"
#include "rt_shading.h"
#include "rt_traverse.h"

namespace rt {

// One generation of rays, start to finish, one ray per lane:
//   ray (camera or queue) -> closest hit (World::intersect + Intersections::hit)
//   -> IntersectionComputation::new (src/intersect.rs:152-184) -> lighting terms
//   (src/light.rs:125-174) -> reflection / refraction rays appended to the next
//   generation (src/world.rs:142-215) -> shadow query (src/world.rs:123-137)
//   -> surface colour.
// MODE 0: camera rays of a scene that never spawns rays: the pixel is final, written straight
//         to the image.  MODE 1: camera rays with ray records.  MODE 2: rays read from g.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_XCD_RUN
#define RT_XCD_RUN 16  // pixel blocks (waves) per run on one XCD: 16 = one 32x32 tile; 1 = plain order
#endif
#ifndef RT_MIN_WAVES_SPAWN
#define RT_MIN_WAVES_SPAWN 3
#endif
template <bool COUNT, int MODE, bool CSG, bool UNIFORM, bool TRIPF = false>
// Waves per SIMD a kernel is compiled for.  The ray-spawning modes need about 170 registers: at three waves
// nothing much spills, and small scenes, whose records come from L1, run fastest that way (glass 4K: 0.67 ms
// against 0.88 at four waves, 0.76 at two).  The kernels for large hierarchies wait on L2 and beyond, and a
// fourth wave hides more of that than the spilled registers cost (100k primitives 8K: 11.1 -> 10.7 ms).
__global__ void __launch_bounds__(BLOCK, MODE == 0 || UNIFORM ? RT_MIN_WAVES : RT_MIN_WAVES_SPAWN)
    k_generation(SceneDev S, GenView g, uint32_t n, GenView next, uint32_t next_segcap, int remaining,
                 uint32_t* next_counts, unsigned long long* stats, CameraDev cam, const uint32_t* tiles,
                 uint32_t tiles_x, ImageOut img, const uint32_t* queue_counts, uint32_t segcap, ShadowView aside,
                 FrameBatch batch) {
  // dynamic LDS: SceneDev::stack_entries * BLOCK words of traversal stack, then SceneDev::csg_hits * 3 * BLOCK words
  // of CSG hit list when the scene has operators
  extern __shared__ uint32_t dyn_lds[];
  LdsStack<BLOCK, CSG, UNIFORM, TRIPF> st;
  st.init(dyn_lds, S.stack_entries);
#ifdef RT_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef RT_WAVE_CLOCK_IMAGE  // diagnostic build: the frame shows how long each wave ran instead of the colours
  const unsigned long long wave_begin_ = __builtin_amdgcn_s_memtime();
#endif
  // i: the thread's ray; rec: where the ray's record lives.  Camera rays and rays handed in by the
  // caller are dense (rec == i, n of them); spawned rays are found through the segment counters the
  // generation before left in queue_counts, and n only bounds the grid.
  // Camera rays: which 8x8 pixel block a wave takes.  Workgroups are dealt round-robin over the eight
  // XCDs (blocks b and b + 8 share one, each XCD with its own L2), so consecutive block numbers
  // would spread the sixteen blocks of a tile over all eight L2s: every L2 would hold the nodes and
  // triangles behind every part of the image, and every 128-byte line of the frame would leave
  // several L2s as partial writes.  Instead the blocks that share an XCD take runs of XCD_RUN
  // consecutive pixel blocks: one 32 x 32 tile.  Measured on the 1M-triangle mesh at 4K: 2.96 ms in
  // plain order, 2.74 ms with runs of 16, 2.77 with 64, 3.14 with 256 (longer runs leave the XCDs
  // unevenly loaded: the teapot frame gets 5 % slower at 64 and is unchanged at 16).
  uint32_t wave_index = blockIdx.x;
  if (MODE != 2) {
    constexpr uint32_t XCD_RUN = RT_XCD_RUN, XCDS = 8;
    const uint32_t full = gridDim.x - gridDim.x % (XCDS * XCD_RUN);  // the rest keeps its order
    if (XCD_RUN > 1 && wave_index < full) {
      const uint32_t xcd = wave_index % XCDS, k = wave_index / XCDS;
      wave_index = ((k / XCD_RUN) * XCDS + xcd) * XCD_RUN + k % XCD_RUN;
    }
  }
  uint32_t frame = 0;  // which frame of the launch this wave belongs to
  if (MODE == 0 && batch.frames > 1) {
    frame = wave_index / batch.waves_per_frame;
    wave_index -= frame * batch.waves_per_frame;
  }
  wave_index += S.first_block;  // a job launched in pieces (SceneDev::csg_spill_slots)
  const uint32_t i = wave_index * BLOCK + thread_in_block();
  uint32_t rec = i;
  bool active = i < n;
  if (MODE == 2 && queue_counts) active = queue_slot(queue_counts, segcap, i, rec) && active;
  const bool has_record = active;
  uint32_t px = 0, py = 0;
  Ray ray{v3(0, 0, 0), v3(0, 0, 1)};
  if (MODE != 2) {
    active = active && ray_pixel(tiles, tiles_x, cam, i, px, py);
    if (active) ray = camera_ray(MODE == 0 && frame ? batch.cam[frame - 1] : cam, px, py);
  } else if (active) {
    ray = load_ray(g, rec);
  }

  Counters ct{}, cs{};
#ifdef RT_STAMPS
  ct.sink = stats + STAT_RAYS + 3 * STAT_SLOTS + 8;   // time sums at sink[0..3], wave iterations at sink[4..7]:
  cs.sink = stats + STAT_RAYS + 3 * STAT_SLOTS + 16;  // two blocks of eight
#endif
  ClosestQuery Q;
  Q.init();
  RT_STAMP(0);  // ray generation
  if (active) traverse<ClosestQuery, LdsStack<BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, ray, Q, st, ct);
  RT_STAMP(1);  // closest-hit traversal
  const bool hit = active && Q.best_leaf != 0x7FFFFFFF;
  const int leaf = hit ? Q.best_leaf : -1;

  bool want_reflect = false, want_refract = false;
  Ray reflect_ray{v3(0, 0, 0), v3(0, 0, 0)}, refract_ray{v3(0, 0, 0), v3(0, 0, 0)};
  V3 over_point = v3(0, 0, 0), lightv = v3(0, 0, 0);
  double light_distance = 0;
  double ar = 0, ag = 0, ab = 0, lr = 0, lg = 0, lb = 0;
  double k_reflective = 0, k_transparency = 0, reflectance = 0;
  uint32_t flags = hit ? 1u : 0u;
  if (hit) {
    const LeafShade L = S.leaves[leaf];
    const MaterialDev* material = S.materials + L.material;
    // n1 / n2 (src/intersect.rs:191-234) first: the second walk then runs while little else is live
    double n1 = 1.0, n2 = 1.0;
    if (MODE != 0 && (material->flags & 8)) {  // they are consumed: refraction or Schlick
      ContainerQuery C;
      C.init(leaf);
      traverse<ContainerQuery, LdsStack<BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, ray, C, st, ct);
      double ri_hit = material->refractive_index;
      double ri_other = C.other_leaf >= 0 ? S.materials[S.leaves[C.other_leaf].material].refractive_index : 1.0;
      if (C.hit_inside) {
        bool hit_last = C.other_leaf < 0 || C.hit_key > C.other_key ||
                        (C.hit_key == C.other_key && C.hit_leaf > C.other_leaf);
        n1 = hit_last ? ri_hit : ri_other;
        n2 = ri_other;
      } else {
        n1 = ri_other;
        n2 = ri_hit;
      }
    }
    const MaterialDev m = *material;
    const XformDev& X = S.xforms[L.xform];
    const double t = Q.best_t;
    V3 point = ray.o + (ray.d * t);  // Ray4D::position
    V3 eyev = -ray.d;
    // normal_at (src/shape.rs:1478-1484).  A plane and a flat triangle have one normal wherever they are hit:
    // the host worked it out with these same operations (ShadingTables::build), and the point need not be
    // taken to object space.
    V3 normalv;
    if (L.fixed_normal) {
      normalv = v3(L.plane_normal[0], L.plane_normal[1], L.plane_normal[2]);
    } else {
      V3 local_point = mul_point(X.full_inv, point);
      V3 ln = local_normal(S, L, local_point, Q.u, Q.v);
      const double* fi = X.full_inv;  // transposition() * normal, w dropped, normalize
      normalv = normalize(v3(fi[0] * ln.x + fi[4] * ln.y + fi[8] * ln.z, fi[1] * ln.x + fi[5] * ln.y + fi[9] * ln.z,
                             fi[2] * ln.x + fi[6] * ln.y + fi[10] * ln.z));
    }
    if (dot(normalv, eyev) < 0.0) normalv = -normalv;
    over_point = point + normalv * FEQ_EPSILON;

    // lighting(): evaluated at over_point, pattern included (src/world.rs:223-227)
    double cr, cg, cb;
    if (m.pattern >= 0) {
      V3 object_point = mul_point(X.full_inv, over_point);
      V3 pattern_point = mul_point(S.patterns[m.pattern].inv, object_point);
      pattern_at<MAX_BLEND_DEPTH>(S, m.pattern, pattern_point, cr, cg, cb);
    } else {
      cr = m.color[0]; cg = m.color[1]; cb = m.color[2];
    }
    double er = cr * S.light_intensity[0], eg = cg * S.light_intensity[1], eb = cb * S.light_intensity[2];
    // the vector to the light serves lighting() and is_shadowed() (src/light.rs:137, src/world.rs:124-128)
    const V3 to_light = v3(S.light_position[0], S.light_position[1], S.light_position[2]) - over_point;
    light_distance = magnitude(to_light);
    lightv = to_light * (1.0 / light_distance);  // Tuple4D::normalize
    ar = er * m.ambient; ag = eg * m.ambient; ab = eb * m.ambient;
    double dr = 0.0, dg = 0.0, db = 0.0, sr = 0.0, sg = 0.0, sb = 0.0;
    double light_dot_normal = dot(lightv, normalv);
    if (!(light_dot_normal < 0.0)) {
      dr = er * m.diffuse * light_dot_normal;
      dg = eg * m.diffuse * light_dot_normal;
      db = eb * m.diffuse * light_dot_normal;
      V3 rv = reflect(-lightv, normalv);
      double reflect_dot_eye = dot(rv, eyev);
      if (!(reflect_dot_eye <= 0.0)) {
        double factor = specular_power(reflect_dot_eye, m.shininess);
        sr = S.light_intensity[0] * m.specular * factor;
        sg = S.light_intensity[1] * m.specular * factor;
        sb = S.light_intensity[2] * m.specular * factor;
      }
    }
    lr = ar + dr + sr; lg = ag + dg + sg; lb = ab + db + sb;

    if (MODE != 0) {
      k_reflective = m.reflective;
      k_transparency = m.transparency;
      // reflected_color (src/world.rs:142-159)
      if (remaining != 0 && (m.flags & 1)) {
        want_reflect = true;
        reflect_ray = Ray{over_point, reflect(ray.d, normalv)};
      }
      // refracted_color (src/world.rs:181-215)
      if (remaining != 0 && (m.flags & 2)) {
        double n_ratio = n1 / n2;
        double cos_i = dot(eyev, normalv);
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
          double cos_t = sqrt(1.0 - sin2_t);
          V3 under_point = point - normalv * FEQ_EPSILON;
          V3 direction = normalv * (n_ratio * cos_i - cos_t) - eyev * n_ratio;
          want_refract = true;
          refract_ray = Ray{under_point, direction};
        }
      }
      if (m.flags & 4) {  // schlick (src/intersect.rs:240-260)
        double cosv = dot(eyev, normalv);
        bool total = false;
        if (n1 > n2) {
          double nn = n1 / n2;
          double sin2_t = (nn * nn) * (1.0 - cosv * cosv);
          if (sin2_t > 1.0) total = true;
          else cosv = sqrt(1.0 - sin2_t);
        }
        if (total) {
          reflectance = 1.0;
        } else {
          double q = (n1 - n2) / (n1 + n2);
          double r0 = q * q;
          double x = 1.0 - cosv;
          double x2 = x * x;
          reflectance = r0 + (1.0 - r0) * (x * (x2 * x2));  // powi(5)
        }
        flags = 3u;
      }
    }
  }

  bool keep_record = false;
  if (MODE != 0) {
    // next generation: one atomic per wave that has something to append
    uint32_t child_reflect = REF_NONE, child_refract = REF_NONE;
    // this wave's segment of the next generation's queue
    const uint32_t seg = (blockIdx.x * (BLOCK / 64) + (thread_in_block() >> 6)) % QUEUE_SEGMENTS;
    uint32_t ri = wave_append(next_counts + seg, want_reflect);
    if (want_reflect && ri < next_segcap) {
      ri += seg * next_segcap;
      next.ox[ri] = reflect_ray.o.x; next.oy[ri] = reflect_ray.o.y; next.oz[ri] = reflect_ray.o.z;
      next.dx[ri] = reflect_ray.d.x; next.dy[ri] = reflect_ray.d.y; next.dz[ri] = reflect_ray.d.z;
      child_reflect = ri;
    }
    uint32_t ti = wave_append(next_counts + seg, want_refract);
    if (want_refract && ti < next_segcap) {
      ti += seg * next_segcap;
      next.ox[ti] = refract_ray.o.x; next.oy[ti] = refract_ray.o.y; next.oz[ti] = refract_ray.o.z;
      next.dx[ti] = refract_ray.d.x; next.dy[ti] = refract_ray.d.y; next.dz[ti] = refract_ray.d.z;
      child_refract = ti;
    }
    // Camera rays write their pixel below; only those with children are looked at again
    // (k_resolve_pixels), so the others leave nothing but their flags.
    if (child_reflect != REF_NONE || child_refract != REF_NONE) flags |= 4u;
    keep_record = has_record && (MODE == 2 || (flags & 4u));
    if (has_record) g.flags[rec] = flags;
    if (keep_record) {
      g.leaf[rec] = leaf;
      g.child_reflect[rec] = child_reflect;
      g.child_refract[rec] = child_refract;
      if (hit) {
        g.t[rec] = Q.best_t;
        g.reflective[rec] = k_reflective;
        g.transparency[rec] = k_transparency;
        g.reflectance[rec] = reflectance;
      }
    }
  }

  // World::is_shadowed (src/world.rs:123-137), then the in_shadow branch of lighting
  double out_r = 0.0, out_g = 0.0, out_b = 0.0;  // World::color_at: no hit => black
  RT_STAMP(2);  // shading, spawning
  // lighting() returns `ambient` in shadow and `ambient + diffuse + specular` otherwise
  // (src/light.rs:135-173).  Where the two are the same bits -- the light behind the surface, or a
  // material without diffuse and specular terms -- the answer of is_shadowed cannot reach the
  // colour, and the query (still counted as a ray) is not traversed.
  const bool decides = hit && (__double_as_longlong(ar) != __double_as_longlong(lr) ||
                               __double_as_longlong(ag) != __double_as_longlong(lg) ||
                               __double_as_longlong(ab) != __double_as_longlong(lb));
  // A hit on a cone's extension (src/shape.rs:1089-1103) lies far from the hits of the lanes around it, and so
  // does its shadow query: walked here, it would hold the whole wave until it is done.  Such queries are set
  // aside (point, ambient colour, record) for k_shadow, which walks them in waves of their own and replaces
  // the lit colour stored here where the point is in shadow.  When the queue is full the query stays.
  bool walk = decides;
  if (aside.segcap != 0u && S.n_bands > 0) {
    const uint32_t seg = blockIdx.x % QUEUE_SEGMENTS;
    const bool far = decides && Q.on_cone_extension();
    const uint32_t k = wave_append(aside.counts + seg, far);
    if (far && k < aside.segcap) {
      const uint32_t e = seg * aside.segcap + k;
      aside.px[e] = over_point.x; aside.py[e] = over_point.y; aside.pz[e] = over_point.z;
      aside.ar[e] = ar; aside.ag[e] = ag; aside.ab[e] = ab;
      aside.target[e] = rec;
      walk = false;
    }
  }
  if (hit) {
    ShadowQuery SQ;
    SQ.distance = light_distance;
    SQ.found = false;
    Ray shadow_ray{over_point, lightv};
    if (walk)
      traverse<ShadowQuery, LdsStack<BLOCK, CSG, UNIFORM, TRIPF>, COUNT>(S, shadow_ray, SQ, st, cs);
    out_r = SQ.found ? ar : lr;
    out_g = SQ.found ? ag : lg;
    out_b = SQ.found ? ab : lb;
  }
  RT_STAMP(3);  // shadow traversal
#ifdef RT_WAVE_CLOCK_IMAGE
  {
#if RT_WAVE_CLOCK_IMAGE == 2  // wave-level iterations instead: node visits in the low 12 bits, triangle tests above
    unsigned nodes_ = ct.wi[0] + cs.wi[0], tris_ = ct.wi[1] + cs.wi[1];
    for (int off = 32; off > 0; off >>= 1) {
      nodes_ += __shfl_xor(nodes_, off);
      tris_ += __shfl_xor(tris_, off);
    }
    const unsigned long long ticks = (nodes_ < 4095u ? nodes_ : 4095u) | ((tris_ < 4095u ? tris_ : 4095u) << 12);
#else
    const unsigned long long ticks = __builtin_amdgcn_s_memtime() - wave_begin_;  // 24 bits into the three bytes
#endif
    const unsigned b0 = ticks & 255u, b1 = (ticks >> 8) & 255u, b2 = (ticks >> 16) & 255u;
    out_r = b0 ? (b0 - 0.5) / 255.0 : 0.0;
    out_g = b1 ? (b1 - 0.5) / 255.0 : 0.0;
    out_b = b2 ? (b2 - 0.5) / 255.0 : 0.0;
  }
#endif
  if (MODE == 0 && frame) img.rgb8 = batch.rgb8[frame - 1];
  if (MODE != 2) write_pixel(img, cam.hsize, px, py, i, active, out_r, out_g, out_b, leaf);
  if (MODE != 0 && keep_record) {
    g.cr[rec] = out_r; g.cg[rec] = out_g; g.cb[rec] = out_b;
  }

  // ray statistics, one add per wave and kind, spread over STAT_SLOTS addresses
  unsigned long long hm = __ballot(hit), rm = __ballot(want_reflect), fm = __ballot(want_refract);
  if ((thread_in_block() & 63) == 0) {
    uint32_t slot = (blockIdx.x * (BLOCK / 64) + (thread_in_block() >> 6)) % STAT_SLOTS;
    if (hm) atomicAdd(&stats[STAT_RAYS + 0 * STAT_SLOTS + slot], (unsigned long long)__popcll(hm));
    if (rm) atomicAdd(&stats[STAT_RAYS + 1 * STAT_SLOTS + slot], (unsigned long long)__popcll(rm));
    if (fm) atomicAdd(&stats[STAT_RAYS + 2 * STAT_SLOTS + slot], (unsigned long long)__popcll(fm));
  }
  if (COUNT) {
    flush_counters(ct, stats);
    flush_counters(cs, stats + 4);
  }
  RT_STAMP(4);  // output
}

// The shadow queries a generation kernel set aside, one per lane in waves of their own: World::is_shadowed
// (src/world.rs:123-137), then lighting()'s in_shadow branch (src/light.rs:135-141) for the points that
// are: the ambient colour replaces the lit one the generation kernel stored -- in the ray's record when it
// has one that is read again (mode 2: every ray; mode 1: camera rays with children), else in its pixel.
template <bool COUNT, bool CSG, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK, RT_MIN_WAVES)
    k_shadow(SceneDev S, ShadowView q, uint32_t n, GenView g, int mode, unsigned long long* stats, CameraDev cam,
             const uint32_t* tiles, uint32_t tiles_x, ImageOut img) {
  extern __shared__ uint32_t dyn_lds[];
  LdsStack<BLOCK, CSG, UNIFORM, false> st;
  st.init(dyn_lds, S.stack_entries);
  const uint32_t i = (blockIdx.x + S.first_block) * BLOCK + thread_in_block();
  uint32_t slot = 0;
  const bool active = queue_slot(q.counts, q.segcap, i, slot) && i < n;  // n only bounds the grid
  if (!__any(active)) return;
  Counters cs{};
  ShadowQuery SQ;
  SQ.distance = 0.0;
  SQ.found = false;
  if (active) {
    const V3 p = v3(q.px[slot], q.py[slot], q.pz[slot]);
    const V3 v = v3(S.light_position[0], S.light_position[1], S.light_position[2]) - p;
    SQ.distance = magnitude(v);
    const Ray shadow_ray{p, normalize(v)};
    traverse<ShadowQuery, LdsStack<BLOCK, CSG, UNIFORM, false>, COUNT>(S, shadow_ray, SQ, st, cs);
  }
  if (active && SQ.found) {
    const uint32_t rec = q.target[slot];
    const double r = q.ar[slot], gg = q.ag[slot], b = q.ab[slot];
    if (mode == 2 || (mode == 1 && (g.flags[rec] & 4u))) {
      g.cr[rec] = r; g.cg[rec] = gg; g.cb[rec] = b;
    } else {
      uint32_t x = 0, y = 0;
      if (ray_pixel(tiles, tiles_x, cam, rec, x, y)) {
        const size_t p = (size_t)y * cam.hsize + x;
        if (img.rgb64) {
          img.rgb64[p * 3 + 0] = r; img.rgb64[p * 3 + 1] = gg; img.rgb64[p * 3 + 2] = b;
        }
        if (img.rgb8) {
          uint8_t* dst = img.rgb8 + (img.compact ? (size_t)rec : p) * 3;
          dst[0] = quantise(r); dst[1] = quantise(gg); dst[2] = quantise(b);
        }
      }
    }
  }
  if (COUNT) flush_counters(cs, stats + 4);
}

// Launches the kernel of one generation for this combination of template arguments, and after it the kernel
// of the shadow queries it set aside (a.aside.segcap != 0: scenes with bounded cones).  Defined where the
// kernels are instantiated (rt_gen_*.hip); rt_launch.h declares it for the host pipeline.
template <bool COUNT, bool CSG, bool UNIFORM, bool TRIPF>
void launch_generation_variant(const SceneDev& scene, unsigned lds_bytes, int mode, const GenLaunch& a) {
  const dim3 block(BLOCK);
  // a scene whose CSG hit lists live in device memory owns csg_spill_slots of them: its jobs go in pieces
  auto in_pieces = [&](uint32_t blocks, auto&& launch) {
    if (!scene.csg_spill) {
      launch(scene, dim3(blocks));
      return;
    }
    for (uint32_t first = 0; first < blocks; first += scene.csg_spill_slots) {
      SceneDev piece = scene;
      piece.first_block = first;
      launch(piece, dim3(std::min(scene.csg_spill_slots, blocks - first)));
    }
  };
#define RT_GEN_ARGS dev, a.cur, a.n, a.next, a.next_segcap, a.remaining, a.next_counts, a.stats, a.cam, a.tiles, a.tiles_x, a.img, a.queue_counts, a.segcap, a.aside, a.batch
  const uint32_t frames = mode == 0 && a.batch.frames > 1 ? a.batch.frames : 1u;  // a.n rays each
  in_pieces((a.n + BLOCK - 1) / BLOCK * frames, [&](const SceneDev& dev, dim3 grid) {
    if constexpr (TRIPF) {  // only camera rays of scenes that spawn nothing fetch triangles ahead
      hipLaunchKernelGGL((k_generation<COUNT, 0, CSG, UNIFORM, true>), grid, block, lds_bytes, a.stream, RT_GEN_ARGS);
    } else {
      switch (mode) {
        case 0: hipLaunchKernelGGL((k_generation<COUNT, 0, CSG, UNIFORM>), grid, block, lds_bytes, a.stream, RT_GEN_ARGS); break;
        case 1: hipLaunchKernelGGL((k_generation<COUNT, 1, CSG, UNIFORM>), grid, block, lds_bytes, a.stream, RT_GEN_ARGS); break;
        default: hipLaunchKernelGGL((k_generation<COUNT, 2, CSG, UNIFORM>), grid, block, lds_bytes, a.stream, RT_GEN_ARGS); break;
      }
    }
  });
#undef RT_GEN_ARGS
  if (a.aside.segcap != 0u) {  // the grid covers the queue's room; waves beyond its end exit at once
    const uint32_t room = a.aside.segcap * QUEUE_SEGMENTS;
    in_pieces((room + BLOCK - 1) / BLOCK, [&](const SceneDev& dev, dim3 grid) {
      hipLaunchKernelGGL((k_shadow<COUNT, CSG, UNIFORM>), grid, block, lds_bytes, a.stream, dev, a.aside, room, a.cur, mode, a.stats,
                         a.cam, a.tiles, a.tiles_x, a.img);
    });
  }
}

}  // namespace rt
