// World::intersect (src/world.rs:107-117) over the acceleration structure: per-lane stack in LDS,
// conservative f32 box tests, CSG operator trees, the cones' direction index, the 4-wide BVH walk.
#pragma once
#include "band_grid_test.h"
#include "rt_primitives.h"
#include "rt_queries.h"

namespace rt {

// Per-lane stack in LDS: entry e of thread t lives at stack[e * BLOCK + t], so
// the 64 lanes of a wave hit 64 consecutive banks.  The LDS is dynamic: a launch asks for as many entries as
// the scene's hierarchy can need, so a shallow scene leaves the LDS to more waves and a deep one still fits.
template <int BLOCK, bool CSG = false, bool UNIFORM = false, bool TRIPF = false>
struct LdsStack {
  static constexpr bool WITH_CSG = CSG;  // the kernels built for scenes with CSG operators carry eval_csg
  static constexpr bool UNIFORM_NODES = UNIFORM;  // ... and those for large hierarchies the scalar node path
  static constexpr bool TRI_PREFETCH = TRIPF;  // ... and those for large meshes fetch a run's next triangle early
  static constexpr int THREADS = BLOCK;
  uint32_t* base;
  uint32_t* csg;  // this lane's column of the CSG hit list (behind the stack), or null when the scene has no CSG
  int sp;
  int entries;    // per lane: what the scene's hierarchy can need (SceneDev::stack_entries), sized at launch
  // lds: the workgroup's dynamic LDS, `entries` x BLOCK words of stack, then the CSG hit list
  DEV void init(uint32_t* lds, int n_entries) {
    const uint32_t tid = (uint32_t)__lane_id();  // one wave per workgroup
    base = lds + tid;
    csg = CSG ? lds + (size_t)n_entries * BLOCK + tid : nullptr;
    sp = 0;
    entries = n_entries;
  }
  DEV void push(uint32_t r) {
    if (sp < entries) base[(sp++) * BLOCK] = r;
  }
  DEV void push_if(uint32_t r, bool take) {
    base[sp * BLOCK] = r;
    sp += take ? 1 : 0;
  }
  DEV uint32_t pop() { return base[(--sp) * BLOCK]; }
  DEV uint32_t peek(int i) const { return base[i * BLOCK]; }  // entry i of this lane's column, used as a queue
  DEV void poke(int i, uint32_t r) { base[i * BLOCK] = r; }
  DEV uint32_t pop_or_none() { return sp > 0 ? base[(--sp) * BLOCK] : REF_NONE; }
  DEV bool empty() const { return sp == 0; }
  DEV void clear() { sp = 0; }
};

// What a ray needs for the f32 box tests.  Per axis t(plane) = (plane - origin) / direction; the
// near plane of an axis is the lower one when the direction is positive.  Only t >= 0 matters
// (queries behind the origin walk the reversed ray), so a test stays conservative when every near
// value is computed too small and every far value too large:
//   inv          1 / direction from v_rcp_f32 on the direction rounded to f32 (relative error below
//                2^-22); origin * inv is formed with that same inv, so the error of inv only scales t
//   near / far   inv and origin * inv scaled by (1 -/+ 1.5e-6), which covers the error of inv, the
//                rounding of the scaled values and that of the box test's multiply-add (2^-24 |t| each)
//   c_n / c_f    additionally moved outward by 1e-6 |origin * inv|: covers the f32 roundings of the
//                origin, of origin * inv and of the scaled value, which are relative to
//                |origin * inv|, not to t (2^-24 each, 2^-22 in all)
// A box test can therefore pass wrongly but never fail wrongly.
struct RayBox {
  float inv_n[3], c_n[3], inv_f[3], c_f[3];
  uint32_t off_near[3];  // byte offset of the near plane of each axis in a BvhNode; far = near ^ 16
  bool tiny;             // some |direction| component is below FEQ_EPSILON
};

// Rounding of the reference's quadratic for round shapes, relative to |object-space origin|^2 (derived at
// round_shape_noise() below), with a factor of four in hand.
constexpr double ROUND_NOISE = 4.0 * 6.6e-16;

// widen: a distance (world units) by which every box is to be taken larger for this ray, see round_shape_noise().
DEV RayBox make_raybox(const Ray& r, bool reversed, float widen = 0.0f) {
  RayBox b;
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  b.tiny = false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double dir = reversed ? -d[a] : d[a];
    if (fabs(dir) < FEQ_EPSILON) b.tiny = true;
    float dirf = (float)dir;
    if (!(fabsf(dirf) >= 1e-20f)) dirf = copysignf(1e-20f, dirf);  // parallel: a huge finite slope, no inf * 0
    // Everything below in single precision.  inv is within 2^-22 of 1 / direction (rounding of the
    // direction, v_rcp_f32), oinv within 2^-22 |oinv| of origin / direction (rounding of the origin,
    // of inv, of the product); scaling by (1 -/+ 1.5e-6) covers the first and the roundings of the
    // scaled values and of the box test's multiply-add, the slack of 1e-6 |oinv| the second.
    const float inv = __builtin_amdgcn_rcpf(dirf);
    const float oinv = (float)o[a] * inv;
    const float slack = fabsf(oinv) * 1e-6f + 1e-30f + widen * fabsf(inv) * (1.0f + 1e-6f);
    const float lo_s = 1.0f - 1.5e-6f, hi_s = 1.0f + 1.5e-6f;
    b.inv_n[a] = inv * lo_s;
    b.inv_f[a] = inv * hi_s;
    b.c_n[a] = fmaf(oinv, lo_s, slack);
    b.c_f[a] = fmaf(oinv, hi_s, -slack);
    b.off_near[a] = 32u * a + (inv < 0.0f ? 16u : 0u);
  }
  return b;
}

struct F4 {
  float v[4];
};
DEV F4 load_f4(const char* base, uint32_t off) {
  float4 q = *reinterpret_cast<const float4*>(base + off);
  return F4{{q.x, q.y, q.z, q.w}};
}

// The same 16 bytes for a wave-uniform address: read through the constant address space, so the
// load goes through the scalar cache into SGPRs and not through the per-CU vector data path, whose
// return bandwidth (64 B per clock, 1 KB per wave instruction whatever the addresses) bounds the
// node loop.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEV F4 load_f4_uniform(const char* base, uint32_t off) {
  f32x4 q = *(const __attribute__((address_space(4))) f32x4*)(base + off);
  return F4{{q.x, q.y, q.z, q.w}};
}

// Slab tests of the four children of one node: entry distance (INFINITY when missed) per child,
// and the references of missed children cleared.
DEV void slab4(const RayBox& rb, float qhi, const F4& nx, const F4& ny, const F4& nz, const F4& fx, const F4& fy,
               const F4& fz, float tn[4], uint32_t c[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float tnear = fmaxf(fmaxf(fmaf(nx.v[k], rb.inv_n[0], -rb.c_n[0]), fmaf(ny.v[k], rb.inv_n[1], -rb.c_n[1])),
                        fmaf(nz.v[k], rb.inv_n[2], -rb.c_n[2]));
    float tfar = fminf(fminf(fmaf(fx.v[k], rb.inv_f[0], -rb.c_f[0]), fmaf(fy.v[k], rb.inv_f[1], -rb.c_f[1])),
                       fmaf(fz.v[k], rb.inv_f[2], -rb.c_f[2]));
    bool hit = fmaxf(tnear, 0.0f) <= fminf(tfar, qhi);
    tn[k] = hit ? tnear : INFINITY;
    if (!hit) c[k] = REF_NONE;
  }
}

struct __attribute__((packed, aligned(4))) StripWords {  // four strip words of a cell's list, at any word address
  uint32_t x, y, z, w;
};

// The intersections of one primitive with a world-space ray, its enclosing groups' gates included when
// asked for: how many (0..4), written to t in the reference's emission order.
template <bool GENERAL>
DEV int eval_quad(const SceneDev& S, const QuadPrim& q, const Ray& ray, double t[4], bool check_gates) {
  if (check_gates) {
    for (int g = q.gate; g >= 0; g = S.gates[g].parent)
      if (!gate_passes<GENERAL>(S, S.gates[g], ray)) return 0;
  }
  if (q.type == RT_PLANE) {
    // only the y row of the transformed ray enters intersect_plane (src/shape.rs:891-908)
    const double oy = q.m[4] * ray.o.x + q.m[5] * ray.o.y + q.m[6] * ray.o.z + q.m[7];
    const double dy = q.m[4] * ray.d.x + q.m[5] * ray.d.y + q.m[6] * ray.d.z;
    if (fabs(dy) <= FEQ_EPSILON) return 0;
    t[0] = -oy / dy;
    return 1;
  }
  Ray local = quad_ray<GENERAL>(S, q, ray);  // src/shape.rs:1468
  return intersect_quadric(q.type, q.closed, q.minimum, q.maximum, local, t);
}
template <class Query, bool GENERAL>
DEV void test_quad(const SceneDev& S, const QuadPrim& q, const Ray& ray, Query& Q, bool check_gates) {
  double t[4];
  const int n = eval_quad<GENERAL>(S, q, ray, t, check_gates);
  if (n) Q.leaf_ts(q.leaf, n, t);
}

// intersect_csg + filter_intersections (src/shape.rs:1339-1354, src/intersect.rs:278-328) for one
// root operator tree.  The reference intersects both operands, sorts, and lets an intersection
// through depending on which operand it came from and whether the ray is inside the other; nested
// operators filter their own lists first.  Here all leaf intersections below the root go into one
// list sorted by (t, leaf order) -- the order every stable sort of the reference yields -- and the
// operators run innermost first over the entries that are theirs and still alive.
// The list lives in LDS: entry e of a lane at words (3 e + {0, 1, 2}) * BLOCK: t low, t high, leaf slot.  SPILL:
// the scene's lists are longer than the LDS holds; they live in device memory, one per workgroup of the launch,
// and an entry that an operator filtered out is marked in its leaf-slot word instead of a 64-bit mask.
template <class Query, class Stack, int BLOCK_, bool SPILL>
DEV void eval_csg_list(const SceneDev& S, uint32_t root_index, const Ray& ray, Query& Q, uint32_t* list, bool outer_gates) {
  const CsgRoot R = S.csg_roots[root_index];
  if (!list) return;
  if (outer_gates)
    for (int g = R.gate; g >= 0; g = S.gates[g].parent)
      if (!gate_passes<true>(S, S.gates[g], ray)) return;
  constexpr uint32_t DEAD = 0x80000000u;
  auto put = [&](int e, double t, uint32_t who) {
    list[(3 * e + 0) * BLOCK_] = (uint32_t)__double2loint(t);
    list[(3 * e + 1) * BLOCK_] = (uint32_t)__double2hiint(t);
    list[(3 * e + 2) * BLOCK_] = who;
  };
  auto get_t = [&](int e) { return __hiloint2double((int)list[(3 * e + 1) * BLOCK_], (int)list[(3 * e + 0) * BLOCK_]); };
  auto get_who = [&](int e) { return list[(3 * e + 2) * BLOCK_] & ~DEAD; };
  int n = 0;
  for (uint32_t j = 0; j < R.n_leaves; j++) {
    const CsgLeaf L = S.csg_leaves[R.first_leaf + j];
    bool open = true;
    for (int g = L.gate; g != R.gate && g >= 0; g = S.gates[g].parent)  // groups between the root and the leaf
      if (!gate_passes<true>(S, S.gates[g], ray)) {
        open = false;
        break;
      }
    if (!open) continue;
    double t[4];
    int m = 0;
    const uint32_t idx = L.ref & MAX_RUN_FIRST;
    if ((L.ref >> 30) == KIND_TRI) {
      const TriIsect& tr = S.tris[idx];
      double u, v;
      bool hit = tr.xform >= 0 ? intersect_triangle(tr, tri_ray<true>(S, tr, ray), t[0], u, v)
                               : intersect_triangle(tr, ray, t[0], u, v);
      m = hit ? 1 : 0;
    } else {
      const QuadPrim& q = S.quads[idx];
      m = intersect_quadric(q.type, q.closed, q.minimum, q.maximum, quad_ray<true>(S, q, ray), t);
    }
    for (int k = 0; k < m; k++) {
      if (!(t[k] == t[k]) || n >= S.csg_hits) continue;  // NaN never orders (the reference's sort would panic)
      int e = n;
      while (e > 0 && get_t(e - 1) > t[k]) {  // stable: equal t stays behind earlier leaves
        put(e, get_t(e - 1), get_who(e - 1));
        e--;
      }
      put(e, t[k], j);
      n++;
    }
  }
  if (n == 0) return;
  static_assert(CSG_MAX_HITS <= 64, "one bit of `alive` per list entry while the list is in LDS");
  unsigned long long alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);  // !SPILL
  int n_alive = n;                                                     // SPILL
  auto is_alive = [&](int e) { return SPILL ? !(list[(3 * e + 2) * BLOCK_] & DEAD) : (bool)((alive >> e) & 1ull); };
  auto kill = [&](int e) {
    if constexpr (SPILL) {
      list[(3 * e + 2) * BLOCK_] |= DEAD;
      n_alive--;
    } else {
      alive &= ~(1ull << e);
    }
  };
  for (int c = (int)R.n_nodes - 1; c >= 0; c--) {  // children carry larger indices than their parents
    const int op = S.csg_nodes[R.first_node + c].op;
    bool inl = false, inr = false;
    for (int e = 0; e < n; e++) {
      if (!is_alive(e)) continue;
      const CsgLeaf& L = S.csg_leaves[R.first_leaf + get_who(e)];
      int a = L.node, level = 0;
      while (a > c) {
        a = S.csg_nodes[R.first_node + a].parent;
        level++;
      }
      if (a != c) continue;  // not below this operator
      const bool lhit = (L.left_mask >> level) & 1u;
      bool allowed;  // intersection_allowed (src/intersect.rs:278-290)
      if (op == RT_CSG_UNION) allowed = lhit ? !inr : !inl;
      else if (op == RT_CSG_INTERSECTION) allowed = lhit ? inr : inl;
      else allowed = lhit ? !inr : inl;
      if (lhit) inl = !inl;
      else inr = !inr;
      if (!allowed) kill(e);
    }
  }
  if (SPILL ? n_alive == 0 : alive == 0ull) return;
  for (uint32_t j = 0; j < R.n_leaves; j++) {
    double t[4];
    int m = 0;
    for (int e = 0; e < n; e++)
      if (is_alive(e) && get_who(e) == j && m < 4) t[m++] = get_t(e);
    if (!m) continue;
    const CsgLeaf L = S.csg_leaves[R.first_leaf + j];
    const uint32_t idx = L.ref & MAX_RUN_FIRST;
    if ((L.ref >> 30) == KIND_TRI) {
      const TriIsect& tr = S.tris[idx];
      double tt, u, v;  // again, for the barycentric coordinates of a smooth triangle
      bool hit = tr.xform >= 0 ? intersect_triangle(tr, tri_ray<true>(S, tr, ray), tt, u, v)
                               : intersect_triangle(tr, ray, tt, u, v);
      if (hit) Q.leaf_tri(tr.leaf, tt, u, v);
    } else {
      Q.leaf_ts(S.quads[idx].leaf, m, t);
    }
  }
}

template <class Query, class Stack, int BLOCK_>
DEV void eval_csg(const SceneDev& S, uint32_t root_index, const Ray& ray, Query& Q, Stack& st, bool outer_gates) {
  if (S.csg_spill) {  // this workgroup's list (one wave per workgroup): csg_hits entries of 3 x BLOCK words
    uint32_t* list = S.csg_spill + (size_t)blockIdx.x * (size_t)S.csg_hits * 3u * BLOCK_ + __lane_id();
    eval_csg_list<Query, Stack, BLOCK_, true>(S, root_index, ray, Q, list, outer_gates);
  } else {
    eval_csg_list<Query, Stack, BLOCK_, false>(S, root_index, ray, Q, st.csg, outer_gates);
  }
}

// World::intersect (src/world.rs:107-117) over the acceleration structure: first the primitives
// without a finite box, then the 4-wide BVH.  While-while traversal: every lane first descends
// inner nodes (the nearest hit child stays in a register, the others go on the LDS stack), and
// only when all lanes of the wave have reached a primitive run or a gate is that work done.
#ifndef RT_BAND_BATCH
#define RT_BAND_BATCH 2  // cone-band list entries fetched together
#endif
template <class Query, class Stack, bool COUNT>
DEV void traverse(const SceneDev& S, const Ray& ray, Query& Q, Stack& st, Counters& cnt) {
  RT_TBEGIN(cnt);
#pragma unroll
  for (int i = 0; i < MAX_INLINE_PLANES; i++) {
    if (i < S.n_inline_planes) {  // intersect_plane (src/shape.rs:891-908) on the y row of the transformed ray
      const InlinePlane& P = S.inline_planes[i];
      if (COUNT) cnt.quad_tests++;
      const double oy = P.row[0] * ray.o.x + P.row[1] * ray.o.y + P.row[2] * ray.o.z + P.row[3];
      const double dy = P.row[0] * ray.d.x + P.row[1] * ray.d.y + P.row[2] * ray.d.z;
      if (!(fabs(dy) <= FEQ_EPSILON)) {
        double t[1] = {-oy / dy};
        Q.leaf_ts(P.leaf, 1, t);
      }
    }
  }
  for (int i = 0; i < S.n_unbounded; i++) {
    if (COUNT) cnt.quad_tests++;
    test_quad<Query, Stack::WITH_CSG>(S, S.unbounded[i], ray, Q, true);
  }
  if constexpr (Stack::WITH_CSG)
    for (int i = 0; i < S.n_csg_unbounded; i++) eval_csg<Query, Stack, Stack::THREADS>(S, S.csg_unbounded[i], ray, Q, st, true);
  // The cones' parallel-ray branch (src/shape.rs:1089-1103): the cones whose band holds the ray's direction.
  auto band_list = [&]() {
    const double dd = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
    if (S.band_grid_n > 0 && dd >= 1.0 - 1e-6) {
      // cell of the direction on the sign-folded cube map; the grid was built for |d| >= 1 - 1e-6
      const int n = S.band_grid_n;
      const float fx = (float)ray.d.x, fy = (float)ray.d.y, fz = (float)ray.d.z;
      const float fxx = fx * fx, fyy = fy * fy, fzz = fz * fz, fxy = fx * fy, fxz = fx * fz, fyz = fy * fz;
      int m = fzz > fyy ? 2 : 1;
      m = fxx >= (m == 2 ? fzz : fyy) ? 0 : m;
      const float dm = m == 0 ? fx : (m == 1 ? fy : fz);
      const float da = m == 0 ? fy : (m == 1 ? fz : fx);
      const float dc = m == 0 ? fz : (m == 1 ? fx : fy);
      const float inv = __builtin_amdgcn_rcpf(dm);
      const float half_n = 0.5f * (float)n;
      int iu, iv;
      float ls, lt;  // where in its cell the direction lies, -1..1 per axis
      band_strip_coords((da * inv + 1.0f) * half_n, (dc * inv + 1.0f) * half_n, n, iu, iv, ls, lt);
      const uint32_t cell = ((uint32_t)m * n + iv) * n + iu;
      // A cell lists every cone whose band crosses it, and with each entry a word that says where in the
      // cell the band lies (a strip, band_grid_test.h).  Three passes, each over all lanes in step, because
      // what is rare for one ray happens in every wave:
      //   1. the strip words of the cell, four per load from one contiguous array: one multiply-add and a
      //      compare per entry, no look-up; the few entries whose strip holds the direction are queued on
      //      the traversal stack (free until the hierarchy is walked);
      //   2. the queued entries: band number, then the band's quadratic form in single precision (it passes
      //      slightly more than the double-precision band); survivors stay queued as primitive numbers;
      //   3. the cone's own test (src/shape.rs:1089) for what is left: it runs as often as the fullest
      //      queue is long, not once per entry some lane happens to pass.  (Dealing these tests evenly over
      //      the lanes of the wave through LDS was measured and is no faster: with 25 000 cones the tests
      //      are 1.3 ms of a 15.7 ms frame.)
      const uint32_t end = S.band_start[cell + 1];
#ifdef RT_DIAG_NO_BANDS  // timing builds (wrong frames): the passes below stop after the cell look-up (0) .. run in full (3)
      const int band_passes = S.band_diag_passes;
#else
      constexpr int band_passes = 3;
#endif
      const float fdd = (fxx + fyy + fzz) * (1.0f + 1e-6f);
      const uint16_t* items16 = reinterpret_cast<const uint16_t*>(S.band_items);
      st.clear();
      auto flush = [&]() {
        if (band_passes <= 1) { st.sp = 0; return; }
        int wp = 0;  // survivors are written over the entries already read (wp <= rp)
        for (int rp = 0; __any(rp < st.sp); rp += RT_BAND_BATCH) {
          uint32_t id[RT_BAND_BATCH];
          ConeBandF32 f[RT_BAND_BATCH];
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) {
            const uint32_t at = rp + j < st.sp ? st.peek(rp + j) : 0u;  // entry 0 exists: the index is not empty
            id[j] = S.band_items16 ? (uint32_t)items16[at] : S.band_items[at];
          }
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) f[j] = S.bands32[id[j]];
#pragma unroll
          for (int j = 0; j < RT_BAND_BATCH; j++) {
            const float a = fmaf(f[j].q[0], fxx, fmaf(f[j].q[1], fyy, fmaf(f[j].q[2], fzz,
                            fmaf(f[j].q[3], fxy, fmaf(f[j].q[4], fxz, f[j].q[5] * fyz)))));
            if (rp + j < st.sp && fabsf(a) < 1.0001e-4f + f[j].slack * fdd) st.poke(wp++, f[j].prim);
          }
        }
        st.sp = wp;
        if (band_passes <= 2) st.sp = 0;
#ifdef RT_DIAG_NO_BANDS
        Query dropped = Q;  // RT_AMD_TIMING_BAND_PASSES=4: the tests run, their intersections are dropped
#endif
        while (__any(st.sp > 0)) {
          if (st.sp > 0) {
            const uint32_t prim = st.pop();
            if (COUNT) cnt.quad_tests++;
#ifdef RT_DIAG_NO_BANDS
            if (band_passes == 4) test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[prim], ray, dropped, true);
            else
#endif
            test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[prim], ray, Q, true);
          }
        }
      };
      const uint32_t begin = S.band_start[cell];
      uint32_t stop = end;
      if (band_passes == 0 && end != 0xFFFFFFFFu) stop = begin;  // the offsets are still fetched
      // all lanes stay in the loop until the longest list is through: flush() works on the whole wave
      const float ls1024 = ls * 1024.0f, lt1024 = lt * 1024.0f;
      auto words_at = [&](uint32_t k) {  // four strip words of the list, or padding behind its end (four words)
        return *reinterpret_cast<const StripWords*>(S.band_strip + (k < stop ? k : stop));
      };
      StripWords next = words_at(begin);
      for (uint32_t k = begin; __any(k < stop); k += 4) {
        const StripWords w = next;
        next = words_at(k + 4);  // in flight while these four are looked at
        if (__any(st.sp + 4 > st.entries)) flush();
        st.push_if(k, k < stop && band_strip_passes(w.x, ls, lt, ls1024, lt1024));  // room for four was made above
        st.push_if(k + 1, k + 1 < stop && band_strip_passes(w.y, ls, lt, ls1024, lt1024));
        st.push_if(k + 2, k + 2 < stop && band_strip_passes(w.z, ls, lt, ls1024, lt1024));
        st.push_if(k + 3, k + 3 < stop && band_strip_passes(w.w, ls, lt, ls1024, lt1024));
      }
      flush();
    } else {
      const double xx = ray.d.x * ray.d.x, yy = ray.d.y * ray.d.y, zz = ray.d.z * ray.d.z;
      const double xy = ray.d.x * ray.d.y, xz = ray.d.x * ray.d.z, yz = ray.d.y * ray.d.z;
      for (int i = 0; i < S.n_bands; i++) {
        const ConeBand& cb = S.bands[i];
        double a = fma(cb.q[0], xx, fma(cb.q[1], yy, fma(cb.q[2], zz, fma(cb.q[3], xy, fma(cb.q[4], xz, cb.q[5] * yz)))));
        if (fabs(a) < FEQ_EPSILON * (1.0 + 1e-6) + cb.slack * dd) {
          if (COUNT) cnt.quad_tests++;
          test_quad<Query, Stack::WITH_CSG>(S, S.band_prims[cb.prim], ray, Q, true);
        }
      }
    }
  };
  // The order in which a query meets the shapes does not change its answer; the bands come first because
  // their hits shorten the walk.  (A shadow query that walks the hierarchy first and looks the bands up
  // only when it found nothing was measured: correct, but the shadow ray of a hit far out on a cone's
  // extension runs back along that cone, and the band list ends it at once.)
  if (S.n_bands > 0) {
    band_list();
    Q.after_bands();
  }
  auto hierarchy = [&]() {
  if (S.root == REF_NONE || Q.done()) return;
  const char* nodes = reinterpret_cast<const char*>(S.nodes);
  // The reference finds the roots of a round shape in object space, where b^2 - 4ac cancels for a far
  // origin: with |origin|^2 = 1e14 a unit sphere's silhouette moves by a hundredth of its radius, and the
  // reference's hits then lie outside the shape's box.  The boxes are taken larger by that much for this
  // ray (nothing measurable for a ray that starts within 1e5 units of the scene).
  float widen = 0.0f;
  if (S.box_noise[0] > 0.0 || S.box_noise[1] > 0.0) {
    const double q = 2.0 * (ray.o.x * ray.o.x + ray.o.y * ray.o.y + ray.o.z * ray.o.z + S.box_noise[2]);
    if (q > 1e9) {  // within 2e4 units the rounding stays below the boxes' own pads on every scene tested
      const double w = 0.5 * ROUND_NOISE * q * S.box_noise[0] + sqrt(ROUND_NOISE * q) * S.box_noise[1];
      widen = w < 1e30 ? __double2float_ru(w) : 1e30f;  // NaN origins: the box tests see NaN either way
    }
  }
  const RayBox rb = make_raybox(ray, Query::BEHIND, widen);
  auto hi_f = [&]() { return __double2float_ru(Q.hi_d()); };  // the query's far bound, rounded up
  float qhi = hi_f();
  {  // The box of the whole hierarchy comes with the kernel arguments: a ray that misses it (most rays of a
     // floor) is done without fetching the root node.  Same test and constants as slab4, on a box that holds
     // every child box, so it fails only where every child's test would.
    float tnear = -INFINITY, tfar = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const bool neg = (rb.off_near[a] & 16u) != 0u;
      tnear = fmaxf(tnear, fmaf(neg ? S.scene_box[3 + a] : S.scene_box[a], rb.inv_n[a], -rb.c_n[a]));
      tfar = fminf(tfar, fmaf(neg ? S.scene_box[a] : S.scene_box[3 + a], rb.inv_f[a], -rb.c_f[a]));
    }
    if (!(fmaxf(tnear, 0.0f) <= fminf(tfar, qhi))) return;
  }
  st.clear();
  // rays without a near-zero direction component read the child references in which skippable
  // gates are already replaced by their subtrees
  const uint32_t child_off = rb.tiny ? 96u : 112u;
  uint32_t cur = rb.tiny ? S.root : S.root_alt;
  // do all lanes that traverse pick the same near planes and the same child list?
  uint32_t un[3] = {0, 0, 0}, uchild = 0;
  bool same_signs = false;
  if constexpr (Stack::UNIFORM_NODES) {
    for (int a = 0; a < 3; a++) un[a] = (uint32_t)__builtin_amdgcn_readfirstlane(rb.off_near[a]);
    uchild = __builtin_amdgcn_readfirstlane(child_off);
    same_signs = __builtin_amdgcn_ballot_w64(rb.off_near[0] != un[0] || rb.off_near[1] != un[1] ||
                                             rb.off_near[2] != un[2] || child_off != uchild) == 0;
  }
  RT_TSTAMP(cnt, 0);
  while (cur != REF_NONE) {
    while ((cur >> 30) == KIND_INNER) {
      const uint32_t nd = cur << 7;  // byte offset of the node (inner kind bits are zero)
      if (COUNT) cnt.node_visits++;
      RT_WAVE_ITER(cnt, 0);
      float tn[4];
      uint32_t c[4];
      bool together = false;
      uint32_t ucur = 0;
      if constexpr (Stack::UNIFORM_NODES) {
        ucur = __builtin_amdgcn_readfirstlane(cur);
        together = same_signs && __builtin_amdgcn_ballot_w64(cur != ucur) == 0;
      }
      if (together) {
        // every lane still in the loop is at this node and picks the same near / far planes
        const char* np = nodes + ((size_t)ucur << 7);
        F4 nx = load_f4_uniform(np, un[0]), ny = load_f4_uniform(np, un[1]), nz = load_f4_uniform(np, un[2]);
        F4 fx = load_f4_uniform(np, un[0] ^ 16u), fy = load_f4_uniform(np, un[1] ^ 16u), fz = load_f4_uniform(np, un[2] ^ 16u);
        u32x4 ch = *(const __attribute__((address_space(4))) u32x4*)(np + uchild);
        c[0] = ch.x; c[1] = ch.y; c[2] = ch.z; c[3] = ch.w;
        slab4(rb, qhi, nx, ny, nz, fx, fy, fz, tn, c);
      } else {
        F4 nx = load_f4(nodes, nd + rb.off_near[0]), ny = load_f4(nodes, nd + rb.off_near[1]),
           nz = load_f4(nodes, nd + rb.off_near[2]);
        F4 fx = load_f4(nodes, nd + (rb.off_near[0] ^ 16u)), fy = load_f4(nodes, nd + (rb.off_near[1] ^ 16u)),
           fz = load_f4(nodes, nd + (rb.off_near[2] ^ 16u));
        uint4 ch = *reinterpret_cast<const uint4*>(nodes + nd + child_off);
        c[0] = ch.x; c[1] = ch.y; c[2] = ch.z; c[3] = ch.w;
        slab4(rb, qhi, nx, ny, nz, fx, fy, fz, tn, c);
      }
      // nearest hit child stays in a register, the other hit children go on the stack
      bool lt = tn[1] < tn[0];
      float best = lt ? tn[1] : tn[0];
      uint32_t bc = lt ? c[1] : c[0], o1 = lt ? c[0] : c[1];
      lt = tn[2] < best;
      best = lt ? tn[2] : best;
      uint32_t o2 = lt ? bc : c[2];
      bc = lt ? c[2] : bc;
      lt = tn[3] < best;
      uint32_t o3 = lt ? bc : c[3];
      bc = lt ? c[3] : bc;
      st.push_if(o1, o1 != REF_NONE);
      st.push_if(o2, o2 != REF_NONE);
      st.push_if(o3, o3 != REF_NONE);
      cur = bc != REF_NONE ? bc : st.pop_or_none();
    }
    RT_TSTAMP(cnt, 1);
    RT_WAVE_ITER(cnt, 3);  // rounds of the outer loop
    if (cur == REF_NONE) break;
    const uint32_t kind = cur >> 30;
    if (kind == KIND_GATE) {
      const Gate& g = S.gates[cur & 0x3FFFFFFFu];
      if (g.csg) {
        if constexpr (Stack::WITH_CSG) eval_csg<Query, Stack, Stack::THREADS>(S, g.csg - 1u, ray, Q, st, false);
        if (Q.done()) break;
        qhi = hi_f();
        cur = st.pop_or_none();
        RT_TSTAMP(cnt, 3);
        continue;
      }
      bool pass;
      if (g.skippable && !rb.tiny) {
        pass = true;
      } else {
        if (COUNT) cnt.gate_tests++;
        pass = gate_passes<Stack::WITH_CSG>(S, g, ray);
      }
      cur = pass ? g.root : st.pop_or_none();
      RT_TSTAMP(cnt, 3);
      continue;
    }
    const uint32_t first = cur & MAX_RUN_FIRST, count = ((cur >> 27) & 7u) + 1u;
    if (kind == KIND_TRI) {
      int32_t local_of = -1;  // the transform `local` was made with: the triangles of a mesh share one
      Ray local = ray;
      // Kernels for large meshes hold the next record of the run in registers while this one is tested.
      TriIsect ahead;
      if constexpr (Stack::TRI_PREFETCH) ahead = S.tris[first];
      for (uint32_t i = 0; i < count; i++) {
        TriIsect held;
        if constexpr (Stack::TRI_PREFETCH) {
          held = ahead;
          if (i + 1 < count) ahead = S.tris[first + i + 1];
        }
        const TriIsect& tr = Stack::TRI_PREFETCH ? held : S.tris[first + i];
        if (COUNT) cnt.tri_tests++;
        RT_WAVE_ITER(cnt, 1);
        double t, u, v;
        bool hit;
        if (tr.xform >= 0) {
          if (tr.xform != local_of) {  // the same ray, the same matrix: the same object-space ray, made once
            local = tri_ray<Stack::WITH_CSG>(S, tr, ray);
            local_of = tr.xform;
          }
          hit = intersect_triangle(tr, local, t, u, v);
        } else {
          hit = intersect_triangle(tr, ray, t, u, v);
        }
        if (hit) Q.leaf_tri(tr.leaf, t, u, v);
      }
      RT_TSTAMP(cnt, 2);
    } else {
      for (uint32_t i = 0; i < count; i++) {
        if (COUNT) cnt.quad_tests++;
        RT_WAVE_ITER(cnt, 2);
        test_quad<Query, Stack::WITH_CSG>(S, S.quads[first + i], ray, Q, false);
      }
      RT_TSTAMP(cnt, 3);
    }
    if (Q.done()) return;
    qhi = hi_f();
    cur = st.pop_or_none();
  }
  };
  hierarchy();
}

}  // namespace rt
