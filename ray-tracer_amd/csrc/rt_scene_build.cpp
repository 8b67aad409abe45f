// Host side of the scene object: flattening a description into the device tables (boxes, hierarchy,
// CSG records, cone bands, shading tables), rt_scene_create / _update / _clone / _destroy.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "band_grid.h"
#include "bvh_build.h"
#include "rt_error.h"
#include "rt_scene.h"

using namespace rt;

bool rt::affine(const double* m) {
  return fabs(m[12]) < 1e-9 && fabs(m[13]) < 1e-9 && fabs(m[14]) < 1e-9 && fabs(m[15] - 1.0) < 1e-9;
}

namespace {

// `out` points at one of the pointers inside s->dev
template <class T>
int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
  *out = nullptr;
  if (v.empty()) return RT_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  s->allocs.push_back(rt_scene::Alloc{p, v.size() * sizeof(T), (size_t)((const char*)out - (const char*)&s->dev)});
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  s->scene_bytes += v.size() * sizeof(T);
  *out = (const T*)p;
  return RT_OK;
}

bool is_identity(const double* m) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++)
      if (m[r * 4 + c] != (r == c ? 1.0 : 0.0)) return false;  // -0.0 == 0.0
  return true;
}

void box_of_points(const double* fwd, const double (*pts)[3], int n, double lo[3], double hi[3]) {
  for (int a = 0; a < 3; a++) {
    lo[a] = INFINITY;
    hi[a] = -INFINITY;
  }
  for (int i = 0; i < n; i++) {
    for (int a = 0; a < 3; a++) {
      double v = fwd[a * 4 + 0] * pts[i][0] + fwd[a * 4 + 1] * pts[i][1] + fwd[a * 4 + 2] * pts[i][2] + fwd[a * 4 + 3];
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
  }
  // Pad: the reference finds hits by rounded object-space arithmetic, so a hit can sit a few
  // ulps outside the exact shape; the pad also absorbs the rounding of the slab test.
  for (int a = 0; a < 3; a++) {
    double pad = 1e-7 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-9;
    lo[a] -= pad;
    hi[a] += pad;
  }
}

// Rounding of the reference's quadratic (src/shape.rs:848-865, 987-1010, 1077-1112) for a ray whose
// object-space origin o is far from the shape: b = 2 o.d, c = o.o - 1 and b^2 - 4ac each carry a few
// relative roundings of |o|^2 |d|^2, about 24 * 1.1e-16 |o|^2 |d|^2 in the discriminant, which is
// 4 |d|^2 (1 - p^2) for a line at distance p from the centre (axis): the reference reports hits up to
// p^2 <= 1 + 6.6e-16 |o|^2.  ROUND_NOISE is that with a factor of four in hand.  For a shape with
// object -> world matrix F and world -> object matrix M, |o| <= |M| (|origin| + |centre|) and a distance
// in object space is at most |F| times as long in world space, so the silhouette moves by at most
//   sphere, cylinder (radius 1):  |F| (sqrt(1 + x) - 1) <= |F| x / 2,   x = ROUND_NOISE |M|^2 q,
//   cone (radius |y|, down to 0): |F| sqrt(x),
// with q = 2 (|origin|^2 + |centre|^2) and Frobenius norms.  round_shape_noise() takes the largest
// |F| |M|^2, |F| |M| and |centre|^2 over the scene's round shapes; the traversal widens its box tests by
// the two terms for every ray (rt_types.h: SceneDev::box_noise).
void round_shape_noise(const rt_scene_desc* d, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (int32_t i = 0; i < d->n_leaves; i++) {
    const rt_leaf& L = d->leaves[i];
    if (L.type != RT_SPHERE && L.type != RT_CYLINDER && L.type != RT_CONE) continue;
    const rt_xform& X = d->xforms[L.xform];
    double f2 = 0, m2 = 0, c2 = 0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) {
        f2 += X.fwd[r * 4 + c] * X.fwd[r * 4 + c];
        m2 += X.ray_inv[r * 4 + c] * X.ray_inv[r * 4 + c];
      }
      c2 += X.fwd[r * 4 + 3] * X.fwd[r * 4 + 3];
    }
    if (!(f2 < 1e300) || !(m2 < 1e300) || !(c2 < 1e300)) continue;  // degenerate matrices: such shapes have no box
    const double f = sqrt(f2), m = sqrt(m2);
    if (L.type == RT_CONE) out[1] = fmax(out[1], f * m);
    else out[0] = fmax(out[0], f * m2);
    out[2] = fmax(out[2], c2);
  }
}

// World-space box of a leaf, or false when it has to be tested for every ray.
bool leaf_box(const rt_leaf& L, const rt_xform& X, const rt_triangle* tri, double lo[3], double hi[3]) {
  const double* f = X.fwd;
  double scale = 0;
  for (int c = 0; c < 3; c++)
    scale = fmax(scale, sqrt(f[c] * f[c] + f[4 + c] * f[4 + c] + f[8 + c] * f[8 + c]));
  double x0, x1, y0, y1;
  switch (L.type) {
    case RT_SPHERE: x0 = y0 = -1; x1 = y1 = 1; break;
    case RT_CUBE:
      // A cube treats a ray whose object-space direction component is below 1e-4 as parallel to
      // that slab (src/shape.rs:1372-1377); for a very large cube that widens its silhouette by
      // far more than any pad, so such cubes are tested for every ray.
      if (scale > 100.0) return false;
      x0 = y0 = -1 - 1e-3; x1 = y1 = 1 + 1e-3;
      break;
    case RT_CYLINDER:
      if (!std::isfinite(L.minimum) || !std::isfinite(L.maximum)) return false;
      x0 = -1; x1 = 1; y0 = L.minimum; y1 = L.maximum;
      break;
    case RT_CONE: {
      if (!std::isfinite(L.minimum) || !std::isfinite(L.maximum) || scale > 100.0) return false;
      double r = fmax(fabs(L.minimum), fabs(L.maximum));
      x0 = -r; x1 = r; y0 = L.minimum; y1 = L.maximum;
      break;
    }
    case RT_TRIANGLE:
    case RT_SMOOTH_TRIANGLE: {
      double pts[3][3];
      for (int a = 0; a < 3; a++) {
        pts[0][a] = tri->p1[a];
        pts[1][a] = tri->p1[a] + tri->e1[a];
        pts[2][a] = tri->p1[a] + tri->e2[a];
      }
      box_of_points(f, pts, 3, lo, hi);
      return true;
    }
    default: return false;  // plane
  }
  static const bool corner_boxes = getenv("RT_AMD_CORNER_BOXES") != nullptr;  // the looser boxes, for comparison
  if (L.type != RT_CUBE && !corner_boxes) {
    // Round shapes get their own extent along each world axis instead of the box of their object-space
    // box's corners (up to sqrt(3) wider per axis under a rotation).  A sphere reaches
    // sqrt(f_i0^2 + f_i1^2 + f_i2^2) from its centre along axis i.  A cylinder or cone lies in the hull of
    // its two end discs (a cone's lines pass through the apex from one rim to the other), and a disc of
    // radius r at height y reaches r sqrt(f_i0^2 + f_i2^2) from its centre f_i1 y + f_i3.
    const bool sphere = L.type == RT_SPHERE;
    const double ry0 = L.type == RT_CONE ? fabs(y0) : 1.0, ry1 = L.type == RT_CONE ? fabs(y1) : 1.0;
    for (int a = 0; a < 3; a++) {
      const double fx = f[a * 4 + 0], fy = f[a * 4 + 1], fz = f[a * 4 + 2], c = f[a * 4 + 3];
      if (sphere) {
        const double h = sqrt(fx * fx + fy * fy + fz * fz);
        lo[a] = c - h;
        hi[a] = c + h;
      } else {
        const double rho = sqrt(fx * fx + fz * fz);
        lo[a] = c + fmin(fy * y0 - ry0 * rho, fy * y1 - ry1 * rho);
        hi[a] = c + fmax(fy * y0 + ry0 * rho, fy * y1 + ry1 * rho);
      }
      // the same pad as box_of_points, and the rounding of the extent itself
      const double pad = 2e-7 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-9;
      lo[a] -= pad;
      hi[a] += pad;
    }
    return true;
  }
  double pts[8][3];
  for (int k = 0; k < 8; k++) {
    pts[k][0] = (k & 1) ? x1 : x0;
    pts[k][1] = (k & 2) ? y1 : y0;
    pts[k][2] = (k & 4) ? x1 : x0;
  }
  box_of_points(f, pts, 8, lo, hi);
  return true;
}

}  // namespace

// normalize(transposition(full_inv) * vector(ln)) with w dropped, as shape::normal_at forms it
// (src/shape.rs:1481-1483, src/matrix.rs:644-649, src/tuple.rs:111-135)
void rt::world_normal(const double* fi, const double* ln, double* out) {
  const double x = fi[0] * ln[0] + fi[4] * ln[1] + fi[8] * ln[2];
  const double y = fi[1] * ln[0] + fi[5] * ln[1] + fi[9] * ln[2];
  const double z = fi[2] * ln[0] + fi[6] * ln[1] + fi[10] * ln[2];
  const double mag = sqrt(x * x + y * y + z * z);
  out[0] = x * (1.0 / mag);
  out[1] = y * (1.0 / mag);
  out[2] = z * (1.0 / mag);
}

namespace {

// The tables the shading code reads, from a description.
struct ShadingTables {
  std::vector<LeafShade> leaves;
  std::vector<XformDev> xforms;
  std::vector<TriShade> tri_shade;
  std::vector<MaterialDev> mats;
  std::vector<PatternDev> pats;
  int max_children = 0;  // rays a hit can spawn: reflective? + transparent?
  void build(const rt_scene_desc* d) {
    leaves.resize(d->n_leaves);
    for (int i = 0; i < d->n_leaves; i++) {
      const rt_leaf& L = d->leaves[i];
      LeafShade& o = leaves[i];
      o.type = L.type; o.closed = L.closed; o.material = L.material; o.xform = L.xform;
      o.tri = L.tri; o.fixed_normal = 0; o.minimum = L.minimum; o.maximum = L.maximum;
      for (int a = 0; a < 3; a++) o.plane_normal[a] = L.plane_normal[a];
      // normal_at (src/shape.rs:1478-1484) of a shape with one normal: transposition() * local normal, w dropped,
      // normalized -- the operations and their order are those of the generation kernel, which then only loads it
      const bool flat = L.type == RT_TRIANGLE && L.tri >= 0 && L.tri < d->n_triangles;
      if ((L.type == RT_PLANE || flat) && L.xform >= 0 && L.xform < d->n_xforms) {
        const double* ln = flat ? d->triangles[L.tri].normal : L.plane_normal;
        world_normal(d->xforms[L.xform].full_inv, ln, o.plane_normal);
        o.fixed_normal = 1;
      }
    }
    xforms.resize(d->n_xforms);
    for (int i = 0; i < d->n_xforms; i++) {
      copy3x4(xforms[i].ray_inv, d->xforms[i].ray_inv);
      copy3x4(xforms[i].full_inv, d->xforms[i].full_inv);
    }
    tri_shade.resize(d->n_triangles);
    for (int i = 0; i < d->n_triangles; i++) {
      const rt_triangle& T = d->triangles[i];
      for (int a = 0; a < 3; a++) {
        tri_shade[i].normal[a] = T.normal[a];
        tri_shade[i].n1[a] = T.n1[a];
        tri_shade[i].n2[a] = T.n2[a];
        tri_shade[i].n3[a] = T.n3[a];
      }
    }
    mats.resize(d->n_materials);
    bool any_reflect = false, any_refract = false;
    for (int i = 0; i < d->n_materials; i++) {
      const rt_material& M = d->materials[i];
      MaterialDev& o = mats[i];
      for (int a = 0; a < 3; a++) o.color[a] = M.color[a];
      o.ambient = M.ambient; o.diffuse = M.diffuse; o.specular = M.specular; o.shininess = M.shininess;
      o.reflective = M.reflective; o.refractive_index = M.refractive_index; o.transparency = M.transparency;
      o.pattern = M.pattern;
      bool refl = !(fabs(M.reflective - 0.0) < FEQ_EPSILON);      // world.rs:151
      bool refr = !(fabs(M.transparency - 0.0) < FEQ_EPSILON);    // world.rs:190
      bool schlick = M.reflective > 0.0 && M.transparency > 0.0;  // world.rs:237
      o.flags = (refl ? 1 : 0) | (refr ? 2 : 0) | (schlick ? 4 : 0) | ((refr || schlick) ? 8 : 0);
      any_reflect |= refl;
      any_refract |= refr;
    }
    max_children = (any_reflect ? 1 : 0) + (any_refract ? 1 : 0);
    pats.resize(d->n_patterns);
    for (int i = 0; i < d->n_patterns; i++) {
      const rt_pattern& P = d->patterns[i];
      pats[i].type = P.type; pats[i].first = P.first; pats[i].second = P.second; pats[i].pad = 0;
      copy3x4(pats[i].inv, P.inv);
      for (int a = 0; a < 3; a++) {
        pats[i].a[a] = P.a[a];
        pats[i].b[a] = P.b[a];
      }
    }
  }
};

// The direction pre-filter of a bounded cone's parallel-ray branch (ConeBand) from its world -> object
// matrix: a(d) = |M d|^2 under diag(1, -1, 1), M = the linear part.
ConeBand make_cone_band(const QuadPrim& q, uint32_t prim) {
  const double* m = q.m;
  const double sg[3] = {1.0, -1.0, 1.0};
  double Q[3][3], sum = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Q[i][j] = 0.0;
      for (int r = 0; r < 3; r++) Q[i][j] += sg[r] * m[r * 4 + i] * m[r * 4 + j];
    }
  ConeBand cb{};
  cb.q[0] = Q[0][0]; cb.q[1] = Q[1][1]; cb.q[2] = Q[2][2];
  cb.q[3] = 2.0 * Q[0][1]; cb.q[4] = 2.0 * Q[0][2]; cb.q[5] = 2.0 * Q[1][2];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) sum += m[r * 4 + c] * m[r * 4 + c];
  cb.slack = 1e-11 * sum;
  cb.prim = prim;
  return cb;
}

// Single-precision copies: the direction's components are rounded to single (2^-24 each, so 2^-23 per
// product), then six products and five sums within 2^-24 relative each, on top of the rounding of the
// coefficients: below 1e-6 * sum |q| in all; 4e-6 * sum |q| is added to the slack.
std::vector<ConeBandF32> cone_bands_f32(const std::vector<ConeBand>& bands) {
  std::vector<ConeBandF32> out(bands.size());
  for (size_t i = 0; i < bands.size(); i++) {
    double sum = 0;
    for (int c = 0; c < 6; c++) {
      out[i].q[c] = (float)bands[i].q[c];
      sum += fabs(bands[i].q[c]);
    }
    out[i].slack = std::nextafterf((float)(bands[i].slack + 4e-6 * sum), INFINITY);
    out[i].prim = bands[i].prim;
  }
  return out;
}

struct SceneBuild {
  const rt_scene_desc* d;
  BvhBuilder bvh;
  std::vector<std::vector<int>> gate_children, gate_leaves;
  std::vector<int> top_gates, top_leaves;
  std::vector<Gate> gates;
  std::vector<QuadPrim> quad_src, unbounded, band_prims;
  std::vector<ConeBand> bands;
  // CSG: one root per maximal operator tree
  std::vector<std::vector<int>> gate_csg;  // roots directly inside each group
  std::vector<int> top_csg;
  std::vector<CsgRoot> csg_roots;
  std::vector<CsgNode> csg_nodes;
  std::vector<CsgLeaf> csg_leaves;         // ref filled in once the BVH orders are final
  std::vector<QuadPrim> csg_quads;
  std::vector<TriIsect> csg_tris;
  std::vector<uint32_t> csg_unbounded;
  std::vector<int> csg_root_of;            // operator index -> root index (for roots), else -1
  int csg_hits = CSG_MIN_HITS;             // the longest hit list a root of this scene can produce
  std::vector<TriIsect> tri_src;
  std::vector<int> tri_xform;  // device xform index or -1 (identity)
  int stack_need = 1;

  bool xform_has_chain(int x) const { return d->xform_chain && d->xform_chain[x].count > 0; }
  void gate_chain(int gi, Gate& g) const {
    g.chain_first = g.chain_len = 0;
    if (d->gate_chain && d->gate_chain[gi].count > 0) {
      g.chain_first = (uint32_t)d->gate_chain[gi].first;
      g.chain_len = (uint32_t)d->gate_chain[gi].count;
    }
  }

  QuadPrim make_quad(int li) const {
    const rt_leaf& L = d->leaves[li];
    QuadPrim q{};
    copy3x4(q.m, d->xforms[L.xform].ray_inv);
    q.minimum = L.minimum;
    q.maximum = L.maximum;
    q.type = L.type | (xform_has_chain(L.xform) ? QUAD_CHAIN : 0);
    q.closed = L.closed;
    q.leaf = li;
    q.gate = L.gate;
    return q;
  }

  TriIsect make_tri(int li) const {
    const rt_leaf& L = d->leaves[li];
    const rt_triangle& T = d->triangles[L.tri];
    TriIsect ti{};
    for (int a = 0; a < 3; a++) {
      ti.p1[a] = T.p1[a];
      ti.e1[a] = T.e1[a];
      ti.e2[a] = T.e2[a];
    }
    ti.leaf = li;
    ti.xform = is_identity(d->xforms[L.xform].ray_inv) ? -1 : L.xform;
    if (xform_has_chain(L.xform)) ti.xform = L.xform | TRI_XFORM_CHAIN;
    return ti;
  }

  // Records of one CSG root (operator `top` of the description and everything below it) and its item
  // for the enclosing level; a root without a finite box goes on the list tested for every ray.
  void csg_items(const std::vector<int>& roots, std::vector<BuildItem>& items) {
    for (int top : roots) {
      const rt_csg& C = d->csg[top];
      CsgRoot R{};
      R.first_leaf = (uint32_t)csg_leaves.size();
      R.first_node = (uint32_t)csg_nodes.size();
      R.gate = C.gate;
      int last = top;  // operators below `top` follow it directly (parents first)
      while (last + 1 < d->n_csg) {
        int a = last + 1;
        while (a > top) a = d->csg[a].parent;
        if (a != top) break;
        last++;
      }
      for (int k = top; k <= last; k++) csg_nodes.push_back(CsgNode{d->csg[k].op, k == top ? -1 : d->csg[k].parent - top});
      R.n_nodes = (uint32_t)(last - top + 1);
      BuildItem it{};
      for (int a = 0; a < 3; a++) {
        it.lo[a] = INFINITY;
        it.hi[a] = -INFINITY;
      }
      bool bounded = true;
      int possible = 0;
      for (int li = C.first_leaf; li < C.first_leaf + C.n_leaves; li++) {
        const rt_leaf& L = d->leaves[li];
        if (L.type == RT_EMPTY) continue;
        bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
        double lo[3], hi[3];
        // a cone's parallel-ray root can lie anywhere (src/shape.rs:1097-1102): no box bounds the operator then
        if (L.type == RT_CONE || !leaf_box(L, d->xforms[L.xform], tri ? &d->triangles[L.tri] : nullptr, lo, hi)) bounded = false;
        else
          for (int a = 0; a < 3; a++) {
            it.lo[a] = fmin(it.lo[a], lo[a]);
            it.hi[a] = fmax(it.hi[a], hi[a]);
          }
        CsgLeaf cl{};
        if (tri) {
          cl.ref = make_run(KIND_TRI, (uint32_t)csg_tris.size(), 1);
          csg_tris.push_back(make_tri(li));
          possible += 1;
        } else {
          cl.ref = make_run(KIND_QUAD, (uint32_t)csg_quads.size(), 1);
          csg_quads.push_back(make_quad(li));
          possible += (L.type == RT_SPHERE || L.type == RT_CUBE) ? 2 : (L.type == RT_PLANE ? 1 : 4);
        }
        cl.node = L.csg - top;
        cl.left_mask = L.csg_left;
        cl.gate = L.gate;
        csg_leaves.push_back(cl);
      }
      R.n_leaves = (uint32_t)csg_leaves.size() - R.first_leaf;
      if (possible > CSG_SPILL_MAX_HITS)
        throw std::length_error("a CSG operator tree whose shapes can yield more than 4096 intersections per ray");
      csg_hits = std::max(csg_hits, possible);
      const uint32_t root_index = (uint32_t)csg_roots.size();
      csg_roots.push_back(R);
      if (R.n_leaves == 0) continue;
      if (!bounded) {
        csg_unbounded.push_back(root_index);
        continue;
      }
      Gate g{};  // stands for the root in the BVH
      g.root = REF_NONE;
      g.parent = C.gate;
      g.csg = root_index + 1;
      it.kind = KIND_GATE;
      it.index = (uint32_t)gates.size();
      gates.push_back(g);
      if (bvh.gate_alt.size() < gates.size()) bvh.gate_alt.resize(gates.size(), REF_NONE);
      bvh.gate_alt[it.index] = make_ref(KIND_GATE, it.index);
      if (bvh.gate_need.size() < gates.size()) bvh.gate_need.resize(gates.size(), 0);
      items.push_back(it);
    }
  }

  // Items of the leaves directly under one group (or the top level).
  void leaf_items(const std::vector<int>& leaves, std::vector<BuildItem>& items) {
    for (int li : leaves) {
      const rt_leaf& L = d->leaves[li];
      if (L.type == RT_EMPTY) continue;  // never intersects (src/shape.rs:789)
      const rt_xform& X = d->xforms[L.xform];
      BuildItem it{};
      bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
      if (!leaf_box(L, X, tri ? &d->triangles[L.tri] : nullptr, it.lo, it.hi)) {
        unbounded.push_back(make_quad(li));
        continue;
      }
      if (tri) {
        it.kind = KIND_TRI;
        it.index = (uint32_t)tri_src.size();
        tri_src.push_back(make_tri(li));
      } else {
        it.kind = KIND_QUAD;
        it.index = (uint32_t)quad_src.size();
        QuadPrim q = make_quad(li);
        if (L.type == RT_CONE) {
          ConeBand cb = make_cone_band(q, (uint32_t)band_prims.size());
          bands.push_back(cb);
          QuadPrim only = q;
          only.type |= QUAD_ONLY_PARALLEL;
          band_prims.push_back(only);
          q.type |= QUAD_SKIP_PARALLEL;
        }
        quad_src.push_back(q);
      }
      items.push_back(it);
    }
  }

  // Builds the subtree of gate gi and its item for the enclosing level.
  void build_gate(int gi, BuildItem* out_item, bool* has_item) {
    std::vector<BuildItem> items;
    for (int c : gate_children[gi]) {
      BuildItem it{};
      bool ok = false;
      build_gate(c, &it, &ok);
      if (ok) items.push_back(it);
    }
    leaf_items(gate_leaves[gi], items);
    csg_items(gate_csg[gi], items);
    const rt_gate& G = d->gates[gi];
    Gate& g = gates[gi];
    copy3x4(g.m, G.to_local);
    gate_chain(gi, g);
    for (int a = 0; a < 3; a++) {
      g.bmin[a] = G.bmin[a];
      g.bmax[a] = G.bmax[a];
    }
    g.parent = G.parent;
    g.csg = 0;
    double lo[3], hi[3];
    g.root = bvh.build(items, lo, hi);
    if ((size_t)gi >= bvh.gate_need.size()) bvh.gate_need.resize(gi + 1, 0);
    bvh.gate_need[gi] = bvh.stack_need + 1;
    // The exact test can be skipped for rays without a near-zero direction component when the
    // group space is world space and the reference box holds everything below (boxes are padded
    // by about 1e-7 relative, hence the tolerance).
    bool contains = g.root != REF_NONE;
    for (int a = 0; a < 3 && contains; a++) {
      double tol = 1e-6 * (fabs(lo[a]) + fabs(hi[a]) + (hi[a] - lo[a])) + 1e-8;
      if (!(G.bmin[a] <= lo[a] + tol) || !(G.bmax[a] >= hi[a] - tol)) contains = false;
    }
    g.skippable = (contains && is_identity(G.to_local)) ? 1u : 0u;
    if ((size_t)gi >= bvh.gate_alt.size()) bvh.gate_alt.resize(gi + 1, REF_NONE);
    bvh.gate_alt[gi] = g.skippable ? bvh.alt(g.root) : make_ref(KIND_GATE, (uint32_t)gi);
    *has_item = g.root != REF_NONE;
    if (*has_item) {
      out_item->kind = KIND_GATE;
      out_item->index = (uint32_t)gi;
      for (int a = 0; a < 3; a++) {
        out_item->lo[a] = lo[a];
        out_item->hi[a] = hi[a];
      }
    }
  }
};

}  // namespace

extern "C" void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  // frames may still run on any render stream and copies on the copy stream: drain them all
  // before any of the memory they use is released
  for (hipStream_t st : {s->stream, s->spare_lane[0].stream, s->spare_lane[1].stream, s->spare_lane[2].stream, s->frame_stream[0],
                         s->frame_stream[1], s->frame_stream[2], s->copy_stream})
    if (st) (void)hipStreamSynchronize(st);
  for (const rt_scene::Alloc& a : s->allocs) (void)hipFree(a.p);
  for (GenBuf& g : s->gens)
    if (g.slab) (void)hipFree(g.slab);
  if (s->aside_buf.slab) (void)hipFree(s->aside_buf.slab);
  if (s->csg_spill_buf) (void)hipFree(s->csg_spill_buf);
  if (s->d_counts) (void)hipFree(s->d_counts);
  if (s->d_stats) (void)hipFree(s->d_stats);
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  if (s->d_tiles) (void)hipFree(s->d_tiles);
  if (s->stage) (void)hipFree(s->stage);
  for (void* p : {(void*)s->host.d_tri_box, (void*)s->host.d_quad_box, (void*)s->host.d_parent, (void*)s->host.d_waits,
                  (void*)s->host.d_arrived})
    if (p) (void)hipFree(p);
  for (hipEvent_t e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->d_async_stats) (void)hipFree(s->d_async_stats);
  if (s->h_async_stats) (void)hipHostFree(s->h_async_stats);
  for (rt_scene::FrameSlot& f : s->slots)
    for (hipEvent_t e : {f.begin, f.end})
      if (e) (void)hipEventDestroy(e);
  for (rt_scene::Lane& o : s->spare_lane) {
    if (o.stream) (void)hipStreamSynchronize(o.stream);
    for (GenBuf& g : o.gens)
      if (g.slab) (void)hipFree(g.slab);
    if (o.aside_buf.slab) (void)hipFree(o.aside_buf.slab);
    if (o.d_counts) (void)hipFree(o.d_counts);
    if (o.d_stats) (void)hipFree(o.d_stats);
    if (o.stream) (void)hipStreamDestroy(o.stream);
  }
  for (rt_scene::SpecFrame& f : s->spec) {
    for (hipEvent_t e : {f.begin, f.end, f.done})
      if (e) (void)hipEventDestroy(e);
    if (f.h_counts) (void)hipHostFree(f.h_counts);
    if (f.h_stats) (void)hipHostFree(f.h_stats);
  }
  if (s->copy_stream) {
    (void)hipStreamSynchronize(s->copy_stream);
    (void)hipStreamDestroy(s->copy_stream);
  }
  if (s->copy_order) (void)hipEventDestroy(s->copy_order);
  for (hipEvent_t e : s->copy_order2)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->copy_fence)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : s->frame_stream)
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

static void box_to_f32(const double lo[3], const double hi[3], float* out6) {
  for (int a = 0; a < 3; a++) {
    float l = (float)lo[a], h = (float)hi[a];
    out6[a] = (double)l > lo[a] ? std::nextafterf(l, -INFINITY) : l;
    out6[3 + a] = (double)h < hi[a] ? std::nextafterf(h, INFINITY) : h;
  }
}

// `want` streams on hardware queues of their own, as far as the process has them.  The runtime spreads a process's
// streams over a few hardware queues (four by default) by a rule of its own, and two streams on one queue take turns:
// a copy stream that shares its queue with a render stream waits behind that stream's kernels and holds them up in
// turn (whole teapot frames at 1080p: 0.17-0.19 instead of 0.116 ms per frame, which of the two depending on what
// other streams the process had made before), and render streams that share one do not overlap their launches.  So
// the streams are chosen by measurement: out of a pool of eight, those on which two single-wave kernels that linger
// for 100 us run side by side rather than one after the other (about 5 ms per scene).  RT_AMD_NO_QUEUE_TEST=1: the
// first `want` streams of the pool, as made.  `avoid`: streams the new ones must not share a queue with either.
int rt::pick_streams(const hipStream_t* avoid, int n_avoid, int want, hipStream_t* out) {
  constexpr int POOL = 8;
  hipStream_t pool[POOL] = {};
  for (hipStream_t& st : pool) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  bool taken[POOL] = {};
  int n = 0;
  auto release_rest = [&]() {
    for (int i = 0; i < POOL; i++)
      if (!taken[i]) (void)hipStreamDestroy(pool[i]);
  };
  if (getenv("RT_AMD_NO_QUEUE_TEST")) {
    for (int i = 0; i < want; i++) { out[n++] = pool[i]; taken[i] = true; }
    release_rest();
    return RT_OK;
  }
  using clock = std::chrono::steady_clock;
  const unsigned long long ticks = 200000;  // about 100 us of the shader clock
  auto run = [&](hipStream_t a, hipStream_t b, double* us) -> int {  // both idle before and after
    const auto t0 = clock::now();
    launch_linger(a, ticks, nullptr);
    if (b) launch_linger(b, ticks, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(a));
    if (b) HIP_TRY(hipStreamSynchronize(b));
    *us = std::chrono::duration<double, std::micro>(clock::now() - t0).count();
    return RT_OK;
  };
  double alone = 0, t = 0;
  int rc;
  for (hipStream_t st : pool)  // every stream has made its first launch (queues are handed out then), code is loaded
    if ((rc = run(st, nullptr, &t))) { release_rest(); return rc; }
  alone = 1e30;
  for (int k = 0; k < 3; k++) {
    if ((rc = run(pool[0], nullptr, &t))) { release_rest(); return rc; }
    alone = std::min(alone, t);
  }
  auto side_by_side = [&](int i, int j, bool* yes) -> int {
    double best = 1e30;
    for (int k = 0; k < 2; k++) {
      double us = 0;
      int rc2 = run(pool[i], pool[j], &us);
      if (rc2) return rc2;
      best = std::min(best, us);
    }
    *yes = best < 1.6 * alone;
    return RT_OK;
  };
  for (int i = 0; i < POOL && n < want; i++) {
    bool ok = true;
    for (int j = 0; j < n_avoid && ok; j++) {
      double best = 1e30, us = 0;
      for (int k = 0; k < 2; k++) {
        if ((rc = run(pool[i], avoid[j], &us))) { release_rest(); return rc; }
        best = std::min(best, us);
      }
      ok = best < 1.6 * alone;
    }
    for (int j = 0; j < POOL && ok; j++)
      if (taken[j] && (rc = side_by_side(i, j, &ok))) { release_rest(); return rc; }
    if (ok) { taken[i] = true; n++; }
  }
  const int distinct = n;
  for (int i = 0; i < POOL && n < want; i++)  // fewer hardware queues than streams: the rest as they come
    if (!taken[i]) { taken[i] = true; n++; }
  int k = 0;
  for (int i = 0; i < POOL; i++)
    if (taken[i]) out[k++] = pool[i];
  release_rest();
  if (getenv("RT_AMD_DEBUG"))
    fprintf(stderr, "[rt_amd] streams: %d of %d on hardware queues of their own (a lingering kernel alone %.0f us)\n", distinct, want, alone);
  return RT_OK;
}

// Streams, events and the per-frame counters of a scene on `device` (which becomes current).
static int scene_runtime_init(rt_scene* s, int device) {
  HIP_TRY(hipSetDevice(device));
  s->device = device;
  if (const char* e = getenv("RT_AMD_FRAME_STREAMS")) s->n_frame_streams = std::min(std::max(atoi(e), 1), (int)rt_scene::FRAME_STREAMS);
  {  // the copy stream first: it is the one that must not share
    hipStream_t picked[rt_scene::FRAME_STREAMS + 1] = {};
    int rc = pick_streams(nullptr, 0, s->n_frame_streams + 1, picked);
    if (rc) return rc;
    s->copy_stream = picked[0];
    s->stream = picked[1];
    for (int k = 0; k + 1 < s->n_frame_streams; k++) s->frame_stream[k] = picked[2 + k];
  }
  s->last_stream = s->stream;
  HIP_TRY(hipEventCreateWithFlags(&s->copy_order, hipEventDisableTiming));
  for (hipEvent_t& e : s->copy_order2) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : s->copy_fence) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipMalloc((void**)&s->d_stats, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc((void**)&s->h_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  { int rc = ensure_depth(s, RECURSION_DEPTH); if (rc) return rc; }
  HIP_TRY(hipMalloc((void**)&s->d_async_stats, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(s->d_async_stats, 0, STAT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipStreamSynchronize(nullptr));  // the fill is asynchronous and the scene's streams do not wait for the null stream
  HIP_TRY(hipHostMalloc((void**)&s->h_async_stats, STAT_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  for (rt_scene::FrameSlot& f : s->slots) {
    HIP_TRY(hipEventCreate(&f.begin));
    HIP_TRY(hipEventCreate(&f.end));
  }
  return RT_OK;
}

// Hit lists longer than the LDS holds (SceneDev::csg_hits > CSG_MAX_HITS) live in device memory: one list per
// workgroup of a launch, as many as fit about a gigabyte (the scene's jobs are then launched in pieces of that
// many workgroups).  On the current device.
static int attach_csg_spill(rt_scene* s) {
  s->dev.csg_spill = nullptr;
  s->dev.csg_spill_slots = 0;
  s->dev.first_block = 0;
  if (s->dev.csg_hits <= CSG_MAX_HITS) return RT_OK;
  const size_t list_bytes = (size_t)s->dev.csg_hits * 3u * BLOCK * sizeof(uint32_t);
  size_t slots = ((size_t)1 << 30) / list_bytes;
  slots = std::min<size_t>(8192, std::max<size_t>(128, slots)) & ~(size_t)127;  // whole runs of the XCD-aware block order
  HIP_TRY(hipMalloc(&s->csg_spill_buf, slots * list_bytes));
  s->dev.csg_spill = (uint32_t*)s->csg_spill_buf;
  s->dev.csg_spill_slots = (uint32_t)slots;
  s->scene_bytes += slots * list_bytes;
  return RT_OK;
}

static int scene_create_impl(const rt_scene_desc* d, int device, uint32_t build_flags, rt_scene* s) {
  // ---- validate
  if (d->n_leaves < 0 || d->n_gates < 0 || d->n_xforms < 0 || d->n_triangles < 0 || d->n_materials < 0 ||
      d->n_patterns < 0)
    return fail(RT_ERR_INVALID, "negative count in scene description");
  if ((d->n_leaves > 0 && !d->leaves) || (d->n_gates > 0 && !d->gates) || (d->n_xforms > 0 && !d->xforms) ||
      (d->n_triangles > 0 && !d->triangles) || (d->n_materials > 0 && !d->materials) || (d->n_patterns > 0 && !d->patterns))
    return fail(RT_ERR_INVALID, "null array with a positive count in scene description");
  for (int i = 0; i < d->n_xforms; i++) {
    if (!affine(d->xforms[i].ray_inv) || !affine(d->xforms[i].full_inv) || !affine(d->xforms[i].fwd))
      return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) shape transform: last row must be 0 0 0 1");
  }
  for (int i = 0; i < d->n_gates; i++) {
    if (!affine(d->gates[i].to_local)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) group transform");
    if (d->gates[i].parent >= i || d->gates[i].parent < -1)
      return fail(RT_ERR_INVALID, "gate parent must precede its children");
  }
  if (d->n_csg < 0 || (d->n_csg > 0 && !d->csg)) return fail(RT_ERR_INVALID, "bad CSG operator list");
  if (d->n_chain_mats < 0 || (d->n_chain_mats > 0 && !d->chain_mats)) return fail(RT_ERR_INVALID, "bad transform chain list");
  for (int i = 0; i < d->n_chain_mats; i++)
    if (!affine(d->chain_mats + 16 * (size_t)i)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) matrix in a transform chain");
  bool any_chain = false;
  for (int pass = 0; pass < 2; pass++) {
    const rt_chain_ref* refs = pass ? d->gate_chain : d->xform_chain;
    const int n = pass ? d->n_gates : d->n_xforms;
    for (int i = 0; refs && i < n; i++) {
      if (refs[i].count < 0 || refs[i].first < 0 || (int64_t)refs[i].first + refs[i].count > d->n_chain_mats)
        return fail(RT_ERR_INVALID, "transform chain out of range");
      any_chain = any_chain || refs[i].count > 0;
    }
  }
  for (int i = 0; i < d->n_csg; i++) {
    const rt_csg& c = d->csg[i];
    if (c.op < RT_CSG_UNION || c.op > RT_CSG_DIFFERENCE) return fail(RT_ERR_INVALID, "unknown CSG operator");
    if (c.parent >= i || c.parent < -1) return fail(RT_ERR_INVALID, "CSG parent must precede its children");
    if (c.gate < -1 || c.gate >= d->n_gates) return fail(RT_ERR_INVALID, "CSG gate out of range");
    if (c.first_leaf < 0 || c.n_leaves < 0 || (int64_t)c.first_leaf + c.n_leaves > d->n_leaves)
      return fail(RT_ERR_INVALID, "CSG leaf range out of bounds");
    if (c.parent >= 0) {
      const rt_csg& p = d->csg[c.parent];
      if (c.first_leaf < p.first_leaf || c.first_leaf + c.n_leaves > p.first_leaf + p.n_leaves)
        return fail(RT_ERR_INVALID, "CSG leaf range not inside its parent's");
    }
  }
  for (int i = 0; i < d->n_gates; i++) {
    const rt_gate& G = d->gates[i];
    if (G.csg < -1 || G.csg >= d->n_csg) return fail(RT_ERR_INVALID, "gate CSG index out of range");
    // A group below an operator lies below the same operator as its parent group, or below one nested in it
    // (an operator between the two groups); a group outside every operator has no parent inside one.
    if (G.parent >= 0) {
      const int pc = d->gates[G.parent].csg;
      int a = G.csg;
      while (a >= 0 && a != pc) a = d->csg[a].parent;
      if (a != pc) return fail(RT_ERR_INVALID, "gate CSG operator does not lie below its parent gate's");
    }
  }
  for (int i = 0; i < d->n_patterns; i++) {
    const rt_pattern& p = d->patterns[i];
    if (!affine(p.inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) pattern transform");
    if (p.type < RT_PAT_NULL || p.type > RT_PAT_BLEND) return fail(RT_ERR_INVALID, "unknown pattern type");
    if (p.type == RT_PAT_BLEND && (p.first < 0 || p.first >= d->n_patterns || p.second < 0 || p.second >= d->n_patterns))
      return fail(RT_ERR_INVALID, "blend pattern operand out of range");
  }
  for (int i = 0; i < d->n_materials; i++)
    if (d->materials[i].pattern < -1 || d->materials[i].pattern >= d->n_patterns)
      return fail(RT_ERR_INVALID, "material pattern index out of range");
  for (int i = 0; i < d->n_leaves; i++) {
    const rt_leaf& L = d->leaves[i];
    if (L.type < RT_EMPTY || L.type > RT_SMOOTH_TRIANGLE) return fail(RT_ERR_INVALID, "unknown leaf type");
    if (L.material < 0 || L.material >= d->n_materials) return fail(RT_ERR_INVALID, "leaf material out of range");
    if (L.xform < 0 || L.xform >= d->n_xforms) return fail(RT_ERR_INVALID, "leaf xform out of range");
    if (L.gate < -1 || L.gate >= d->n_gates) return fail(RT_ERR_INVALID, "leaf gate out of range");
    if (L.csg < -1 || L.csg >= d->n_csg) return fail(RT_ERR_INVALID, "leaf CSG index out of range");
    if (L.csg >= 0 && (i < d->csg[L.csg].first_leaf || i >= d->csg[L.csg].first_leaf + d->csg[L.csg].n_leaves))
      return fail(RT_ERR_INVALID, "leaf outside the leaf range of its CSG operator");
    if (L.gate >= 0) {
      // the leaf's innermost operator is its group's, or one nested below the group's
      const int gc = d->gates[L.gate].csg;
      int a = L.csg;
      while (a >= 0 && a != gc) a = d->csg[a].parent;
      if (a != gc) return fail(RT_ERR_INVALID, "leaf CSG operator does not lie below its gate's");
    }
    bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
    if (tri && (L.tri < 0 || L.tri >= d->n_triangles)) return fail(RT_ERR_INVALID, "leaf triangle out of range");
  }
  if ((uint64_t)d->n_triangles > MAX_RUN_FIRST || (uint64_t)d->n_leaves > (1u << 24))
    return fail(RT_ERR_UNSUPPORTED, "too many primitives");  // node byte offsets are 32-bit

  int rci = scene_runtime_init(s, device);
  if (rci) return rci;
  s->build_flags = build_flags;

  // ---- acceleration structure
  auto t0 = std::chrono::steady_clock::now();
  SceneBuild b;
  b.d = d;
  {  // where the hierarchy is built: large subtrees on the GPU unless the caller or the environment says otherwise
    bool on_device = (build_flags & RT_BUILD_HOST) == 0;
    size_t min_items = (build_flags & RT_BUILD_DEVICE) ? 2048 : 32768;
    if (const char* e = getenv("RT_AMD_DEVICE_BVH")) {
      on_device = atoi(e) != 0;
      if (on_device) min_items = 2048;
    }
    if (const char* e = getenv("RT_AMD_DEVICE_BVH_MIN")) min_items = (size_t)std::max(2, atoi(e));
    b.bvh.on_device = on_device;
    b.bvh.device_min_items = min_items;
  }
  b.gate_children.resize(d->n_gates);
  b.gate_leaves.resize(d->n_gates);
  b.gates.reserve((size_t)d->n_gates + (size_t)d->n_csg);
  b.gates.resize(d->n_gates);
  b.gate_csg.resize(d->n_gates);
  for (int i = 0; i < d->n_gates; i++) {
    const rt_gate& G = d->gates[i];
    if (G.csg >= 0) {
      // a group below a CSG operator: only its gate record is needed (eval_csg tests it per leaf)
      Gate& g = b.gates[i];
      copy3x4(g.m, G.to_local);
      b.gate_chain(i, g);
      for (int a = 0; a < 3; a++) {
        g.bmin[a] = G.bmin[a];
        g.bmax[a] = G.bmax[a];
      }
      g.root = REF_NONE;
      g.parent = G.parent;
      continue;
    }
    if (G.parent < 0) b.top_gates.push_back(i);
    else b.gate_children[G.parent].push_back(i);
  }
  for (int i = 0; i < d->n_leaves; i++) {
    if (d->leaves[i].csg >= 0) continue;  // reached through its CSG root
    if (d->leaves[i].gate < 0) b.top_leaves.push_back(i);
    else b.gate_leaves[d->leaves[i].gate].push_back(i);
  }
  for (int i = 0; i < d->n_csg; i++) {
    if (d->csg[i].parent >= 0) continue;
    if (d->csg[i].gate < 0) b.top_csg.push_back(i);
    else b.gate_csg[d->csg[i].gate].push_back(i);
  }
  std::vector<BuildItem> items;
  for (int gi : b.top_gates) {
    BuildItem it{};
    bool ok = false;
    b.build_gate(gi, &it, &ok);
    if (ok) items.push_back(it);
  }
  b.leaf_items(b.top_leaves, items);
  b.csg_items(b.top_csg, items);
  double lo[3], hi[3];
  uint32_t root = b.bvh.build(items, lo, hi);
  s->stack_need = b.bvh.stack_need + 2 + 3;  // + root and gate hand-over, + 3 scratch slots of push_if
  if (s->stack_need > TRAVERSAL_STACK)
    return fail(RT_ERR_UNSUPPORTED, "scene hierarchy is deeper than the traversal stack");
  // what the kernels are launched with: the need, in steps of four, and room for the cone index's queue
  s->dev.stack_entries = std::min(TRAVERSAL_STACK, std::max(16, (s->stack_need + 3) & ~3));
  if (getenv("RT_AMD_DEBUG")) {
    fprintf(stderr, "[rt_amd] leaves=%d gates=%d nodes=%zu tris=%zu quads=%zu unbounded=%zu bands=%zu stack_need=%d\n", d->n_leaves,
            d->n_gates, b.bvh.nodes.size(), b.bvh.tri_order.size(), b.bvh.quad_order.size(), b.unbounded.size(), b.bands.size(),
            s->stack_need);
  }

  std::vector<TriIsect> tris(b.bvh.tri_order.size());
  for (size_t i = 0; i < tris.size(); i++) tris[i] = b.tri_src[b.bvh.tri_order[i]];
  std::vector<QuadPrim> quads(b.bvh.quad_order.size());
  for (size_t i = 0; i < quads.size(); i++) quads[i] = b.quad_src[b.bvh.quad_order[i]];
  // the shapes below CSG operators follow the BVH-ordered records
  for (CsgLeaf& cl : b.csg_leaves) cl.ref += (uint32_t)((cl.ref >> 30) == KIND_TRI ? tris.size() : quads.size());
  tris.insert(tris.end(), b.csg_tris.begin(), b.csg_tris.end());
  quads.insert(quads.end(), b.csg_quads.begin(), b.csg_quads.end());
  if (tris.size() > MAX_RUN_FIRST || quads.size() > MAX_RUN_FIRST) return fail(RT_ERR_UNSUPPORTED, "too many primitives");
  s->has_csg = !b.csg_roots.empty() || any_chain;  // either needs the general kernels
  s->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  s->device_build_seconds = b.bvh.device_seconds;
  s->device_builds = b.bvh.device_builds;

  // ---- shading tables
  ShadingTables sh;
  sh.build(d);
  s->max_children = sh.max_children;
  std::vector<LeafShade>& leaves = sh.leaves;
  std::vector<XformDev>& xforms = sh.xforms;
  std::vector<TriShade>& tri_shade = sh.tri_shade;
  std::vector<MaterialDev>& mats = sh.mats;
  std::vector<PatternDev>& pats = sh.pats;

  // ---- upload
  int rc;
  if ((rc = upload(s, b.bvh.nodes, &s->dev.nodes))) return rc;
  if ((rc = upload(s, tris, &s->dev.tris))) return rc;
  if ((rc = upload(s, quads, &s->dev.quads))) return rc;
  if ((rc = upload(s, b.gates, &s->dev.gates))) return rc;
  // top-level planes without a transform chain go into the kernel arguments (the order among the
  // shapes tested for every ray does not matter: equal distances resolve by leaf index)
  s->dev.n_inline_planes = 0;
  for (size_t i = 0; i < b.unbounded.size();) {
    const QuadPrim& q = b.unbounded[i];
    if (q.type == RT_PLANE && q.gate < 0 && s->dev.n_inline_planes < MAX_INLINE_PLANES) {
      InlinePlane& P = s->dev.inline_planes[s->dev.n_inline_planes++];
      for (int k = 0; k < 4; k++) P.row[k] = q.m[4 + k];
      P.leaf = q.leaf;
      P.pad = 0;
      b.unbounded.erase(b.unbounded.begin() + (long)i);
    } else {
      i++;
    }
  }
  if ((rc = upload(s, b.unbounded, &s->dev.unbounded))) return rc;
  if ((rc = upload(s, b.csg_roots, &s->dev.csg_roots))) return rc;
  if ((rc = upload(s, b.csg_nodes, &s->dev.csg_nodes))) return rc;
  if ((rc = upload(s, b.csg_leaves, &s->dev.csg_leaves))) return rc;
  if ((rc = upload(s, b.csg_unbounded, &s->dev.csg_unbounded))) return rc;
  s->dev.n_csg_unbounded = (int32_t)b.csg_unbounded.size();
  s->dev.csg_hits = b.csg_hits;
  if ((rc = attach_csg_spill(s))) return rc;
  if ((rc = upload(s, b.bands, &s->dev.bands))) return rc;
  {
    std::vector<ConeBandF32> bands32 = cone_bands_f32(b.bands);
    if ((rc = upload(s, bands32, &s->dev.bands32))) return rc;
  }
  if ((rc = upload(s, b.band_prims, &s->dev.band_prims))) return rc;
  s->dev.n_bands = (int32_t)b.bands.size();
#ifdef RT_DIAG_NO_BANDS  // diagnostic builds only (timing of the band list): drops the cones' parallel-ray hits
  if (getenv("RT_AMD_TIMING_NO_BANDS")) s->dev.n_bands = 0;
  s->dev.band_diag_passes = getenv("RT_AMD_TIMING_BAND_PASSES") ? atoi(getenv("RT_AMD_TIMING_BAND_PASSES")) : 3;
#endif
  if (getenv("RT_AMD_BAND_GRID_HOST")) {  // the host builder (what the device builder is checked against)
    BandGrid grid = build_band_grid(b.bands, (size_t)1 << 29);  // at most 2 GiB of entries
    s->dev.band_grid_n = grid.n;
    if ((rc = upload(s, grid.start, &s->dev.band_start))) return rc;
    if ((rc = upload(s, grid.items, &s->dev.band_items))) return rc;
    grid.strip.resize(grid.strip.size() + 4, 0u);  // a ray reads four words at a time
    if ((rc = upload(s, grid.strip, &s->dev.band_strip))) return rc;
    if (getenv("RT_AMD_DEBUG"))
      fprintf(stderr, "[rt_amd] band grid n=%d cells=%zu entries=%zu\n", grid.n, grid.start.size(), grid.items.size());
  } else {
    // built on this GPU from the bands just uploaded: milliseconds where the host builder takes seconds
    if ((rc = attach_band_grid(s, b.bands.size()))) return rc;
  }
  if ((rc = upload(s, leaves, &s->dev.leaves))) return rc;
  if ((rc = upload(s, xforms, &s->dev.xforms))) return rc;
  if (any_chain) {
    std::vector<double> chain_mats((size_t)d->n_chain_mats * 12);
    for (int i = 0; i < d->n_chain_mats; i++) copy3x4(&chain_mats[(size_t)i * 12], d->chain_mats + 16 * (size_t)i);
    std::vector<uint32_t> xform_chain((size_t)d->n_xforms * 2, 0u);
    for (int i = 0; d->xform_chain && i < d->n_xforms; i++) {
      xform_chain[2 * (size_t)i] = (uint32_t)d->xform_chain[i].first;
      xform_chain[2 * (size_t)i + 1] = (uint32_t)d->xform_chain[i].count;
    }
    if ((rc = upload(s, chain_mats, &s->dev.chain_mats))) return rc;
    if ((rc = upload(s, xform_chain, &s->dev.xform_chain))) return rc;
  }
  if ((rc = upload(s, tri_shade, &s->dev.tri_shade))) return rc;
  if ((rc = upload(s, mats, &s->dev.materials))) return rc;
  if ((rc = upload(s, pats, &s->dev.patterns))) return rc;
  s->dev.root = root;
  s->dev.root_alt = b.bvh.alt(root);
  if (root != REF_NONE) box_to_f32(lo, hi, s->dev.scene_box);  // the node planes are these boxes' subsets, rounded the same way
  {  // the host copy rt_scene_update works from
    rt_scene::Host& H = s->host;
    H.tris = tris;
    H.quads = quads;
    H.unbounded = b.unbounded;
    H.band_prims = b.band_prims;
    H.gates = b.gates;
    H.leaves.assign(d->leaves, d->leaves + d->n_leaves);
    H.gate_parent.resize(d->n_gates);
    for (int i = 0; i < d->n_gates; i++) H.gate_parent[i] = d->gates[i].parent;
    if (d->xform_chain) H.xform_chain.assign(d->xform_chain, d->xform_chain + d->n_xforms);
    if (d->gate_chain) H.gate_chain.assign(d->gate_chain, d->gate_chain + d->n_gates);
    H.boxed.assign((size_t)d->n_leaves, 0);
    for (size_t i = 0; i < b.bvh.tri_order.size(); i++) H.boxed[(size_t)tris[i].leaf] = 1;
    for (size_t i = 0; i < b.bvh.quad_order.size(); i++) H.boxed[(size_t)quads[i].leaf] = 1;
    for (int i = 0; i < s->dev.n_inline_planes; i++) H.inline_plane_leaf.push_back(s->dev.inline_planes[i].leaf);
    H.n_xforms = d->n_xforms; H.n_triangles = d->n_triangles; H.n_materials = d->n_materials;
    H.n_patterns = d->n_patterns; H.n_chain_mats = d->n_chain_mats; H.n_csg = d->n_csg;
    // who delivers the box of every node: its parent slot, through a gate when the node is a group's root
    const std::vector<BvhNode>& nodes = b.bvh.nodes;
    H.n_nodes = (uint32_t)nodes.size();
    H.parent.assign(nodes.size(), REF_NONE);
    H.waits.assign(nodes.size(), 0u);
    for (size_t n = 0; n < nodes.size(); n++)
      for (int c = 0; c < 4; c++) {
        uint32_t ref = nodes[n].child[c];
        if (ref != REF_NONE && (ref >> 30) == KIND_GATE) {
          const Gate& g = b.gates[ref & 0x3FFFFFFFu];
          ref = g.csg ? REF_NONE : g.root;
        }
        if (ref != REF_NONE && (ref >> 30) == KIND_INNER) {
          H.parent[ref & 0x3FFFFFFFu] = (uint32_t)(n << 2) | (uint32_t)c;
          H.waits[n]++;
        }
      }
  }
  // Measured: the scalar path for wave-uniform nodes pays off for hierarchies far beyond the 32 KB L1
  // of a CU (1M-triangle mesh: -6 % frame time) and costs a few percent on small ones (12k triangles:
  // +5 %), whose upper levels stay in L1 anyway.
  s->uniform_nodes = b.bvh.nodes.size() >= 8192;
  if (const char* e = getenv("RT_AMD_UNIFORM_NODES")) s->uniform_nodes = atoi(e) != 0;
  // Measured: in a run of triangles, fetching the next record while one is tested takes a tenth off the
  // frame of a 1M-triangle mesh (2.72 -> 2.44 ms at 4K: ten triangle tests per ray, records far beyond a
  // CU's L1) and costs the 12k-triangle teapot 6 % (the twenty registers held across a test are spilled,
  // and its one or two tests per ray find their records in L1).  Scenes that spawn rays keep the plain
  // loop (their kernels spill more than they gain).
  s->tri_prefetch = s->uniform_nodes && b.tri_src.size() * sizeof(TriIsect) >= (4u << 20);
  if (const char* e = getenv("RT_AMD_TRI_PREFETCH")) s->tri_prefetch = atoi(e) != 0;
  // Scenes with bounded cones: the shadow queries of hits on a cone's extension are walked by a kernel of
  // their own (k_shadow) instead of holding up the wave that found them.
  s->set_aside = !b.bands.empty();
  if (const char* e = getenv("RT_AMD_SET_ASIDE")) s->set_aside = atoi(e) != 0 && !b.bands.empty();
  s->dev.n_unbounded = (int32_t)b.unbounded.size();
  s->dev.n_leaves = d->n_leaves;
  for (int a = 0; a < 3; a++) {
    s->dev.light_intensity[a] = d->light.intensity[a];
    s->dev.light_position[a] = d->light.position[a];
  }
  round_shape_noise(d, s->dev.box_noise);
  // Not in scenes with bounded cones: their parallel-ray branch (the band list) puts hits up to 1e8 units
  // out on the cones' extensions, and a ray that starts there would, widened, meet every box of the scene
  // one after the other in a single lane (measured: 12 -> 200 ms per frame with 25 000 cones, for rays
  // whose reference results are rounding noise).  Far cameras and lights are the case this is for.
  if (s->dev.n_bands > 0) s->dev.box_noise[0] = s->dev.box_noise[1] = 0.0;

  return RT_OK;
}

extern "C" int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
  return rt_scene_create_with(desc, device, 0u, out);
}

extern "C" int rt_scene_create_with(const rt_scene_desc* desc, int device, uint32_t build_flags, rt_scene** out) {
  if (!desc || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  rt_scene* s = new (std::nothrow) rt_scene();
  if (!s) return fail(RT_ERR_NOMEM, "out of host memory");
  int rc;
  try {
    rc = scene_create_impl(desc, device, build_flags, s);
  } catch (const std::bad_alloc&) {
    rc = fail(RT_ERR_NOMEM, "out of host memory while building the scene");
  } catch (const std::length_error& e) {
    rc = fail(RT_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) {
    rc = fail(RT_ERR_INVALID, e.what());
  }
  if (rc != RT_OK) {
    std::string keep = last_error_text();
    rt_scene_destroy(s);
    last_error_text() = keep;
    return rc;
  }
  *out = s;
  return RT_OK;
}

// Builds the cones' direction index on the current device from the bands already uploaded and hangs it
// on the scene (16-bit entries when the band numbers fit).
int rt::attach_band_grid(rt_scene* s, size_t n_bands) {
  s->dev.band_grid_n = 0;
  s->dev.band_start = s->dev.band_items = s->dev.band_strip = nullptr;
  s->dev.band_items16 = 0;
  int n = 0;
  uint32_t *d_start = nullptr, *d_items = nullptr;
  size_t n_items = 0;
  int rc = build_band_grid_device(s->dev.bands, n_bands, (size_t)1 << 30, &n, &d_start, &d_items, &n_items);
  if (rc) return rc;
  size_t item_bytes = (n_items + 1) * 4;
  uint32_t* d_strip = nullptr;
  if (d_items && (rc = build_band_strips_device(s->dev.bands, n, d_start, d_items, n_items, &d_strip))) {
    (void)hipFree(d_items);
    if (d_start) (void)hipFree(d_start);
    return rc;
  }
  if (d_items && n_bands < 65536) {
    uint16_t* small = nullptr;
    if ((rc = compress_band_items_u16(d_items, n_items, &small))) {
      (void)hipFree(d_items);
      if (d_start) (void)hipFree(d_start);
      if (d_strip) (void)hipFree(d_strip);
      return rc;
    }
    d_items = reinterpret_cast<uint32_t*>(small);
    item_bytes = (n_items + 2) * 2;
    s->dev.band_items16 = 1;
  }
  s->dev.band_grid_n = n;
  s->dev.band_start = d_start;
  s->dev.band_items = d_items;
  if (d_start) {
    s->allocs.push_back(rt_scene::Alloc{d_start, ((size_t)3 * n * n + 1) * 4, offsetof(SceneDev, band_start)});
    s->scene_bytes += ((size_t)3 * n * n + 1) * 4;
  }
  if (d_items) {
    s->allocs.push_back(rt_scene::Alloc{d_items, item_bytes, offsetof(SceneDev, band_items)});
    s->scene_bytes += item_bytes;
  }
  s->dev.band_strip = d_strip;
  if (d_strip) {
    s->allocs.push_back(rt_scene::Alloc{d_strip, (n_items + 4) * 4, offsetof(SceneDev, band_strip)});
    s->scene_bytes += (n_items + 4) * 4;
  }
  if (getenv("RT_AMD_DEBUG"))
    fprintf(stderr, "[rt_amd] band grid n=%d entries=%zu (%d-bit) built on the device\n", n, n_items, s->dev.band_items16 ? 16 : 32);
  return RT_OK;
}

// Finds the device array a pointer of `dev` owns.
static const rt_scene::Alloc* alloc_of(const rt_scene* s, size_t field) {
  for (const rt_scene::Alloc& a : s->allocs)
    if (a.field == field) return &a;
  return nullptr;
}
template <class T>
static int reupload(rt_scene* s, size_t field, const std::vector<T>& v) {
  if (v.empty()) return RT_OK;
  const rt_scene::Alloc* a = alloc_of(s, field);
  if (!a || a->bytes != v.size() * sizeof(T)) return fail(RT_ERR_INVALID, "rt_scene_update: a table changed size");
  HIP_TRY(hipMemcpy(a->p, v.data(), a->bytes, hipMemcpyHostToDevice));
  return RT_OK;
}
#define RT_FIELD(f) offsetof(SceneDev, f)

// *rebuild: the description is the same shape tree but the hierarchy cannot keep its shape (nothing was changed).
// *touched: device tables have been written (a failure from here on leaves the scene half updated).
static int scene_update_impl(rt_scene* s, const rt_scene_desc* d, bool* rebuild, bool* touched) {
  rt_scene::Host& H = s->host;
  // ---- the shape tree must be the one the scene was built from
  if (d->n_leaves != (int32_t)H.leaves.size() || d->n_gates != (int32_t)H.gate_parent.size() || d->n_xforms != H.n_xforms ||
      d->n_triangles != H.n_triangles || d->n_materials != H.n_materials || d->n_patterns != H.n_patterns || d->n_csg != H.n_csg)
    return fail(RT_ERR_INVALID, "rt_scene_update: the description has other counts than the scene was built from");
  if (d->n_chain_mats < 0 || (d->n_chain_mats > 0 && !d->chain_mats)) return fail(RT_ERR_INVALID, "bad transform chain list");
  if ((d->n_leaves > 0 && !d->leaves) || (d->n_gates > 0 && !d->gates) || (d->n_xforms > 0 && !d->xforms) ||
      (d->n_triangles > 0 && !d->triangles) || (d->n_materials > 0 && !d->materials) || (d->n_patterns > 0 && !d->patterns) ||
      (d->n_chain_mats > 0 && !d->chain_mats))
    return fail(RT_ERR_INVALID, "null array with a positive count in scene description");
  for (int i = 0; i < d->n_leaves; i++) {
    const rt_leaf &A = d->leaves[i], &B = H.leaves[(size_t)i];
    if (A.type != B.type || A.gate != B.gate || A.csg != B.csg || A.xform != B.xform || A.tri != B.tri || A.closed != B.closed)
      return fail(RT_ERR_INVALID, "rt_scene_update: the shape tree differs from the one the scene was built from");
    if (A.material < 0 || A.material >= d->n_materials) return fail(RT_ERR_INVALID, "leaf material out of range");
  }
  for (int i = 0; i < d->n_gates; i++) {
    if (d->gates[i].parent != H.gate_parent[(size_t)i])
      return fail(RT_ERR_INVALID, "rt_scene_update: the group tree differs from the one the scene was built from");
    if (!affine(d->gates[i].to_local)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) group transform");
  }
  for (int i = 0; i < d->n_xforms; i++)
    if (!affine(d->xforms[i].ray_inv) || !affine(d->xforms[i].full_inv) || !affine(d->xforms[i].fwd))
      return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) shape transform: last row must be 0 0 0 1");
  for (int i = 0; i < d->n_chain_mats; i++)
    if (!affine(d->chain_mats + 16 * (size_t)i)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) matrix in a transform chain");
  if (d->n_chain_mats != H.n_chain_mats) {  // some chain changed its length
    *rebuild = true;
    return RT_OK;
  }
  for (int pass = 0; pass < 2; pass++) {  // the chains keep their places (their matrices may change)
    const rt_chain_ref* now = pass ? d->gate_chain : d->xform_chain;
    const std::vector<rt_chain_ref>& then = pass ? H.gate_chain : H.xform_chain;
    const int n = pass ? d->n_gates : d->n_xforms;
    for (int i = 0; i < n; i++) {
      const rt_chain_ref a = now ? now[i] : rt_chain_ref{0, 0}, b = then.empty() ? rt_chain_ref{0, 0} : then[(size_t)i];
      if (a.count != b.count || (a.count > 0 && a.first != b.first)) {
        *rebuild = true;  // an identity level became a transform or the reverse: the chains are laid out anew
        return RT_OK;
      }
    }
  }
  for (int i = 0; i < d->n_patterns; i++) {
    const rt_pattern& P = d->patterns[i];
    if (!affine(P.inv)) return fail(RT_ERR_UNSUPPORTED, "projective (non-affine) pattern transform");
    if (P.type < RT_PAT_NULL || P.type > RT_PAT_BLEND) return fail(RT_ERR_INVALID, "unknown pattern type");
    if (P.type == RT_PAT_BLEND && (P.first < 0 || P.first >= d->n_patterns || P.second < 0 || P.second >= d->n_patterns))
      return fail(RT_ERR_INVALID, "blend pattern operand out of range");
  }
  for (int i = 0; i < d->n_materials; i++)
    if (d->materials[i].pattern < -1 || d->materials[i].pattern >= d->n_patterns)
      return fail(RT_ERR_INVALID, "material pattern index out of range");

  if (H.n_csg > 0) {  // the operators' records and boxes are laid out by the build
    *rebuild = true;
    return RT_OK;
  }
  SceneBuild b;
  b.d = d;
  // ---- every shape stays where it is: in the hierarchy, or among the shapes tested for every ray
  std::vector<float> tri_box(H.tris.size() * 6, 0.0f), quad_box(H.quads.size() * 6, 0.0f);
  std::vector<TriIsect> tris(H.tris.size());
  std::vector<QuadPrim> quads(H.quads.size()), unbounded(H.unbounded.size()), band_prims(H.band_prims.size());
  std::vector<ConeBand> bands(H.band_prims.size());
  auto box_of_leaf = [&](int li, double lo[3], double hi[3]) {
    const rt_leaf& L = d->leaves[li];
    const bool tri = L.type == RT_TRIANGLE || L.type == RT_SMOOTH_TRIANGLE;
    return leaf_box(L, d->xforms[L.xform], tri ? &d->triangles[L.tri] : nullptr, lo, hi);
  };
  for (int li = 0; li < d->n_leaves; li++) {
    if (d->leaves[li].type == RT_EMPTY) continue;
    double lo[3], hi[3];
    if (d->leaves[li].csg < 0 && box_of_leaf(li, lo, hi) != (H.boxed[(size_t)li] != 0)) {
      *rebuild = true;  // a shape crossed between the hierarchy and the shapes tested for every ray
      return RT_OK;
    }
  }
  for (size_t i = 0; i < tris.size(); i++) {
    const int li = H.tris[i].leaf;
    tris[i] = b.make_tri(li);
    double lo[3], hi[3];
    box_of_leaf(li, lo, hi);
    box_to_f32(lo, hi, &tri_box[i * 6]);
  }
  for (size_t i = 0; i < quads.size(); i++) {
    const int li = H.quads[i].leaf;
    quads[i] = b.make_quad(li);
    quads[i].type |= H.quads[i].type & QUAD_SKIP_PARALLEL;
    double lo[3], hi[3];
    box_of_leaf(li, lo, hi);
    box_to_f32(lo, hi, &quad_box[i * 6]);
  }
  for (size_t i = 0; i < unbounded.size(); i++) unbounded[i] = b.make_quad(H.unbounded[i].leaf);
  for (size_t i = 0; i < band_prims.size(); i++) {
    band_prims[i] = b.make_quad(H.band_prims[i].leaf);
    bands[i] = make_cone_band(band_prims[i], (uint32_t)i);
    band_prims[i].type |= QUAD_ONLY_PARALLEL;
  }
  std::vector<Gate> gates = H.gates;
  for (int gi = 0; gi < d->n_gates; gi++) {
    Gate& g = gates[(size_t)gi];
    const rt_gate& G = d->gates[gi];
    copy3x4(g.m, G.to_local);
    b.gate_chain(gi, g);
    for (int a = 0; a < 3; a++) {
      g.bmin[a] = G.bmin[a];
      g.bmax[a] = G.bmax[a];
    }
    // Whether the exact test of a group may be skipped depends on its box containing everything below,
    // which only the refit knows: from the first update on every gate is tested.
    g.skippable = 0;
  }
  ShadingTables sh;
  sh.build(d);

  // ---- nothing may be in flight while the tables change
  int rc = collect_all(s);
  if (rc) return rc;
  for (hipStream_t st : {s->stream, s->frame_stream[0], s->frame_stream[1], s->frame_stream[2], s->copy_stream}) HIP_TRY(hipStreamSynchronize(st));
  *touched = true;
  if ((rc = reupload(s, RT_FIELD(tris), tris))) return rc;
  if ((rc = reupload(s, RT_FIELD(quads), quads))) return rc;
  if ((rc = reupload(s, RT_FIELD(gates), gates))) return rc;
  if ((rc = reupload(s, RT_FIELD(unbounded), unbounded))) return rc;
  if ((rc = reupload(s, RT_FIELD(band_prims), band_prims))) return rc;
  if ((rc = reupload(s, RT_FIELD(bands), bands))) return rc;
  if ((rc = reupload(s, RT_FIELD(bands32), cone_bands_f32(bands)))) return rc;
  if ((rc = reupload(s, RT_FIELD(leaves), sh.leaves))) return rc;
  if ((rc = reupload(s, RT_FIELD(xforms), sh.xforms))) return rc;
  if ((rc = reupload(s, RT_FIELD(tri_shade), sh.tri_shade))) return rc;
  if ((rc = reupload(s, RT_FIELD(materials), sh.mats))) return rc;
  if ((rc = reupload(s, RT_FIELD(patterns), sh.pats))) return rc;
  if (s->dev.chain_mats) {
    std::vector<double> chain_mats((size_t)d->n_chain_mats * 12);
    for (int i = 0; i < d->n_chain_mats; i++) copy3x4(&chain_mats[(size_t)i * 12], d->chain_mats + 16 * (size_t)i);
    if ((rc = reupload(s, RT_FIELD(chain_mats), chain_mats))) return rc;
  }
  for (int i = 0; i < s->dev.n_inline_planes; i++) {
    const QuadPrim q = b.make_quad(H.inline_plane_leaf[(size_t)i]);
    for (int k = 0; k < 4; k++) s->dev.inline_planes[i].row[k] = q.m[4 + k];
  }
  for (int a = 0; a < 3; a++) {
    s->dev.light_intensity[a] = d->light.intensity[a];
    s->dev.light_position[a] = d->light.position[a];
  }
  round_shape_noise(d, s->dev.box_noise);
  // Not in scenes with bounded cones: their parallel-ray branch (the band list) puts hits up to 1e8 units
  // out on the cones' extensions, and a ray that starts there would, widened, meet every box of the scene
  // one after the other in a single lane (measured: 12 -> 200 ms per frame with 25 000 cones, for rays
  // whose reference results are rounding noise).  Far cameras and lights are the case this is for.
  if (s->dev.n_bands > 0) s->dev.box_noise[0] = s->dev.box_noise[1] = 0.0;
  s->max_children = sh.max_children;
  s->hint_valid = false;
  // ---- the cones' direction index follows their new matrices
  if (s->dev.band_grid_n > 0) {
    for (size_t field : {RT_FIELD(band_start), RT_FIELD(band_items), RT_FIELD(band_strip)})
      for (size_t k = 0; k < s->allocs.size(); k++)
        if (s->allocs[k].field == field) {
          s->scene_bytes -= s->allocs[k].bytes;
          (void)hipFree(s->allocs[k].p);
          s->allocs.erase(s->allocs.begin() + (long)k);
          break;
        }
    if ((rc = attach_band_grid(s, bands.size()))) return rc;
  }
  // ---- the hierarchy keeps its shape and gets new boxes
  if (H.n_nodes > 0) {
    if (!H.d_parent) {
      HIP_TRY(hipMalloc((void**)&H.d_parent, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_waits, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_arrived, H.n_nodes * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void**)&H.d_tri_box, std::max<size_t>(1, tri_box.size()) * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&H.d_quad_box, std::max<size_t>(1, quad_box.size()) * sizeof(float)));
      HIP_TRY(hipMemcpy(H.d_parent, H.parent.data(), H.n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(H.d_waits, H.waits.data(), H.n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!tri_box.empty()) HIP_TRY(hipMemcpy(H.d_tri_box, tri_box.data(), tri_box.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!quad_box.empty()) HIP_TRY(hipMemcpy(H.d_quad_box, quad_box.data(), quad_box.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(H.d_arrived, 0, H.n_nodes * sizeof(uint32_t), main_stream(s)));
    RefitView v{const_cast<BvhNode*>(s->dev.nodes), H.d_tri_box, H.d_quad_box, s->dev.gates, H.d_parent, H.d_waits, H.d_arrived, H.n_nodes};
    launch_refit(main_stream(s), v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(main_stream(s)));
  }
  s->dev.root_alt = s->dev.root;  // with every gate tested, both kinds of ray walk the same references
  // the box of the whole hierarchy: from the root node's new planes, or none (everything passes)
  for (int a = 0; a < 3; a++) {
    s->dev.scene_box[a] = -INFINITY;
    s->dev.scene_box[3 + a] = INFINITY;
  }
  if (s->dev.root != REF_NONE && (s->dev.root >> 30) == KIND_INNER) {
    BvhNode top;
    HIP_TRY(hipMemcpy(&top, s->dev.nodes + s->dev.root, sizeof top, hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; a++) {
      float l = INFINITY, h = -INFINITY;
      for (int c = 0; c < 4; c++)
        if (top.child[c] != REF_NONE) {
          l = fminf(l, top.plane[a][0][c]);
          h = fmaxf(h, top.plane[a][1][c]);
        }
      s->dev.scene_box[a] = l;
      s->dev.scene_box[3 + a] = h;
    }
  }
  H.tris.swap(tris);
  H.quads.swap(quads);
  H.unbounded.swap(unbounded);
  H.band_prims.swap(band_prims);
  H.gates.swap(gates);
  H.updated = true;
  return RT_OK;
}
#undef RT_FIELD

// The acceleration structure built again from desc for a scene that keeps its handle, streams and frame buffers:
// a new scene is created beside the old one and the two trade their scene data, so a failure leaves the old
// scene as it was.
static int scene_rebuild(rt_scene* s, const rt_scene_desc* d) {
  rt_scene* fresh = nullptr;
  int rc = rt_scene_create_with(d, s->device, s->build_flags, &fresh);
  if (rc != RT_OK) return rc;
  rc = collect_all(s);
  if (rc == RT_OK)
    for (hipStream_t st : {s->stream, s->spare_lane[0].stream, s->spare_lane[1].stream, s->spare_lane[2].stream, s->frame_stream[0],
                           s->frame_stream[1], s->frame_stream[2], s->copy_stream})
      if (st && hipStreamSynchronize(st) != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_scene_update: a stream of the scene failed");
  if (rc != RT_OK) {
    std::string keep = last_error_text();
    rt_scene_destroy(fresh);
    last_error_text() = keep;
    return rc;
  }
  std::swap(s->dev, fresh->dev);
  std::swap(s->allocs, fresh->allocs);
  std::swap(s->scene_bytes, fresh->scene_bytes);
  std::swap(s->stack_need, fresh->stack_need);
  std::swap(s->has_csg, fresh->has_csg);
  std::swap(s->uniform_nodes, fresh->uniform_nodes);
  std::swap(s->tri_prefetch, fresh->tri_prefetch);
  std::swap(s->max_children, fresh->max_children);
  std::swap(s->set_aside, fresh->set_aside);
  std::swap(s->host, fresh->host);  // the refit's device arrays travel with their owner
  std::swap(s->csg_spill_buf, fresh->csg_spill_buf);
  s->build_seconds += fresh->build_seconds;
  s->device_build_seconds += fresh->device_build_seconds;
  s->device_builds += fresh->device_builds;
  s->hint_valid = false;
  s->rebuilds++;
  rt_scene_destroy(fresh);  // the old tables
  return RT_OK;
}

extern "C" int rt_scene_update(rt_scene* s, const rt_scene_desc* desc) {
  if (!s || !desc) return fail(RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  bool rebuild = false, touched = false;
  int rc;
  try {
    rc = scene_update_impl(s, desc, &rebuild, &touched);
    // Shape::set_transform works for any shape (src/shape.rs:592-656).  Where the hierarchy cannot keep its shape --
    // a shape crossed between bounded and unbounded, a transform chain changed its length, the scene has CSG
    // operators -- it is built again.
    if (rc == RT_OK && rebuild) rc = scene_rebuild(s, desc);
  } catch (const std::bad_alloc&) {
    rc = fail(RT_ERR_NOMEM, "out of host memory while updating the scene");
  } catch (const std::length_error& e) {
    rc = fail(RT_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) {
    rc = fail(RT_ERR_INVALID, e.what());
  }
  if (rc == RT_OK) s->broken = false;
  else if (touched) s->broken = true;  // some tables are new, some old
  return rc;
}

extern "C" int rt_scene_rebuilds(const rt_scene* s, uint64_t* rebuilds) {
  if (!s || !rebuilds) return fail(RT_ERR_INVALID, "null argument");
  *rebuilds = s->rebuilds;
  return RT_OK;
}

// A second copy of a scene on another GPU (or the same one): the acceleration structure is built once
// and its arrays are copied device to device.
extern "C" int rt_scene_clone(const rt_scene* src, int device, rt_scene** out) {
  if (!src || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  RT_REQUIRE_USABLE(src);
  rt_scene* s = new (std::nothrow) rt_scene();
  if (!s) return fail(RT_ERR_NOMEM, "out of host memory");
  auto work = [&]() -> int {
    int rc = scene_runtime_init(s, device);
    if (rc) return rc;
    s->dev = src->dev;
    for (const rt_scene::Alloc& a : src->allocs) *(const void**)((char*)&s->dev + a.field) = nullptr;
    for (const rt_scene::Alloc& a : src->allocs) {
      void* p = nullptr;
      HIP_TRY(hipMalloc(&p, a.bytes));
      s->allocs.push_back(rt_scene::Alloc{p, a.bytes, a.field});
      HIP_TRY(hipMemcpyPeer(p, device, a.p, src->device, a.bytes));
      *(const void**)((char*)&s->dev + a.field) = p;
    }
    HIP_TRY(hipDeviceSynchronize());
    s->scene_bytes = src->scene_bytes;
    if (src->dev.csg_spill) {
      s->scene_bytes -= (size_t)src->dev.csg_spill_slots * src->dev.csg_hits * 3u * BLOCK * sizeof(uint32_t);
      int rcs = attach_csg_spill(s);
      if (rcs) return rcs;
    }
    s->build_seconds = 0;
    s->stack_need = src->stack_need;
    s->has_csg = src->has_csg;
    s->uniform_nodes = src->uniform_nodes;
    s->tri_prefetch = src->tri_prefetch;
    s->set_aside = src->set_aside;
    s->build_flags = src->build_flags;
    s->max_children = src->max_children;
    s->host = src->host;  // vectors copied; the refit's device arrays are made on the first update
    s->host.d_tri_box = s->host.d_quad_box = nullptr;
    s->host.d_parent = s->host.d_waits = s->host.d_arrived = nullptr;
    return RT_OK;
  };
  int rc = work();
  if (rc != RT_OK) {
    std::string keep = last_error_text();
    rt_scene_destroy(s);
    last_error_text() = keep;
    return rc;
  }
  *out = s;
  return RT_OK;
}
