// Device-side scene layout of the MI355X render path (shared by the host
// builder and the HIP kernels).  Everything a ray touches while traversing is
// packed into 128-byte records (one MI355X cache line): a BVH node with both
// child boxes, or one transformed primitive.
#pragma once
#include <stdint.h>

namespace rt {

constexpr double FEQ_EPSILON = 0.0001;  // reference src/consts.rs:7
constexpr int RECURSION_DEPTH = 5;      // reference src/consts.rs:10 (the depth parallel.rs:47 uses)
constexpr int MAX_GENERATIONS = RECURSION_DEPTH + 1;

// 32-bit traversal reference: kind in the top two bits.
//   inner node : index in the low 30 bits
//   tri / quad : run of (count) records, count-1 in bits 27..29, first index in the low 27 bits
//   gate       : index in the low 30 bits
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
constexpr uint32_t KIND_INNER = 0, KIND_TRI = 1, KIND_QUAD = 2, KIND_GATE = 3;
constexpr int MAX_RUN = 8;
constexpr uint32_t MAX_RUN_FIRST = (1u << 27) - 1;
constexpr int TRAVERSAL_STACK = 64;  // entries per ray; the builder keeps tree depth below this

inline uint32_t make_ref(uint32_t kind, uint32_t index) { return (kind << 30) | index; }
inline uint32_t make_run(uint32_t kind, uint32_t first, uint32_t count) {
  return (kind << 30) | ((count - 1) << 27) | first;
}

// 4-wide BVH node: the world-space boxes of up to four children as f32, rounded outward from the
// f64 boxes (a box test only has to be conservative; the primitives behind it are tested in f64).
// plane[a][k][c] is the lower (k = 0) or upper (k = 1) plane of axis a of child c: a ray picks the
// near plane of an axis by byte offset (32 a + 0 or 16), the far plane is that offset ^ 16, and one
// 16-byte load brings the plane of all four children.  Empty slots hold an inverted box
// (+inf, -inf) and REF_NONE.  alt[] repeats child[] with every skippable gate replaced by the
// subtree behind it: a ray that may skip those gates (Gate::skippable) reads alt[] and never leaves
// the node loop for them.
struct alignas(128) BvhNode {
  float plane[3][2][4];
  uint32_t child[4];
  uint32_t alt[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be one cache line");

// Triangle as the intersection routine needs it (reference src/shape.rs:1225-1260).
struct TriIsect {
  double p1[3], e1[3], e2[3];
  int32_t leaf;   // depth-first leaf index (tie-break order and shading lookup)
  int32_t xform;  // index into xforms, or -1 when the ray needs no transform (identity)
};
static_assert(sizeof(TriIsect) == 80, "TriIsect layout");

// Sphere / plane / cube / cylinder / cone with its world->object matrix.
struct alignas(128) QuadPrim {
  double m[12];  // rows 0..2 of ray_inv
  double minimum, maximum;
  int32_t type, closed, leaf, gate;  // gate: innermost group (used for unbounded primitives only)
};
static_assert(sizeof(QuadPrim) == 128, "QuadPrim must be one cache line");

// A cone treats a ray whose object-space direction satisfies |dx^2 - dy^2 + dz^2| < FEQ_EPSILON as
// parallel to one of its halves and then reports the single root -c / 2b without the y-range check
// (reference src/shape.rs:1089-1103): that hit can lie anywhere on the infinite cone, outside any
// box.  A bounded cone is therefore stored twice: in the BVH with QUAD_SKIP_PARALLEL (every other
// ray), and in the band list with QUAD_ONLY_PARALLEL, which every ray scans by direction only.
constexpr int32_t QUAD_TYPE_MASK = 0xFF, QUAD_SKIP_PARALLEL = 0x100, QUAD_ONLY_PARALLEL = 0x200;

// The reference moves a ray into object space level by level: every enclosing group or CSG operator
// applies its own transform_inverse to the ray its parent handed down (src/shape.rs:1467-1470,
// 1213-1218).  One matrix product applied once gives the same ray only up to rounding as soon as two
// of those matrices differ from the identity, and a last-bit difference in t can turn a grazing
// reflection or shadow ray.  Such a shape or group carries a chain: its non-identity local inverses,
// outermost first, applied one after the other (SceneDev::chain_mats).  QUAD_CHAIN marks a QuadPrim
// whose chain is found through its leaf's transform record, TRI_XFORM_CHAIN the same in TriIsect::xform.
constexpr int32_t QUAD_CHAIN = 0x400;
constexpr int32_t TRI_XFORM_CHAIN = 0x40000000;

// Direction pre-filter of one band-list cone: a(d) = q . (dx^2, dy^2, dz^2, dx dy, dx dz, dy dz) is the
// cone's quadratic coefficient for a world-space direction d, up to rounding covered by slack |d|^2.
struct alignas(64) ConeBand {
  double q[6];
  double slack;
  uint32_t prim;  // index into band_prims
  uint32_t pad;
};
static_assert(sizeof(ConeBand) == 64, "ConeBand layout");

// The same quadratic form in single precision with the rounding of that evaluation added to the
// slack: a first, cheaper filter in front of ConeBand (never rejects what ConeBand accepts).
struct alignas(32) ConeBandF32 {
  float q[6];
  float slack;
  uint32_t prim;  // index into band_prims
};
static_assert(sizeof(ConeBandF32) == 32, "ConeBandF32 layout");

// One reference Group: exact bounding-box gate in group space, then its subtree.
struct alignas(32) Gate {
  uint32_t root;  // traversal reference of the subtree (REF_NONE when empty)
  int32_t parent;
  // 1 when the exact test may be skipped for a ray with no direction component below FEQ_EPSILON:
  // the group's ray transform is the identity and its reference box contains everything under it,
  // so for such a ray the reference's gate is an ordinary slab test that the BVH already performs.
  uint32_t skippable;
  uint32_t csg;  // 0: a group; k + 1: stands for CSG root k, nothing else of the record is used
  double m[12];  // rows 0..2 of to_local
  double bmin[3], bmax[3];
  uint32_t chain_first, chain_len;  // chain_len > 0: the matrices to apply instead of m, see QUAD_CHAIN
  uint32_t pad[6];
};
static_assert(sizeof(Gate) == 192, "Gate layout");

// CSG (reference src/shape.rs:1339-1354, src/intersect.rs:278-328).  A root is a maximal operator
// tree; a ray that reaches it collects the intersections of all leaves below (at most
// SceneDev::csg_hits, the most any root of the scene can yield: 2 per sphere or cube, 4 per cylinder or
// cone, 1 per plane or triangle), sorts them and runs every operator's filter, innermost first.  The list
// lives in LDS, 768 bytes per entry and wave, behind the traversal stack: CSG_MAX_HITS entries fit the 64 KB a
// workgroup may use.  A scene with a root that can yield more keeps its lists in device memory instead
// (SceneDev::csg_spill: one list per workgroup of a launch, so such scenes are launched in pieces of
// csg_spill_slots workgroups), up to CSG_SPILL_MAX_HITS entries.
constexpr int CSG_MIN_HITS = 16, CSG_MAX_HITS = 60, CSG_SPILL_MAX_HITS = 4096;
struct CsgRoot {
  uint32_t first_leaf, n_leaves;  // into csg_leaves
  uint32_t first_node, n_nodes;   // into csg_nodes, parents before children
  int32_t gate;                   // innermost group around the root (its gates are the traversal's business)
  int32_t pad[3];
};
struct CsgNode {
  int32_t op;      // rt_csg_op
  int32_t parent;  // index into csg_nodes relative to the root's first_node, or -1
};
struct CsgLeaf {
  uint32_t ref;        // KIND_TRI / KIND_QUAD reference into tris / quads (run of one)
  int32_t node;        // innermost operator, relative to the root's first_node
  uint32_t left_mask;  // bit k: counts as a left-operand intersection for the k-th operator above
  int32_t gate;        // innermost group around the leaf
};

// Per-leaf shading lookup, indexed by depth-first leaf index.
struct LeafShade {
  int32_t type, closed, material, xform;
  int32_t tri;  // triangle index or -1
  int32_t fixed_normal;  // 1: plane_normal is the world-space unit normal of the whole shape (planes, flat triangles)
  double minimum, maximum;
  double plane_normal[3];  // a plane's object-space normal, or the normal fixed_normal speaks of
};
static_assert(sizeof(LeafShade) == 64, "LeafShade layout");

struct XformDev {
  double ray_inv[12];
  double full_inv[12];
};

struct TriShade {
  double normal[3];
  double n1[3], n2[3], n3[3];
};

struct MaterialDev {
  double color[3];
  double ambient, diffuse, specular, shininess;
  double reflective, refractive_index, transparency;
  int32_t pattern;
  int32_t flags;  // bit0: !feq(reflective,0); bit1: !feq(transparency,0); bit2: schlick mix
};

struct PatternDev {
  int32_t type, first, second, pad;
  double inv[12];
  double a[3], b[3];
};

// A top-level plane kept in the kernel arguments (the floor of nearly every scene): its test needs only
// the y row of the world -> object matrix (reference src/shape.rs:891-908), so a ray meets it without a
// memory access.  Planes inside groups, with a transform chain or beyond MAX_INLINE_PLANES stay in
// `unbounded`.
constexpr int MAX_INLINE_PLANES = 2;
struct InlinePlane {
  double row[4];  // m[4..7]: y of the transformed origin = row . (o, 1), of the direction = row . (d, 0)
  int32_t leaf;
  int32_t pad;
};

struct SceneDev {
  const BvhNode* nodes;
  const TriIsect* tris;
  const QuadPrim* quads;
  const Gate* gates;
  const QuadPrim* unbounded;  // tested for every ray (planes, infinite cylinders, very large shapes)
  const ConeBand* bands;      // bounded cones: direction pre-filter of their parallel-ray branch
  const ConeBandF32* bands32;  // first pass of that filter
  const QuadPrim* band_prims;
  const uint32_t* band_start;  // direction grid over bands (band_grid.h), null when band_grid_n == 0
  const uint32_t* band_items;  // the cells' entries: 32-bit band indices, or 16-bit ones when band_items16 is set
  const uint32_t* band_strip;  // per entry: the strip of the cell the band lies in (band_grid_test.h), padded by four words
  const CsgRoot* csg_roots;
  const CsgNode* csg_nodes;
  const CsgLeaf* csg_leaves;
  const uint32_t* csg_unbounded;  // roots without a finite box: evaluated for every ray
  uint32_t* csg_spill;            // hit lists in device memory (csg_hits > CSG_MAX_HITS), else null: they live in LDS
  const LeafShade* leaves;
  const XformDev* xforms;
  const TriShade* tri_shade;
  const MaterialDev* materials;
  const PatternDev* patterns;
  const double* chain_mats;     // 12 doubles (rows 0..2) per matrix
  const uint32_t* xform_chain;  // per transform record: first matrix, count (0: the product is exact)
  uint32_t root;      // for rays that take every gate test
  uint32_t root_alt;  // for rays that may skip skippable gates
  int32_t n_unbounded;
  int32_t n_leaves;
  int32_t n_bands;
  int32_t band_grid_n;
  int32_t n_csg_unbounded;
  int32_t n_inline_planes;
  int32_t band_items16;  // 1: band_items holds uint16_t entries (scenes of fewer than 65536 bounded cones)
  int32_t csg_hits;      // entries of the per-ray CSG hit list (at least CSG_MIN_HITS)
  int32_t stack_entries; // entries of the per-lane traversal stack the kernels are launched with (<= TRAVERSAL_STACK)
  uint32_t csg_spill_slots;  // workgroups one launch of such a scene may have (each owns one list of csg_spill)
  // first workgroup of this launch among those of the whole job (launches in pieces: 0 otherwise); set per launch
  uint32_t first_block;
  int32_t stack_pad;
#ifdef RT_DIAG_NO_BANDS
  int32_t band_diag_passes;  // timing builds only: how many passes of the cone direction index run (0..3)
  int32_t band_diag_pad;
#endif
  // How far outside their exact shape the reference can report hits of round shapes for a ray from a
  // far origin (make_raybox): [0] sphere / cylinder term, [1] cone term, [2] largest |centre|^2 of a round shape.
  double box_noise[3];
  // Box of everything in the hierarchy (lo xyz, hi xyz, rounded outward): a ray that misses it fetches no node.
  float scene_box[6];
  InlinePlane inline_planes[MAX_INLINE_PLANES];
  double light_intensity[3];
  double light_position[3];
};

}  // namespace rt
