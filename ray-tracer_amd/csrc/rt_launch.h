// Host-side entry points of the kernels that live in translation units of their own.
#pragma once
#include "rt_records.h"

namespace rt {

// One generation of rays (rt_generation.h).  STACK: entries of the per-lane traversal stack; COUNT:
// traversal counters; CSG: the general kernels (CSG operators, transform chains); UNIFORM: scalar path
// for nodes a whole wave visits; TRIPF: a triangle run's next record fetched ahead (mode 0 only).
// extra_lds: bytes of dynamic LDS (the CSG hit list).
template <int STACK, bool COUNT, bool CSG, bool UNIFORM, bool TRIPF>
void launch_generation_variant(const SceneDev& dev, unsigned extra_lds, int mode, const GenLaunch& a);

// 40 entries x 64 lanes x 4 B = 10 KB per wave: sixteen waves still fit the 160 KB of LDS of a CU,
// so the stack does not limit occupancy; deeper scenes take the 64-entry kernels (10 waves per CU).
constexpr int STACK_SMALL = 40;
inline int pick_stack(int need) { return need <= STACK_SMALL ? STACK_SMALL : 64; }

// What k_refit works on (rt_scene_update): new boxes for a hierarchy whose primitives moved.
struct RefitView {
  BvhNode* nodes;
  const float* tri_box;    // 6 floats per TriIsect slot: lo xyz, hi xyz (rounded outward)
  const float* quad_box;   // the same per QuadPrim slot
  const Gate* gates;
  const uint32_t* parent;  // per node: (parent node << 2 | slot), REF_NONE for a node without one
  const uint32_t* waits;   // per node: children whose box another node delivers
  uint32_t* arrived;       // per node, zeroed before the launch
  uint32_t n_nodes;
};

// rt_kernels.hip.  `general`: the scene has CSG operators or transform chains; stack: pick_stack().
void launch_load_rays(hipStream_t stream, GenView g, uint32_t n, const double* rays6);
void launch_resolve(hipStream_t stream, GenView g, uint32_t n, GenView next, const uint32_t* queue_counts, uint32_t segcap);
void launch_resolve_pixels(hipStream_t stream, GenView g, uint32_t n, GenView next, const CameraDev& cam, const uint32_t* tiles,
                           uint32_t tiles_x, const ImageOut& img);
void launch_output_rays(hipStream_t stream, GenView g, uint32_t n, double* rgb, int32_t* leaf, double* t);
void launch_intersect(hipStream_t stream, const SceneDev& dev, int stack, bool general, unsigned extra_lds, GenView g, uint32_t n);
void launch_points_shadowed(hipStream_t stream, const SceneDev& dev, int stack, bool general, unsigned extra_lds, uint32_t n,
                            const double* pts, uint8_t* out);
void launch_pixel_rays(hipStream_t stream, const CameraDev& cam, uint32_t n, const uint32_t* xy, double* rays6);
void launch_refit(hipStream_t stream, const RefitView& v);

}  // namespace rt
