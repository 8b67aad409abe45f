// Host-side entry points of the kernels that live in translation units of their own.
#pragma once
#include "rt_records.h"

namespace rt {

// One generation of rays (rt_generation.h).  COUNT: traversal counters; CSG: the general kernels (CSG operators,
// transform chains); UNIFORM: scalar path for nodes a whole wave visits; TRIPF: a triangle run's next record fetched
// ahead (mode 0 only).  lds_bytes: the dynamic LDS of a workgroup (traversal stack, then the CSG hit list).
template <bool COUNT, bool CSG, bool UNIFORM, bool TRIPF>
void launch_generation_variant(const SceneDev& dev, unsigned lds_bytes, int mode, const GenLaunch& a);

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

// rt_kernels.hip.  `general`: the scene has CSG operators or transform chains; lds_bytes as above.
void launch_load_rays(hipStream_t stream, GenView g, uint32_t n, const double* rays6);
void launch_resolve(hipStream_t stream, GenView g, uint32_t n, GenView next, const uint32_t* queue_counts, uint32_t segcap);
void launch_resolve_pixels(hipStream_t stream, GenView g, uint32_t n, GenView next, const CameraDev& cam, const uint32_t* tiles,
                           uint32_t tiles_x, const ImageOut& img);
void launch_output_rays(hipStream_t stream, GenView g, uint32_t n, double* rgb, int32_t* leaf, double* t);
void launch_intersect(hipStream_t stream, const SceneDev& dev, bool general, unsigned lds_bytes, GenView g, uint32_t n);
void launch_points_shadowed(hipStream_t stream, const SceneDev& dev, bool general, unsigned lds_bytes, uint32_t n,
                            const double* pts, uint8_t* out);
void launch_pixel_rays(hipStream_t stream, const CameraDev& cam, uint32_t n, const uint32_t* xy, double* rays6);
void launch_refit(hipStream_t stream, const RefitView& v);
void launch_linger(hipStream_t stream, unsigned long long ticks, unsigned long long* sink);

}  // namespace rt
