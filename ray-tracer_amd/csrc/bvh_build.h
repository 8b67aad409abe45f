// Host-side builder of the world-space BVH the traversal kernels walk.
#pragma once
#include <vector>

#include "rt_types.h"

namespace rt {

struct BuildItem {
  double lo[3], hi[3];
  uint32_t kind;   // KIND_TRI / KIND_QUAD: index into the source array; KIND_GATE: gate index
  uint32_t index;
};

struct BinNode {  // binary SAH node, collapsed into 4-wide BvhNodes afterwards
  double lo[2][3], hi[2][3];
  uint32_t c[2];
};

// What the builder on the GPU (bvh_device.hip) returns: nodes and primitive orders numbered from zero.
struct DeviceBvh {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> tri_order, quad_order;
  uint32_t root = REF_NONE;
  int stack_need = 0;
  int levels = 0;
  double seconds = 0;  // upload, kernels, download
};
// gate_alt / gate_need: BvhBuilder's tables (indexed by gate).  Runs on the current device.  RT_OK, or a
// negative rt_status with the message set (RT_ERR_UNSUPPORTED: let the host builder do it).
int build_bvh_device(const BuildItem* items, size_t n_items, const std::vector<uint32_t>& gate_alt, const std::vector<int>& gate_need,
                     DeviceBvh* out);

struct BvhBuilder {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> tri_order;   // source triangle index of every emitted TriIsect slot
  std::vector<uint32_t> quad_order;  // same for QuadPrim slots
  int threads = 0;                   // build threads; 0: RT_AMD_BUILD_THREADS or the host's cores, at most 16
  bool on_device = false;            // builds of at least device_min_items items run on the current GPU (bvh_device.hip)
  size_t device_min_items = 2048;
  double device_seconds = 0;         // time spent in device builds
  size_t device_builds = 0;
  int stack_need = 0;                // traversal stack entries the last build()'s subtree can need
  std::vector<int> gate_need;        // the same for the subtree behind each gate (indexed by gate)
  // What a ray that may skip the exact gate test walks instead of gate g: the subtree behind it (with
  // skippable gates at its root resolved the same way), or the gate itself when it cannot be skipped.
  std::vector<uint32_t> gate_alt;
  uint32_t alt(uint32_t ref) const {
    if (ref != REF_NONE && (ref >> 30) == KIND_GATE && (ref & 0x3FFFFFFFu) < gate_alt.size())
      return gate_alt[ref & 0x3FFFFFFFu];
    return ref;
  }

  // Builds a subtree over items (reordered in place).  Returns its traversal
  // reference and the union box; REF_NONE when items is empty.
  uint32_t build(std::vector<BuildItem>& items, double lo[3], double hi[3]);

 private:
  std::vector<BinNode> bin_;
  // threads: how many threads this subtree may use; large ranges hand their two halves to builders of
  // their own and splice the results in the order a single thread would have produced them
  uint32_t build_range(BuildItem* items, size_t n, int depth, double lo[3], double hi[3], int threads);
  uint32_t splice(const BvhBuilder& sub, uint32_t ref);
  uint32_t make_leaf(BuildItem* items, size_t n);
  uint32_t collapse(uint32_t bin_ref, int* need);
};

}  // namespace rt
