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

struct BvhBuilder {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> tri_order;   // source triangle index of every emitted TriIsect slot
  std::vector<uint32_t> quad_order;  // same for QuadPrim slots
  int max_depth = 0;                 // deepest chain of inner nodes produced by one build() call

  // Builds a subtree over items (reordered in place).  Returns its traversal
  // reference and the union box; REF_NONE when items is empty.
  uint32_t build(std::vector<BuildItem>& items, double lo[3], double hi[3]);

 private:
  uint32_t build_range(BuildItem* items, size_t n, int depth, double lo[3], double hi[3]);
  uint32_t make_leaf(BuildItem* items, size_t n);
};

}  // namespace rt
