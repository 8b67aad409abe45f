// Direction index over the cone bands.  A bounded cone reports an unbounded hit for the thin set of
// ray directions d with |a(d)| < FEQ_EPSILON (ConeBand in rt_types.h); with many cones, scanning all
// of them per ray costs more than the traversal, so the directions are bucketed: a cube map folded
// by the sign of the major axis (a is even in d), N x N cells per axis, each cell listing the cones
// whose band can reach it.
#pragma once
#include <cstdint>
#include <vector>

#include "rt_types.h"

namespace rt {

struct BandGrid {
  int n = 0;                    // cells per side of one face; 0 = no grid (scan the list)
  std::vector<uint32_t> start;  // 3 n n + 1 offsets into items, cell = (axis n + iv) n + iu
  std::vector<uint32_t> items;  // band indices
};

// Builds the grid for bands (empty grid when there are few of them).  max_items bounds the memory.
BandGrid build_band_grid(const std::vector<ConeBand>& bands, size_t max_items);

}  // namespace rt
