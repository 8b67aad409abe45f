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
  std::vector<uint32_t> strip;  // per entry: where in the cell the band lies (band_strip_word, band_grid_test.h)
};

// Builds the grid for bands (empty grid when there are few of them).  max_items bounds the memory:
// while the grid holds more entries the resolution is halved.
BandGrid build_band_grid(const std::vector<ConeBand>& bands, size_t max_items);

// The resolution both builders try first for n_bands cones.
int band_grid_first_n(size_t n_bands);

// The same grid built on the GPU from bands already in device memory (the current device's): same
// cells, same entries in the same order as build_band_grid.  On success *d_start (3 n n + 1 words) and
// *d_items (*n_items words) are hipMalloc'ed buffers the caller owns; *n_out = 0 and null buffers when
// there are too few bands for a grid.  Returns 0 or a negative rt_status (rt_last_error() says why).
int build_band_grid_device(const ConeBand* d_bands, size_t n_bands, size_t max_items, int* n_out, uint32_t** d_start,
                           uint32_t** d_items, size_t* n_items);

// The strip words of a grid on the device (one per entry, in the entries' order): *out is a new
// hipMalloc'ed array of n_items words.  Returns 0 or a negative rt_status.
int build_band_strips_device(const ConeBand* d_bands, int n, const uint32_t* d_start, const uint32_t* d_items, size_t n_items,
                             uint32_t** out);

// The entries of a grid on the device as 16-bit numbers (for fewer than 65536 bands): *out is a new
// hipMalloc'ed array of n_items entries, d_items is freed.  Returns 0 or a negative rt_status.
int compress_band_items_u16(uint32_t* d_items, size_t n_items, uint16_t** out);

}  // namespace rt
