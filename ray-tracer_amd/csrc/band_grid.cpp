#include "band_grid.h"

#include "band_grid_test.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <thread>

namespace rt {
namespace {

// Visits every finest-level cell of one face the band can reach (band_cell_passes at every level).
template <class Emit>
void raster(const FacePoly& p, double thr, int n, int level_n, int iu, int iv, Emit&& emit) {
  if (!band_cell_passes(p, thr, level_n, iu, iv)) return;
  if (level_n == n) {
    emit(iv * n + iu);
    return;
  }
  for (int k = 0; k < 4; k++) raster(p, thr, n, level_n * 2, iu * 2 + (k & 1), iv * 2 + (k >> 1), emit);
}

template <class Emit>
void raster_band(const ConeBand& b, int n, Emit&& emit) {
  const double thr = band_threshold(b);
  for (int m = 0; m < 3; m++) {
    FacePoly p = face_poly(b, m);
    raster(p, thr, n, 1, 0, 0, [&](int cell) { emit(m * n * n + cell); });
  }
}

}  // namespace

int band_grid_first_n(size_t n_bands) {
  int n = 32;
  static const int n_max = getenv("RT_AMD_BAND_GRID_MAX") ? atoi(getenv("RT_AMD_BAND_GRID_MAX")) : 4096;  // cells per side of a face, at most
  while (n < n_max && n < 1.7 * (double)n_bands) n *= 2;
  return n;
}

BandGrid build_band_grid(const std::vector<ConeBand>& bands, size_t max_items) {
  BandGrid grid;
  const size_t nb = bands.size();
  if (nb <= 32) return grid;  // a short list is cheaper to scan than to index
  int n = band_grid_first_n(nb);
  const unsigned workers = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));

  for (;; n /= 2) {
    const size_t cells = (size_t)3 * n * n;
    std::vector<std::atomic<uint32_t>> count(cells + 1);
    for (auto& c : count) c.store(0, std::memory_order_relaxed);
    auto each_band = [&](auto&& fn) {
      std::vector<std::thread> pool;
      std::atomic<size_t> next{0};
      for (unsigned w = 0; w < workers; w++)
        pool.emplace_back([&] {
          for (;;) {
            size_t lo = next.fetch_add(64);
            if (lo >= nb) return;
            for (size_t i = lo; i < std::min(nb, lo + 64); i++) fn(i);
          }
        });
      for (auto& t : pool) t.join();
    };
    each_band([&](size_t i) {
      raster_band(bands[i], n, [&](int cell) { count[cell].fetch_add(1, std::memory_order_relaxed); });
    });
    size_t total = 0;
    for (size_t c = 0; c < cells; c++) total += count[c].load(std::memory_order_relaxed);
    if (total > max_items && n > 32) continue;  // coarser cells: fewer entries per band
    grid.n = n;
    grid.start.resize(cells + 1);
    uint32_t run = 0;
    for (size_t c = 0; c < cells; c++) {
      grid.start[c] = run;
      run += count[c].load(std::memory_order_relaxed);
      count[c].store(grid.start[c], std::memory_order_relaxed);  // becomes the fill cursor
    }
    grid.start[cells] = run;
    grid.items.resize(total);
    each_band([&](size_t i) {
      raster_band(bands[i], n, [&](int cell) {
        grid.items[count[cell].fetch_add(1, std::memory_order_relaxed)] = (uint32_t)i;
      });
    });
    // a fixed order inside every cell, whatever the thread schedule was
    std::vector<std::thread> pool;
    std::atomic<size_t> next{0};
    for (unsigned w = 0; w < workers; w++)
      pool.emplace_back([&] {
        for (;;) {
          size_t lo = next.fetch_add(4096);
          if (lo >= cells) return;
          for (size_t c = lo; c < std::min(cells, lo + 4096); c++)
            std::sort(grid.items.begin() + grid.start[c], grid.items.begin() + grid.start[c + 1]);
        }
      });
    for (auto& t : pool) t.join();
    return grid;
  }
}

}  // namespace rt
