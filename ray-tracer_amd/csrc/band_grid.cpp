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

// Cells per side of a face: a power of two in [32, 8192].  The ceiling is what the device builder's
// walk() can hold: its stack entries keep du, dv in 14 bits each (8192 / 1 cells below a start cell at
// most) and the level in 4 bits, its private stack of 40 entries covers 13 levels (3 per level + 1), and
// the scans index 3 n^2 + 1 cells with an int.
constexpr int BAND_GRID_MIN = 32, BAND_GRID_CEILING = 8192;
static_assert(3ll * BAND_GRID_CEILING * BAND_GRID_CEILING + 1 < 0x7FFFFFFFll, "cell count must fit an int");
static_assert(BAND_GRID_CEILING <= (1 << 14) / 2, "walk() packs cell offsets into 14 bits");

int band_grid_first_n(size_t n_bands) {
  static const int n_max = [] {
    int v = 1024;
    if (const char* e = getenv("RT_AMD_BAND_GRID_MAX")) v = atoi(e);
    v = std::max(BAND_GRID_MIN, std::min(BAND_GRID_CEILING, v));
    int p = BAND_GRID_MIN;
    while (p * 2 <= v) p *= 2;  // rounded down to a power of two
    return p;
  }();
  int n = BAND_GRID_MIN;
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
    grid.strip.resize(total);
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
          for (size_t c = lo; c < std::min(cells, lo + 4096); c++) {
            std::sort(grid.items.begin() + grid.start[c], grid.items.begin() + grid.start[c + 1]);
            const int face = (int)(c / ((size_t)n * n)), iv = (int)((c / (size_t)n) % (size_t)n), iu = (int)(c % (size_t)n);
            for (uint32_t k = grid.start[c]; k < grid.start[c + 1]; k++) {
              const ConeBand& b = bands[grid.items[k]];
              grid.strip[k] = band_strip_word(face_poly(b, face), band_threshold(b), n, iu, iv);
            }
          }
        }
      });
    for (auto& t : pool) t.join();
    return grid;
  }
}

}  // namespace rt
