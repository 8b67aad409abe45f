#include "band_grid.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <thread>

namespace rt {
namespace {

// On the face of major axis m a direction is (1 at m, u at axis a, v at axis b), u, v in [-1, 1], and
// a(d) / d_m^2 = g(u, v) = c0 + cu u + cv v + cuu u^2 + cuv u v + cvv v^2.
struct FacePoly {
  double c0, cu, cv, cuu, cuv, cvv;
};

FacePoly face_poly(const ConeBand& b, int m) {
  // b.q = coefficients of (xx, yy, zz, xy, xz, yz)
  const double sq[3] = {b.q[0], b.q[1], b.q[2]};
  auto cross = [&](int i, int j) {  // coefficient of d_i d_j, i != j
    int lo = std::min(i, j), hi = std::max(i, j);
    return lo == 0 ? (hi == 1 ? b.q[3] : b.q[4]) : b.q[5];
  };
  const int a = (m + 1) % 3, c = (m + 2) % 3;
  return FacePoly{sq[m], cross(m, a), cross(m, c), sq[a], cross(a, c), sq[c]};
}

// Visits every finest-level cell of one face the band can reach.  A cell passes when |g| can drop
// below the band threshold somewhere inside it: centred form of the quadratic, which is an exact
// enclosure of the linear part and bounds the second-order part by its coefficients.
template <class Emit>
void raster(const FacePoly& p, double thr, int n, int level_n, int iu, int iv, Emit&& emit) {
  const double size = 2.0 / level_n;
  const double h = 0.5 * size * (1.0 + 1e-9) + 2e-6;  // the lookup computes u, v in f32
  const double uc = -1.0 + (iu + 0.5) * size, vc = -1.0 + (iv + 0.5) * size;
  const double g = p.c0 + p.cu * uc + p.cv * vc + p.cuu * uc * uc + p.cuv * uc * vc + p.cvv * vc * vc;
  const double gu = p.cu + 2.0 * p.cuu * uc + p.cuv * vc, gv = p.cv + 2.0 * p.cvv * vc + p.cuv * uc;
  const double reach = (std::fabs(gu) + std::fabs(gv)) * h + (std::fabs(p.cuu) + std::fabs(p.cuv) + std::fabs(p.cvv)) * h * h;
  const double um = std::fabs(uc) + h, vm = std::fabs(vc) + h;
  if (std::fabs(g) > reach + thr * (1.0 + um * um + vm * vm)) return;
  if (level_n == n) {
    emit(iv * n + iu);
    return;
  }
  for (int k = 0; k < 4; k++) raster(p, thr, n, level_n * 2, iu * 2 + (k & 1), iv * 2 + (k >> 1), emit);
}

template <class Emit>
void raster_band(const ConeBand& b, int n, Emit&& emit) {
  // |a(d)| < FEQ_EPSILON (1 + 1e-6) + slack |d|^2 is the device pre-filter for |d|^2 >= 1 - 1e-6
  const double thr = (FEQ_EPSILON * (1.0 + 1e-6) + b.slack) * (1.0 + 4e-6);
  for (int m = 0; m < 3; m++) {
    FacePoly p = face_poly(b, m);
    raster(p, thr, n, 1, 0, 0, [&](int cell) { emit(m * n * n + cell); });
  }
}

}  // namespace

BandGrid build_band_grid(const std::vector<ConeBand>& bands, size_t max_items) {
  BandGrid grid;
  const size_t nb = bands.size();
  if (nb <= 32) return grid;  // a short list is cheaper to scan than to index
  int n = 32;
  static const int n_max = getenv("RT_AMD_BAND_GRID_MAX") ? atoi(getenv("RT_AMD_BAND_GRID_MAX")) : 2048;  // experiment knob
  while (n < n_max && n < 1.7 * (double)nb) n *= 2;
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
