// Host-only self-checks exported through the C ABI: the threaded hierarchy builder and the cone
// direction index (run by the CPU test suite, no GPU needed).
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <vector>

#include "band_grid.h"
#include "band_grid_test.h"
#include "bvh_build.h"
#include "rt_scene.h"

using namespace rt;

// Host-only self-check of the cone direction index: n_bands random cone bands, n_dirs random unit
// directions (half of them placed inside a band); returns how many (direction, band) pairs that pass
// the band pre-filter are missing from the direction's grid cell (0 when the index is conservative),
// or a negative rt_status.  *grid_n receives the grid resolution, *entries its size.
extern "C" int rt_selftest_bvh_build(int n_items, uint64_t seed, int threads, uint64_t* hash, uint64_t* n_nodes,
                                     double* seconds) {
  if (n_items < 0 || threads < 0 || !hash) return fail(RT_ERR_INVALID, "rt_selftest_bvh_build: bad argument");
  uint64_t state = seed;
  auto uniform = [&state]() {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
  };
  std::vector<BuildItem> items((size_t)n_items);
  for (int i = 0; i < n_items; i++) {
    BuildItem& it = items[i];
    for (int a = 0; a < 3; a++) {
      double c = uniform() * 100.0, e = uniform() * 0.5;
      it.lo[a] = c - e;
      it.hi[a] = c + e;
    }
    it.kind = uniform() < 0.7 ? KIND_TRI : KIND_QUAD;
    it.index = (uint32_t)i;
  }
  BvhBuilder b;
  b.threads = threads;
  double lo[3], hi[3];
  auto t0 = std::chrono::steady_clock::now();
  uint32_t root = b.build(items, lo, hi);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t h = 0xCBF29CE484222325ull ^ root;  // FNV-1a over everything the build produced
  auto mix = [&h](const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 0x100000001B3ull;
  };
  if (!b.nodes.empty()) mix(b.nodes.data(), b.nodes.size() * sizeof(BvhNode));
  if (!b.tri_order.empty()) mix(b.tri_order.data(), b.tri_order.size() * sizeof(uint32_t));
  if (!b.quad_order.empty()) mix(b.quad_order.data(), b.quad_order.size() * sizeof(uint32_t));
  mix(lo, sizeof lo);
  mix(hi, sizeof hi);
  mix(&b.stack_need, sizeof b.stack_need);
  *hash = h;
  if (n_nodes) *n_nodes = b.nodes.size();
  return RT_OK;
}

extern "C" int rt_selftest_band_grid(int n_bands, int n_dirs, uint64_t seed, int* grid_n, uint64_t* entries) {
  if (n_bands < 0 || n_dirs < 0) return fail(RT_ERR_INVALID, "negative count");
  uint64_t state = seed;
  auto next = [&]() {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto uni = [&](double lo, double hi) { return lo + (hi - lo) * ((next() >> 11) * (1.0 / 9007199254740992.0)); };
  std::vector<ConeBand> bands(n_bands);
  std::vector<std::array<double, 9>> mats(n_bands);
  for (int i = 0; i < n_bands; i++) {
    // rows of a random rotation (from a random quaternion) divided by random scales: world -> object
    double qv[4], nq = 0;
    for (double& v : qv) { v = uni(-1, 1); nq += v * v; }
    nq = sqrt(nq);
    double w = qv[0] / nq, x = qv[1] / nq, y = qv[2] / nq, z = qv[3] / nq;
    double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                   2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    double sc[3] = {uni(0.2, 1.0), uni(0.2, 1.0), uni(0.2, 1.0)};
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) mats[i][r * 3 + c] = R[c * 3 + r] / sc[r];
    const double sg[3] = {1.0, -1.0, 1.0};
    double Q[3][3], sum = 0;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        Q[a][b] = 0;
        for (int r = 0; r < 3; r++) Q[a][b] += sg[r] * mats[i][r * 3 + a] * mats[i][r * 3 + b];
      }
    for (double v : mats[i]) sum += v * v;
    ConeBand& cb = bands[i];
    cb.q[0] = Q[0][0]; cb.q[1] = Q[1][1]; cb.q[2] = Q[2][2];
    cb.q[3] = 2 * Q[0][1]; cb.q[4] = 2 * Q[0][2]; cb.q[5] = 2 * Q[1][2];
    cb.slack = 1e-11 * sum;
    cb.prim = (uint32_t)i;
  }
  BandGrid grid = build_band_grid(bands, (size_t)1 << 28);
  if (grid_n) *grid_n = grid.n;
  if (entries) *entries = grid.items.size();
  if (grid.n == 0) return 0;
  int missing = 0;
  const int n = grid.n;
  for (int k = 0; k < n_dirs; k++) {
    double d[3];
    if ((k & 1) && n_bands > 0) {
      // a direction on the surface of cone i: object direction (cos p, +-1, sin p) mapped to world
      int i = (int)(next() % (uint64_t)n_bands);
      double p = uni(0, 6.283185307179586), ol[3] = {cos(p), (next() & 1) ? 1.0 : -1.0, sin(p)};
      // world = M^-1 object; M = diag(1/s) R^T, so M^-1 = R diag(s): solve by Cramer on the 3x3
      const double* m = mats[i].data();
      double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
      double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                       (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                       (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
      for (int r = 0; r < 3; r++) d[r] = inv[r * 3] * ol[0] + inv[r * 3 + 1] * ol[1] + inv[r * 3 + 2] * ol[2];
      if (next() & 1) {
        // pushed across the band along the gradient of a, to anywhere between its two edges
        const ConeBand& cb = bands[i];
        double l = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (double& v : d) v /= l;
        const double gr[3] = {2 * cb.q[0] * d[0] + cb.q[3] * d[1] + cb.q[4] * d[2], 2 * cb.q[1] * d[1] + cb.q[3] * d[0] + cb.q[5] * d[2],
                              2 * cb.q[2] * d[2] + cb.q[4] * d[0] + cb.q[5] * d[1]};
        const double g2 = gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2], target = uni(-0.999, 0.999) * FEQ_EPSILON;
        if (g2 > 1e-12)
          for (int r = 0; r < 3; r++) d[r] += target / g2 * gr[r];
      }
    } else {
      for (double& v : d) v = uni(-1, 1);
    }
    double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(len > 1e-3)) continue;
    for (double& v : d) v /= len;
    const double xx = d[0] * d[0], yy = d[1] * d[1], zz = d[2] * d[2], xy = d[0] * d[1], xz = d[0] * d[2], yz = d[1] * d[2];
    const double dd = xx + yy + zz;
    int m = zz > yy ? 2 : 1;
    m = xx >= (m == 2 ? zz : yy) ? 0 : m;
    const double dm = d[m], da = d[(m + 1) % 3], dc = d[(m + 2) % 3];
    const float inv = 1.0f / (float)dm, half_n = 0.5f * (float)n;
    int iu, iv;
    float ls, lt;  // the cell and the local coordinates as the traversal computes them
    band_strip_coords(((float)da * inv + 1.0f) * half_n, ((float)dc * inv + 1.0f) * half_n, n, iu, iv, ls, lt);
    const uint32_t cell = ((uint32_t)m * n + iv) * n + iu;
    for (int i = 0; i < n_bands; i++) {
      const ConeBand& cb = bands[i];
      double a = cb.q[0] * xx + cb.q[1] * yy + cb.q[2] * zz + cb.q[3] * xy + cb.q[4] * xz + cb.q[5] * yz;
      if (!(fabs(a) < FEQ_EPSILON * (1.0 + 1e-6) + cb.slack * dd)) continue;
      auto at = std::lower_bound(grid.items.begin() + grid.start[cell], grid.items.begin() + grid.start[cell + 1], (uint32_t)i);
      bool found = at != grid.items.begin() + grid.start[cell + 1] && *at == (uint32_t)i;
      // listed in the cell, and the entry's strip holds the direction
      if (!found || !band_strip_passes(grid.strip[(size_t)(at - grid.items.begin())], ls, lt, ls * 1024.0f, lt * 1024.0f)) missing++;
    }
  }
  return missing;
}

extern "C" int rt_selftest_world_normal(const double* full_inv16, const double* local_normal3, double* out3) {
  if (!full_inv16 || !local_normal3 || !out3) return fail(RT_ERR_INVALID, "rt_selftest_world_normal: null argument");
  world_normal(full_inv16, local_normal3, out3);
  return RT_OK;
}
