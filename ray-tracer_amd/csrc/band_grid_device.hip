// The cone direction index (band_grid.h) built on the GPU.  The host builder walks a quadtree per cone
// and face; with 25 000 cones and 2048 x 2048 cells per face that is about 4e8 cell tests and 2e8
// entries, two seconds on sixteen host cores and the largest part of loading such a scene.  Here:
//   1. one thread per (cone, face) walks the quadtree down to MID x MID cells and counts, then (after a
//      scan over the counts) lists the cells the band reaches;
//   2. one thread per listed cell walks the rest of the way to the finest level and counts every cell
//      it reaches in that cell's counter; a scan turns the counters into the offsets of the index;
//   3. the same walk again, now storing the cone at the next free place of its cell;
//   4. one thread per cell sorts its entries (the order the atomics produced is arbitrary; the host
//      builder sorts too).
// Every test is band_cell_passes, the function the host builder uses, so the two indexes are equal
// word for word (checked by rt_selftest_band_grid_device).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "band_grid.h"
#include "band_grid_test.h"
#include "rt_error.h"

namespace rt {
namespace {

#define BG_TRY(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      rc = rt_fail_message(RT_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str());    \
      goto done;                                                                                            \
    }                                                                                                       \
  } while (0)

constexpr int THREADS = 256;

// Depth-first walk from cell (iu, iv) of a from_n x from_n face (which passed) to the to_n x to_n level;
// visit(iu, iv) for every cell reached there.  Children in the host builder's order.
template <class Visit>
__device__ void walk(const FacePoly& p, double thr, int from_n, int to_n, int iu, int iv, Visit&& visit) {
  // entry: level shift in the top 4 bits (level_n = from_n << shift), then iu, iv relative to the start cell
  uint32_t stack[40];
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const uint32_t e = stack[--sp];
    const int shift = (int)(e >> 28), du = (int)((e >> 14) & 0x3FFFu), dv = (int)(e & 0x3FFFu);
    const int level_n = from_n << shift, cu = (iu << shift) + du, cv = (iv << shift) + dv;
    if (shift && !band_cell_passes(p, thr, level_n, cu, cv)) continue;  // the start cell passed already
    if (level_n == to_n) {
      visit(cu, cv);
      continue;
    }
    for (int k = 3; k >= 0; k--)  // popped in the order k = 0, 1, 2, 3
      stack[sp++] = ((uint32_t)(shift + 1) << 28) | ((uint32_t)(du * 2 + (k & 1)) << 14) | (uint32_t)(dv * 2 + (k >> 1));
  }
}

// Adds the lanes' counts of one wave to *total with one atomic (the 32-bit scans cannot tell a
// wrapped sum from a small one; this 64-bit total can).
__device__ void add_to_total(unsigned long long* total, uint32_t mine) {
  unsigned long long v = mine;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63u) == 0 && v) atomicAdd(total, v);
}

// Step 1.  FILL = false: counts[t] = cells of the MID level band-and-face t reaches.  FILL = true:
// lists them at offsets[t]: (band, face * mid * mid + iv * mid + iu).
template <bool FILL>
__global__ void __launch_bounds__(THREADS) k_mid(const ConeBand* bands, uint32_t n_bands, int mid, uint32_t* counts,
                                                 const uint32_t* offsets, uint2* list, unsigned long long* total) {
  const uint32_t t = blockIdx.x * THREADS + threadIdx.x;
  const bool live = t < 3u * n_bands;  // no early return: the wave sums below need every lane
  uint32_t n = 0;
  if (live) {
    const uint32_t band = t / 3u, face = t % 3u;
    const ConeBand b = bands[band];
    const FacePoly p = face_poly(b, (int)face);
    const double thr = band_threshold(b);
    const uint32_t base = FILL ? offsets[t] : 0u;
    if (band_cell_passes(p, thr, 1, 0, 0))
      walk(p, thr, 1, mid, 0, 0, [&](int iu, int iv) {
        if (FILL) list[base + n] = make_uint2(band, (face * (uint32_t)mid + (uint32_t)iv) * (uint32_t)mid + (uint32_t)iu);
        n++;
      });
    if (!FILL) counts[t] = n;
  }
  if (!FILL) add_to_total(total, n);
}

// Steps 2 and 3.  FILL = false: cell_count[cell]++ for every finest cell below a listed cell.
// FILL = true: items[cursor[cell]++] = band.
template <bool FILL>
__global__ void __launch_bounds__(THREADS) k_fine(const ConeBand* bands, const uint2* list, uint32_t n_list, int mid, int n,
                                                  uint32_t* cell_counter, uint32_t* items, unsigned long long* total) {
  const uint32_t t = blockIdx.x * THREADS + threadIdx.x;
  uint32_t mine = 0;
  if (t < n_list) {
    const uint2 e = list[t];
    const uint32_t face = e.y / ((uint32_t)mid * (uint32_t)mid), in_face = e.y % ((uint32_t)mid * (uint32_t)mid);
    const ConeBand b = bands[e.x];
    const FacePoly p = face_poly(b, (int)face);
    const double thr = band_threshold(b);
    walk(p, thr, mid, n, (int)(in_face % (uint32_t)mid), (int)(in_face / (uint32_t)mid), [&](int iu, int iv) {
      const size_t cell = ((size_t)face * n + iv) * n + iu;
      const uint32_t at = atomicAdd(&cell_counter[cell], 1u);
      if (FILL) items[at] = e.x;
      mine++;
    });
  }
  if (!FILL) add_to_total(total, mine);
}

// Step 4: ascending band index inside every cell.
__global__ void __launch_bounds__(THREADS) k_sort_cells(const uint32_t* start, uint32_t n_cells, uint32_t* items) {
  const uint32_t c = blockIdx.x * THREADS + threadIdx.x;
  if (c >= n_cells) return;
  const uint32_t lo = start[c], hi = start[c + 1];
  for (uint32_t i = lo + 1; i < hi; i++) {
    const uint32_t v = items[i];
    uint32_t j = i;
    while (j > lo && items[j - 1] > v) {
      items[j] = items[j - 1];
      j--;
    }
    items[j] = v;
  }
}

int exclusive_sum(const uint32_t* in, uint32_t* out, size_t n) {  // out[i] = in[0] + ... + in[i - 1], 0 or a hipError
  void* tmp = nullptr;
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n);
  if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 1);
  if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, (int)n);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (tmp) (void)hipFree(tmp);
  return (int)e;
}

}  // namespace

int build_band_grid_device(const ConeBand* d_bands, size_t n_bands, size_t max_items, int* n_out, uint32_t** d_start,
                           uint32_t** d_items, size_t* n_items) {
  *n_out = 0;
  *d_start = *d_items = nullptr;
  *n_items = 0;
  if (n_bands <= 32) return RT_OK;  // a short list is cheaper to scan than to index
  // (the scans below index with int; 2^27 is also the most primitives a scene can hold)
  if (n_bands > ((size_t)1 << 27)) return rt_fail_message(RT_ERR_UNSUPPORTED, "too many cones for the direction index");
  int rc = RT_OK;
  const uint32_t nb = (uint32_t)n_bands, n_threads = 3u * nb;
  uint32_t *mid_counts = nullptr, *mid_offsets = nullptr, *counter = nullptr, *start = nullptr, *items = nullptr;
  uint2* list = nullptr;
  unsigned long long* totals = nullptr;  // [0]: cells listed at the MID level, [1]: entries of the index
  unsigned long long host_totals[2] = {0, 0};
  int mid_cap = 256;
  BG_TRY(hipMalloc(&totals, 16));
  for (int n = band_grid_first_n(n_bands);;) {
    const int mid = n < mid_cap ? n : mid_cap;
    const size_t cells = (size_t)3 * n * n;
    uint32_t last[2] = {0, 0};
    uint32_t n_list = 0, total = 0;
    // step 1
    BG_TRY(hipMalloc(&mid_counts, ((size_t)n_threads + 1) * 4));
    BG_TRY(hipMalloc(&mid_offsets, ((size_t)n_threads + 1) * 4));
    BG_TRY(hipMemset(mid_counts, 0, ((size_t)n_threads + 1) * 4));
    BG_TRY(hipMemset(totals, 0, 16));
    hipLaunchKernelGGL(k_mid<false>, dim3((n_threads + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, d_bands, nb, mid,
                       mid_counts, (const uint32_t*)nullptr, (uint2*)nullptr, totals);
    BG_TRY(hipGetLastError());
    BG_TRY(hipMemcpy(host_totals, totals, 8, hipMemcpyDeviceToHost));
    if (host_totals[0] > 0x7FFFFFFFull && mid_cap > 1) {  // too many cells for the 32-bit list: a coarser MID level
      (void)hipFree(mid_counts); (void)hipFree(mid_offsets);
      mid_counts = mid_offsets = nullptr;
      mid_cap /= 2;
      continue;
    }
    BG_TRY((hipError_t)exclusive_sum(mid_counts, mid_offsets, (size_t)n_threads + 1));
    BG_TRY(hipMemcpy(&n_list, mid_offsets + n_threads, 4, hipMemcpyDeviceToHost));
    BG_TRY(hipMalloc(&list, ((size_t)n_list + 1) * sizeof(uint2)));
    hipLaunchKernelGGL(k_mid<true>, dim3((n_threads + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, d_bands, nb, mid,
                       mid_counts, (const uint32_t*)mid_offsets, list, totals);
    BG_TRY(hipGetLastError());
    // step 2
    BG_TRY(hipMalloc(&counter, (cells + 1) * 4));
    BG_TRY(hipMalloc(&start, (cells + 1) * 4));
    BG_TRY(hipMemset(counter, 0, (cells + 1) * 4));
    if (n_list)
      hipLaunchKernelGGL(k_fine<false>, dim3((n_list + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, d_bands,
                         (const uint2*)list, n_list, mid, n, counter, (uint32_t*)nullptr, totals + 1);
    BG_TRY(hipGetLastError());
    BG_TRY(hipMemcpy(host_totals, totals, 16, hipMemcpyDeviceToHost));
    if (host_totals[1] > (unsigned long long)max_items && n > 32) {  // coarser cells: fewer entries per band
      (void)hipFree(mid_counts); (void)hipFree(mid_offsets); (void)hipFree(list); (void)hipFree(counter); (void)hipFree(start);
      mid_counts = mid_offsets = counter = start = nullptr;
      list = nullptr;
      n /= 2;
      continue;
    }
    if (host_totals[1] > 0xFFFFFFFFull) {
      rc = rt_fail_message(RT_ERR_UNSUPPORTED, "the cone direction index would hold more than 2^32 entries");
      goto done;
    }
    BG_TRY((hipError_t)exclusive_sum(counter, start, cells + 1));
    BG_TRY(hipMemcpy(last, start + cells, 4, hipMemcpyDeviceToHost));
    total = last[0];
    // steps 3 and 4: the offsets double as the fill cursors of a copy
    BG_TRY(hipMalloc(&items, ((size_t)total + 1) * 4));
    BG_TRY(hipMemcpy(counter, start, (cells + 1) * 4, hipMemcpyDeviceToDevice));
    if (n_list)
      hipLaunchKernelGGL(k_fine<true>, dim3((n_list + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, d_bands,
                         (const uint2*)list, n_list, mid, n, counter, items, totals + 1);
    BG_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_sort_cells, dim3((uint32_t)((cells + THREADS - 1) / THREADS)), dim3(THREADS), 0, 0,
                       (const uint32_t*)start, (uint32_t)cells, items);
    BG_TRY(hipGetLastError());
    BG_TRY(hipDeviceSynchronize());
    *n_out = n;
    *d_start = start;
    *d_items = items;
    *n_items = total;
    start = items = nullptr;
    break;
  }
done:
  if (totals) (void)hipFree(totals);
  if (mid_counts) (void)hipFree(mid_counts);
  if (mid_offsets) (void)hipFree(mid_offsets);
  if (list) (void)hipFree(list);
  if (counter) (void)hipFree(counter);
  if (start) (void)hipFree(start);
  if (items) (void)hipFree(items);
  return rc;
}

// The strip word of every entry: one thread per cell walks the cell's entries.
namespace {
__global__ void __launch_bounds__(THREADS) k_strips(const ConeBand* bands, int n, const uint32_t* start, const uint32_t* items,
                                                    uint32_t n_cells, uint32_t* strip) {
  const uint32_t c = blockIdx.x * THREADS + threadIdx.x;
  if (c >= n_cells) return;
  const uint32_t nn = (uint32_t)n * (uint32_t)n;
  const int face = (int)(c / nn), iv = (int)((c % nn) / (uint32_t)n), iu = (int)(c % (uint32_t)n);
  for (uint32_t k = start[c]; k < start[c + 1]; k++) {
    const ConeBand b = bands[items[k]];
    strip[k] = band_strip_word(face_poly(b, face), band_threshold(b), n, iu, iv);
  }
}
}  // namespace

int build_band_strips_device(const ConeBand* d_bands, int n, const uint32_t* d_start, const uint32_t* d_items, size_t n_items,
                             uint32_t** out) {
  *out = nullptr;
  uint32_t* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, (n_items + 4) * sizeof(uint32_t));  // a ray reads four words at a time
  if (e != hipSuccess) return rt_fail_message(RT_ERR_NOMEM, "no device memory for the cone direction index");
  e = hipMemset(p, 0, (n_items + 4) * sizeof(uint32_t));
  const uint32_t cells = 3u * (uint32_t)n * (uint32_t)n;
  if (e == hipSuccess && n_items)
    hipLaunchKernelGGL(k_strips, dim3((cells + THREADS - 1) / THREADS), dim3(THREADS), 0, 0, d_bands, n, d_start, d_items, cells, p);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    return rt_fail_message(RT_ERR_DEVICE, (std::string("k_strips: ") + hipGetErrorString(e)).c_str());
  }
  *out = p;
  return RT_OK;
}

// The entries of a finished index as 16-bit numbers (for fewer than 65536 cones): half the memory, and
// half the bytes a ray reads per entry.  Frees the 32-bit array.
namespace {
__global__ void __launch_bounds__(THREADS) k_items_to_u16(const uint32_t* in, size_t n, uint16_t* out) {
  const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  if (i < n) out[i] = (uint16_t)in[i];
}
}  // namespace

int compress_band_items_u16(uint32_t* d_items, size_t n_items, uint16_t** out) {
  *out = nullptr;
  uint16_t* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, (n_items + 2) * sizeof(uint16_t));
  if (e != hipSuccess) return rt_fail_message(RT_ERR_NOMEM, "no device memory for the cone direction index");
  if (n_items) hipLaunchKernelGGL(k_items_to_u16, dim3((unsigned)((n_items + THREADS - 1) / THREADS)), dim3(THREADS), 0, 0, (const uint32_t*)d_items, n_items, p);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    return rt_fail_message(RT_ERR_DEVICE, (std::string("k_items_to_u16: ") + hipGetErrorString(e)).c_str());
  }
  (void)hipFree(d_items);
  *out = p;
  return RT_OK;
}

}  // namespace rt

// Diagnostic: builds the index over n_bands random cones on the host and on `device` and compares
// the two word for word.  Returns 0 when they are equal, 1 when they differ, or a negative rt_status.
extern "C" int rt_selftest_band_grid_device(int n_bands, uint64_t seed, int device, int* grid_n, uint64_t* entries,
                                            double* host_seconds, double* device_seconds) {
  using namespace rt;
  if (n_bands < 0) return rt_fail_message(RT_ERR_INVALID, "negative count");
  uint64_t state = seed;
  auto uni = [&](double lo, double hi) {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return lo + (hi - lo) * (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
  };
  std::vector<ConeBand> bands((size_t)n_bands);
  for (int i = 0; i < n_bands; i++) {
    // world -> object: rows of a random rotation divided by random scales; a(d) = |M d|^2 under diag(1, -1, 1)
    double qv[4], nq = 0;
    for (double& v : qv) {
      v = uni(-1, 1);
      nq += v * v;
    }
    nq = sqrt(nq) + 1e-300;
    const double w = qv[0] / nq, x = qv[1] / nq, y = qv[2] / nq, z = qv[3] / nq;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w),
                         1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w),
                         1 - 2 * (x * x + y * y)};
    double M[9], sum = 0;
    for (int r = 0; r < 3; r++) {
      const double sc = uni(0.2, 1.0);
      for (int c = 0; c < 3; c++) {
        M[r * 3 + c] = R[c * 3 + r] / sc;
        sum += M[r * 3 + c] * M[r * 3 + c];
      }
    }
    const double sg[3] = {1.0, -1.0, 1.0};
    double Q[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        Q[a][b] = 0;
        for (int r = 0; r < 3; r++) Q[a][b] += sg[r] * M[r * 3 + a] * M[r * 3 + b];
      }
    ConeBand& cb = bands[(size_t)i];
    cb.q[0] = Q[0][0]; cb.q[1] = Q[1][1]; cb.q[2] = Q[2][2];
    cb.q[3] = 2 * Q[0][1]; cb.q[4] = 2 * Q[0][2]; cb.q[5] = 2 * Q[1][2];
    cb.slack = 1e-11 * sum;
    cb.prim = (uint32_t)i;
    cb.pad = 0;
  }
  const size_t max_items = (size_t)1 << 28;
  auto t0 = std::chrono::steady_clock::now();
  BandGrid host = build_band_grid(bands, max_items);
  auto t1 = std::chrono::steady_clock::now();
  if (host_seconds) *host_seconds = std::chrono::duration<double>(t1 - t0).count();
  if (hipSetDevice(device) != hipSuccess) return rt_fail_message(RT_ERR_DEVICE, "hipSetDevice failed");
  ConeBand* d_bands = nullptr;
  if (hipMalloc(&d_bands, (bands.size() + 1) * sizeof(ConeBand)) != hipSuccess) return rt_fail_message(RT_ERR_DEVICE, "hipMalloc failed");
  (void)hipMemcpy(d_bands, bands.data(), bands.size() * sizeof(ConeBand), hipMemcpyHostToDevice);
  int n = 0;
  uint32_t *d_start = nullptr, *d_items = nullptr;
  size_t n_items = 0;
  t0 = std::chrono::steady_clock::now();
  int rc = build_band_grid_device(d_bands, bands.size(), max_items, &n, &d_start, &d_items, &n_items);
  t1 = std::chrono::steady_clock::now();
  if (device_seconds) *device_seconds = std::chrono::duration<double>(t1 - t0).count();
  (void)hipFree(d_bands);
  if (rc != RT_OK) return rc;
  if (grid_n) *grid_n = n;
  if (entries) *entries = n_items;
  int differ = (n != host.n || n_items != host.items.size()) ? 1 : 0;
  if (!differ && n > 0) {
    std::vector<uint32_t> start(host.start.size()), items(host.items.size());
    (void)hipMemcpy(start.data(), d_start, start.size() * 4, hipMemcpyDeviceToHost);
    if (!items.empty()) (void)hipMemcpy(items.data(), d_items, items.size() * 4, hipMemcpyDeviceToHost);
    differ = (start != host.start || items != host.items) ? 1 : 0;
    if (!differ) {  // the strip words, computed on the device from the device's own index
      uint32_t* d_strip = nullptr;
      ConeBand* d_again = nullptr;
      if (hipMalloc(&d_again, (bands.size() + 1) * sizeof(ConeBand)) != hipSuccess) return rt_fail_message(RT_ERR_DEVICE, "hipMalloc failed");
      (void)hipMemcpy(d_again, bands.data(), bands.size() * sizeof(ConeBand), hipMemcpyHostToDevice);
      rc = build_band_strips_device(d_again, n, d_start, d_items, n_items, &d_strip);
      (void)hipFree(d_again);
      if (rc == RT_OK) {
        std::vector<uint32_t> strip(host.strip.size());
        if (!strip.empty()) (void)hipMemcpy(strip.data(), d_strip, strip.size() * 4, hipMemcpyDeviceToHost);
        differ = strip != host.strip ? 1 : 0;
        (void)hipFree(d_strip);
      }
    }
  }
  if (d_start) (void)hipFree(d_start);
  if (d_items) (void)hipFree(d_items);
  if (rc != RT_OK) return rc;
  return differ;
}
