// Canvas::save (src/canvas.rs:47-94) on the device: an rgb8 frame in HBM becomes the text of the
// reference's PPM file, byte for byte, without the frame crossing to the host first.
//
// The writer of the reference is a state machine: `col`, the column it believes the line has
// reached, decides for every pixel which of its three separators is the line break.  Two facts
// make it parallel:
//   * the number of bytes a pixel emits does not depend on col (three numbers and three separators;
//     only the very first pixel of the file has no separator in front), so every byte's position is
//     a running sum of digit counts;
//   * col -> col' per pixel is a function on a domain of 73 values (col never exceeds 70 + 3), and
//     functions compose associatively: a block of pixels is summarised by a table of 73 entries.
// Pixels are cut into chunks of 64 (one wave) and spans of 256 chunks:
//   k_ppm_chunk_maps   table and byte count of every chunk            (parallel over chunks)
//   k_ppm_span_maps    table and byte count of every span             (parallel over spans)
//   k_ppm_walk_spans   the true col and byte offset at every span     (one wave, n / 16384 steps)
//   k_ppm_walk_chunks  the true col and byte offset at every chunk    (parallel over spans)
//   k_ppm_emit         the text                                       (parallel over chunks)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/rt_amd.h"
#include "rt_error.h"

namespace {

constexpr int CHUNK = 64;        // pixels per chunk: one per lane
constexpr int SPAN = 256;        // chunks per span
constexpr int STATES = 128;      // table entries per chunk; entry k is col = k + 1 (only 1..73 occur)
constexpr int WAVES_PER_BLOCK = 4;
constexpr int CHUNK_TEXT = CHUNK * 12;  // most bytes a chunk can emit

#define PPM_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      rc = rt_fail_message(RT_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
      goto done;                                                                                   \
    }                                                                                              \
  } while (0)

__device__ __forceinline__ int digits(uint32_t v) { return v < 10u ? 1 : (v < 100u ? 2 : 3); }

// One pixel of the loop in Canvas::save (src/canvas.rs:68-86): the column after the pixel.
__device__ __forceinline__ int ppm_step(int col, int lr, int lg, int lb) {
  if (col + lr > 70) return lr + lg + lb + 3;
  if (col + lr + lg > 70) return lg + lb + 2;
  if (col + lr + lg + lb > 70) return lb + 1;
  return col + (col != 1 ? 1 : 0) + lr + lg + lb + 2;
}

__device__ __forceinline__ int clamp_state(int col) { return col < STATES ? col : STATES; }

// The same step with everything that depends on the pixel alone worked out once per wave (the pixel
// is the same for all lanes, so these live in scalar registers) and no branches per state: seven
// vector instructions.  Valid for col >= 2, which is every column but the one before the first pixel.
struct PixelStep {
  int grow, t1, t2, t3, r1, r2, r3;
  __device__ __forceinline__ explicit PixelStep(uint32_t lens) {
    const int lr = lens & 3, lg = (lens >> 2) & 3, lb = (lens >> 4) & 3;
    grow = lr + lg + lb + 3;
    t1 = 70 - lr; t2 = 70 - lr - lg; t3 = 70 - lr - lg - lb;
    r1 = lr + lg + lb + 3; r2 = lg + lb + 2; r3 = lb + 1;
  }
  __device__ __forceinline__ int operator()(int col) const {
    int next = col + grow;
    next = col > t3 ? r3 : next;
    next = col > t2 ? r2 : next;
    next = col > t1 ? r1 : next;  // t1 >= t2 >= t3: the first condition of the reference wins
    return next;
  }
};

// lens of pixel k of the wave, as a scalar
__device__ __forceinline__ uint32_t lens_of(uint32_t lens, int k) {
  return (uint32_t)__builtin_amdgcn_readlane((int)lens, k);
}

// digit counts of pixel p packed as lr | lg << 2 | lb << 4 (0 for a lane beyond the image)
__device__ __forceinline__ uint32_t pixel_lens(const uint8_t* rgb8, uint64_t p, uint64_t n, uint32_t* rgb) {
  if (p >= n) {
    *rgb = 0;
    return 0;
  }
  const uint32_t r = rgb8[p * 3 + 0], g = rgb8[p * 3 + 1], b = rgb8[p * 3 + 2];
  *rgb = r | (g << 8) | (b << 16);
  return (uint32_t)digits(r) | ((uint32_t)digits(g) << 2) | ((uint32_t)digits(b) << 4);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t lane) {
  uint32_t p = v;
  for (uint32_t d = 1; d < 64; d <<= 1) {
    uint32_t u = __shfl_up(p, d, 64);
    if (lane >= d) p += u;
  }
  return p - v;
}

// Table lookup: entry (col - 1) of a 128-entry table held as two values per lane.
__device__ __forceinline__ int table_at(uint32_t lo, uint32_t hi, int col) {
  const int k = col - 1;
  const uint32_t a = __shfl(lo, k & 63, 64), b = __shfl(hi, k & 63, 64);
  return (int)(k < 64 ? a : b);
}

__global__ void __launch_bounds__(CHUNK* WAVES_PER_BLOCK)
    k_ppm_chunk_maps(const uint8_t* rgb8, uint64_t n, uint64_t n_chunks, uint8_t* maps, uint32_t* chunk_bytes) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t chunk = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;  // whole wave
  uint32_t rgb;
  const uint32_t lens = pixel_lens(rgb8, chunk * CHUNK + lane, n, &rgb);
  const uint32_t mine = lens ? (lens & 3u) + ((lens >> 2) & 3u) + ((lens >> 4) & 3u) + 3u : 0u;
  uint32_t total = wave_sum(mine);
  if (chunk == 0) total -= 1;  // the first pixel of the file has no separator in front
  const int count = (int)min((uint64_t)CHUNK, n - chunk * CHUNK);
  int s0 = (int)lane + 1, s1 = (int)lane + 65;
  {  // the first pixel with the general step: column 1 is among the states
    const uint32_t l = lens_of(lens, 0);
    const int lr = l & 3, lg = (l >> 2) & 3, lb = (l >> 4) & 3;
    s0 = ppm_step(s0, lr, lg, lb);
    s1 = ppm_step(s1, lr, lg, lb);
  }
#pragma unroll 8
  for (int k = 1; k < count; k++) {
    const PixelStep step(lens_of(lens, k));
    s0 = step(s0);
    s1 = step(s1);
  }
  s1 = clamp_state(s1);
  maps[chunk * STATES + lane] = (uint8_t)s0;
  maps[chunk * STATES + 64 + lane] = (uint8_t)s1;
  if (lane == 0) chunk_bytes[chunk] = total;
}

__global__ void __launch_bounds__(CHUNK* WAVES_PER_BLOCK)
    k_ppm_span_maps(const uint8_t* maps, const uint32_t* chunk_bytes, uint64_t n_chunks, uint64_t n_spans,
                    uint8_t* span_maps, unsigned long long* span_bytes) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t span = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (span >= n_spans) return;
  const uint64_t first = span * SPAN;
  const int count = (int)min((uint64_t)SPAN, n_chunks - first);
  int s0 = (int)lane + 1, s1 = (int)lane + 65;
  unsigned long long bytes = 0;
  for (int base = 0; base < count; base += 8) {  // eight tables in flight: their addresses are known
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t c = first + min(base + j, count - 1);
      lo[j] = maps[c * STATES + lane];
      hi[j] = maps[c * STATES + 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (base + j < count) {
        s0 = table_at(lo[j], hi[j], s0);
        s1 = table_at(lo[j], hi[j], s1);
      }
  }
  for (int k = (int)lane; k < count; k += 64) bytes += chunk_bytes[first + k];
  for (int d = 32; d > 0; d >>= 1) bytes += __shfl_xor(bytes, d, 64);
  span_maps[span * STATES + lane] = (uint8_t)s0;
  span_maps[span * STATES + 64 + lane] = (uint8_t)s1;
  if (lane == 0) span_bytes[span] = bytes;
}

struct PpmHeader {
  char text[48];
  uint32_t length;
};

// One wave walks the spans in order; the tables of eight spans are fetched together (their addresses
// do not depend on the walk), then applied one after the other.
__global__ void __launch_bounds__(64)
    k_ppm_walk_spans(const uint8_t* span_maps, const unsigned long long* span_bytes, uint64_t n_spans, PpmHeader header,
                     uint8_t* span_col, unsigned long long* span_offset, char* out, unsigned long long* out_length) {
  const uint32_t lane = threadIdx.x;
  int col = 1;  // src/canvas.rs:57
  unsigned long long offset = header.length;
  if (lane < header.length) out[lane] = header.text[lane];
  for (uint64_t base = 0; base < n_spans; base += 8) {
    uint32_t lo[8], hi[8];
    unsigned long long by[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t s = min(base + j, n_spans - 1);
      lo[j] = span_maps[s * STATES + lane];
      hi[j] = span_maps[s * STATES + 64 + lane];
      by[j] = span_bytes[s];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (base + j < n_spans) {
        if (lane == 0) {
          span_col[base + j] = (uint8_t)col;
          span_offset[base + j] = offset;
        }
        col = table_at(lo[j], hi[j], col);
        offset += by[j];
      }
    }
  }
  if (lane == 0) {
    out[offset] = '\n';  // src/canvas.rs:90
    *out_length = offset + 1;
  }
}

__global__ void __launch_bounds__(CHUNK* WAVES_PER_BLOCK)
    k_ppm_walk_chunks(const uint8_t* maps, const uint32_t* chunk_bytes, uint64_t n_chunks, uint64_t n_spans,
                      const uint8_t* span_col, const unsigned long long* span_offset, uint8_t* chunk_col,
                      unsigned long long* chunk_offset) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t span = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (span >= n_spans) return;
  const uint64_t first = span * SPAN;
  const int count = (int)min((uint64_t)SPAN, n_chunks - first);
  int col = span_col[span];
  unsigned long long offset = span_offset[span];
  for (int base = 0; base < count; base += 64) {
    // byte offsets of 64 chunks at a time
    const int k = base + (int)lane;
    const uint32_t mine = k < count ? chunk_bytes[first + k] : 0u;
    const uint32_t before = wave_exclusive_scan(mine, lane);
    if (k < count) chunk_offset[first + k] = offset + before;
    offset += wave_sum(mine);
  }
  for (int base = 0; base < count; base += 8) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t c = first + min(base + j, count - 1);
      lo[j] = maps[c * STATES + lane];
      hi[j] = maps[c * STATES + 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (base + j < count) {
        if (lane == 0) chunk_col[first + base + j] = (uint8_t)col;
        col = table_at(lo[j], hi[j], col);
      }
  }
}

__device__ __forceinline__ int put_number(char* dst, uint32_t v) {
  int k = 0;
  if (v >= 100u) dst[k++] = (char)('0' + v / 100u);
  if (v >= 10u) dst[k++] = (char)('0' + (v / 10u) % 10u);
  dst[k++] = (char)('0' + v % 10u);
  return k;
}

__global__ void __launch_bounds__(CHUNK* WAVES_PER_BLOCK)
    k_ppm_emit(const uint8_t* rgb8, uint64_t n, uint64_t n_chunks, const uint8_t* chunk_col,
               const unsigned long long* chunk_offset, char* out) {
  __shared__ char text[WAVES_PER_BLOCK][CHUNK_TEXT];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t chunk = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
  if (chunk >= n_chunks) return;
  uint32_t rgb;
  const uint32_t lens = pixel_lens(rgb8, chunk * CHUNK + lane, n, &rgb);
  const int lr = lens & 3, lg = (lens >> 2) & 3, lb = (lens >> 4) & 3;
  const int count = (int)min((uint64_t)CHUNK, n - chunk * CHUNK);
  // the column every pixel of the chunk starts from
  int col = chunk_col[chunk], my_col = 0;
  for (int k = 0; k < count; k++) {
    if ((int)lane == k) my_col = col;
    const uint32_t l = __shfl(lens, k, 64);
    col = ppm_step(col, l & 3, (l >> 2) & 3, (l >> 4) & 3);
  }
  const bool first_of_file = chunk == 0 && lane == 0;  // the only pixel that sees col == 1
  const uint32_t mine = lens ? (uint32_t)(lr + lg + lb + 3) - (first_of_file ? 1u : 0u) : 0u;
  const uint32_t at = wave_exclusive_scan(mine, lane);
  const uint32_t total = wave_sum(mine);
  if (lens) {
    // which separator is the line break (src/canvas.rs:68-86)
    const bool br0 = my_col + lr > 70;
    const bool br1 = !br0 && my_col + lr + lg > 70;
    const bool br2 = !br0 && !br1 && my_col + lr + lg + lb > 70;
    char* dst = text[wave] + at;
    int k = 0;
    if (!first_of_file) dst[k++] = br0 ? '\n' : ' ';
    k += put_number(dst + k, rgb & 255u);
    dst[k++] = br1 ? '\n' : ' ';
    k += put_number(dst + k, (rgb >> 8) & 255u);
    dst[k++] = br2 ? '\n' : ' ';
    k += put_number(dst + k, (rgb >> 16) & 255u);
  }
  // the other lanes' bytes: one wave, so its LDS accesses execute in program order
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  char* dst = out + chunk_offset[chunk];
  for (uint32_t j = lane; j < total; j += 64) dst[j] = text[wave][j];
}

}  // namespace

extern "C" uint64_t rt_ppm_max_bytes(uint64_t width, uint64_t height) { return 48 + 12 * width * height + 1; }

namespace {

// Workspace of one encode: tables, byte counts, columns, offsets, the length (about 2.2 bytes per pixel).
struct PpmLayout {
  uint64_t n_chunks, n_spans;
  uint64_t maps, span_maps, chunk_bytes, span_bytes, chunk_col, span_col, chunk_off, span_off, length, bytes;
};
constexpr uint64_t up256(uint64_t v) { return (v + 255) & ~255ull; }
constexpr PpmLayout ppm_layout(uint64_t n) {
  PpmLayout L{};
  L.n_chunks = (n + CHUNK - 1) / CHUNK;
  L.n_spans = (L.n_chunks + SPAN - 1) / SPAN;
  L.maps = 0;
  L.span_maps = L.maps + up256(L.n_chunks * STATES);
  L.chunk_bytes = L.span_maps + up256(L.n_spans * STATES);
  L.span_bytes = L.chunk_bytes + up256(L.n_chunks * 4);
  L.chunk_col = L.span_bytes + up256(L.n_spans * 8);
  L.span_col = L.chunk_col + up256(L.n_chunks);
  L.chunk_off = L.span_col + up256(L.n_spans);
  L.span_off = L.chunk_off + up256(L.n_chunks * 8);
  L.length = L.span_off + up256(L.n_spans * 8);
  L.bytes = L.length + 256;
  return L;
}

int ppm_check_arguments(const char* who, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                        uint64_t capacity, uint64_t* length) {
  const std::string name(who);
  if (!out_dev || !length || (!rgb8_dev && width * height > 0))
    return rt_fail_message(RT_ERR_INVALID, (name + ": null argument").c_str());
  if (width > 0xFFFFFFFFull || height > 0xFFFFFFFFull || (height && width > (1ull << 40) / height))
    return rt_fail_message(RT_ERR_INVALID, (name + ": image too large").c_str());
  if (capacity < rt_ppm_max_bytes(width, height))
    return rt_fail_message(RT_ERR_INVALID, (name + ": output buffer smaller than rt_ppm_max_bytes(width, height)").c_str());
  return RT_OK;
}

// The five kernels on `stream` over the workspace at ws (device memory of the current device, at least
// ppm_layout(width * height).bytes, contents irrelevant: every word the kernels read was written by an earlier
// kernel of the same call), then the length of the file.  Blocking.  Allocates nothing.
int ppm_run(char* ws, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev, uint64_t* length,
            hipStream_t stream) {
  int rc = RT_OK;
  const uint64_t n = width * height;
  const PpmLayout L = ppm_layout(n);
  const uint64_t n_chunks = L.n_chunks, n_spans = L.n_spans;
  PpmHeader header;
  memset(&header, 0, sizeof header);
  header.length = (uint32_t)snprintf(header.text, sizeof header.text, "P3\n%llu %llu\n255\n", (unsigned long long)width,
                                     (unsigned long long)height);  // src/canvas.rs:52-54
  unsigned long long host_length = 0;
  uint8_t* maps = (uint8_t*)(ws + L.maps);
  uint8_t* span_maps = (uint8_t*)(ws + L.span_maps);
  uint32_t* chunk_bytes = (uint32_t*)(ws + L.chunk_bytes);
  unsigned long long* span_bytes = (unsigned long long*)(ws + L.span_bytes);
  uint8_t* chunk_col = (uint8_t*)(ws + L.chunk_col);
  uint8_t* span_col = (uint8_t*)(ws + L.span_col);
  unsigned long long* chunk_off = (unsigned long long*)(ws + L.chunk_off);
  unsigned long long* span_off = (unsigned long long*)(ws + L.span_off);
  unsigned long long* d_length = (unsigned long long*)(ws + L.length);
  const dim3 block(CHUNK * WAVES_PER_BLOCK);
  const dim3 chunk_grid((unsigned)((n_chunks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
  const dim3 span_grid((unsigned)((n_spans + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
  if (n_chunks) {
    hipLaunchKernelGGL(k_ppm_chunk_maps, chunk_grid, block, 0, stream, rgb8_dev, n, n_chunks, maps, chunk_bytes);
    hipLaunchKernelGGL(k_ppm_span_maps, span_grid, block, 0, stream, maps, chunk_bytes, n_chunks, n_spans, span_maps,
                       span_bytes);
  }
  hipLaunchKernelGGL(k_ppm_walk_spans, dim3(1), dim3(64), 0, stream, span_maps, span_bytes, n_spans, header, span_col,
                     span_off, out_dev, d_length);
  if (n_chunks) {
    hipLaunchKernelGGL(k_ppm_walk_chunks, span_grid, block, 0, stream, maps, chunk_bytes, n_chunks, n_spans, span_col,
                       span_off, chunk_col, chunk_off);
    hipLaunchKernelGGL(k_ppm_emit, chunk_grid, block, 0, stream, rgb8_dev, n, n_chunks, chunk_col, chunk_off, out_dev);
  }
  PPM_TRY(hipGetLastError());
  PPM_TRY(hipMemcpyAsync(&host_length, d_length, sizeof host_length, hipMemcpyDeviceToHost, stream));
  PPM_TRY(hipStreamSynchronize(stream));
  *length = host_length;
done:
  return rc;
}

}  // namespace

// ---------------------------------------------------------------------------
// The same encoder for callers that cannot allocate (serving loops and harnesses that forbid hipMalloc):
// the workspace is a fixed array in the library's device image, one copy per GPU.  Like rt_ppm_encode it
// blocks until the file is written (it launches on the null stream and waits for the length), so it is not
// for use while a stream is being captured into a graph.
// ---------------------------------------------------------------------------
namespace {
constexpr uint64_t PPM_STATIC_MAX_PIXELS = 3840ull * 2160ull;  // one 4K frame
constexpr uint64_t PPM_STATIC_BYTES = ppm_layout(PPM_STATIC_MAX_PIXELS).bytes;  // about 18.4 MB per GPU
}  // namespace
// File scope: every GPU the library runs on holds its own copy at its own address until the process ends.
// In host code this name is only a handle for hipGetSymbolAddress; it is never passed to a kernel.
static __device__ __align__(256) unsigned char scratch[PPM_STATIC_BYTES];

namespace {
constexpr int PPM_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
char* g_scratch_ptr[PPM_MAX_DEVICES] = {nullptr};  // device address of `scratch` on each GPU, once looked up

// The address of this device's copy of the array.  The device must be current.
int scratch_on_current_device(int device, char** out) {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
      return rt_fail_message(RT_ERR_DEVICE, (std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)).c_str());
    g_scratch_ptr[device] = (char*)p;
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}

// One call at a time works in a GPU's array: calls on one GPU take turns.
std::mutex g_scratch_turn[PPM_MAX_DEVICES];

bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && *e && *e != '0';
}

// The encoder in the fixed workspace of `device` (current, frame within PPM_STATIC_MAX_PIXELS).  wait_for_turn:
// block until the array is free; otherwise return at once with *ran = false when another call is using it.
int ppm_run_in_scratch(int device, bool wait_for_turn, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                       uint64_t* length, bool* ran) {
  *ran = false;
  std::unique_lock<std::mutex> turn(g_scratch_turn[device], std::defer_lock);
  if (wait_for_turn) turn.lock();
  else if (!turn.try_lock()) return RT_OK;
  char* ws = nullptr;
  int rc = scratch_on_current_device(device, &ws);
  if (rc != RT_OK) return wait_for_turn ? rc : RT_OK;  // a caller that may allocate does so instead (*ran stays false)
  *ran = true;
  // nothing in the workspace has to start at zero: every word a kernel reads was written earlier in the same call
  return ppm_run(ws, rgb8_dev, width, height, out_dev, length, nullptr);
}
}  // namespace

// Frames up to one 3840x2160 work in the GPU's fixed array when no other call is using it (no hipMalloc / hipFree
// per frame); larger frames, and calls that find the array taken, allocate their own workspace.
extern "C" int rt_ppm_encode(int device, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                             uint64_t capacity, uint64_t* length) {
  int rc = ppm_check_arguments("rt_ppm_encode", rgb8_dev, width, height, out_dev, capacity, length);
  if (rc != RT_OK) return rc;
  char* ws = nullptr;
  int previous = -1;
  bool ran = false;
  PPM_TRY(hipGetDevice(&previous));
  PPM_TRY(hipSetDevice(device));
  // RT_AMD_PPM_ALLOCATE=1 (read per call): always allocate, as every call did before the fixed array existed
  if (width * height <= PPM_STATIC_MAX_PIXELS && device >= 0 && device < PPM_MAX_DEVICES && !env_flag("RT_AMD_PPM_ALLOCATE")) {
    rc = ppm_run_in_scratch(device, false, rgb8_dev, width, height, out_dev, length, &ran);
    if (rc != RT_OK || ran) goto done;
  }
  PPM_TRY(hipMalloc((void**)&ws, ppm_layout(width * height).bytes));
  rc = ppm_run(ws, rgb8_dev, width, height, out_dev, length, nullptr);
done:
  if (ws) (void)hipFree(ws);
  if (previous >= 0) (void)hipSetDevice(previous);
  return rc;
}

extern "C" uint64_t rt_ppm_static_max_pixels(void) { return PPM_STATIC_MAX_PIXELS; }

extern "C" int rt_ppm_encode_static(int device, const uint8_t* rgb8_dev, uint64_t width, uint64_t height, char* out_dev,
                                    uint64_t capacity, uint64_t* length) {
  int rc = ppm_check_arguments("rt_ppm_encode_static", rgb8_dev, width, height, out_dev, capacity, length);
  if (rc != RT_OK) return rc;
  if (width * height > PPM_STATIC_MAX_PIXELS)
    return rt_fail_message(RT_ERR_UNSUPPORTED, ("rt_ppm_encode_static: " + std::to_string(width * height) + " pixels exceed the fixed workspace of " +
                                                std::to_string(PPM_STATIC_MAX_PIXELS) + " (one 3840x2160 frame); use rt_ppm_encode").c_str());
  if (device < 0 || device >= PPM_MAX_DEVICES) return rt_fail_message(RT_ERR_INVALID, "rt_ppm_encode_static: device out of range");
  static_assert(ppm_layout(PPM_STATIC_MAX_PIXELS).bytes <= sizeof(scratch), "the fixed workspace holds the largest frame");
  int previous = -1;
  bool ran = false;
  PPM_TRY(hipGetDevice(&previous));
  PPM_TRY(hipSetDevice(device));
  rc = ppm_run_in_scratch(device, true, rgb8_dev, width, height, out_dev, length, &ran);
  (void)ran;
done:
  if (previous >= 0) (void)hipSetDevice(previous);
  return rc;
}

// The same for a frame and a file in host memory: the frame goes up, the text comes down.
extern "C" int rt_ppm_encode_host(int device, const uint8_t* rgb8_host, uint64_t width, uint64_t height, char* out_host,
                                  uint64_t capacity, uint64_t* length) {
  if (!out_host || !length || (!rgb8_host && width * height > 0))
    return rt_fail_message(RT_ERR_INVALID, "rt_ppm_encode_host: null argument");
  if (width > 0xFFFFFFFFull || height > 0xFFFFFFFFull || (height && width > (1ull << 40) / height))
    return rt_fail_message(RT_ERR_INVALID, "rt_ppm_encode_host: image too large");
  int rc = RT_OK;
  const uint64_t n = width * height, cap = rt_ppm_max_bytes(width, height);
  uint8_t* d_rgb = nullptr;
  char* d_out = nullptr;
  int previous = -1;
  uint64_t len = 0;
  PPM_TRY(hipGetDevice(&previous));
  PPM_TRY(hipSetDevice(device));
  PPM_TRY(hipMalloc((void**)&d_rgb, n * 3 + 4));
  PPM_TRY(hipMalloc((void**)&d_out, cap));
  if (n) PPM_TRY(hipMemcpy(d_rgb, rgb8_host, n * 3, hipMemcpyHostToDevice));
  rc = rt_ppm_encode(device, d_rgb, width, height, d_out, cap, &len);
  if (rc != RT_OK) goto done;
  if (len > capacity) {
    rc = rt_fail_message(RT_ERR_INVALID, "rt_ppm_encode_host: output buffer smaller than the file");
    *length = len;  // the size it takes
    goto done;
  }
  PPM_TRY(hipMemcpy(out_host, d_out, len, hipMemcpyDeviceToHost));
  *length = len;
done:
  if (d_rgb) (void)hipFree(d_rgb);
  if (d_out) (void)hipFree(d_out);
  if (previous >= 0) (void)hipSetDevice(previous);
  return rc;
}
