// Device-to-host copy rate for one 1080p RGB8 frame (6.2 MB) into page-locked memory:
// Build: make -C ray-tracer_amd probes (lib/d2h_probe; run it on the GPU box).
// one copy per frame on one stream, or the frame split over several streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const size_t bytes = 1920 * 1080 * 3;
  const int frames = 400;
  unsigned char* dev; unsigned char* host;
  CK(hipMalloc(&dev, bytes * 4));
  CK(hipHostMalloc(&host, bytes * 4, hipHostMallocDefault));
  CK(hipMemset(dev, 1, bytes * 4));
  for (int n_streams : {1, 2, 3, 4}) {
    std::vector<hipStream_t> st(n_streams);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int split = 0; split < 2; split++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
          size_t slot = (f & 3) * bytes;
          if (split) {
            size_t part = bytes / n_streams;
            for (int k = 0; k < n_streams; k++)
              CK(hipMemcpyAsync(host + slot + k * part, dev + slot + k * part, k == n_streams - 1 ? bytes - k * part : part, hipMemcpyDeviceToHost, st[k]));
          } else {
            CK(hipMemcpyAsync(host + slot, dev + slot, bytes, hipMemcpyDeviceToHost, st[f % n_streams]));
          }
        }
        CK(hipDeviceSynchronize());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        if (rep) printf("streams %d %s: %.4f ms per frame, %.1f GB/s\n", n_streams, split ? "frame split" : "whole frames", ms, bytes / ms * 1e-6);
      }
    }
  }
  // the frame loop's pattern: the copy stream waits for an event of a render stream, copies, records a fence
  {
    hipStream_t render, copy;
    CK(hipStreamCreateWithFlags(&render, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    hipEvent_t order, fence[4];
    CK(hipEventCreateWithFlags(&order, hipEventDisableTiming));
    for (auto& fe : fence) CK(hipEventCreateWithFlags(&fe, hipEventDisableTiming));
    for (int mode = 0; mode < 4; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
          size_t slot = (f & 3) * bytes;
          if (mode & 1) {
            CK(hipEventRecord(order, render));
            CK(hipStreamWaitEvent(copy, order, 0));
          }
          CK(hipMemcpyAsync(host + slot, dev + slot, bytes, hipMemcpyDeviceToHost, copy));
          if (mode & 2) CK(hipEventRecord(fence[f & 3], copy));
        }
        CK(hipDeviceSynchronize());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        if (rep) printf("wait on a render-stream event %d, fence after the copy %d: %.4f ms per frame, %.1f GB/s\n", mode & 1, (mode >> 1) & 1, ms, bytes / ms * 1e-6);
      }
    }
  }
  // one GPU's bands of a frame split over n GPUs: a strided copy, one copy per band, or (for scale) one
  // contiguous copy of as many bytes
  {
    hipStream_t copy;
    CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    const size_t band = 32 * 1920 * 3;
    for (int n : {2, 4, 8}) {
      const size_t bands = (1080 / 32 + n - 1) / n;  // whole bands of part 0, roughly
      for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          for (int f = 0; f < frames; f++) {
            size_t slot = (f & 3) * bytes;
            if (mode == 0) CK(hipMemcpy2DAsync(host + slot, n * band, dev + slot, n * band, band, bands, hipMemcpyDeviceToHost, copy));
            if (mode == 1)
              for (size_t b = 0; b < bands; b++)
                CK(hipMemcpyAsync(host + slot + b * n * band, dev + slot + b * n * band, band, hipMemcpyDeviceToHost, copy));
            if (mode == 2) CK(hipMemcpyAsync(host + slot, dev + slot, band * bands, hipMemcpyDeviceToHost, copy));
          }
          CK(hipDeviceSynchronize());
          double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
          if (rep) printf("%d parts, %zu bands of %zu bytes, %s: %.4f ms per frame, %.1f GB/s\n", n, bands, band,
                          mode == 0 ? "one strided copy" : mode == 1 ? "one copy per band" : "one contiguous copy", ms, band * bands / ms * 1e-6);
        }
      }
    }
  }
  // several whole frames per copy: what the fixed cost of a copy is worth
  {
    hipStream_t copy;
    CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    for (int per_copy : {1, 2, 4}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f += per_copy) {
          size_t slot = (f & 3) / per_copy * per_copy * bytes;
          CK(hipMemcpyAsync(host + slot, dev + slot, bytes * per_copy, hipMemcpyDeviceToHost, copy));
        }
        CK(hipDeviceSynchronize());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        if (rep) printf("%d whole frame(s) per copy: %.4f ms per frame, %.1f GB/s\n", per_copy, ms, bytes / ms * 1e-6);
      }
    }
  }
  // the same for bands of 8 rows (what the frame loop deals), the copies of consecutive frames going round-robin
  // to one, two or three copy streams: does the fixed cost of one copy overlap the transfer of the next?
  {
    hipStream_t copy[3];
    for (auto& c : copy) CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    const size_t band = 8 * 1920 * 3;
    for (int n : {2, 4, 8}) {
      const size_t bands = (1080 / 8 + n - 1) / n;
      for (int streams = 1; streams <= 3; streams++) {
        for (int rep = 0; rep < 2; rep++) {
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          for (int f = 0; f < frames; f++) {
            size_t slot = (f & 3) * bytes;
            CK(hipMemcpy2DAsync(host + slot, n * band, dev + slot, n * band, band, bands, hipMemcpyDeviceToHost, copy[f % streams]));
          }
          CK(hipDeviceSynchronize());
          double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
          if (rep) printf("%d parts, %zu bands of %zu bytes, one strided copy per frame, %d copy stream(s): %.4f ms per frame, %.1f GB/s\n", n, bands,
                          band, streams, ms, band * bands / ms * 1e-6);
        }
      }
    }
  }
  return 0;
}
