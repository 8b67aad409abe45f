// How fast one SIMD issues the vector instructions a triangle test could be made of: v_fma_f64, v_fma_f32,
// v_pk_fma_f32, v_cvt_f32_f64 and the f64 division, at one and at four waves per SIMD.  Eight independent chains
// per lane, so latency does not bound a wave.  Prints nanoseconds per wave-level instruction and SIMD.
// Build: make -C ray-tracer_amd probes (lib/issue_rate_probe; run it on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CHAINS = 8, ROUNDS = 4096;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) k_issue(double* out, double seed) {
  double a[CHAINS];
  float f[CHAINS];
  f32x2 p[CHAINS];
  const double m = seed * 0.999, c = seed * 1e-9;
  const float mf = (float)m, cf = (float)c;
  const f32x2 mp = {mf, mf}, cp = {cf, cf};
#pragma unroll
  for (int k = 0; k < CHAINS; k++) {
    a[k] = seed + k + threadIdx.x;
    f[k] = (float)a[k];
    p[k] = f32x2{f[k], f[k] + 1.0f};
  }
  for (int r = 0; r < ROUNDS; r++) {
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
      if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(mf), "v"(cf));
      if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(mp), "v"(cp));
      if (KIND == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));
      if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(m));
      if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      if (KIND == 6) a[k] = 1.0 / (a[k] + c);  // the compiler's division sequence
      if (KIND == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < CHAINS; k++) s += a[k] + f[k] + p[k].x + p[k].y;
  if (s == 12345.678) out[0] = s;  // never, but the compiler cannot know
}

template <int KIND>
static void run(const char* what, double* out, int cus) {
  for (int waves = 1; waves <= 4; waves *= 4) {
    const int blocks = cus * 4 * waves;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_issue<KIND><<<blocks, 64>>>(out, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_issue<KIND><<<blocks, 64>>>(out, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)waves * CHAINS * ROUNDS;  // wave-level statements one SIMD went through
    printf("%-28s %d wave(s) per SIMD: %7.3f ms, %6.2f ns per wave-level statement and SIMD\n", what, waves, ms,
           ms * 1e6 / per_simd);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs, %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  double* out;
  CHECK(hipMalloc(&out, 64));
  run<0>("v_fma_f64", out, prop.multiProcessorCount);
  run<4>("v_mul_f64", out, prop.multiProcessorCount);
  run<5>("v_add_f64", out, prop.multiProcessorCount);
  run<1>("v_fma_f32", out, prop.multiProcessorCount);
  run<2>("v_pk_fma_f32", out, prop.multiProcessorCount);
  run<3>("v_cvt_f32_f64", out, prop.multiProcessorCount);
  run<7>("v_rcp_f64", out, prop.multiProcessorCount);
  run<6>("1.0 / x in f64 (sequence)", out, prop.multiProcessorCount);
  CHECK(hipFree(out));
  return 0;
}
