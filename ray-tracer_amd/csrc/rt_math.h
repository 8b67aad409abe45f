// Device-side arithmetic of the render path: points, vectors and rays in f64 with the reference's
// operation order (src/tuple.rs, src/matrix.rs, src/ray.rs), and the traversal counters.
// Build with -ffp-contract=off: the reference (Rust) never fuses a multiply with an add.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rt_amd.h"
#include "rt_types.h"

namespace rt {

#define DEV __device__ __forceinline__

struct V3 {
  double x, y, z;
};
struct Ray {
  V3 o, d;
};

DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
// Tuple4D::dot with both w == 0 (src/tuple.rs:145-150): the w product adds +0.0
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) {  // src/tuple.rs:160-167
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DEV double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // src/tuple.rs:111-118
DEV V3 normalize(V3 a) {  // src/tuple.rs:126-135: multiply by the reciprocal
  double mag = magnitude(a);
  return v3(a.x * (1.0 / mag), a.y * (1.0 / mag), a.z * (1.0 / mag));
}
DEV V3 reflect(V3 v, V3 n) { return v - ((n * 2.0) * dot(v, n)); }  // src/tuple.rs:170-172
DEV bool feq(double a, double b) { return fabs(a - b) < FEQ_EPSILON; }

// Matrix4D * point (w = 1) and * vector (w = 0), rows 0..2 (src/matrix.rs:638-653)
DEV V3 mul_point(const double* m, V3 p) {
  return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
DEV V3 mul_vector(const double* m, V3 v) {
  return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
DEV Ray transform_ray(const double* m, const Ray& r) {  // src/ray.rs:50-55
  return Ray{mul_point(m, r.o), mul_vector(m, r.d)};
}

struct Counters {
  unsigned long long node_visits, tri_tests, quad_tests, gate_tests;
#ifdef RT_STAMPS
  unsigned long long* sink;  // four sums: unbounded + setup, node loop, triangle runs, quadric runs + gates
  unsigned long long tt;
#endif
#if defined(RT_WAVE_CLOCK_IMAGE) && RT_WAVE_CLOCK_IMAGE == 2
  unsigned wi[4];  // this lane's share of the wave-level loop iterations (it counts those it leads)
#endif
};
#ifdef RT_STAMPS
// one count per wave and loop iteration (by the first active lane): lane events / (64 x these) = lane utilisation
#define RT_WAVE_ITER(c, k) do { if ((c).sink && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd((c).sink + 4 + (k), 1ull); } while (0)
#define RT_TBEGIN(c) (c).tt = __builtin_amdgcn_s_memtime()
#define RT_TSTAMP(c, k) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if ((c).sink && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd((c).sink + (k), now_ - (c).tt); \
    (c).tt = now_; } while (0)
#elif defined(RT_WAVE_CLOCK_IMAGE) && RT_WAVE_CLOCK_IMAGE == 2
#define RT_TBEGIN(c) do { } while (0)
#define RT_TSTAMP(c, k) do { } while (0)
#define RT_WAVE_ITER(c, k) do { if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) (c).wi[k]++; } while (0)
#else
#define RT_TBEGIN(c) do { } while (0)
#define RT_TSTAMP(c, k) do { } while (0)
#define RT_WAVE_ITER(c, k) do { } while (0)
#endif

}  // namespace rt
