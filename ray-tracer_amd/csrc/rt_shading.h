// Shading-side device routines: patterns (src/pattern.rs:97-189), local normals (src/shape.rs:805-1332),
// the specular power, quantisation and pixel stores (src/canvas.rs:60-62, 114-121), ray queues.
#pragma once
#include "rt_records.h"

namespace rt {

DEV double rem_euclid2(double x) {  // f64::rem_euclid(2.0)
  double r = fmod(x, 2.0);
  return r < 0.0 ? r + 2.0 : r;
}

// Pattern::pattern_at (src/pattern.rs:97-189).  Blend operands are evaluated at
// the same point with no transform of their own (src/pattern.rs:185-186).
template <int DEPTH>
DEV void pattern_at(const SceneDev& S, int idx, V3 p, double& r, double& g, double& b) {
  const PatternDev& pt = S.patterns[idx];
  bool primary = true;
  switch (pt.type) {
    case RT_PAT_NULL: r = g = b = 0.0; return;
    case RT_PAT_POINT: r = p.x; g = p.y; b = p.z; return;
    case RT_PAT_IDENTITY: r = pt.a[0]; g = pt.a[1]; b = pt.a[2]; return;
    case RT_PAT_STRIPE: primary = feq(rem_euclid2(floor(p.x)), 0.0); break;
    case RT_PAT_RING: primary = feq(rem_euclid2(floor(sqrt(p.x * p.x + p.z * p.z))), 0.0); break;
    case RT_PAT_CHECKER: primary = feq(rem_euclid2(floor(p.x) + floor(p.y) + floor(p.z)), 0.0); break;
    case RT_PAT_GRADIENT: {
      double fraction = p.x - floor(p.x);
      r = pt.a[0] + (pt.b[0] - pt.a[0]) * fraction;
      g = pt.a[1] + (pt.b[1] - pt.a[1]) * fraction;
      b = pt.a[2] + (pt.b[2] - pt.a[2]) * fraction;
      return;
    }
    case RT_PAT_BLEND: {
      if constexpr (DEPTH > 0) {
        double r1, g1, b1, r2, g2, b2;
        pattern_at<DEPTH - 1>(S, pt.first, p, r1, g1, b1);
        pattern_at<DEPTH - 1>(S, pt.second, p, r2, g2, b2);
        r = (r1 + r2) / 2.0; g = (g1 + g2) / 2.0; b = (b1 + b2) / 2.0;  // src/color.rs:169-175
      } else {
        r = g = b = 0.0;
      }
      return;
    }
    default: r = g = b = 0.0; return;
  }
  r = primary ? pt.a[0] : pt.b[0];
  g = primary ? pt.a[1] : pt.b[1];
  b = primary ? pt.a[2] : pt.b[2];
}
constexpr int MAX_BLEND_DEPTH = 4;

// local_normal_at (src/shape.rs:805-829, per-type routines 880-1173, 1263-1332)
DEV V3 local_normal(const SceneDev& S, const LeafShade& L, V3 at, double u, double v) {
  switch (L.type) {
    case RT_PLANE: return v3(L.plane_normal[0], L.plane_normal[1], L.plane_normal[2]);
    case RT_CUBE: {
      double xa = fabs(at.x), ya = fabs(at.y), za = fabs(at.z);
      double max_component = fmax(fmax(xa, ya), za);
      if (max_component == xa) return v3(at.x, 0.0, 0.0);
      if (max_component == ya) return v3(0.0, at.y, 0.0);
      return v3(0.0, 0.0, at.z);
    }
    case RT_CYLINDER:
    case RT_CONE: {
      double dist = at.x * at.x + at.z * at.z;
      if (dist < 1.0 && at.y >= L.maximum - FEQ_EPSILON) return v3(0.0, 1.0, 0.0);
      if (dist < 1.0 && at.y <= L.minimum + FEQ_EPSILON) return v3(0.0, -1.0, 0.0);
      if (L.type == RT_CYLINDER) return v3(at.x, 0.0, at.z);
      double y = sqrt(dist);
      if (at.y > 0.0) y = -y;
      return v3(at.x, y, at.z);
    }
    case RT_TRIANGLE: {
      const TriShade& ts = S.tri_shade[L.tri];
      return v3(ts.normal[0], ts.normal[1], ts.normal[2]);
    }
    case RT_SMOOTH_TRIANGLE: {
      const TriShade& ts = S.tri_shade[L.tri];
      double w = 1.0 - u - v;  // n2 * u + n3 * v + n1 * (1 - u - v)
      return v3(ts.n2[0] * u + ts.n3[0] * v + ts.n1[0] * w, ts.n2[1] * u + ts.n3[1] * v + ts.n1[1] * w,
                ts.n2[2] * u + ts.n3[2] * v + ts.n1[2] * w);
    }
    default: return at;  // sphere, empty: the point with w = 0
  }
}

// Appends `mine` items per lane to a queue: one atomic per wave, prefix by ballot.
DEV uint32_t wave_append(uint32_t* counter, bool mine) {
  unsigned long long mask = __ballot(mine);
  uint32_t lane = thread_in_block() & 63u;
  uint32_t total = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  if (lane == 0 && total) base = atomicAdd(counter, total);
  base = __shfl(base, 0);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// The rays a generation spawns are queued in QUEUE_SEGMENTS segments, each with its own counter and
// its own stretch of `segcap` record slots (segment s holds slots [s * segcap, (s + 1) * segcap)): a
// wave appends to segment (wave number % QUEUE_SEGMENTS).  One counter for all waves serialises the
// returning atomics at one L2 address, about 10 ns each, which for a generation of half a million rays
// (15 000 atomics) is more than the tracing itself takes.
//
// The next generation runs one thread per queued ray: thread i finds its segment and record slot from
// the running sum of the segment counters.  Every lane of the wave must call this (it shuffles).
DEV bool queue_slot(const uint32_t* counts, uint32_t segcap, uint32_t i, uint32_t& slot) {
  static_assert(QUEUE_SEGMENTS == 64, "one segment counter per lane of a wave");
  const uint32_t lane = thread_in_block() & 63u;
  const uint32_t c = min(counts[lane], segcap);
  uint32_t p = c;  // inclusive running sum over the segments
  for (uint32_t d = 1; d < 64; d <<= 1) {
    uint32_t v = __shfl_up(p, d, 64);
    if (lane >= d) p += v;
  }
  uint32_t seg = 0;  // the number of segments that end at or before ray i, at most 63
  for (uint32_t step = 32; step > 0; step >>= 1) {
    uint32_t v = __shfl(p, (int)(seg + step - 1), 64);
    if (v <= i) seg += step;
  }
  const uint32_t end = __shfl(p, (int)seg, 64), len = __shfl(c, (int)seg, 64);
  slot = seg * segcap + (i - (end - len));
  return i < end;
}

// reflect_dot_eye.powf(shininess) (src/light.rs:168).  A whole-number exponent up to 1024 (the
// default material's is 200) is raised by repeated squaring: about ten multiplications, each
// correctly rounded, so the result stays within a few ulps of the libm value; anything else goes
// through pow().
DEV double specular_power(double x, double shininess) {
  if (shininess >= 0.0 && shininess <= 1024.0 && shininess == floor(shininess)) {
    uint32_t n = (uint32_t)shininess;
    double result = 1.0, base = x;
    while (n) {
      if (n & 1u) result *= base;
      base *= base;
      n >>= 1;
    }
    return result;
  }
  return pow(x, shininess);
}

// Canvas::save quantisation (src/canvas.rs:60-62): ceil(clamp(c * 255, 0, 255)); NaN -> 0
DEV uint8_t quantise(double c) {
  double v = c * 255.0;
  if (v < 0.0) v = 0.0;
  else if (v > 255.0) v = 255.0;
  v = ceil(v);
  return (v == v) ? (uint8_t)v : (uint8_t)0;
}

// Canvas::write_pixel for the 8x8 pixel block of a wave; all 64 lanes call it together.  The bytes
// of one block row (8 pixels, 24 bytes) are gathered across lanes and stored as six dwords, so the
// framebuffer leaves the chip as full dwords instead of three single-byte stores per pixel.  In the
// compact layout the 64 pixels of the wave are contiguous (192 bytes, 48 dwords).
DEV void write_pixel(const ImageOut& img, uint32_t hsize, uint32_t x, uint32_t y, uint32_t i, bool active, double r,
                     double g, double b, int leaf) {
  const size_t p = (size_t)y * hsize + x;
  if (active) {
    if (img.rgb64) {
      img.rgb64[p * 3 + 0] = r; img.rgb64[p * 3 + 1] = g; img.rgb64[p * 3 + 2] = b;
    }
    if (img.hit_leaf) img.hit_leaf[p] = leaf;
  }
  if (!img.rgb8) return;
  const uint32_t v = (uint32_t)quantise(r) | ((uint32_t)quantise(g) << 8) | ((uint32_t)quantise(b) << 16);
  const uint32_t lane = thread_in_block() & 63u;
  if (img.compact) {
    // dword `lane` (0..47) of the wave's 192 bytes holds bytes 4 lane .. 4 lane + 3: pixels p0 and p0 + 1
    const uint32_t p0 = (4u * lane) / 3u, shift = 8u * (4u * lane - 3u * p0);
    const uint32_t lo = __shfl(v, (int)(p0 < 63u ? p0 : 63u));
    const uint32_t hi = __shfl(v, (int)(p0 + 1u < 63u ? p0 + 1u : 63u));
    if (((uintptr_t)img.rgb8 & 3u) == 0u) {  // wave-uniform; i - lane is a multiple of 64
      if (lane < 48u) {
        const unsigned long long cat = (unsigned long long)lo | ((unsigned long long)hi << 24);
        reinterpret_cast<uint32_t*>(img.rgb8 + (size_t)(i - lane) * 3u)[lane] = (uint32_t)(cat >> shift);
      }
    } else {
      img.rgb8[(size_t)i * 3 + 0] = (uint8_t)v; img.rgb8[(size_t)i * 3 + 1] = (uint8_t)(v >> 8);
      img.rgb8[(size_t)i * 3 + 2] = (uint8_t)(v >> 16);
    }
    return;
  }
  const uint32_t col = lane & 7u, row_base = lane & ~7u;
  // dword `col` (0..5) of the row holds bytes 4 col .. 4 col + 3, which belong to pixels p0 and p0 + 1
  const uint32_t p0 = (4u * col) / 3u, shift = 8u * (4u * col - 3u * p0);
  const uint32_t lo = __shfl(v, (int)(row_base + (p0 < 7u ? p0 : 7u)));
  const uint32_t hi = __shfl(v, (int)(row_base + (p0 + 1u < 7u ? p0 + 1u : 7u)));
  const uint32_t x0 = x - col;
  const bool whole_rows = (hsize & 3u) == 0u && x0 + 8u <= hsize && ((uintptr_t)img.rgb8 & 3u) == 0u;  // wave-uniform
  if (whole_rows) {
    if (active && col < 6u) {
      const unsigned long long cat = (unsigned long long)lo | ((unsigned long long)hi << 24);
      uint32_t* dst = reinterpret_cast<uint32_t*>(img.rgb8 + ((size_t)y * hsize + x0) * 3u) + col;
      *dst = (uint32_t)(cat >> shift);
    }
  } else if (active) {
    img.rgb8[p * 3 + 0] = (uint8_t)v; img.rgb8[p * 3 + 1] = (uint8_t)(v >> 8); img.rgb8[p * 3 + 2] = (uint8_t)(v >> 16);
  }
}

}  // namespace rt
