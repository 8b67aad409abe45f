// The cell test of the cone direction index, shared by the host builder (band_grid.cpp) and the
// device builder (band_grid_device.hip): both must accept exactly the same cells, so both compile
// this one definition (without multiply-add contraction).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "rt_types.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

namespace rt {

// On the face of major axis m a direction is (1 at m, u at axis a, v at axis b), u, v in [-1, 1], and
// a(d) / d_m^2 = g(u, v) = c0 + cu u + cv v + cuu u^2 + cuv u v + cvv v^2.
struct FacePoly {
  double c0, cu, cv, cuu, cuv, cvv;
};

RT_HD FacePoly face_poly(const ConeBand& b, int m) {
  // b.q = coefficients of (xx, yy, zz, xy, xz, yz)
  const int a = (m + 1) % 3, c = (m + 2) % 3;
  // coefficient of d_i d_j, i != j: (0,1) -> q[3], (0,2) -> q[4], (1,2) -> q[5]
  const double cross_ma = b.q[2 + m + a], cross_mc = b.q[2 + m + c], cross_ac = b.q[2 + a + c];
  return FacePoly{b.q[m], cross_ma, cross_mc, b.q[a], cross_ac, b.q[c]};
}

// |a(d)| < FEQ_EPSILON (1 + 1e-6) + slack |d|^2 is the device pre-filter for |d|^2 >= 1 - 1e-6
RT_HD double band_threshold(const ConeBand& b) { return (FEQ_EPSILON * (1.0 + 1e-6) + b.slack) * (1.0 + 4e-6); }

// Can |g| drop below the band threshold somewhere inside cell (iu, iv) of a level_n x level_n face?
// Centred form of the quadratic: an exact enclosure of the linear part, the second-order part
// bounded by its coefficients.
RT_HD bool band_cell_passes(const FacePoly& p, double thr, int level_n, int iu, int iv) {
  const double size = 2.0 / level_n;
  const double h = 0.5 * size * (1.0 + 1e-9) + 2e-6;  // the lookup computes u, v in f32
  const double uc = -1.0 + (iu + 0.5) * size, vc = -1.0 + (iv + 0.5) * size;
  const double g = p.c0 + p.cu * uc + p.cv * vc + p.cuu * uc * uc + p.cuv * uc * vc + p.cvv * vc * vc;
  const double gu = p.cu + 2.0 * p.cuu * uc + p.cuv * vc, gv = p.cv + 2.0 * p.cvv * vc + p.cuv * uc;
  const double reach = (fabs(gu) + fabs(gv)) * h + (fabs(p.cuu) + fabs(p.cuv) + fabs(p.cvv)) * h * h;
  const double um = fabs(uc) + h, vm = fabs(vc) + h;
  return !(fabs(g) > reach + thr * (1.0 + um * um + vm * vm));
}

// ---------------------------------------------------------------------------------------------
// Strip word of one (band, cell) entry.  Inside a cell the band is close to a straight strip: with
// cell-local coordinates s, t in [-1, 1] (u = uc + s / n, v = vc + t / n) and the expansion of g
// about the cell centre,  g(u, v) = g + gu du + gv dv + second-order part,  a direction of the cell
// can lie in the band only if  |g + gu du + gv dv| <= thr (1 + um^2 + vm^2) + (|cuu| + |cuv| + |cvv|) h^2.
// Divided by the larger of |gu|, |gv| (times the half cell size) that reads |g0 + a + m b| <= w with a
// the coordinate of the larger coefficient, b the other one and |m| <= 1.  The word keeps g0 (12 bits,
// steps of 1/512), m (11 bits, steps of 1/1024), w rounded up to an 8-bit float and which coordinate a
// is; w also takes up the rounding of g0 and m and what the ray's single-precision s, t can be off by.
// A ray tests the word of every entry of its cell (one multiply-add and a compare) and only looks up
// the bands whose strip it lies in: a few per cent of the entries.
//   bits 31..20 g0, 19..9 m, 8..1 width code, 0: 1 when a = t.
constexpr uint32_t BAND_STRIP_ALWAYS = 0xFFu;  // width code: every direction of the cell passes (a width of 31)

// Width code c < 255 stands for 2^(c / 16 - 11) (1 + (c % 16) / 16): the f32 with bit pattern
// (c << 19) + (116 << 23).
RT_HD uint32_t band_strip_width_code(double w) {  // the smallest code whose width is >= w
  if (!(w > 0.00048828125)) return 0u;            // 2^-11
  uint32_t e = 0;
  double base = 0.00048828125;
  while (e < 15u && w >= base * 2.0) {
    base *= 2.0;
    e++;
  }
  if (w >= base * 2.0) return BAND_STRIP_ALWAYS;
  const double f = (w / base - 1.0) * 16.0;  // in [0, 16)
  uint32_t mant = (uint32_t)f;
  if ((double)mant < f) mant++;
  uint32_t code = (e << 4) + mant;  // mant == 16 carries into the exponent
  return code >= BAND_STRIP_ALWAYS ? BAND_STRIP_ALWAYS : code;
}

// What a ray's s, t (computed in single precision from the direction rounded to f32, band_strip_coords)
// can differ by from the exact local coordinates, for an n x n face.
RT_HD double band_strip_coord_error(int n) { return 1.0e-3 * ((double)n / 1024.0) + 1e-5; }

RT_HD uint32_t band_strip_word(const FacePoly& p, double thr, int n, int iu, int iv) {
  const double size = 2.0 / n;
  const double h = 0.5 * size * (1.0 + 1e-9) + 2e-6;  // as in band_cell_passes: the cell the lookup may pick
  const double hn = 0.5 * size;
  const double uc = -1.0 + (iu + 0.5) * size, vc = -1.0 + (iv + 0.5) * size;
  const double g = p.c0 + p.cu * uc + p.cv * vc + p.cuu * uc * uc + p.cuv * uc * vc + p.cvv * vc * vc;
  const double gu = p.cu + 2.0 * p.cuu * uc + p.cuv * vc, gv = p.cv + 2.0 * p.cvv * vc + p.cuv * uc;
  const double um = fabs(uc) + h, vm = fabs(vc) + h;
  const double w = thr * (1.0 + um * um + vm * vm) + (fabs(p.cuu) + fabs(p.cuv) + fabs(p.cvv)) * h * h;
  const double du = gu * hn, dv = gv * hn;
  const uint32_t which = fabs(dv) > fabs(du) ? 1u : 0u;
  const double major = which ? dv : du, minor = which ? du : dv;
  const uint32_t always = BAND_STRIP_ALWAYS << 1;
  if (!(fabs(major) > 1e-300)) return always;
  const double g0 = g / major, m = minor / major, wn = w / fabs(major);
  if (!(wn < 3.0) || !(fabs(g0) < 3.9)) return always;
  double q0 = floor(g0 * 512.0 + 0.5), qm = floor(m * 1024.0 + 0.5);
  q0 = q0 < -2048.0 ? -2048.0 : (q0 > 2047.0 ? 2047.0 : q0);
  qm = qm < -1024.0 ? -1024.0 : (qm > 1023.0 ? 1023.0 : qm);
  const double e0 = fabs(g0 - q0 / 512.0), em = fabs(m - qm / 1024.0);
  const double reach = h / hn;  // how far s, t of a direction of the (padded) cell can go
  const double delta = band_strip_coord_error(n);
  const double slack = e0 + em * reach + delta * (1.0 + fabs(qm / 1024.0)) + 1e-5;
  const uint32_t code = band_strip_width_code(wn + slack);
  if (code >= BAND_STRIP_ALWAYS) return always;
  return (((uint32_t)(int)q0 & 0xFFFu) << 20) | (((uint32_t)(int)qm & 0x7FFu) << 9) | (code << 1) | which;
}

// The ray's side.  (x, y): the direction's face coordinates in cell units, in [0, n]; the cell index and
// the local coordinates come from the same rounded values.
RT_HD void band_strip_coords(float x, float y, int n, int& iu, int& iv, float& s, float& t) {
  iu = (int)floorf(x);
  iv = (int)floorf(y);
  iu = iu < 0 ? 0 : (iu >= n ? n - 1 : iu);
  iv = iv < 0 ? 0 : (iv >= n ? n - 1 : iv);
  s = 2.0f * (x - (float)iu) - 1.0f;
  t = 2.0f * (y - (float)iv) - 1.0f;
}
// s1024, t1024: the local coordinates times 1024 (the unit of m's steps; g0's steps are two of them).
// Width code 255 needs no case of its own: its bit pattern is a width of 31, above any |g0 + a + m b|.
RT_HD bool band_strip_passes(uint32_t word, float s, float t, float s1024, float t1024) {
  const float q0 = (float)((int32_t)word >> 20), qm = (float)((int32_t)(word << 12) >> 21);
  const uint32_t wbits = (((word >> 1) & 0xFFu) << 19) + (126u << 23);  // the width times 1024
  float w;
  memcpy(&w, &wbits, 4);
  const bool t_major = (word & 1u) != 0u;
  const float a = t_major ? t1024 : s1024, b = t_major ? s : t;
  return fabsf((q0 * 2.0f + a) + qm * b) <= w;
}

}  // namespace rt
