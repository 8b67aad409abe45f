// The cell test of the cone direction index, shared by the host builder (band_grid.cpp) and the
// device builder (band_grid_device.hip): both must accept exactly the same cells, so both compile
// this one definition (without multiply-add contraction).
#pragma once
#include <math.h>

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

}  // namespace rt
