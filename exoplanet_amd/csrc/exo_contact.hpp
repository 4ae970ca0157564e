// exo_contact.hpp -- first / fourth contact of an eccentric orbit (device only).
//
// Replaces exoplanet_core's contact_points Op (reference call site
// /root/reference/src/exoplanet/orbits/keplerian.py:744-753): the two roots
// nearest mid-transit of  rho(f)^2 (1 - sin^2 i sin^2(omega+f)) = L^2,
// L = R_star + r, returned as mean anomalies.  Coarse outward scan for the
// bracket, then bracketed false position on the definition: O(planets) work per draw.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "exo_math.hpp"

namespace exo {

__device__ __forceinline__ double contact_g(double th, double p, double e, double cw, double sw,
                                            double ci, double L) {
  double st, ct;
  exo::sincos_any(th, &st, &ct);          // |th| <= pi / 2: the branch-free kernel, a third of libm's instructions
  const double cosf = sw * ct - cw * st;  // th = omega + f - pi/2
  const double rho = p / (1.0 + e * cosf);
  return rho * rho * (st * st + ci * ci * ct * ct) - L * L;
}

// returns true on failure (no contact: the caller evaluates every cadence, keplerian.py:771-775)
__device__ inline bool contact_solve(double a, double e, double cw, double sw, double ci, double L,
                                     double* M_left, double* M_right) {
  const double p = a * (1.0 - e * e);
  double out[2] = {0.0, 0.0};
  bool bad = !(contact_g(0.0, p, e, cw, sw, ci, L) < 0.0);
  for (int side = 0; side < 2 && !bad; ++side) {
    const double sgn = side == 0 ? -1.0 : 1.0;
    double lo = 0.0, hi = 0.0;
    bool found = false;
    for (int k = 1; k <= 32; ++k) {
      const double th = sgn * k * (kHalfPi / 32.0);
      if (contact_g(th, p, e, cw, sw, ci, L) > 0.0) { hi = th; found = true; break; }
      lo = th;
    }
    if (!found) { bad = true; break; }
    // bracketed false position with the Illinois correction (superlinear: ~10 evaluations of the trigonometry where
    // plain bisection to the last bit takes 53 -- this runs on ONE lane per (draw, planet) inside the packing kernel
    // and its latency is the kernel's), bisection whenever the secant point leaves the bracket
    double flo = contact_g(lo, p, e, cw, sw, ci, L), fhi = contact_g(hi, p, e, cw, sw, ci, L);
    int last = 0;
    for (int it = 0; it < 64; ++it) {
      if (!(fabs(hi - lo) > 4.0e-16 * fmax(1.0, fabs(hi)))) break;
      double mid = (lo * fhi - hi * flo) / (fhi - flo);
      const double a_lo = fmin(lo, hi), a_hi = fmax(lo, hi);
      if (!(mid > a_lo && mid < a_hi)) mid = 0.5 * (lo + hi);
      const double fm = contact_g(mid, p, e, cw, sw, ci, L);
      if (fm > 0.0) {
        hi = mid; fhi = fm;
        if (last == 1) flo *= 0.5;
        last = 1;
      } else {
        lo = mid; flo = fm;
        if (last == -1) fhi *= 0.5;
        last = -1;
        if (fm == 0.0) { hi = mid; break; }
      }
    }
    const double th = 0.5 * (lo + hi);
    const double f = th + kHalfPi - atan2(sw, cw);
    double shf, chf;
    sincos(0.5 * f, &shf, &chf);
    const double E = 2.0 * atan2(sqrt(1.0 - e) * shf, sqrt(1.0 + e) * chf);
    out[side] = E - e * sin(E);
  }
  *M_left = bad ? 0.0 : out[0];
  *M_right = bad ? 0.0 : out[1];
  return bad;
}

}  // namespace exo
