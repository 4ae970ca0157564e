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

// g and dg/ds in terms of s = sin(th), th in (-pi/2, pi/2): no trigonometry
__device__ __forceinline__ double contact_gs(double s, double p, double e, double cw, double sw, double ci, double L,
                                             double* dg) {
  const double c = sqrt(fmax(1.0 - s * s, 0.0));
  const double D = 1.0 + e * (sw * c - cw * s);
  const double rho = p / D;
  const double q = ci * ci + (1.0 - ci * ci) * s * s;
  // dD/ds = -e (sw s / c + cw);  d(rho^2 q)/ds = rho^2 (q' - 2 q D' / D)
  const double dD = -e * (sw * s / fmax(c, 1e-300) + cw);
  *dg = rho * rho * (2.0 * (1.0 - ci * ci) * s - 2.0 * q * dD / D);
  return rho * rho * q - L * L;
}

// mean anomaly of the point th = omega + f - pi/2 given s = sin(th): E from (sin E, cos E), on the revolution of
// f = th + pi/2 - omega (E and f never differ by half a turn) -- the branch the bracketing path below returns
__device__ __forceinline__ double contact_mean_anomaly(double s, double e, double cw, double sw) {
  const double c = sqrt(fmax(1.0 - s * s, 0.0));
  const double cosf = sw * c - cw * s, sinf = sw * s + cw * c;      // f = th + (pi/2 - omega)
  const double den = 1.0 + e * cosf;
  const double cosE = (e + cosf) / den, sinE = sqrt(1.0 - e * e) * sinf / den;
  const double f = asin(s) + kHalfPi - atan2(sw, cw);
  double E = atan2(sinE, cosE);
  E = fma(rint((f - E) * (1.0 / kTwoPiHi)), kTwoPiHi, E);
  return E - e * sinE;
}

// returns true on failure (no contact: the caller evaluates every cadence, keplerian.py:771-775)
__device__ inline bool contact_solve(double a, double e, double cw, double sw, double ci, double L,
                                     double* M_left, double* M_right) {
  const double p = a * (1.0 - e * e);
  double out[2] = {0.0, 0.0};
  bool bad = !(contact_g(0.0, p, e, cw, sw, ci, L) < 0.0);
  for (int side = 0; side < 2 && !bad; ++side) {
    const double sgn = side == 0 ? -1.0 : 1.0;
    // Fast path (this runs on ONE lane per (draw, planet) inside the packing kernel and its latency is the kernel's:
    // 28 us at C3, 51 us with an occultation, of which ~6 are the packing): the contact in terms of s = sin(th) -- no
    // trigonometry -- from the fixed point of "the star-planet distance is what it is at the current estimate" (two
    // rounds: the window kernel's refinement) polished by Newton steps with the analytic slope.  Accepted only if it
    // converged (see below); anything else -- grazing chords, a distance that changes
    // fast across the window -- takes the bracketing solver below, as before.
    {
      const double si2 = 1.0 - ci * ci;
      double s = 0.0, dg;
      bool ok = si2 > 1e-12;
      for (int it = 0; it < 2 && ok; ++it) {
        const double c = sqrt(fmax(1.0 - s * s, 0.0));
        const double rho = p / (1.0 + e * (sw * c - cw * s));
        const double S = (L * L / (rho * rho) - ci * ci) / si2;
        ok = S > 0.0 && S < 0.98;
        s = sgn * sqrt(fmax(S, 0.0));
      }
      double step = 1.0;
      for (int it = 0; it < 8 && ok; ++it) {
        const double g = contact_gs(s, p, e, cw, sw, ci, L, &dg);
        step = g / dg;
        s -= step;
        ok = (s * sgn > 0.0) && (fabs(s) < 0.99) && (step == step);
        if (fabs(step) <= 1e-16 * fabs(s)) break;
      }
      // (the nearest root, not a later one: no overlap lost at a quarter, a half, three quarters of the way; and only
      // windows within 30 degrees of the conjunction -- 200 000 random geometries, e up to 0.95, a / R* from 2: 91 %
      // take this path, all of them within 1e-15 rad of the bracketing solver's root)
      ok = ok && fabs(step) <= 4e-16 * fabs(s) && fabs(s) <= 0.5;
      for (int k = 1; k <= 3 && ok; ++k) ok = contact_gs(0.25 * k * s, p, e, cw, sw, ci, L, &dg) < 0.0;
      if (ok) {
        out[side] = contact_mean_anomaly(s, e, cw, sw);
        continue;
      }
    }
    double lo = 0.0, hi = 0.0;
    bool found = false;
    for (int k = 1; k <= 32; ++k) {
      const double th = sgn * k * (kHalfPi / 32.0);
      if (contact_g(th, p, e, cw, sw, ci, L) > 0.0) { hi = th; found = true; break; }
      lo = th;
    }
    if (!found) { bad = true; break; }
    // bracketed false position with the Illinois correction (superlinear: ~10 evaluations of the trigonometry where
    // plain bisection to the last bit takes 53), bisection whenever the secant point leaves the bracket
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
