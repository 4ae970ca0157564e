// Test harness ONLY: compiles the device math header for the host (g++) so the
// arithmetic can be checked against the oracle on a machine without a GPU.
// Not part of the product; nothing in exoplanet_amd/ loads this.
#define EXO_HOST_BUILD 1
#include "../exoplanet_amd/csrc/exo_math.hpp"
#include <stdint.h>

extern "C" {
void harness_quad_sv(const double* b, const double* r, double* s, double* dsdb, double* dsdr, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    exo::SV o;
    const double bb = fabs(b[i]);
    exo::quad_sv<true>(bb, r[i], o);
    const double sg = b[i] < 0 ? -1.0 : 1.0;
    s[3 * i] = o.s0; s[3 * i + 1] = o.s1; s[3 * i + 2] = o.s2;
    dsdb[3 * i] = sg * o.db0; dsdb[3 * i + 1] = sg * o.db1; dsdb[3 * i + 2] = sg * o.db2;
    dsdr[3 * i] = o.dr0; dsdr[3 * i + 1] = o.dr1; dsdr[3 * i + 2] = o.dr2;
  }
}
void harness_quad_sv_nograd(const double* b, const double* r, double* s, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    exo::SV o;
    exo::quad_sv<false>(fabs(b[i]), r[i], o);
    s[3 * i] = o.s0; s[3 * i + 1] = o.s1; s[3 * i + 2] = o.s2;
  }
}
void harness_kepler(const double* M, const double* e, double* sinf, double* cosf, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const double ee = e[i];
    if (!(ee >= 0.0 && ee < 1.0)) { sinf[i] = cosf[i] = NAN; continue; }
    exo::KeplerHalf h = exo::kepler_half(M[i], ee, sqrt(1.0 - ee), sqrt(1.0 + ee));
    const double den = h.X * h.X + h.Y * h.Y;
    cosf[i] = (h.X * h.X - h.Y * h.Y) / den;
    sinf[i] = 2.0 * h.X * h.Y / den;
  }
}
}
