// exo_pack_core.hpp -- the record-packing algebra (KeplerianOrbit.__init__ + get_cl + windows) as device functions, shared by
// the packing kernels (exo_pack.hip) and the light-curve sweep, which packs the records ITSELF when it is handed the
// constructor's columns (exo_transit.hip: transit_enum_kernel<.., PACK>, and the packing VJP folded into the sweep's last
// kernel).  Reference lines: exo_pack.hip's header comment.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"
#include "exo_contact.hpp"

namespace exo_pack {

constexpr double kG = 2942.2062175044193;  // R_sun^3 / M_sun / day^2 (orbits/constants.py:32)
constexpr double kPi = 3.14159265358979323846;
constexpr double kCLight = 37231.66360672704;  // R_sun / day (orbits/constants.py:36)

struct Derived {
  double a, n, cw, sw, E0, M0, f, cosi, x, y, mtot;
};

__device__ __forceinline__ Derived derive(const double* in, bool circular) {
  Derived d;
  const double P = in[EXO_IN_PERIOD], e = circular ? 0.0 : in[EXO_IN_ECC];
  d.mtot = in[EXO_IN_MSTAR] + in[EXO_IN_MPLANET];
  d.a = cbrt(kG * d.mtot * P * P * (1.0 / (4.0 * kPi * kPi)));
  d.n = 2.0 * kPi / P;
  if (circular) {
    d.cw = 1.0; d.sw = 0.0; d.E0 = 0.5 * kPi; d.M0 = 0.5 * kPi; d.f = 1.0; d.x = 1.0; d.y = 1.0;
  } else {
    sincos(in[EXO_IN_OMEGA], &d.sw, &d.cw);
    d.y = sqrt(1.0 - e) * d.cw;
    d.x = sqrt(1.0 + e) * (1.0 + d.sw);
    d.E0 = 2.0 * atan2(d.y, d.x);
    d.M0 = d.E0 - e * sin(d.E0);
    d.f = (1.0 + e * d.sw) / (1.0 - e * e);
  }
  d.cosi = d.f * in[EXO_IN_RSTAR] / d.a * in[EXO_IN_B];
  return d;
}

// Where the inputs come from / the input cotangents go.  Packed: the (n_draw, n_planet, EXO_NIN) and (n_draw, 2|4)
// arrays of exo_pack_records_f64.  Cols: every input its own device array read with per-column strides (0 =
// broadcast) or absent (a default value) -- a caller whose parameters are separate tensors needs no packing pass,
// and no unpacking pass for the cotangents, which are written densely, one (n_draw, n_planet) array per column.
struct PackedSrc {
  const double* orbit_in;
  const double* ld_in;
  __device__ __forceinline__ void load(int64_t i, int64_t, int, double* in) const {
#pragma unroll
    for (int k = 0; k < EXO_NIN; ++k) in[k] = orbit_in[i * EXO_NIN + k];
  }
  __device__ __forceinline__ double ld(int64_t draw, int k, int nld) const { return ld_in[draw * nld + k]; }
};
struct PackedDst {
  double* gorbit_in;
  double* gld_in;
  __device__ __forceinline__ void store(int64_t i, int k, double v) const { gorbit_in[i * EXO_NIN + k] = v; }
  __device__ __forceinline__ void store_ld(int64_t draw, int k, int nld, double v) const { gld_in[draw * nld + k] = v; }
};
struct ColsSrc {
  const double* ptr[EXO_NIN];
  int64_t ds[EXO_NIN], ps[EXO_NIN];
  double def[EXO_NIN];
  const double* ldp[4];
  int64_t lds[4];
  __device__ __forceinline__ void load(int64_t, int64_t draw, int planet, double* in) const {
#pragma unroll
    for (int k = 0; k < EXO_NIN; ++k) in[k] = ptr[k] ? ptr[k][draw * ds[k] + planet * ps[k]] : def[k];
  }
  __device__ __forceinline__ double ld(int64_t draw, int k, int) const { return ldp[k][draw * lds[k]]; }
};
struct ColsDst {
  double* ptr[EXO_NIN];
  double* ldp[4];
  __device__ __forceinline__ void store(int64_t i, int k, double v) const { if (ptr[k]) ptr[k][i] = v; }
  __device__ __forceinline__ void store_ld(int64_t draw, int k, int, double v) const { if (ldp[k]) ldp[k][draw] = v; }
};

// the record of (draw, planet) -- i = draw * n_planet + planet -- into o[EXO_NPAR]
template <class Src>
__device__ __forceinline__ void pack_record(const Src& src, int64_t i, int64_t draw, int planet, uint32_t flags, double* o) {
  const bool circular = flags & EXO_PACK_CIRCULAR, secondary = flags & EXO_FLAG_SECONDARY;
  const bool window = flags & EXO_FLAG_WINDOW;
  const double inf = __builtin_inf();
    double in[EXO_NIN];
    src.load(i, draw, planet, in);
    const Derived d = derive(in, circular);
    const double e = circular ? 0.0 : in[EXO_IN_ECC];
    const double P = in[EXO_IN_PERIOD], Rs = in[EXO_IN_RSTAR], r = in[EXO_IN_R];
    // sin(acos(cos i)); |cos i| > 1 (b beyond the orbit's largest impact parameter): sin i = 0, i.e.
    // never in front of the star and flux 0, which is what the reference's `switch(los > 0, lc, 0)` gives
    // for the NaN it computes there (limb_dark.py:252)
    const double sini = sqrt(fmax(0.0, 1.0 - d.cosi * d.cosi));
    o[EXO_P_N] = d.n;
    o[EXO_P_TP] = in[EXO_IN_T0] - d.M0 / d.n;
    o[EXO_P_ECC] = e;
    o[EXO_P_COSW] = d.cw;
    o[EXO_P_SINW] = d.sw;
    o[EXO_P_COSI] = d.cosi;
    o[EXO_P_SINI] = sini;
    o[EXO_P_AOR] = d.a / Rs;
    o[EXO_P_ROR] = r / Rs;
    o[EXO_P_T0] = in[EXO_IN_T0];
    o[EXO_P_PERIOD] = P;
    const double k = r / Rs;
    o[EXO_P_FRATIO] = secondary ? in[EXO_IN_SBR] * k * k : 0.0;
    double ts = -inf, te = inf, ts2 = -inf, te2 = inf;
    if (window) {
      const double hp = 0.5 * P;
      if (circular) {
        // Winn (2010) eq. 14, keplerian.py:733-741
        const double arg = (1.0 + k) * (1.0 + k) - in[EXO_IN_B] * in[EXO_IN_B];
        const double hdur = hp * asin(Rs / (d.a * sini) * sqrt(arg)) * (1.0 / kPi);
        ts = -hdur; te = hdur;
        if (secondary) { ts2 = hp - hdur; te2 = hp + hdur; }  // flipped orbit: half a period later
      } else {
        double ml, mr;
        if (!exo::contact_solve(d.a, e, d.cw, d.sw, d.cosi, Rs + r, &ml, &mr)) {
          double a0 = (ml - d.M0) / d.n + hp, a1 = (mr - d.M0) / d.n + hp;
          a0 = a0 - P * floor(a0 / P) - hp;
          a1 = a1 - P * floor(a1 / P) - hp;
          ts = a0 > 0.0 ? a0 - P : a0;
          te = a1 < 0.0 ? a1 + P : a1;
        }
        if (secondary) {
          // occultation = transit of the flipped orbit (omega - pi): same ellipse, same t_periastron
          const double y2 = -sqrt(1.0 - e) * d.cw, x2 = sqrt(1.0 + e) * (1.0 - d.sw);
          const double E02 = 2.0 * atan2(y2, x2);
          const double M02 = E02 - e * sin(E02);
          if (!exo::contact_solve(d.a, e, -d.cw, -d.sw, d.cosi, Rs + r, &ml, &mr)) {
            double a0 = (ml - M02) / d.n + hp, a1 = (mr - M02) / d.n + hp;
            a0 = a0 - P * floor(a0 / P) - hp;
            a1 = a1 - P * floor(a1 / P) - hp;
            const double s0 = a0 > 0.0 ? a0 - P : a0, s1 = a1 < 0.0 ? a1 + P : a1;
            double shift = (M02 - d.M0) / d.n;          // t0(flipped) - t0
            shift = shift - P * floor(shift / P);       // [0, P)
            const double lo = shift + s0, hi = shift + s1;
            if (lo >= 0.0 && hi <= P) { ts2 = lo; te2 = hi; }
          }
        }
      }
    }
    o[EXO_P_TS] = ts; o[EXO_P_TE] = te; o[EXO_P_TS2] = ts2; o[EXO_P_TE2] = te2;
    o[EXO_P_CLIGHT] = kCLight / Rs;   // read only by light-delay sweeps
    o[EXO_P_CLIGHT + 1] = o[EXO_P_CLIGHT + 2] = o[EXO_P_CLIGHT + 3] = 0.0;
}
// limb darkening of one draw: u -> c (get_cl) into ld_row[3 | 6]
template <class Src>
__device__ __forceinline__ void pack_ld(const Src& src, int64_t draw, uint32_t flags, double* ld_row) {
  const bool secondary = flags & EXO_FLAG_SECONDARY;
    const int nset = secondary ? 2 : 1;
    for (int s = 0; s < nset; ++s) {
      const double u1 = src.ld(draw, 2 * s, 2 * nset), u2 = src.ld(draw, 2 * s + 1, 2 * nset);
      const double c0 = 1.0 - u1 - 1.5 * u2, c1 = u1 + 2.0 * u2, c2 = -0.25 * u2;
      const double inorm = 1.0 / (kPi * (c0 + c1 * (1.0 / 1.5)));
      double* o = ld_row + 3 * s;
      o[0] = c0 * inorm; o[1] = c1 * inorm; o[2] = c2 * inorm;
    }
  }

// reverse of pack_record: grec[EXO_NPAR] = the record's cotangents, multiplied by sc as they are read; the input cotangents
// go to dst
template <class Src, class Dst>
__device__ __forceinline__ void pack_vjp_record(const Src& src, int64_t i, int64_t draw, int planet, uint32_t flags,
                                                const double* __restrict__ grec, double sc, const Dst& dst) {
  const bool circular = flags & EXO_PACK_CIRCULAR, secondary = flags & EXO_FLAG_SECONDARY;
    double in[EXO_NIN], g[EXO_NPAR], o[EXO_NIN];
    src.load(i, draw, planet, in);
#pragma unroll
    for (int k = 0; k < EXO_NPAR; ++k) g[k] = sc * grec[k];
    const Derived d = derive(in, circular);
    const double e = circular ? 0.0 : in[EXO_IN_ECC];
    const double P = in[EXO_IN_PERIOD], Rs = in[EXO_IN_RSTAR], r = in[EXO_IN_R], b = in[EXO_IN_B];
    double Pb = 0, t0b = 0, bb = 0, eb = 0, wb = 0, rb = 0, Msb = 0, Rsb = 0, ab = 0, nb = 0, cwb = 0, swb = 0;
    // ror = r / Rs ;  aor = a / Rs
    rb += g[EXO_P_ROR] / Rs;
    Rsb -= g[EXO_P_ROR] * r / (Rs * Rs);
    ab += g[EXO_P_AOR] / Rs;
    Rsb -= g[EXO_P_AOR] * d.a / (Rs * Rs);
    if (secondary) {  // fratio = sbr r^2 / Rs^2
      const double gf = g[EXO_P_FRATIO], k = r / Rs;
      o[EXO_IN_SBR] = gf * k * k;
      rb += gf * in[EXO_IN_SBR] * 2.0 * k / Rs;
      Rsb -= gf * in[EXO_IN_SBR] * 2.0 * k * k / Rs;
    } else {
      o[EXO_IN_SBR] = 0.0;
    }
    // light delay: c / Rs
    Rsb -= g[EXO_P_CLIGHT] * kCLight / (Rs * Rs);
    // cos i = f Rs b / a ;  sin i = sqrt(1 - cos^2 i) carries a cotangent with light delay only
    const double sini_ = sqrt(fmax(0.0, 1.0 - d.cosi * d.cosi));
    const double gci = g[EXO_P_COSI] - (sini_ > 0.0 ? g[EXO_P_SINI] * d.cosi / sini_ : 0.0);
    const double fb = gci * Rs * b / d.a;
    Rsb += gci * d.f * b / d.a;
    bb += gci * d.f * Rs / d.a;
    ab -= gci * d.cosi / d.a;
    // t_peri = t0 - M0 / n
    const double gtp = g[EXO_P_TP];
    t0b += gtp;
    const double M0b = -gtp / d.n;
    nb += g[EXO_P_N] + gtp * d.M0 / (d.n * d.n);
    if (!circular) {
      const double ome2 = 1.0 - e * e;
      eb += g[EXO_P_ECC] + fb * (d.sw / ome2 + (1.0 + e * d.sw) * 2.0 * e / (ome2 * ome2));
      swb += g[EXO_P_SINW] + fb * e / ome2;
      cwb += g[EXO_P_COSW];
      // M0 = E0 - e sin E0 ; E0 = 2 atan2(y, x)
      double sE, cE;
      sincos(d.E0, &sE, &cE);
      const double E0b = M0b * (1.0 - e * cE);
      eb -= M0b * sE;
      const double h2 = d.x * d.x + d.y * d.y;
      const double yb = E0b * 2.0 * d.x / h2, xb = -E0b * 2.0 * d.y / h2;
      const double se = sqrt(1.0 - e), pe = sqrt(1.0 + e);
      cwb += yb * se;
      eb -= yb * d.cw * 0.5 / se;
      swb += xb * pe;
      eb += xb * (1.0 + d.sw) * 0.5 / pe;
      wb = -cwb * d.sw + swb * d.cw;
    }
    // n = 2 pi / P ;  a = (G mtot P^2 / 4 pi^2)^(1/3)
    Pb += -nb * d.n / P + ab * 2.0 * d.a / (3.0 * P);
    Msb += ab * d.a / (3.0 * d.mtot);
    o[EXO_IN_PERIOD] = Pb; o[EXO_IN_T0] = t0b; o[EXO_IN_B] = bb; o[EXO_IN_ECC] = eb; o[EXO_IN_OMEGA] = wb;
    o[EXO_IN_R] = rb; o[EXO_IN_MSTAR] = Msb; o[EXO_IN_RSTAR] = Rsb; o[EXO_IN_MPLANET] = Msb;
#pragma unroll
    for (int k = 0; k < EXO_NIN; ++k) dst.store(i, k, o[k]);
}
// reverse of pack_ld: gld_row[3 | 6] = the cotangents of the draw's Green's-basis coefficients
template <class Src, class Dst>
__device__ __forceinline__ void pack_vjp_ld(const Src& src, int64_t draw, uint32_t flags, const double* __restrict__ gld_row,
                                            double sc, const Dst& dst) {
  const bool secondary = flags & EXO_FLAG_SECONDARY;
    const int nset = secondary ? 2 : 1;
    for (int s = 0; s < nset; ++s) {
      const double u1 = src.ld(draw, 2 * s, 2 * nset), u2 = src.ld(draw, 2 * s + 1, 2 * nset);
      const double g[3] = {sc * gld_row[3 * s], sc * gld_row[3 * s + 1], sc * gld_row[3 * s + 2]};
      const double c0 = 1.0 - u1 - 1.5 * u2, c1 = u1 + 2.0 * u2, c2 = -0.25 * u2;
      const double nrm = kPi * (c0 + c1 * (1.0 / 1.5)), inorm = 1.0 / nrm;
      // c_k = C_k / nrm :  dC/du1 = (-1, 1, 0), dC/du2 = (-1.5, 2, -0.25), dnrm/du1 = pi(-1 + 2/3), dnrm/du2 = pi(-1.5 + 4/3)
      const double dot = (g[0] * c0 + g[1] * c1 + g[2] * c2) * inorm * inorm;
      dst.store_ld(draw, 2 * s, 2 * nset, (-g[0] + g[1]) * inorm - dot * kPi * (-1.0 + 2.0 / 3.0));
      dst.store_ld(draw, 2 * s + 1, 2 * nset, (-1.5 * g[0] + 2.0 * g[1] - 0.25 * g[2]) * inorm - dot * kPi * (-1.5 + 4.0 / 3.0));
    }
  }

}  // namespace exo_pack
