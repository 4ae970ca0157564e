// exo_transit.hip -- HIP kernels + C ABI for the Kepler / limb-darkened transit
// part of the hot path (gfx950, wave64, fp64 VALU; no MFMA -- nothing here is a
// dense contraction).
//
// TWO paths through this file (runs_path() decides; section "Run-enumeration path" below has the details):
//
// * RUN ENUMERATION (sorted times, one exposure time -- or none -- for all cadences, no EXO_FLAG_EXACT_SCAN; timing tables
//   allowed when the sweep has transits only and no light delay): what every BASELINE config and every sampler leg takes.
//   A sweep (value, or value + VJP) is two launches, three when a draw is shared by several blocks:
//     transit_enum_kernel<true>   one wave per list (draw, planet, event): the record's conjunction windows (closed form,
//                                 refined to the contacts) and, by binary search in t, the RUN of cadences of every window,
//                                 with prefix sums -- caller vouched for sorted times (EXO_FLAG_SORTED_TIMES); otherwise
//                                 transit_window_kernel (+ sortedness blocks) and transit_enum_kernel<false>
//                                 (timing tables: transit_enum_ttv_kernel, a bin's windows periodic in t - shift[bin])
//     transit_runs_kernel         the solved cadences of the runs, dense, in full fp64 (eval_sample: Kepler solve, solution
//                                 vector, flux, reverse sweep into LDS gradient columns), values to a run-ordered array; the
//                                 dense output's zero fill is interleaved with the arithmetic (FillCursor); a block that owns
//                                 its draw (>= 512 draws) also finishes it; EXO_FLAG_SPARSE: runs + values ARE the output;
//                                 chi2 variants: the white-noise likelihood with its gradient, no (draw, cadence) array
//     transit_finish_kernel       (draws shared by several blocks, cadence-major flux, the three-sweep chi2) block partials ->
//                                 gparams / gld / sum(gflux flux); the runs' values to their cadences
//
// * LIST PATH (per-cadence exposure times, EXO_FLAG_EXACT_SCAN, timing tables together with occultations or light delay):
//   four launches, every cadence classified:
//     transit_window_kernel       per (draw, planet): where in mean anomaly an overlap is possible
//     transit_scan_kernel         classify blocks: which cadences can overlap the disk -> per-wave work lists (inside / limb);
//                                 fill blocks: flux = 0 everywhere
//     transit_heavy_kernel        the listed cadences, dense: the same eval_sample; per-block gradient partials
//     transit_vjp_reduce_kernel   block partials -> gparams, gld, sum(gflux flux)
//   Work units: a (draw, run of tiles_per_block tiles of 512 consecutive cadences); a lane owns cadences (sub-exposures and
//   planets are register loops), so a wave is a run of consecutive cadences: transits are contiguous in time and whole waves
//   are in / out of transit except at the edges.
// Per-(draw, planet) constants are derived once per block, staged in LDS and, in the heavy / runs kernels, pinned to scalar
// registers.
//
// Reference lines restated by the fused kernel (all under /root/reference/src/exoplanet):
//   orbits/keplerian.py:324-334   M = (t - t0 - tref) n ; kepler(M, e)
//   orbits/keplerian.py:400-409   r = a (1-e^2)/(1+e cos f) ; rotate
//   orbits/keplerian.py:303-314   omega rotation, inclination projection
//   orbits/keplerian.py:729-731,765-769   in-transit window test
//   light_curves/limb_dark.py:178-226     exposure stencil, b, los, s.c - 1, los > 0
//   light_curves/secondary_eclipse.py:45-70   flipped orbit + blend
//   orbits/ttv.py:158-187         bin edges / values, searchsorted, t - transit time of the bin
//                                 (the TTV template variants of the scan and heavy kernels)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"
#include "exo_contact.hpp"
#include "exo_math.hpp"
#include "exo_pack_core.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 2 * kBlock;         // cadences per tile: two per lane
#ifndef EXO_TARGET_BLOCKS
#define EXO_TARGET_BLOCKS (256 * 16)
#endif
constexpr int kTargetBlocks = EXO_TARGET_BLOCKS;  // ~16 resident-or-queued blocks per CU: fills the chip, amortises
                                         // the per-block constant staging and gradient reduction
constexpr uint32_t kFlagNoFluxDev = 0x80000000u;
constexpr int kNG = 12;                 // compact gradient slots per planet
constexpr int kMaxMerge = 8;            // scan blocks per heavy block, at most
constexpr int kWin = 7;                 // doubles per record written by transit_window_kernel
// compact slot order
enum { G_N = 0, G_TP, G_ECC, G_COSW, G_SINW, G_COSI, G_AOR, G_ROR, G_FR, G_PAD, G_SINI, G_CL };   // (SINI, CL: light delay only)
constexpr double kCLight = 37231.66360672704;   // R_sun / day (orbits/constants.py:36)

// Per-(draw, planet) constants derived once per block and staged in LDS.
struct PlanetConst {
  double n, tp, e, se, pe, sq1me2, cw, sw, ci, si, aor, ror, iror;
  double t0, period, iperiod, ts, te, fr, ts2, te2, isq1me2;
  double clr;   // speed of light, stellar radii per day (light delay)
  // fp32 copies for the conservative classifier of the scan kernel
  float ef, omf, sqf, cwf, swf, cif, zsf, thrf, zthrf, inthrf;
  // conjunction windows of the scan kernel's first test (see transit_window_kernel)
  double nrev, c0, dmid, half[2];
  // timing tables: first edge, bins per unit time, number of finite edges (see TtvRow::locate)
  double te0, tinv;
  int tfin;
};

struct Shared {
  PlanetConst pc[EXO_MAX_PLANETS];
  double c[6];
  double sdt[EXO_MAX_SUBEXP + 1];
  double sw[EXO_MAX_SUBEXP + 1];
  double red[16][16];  // reduce_columns: up to 16 slots x 16 partial sums
};

// mean anomaly of true anomaly f, continuous and increasing over all f
// Where can the planet overlap the disk at all?  Sky-plane separation (units of R*) is
//   rho sqrt(cos^2(w+f) + cos^2 i sin^2(w+f)) >= (a/R)(1-e) |cos(w+f)|,
// so b < 1 + r needs |cos(w+f)| < q = (1+r) / ((a/R)(1-e)) and, for the planet to be in
// front, sin(w+f) sin i > 0: f within asin(q) of the conjunction f_c = +-pi/2 - w -- a first bound,
// then tightened side by side to the contacts with the inclination and the distance actually reached
// (below: never inside them).  Mapped
// through E(f), M(E) (closed forms in this direction) that is a window of mean anomaly; the
// occultation window is the same about f_c + pi.  One thread per (draw, planet), run ahead of
// the scan kernel (libm's fp64 trigonometry would cost the scan kernel half its occupancy);
// the scan kernel's per-cadence test is then a phase wrap and a compare, and only cadences
// inside a window go on to the position-based classifier.
//   out[kWin] = { nrev = n / 2pi, c0 = -(tp nrev + mid_transit), mid_transit - mid_occultation,
//                 half_transit, half_occultation, inner_transit, inner_occultation }
// in revolutions of mean anomaly: the phase of cadence t is fma(t, nrev, c0), wrapped to +-1/2.
// q >= 1 or anything non-finite: halves = inf, every cadence goes on.  inner_*: an ESTIMATE of the
// half-width of the part of the window in which the small disk is wholly inside the large one
// (b + r < 1; 0 if never): the run-enumeration path sorts a window's cadences into "inside" and
// "limb" with it, so that a wave's vote on entering the arc geometry of the solution vector is
// nearly always unanimous.  A wrong estimate costs time, never a result.
//
// With EXO_FLAG_WINDOW the caller's contact-point windows (record slots T0, PERIOD, TS, TE[, TS2,
// TE2]; keplerian.py:729-731,765-769) are put in the same form instead -- revolutions of the
// orbit, centre t0 + (ts + te)/2, half-width (te - ts)/2 -- and they alone decide what is
// evaluated.
// (Blocks past the records' -- the run-enumeration path launches n_sorted more -- check that t is
// non-decreasing: one flag per kSortBlock cadences, the pair straddling the block's end included.)
#ifndef EXO_WINDOW_REFINE
#define EXO_WINDOW_REFINE 1   // (0: the first bound only -- A/B builds)
#endif
constexpr int kSortBlock = 4096;
constexpr int kWinLanes = 8;   // threads per record: (event, side of the conjunction, contact | inner point)
// The window of one record on EIGHT lanes (sub = 0 .. 7: (point, event, side); all eight must call it together: shuffles).
// On return: with EXO_FLAG_WINDOW every lane holds all seven numbers; otherwise lane sub = 0 holds w[0..3] and w[5] (the
// transit's), lane sub = 2 holds w[4] and w[6] (the occultation's).
__device__ __forceinline__ void window_lanes(const double* __restrict__ p, uint32_t flags, int sub, double* w) {
  const int which = sub >> 2, k = (sub >> 1) & 1, sd = sub & 1;   // point, event (0 transit, 1 occultation), side
  const double e = p[EXO_P_ECC], cw = p[EXO_P_COSW], sw = p[EXO_P_SINW];
  if (flags & EXO_FLAG_WINDOW) {   // (every lane: a dozen operations, and then every lane holds all seven)
    const double ip = 1.0 / p[EXO_P_PERIOD];
    const double ts = p[EXO_P_TS], te = p[EXO_P_TE], ts2 = p[EXO_P_TS2], te2 = p[EXO_P_TE2];
    const bool fin = (fabs(ts) < __builtin_inf()) && (fabs(te) < __builtin_inf());
    const bool fin2 = (fabs(ts2) < __builtin_inf()) && (fabs(te2) < __builtin_inf());
    const double mid = fin ? 0.5 * (ts + te) : 0.0, mid2 = fin2 ? 0.5 * (ts2 + te2) : 0.0;
    w[0] = ip;
    w[1] = -(p[EXO_P_T0] + mid) * ip;
    w[2] = (mid - mid2) * ip;
    // a hair wider than the reference's closed interval: a cadence exactly at a contact has zero flux
    w[3] = fin ? fma(0.5 * (te - ts) * ip, 1.0 + 1e-12, 1e-14) : __builtin_inf();
    w[4] = fin2 ? fma(0.5 * (te2 - ts2) * ip, 1.0 + 1e-12, 1e-14) : __builtin_inf();
    // inner parts: chord ratio sqrt((1-r)^2 - b^2) / sqrt((1+r)^2 - b^2) of the contact window,
    // b = impact parameter at the conjunction
    const double wn_ = sqrt(cw * cw + sw * sw), r_ = fabs(p[EXO_P_ROR]);
    const double sinw_ = wn_ > 0.0 ? sw / wn_ : 0.0;
    for (int q = 0; q < 2; ++q) {
      const double bk = fabs(p[EXO_P_AOR] * p[EXO_P_COSI]) * (1.0 - e * e) / (1.0 + (q ? -e : e) * sinw_);
      const double in2 = (1.0 - r_) * (1.0 - r_) - bk * bk, out2 = (1.0 + r_) * (1.0 + r_) - bk * bk;
      const double h = w[3 + q];
      w[5 + q] = (r_ < 1.0 && in2 > 0.0 && out2 > 0.0 && h < __builtin_inf()) ? 0.95 * h * sqrt(in2 / out2) : 0.0;
    }
    if (flags & EXO_FLAG_LIGHT_DELAY) {   // as below: the retarded time differs from t by at most this
      const double vmax = fabs(p[EXO_P_N] * p[EXO_P_AOR]) * (1.0 + e) / sqrt(1.0 - e * e);
      const double dmax = fabs(p[EXO_P_AOR]) * (1.0 + e) / (fabs(p[EXO_P_CLIGHT]) - vmax);
      const double wd = (dmax >= 0.0 ? dmax : __builtin_inf()) * ip * 1.05;
      w[3] += wd; w[4] += wd;
    }
    return;     // (flag-uniform: every lane of the launch takes this branch or none does)
  }
  // Eight threads per record -- (event, side, contact | inner point) -- each with the short serial chain of its own
  // point.  No forward trigonometry: the conjunction's true anomaly f0 = +-pi/2 - w (+ pi) has cos f0 = +-sin w,
  // sin f0 = +-cos w; every angle of the refinement is carried as its sine (all lie in [0, pi/2)) and composed with f0
  // by the addition formulas; distances enter as 1 / dist = (1 + e cos f) / (a (1 - e^2)).  What is left is one atan2
  // for w, one asin for the angle reached and one atan2 for E(f).  (One thread per record with libm's sin / cos / asin
  // in the loop: 19 us of a 320 us sweep; four threads: 11.6 us; this form: see docs/DESIGN_r1_r4.md 4.)
  const double nrev = p[EXO_P_N] * (0.5 / exo::kPi);
  const double wn = sqrt(cw * cw + sw * sw);
  const double q = (1.0 + fabs(p[EXO_P_ROR])) / (fabs(p[EXO_P_AOR]) * (1.0 - e) * wn);
  const bool bounded = (e >= 0.0 && e < 1.0) && (q < 0.999);   // else NaN everywhere / no bound: every cadence goes on
  const bool want = bounded && (k == 0 || (flags & EXO_FLAG_SECONDARY));
  double m_pt = 0.0;       // this thread's point (contact or inner point of its side), revolutions of mean anomaly
  bool has_in = false;
  if (want) {
    const double se = sqrt(1.0 - e), pe = sqrt(1.0 + e);
    const double s0 = p[EXO_P_SINI] < 0.0 ? -1.0 : 1.0, sk = k ? -1.0 : 1.0;
    const double sinw = sw / wn, cosw = cw / wn;
    const double cf0 = sk * s0 * sinw, sf0 = sk * s0 * cosw;
    const double f0 = 0.5 * s0 * exo::kPi - atan2(sw, cw) + k * exo::kPi;
    const double si2 = p[EXO_P_SINI] * p[EXO_P_SINI], ci2 = p[EXO_P_COSI] * p[EXO_P_COSI];
    const double lim = 1.0 + fabs(p[EXO_P_ROR]), semi = fabs(p[EXO_P_AOR]) * (1.0 - e * e);
    const double sgn = sd ? 1.0 : -1.0;
    double sphi;   // sine of the angle from the conjunction to this thread's point
    if (which == 0) {
      // The bound above is that of an edge-on orbit at its periastron distance.  At phase angle phi from the
      // conjunction the sky-plane separation is dist(f) sqrt(cos^2 i + sin^2 i sin^2 phi): the disks overlap only where
      // phi <= G(phi) = asin sqrt(((1 + r)^2 / dist(f0 +- phi)^2 - cos^2 i) / sin^2 i).  Each side of the conjunction on
      // its own, from the upper bound ub = asin q, never below the contact:
      //   dist falling away from the conjunction (G rising): ub <- G(ub);
      //   dist rising (G falling): lb = G(ub) is a lower bound of the contact, so G(lb) an upper one;
      //   an apsis inside the half-window: G at the smallest distance in it.
      // Three rounds leave the window within ~0.1 % of the contacts (C2: it was 6.8 % wider than them); a planet that
      // never reaches the disk (b > 1 + r) keeps only the safety margin.
      double su = q;
      if (EXO_WINDOW_REFINE && si2 > 1e-12) {
        const double isemi = 1.0 / semi, isi2 = 1.0 / si2, lim2 = lim * lim;
        auto sin_ang = [&](double u) {   // u = 1 / dist
          const double S = (lim2 * u * u - ci2) * isi2;
          return S <= 0.0 ? 0.0 : (S < 1.0 ? sqrt(S) : q);   // (NaN: no information)
        };
        auto u_at = [&](double sa) {     // 1 / dist at f0 + sgn * asin(sa)
          return fma(e, cf0 * sqrt(1.0 - sa * sa) - sgn * sf0 * sa, 1.0) * isemi;
        };
        const double u_c = fma(e, cf0, 1.0) * isemi;
        for (int it = 0; it < 3; ++it) {
          // an apsis (f = m pi) in [f0, f_end] <=> sin f changes sign over it (the interval is shorter than pi / 2);
          // it is the periastron <=> cos f0 > 0
          const double s_end = sf0 * sqrt(1.0 - su * su) + sgn * cf0 * su;
          const bool apsis = sf0 * s_end <= 0.0;
          const double u_end = u_at(su);
          if (!apsis && u_end <= u_c) {
            const double lb = fmin(sin_ang(u_end), su);
            su = fmin(su, sin_ang(u_at(lb)));
          } else {
            su = fmin(su, sin_ang((apsis && cf0 > 0.0) ? (1.0 + e) * isemi : fmax(u_end, u_c)));
          }
        }
      }
      sphi = su;
    } else {
      // inner part: |sky-plane x| < sqrt((1-r)^2 - b^2) at the conjunction's star-planet distance
      const double r = fabs(p[EXO_P_ROR]);
      const double dist = fabs(p[EXO_P_AOR]) * (1.0 - e * e) / (1.0 + (k ? -e : e) * sinw);
      const double bk = dist * fabs(p[EXO_P_COSI]);
      const double in2 = (1.0 - r) * (1.0 - r) - bk * bk;
      has_in = r < 1.0 && in2 > 0.0 && dist > 0.0;
      sphi = has_in ? fmin(0.95 * sqrt(in2) / dist, 1.0) : 0.0;
    }
    // the point's true anomaly f = f0 + sgn (phi (1 + 1e-6) + 1e-6) for a contact, f0 + sgn phi for an inner point:
    // its angle for the revolution count, its sine and cosine by the addition formulas (the margin: a rotation by
    // delta <= 2.6e-6, second order)
    const double phi = asin(sphi), cphi = sqrt(fmax(1.0 - sphi * sphi, 0.0));
    const double delta = which == 0 ? fma(phi, 1e-6, 1e-6) : 0.0;
    const double c1 = cf0 * cphi - sgn * sf0 * sphi, s1 = sf0 * cphi + sgn * cf0 * sphi;
    const double hd = 1.0 - 0.5 * delta * delta;
    const double cf = c1 * hd - sgn * delta * s1, sf = s1 * hd + sgn * delta * c1;
    // E(f) = f - 2 atan(beta sin f / (1 + beta cos f)), beta = e / (1 + sqrt(1 - e^2)): continuous in f, no wrap
    const double rt = se * pe, beta = e / (1.0 + rt);
    const double E = (f0 + sgn * (phi + delta)) - 2.0 * atan2(beta * sf, fma(beta, cf, 1.0));
    m_pt = (E - e * (rt * sf / fma(e, cf, 1.0))) * (0.5 / exo::kPi);
  }
  // the record's other numbers (lanes 8j .. 8j + 7 hold one record: no record straddles a wave)
  const double m_other = __shfl_xor(m_pt, 4, 64);    // (every shuffle outside the branches: all eight lanes take part)
  const int in_other = __shfl_xor((int)has_in, 4, 64);
  const double m_edge = which ? m_other : m_pt, m_in = which ? m_pt : m_other;
  has_in = has_in || in_other != 0;
  const double o_edge = __shfl_xor(m_edge, 1, 64), o_in = __shfl_xor(m_in, 1, 64);
  const double lo = sd ? o_edge : m_edge, hi = sd ? m_edge : o_edge;
  const double mid = 0.5 * (lo + hi);
  const double half = want ? 0.5 * (hi - lo) * (1.0 + 1e-5) + 1e-6 : __builtin_inf();
  // (about the window's centre, which the two contacts set: the smaller of the two sides)
  const double in_lo = sd ? o_in : m_in, in_hi = sd ? m_in : o_in;
  const double inner = (want && has_in) ? fmax(fmin(in_hi - mid, mid - in_lo), 0.0) : 0.0;
  const double mid_other = __shfl_xor(mid, 2, 64);   // the other event's centre
  double wd = 0.0;
  if (flags & EXO_FLAG_LIGHT_DELAY) {
    // the body is seen where it was up to |z|max / (c - |vz|max) earlier or later: widen by that much
    const double vmax = fabs(p[EXO_P_N] * p[EXO_P_AOR]) * (1.0 + e) / sqrt(1.0 - e * e);
    const double dmax = fabs(p[EXO_P_AOR]) * (1.0 + e) / (fabs(p[EXO_P_CLIGHT]) - vmax);
    wd = (dmax >= 0.0 ? dmax : __builtin_inf()) * fabs(nrev) * 1.05;   // (NaN or v >= c: no window)
  }
  // (valid on the lanes with sd == 0, which == 0: event 0 -> w[0..3], w[5]; event 1 -> w[4], w[6])
  if (k == 0) {
    w[0] = nrev;
    w[1] = bounded ? -fma(p[EXO_P_TP], nrev, mid) : -p[EXO_P_TP] * nrev;
    w[2] = (bounded && (flags & EXO_FLAG_SECONDARY)) ? mid - mid_other : 0.0;
    w[3] = half + wd;
    w[5] = inner;
  } else {
    w[4] = half + wd;
    w[6] = inner;
  }
}


__global__ __launch_bounds__(kBlock) void transit_window_kernel(const double* __restrict__ params, int64_t n_rec,
                                                                uint32_t flags, double* __restrict__ out,
                                                                const double* __restrict__ t = nullptr, int64_t n_cad = 0,
                                                                int32_t* __restrict__ sorted = nullptr,
                                                                int32_t* __restrict__ done = nullptr, int64_t n_done = 0) {
  const int n_rec_blocks = (int)((n_rec * kWinLanes + kBlock - 1) / kBlock);
  // (the per-draw block counters of the sweep that follows: transit_runs_kernel)
  if (done && (int64_t)blockIdx.x * kBlock + threadIdx.x < n_done) done[(int64_t)blockIdx.x * kBlock + threadIdx.x] = 0;
  if ((int)blockIdx.x >= n_rec_blocks) {
    const int sb = blockIdx.x - n_rec_blocks;
    const int64_t b0 = (int64_t)sb * kSortBlock;
    bool ok = true;
    for (int64_t k = b0 + threadIdx.x; k < b0 + kSortBlock && k + 1 < n_cad; k += kBlock) ok = ok && (t[k] <= t[k + 1]);   // NaN: not sorted
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) sorted[sb] = all;
    return;
  }
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = gid / kWinLanes;
  const int sub = (int)(gid - i * kWinLanes);
  if (i >= n_rec) return;   // (whole groups of eight: the shuffles below stay within a record)
  double w[kWin] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  window_lanes(params + i * EXO_NPAR, flags, sub, w);
  double* o = out + kWin * i;
  if (flags & EXO_FLAG_WINDOW) {
    if (sub == 0)
      for (int q = 0; q < kWin; ++q) o[q] = w[q];
  } else if (sub == 0) {
    o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = w[3]; o[5] = w[5];
  } else if (sub == 2) {
    o[4] = w[4]; o[6] = w[6];
  }
}

// ---------------------------------------------------------------------------
// Transit-timing variations (reference: orbits/ttv.py:158-187).  Every time is measured from its
// nearest labelled transit: planet p of a draw has n_edge bin edges (ascending, padded with +inf)
// and n_edge + 1 shifts; a time t falls in bin k = #{edges < t} (searchsorted, left) and is
// warped to t - shift[k] before anything else happens to it (shift[k] = transit time k - the
// record's t0; the mean anomaly and the window phase are then those of the unperturbed orbit).
// ---------------------------------------------------------------------------
struct Ttv {
  const double* edges;   // [n_draw][n_planet][n_edge]
  const double* shift;   // [n_draw][n_planet][n_edge + 1]
  double* gshift;        // [n_draw][n_planet][n_edge + 1], reverse sweep only
  int n_edge;
};

// one planet's table
struct TtvRow {
  const double* __restrict__ edges;
  const double* __restrict__ shift;
  int n_edge;
  __device__ __forceinline__ TtvRow(const Ttv& tv, int64_t rec)
      : edges(tv.edges + rec * tv.n_edge), shift(tv.shift + rec * (tv.n_edge + 1)), n_edge(tv.n_edge) {}
  // #{edges < t}: lower bound, branch-free steps (NaN t -> 0)
  __device__ __forceinline__ int bin(double t) const {
    int lo = 0, len = n_edge;
    while (len > 0) {
      const int half = len >> 1;
      const bool lt = edges[lo + half] < t;
      lo = lt ? lo + half + 1 : lo;
      len = lt ? len - half - 1 : half;
    }
    return lo;
  }
  // The same bin from a guess: labelled transits are nearly evenly spaced, so bin ~ 1 + (t - e0) *
  // inv; the two edges around the guess confirm it (two independent loads instead of a chain of
  // log2(n_edge) dependent ones), anything else falls back to the search.  lo / hi: the edges of
  // the bin (-inf / +inf at the ends).
  __device__ __forceinline__ int locate(double t, double e0, double inv, int n_fin, double& lo, double& hi) const {
    const double x = (t - e0) * inv;
    int g = (x > 0.0) ? ((x < (double)n_fin) ? (int)x + 1 : n_fin) : 0;
    lo = edges[g > 0 ? g - 1 : 0];
    hi = edges[g < n_edge ? g : n_edge - 1];
    if (!((g == 0 || lo < t) && (g == n_edge || !(hi < t)))) {
      g = bin(t);
      lo = edges[g > 0 ? g - 1 : 0];
      hi = edges[g < n_edge ? g : n_edge - 1];
    }
    lo = g > 0 ? lo : -__builtin_inf();
    hi = g < n_edge ? hi : __builtin_inf();
    return g;
  }
  // bin of a sub-exposure time tt of a cadence in bin k = (lo, hi]: k or, for an exposure that
  // reaches over an edge, a neighbour (one confirming load; anything else is searched for)
  __device__ __forceinline__ int neighbour(double tt, int k, double lo, double hi) const {
    if (!(tt > lo)) {          // lo is -inf for k = 0: never taken there
      k -= 1;
      if (k > 0 && !(edges[k - 1] < tt)) k = bin(tt);
    } else if (tt > hi) {      // hi is +inf for k = n_edge
      k += 1;
      if (k < n_edge && edges[k] < tt) k = bin(tt);
    }
    return k;
  }
  // Two times at once, shifts included: all six loads of the two guesses are issued before
  // anything is checked (one memory latency for the pair instead of four in a row).
  struct Hit { int k; double lo, hi, sh; };
  __device__ __forceinline__ int guess(double t, double e0, double inv, int n_fin) const {
    const double x = (t - e0) * inv;
    return (x > 0.0) ? ((x < (double)n_fin) ? (int)x + 1 : n_fin) : 0;
  }
  __device__ __forceinline__ void settle(Hit& h, double t) const {
    if (!((h.k == 0 || h.lo < t) && (h.k == n_edge || !(h.hi < t)))) {
      h.k = bin(t);
      h.lo = edges[h.k > 0 ? h.k - 1 : 0];
      h.hi = edges[h.k < n_edge ? h.k : n_edge - 1];
      h.sh = shift[h.k];
    }
    h.lo = h.k > 0 ? h.lo : -__builtin_inf();
    h.hi = h.k < n_edge ? h.hi : __builtin_inf();
  }
  __device__ __forceinline__ void locate2(double ta, double tb, double e0, double inv, int n_fin, Hit& a, Hit& b) const {
    a.k = guess(ta, e0, inv, n_fin);
    b.k = guess(tb, e0, inv, n_fin);
    a.lo = edges[a.k > 0 ? a.k - 1 : 0]; a.hi = edges[a.k < n_edge ? a.k : n_edge - 1]; a.sh = shift[a.k];
    b.lo = edges[b.k > 0 ? b.k - 1 : 0]; b.hi = edges[b.k < n_edge ? b.k : n_edge - 1]; b.sh = shift[b.k];
    settle(a, ta);
    settle(b, tb);
  }
};


__device__ __forceinline__ void stage_constants(Shared& sh, const double* __restrict__ params,
                                                const double* __restrict__ ld,
                                                const double* __restrict__ stencil_dt,
                                                const double* __restrict__ stencil_w, int n_sub,
                                                int n_planet, int64_t draw, bool secondary,
                                                const double* __restrict__ windows = nullptr,
                                                const Ttv* ttv = nullptr, int64_t ttv_first = 0) {
  const int tid = threadIdx.x;
  if (tid < n_planet) {
    const double* p = params + (draw * n_planet + tid) * EXO_NPAR;
    PlanetConst& c = sh.pc[tid];
    const double e = p[EXO_P_ECC];
    // e outside [0,1) -> NaN everywhere (docstring keplerian.py:58)
    const bool ok = (e >= 0.0) && (e < 1.0);
    c.n = p[EXO_P_N]; c.tp = p[EXO_P_TP]; c.e = e;
    c.se = ok ? sqrt(1.0 - e) : __builtin_nan("");
    c.pe = sqrt(1.0 + e);
    c.sq1me2 = c.se * c.pe;
    c.isq1me2 = 1.0 / c.sq1me2;
    c.cw = p[EXO_P_COSW]; c.sw = p[EXO_P_SINW];
    c.ci = p[EXO_P_COSI]; c.si = p[EXO_P_SINI];
    c.aor = p[EXO_P_AOR]; c.ror = p[EXO_P_ROR]; c.iror = 1.0 / p[EXO_P_ROR];
    c.t0 = p[EXO_P_T0]; c.period = p[EXO_P_PERIOD]; c.iperiod = 1.0 / p[EXO_P_PERIOD];
    c.ts = p[EXO_P_TS]; c.te = p[EXO_P_TE];
    c.fr = p[EXO_P_FRATIO]; c.ts2 = p[EXO_P_TS2]; c.te2 = p[EXO_P_TE2];
    c.clr = p[EXO_P_CLIGHT];
    // classifier: accept if (x^2 + y^2) (a/R)^2 < (1 + ror + margin)^2 with the fp32 position error
    // bound of exo::orbit_pos_f32 folded into the margin (never a false negative)
    const double margin = 2e-3 + c.aor * 1.6e-3;   // 2x the 8e-4 bound of exo::orbit_pos_f32
    const double lim = (1.0 + c.ror + margin) / c.aor;
    c.ef = (float)e; c.omf = (float)(1.0 - e); c.sqf = (float)c.sq1me2;
    c.cwf = (float)c.cw; c.swf = (float)c.sw; c.cif = (float)c.ci;
    c.zsf = (float)c.si;
    c.thrf = (float)(lim * lim) * 1.00001f;
    c.zthrf = (float)(-margin / c.aor);
    const double lin = fmax(1.0 - c.ror - margin, 0.0) / c.aor;
    c.inthrf = (float)(lin * lin);
    if (windows) {
      const double* wv = windows + kWin * (draw * n_planet + tid);
      c.nrev = wv[0]; c.c0 = wv[1]; c.dmid = wv[2]; c.half[0] = wv[3]; c.half[1] = wv[4];
    }
    if (ttv) {
      const TtvRow row(*ttv, ttv_first + draw * n_planet + tid);
      const int nf = row.bin(__builtin_inf());   // the padding is +inf
      c.tfin = nf;
      c.te0 = row.edges[0];
      const double width = nf > 1 ? row.edges[nf - 1] - row.edges[0] : 0.0;
      c.tinv = width > 0.0 ? (double)(nf - 1) / width : 0.0;
    }
  }
  const int nld = secondary ? 6 : 3;
  if (ld && tid >= 64 && tid < 64 + nld) sh.c[tid - 64] = ld[draw * nld + (tid - 64)];
  if (tid >= 128 && tid < 128 + n_sub) {
    sh.sdt[tid - 128] = stencil_dt ? stencil_dt[tid - 128] : 0.0;
    sh.sw[tid - 128] = stencil_w ? stencil_w[tid - 128] : 1.0;
  }
  __syncthreads();
}

// a wave-uniform double pinned to scalar registers
__device__ __forceinline__ double uniform(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// The heavy kernel's view of one planet: the constants eval_sample touches, pinned to scalar
// registers (they are the same for every lane; read from LDS they would sit in ~50 vector
// registers for the whole block, next to the elliptic-integral code that needs them all).
struct PlanetS {
  double n, tp, e, se, pe, sq1me2, isq1me2, cw, sw, ci, si, aor, ror, iror, fr, clr;
  __device__ __forceinline__ explicit PlanetS(const PlanetConst& c)
      : n(uniform(c.n)), tp(uniform(c.tp)), e(uniform(c.e)), se(uniform(c.se)), pe(uniform(c.pe)),
        sq1me2(uniform(c.sq1me2)), isq1me2(uniform(c.isq1me2)), cw(uniform(c.cw)), sw(uniform(c.sw)),
        ci(uniform(c.ci)), si(uniform(c.si)), aor(uniform(c.aor)), ror(uniform(c.ror)), iror(uniform(c.iror)),
        fr(uniform(c.fr)), clr(uniform(c.clr)) {}
};

// Gradient accumulators of the heavy kernel live in LDS, one column per thread
// ([slot][thread]: consecutive threads hit consecutive banks).  They are touched only
// by samples that overlap the disk, and keeping 17 doubles out of the register file is
// what lets two waves share a SIMD next to the elliptic-integral code.
// `add` is the LDS's own fp64 adder (ds_add_f64, no return value): one instruction, nothing to wait for -- as a
// read-modify-write in the wave (ds_read, wait ~100 cycles, v_add, ds_write) the dozen accumulations of a sample cost
// the kernel more stalled cycles than the arithmetic of its Kepler solve.  A column belongs to one thread and the LDS
// executes a wave's operations in order: the sums are those of the sequential loop, bit for bit, run after run.
struct GradAcc {
  double* col;  // &lds[0][threadIdx.x]
  __device__ __forceinline__ void add(int slot, double v) const {
    __hip_atomic_fetch_add(col + slot * kBlock, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
};

// One (cadence, sub-exposure, planet) sample.  Returns the flux contribution F
// and, if GRAD, adds gw * dF/d(theta) into the LDS accumulator columns.
// LDELAY (EXO_FLAG_LIGHT_DELAY; keplerian.py:411-470 with z0 = 0): the body is seen where it was at
// tt - D.  With the relative orbit's line-of-sight position z, velocity vz and acceleration az at tt
// (a first Kepler solve; a -> -a, r = a (1 - e cos E), vz = n a sin i (e cos w + cos(w + f)) / sqrt(1 -
// e^2), az = -n^2 z / (1 - e cos E)^3) the reference's
//     D = (c / az) ((1 + vz / c) - sqrt((1 + vz / c)^2 - 2 az (z0 - z) / c^2)),   (z0 - z) / (c + vz) if |az| < 1e-10
// is evaluated in the algebraically identical form  D = 2 q / (c (w + s)),  q = z0 - z, w = 1 + vz / c,
// s = sqrt(w^2 - 2 az q / c^2): no cancellation, and the small-az branch is its limit.  An occultation
// is the transit of the flipped orbit (keplerian.py:779-804), whose relative position, velocity and
// acceleration are the negatives: sigma = -1 below.  The reverse sweep takes the cotangent of the
// retarded time (-n Mbar of the second solve) back through D and the first solve by hand.
// CHI2 (one planet, no occultation, no exposure stencil: the sample IS the cadence's flux): `gw` carries the observed
// value and c2w its weight on the way in; the cotangent of F is formed once F is known, 2 w (F - obs), so the value
// and the gradient of a white-noise misfit take ONE evaluation per cadence.
template <bool GRAD, bool SECONDARY, bool LDELAY = false, bool CHI2 = false>
__device__ __forceinline__ double eval_sample(double tt, const PlanetS& c, const double* cld,
                                              double gw, const GradAcc& acc, double c2w = 0.0) {
  // saved by the delay computation for its reverse sweep
  double ld_cx = 0, ld_sx = 0, ld_den = 0, ld_z = 0, ld_vz = 0, ld_az = 0, ld_w = 0, ld_s = 0, ld_D = 0, ld_sig = 1, ld_t = tt;
  if (LDELAY) {
    const exo::KeplerHalf k1 = exo::kepler_half((tt - c.tp) * c.n, c.e, c.se, c.pe);
    const double X2 = k1.X * k1.X, Y2 = k1.Y * k1.Y;
    ld_cx = X2 - Y2; ld_sx = 2.0 * k1.X * k1.Y; ld_den = X2 + Y2;
    const double y1 = -c.aor * (c.sw * ld_cx + c.cw * ld_sx);
    ld_z = -c.si * y1;
    const double iden = exo::fast_div(1.0, ld_den);
    const double cwf = (c.cw * ld_cx - c.sw * ld_sx) * iden;
    ld_vz = -c.n * c.aor * c.isq1me2 * c.si * (c.e * c.cw + cwf);
    ld_az = -c.n * c.n * ld_z * iden * iden * iden;
    ld_sig = (SECONDARY && ld_z < 0.0) ? -1.0 : 1.0;   // behind the star: the flipped orbit's delay
    const double q = -ld_sig * ld_z, ic = exo::fast_div(1.0, c.clr);
    ld_w = fma(ld_sig * ld_vz, ic, 1.0);
    ld_s = sqrt(fma(-2.0 * ld_sig * ld_az * q, ic * ic, ld_w * ld_w));
    ld_D = 2.0 * q * ic / (ld_w + ld_s);
    tt -= ld_D;
  }
  const double M = (tt - c.tp) * c.n;
  const exo::KeplerHalf kh = exo::kepler_half(M, c.e, c.se, c.pe);
  const double X2 = kh.X * kh.X, Y2 = kh.Y * kh.Y;
  const double cx = X2 - Y2;            // (1 - e cos E) cos f = cos E - e
  const double sx = 2.0 * kh.X * kh.Y;  // (1 - e cos E) sin f = sqrt(1-e^2) sin E
  const double den = X2 + Y2;           // 1 - e cos E, without the cancellation at e -> 1
  // position relative to the star in units of R_star; the reference passes a = -self.a
  // (keplerian.py:540) and r = a (1-e^2)/(1+e cos f) = a (1 - e cos E)
  const double xo = -c.aor * cx, yo = -c.aor * sx;
  const double x1 = c.cw * xo - c.sw * yo;
  const double y1 = c.sw * xo + c.cw * yo;
  const double Ys = c.ci * y1;
  const double Z = -c.si * y1;
  const double b2 = x1 * x1 + Ys * Ys;
  const double lim = 1.0 + c.ror;
  const bool front = !(Z <= 0.0);  // NaN counts as in front so that NaN parameters propagate
  const bool behind = SECONDARY && (Z < 0.0);
  // NaN parameters must propagate: treat NaN b2 as active
  const bool act = (front || behind) && !(b2 >= lim * lim);
  if (!EXO_WAVE_ANY(act)) return 0.0;
  double ib;  // 1 / b, for the reverse sweep
  double b = exo::fast_sqrt_rs(b2, &ib);
  if (!(b2 > 0.0)) {  // centre of the disk (no direction: zero gradient through b), or NaN
    b = (b2 == 0.0) ? 0.0 : b2;
    ib = 0.0;
  }
  // transit: (b, ror) on the star; occultation: star of radius 1/ror passes in
  // front of the planet, in units of the planet radius (secondary_eclipse.py:56-58)
  const bool occ = SECONDARY && behind;
  const double bq = occ ? b * c.iror : b;
  const double rq = occ ? c.iror : c.ror;
  exo::SV sv;
  exo::quad_sv<GRAD>(act ? bq : 2.0 + rq, rq, sv);
  const double* cc = occ ? cld + 3 : cld;
  const double Fq = fma(sv.s0, cc[0], fma(sv.s1, cc[1], sv.s2 * cc[2])) - 1.0;
  double F;
  double wq = 1.0;  // dF/dFq
  if (SECONDARY) {
    const double inv = exo::fast_div(1.0, 1.0 + c.fr);
    wq = occ ? c.fr * inv : inv;
    F = act ? Fq * wq : 0.0;
  } else {
    F = act ? Fq : 0.0;
  }
  if (CHI2) gw = 2.0 * c2w * (F - gw);
  if (GRAD) {
    if (act) {
      const double gq = gw * wq;
      // limb-darkening coefficients
      const int o = occ ? 3 : 0;
      acc.add(kNG + o + 0, gq * sv.s0);
      acc.add(kNG + o + 1, gq * sv.s1);
      acc.add(kNG + o + 2, gq * sv.s2);
      double bbar_q = gq * fma(sv.db0, cc[0], fma(sv.db1, cc[1], sv.db2 * cc[2]));
      double rbar_q = gq * fma(sv.dr0, cc[0], fma(sv.dr1, cc[1], sv.dr2 * cc[2]));
      double bbar, rorbar;
      if (occ) {
        // bq = b / ror, rq = 1 / ror ; F = fr Fq / (1 + fr)
        bbar = bbar_q * c.iror;
        rorbar = -(bbar_q * b + rbar_q) * c.iror * c.iror;
        acc.add(G_FR, gw * Fq * (1.0 / ((1.0 + c.fr) * (1.0 + c.fr))));
      } else {
        bbar = bbar_q;
        rorbar = rbar_q;
        if (SECONDARY) acc.add(G_FR, -gw * Fq * (1.0 / ((1.0 + c.fr) * (1.0 + c.fr))));
      }
      acc.add(G_ROR, rorbar);
      const double x1bar = bbar * x1 * ib;
      const double Ysbar = bbar * Ys * ib;
      const double y1bar = Ysbar * c.ci;
      acc.add(G_COSI, Ysbar * y1);
      const double xobar = c.cw * x1bar + c.sw * y1bar;
      const double yobar = -c.sw * x1bar + c.cw * y1bar;
      acc.add(G_COSW, x1bar * xo + y1bar * yo);
      acc.add(G_SINW, -x1bar * yo + y1bar * xo);
      acc.add(G_AOR, -(xobar * cx + yobar * sx));
      const double cxbar = -c.aor * xobar, sxbar = -c.aor * yobar;
      // cx = cos E - e, sx = sqrt(1-e^2) sin E ; dE/dM = 1/den, dE/de = sin E/den
      // sin E, cos E back from (cx, sx) rather than from the half angles: two values live across the
      // solution vector instead of six
      const double sinE = sx * c.isq1me2;
      const double cosE = cx + c.e;
      const double iden = exo::fast_div(1.0, den);
      const double Ebar = fma(-sinE, cxbar, c.sq1me2 * cosE * sxbar);
      const double Mbar = Ebar * iden;
      acc.add(G_ECC, Mbar * sinE - cxbar - c.e * sinE * c.isq1me2 * sxbar);
      acc.add(G_N, Mbar * (tt - c.tp));
      acc.add(G_TP, -Mbar * c.n);
      if (LDELAY) {
        // tt = t_obs - D: Dbar = -(d / d tt) = -n Mbar; back through D = 2 q / (c (w + s))
        const double cl = c.clr, ic = 1.0 / cl;
        const double q = -ld_sig * ld_z, u = ld_w + ld_s;
        const double Dbar = -Mbar * c.n;
        double qbar = Dbar * 2.0 * ic / u;
        double clbar = -Dbar * ld_D * ic;
        const double ubar = -Dbar * ld_D / u;
        // s = sqrt(w^2 - 2 sig az q / c^2)
        const double discbar = ubar / (2.0 * ld_s);
        double wbar = ubar + discbar * 2.0 * ld_w;
        const double azbar = discbar * (-2.0 * ld_sig * q * ic * ic);
        qbar += discbar * (-2.0 * ld_sig * ld_az * ic * ic);
        clbar += discbar * (4.0 * ld_sig * ld_az * q * ic * ic * ic);
        // w = 1 + sig vz / c
        const double vzbar = wbar * ld_sig * ic;
        clbar -= wbar * ld_sig * ld_vz * ic * ic;
        acc.add(G_CL, clbar);
        // q = -sig z ;  az = -n^2 z / den^3
        const double id1 = 1.0 / ld_den, id3 = id1 * id1 * id1;
        double zbar = -ld_sig * qbar - azbar * c.n * c.n * id3;
        double nbar = -azbar * 2.0 * c.n * ld_z * id3;
        double denbar = azbar * 3.0 * c.n * c.n * ld_z * id3 * id1;
        // vz = vamp si P, vamp = -n a / sqrt(1 - e^2), P = e cw + cwf
        const double cwf = (c.cw * ld_cx - c.sw * ld_sx) * id1, Pq = c.e * c.cw + cwf;
        const double vamp = -c.n * c.aor * c.isq1me2;
        const double vampbar = vzbar * c.si * Pq, Pbar = vzbar * vamp * c.si;
        double sibar = vzbar * vamp * Pq;
        nbar += vampbar * (-c.aor * c.isq1me2);
        double aorbar = vampbar * (-c.n * c.isq1me2);
        // d(1/sqrt(1-e^2))/de = e / (1-e^2)^(3/2)
        double ebar = vampbar * (-c.n * c.aor) * c.e * c.isq1me2 * c.isq1me2 * c.isq1me2 + Pbar * c.cw;
        double cwbar = Pbar * c.e, swbar = 0.0;
        // cwf = (cw cx - sw sx) / den
        const double Nbar = Pbar * id1;
        denbar -= Pbar * cwf * id1;
        cwbar += Nbar * ld_cx; swbar -= Nbar * ld_sx;
        double cxbar1 = Nbar * c.cw, sxbar1 = -Nbar * c.sw;
        // z = -si y1 ;  y1 = -a (sw cx + cw sx)
        const double y1 = -c.aor * (c.sw * ld_cx + c.cw * ld_sx);
        sibar -= zbar * y1;
        const double y1bar = -zbar * c.si;
        aorbar -= y1bar * (c.sw * ld_cx + c.cw * ld_sx);
        swbar -= y1bar * c.aor * ld_cx; cwbar -= y1bar * c.aor * ld_sx;
        cxbar1 -= y1bar * c.aor * c.sw; sxbar1 -= y1bar * c.aor * c.cw;
        // first solve: cx = cos E - e, sx = sqrt(1-e^2) sin E, den = 1 - e cos E
        const double sinE1 = ld_sx * c.isq1me2, cosE1 = ld_cx + c.e;
        const double Ebar1 = -cxbar1 * sinE1 + sxbar1 * c.sq1me2 * cosE1 + denbar * c.e * sinE1;
        ebar += -cxbar1 - sxbar1 * c.e * c.isq1me2 * sinE1 - denbar * cosE1;
        const double Mbar1 = Ebar1 * id1;
        ebar += Mbar1 * sinE1;
        nbar += Mbar1 * (ld_t - c.tp);
        acc.add(G_TP, -Mbar1 * c.n);
        acc.add(G_N, nbar);
        acc.add(G_ECC, ebar);
        acc.add(G_COSW, cwbar);
        acc.add(G_SINW, swbar);
        acc.add(G_AOR, aorbar);
        acc.add(G_SINI, sibar);
      }
    }
  }
  return F;
}

// ---------------------------------------------------------------------------
// Scan kernel: the classifier
// ---------------------------------------------------------------------------
// one (planet, sub-exposure) sample of the classifier:
//   0 = cannot overlap the disk,
//   1 = overlaps, and the small disk looks wholly inside the large one (b + r < 1),
//   2 = overlaps, may be on the limb.
// 1 versus 2 only orders the work list (limb cadences run the arc geometry of the solution
// vector, the others do not, and a wave votes on whether to enter it): a wrong guess costs
// time, never a result.
template <bool SECONDARY, bool FAST>
__device__ __forceinline__ int classify_sample(double tt, const PlanetConst& c) {
  if (FAST) {
    // conservative fp32 classification: only the phase is fp64 (see exo::orbit_pos_f32);
    // every accepted cadence is re-evaluated in fp64 by the heavy kernel
    float cx, sx;
    exo::orbit_pos_f32((tt - c.tp) * c.n, c.ef, c.omf, c.sqf, &cx, &sx);
    const float x1 = c.cwf * cx - c.swf * sx;
    const float y1 = c.swf * cx + c.cwf * sx;
    const float Ys = c.cif * y1;
    const float Zs = c.zsf * y1;  // Z / (a/R)
    const bool vis = SECONDARY ? true : !(Zs <= c.zthrf);
    const float b2s = fmaf(x1, x1, Ys * Ys);
    return (vis && !(b2s >= c.thrf)) ? ((b2s < c.inthrf) ? 1 : 2) : 0;
  }
  const exo::KeplerHalf kh = exo::kepler_half((tt - c.tp) * c.n, c.e, c.se, c.pe);
  const double cx = kh.X * kh.X - kh.Y * kh.Y, sx = 2.0 * kh.X * kh.Y;
  const double x1 = c.cw * cx - c.sw * sx;  // position / (-a/R)
  const double y1 = c.sw * cx + c.cw * sx;
  const double Ys = c.ci * y1;
  const double Z = c.si * y1 * c.aor;       // = -sin(i) y1 (-a/R)
  const double b2 = (x1 * x1 + Ys * Ys) * c.aor * c.aor;
  const double lim = 1.0 + c.ror, lin = fmax(1.0 - c.ror, 0.0);
  const bool vis = SECONDARY ? true : !(Z <= 0.0);
  return (vis && !(b2 >= lim * lim)) ? ((b2 < lin * lin) ? 1 : 2) : 0;
}

// The scan kernel.  Two kinds of block share the launch (classify blocks first in the dispatch
// order, fill blocks after them), so that both kinds are resident together:
//   * fill blocks zero flux for their run of cadences -- a pure stream of 16-B stores
//     with nothing to wait for (the heavy kernel, ordered after this one on the stream,
//     overwrites the active cadences).  Stores and loads share one in-order counter on
//     gfx9, so a wave that alternates "load t, store 0" drains its stores every
//     iteration; giving the stores to waves that never load is what lets them run at
//     fill bandwidth;
//   * classify blocks read t (two cadences per lane, the next tile's pair prefetched),
//     decide which cadences can overlap the disk and append their offsets to per-wave
//     lists with ballot + mbcnt (no atomics).
// VEC2 (n_cad even and t 16-B aligned): a lane's two cadences are adjacent and t moves
// as 16-B loads; otherwise they are kBlock apart.
__device__ __forceinline__ void zero_fill(double* __restrict__ dst, int64_t n) {
  if (n <= 0) return;
  const int64_t head = (reinterpret_cast<uintptr_t>(dst) & 8) ? 1 : 0;
  if (threadIdx.x == 0 && head) dst[0] = 0.0;
  double2* __restrict__ d2 = reinterpret_cast<double2*>(dst + head);
  const int64_t n2 = (n - head) >> 1;
  int64_t k = threadIdx.x;
  // non-temporal: the zeros are not read again before they reach HBM, and keeping them out of
  // L2 leaves it to the heavy kernel that follows (measured: scan -21 us, heavy -10 us)
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* __restrict__ q2 = reinterpret_cast<v2d*>(d2);
  const v2d z = {0.0, 0.0};
  for (; k < n2; k += kBlock) __builtin_nontemporal_store(z, q2 + k);
  if (threadIdx.x == 0 && ((n - head) & 1)) dst[n - 1] = 0.0;
}

// first test of the scan kernel: is the wrapped phase within lim of a conjunction?  NaN -> yes.
// x - rint(x) through the 1.5 * 2^52 shift (two full-rate adds; |x| < 2^51 revolutions).
__device__ __forceinline__ double frac_rev(double x) {
  const double kShift = 6755399441055744.0;
  return x - ((x + kShift) - kShift);
}
template <bool SECONDARY>
__device__ __forceinline__ bool near_conjunction(double t, double nrev, double c0, double dmid, double lim0,
                                                 double lim1) {
  const double x = fma(t, nrev, c0);
  bool cand = !(fabs(frac_rev(x)) > lim0);
  if (SECONDARY) cand = cand || !(fabs(frac_rev(x + dmid)) > lim1);
  return cand;
}


// sum over the wave, valid in lane 63: row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31
// (in-register DPP moves; a ds_bpermute butterfly costs an LDS round trip per step)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_last(double v) {
  v = dpp_add<0x111, 0xf>(v);
  v = dpp_add<0x112, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

// Reverse sweep of the warp: d L / d shift[k] is the sum over the samples of bin k of their
// d L / d t_periastron (both enter as t - shift - tp).  eval_sample leaves that sum in the lane's
// G_TP column; after every cadence it is moved to the bin and to the G_PAD column, which ends up
// holding the planet's total.  Each wave keeps kBinSlots bins in LDS (direct-mapped on the bin
// number: a block works through a run of consecutive cadences, i.e. a few transits; one lane per
// wave touches them, so plain loads and stores -- LDS fp64 atomics measured 100 us slower per
// sweep) and sends them to the output table with one hardware fp64 atomic each when the planet is
// done; a bin that finds its slot taken goes to the table directly.  The table is the one place
// where the summation order -- and with it the last bits -- depends on scheduling.
constexpr int kBinSlots = 16;
struct BinCache {
  double sum[kWaves][kBinSlots];
  int id[kWaves][kBinSlots];
};
struct TtvGrad {
  double* __restrict__ col;   // &lds_acc[0][threadIdx.x]
  double* __restrict__ grow;  // gshift row of this (draw, planet)
  BinCache* cache;
  __device__ __forceinline__ double take() const {
    const double d = col[G_TP * kBlock];
    col[G_TP * kBlock] = 0.0;
    col[G_PAD * kBlock] += d;
    return d;
  }
  // one lane on its own (its bin changed in the middle of an exposure)
  __device__ __forceinline__ void flush_lane(int k) const {
    const double d = take();
    if (d != 0.0) unsafeAtomicAdd(grow + k, d);
  }
  // The whole wave, after a cadence.  A wave holds 64 consecutive list entries, i.e. cadences of
  // one transit or of two neighbouring ones: one pass per distinct bin, each a wave sum and one
  // addition by the last lane.
  __device__ __forceinline__ void flush_wave(int k) const {
    const double d = take();
    unsigned long long todo = __ballot(d != 0.0);
    while (todo) {
      const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
      const int k0 = __builtin_amdgcn_readlane(k, first);
      const bool mine = k == k0;
      const double sum = wave_sum_last(mine ? d : 0.0);
      todo &= ~__ballot(mine);
      // (wave number through a scalar register: an address built from threadIdx.x is kept by the
      // compiler across the whole loop -- in scratch, and a scratch reload waits for every load
      // in flight, the prefetched list entries included)
      const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), slot = k0 & (kBinSlots - 1);
      if ((threadIdx.x & 63) == 63) {
        const int owner = cache->id[w][slot];
        if (owner == k0) {
          cache->sum[w][slot] += sum;
        } else if (owner < 0) {
          cache->id[w][slot] = k0;
          cache->sum[w][slot] = sum;
        } else {
          unsafeAtomicAdd(grow + k0, sum);
        }
      }
    }
  }
  // Run-enumeration path, a list whose runs carry their bins: the wave's sums go to its own row of a [wave][run]
  // table in LDS (q = the lane's run within the batch; a wave holds cadences of one or two runs)
  __device__ __forceinline__ void flush_runs(int q, double* __restrict__ tab, int row_len) const {
    const double d = take();
    unsigned long long todo = __ballot(d != 0.0);
    while (todo) {
      const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
      const int q0 = __builtin_amdgcn_readlane(q, first);
      const bool mine = q == q0;
      const double sum = wave_sum_last(mine ? d : 0.0);
      todo &= ~__ballot(mine);
      const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      if ((threadIdx.x & 63) == 63) tab[w * row_len + q0] += sum;
    }
  }
  // block-wide, between planets: bins -> output table
  __device__ __forceinline__ void drain() const {
    __syncthreads();
    if (threadIdx.x < kWaves * kBinSlots) {
      const int k = (&cache->id[0][0])[threadIdx.x];
      if (k >= 0) unsafeAtomicAdd(grow + k, (&cache->sum[0][0])[threadIdx.x]);
      (&cache->id[0][0])[threadIdx.x] = -1;
    }
    __syncthreads();
  }
};


constexpr int kScanDraws = 4;  // draws per classify block on the single-planet path

// ballot + mbcnt append of the active lanes' offsets to a per-wave list (no atomics).  The list
// is two-ended: kind 1 grows up from lst[0], kind 2 grows down from lst[cap - 1].
struct ListCount {
  int in, limb;
};
__device__ __forceinline__ void append_active(int kind, int off, int32_t* __restrict__ lst, int cap, ListCount& cnt) {
  const unsigned long long b1 = __ballot(kind == 1), b2 = __ballot(kind == 2);
  if (kind == 1) {
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0));
    lst[cnt.in + before] = off;
  } else if (kind == 2) {
    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
    lst[cap - 1 - (cnt.limb + before)] = off;
  }
  cnt.in += __popcll(b1);
  cnt.limb += __popcll(b2);
}

// flags & kFlagGrouped: classify blocks take kScanDraws consecutive draws each (single planet,
// conjunction windows, one exposure time): t is loaded once per kScanDraws draws and the per-draw
// window constants sit in scalar registers.
constexpr uint32_t kFlagGrouped = 0x40000000u;

// (TTV: held to five waves per SIMD like the others -- four classify blocks per CU are resident
// at the start of a sweep, and the fifth slot is what lets fill blocks run beside them)
template <bool SECONDARY, bool FAST, bool VEC2, bool TTV = false>
__global__ __launch_bounds__(kBlock, (FAST && TTV) ? 5 : 1) void transit_scan_kernel(
    const double* __restrict__ t, int64_t n_cad, const double* __restrict__ texp, int64_t n_texp,
    const double* __restrict__ stencil_dt, int n_sub, const double* __restrict__ params, int n_planet,
    uint32_t flags, int tiles_per_block, int blocks_per_draw, int64_t n_draw, int64_t n_classify,
    double* __restrict__ flux, int32_t* __restrict__ counts, int32_t* __restrict__ list,
    const double* __restrict__ windows, Ttv ttv) {
  __shared__ Shared sh;
  // 1-D launch: classify blocks first (they feed the next kernel and should start early),
  // fill blocks after them; workgroups go to the 8 XCDs round-robin on the linear id, so both
  // kinds spread over all of them.
  int64_t work = blockIdx.x;
  if (work >= n_classify) {
    work -= n_classify;
    const int64_t draw = work / blocks_per_draw;
    const int bx = (int)(work - draw * blocks_per_draw);
    const int64_t lo = (int64_t)bx * tiles_per_block * kTile;
    const int64_t hi = lo + (int64_t)tiles_per_block * kTile;
    const int64_t npl = (flags & EXO_FLAG_PER_PLANET) ? n_planet : 1;
    zero_fill(flux + (draw * n_cad + lo) * npl, ((hi < n_cad ? hi : n_cad) - lo) * npl);
    return;
  }
  const bool grouped = flags & kFlagGrouped;
  const bool window = flags & EXO_FLAG_WINDOW;
  const bool stage1 = FAST || window;
  const int64_t unit = work / blocks_per_draw;  // draw, or group of kScanDraws draws
  const int bx = (int)(work - unit * blocks_per_draw);
  const int64_t draw = grouped ? unit * kScanDraws : unit;
  const int nd = grouped ? (int)((n_draw - draw) < kScanDraws ? (n_draw - draw) : kScanDraws) : 1;
  // TTV: this block's rows of the timing tables (its draw's planets, or its draws' single planets)
  // are copied to LDS when they fit: under the fill blocks' store stream a lookup that goes to
  // L2 waits microseconds, and a tile that crosses a bin boundary needs two in a row.
  constexpr int kTabMax = TTV ? 2048 : 1;
  __shared__ double s_tab[kTabMax];
  Ttv tl = ttv;                                                  // the tables as this block reads them
  int64_t row0 = grouped ? draw : draw * n_planet;               // table row of the block's first record
  if (TTV) {
    const int rows = grouped ? nd : n_planet, ne = ttv.n_edge;
    if (rows * (2 * ne + 1) <= kTabMax) {
      const double* __restrict__ src_e = ttv.edges + row0 * ne;
      const double* __restrict__ src_s = ttv.shift + row0 * (ne + 1);
      for (int q = threadIdx.x; q < rows * ne; q += kBlock) s_tab[q] = src_e[q];
      for (int q = threadIdx.x; q < rows * (ne + 1); q += kBlock) s_tab[rows * ne + q] = src_s[q];
      tl.edges = s_tab;
      tl.shift = s_tab + rows * ne;
      row0 = 0;
      __syncthreads();
    }
  }
  // grouped: the nd consecutive single-planet records are staged as if they were nd planets of one draw
  stage_constants(sh, params + (grouped ? draw * EXO_NPAR : 0), nullptr, stencil_dt, nullptr, n_sub,
                  grouped ? nd : n_planet, grouped ? 0 : draw, SECONDARY,
                  stage1 ? windows + (grouped ? kWin * draw : 0) : nullptr, TTV ? &tl : nullptr,
                  grouped ? row0 : row0 - draw * n_planet);
  // the windows are widened by the half-span of the exposure stencil; the reference widens its
  // contact windows by texp / 2 whatever the stencil (keplerian.py:765-769)
  double span = window ? 0.5 : 0.0;
  if (!window)
    for (int k = 0; k < n_sub; ++k) span = fmax(span, fabs(sh.sdt[k]));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t blk_base = (int64_t)bx * tiles_per_block * kTile;
  const int64_t list_stride = (int64_t)tiles_per_block * 128;
  const int64_t wave_slot = ((int64_t)draw * blocks_per_draw + bx) * kWaves + wave;  // of the first draw
  const int64_t slot_stride = (int64_t)blocks_per_draw * kWaves;                    // draw to draw
  const int o0 = VEC2 ? 2 * (int)threadIdx.x : (int)threadIdx.x;
  const int o1 = o0 + (VEC2 ? 1 : kBlock);
  auto load_pair = [&](int tile, double& a, double& b) {
    const int64_t i0 = blk_base + tile * kTile + o0, i1 = blk_base + tile * kTile + o1;
    if (VEC2) {
      // n_cad even and i0 even: the pair is valid or invalid together
      const double2 v = (i0 < n_cad) ? *reinterpret_cast<const double2*>(t + i0) : double2{0.0, 0.0};
      a = v.x; b = v.y;
    } else {
      a = (i0 < n_cad) ? t[i0] : 0.0;
      b = (i1 < n_cad) ? t[i1] : 0.0;
    }
  };
  double nx0, nx1;
  load_pair(0, nx0, nx1);
  if (grouped) {
    // Common case per tile: 16 (draw, cadence) phase tests, five full-rate operations each, and
    // no lane near a conjunction.  Otherwise (transits are contiguous in time and aligned across
    // neighbouring draws, so this is ~5% of the tiles) a rolled loop over the draws runs the
    // position-based classifier on the candidates; the per-draw counts live in LDS there.
    __shared__ ListCount s_cnt[kWaves][kScanDraws];
    if (lane < kScanDraws) s_cnt[wave][lane] = ListCount{0, 0};
    const double te = n_texp ? texp[0] : 0.0;
    double nrev[kScanDraws], c0[kScanDraws], dmid[kScanDraws], lim0[kScanDraws], lim1[kScanDraws];
#pragma unroll
    for (int j = 0; j < kScanDraws; ++j) {
      const PlanetConst& c = sh.pc[j < nd ? j : 0];
      nrev[j] = uniform(c.nrev); c0[j] = uniform(c.c0); dmid[j] = uniform(c.dmid);
      const double widen = fabs(te) * span * fabs(c.nrev);
      lim0[j] = uniform(c.half[0] + widen);
      lim1[j] = SECONDARY ? uniform(c.half[1] + widen) : 0.0;
    }
    // TTV: each draw's current bin -- its edges and its shift -- rides along in scalar registers.
    // A tile whose cadences (and their exposures) all lie strictly inside that bin costs four
    // compares and a subtraction more than without timing tables; any other tile (a bin boundary
    // every few tiles; every tile if the times are not sorted) looks its cadences up one by one
    // and leaves the bin of its last cadence behind for the next tile.
    struct BinNow { double lo, hi, sh; };
    __shared__ BinNow s_now[kWaves][kScanDraws];
    double b_lo[kScanDraws], b_hi[kScanDraws], b_sh[kScanDraws];
#pragma unroll
    for (int j = 0; j < kScanDraws; ++j) {
      // no bin yet; draws past the end of the batch: one bin that holds everything
      b_lo[j] = j < nd ? __builtin_inf() : -__builtin_inf();
      b_hi[j] = -b_lo[j];
      b_sh[j] = 0.0;
    }
    // exposures reaching over an edge are looked at sample by sample (never under the caller's
    // windows: those warp the mid-exposure time only, like the reference's in_transit)
    const double hw = (TTV && !window) ? uniform(fma(fabs(te) * span, 1e-12, fabs(te) * span)) : 0.0;
    auto process = [&](int tile, double tv0, double tv1) {
      const double tv[2] = {tv0, tv1};
      unsigned cand = 0;
      unsigned redo = 0;   // TTV: draws whose cached bin does not hold the whole tile (wave-uniform)
#pragma unroll
      for (int j = 0; j < kScanDraws; ++j) {
        double sh_j = 0.0;
        if (TTV) {
          const bool inside = (tv0 - b_lo[j] > hw) && (b_hi[j] - tv0 > hw) && (tv1 - b_lo[j] > hw) && (b_hi[j] - tv1 > hw);
          if (__ballot(!inside) != 0) {
            redo |= 1u << j;
            continue;
          }
          sh_j = b_sh[j];
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
          cand |= near_conjunction<SECONDARY>(tv[v] - sh_j, nrev[j], c0[j], dmid[j], lim0[j], lim1[j]) ? (1u << (2 * j + v)) : 0u;
      }
      if (TTV && redo) {
#pragma unroll 1
        for (int j = 0; j < nd; ++j) {
          if (!((redo >> j) & 1u)) continue;
          const PlanetConst& c = sh.pc[j];
          const TtvRow row(tl, row0 + j);
          const double widen = fabs(te) * span * fabs(c.nrev);
          TtvRow::Hit hit[2];
          row.locate2(tv0, tv1, c.te0, c.tinv, c.tfin, hit[0], hit[1]);
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const bool mixed = !window && n_texp && (!(tv[v] - hit[v].lo > hw) || !(hit[v].hi - tv[v] > hw));
            const bool near = mixed || near_conjunction<SECONDARY>(tv[v] - hit[v].sh, c.nrev, c.c0, c.dmid,
                                                                   c.half[0] + widen, c.half[1] + widen);
            cand |= near ? (1u << (2 * j + v)) : 0u;
          }
          if (lane == 63) s_now[wave][j] = BinNow{hit[1].lo, hit[1].hi, hit[1].sh};
        }
        // (same wave wrote them: program order is enough)
#pragma unroll
        for (int j = 0; j < kScanDraws; ++j) {
          if (((redo >> j) & 1u) && j < nd) {
            const BinNow nb = s_now[wave][j];
            b_lo[j] = uniform(nb.lo); b_hi[j] = uniform(nb.hi); b_sh[j] = uniform(nb.sh);
          }
        }
      }
      // draws past the end of the batch
      cand &= (1u << (2 * nd)) - 1u;
      if (__ballot(cand != 0) == 0) return;
      const int off[2] = {tile * kTile + o0, tile * kTile + o1};
#pragma unroll 1
      for (int j = 0; j < nd; ++j) {
        int32_t* __restrict__ lst = list + (wave_slot + j * slot_stride) * list_stride;
        ListCount cnt = s_cnt[wave][j];
#pragma unroll 1
        for (int v = 0; v < 2; ++v) {
          int kind = 0;
          if ((cand >> (2 * j + v)) & 1u) {
            if (TTV) {
              const PlanetConst& c = sh.pc[j];
              const TtvRow row(tl, row0 + j);
              double e_lo, e_hi;
              const int kb = row.locate(tv[v], c.te0, c.tinv, c.tfin, e_lo, e_hi);
              const double shv = row.shift[kb];
              const bool mixed = !window && n_texp && (!(tv[v] - e_lo > hw) || !(e_hi - tv[v] > hw));
              for (int k = 0; k < n_sub; ++k) {
                const double tt = fma(te, sh.sdt[k], tv[v]);
                double shk = shv;
                if (mixed) shk = row.shift[row.neighbour(tt, kb, e_lo, e_hi)];
                kind = max(kind, classify_sample<SECONDARY, FAST>(tt - shk, c));
              }
            } else {
              for (int k = 0; k < n_sub; ++k)
                kind = max(kind, classify_sample<SECONDARY, FAST>(fma(te, sh.sdt[k], tv[v]), sh.pc[j]));
            }
            if (window) kind = max(kind, 1);  // the caller's window decides; the classifier only sorts
          }
          append_active((blk_base + off[v] < n_cad) ? kind : 0, off[v], lst, (int)list_stride, cnt);
        }
        if (lane == 0) s_cnt[wave][j] = cnt;
      }
    };
    // t runs kAhead tiles ahead of the tests (a block may be alone on its SIMD: no other wave
    // hides the load latency)
    constexpr int kAhead = 4;
    double ring[kAhead][2];
    ring[0][0] = nx0; ring[0][1] = nx1;
#pragma unroll
    for (int u = 1; u < kAhead; ++u) {
      ring[u][0] = ring[u][1] = 0.0;
      if (u < tiles_per_block) load_pair(u, ring[u][0], ring[u][1]);
    }
    for (int tile0 = 0; tile0 < tiles_per_block; tile0 += kAhead) {
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int tile = tile0 + u;
        if (tile < tiles_per_block) {
          const double a = ring[u][0], b = ring[u][1];
          if (tile + kAhead < tiles_per_block) load_pair(tile + kAhead, ring[u][0], ring[u][1]);
          process(tile, a, b);
        }
      }
    }
    if (lane < nd) {
      const ListCount cnt = s_cnt[wave][lane];
      counts[2 * (wave_slot + lane * slot_stride)] = cnt.in;
      counts[2 * (wave_slot + lane * slot_stride) + 1] = cnt.limb;
    }
    return;
  }
  int32_t* __restrict__ my_list = list + wave_slot * list_stride;
  ListCount cnt{0, 0};
  // TTV: per wave and planet, the bin of the wave's last cadence (edges, shift, number)
  struct GenBin { double lo, hi, sh; int k; };
  __shared__ GenBin s_gbin[kWaves][EXO_MAX_PLANETS];
  if (TTV && lane < n_planet) s_gbin[wave][lane] = GenBin{__builtin_inf(), -__builtin_inf(), 0.0, 0};   // no bin yet
  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const double tv[2] = {nx0, nx1};
    if (tile + 1 < tiles_per_block) load_pair(tile + 1, nx0, nx1);
    const int off[2] = {tile * kTile + o0, tile * kTile + o1};
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int64_t i = blk_base + off[v];
      const bool valid = i < n_cad;
      const double te = (n_texp == 0) ? 0.0 : (n_texp == 1 ? texp[0] : (valid ? texp[i] : 0.0));
      int kind = 0;
      for (int p = 0; p < n_planet; ++p) {
        const PlanetConst& c = sh.pc[p];
        bool cand = true;
        if (TTV) {
          // the cadence in its own bin; an exposure that reaches into the next bin has
          // sub-exposures measured from another transit: no window argument covers those,
          // the classifier sees each of them (the caller's windows, like the reference's
          // in_transit, warp the mid-exposure time only)
          // The wave's last bin of this planet is tried first (LDS broadcast): times are usually
          // sorted, and a bin holds thousands of cadences.
          const TtvRow row(tl, row0 + p);
          const double hw = window ? 0.0 : fma(fabs(te) * span, 1e-12, fabs(te) * span);   // (the product was rounded)
          const GenBin nb = s_gbin[wave][p];
          double e_lo = nb.lo, e_hi = nb.hi, shv = nb.sh;
          int kb = nb.k;
          const bool inside = (tv[v] - e_lo > hw) && (e_hi - tv[v] > hw);
          if (__ballot(!inside) != 0) {
            kb = row.locate(tv[v], c.te0, c.tinv, c.tfin, e_lo, e_hi);
            shv = row.shift[kb];
            if (lane == 63) s_gbin[wave][p] = GenBin{e_lo, e_hi, shv, kb};
          }
          const double tw = tv[v] - shv;
          const bool mixed = !window && n_texp && (!(tv[v] - e_lo > hw) || !(e_hi - tv[v] > hw));
          if (stage1 && !mixed) {
            const double widen = fabs(te) * span * fabs(c.nrev);
            cand = near_conjunction<SECONDARY>(tw, c.nrev, c.c0, c.dmid, c.half[0] + widen, c.half[1] + widen);
          }
          if (cand) {
            int kp = window ? 1 : 0;
            for (int k = 0; k < n_sub; ++k) {
              const double tt = fma(te, sh.sdt[k], tv[v]);
              double shk = shv;
              if (mixed) shk = row.shift[row.neighbour(tt, kb, e_lo, e_hi)];
              kp = max(kp, classify_sample<SECONDARY, FAST>(tt - shk, c));
            }
            kind = max(kind, kp);
          }
          continue;
        }
        if (stage1) {
          const double widen = fabs(te) * span * fabs(c.nrev);
          cand = near_conjunction<SECONDARY>(tv[v], c.nrev, c.c0, c.dmid, c.half[0] + widen, c.half[1] + widen);
        }
        if (cand) {
          int kp = window ? 1 : 0;  // the caller's window decides; the classifier only sorts
          for (int k = 0; k < n_sub; ++k)
            kp = max(kp, classify_sample<SECONDARY, FAST>(fma(te, sh.sdt[k], tv[v]), c));
          kind = max(kind, kp);
        }
      }
      append_active(valid ? kind : 0, off[v], my_list, (int)list_stride, cnt);
    }
  }
  if (lane == 0) {
    counts[2 * wave_slot] = cnt.in;
    counts[2 * wave_slot + 1] = cnt.limb;
  }
}

// ---------------------------------------------------------------------------
// Heavy kernel: the cadences on the work lists of up to kMaxMerge scan blocks of one draw,
// concatenated ("inside" runs first, then "limb" runs) and processed densely, 256 at a time:
// Kepler solve in fp64, solution vector with its elliptic integrals, flux, and -- GRAD -- the
// reverse sweep into per-planet gradient slots that live in LDS columns for the whole block
// and are reduced once per planet in a fixed order (bit-reproducible).
// ---------------------------------------------------------------------------
// Sum the kBlock per-thread columns of accumulator slots [first, first + n) and write the n
// totals to out[0..n).  Two passes through LDS in a fixed order (bit-reproducible): thread
// (slot, c) adds the 16 columns c, c + 16, ..., then one thread per slot adds the 16 partials.
// A shuffle tree per slot costs 17 x 6 dependent cross-lane hops per block and was the bulk of
// the heavy kernel's per-block overhead.
__device__ __forceinline__ void reduce_columns(double (*acc)[kBlock], double (*red)[16], int first, int n,
                                               double* __restrict__ out) {
  __syncthreads();
  const int s = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (s < n) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < kBlock / 16; ++i) v += acc[first + s][c + 16 * i];
    red[s][c] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[threadIdx.x][i];
    out[threadIdx.x] = v;
  }
  __syncthreads();
}

// two waves per SIMD (<= 256 registers): measured 0.727 ms vs 0.776 ms per sweep at one wave
// (A/B via EXOPLANET_AMD_LIB), despite ~136 B/lane of scratch in the gradient variant
#ifndef EXO_HEAVY_MIN_WAVES
#define EXO_HEAVY_MIN_WAVES 2
#endif
template <bool GRAD, bool SECONDARY, bool TTV = false>
__global__ __launch_bounds__(kBlock, EXO_HEAVY_MIN_WAVES) void transit_heavy_kernel(
    const double* __restrict__ t, int64_t n_cad, const double* __restrict__ texp, int64_t n_texp,
    const double* __restrict__ stencil_dt, const double* __restrict__ stencil_w, int n_sub,
    const double* __restrict__ params, const double* __restrict__ ld, int n_planet, uint32_t flags,
    int tiles_per_block, int blocks_per_draw, int merge, const int32_t* __restrict__ counts,
    const int32_t* __restrict__ list, const double* __restrict__ gflux, double* __restrict__ flux,
    double* __restrict__ partial, const double* __restrict__ windows, Ttv ttv) {
  __shared__ Shared sh;
  __shared__ int s_pre[2 * kWaves * kMaxMerge + 1];
  const int64_t draw = blockIdx.y;
  __shared__ BinCache s_bins;
  if (GRAD && TTV && threadIdx.x < kWaves * kBinSlots) (&s_bins.id[0][0])[threadIdx.x] = -1;
  // TTV: the draw's timing tables in LDS when they fit (a wave's 64 list entries often span two
  // transits of different planets: a lookup per planet and round; from LDS it costs a tenth)
  constexpr int kTabMax = TTV ? 2048 : 1;
  __shared__ double s_tab[kTabMax];
  Ttv tl = ttv;
  int64_t row0 = draw * n_planet;
  if (TTV && n_planet * (2 * ttv.n_edge + 1) <= kTabMax) {
    const int ne = ttv.n_edge;
    const double* __restrict__ src_e = ttv.edges + row0 * ne;
    const double* __restrict__ src_s = ttv.shift + row0 * (ne + 1);
    for (int q = threadIdx.x; q < n_planet * ne; q += kBlock) s_tab[q] = src_e[q];
    for (int q = threadIdx.x; q < n_planet * (ne + 1); q += kBlock) s_tab[n_planet * ne + q] = src_s[q];
    tl.edges = s_tab;
    tl.shift = s_tab + n_planet * ne;
    row0 = 0;
    __syncthreads();
  }
  stage_constants(sh, params, ld, stencil_dt, stencil_w, n_sub, n_planet, draw, SECONDARY, nullptr,
                  TTV ? &tl : nullptr, row0 - draw * n_planet);
  const bool per_planet = flags & EXO_FLAG_PER_PLANET;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // this block works through the lists of `nsub` consecutive scan blocks of its draw
  const int bx0 = blockIdx.x * merge;
  const int nsub = (blocks_per_draw - bx0 < merge) ? blocks_per_draw - bx0 : merge;
  const int cap = tiles_per_block * 128;
  // segment order: all "inside" runs (scan block by scan block, wave by wave), then all "limb" runs
  const int nseg = 2 * kWaves * nsub;
  if ((int)threadIdx.x < nseg) {
    const int sgm = threadIdx.x;
    const int kind = sgm / (kWaves * nsub), rem = sgm - kind * (kWaves * nsub);
    s_pre[sgm + 1] = counts[2 * (((int64_t)draw * blocks_per_draw + bx0) * kWaves + rem) + kind];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc_n = 0;
    s_pre[0] = 0;
    for (int sgm = 1; sgm <= nseg; ++sgm) {
      acc_n += s_pre[sgm];
      s_pre[sgm] = acc_n;
    }
  }
  __syncthreads();
  const int total = s_pre[nseg];
  const int ng_draw = n_planet * kNG + 7;
  double* __restrict__ pout = GRAD ? partial + ((int64_t)draw * gridDim.x + blockIdx.x) * ng_draw : nullptr;

  // slots [0, kNG): this planet's parameters; [kNG, kNG + 6): limb darkening; kNG + 6: sum(gflux * flux)
  __shared__ double lds_acc[kNG + 7][kBlock];
  const GradAcc acc{GRAD ? &lds_acc[0][threadIdx.x] : nullptr};
  if (GRAD) {
#pragma unroll
    for (int s = 0; s < kNG + 7; ++s) lds_acc[s][threadIdx.x] = 0.0;
  }
  // limb-darkening coefficients, scalar registers as well
  double cld[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cld[k] = uniform((SECONDARY || k < 3) ? sh.c[k] : 0.0);
  // Several planets share one work list (a cadence is listed if ANY planet may overlap): a round
  // whose cadences are all away from planet p's conjunction windows is skipped for p on a wave
  // vote (the same five-operation test as the scan kernel's first stage), before any Kepler solve.
  const bool use_win = windows && n_planet > 1;
  double spanw = (flags & EXO_FLAG_WINDOW) ? 0.5 : 0.0;
  if (use_win && !(flags & EXO_FLAG_WINDOW))
    for (int k = 0; k < n_sub; ++k) spanw = fmax(spanw, fabs(sh.sdt[k]));
  // TTV: how far a sub-exposure can be from its cadence, in units of texp
  double reach = 0.0;
  if (TTV)
    for (int k = 0; k < n_sub; ++k) reach = fmax(reach, fabs(sh.sdt[k]));
  for (int p = 0; p < n_planet; ++p) {
    const PlanetS c(sh.pc[p]);
    const TtvRow row(tl, TTV ? row0 + p : 0);
    const TtvGrad tgrad{GRAD && TTV ? &lds_acc[0][threadIdx.x] : nullptr,
                        GRAD && TTV ? ttv.gshift + (draw * n_planet + p) * (int64_t)(ttv.n_edge + 1) : nullptr,
                        &s_bins};
    const double t_e0 = TTV ? uniform(sh.pc[p].te0) : 0.0, t_inv = TTV ? uniform(sh.pc[p].tinv) : 0.0;
    const int t_fin = TTV ? __builtin_amdgcn_readfirstlane(sh.pc[p].tfin) : 0;
    // the wave's current bin of this planet (scalar registers): list entries are consecutive
    // cadences, so a round usually stays in the bin of the one before
    double c_lo = __builtin_inf(), c_hi = -__builtin_inf(), c_sh = 0.0;
    int c_k = 0;
    double w_nrev = 0.0, w_c0 = 0.0, w_dmid = 0.0, w_h0 = 0.0, w_h1 = 0.0;
    if (use_win) {
      const double* wv = windows + kWin * (draw * n_planet + p);
      w_nrev = uniform(wv[0]); w_c0 = uniform(wv[1]); w_dmid = uniform(wv[2]);
      w_h0 = uniform(wv[3]); w_h1 = uniform(wv[4]);
    }
    if (GRAD && p > 0) {
#pragma unroll
      for (int s = 0; s < kNG; ++s) lds_acc[s][threadIdx.x] = 0.0;
    }
    // Two-deep software pipeline over the rounds: the list entry of round r + 2 and the cadence
    // data (t, texp, gflux) of round r + 1 are in flight while round r computes -- at two waves
    // per SIMD nothing else hides the two dependent loads (list -> t, gflux) of a round.
    auto list_index = [&](int jr, int64_t& base_cad) -> int {
      int sgm = 0;   // last segment whose start is <= jr (empty segments share a start: the search lands past them)
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) {
        const int q = sgm + step;
        if (q < nseg && jr >= s_pre[q]) sgm = q;
      }
      const int pos = jr - s_pre[sgm];
      const int kind = sgm / (kWaves * nsub), rem = sgm - kind * (kWaves * nsub);  // rem = sub * kWaves + wave
      const int64_t lbase = (((int64_t)draw * blocks_per_draw + bx0) * kWaves + rem) * (int64_t)cap;
      base_cad = (int64_t)(bx0 + rem / kWaves) * tiles_per_block * kTile;
      return list[lbase + (kind ? cap - 1 - pos : pos)];
    };
    struct Item { int64_t i; double tv, te, g; };
    auto load_item = [&](bool has_, int off_, int64_t base_) -> Item {
      Item it;
      it.i = has_ ? base_ + off_ : 0;
      it.tv = t[it.i];
      it.te = (n_texp == 0) ? 0.0 : (n_texp == 1 ? texp[0] : texp[it.i]);
      it.g = 0.0;
      if (GRAD && has_) it.g = per_planet ? gflux[(draw * n_cad + it.i) * n_planet + p] : gflux[draw * n_cad + it.i];
      return it;
    };
    int64_t base_n = 0, base_nn = 0;
    int off_n = 0, off_nn = 0;
    {
      const int j = threadIdx.x;
      if (j < total) off_n = list_index(j, base_n);
      if (j + kBlock < total) off_nn = list_index(j + kBlock, base_nn);
    }
    // (the gradient variant has no registers to spare for the data of a second round: it keeps
    // only the list entry one round ahead)
    constexpr bool kDeep = !GRAD;
    Item nxt{0, 0.0, 0.0, 0.0};
    if (kDeep) nxt = load_item((int)threadIdx.x < total, off_n, base_n);
    for (int j0 = 0; j0 < total; j0 += kBlock) {
      const int j = j0 + threadIdx.x;
      const bool has = j < total;
      Item cur;
      if (kDeep) {
        cur = nxt;
        // round r + 1's data (its list entry arrived a round ago), round r + 2's list entry
        if (j0 + kBlock < total) nxt = load_item(j + kBlock < total, off_nn, base_nn);
        if (j + 2 * kBlock < total) off_nn = list_index(j + 2 * kBlock, base_nn);
      } else {
        cur = load_item(has, off_n, base_n);
        off_n = off_nn; base_n = base_nn;
        if (j + 2 * kBlock < total) off_nn = list_index(j + 2 * kBlock, base_nn);
      }
      const int64_t i = cur.i;
      const double tv = cur.tv, te = cur.te;
      // TTV: the cadence's bin and shift; `mixed` = some sub-exposure may belong to another bin
      int kb = 0;
      double dsh = 0.0;
      bool mixed = false;
      if (TTV) {
        const double hw = fma(fabs(te) * reach, 1e-12, fabs(te) * reach);   // the product was rounded
        kb = c_k;
        dsh = c_sh;
        const bool inside = (tv - c_lo > hw) && (c_hi - tv > hw);
        const unsigned long long live = __ballot(has);
        if (__ballot(has && !inside) != 0) {
          double e_lo, e_hi;
          kb = row.locate(tv, t_e0, t_inv, t_fin, e_lo, e_hi);
          dsh = row.shift[kb];
          mixed = n_texp && (!(tv - e_lo > hw) || !(e_hi - tv > hw));
          // the last listed cadence of the wave leaves its bin behind
          const int last = 63 - __builtin_clzll(live);
          c_lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e_lo), last),
                                  __builtin_amdgcn_readlane(__double2loint(e_lo), last));
          c_hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e_hi), last),
                                  __builtin_amdgcn_readlane(__double2loint(e_hi), last));
          c_sh = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dsh), last),
                                  __builtin_amdgcn_readlane(__double2loint(dsh), last));
          c_k = __builtin_amdgcn_readlane(kb, last);
        }
      }
      if (use_win) {
        const double widen = fabs(te) * spanw * fabs(w_nrev);
        const bool near = has && ((mixed && !(flags & EXO_FLAG_WINDOW)) ||
                                  near_conjunction<SECONDARY>(tv - dsh, w_nrev, w_c0, w_dmid, w_h0 + widen, w_h1 + widen));
        if (!EXO_WAVE_ANY(near)) continue;   // the fill left this planet's flux at zero
      }
      const double g = cur.g;
      double f = 0.0;
      int kcur = kb;
      for (int k = 0; k < n_sub; ++k) {
        double tt = fma(te, sh.sdt[k], tv);
        if (TTV) {
          int ks = kb;
          double sh_k = dsh;
          if (mixed) {
            ks = row.bin(tt);
            sh_k = row.shift[ks];
          }
          if (GRAD && ks != kcur) {
            tgrad.flush_lane(kcur);
            kcur = ks;
          }
          tt -= sh_k;
        }
        const double gw = g * sh.sw[k];
        const double F = eval_sample<GRAD, SECONDARY>(tt, c, cld, gw, acc);
        f = fma(sh.sw[k], F, f);
        if (GRAD) acc.add(kNG + 6, gw * F);
      }
      if (GRAD && TTV) tgrad.flush_wave(kcur);
      if (flux && has) {
        if (per_planet) {
          flux[(draw * n_cad + i) * n_planet + p] = f;
        } else {
          double* dst = flux + draw * n_cad + i;
          *dst = (p == 0) ? f : (*dst + f);
        }
      }
    }
    if (GRAD && TTV) {
      tgrad.drain();
      // every sample's t_periastron term went through the bins; the planet's total is in G_PAD
      lds_acc[G_TP][threadIdx.x] = lds_acc[G_PAD][threadIdx.x];
      lds_acc[G_PAD][threadIdx.x] = 0.0;
    }
    if (GRAD) reduce_columns(lds_acc, sh.red, 0, kNG, pout + p * kNG);
  }
  if (GRAD) reduce_columns(lds_acc, sh.red, kNG, 7, pout + n_planet * kNG);
}

// Stage 2: one block per draw; thread s sums slot s over the blocks in order.
__global__ __launch_bounds__(kBlock) void transit_vjp_reduce_kernel(
    const double* __restrict__ partial, int nblk, int n_planet, bool secondary,
    double* __restrict__ gparams, double* __restrict__ gld, double* __restrict__ flux_dot) {
  const int64_t draw = blockIdx.x;
  const int ng_draw = n_planet * kNG + 7;
  const int s = threadIdx.x;
  for (int q = s; q < n_planet * EXO_NPAR; q += kBlock) {
    // record slots that carry no gradient (T0, PERIOD, the windows, the reserved ones) read 0
    const int p = q / EXO_NPAR, slot = q % EXO_NPAR;
    const bool carried = slot == EXO_P_N || slot == EXO_P_TP || slot == EXO_P_ECC || slot == EXO_P_COSW ||
                         slot == EXO_P_SINW || slot == EXO_P_COSI || slot == EXO_P_AOR || slot == EXO_P_ROR ||
                         slot == EXO_P_FRATIO || slot == EXO_P_SINI || slot == EXO_P_CLIGHT;
    if (!carried) gparams[(draw * n_planet + p) * EXO_NPAR + slot] = 0.0;
  }
  if (s >= ng_draw) return;
  const double* __restrict__ src = partial + draw * nblk * (int64_t)ng_draw + s;
  double v = 0.0;
  for (int b = 0; b < nblk; ++b) v += src[(int64_t)b * ng_draw];
  if (s < n_planet * kNG) {
    const int p = s / kNG, k = s % kNG;
    // compact slot -> EXO_P_* slot
    const int map[kNG] = {EXO_P_N, EXO_P_TP, EXO_P_ECC, EXO_P_COSW, EXO_P_SINW,
                          EXO_P_COSI, EXO_P_AOR, EXO_P_ROR, EXO_P_FRATIO, -1, EXO_P_SINI, EXO_P_CLIGHT};
    if (map[k] >= 0) gparams[(draw * n_planet + p) * EXO_NPAR + map[k]] = v;
  } else {
    const int k = s - n_planet * kNG;
    const int nld = secondary ? 6 : 3;
    if (k < nld) gld[draw * nld + k] = v;
    if (k == 6 && flux_dot) flux_dot[draw] = v;
  }
}

// ===========================================================================
// Run-enumeration path: sorted times, one exposure time (or none) for all cadences, no timing tables.
//
// Where a planet can overlap the disk is known in closed form (the conjunction windows of
// transit_window_kernel), the windows are periodic in mean anomaly, and t is sorted: so instead of
// testing every (draw, cadence) -- 1.5e8 phase tests per sweep of C2, 0.12 ms -- each window's run of
// cadences [lo, hi) is found by binary search in t (a few thousand searches per sweep).  The heavy
// kernel then works through the runs densely, in full fp64 (no fp32 pre-filter: a cadence in a
// window but off the disk costs one Kepler solve and returns 0), writes each cadence's flux to a
// compact per-(draw, planet) value array in run order, and -- dense output -- zero-fills its share of
// the flux array WHILE it computes (a few 1 KB non-temporal stores per wave and round: the store
// stream of the dense output hides under the fp64 work instead of preceding it); a last small kernel
// copies the runs' values to their cadences and sums the gradient partials.  With
// EXO_FLAG_SPARSE the flux array is never touched: the runs and the value array ARE the output.
//   t unsorted, a window that cannot be bounded, windows that overlap each other or more than
//   kRunMax windows in the series: that list becomes the single run [0, n_cad) (every cadence solved).
// ===========================================================================
struct Run {
  int32_t lo, a, b, hi;   // cadences [lo, a) and [b, hi): may touch the limb; [a, b): small disk wholly inside (a hint)
};
constexpr int kRunMax = 4096;     // windows per list
#ifndef EXO_RUN_SEG
#define EXO_RUN_SEG 256
#endif
constexpr int kSeg = EXO_RUN_SEG;  // runs of one list a heavy block holds in LDS at a time (a power of two)

struct RunLists {
  int32_t* nrun;     // [n_list]                 windows of list = (draw, planet, event)
  Run* runs;         // [n_list][r_max]
  int32_t* pre_in;   // [n_list][r_max + 1]      exclusive prefix sums of b - a
  int32_t* pre_all;  // [n_list][r_max + 1]      exclusive prefix sums of hi - lo (= position in the value array)
  int32_t* rbin;     // [n_list][r_max]          timing tables: the bin every cadence AND sub-exposure of the run falls
                     //                          in, or -1 (looked up sample by sample)
  double* grun;      // [n_list][r_max]          timing tables, reverse sweep: d(sum)/d(shift) collected run by run
  int r_max;
};

// exclusive prefix sums of the run lengths a list's wave left in s_len: a lane takes a contiguous share of the runs
__device__ __forceinline__ void enum_prefix(int (*s_len)[kRunMax + 1], int K, int lane, int32_t* __restrict__ pin,
                                            int32_t* __restrict__ pall, int32_t* __restrict__ nrun_dst) {
  const int per = (K + 63) / 64, k0 = lane * per, k1 = (k0 + per < K) ? k0 + per : K;
  int sum_in = 0, sum_all = 0;
  for (int k = k0; k < k1; ++k) { sum_in += s_len[0][k]; sum_all += s_len[1][k]; }
  int ex_in = sum_in, ex_all = sum_all;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int o_in = __shfl_up(ex_in, m, 64), o_all = __shfl_up(ex_all, m, 64);
    if (lane >= m) { ex_in += o_in; ex_all += o_all; }
  }
  int run_in = ex_in - sum_in, run_all = ex_all - sum_all;
  for (int k = k0; k < k1; ++k) {
    pin[k] = run_in; pall[k] = run_all;
    run_in += s_len[0][k]; run_all += s_len[1][k];
  }
  if (lane == 63) { pin[K] = ex_in; pall[K] = ex_all; *nrun_dst = K; }
}

// One wave per list (draw, planet, event: 0 = transits, 1 = occultations).
// FUSED (EXO_FLAG_SORTED_TIMES: the caller vouches for non-decreasing times, so nothing has to be checked before the
// searches): the wave works its record's conjunction windows out itself -- every group of eight lanes the same record, lane 0
// and lane 2 hold the result -- and the list of event 0 leaves them in `windows_out` for the sweep: no transit_window_kernel
// launch (each of these short kernels is ~5 us of dispatch and dependent memory round trips before its first useful cycle).
// PACK (with FUSED; exo_transit_flux_cols_vjp_f64): the wave is handed the constructor's COLUMNS and packs its record itself
// (exo_pack_core.hpp: lane 0; the list of event 0 writes it out for the sweep, the first list of a draw the limb-darkening
// coefficients too, on lane 1) -- no pack_kernel launch in front (C2: packing 5.9 + enumeration 10.2 us -> 14.5 us).
// (The packing VJP was folded into the sweep's last kernel as well -- the block that sums a draw's record cotangents taking them
// back to the columns -- measured, and removed: one thread's serial chain at the tail of every block cost the sweep 12.7 us at C2,
// the 1024-lane packing-VJP kernel it replaced costs 7.2.)
struct PackIn {
  exo_pack::ColsSrc src;
  uint32_t flags;          // pack flags
  int n_planet;
  double* params;          // out [n_draw][n_planet][EXO_NPAR]
  double* ld;              // out [n_draw][3 | 6]
};
template <bool FUSED, bool PACK = false>
__global__ __launch_bounds__(64) void transit_enum_kernel(const double* __restrict__ t, int64_t n_cad,
                                                          const double* __restrict__ texp, int64_t n_texp,
                                                          const double* __restrict__ stencil_dt, int n_sub, uint32_t flags,
                                                          const double* __restrict__ windows,
                                                          const int32_t* __restrict__ sorted, int n_sorted, int n_ev,
                                                          RunLists rl, const double* __restrict__ params = nullptr,
                                                          double* __restrict__ windows_out = nullptr, PackIn pk = PackIn{}) {
  static_assert(!PACK || FUSED, "packing rides on the fused windows + enumeration launch");
  __shared__ int s_len[2][kRunMax + 1];
  __shared__ double s_rec[PACK ? EXO_NPAR : 1];
  const int64_t list = blockIdx.x, rec = list / n_ev;
  const int ev = (int)(list - rec * n_ev), lane = threadIdx.x;
  if (PACK) {
    const int64_t draw = rec / pk.n_planet;
    const int planet = (int)(rec - draw * pk.n_planet);
    if (lane == 0) {
      double o[EXO_NPAR];
      exo_pack::pack_record(pk.src, rec, draw, planet, pk.flags, o);
#pragma unroll
      for (int k = 0; k < EXO_NPAR; ++k) s_rec[k] = o[k];
      if (ev == 0) {
#pragma unroll
        for (int k = 0; k < EXO_NPAR; ++k) pk.params[rec * EXO_NPAR + k] = o[k];
      }
    }
    if (lane == 1 && ev == 0 && planet == 0)
      exo_pack::pack_ld(pk.src, draw, pk.flags, pk.ld + draw * ((pk.flags & EXO_FLAG_SECONDARY) ? 6 : 3));
    __syncthreads();
  }
  double wv[kWin];
  if (FUSED) {
    double w[kWin] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    window_lanes(PACK ? s_rec : params + rec * EXO_NPAR, flags, lane & 7, w);
    const bool every = flags & EXO_FLAG_WINDOW;   // (every lane holds all seven)
#pragma unroll
    for (int q = 0; q < kWin; ++q) wv[q] = __shfl(w[q], (!every && (q == 4 || q == 6)) ? 2 : 0, 64);
    if (ev == 0 && lane == 0) {
#pragma unroll
      for (int q = 0; q < kWin; ++q) windows_out[kWin * rec + q] = wv[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < kWin; ++q) wv[q] = windows[kWin * rec + q];
  }
  const double nrev = wv[0], c0 = wv[1], dmid = wv[2];
  // the windows are widened by the half-span of the exposure stencil; the reference widens its
  // contact windows by texp / 2 whatever the stencil (keplerian.py:765-769)
  double span = (flags & EXO_FLAG_WINDOW) ? 0.5 : 0.0;
  if (!(flags & EXO_FLAG_WINDOW) && stencil_dt)
    for (int k = 0; k < n_sub; ++k) span = fmax(span, fabs(stencil_dt[k]));
  const double te = n_texp ? texp[0] : 0.0;
  const double widen = fabs(te) * span * fabs(nrev);
  const double h0 = wv[3] + widen, h1 = wv[4] + widen;
  bool srt = true;
  if (!FUSED) {
    for (int i = lane; i < n_sorted; i += 64) srt = srt && (sorted[i] != 0);
    srt = __all(srt);
  } else {
    // the caller's word covers the order of NEIGHBOURING cadences; the series as a whole is looked at here, coarsely: 65
    // evenly spaced cadences must ascend (a NaN fails), else the list is "every cadence".  What this catches is a time
    // buffer refilled with another, unordered series under a flag that was baked into a captured launch; two swapped
    // neighbours it cannot see (that is what the unflagged sweep's own check is for).  Two independent loads per lane.
    const int64_t i0 = (n_cad - 1) * lane / 64, i1 = (n_cad - 1) * (lane + 1) / 64;
    srt = __all(t[i0] <= t[i1]);
  }
  // the list degenerates to "every cadence" unless its windows are bounded, periodic in t and disjoint
  // (the decision is the same for both events of a planet: it only uses what they share)
  const double x_first = fma(t[0], nrev, c0), x_last = fma(t[n_cad - 1], nrev, c0);
  bool full = !srt || !(nrev > 0.0) || !(x_first == x_first) || !(x_last == x_last) || !(fabs(x_first) < 1e15) ||
              !(fabs(x_last) < 1e15) || !(h0 < 0.5);
  if (n_ev == 2) {
    const double sep = fabs(frac_rev(dmid));   // transit and occultation centres, in revolutions
    full = full || !(h1 < 0.5) || !(h0 + h1 < sep);
  }
  double kmin[2] = {0.0, 0.0}, kcnt[2] = {0.0, 0.0};
  if (!full) {
    for (int e = 0; e < n_ev; ++e) {
      const double off = e ? dmid : 0.0, h = e ? h1 : h0;
      kmin[e] = ceil((x_first + off) - h);
      kcnt[e] = floor((x_last + off) + h) - kmin[e] + 1.0;
      full = full || (kcnt[e] > (double)rl.r_max);
    }
  }
  Run* __restrict__ runs = rl.runs + list * rl.r_max;
  int32_t* __restrict__ pin = rl.pre_in + list * (rl.r_max + 1);
  int32_t* __restrict__ pall = rl.pre_all + list * (rl.r_max + 1);
  int K;
  if (full) {
    // every cadence, as pieces of >= 1024 (the heavy blocks of a draw share a list run by run)
    int64_t piece = (n_cad + rl.r_max - 1) / rl.r_max;
    piece = piece < 1024 ? 1024 : piece;
    K = ev == 0 ? (int)((n_cad + piece - 1) / piece) : 0;
    for (int k = lane; k < K; k += 64) {
      const int64_t lo = k * piece, hi = (lo + piece < n_cad) ? lo + piece : n_cad;
      runs[k] = Run{(int32_t)lo, (int32_t)lo, (int32_t)lo, (int32_t)hi};
      s_len[0][k] = 0;
      s_len[1][k] = (int)(hi - lo);
    }
  } else {
    K = kcnt[ev] > 0.0 ? (int)kcnt[ev] : 0;
    const double off = ev ? dmid : 0.0, h = ev ? h1 : h0, hin = wv[5 + ev];
    // first i in [lo, hi) with x_i >= thr (strict: > thr).  Series are nearly always evenly sampled:
    // the position guessed from the mean sampling rate is confirmed by its two neighbours (two
    // independent loads instead of a chain of log2(n) dependent ones); anything else is searched for.
    const double x_rate = (x_last - x_first) / (double)(n_cad > 1 ? n_cad - 1 : 1);
    auto first_not = [&](double thr, bool strict, int lo, int hi) {
      if (x_rate > 0.0 && hi > lo) {
        const double gq = ceil((thr - off - x_first) / x_rate);
        int g = gq < (double)lo ? lo : (gq > (double)hi ? hi : (int)gq);
        const double xa = g > lo ? fma(t[g - 1], nrev, c0) + off : 0.0, xb = g < hi ? fma(t[g], nrev, c0) + off : 0.0;
        const bool left_before = g == lo || (strict ? (xa <= thr) : (xa < thr));
        const bool here_not = g == hi || !(strict ? (xb <= thr) : (xb < thr));
        if (left_before && here_not) return g;
      }
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double xi = fma(t[mid], nrev, c0) + off;
        const bool before = strict ? (xi <= thr) : (xi < thr);
        lo = before ? mid + 1 : lo;
        hi = before ? hi : mid;
      }
      return lo;
    };
    for (int k = lane; k < K; k += 64) {
      const double kc = kmin[ev] + (double)k;
      Run r;
      r.lo = first_not(kc - h, false, 0, (int)n_cad);
      r.hi = first_not(kc + h, true, r.lo, (int)n_cad);
      if (hin > 0.0) {
        r.a = first_not(kc - hin, false, r.lo, r.hi);
        r.b = first_not(kc + hin, true, r.a, r.hi);
      } else {
        r.a = r.b = r.lo;
      }
      runs[k] = r;
      s_len[0][k] = r.b - r.a;
      s_len[1][k] = r.hi - r.lo;
    }
  }
  __syncthreads();
  enum_prefix(s_len, K, lane, pin, pall, rl.nrun + list);
}

// The same with timing tables (one list per (draw, planet): transits only).  Within a timing bin the warp is a plain
// shift, so the windows of bin k are periodic in t - shift[k]: every bin's windows are enumerated on their own.  A
// list is TRUSTED when each of its windows, widened by the exposure's reach, lies strictly inside its bin -- then every
// cadence of a run and every one of its sub-exposures shares the run's bin (rbin), no sample needs a table lookup and
// d/d(shift) can be collected run by run.  Anything else (a transit across a bin edge, more bins or windows than the
// tables hold, unsorted times) degenerates to "every cadence", each sample looking its own bin up (rbin = -1).
__global__ __launch_bounds__(64) void transit_enum_ttv_kernel(const double* __restrict__ t, int64_t n_cad,
                                                              const double* __restrict__ texp, int64_t n_texp,
                                                              const double* __restrict__ stencil_dt, int n_sub,
                                                              uint32_t flags, const double* __restrict__ windows,
                                                              const int32_t* __restrict__ sorted, int n_sorted,
                                                              RunLists rl, Ttv ttv) {
  __shared__ int s_len[2][kRunMax + 1];
  __shared__ int s_first[kRunMax + 2], s_mlo[kRunMax + 1];
  const int64_t list = blockIdx.x;
  const int lane = threadIdx.x;
  const double* wv = windows + kWin * list;
  const double nrev = wv[0], c0 = wv[1], hin = wv[5];
  double reach = 0.0;
  if (stencil_dt)
    for (int k = 0; k < n_sub; ++k) reach = fmax(reach, fabs(stencil_dt[k]));
  const double span = (flags & EXO_FLAG_WINDOW) ? 0.5 : reach;
  const double te = n_texp ? texp[0] : 0.0;
  const double h0 = wv[3] + fabs(te) * span * fabs(nrev);
  bool srt = true;
  for (int i = lane; i < n_sorted; i += 64) srt = srt && (sorted[i] != 0);
  srt = __all(srt);
  const TtvRow row(ttv, list);
  const int nfin = row.bin(__builtin_inf());   // (the padding is +inf)
  const double t_first = t[0], t_last = t[n_cad - 1];
  const double inf = __builtin_inf();
  bool full = !srt || !(nrev > 0.0) || !(h0 < 0.5) || (nfin + 1 > kRunMax) || !(t_first == t_first) ||
              !(t_last == t_last) || !(fabs(t_first) < inf) || !(fabs(t_last) < inf);
  const double hw_t = h0 / nrev, r_t = n_texp ? fabs(te) * reach : 0.0;
  int K = 0;
  if (!full) {
    bool bad = false;
    int base = 0;
    for (int k0 = 0; k0 <= nfin; k0 += 64) {
      const int k = k0 + lane;
      int cnt = 0, mlo = 0;
      if (k <= nfin) {
        const double lo_t = k > 0 ? row.edges[k - 1] : -inf, hi_t = k < nfin ? row.edges[k] : inf;   // the bin: (lo_t, hi_t]
        const double sh = row.shift[k];
        const double lo_c = fmax(lo_t, t_first), hi_c = fmin(hi_t, t_last);
        if (lo_c <= hi_c) {
          const double x_lo = fma(lo_c - sh, nrev, c0), x_hi = fma(hi_c - sh, nrev, c0);
          if (!(fabs(x_lo) < 1e9) || !(fabs(x_hi) < 1e9)) {
            bad = true;
          } else {
            const double a = ceil(x_lo - h0), b = floor(x_hi + h0);
            if (b >= a) {
              const double n = b - a + 1.0;
              if (n > (double)rl.r_max) {
                bad = true;
              } else {
                cnt = (int)n;
                mlo = (int)a;
              }
              // the bin's first and last window, the exposure's reach included, strictly inside it
              const double tc_a = (a - c0) / nrev + sh, tc_b = (b - c0) / nrev + sh;
              const double slack = 1e-9 * (fabs(tc_a) + fabs(tc_b) + 1.0);
              if (!(tc_a - hw_t - r_t - slack > lo_t) || !(tc_b + hw_t + r_t + slack < hi_t)) bad = true;
            }
          }
        } else if (!(lo_c == lo_c) || !(hi_c == hi_c)) {
          bad = true;
        }
      }
      int ex = cnt;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_up(ex, m, 64);
        if (lane >= m) ex += o;
      }
      if (k <= nfin) { s_first[k] = base + ex - cnt; s_mlo[k] = mlo; }
      base += __shfl(ex, 63, 64);
      if (base > rl.r_max) bad = true;
    }
    K = base;
    full = __any(bad) || K > rl.r_max;
  }
  Run* __restrict__ runs = rl.runs + list * rl.r_max;
  int32_t* __restrict__ rbin = rl.rbin + list * rl.r_max;
  int32_t* __restrict__ pin = rl.pre_in + list * (rl.r_max + 1);
  int32_t* __restrict__ pall = rl.pre_all + list * (rl.r_max + 1);
  __syncthreads();
  if (full) {
    int64_t piece = (n_cad + rl.r_max - 1) / rl.r_max;
    piece = piece < 1024 ? 1024 : piece;
    K = (int)((n_cad + piece - 1) / piece);
    for (int k = lane; k < K; k += 64) {
      const int64_t lo = k * piece, hi = (lo + piece < n_cad) ? lo + piece : n_cad;
      runs[k] = Run{(int32_t)lo, (int32_t)lo, (int32_t)lo, (int32_t)hi};
      rbin[k] = -1;
      s_len[0][k] = 0;
      s_len[1][k] = (int)(hi - lo);
    }
  } else {
    const double t_rate = (t_last - t_first) / (double)(n_cad > 1 ? n_cad - 1 : 1);
    for (int r = lane; r < K; r += 64) {
      int lo_k = 0, hi_k = nfin + 1;     // the run's bin: the last one whose first run is <= r
      while (hi_k - lo_k > 1) {
        const int mid = (lo_k + hi_k) >> 1;
        if (s_first[mid] <= r) lo_k = mid; else hi_k = mid;
      }
      const int k = lo_k;
      const double kc = (double)(s_mlo[k] + (r - s_first[k]));
      const double sh = row.shift[k], off = -sh * nrev;
      // first i in [lo, hi) whose warped phase is >= thr (strict: > thr), guessed from the mean sampling rate first
      auto first_not = [&](double thr, bool strict, int lo, int hi) {
        if (t_rate > 0.0 && hi > lo) {
          const double gq = ceil(((thr - c0) / nrev + sh - t_first) / t_rate);
          const int g = gq < (double)lo ? lo : (gq > (double)hi ? hi : (int)gq);
          const double xa = g > lo ? fma(t[g - 1], nrev, c0) + off : 0.0, xb = g < hi ? fma(t[g], nrev, c0) + off : 0.0;
          const bool left_before = g == lo || (strict ? (xa <= thr) : (xa < thr));
          const bool here_not = g == hi || !(strict ? (xb <= thr) : (xb < thr));
          if (left_before && here_not) return g;
        }
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const double xi = fma(t[mid], nrev, c0) + off;
          const bool before = strict ? (xi <= thr) : (xi < thr);
          lo = before ? mid + 1 : lo;
          hi = before ? hi : mid;
        }
        return lo;
      };
      Run q;
      q.lo = first_not(kc - h0, false, 0, (int)n_cad);
      q.hi = first_not(kc + h0, true, q.lo, (int)n_cad);
      if (hin > 0.0) {
        q.a = first_not(kc - hin, false, q.lo, q.hi);
        q.b = first_not(kc + hin, true, q.a, q.hi);
      } else {
        q.a = q.b = q.lo;
      }
      runs[r] = q;
      rbin[r] = k;
      s_len[0][r] = q.b - q.a;
      s_len[1][r] = q.hi - q.lo;
    }
  }
  __syncthreads();
  enum_prefix(s_len, K, lane, pin, pall, rl.nrun + list);
}

// this wave's share of a block's zero-fill: 1 KB pieces (64 lanes x 16 B, non-temporal), a few per round.
// Everything that steers the stream is WAVE-UNIFORM and lives in scalar registers -- the piece pointer, the count of
// pieces left -- and a store is `global_store_dwordx4 v_lane_offset, v_zero, s[piece]` with no vector arithmetic at
// all (round 2 carried the cursor per lane: 64-bit vector adds, two vector compares and four moves of the zero per
// store, ~200 vector instructions per round of a 1024-draw sweep -- a sixth of the kernel's VALU work).
struct FillCursor {
  typedef double v2d __attribute__((ext_vector_type(2)));
  char* base;      // the block's 16-B aligned share (block-uniform: derived from kernel arguments and block indices)
  int64_t off;     // byte offset of this wave's next full piece (wave-uniform)
  int left;        // full pieces this wave still owes (wave-uniform)
  uint32_t loff;   // lane * 16
  v2d zero;
  __device__ __forceinline__ FillCursor(double* dst, int64_t n) : base(nullptr), off(0), left(0), loff((threadIdx.x & 63) * 16) {
    zero = v2d{0.0, 0.0};
    asm volatile("" : "+v"(zero));   // (an opaque value: kept in four registers, not re-materialised before every store)
    if (!dst || n <= 0) return;
    const int64_t head = (reinterpret_cast<uintptr_t>(dst) & 8) ? 1 : 0;
    if (threadIdx.x == 0 && head) dst[0] = 0.0;
    if (threadIdx.x == 0 && ((n - head) & 1)) dst[n - 1] = 0.0;
    v2d* q2 = reinterpret_cast<v2d*>(dst + head);
    const int64_t n2 = (n - head) >> 1;            // 16-B units
    const int64_t nfull = n2 >> 6;                 // full 1-KB pieces; the partial one goes now
    const int rem = (int)(n2 & 63);
    if ((int)threadIdx.x < rem) __builtin_nontemporal_store(zero, q2 + (nfull << 6) + threadIdx.x);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    left = nfull > w ? (int)((nfull - w + kWaves - 1) / kWaves) : 0;
    left = __builtin_amdgcn_readfirstlane(left);
    base = reinterpret_cast<char*>(q2);
    off = (int64_t)w * 1024;
  }
  __device__ __forceinline__ int pieces_left() const { return left; }
  __device__ __forceinline__ void issue(int count) {   // `count`: wave-uniform
    const int k = count < left ? count : left;
    for (int s = 0; s < k; ++s) {
      __builtin_nontemporal_store(zero, reinterpret_cast<v2d*>(base + off + loff));
      off += kWaves * 1024;
    }
    left -= k;
  }
};

// Last kernel of a sweep on the run-enumeration path, one block per draw: (GRAD) block partials ->
// gparams, gld, sum(gflux * flux), in block order; (dense output) the runs' values to their cadences,
// planet by planet (summed flux: a later planet adds to what the earlier ones left).
// (1024 threads per block for batches of at most 256 draws: the scatter of a draw's values is one block's work, and
// with few draws the loads it keeps in flight are what bounds it -- C4 at 64 draws: 34 -> 10 us)
// (also the tail of transit_runs_kernel when a draw is one block's work -- no restrict on what that kernel wrote)
__device__ __forceinline__ void finish_draw(
    int64_t draw, const double* partial, int nblk, int n_planet, bool secondary, double* __restrict__ gparams,
    double* __restrict__ gld, double* __restrict__ flux_dot, int64_t n_cad, uint32_t flags, int n_ev, const RunLists& rl,
    const double* vals, const int32_t* vcad, double* flux,
    const double* __restrict__ chi2_part, int n_chi2_part, double* __restrict__ chi2_out, const Ttv& ttv,
    int64_t cm_draws = 0) {   // cm_draws: 0, or n_draw -- the summed flux is cadence-major, [n_cad][n_draw]
  if (ttv.gshift) {
    // timing tables, lists whose runs carry their bins: the runs' sums to their bins, in run order (the bins of a
    // list's runs ascend); the samples of any other list added to gshift themselves
    for (int p = 0; p < n_planet; ++p) {
      const int64_t list = draw * n_planet + p;
      const int K = rl.nrun[list];
      const int32_t* __restrict__ rbin = rl.rbin + list * rl.r_max;
      if (K == 0 || rbin[0] < 0) continue;
      const double* grun = rl.grun + list * rl.r_max;
      double* __restrict__ dst = ttv.gshift + list * (int64_t)(ttv.n_edge + 1);
      for (int k = threadIdx.x; k <= ttv.n_edge; k += (int)blockDim.x) {
        int lo = 0, hi = K;   // first run of a bin >= k
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rbin[mid] < k) lo = mid + 1; else hi = mid;
        }
        double v = 0.0;
        for (int r = lo; r < K && rbin[r] == k; ++r) v += grun[r];
        dst[k] = v;
      }
    }
  }
  if (chi2_out && threadIdx.x == blockDim.x - 1) {   // block partials of transit_residual_kernel, in block order
    double v = 0.0;
    for (int b = 0; b < n_chi2_part; ++b) v += chi2_part[draw * n_chi2_part + b];
    chi2_out[draw] = v;
  }
  const int ng_draw = n_planet * kNG + 7;
  const int s = threadIdx.x;
  if (partial) {
    for (int q = s; q < n_planet * EXO_NPAR; q += (int)blockDim.x) {
      // record slots that carry no gradient (T0, PERIOD, the windows, the reserved ones) read 0
      const int p = q / EXO_NPAR, slot = q % EXO_NPAR;
      const bool carried = slot == EXO_P_N || slot == EXO_P_TP || slot == EXO_P_ECC || slot == EXO_P_COSW ||
                           slot == EXO_P_SINW || slot == EXO_P_COSI || slot == EXO_P_AOR || slot == EXO_P_ROR ||
                           slot == EXO_P_FRATIO || slot == EXO_P_SINI || slot == EXO_P_CLIGHT;
      if (!carried) gparams[(draw * n_planet + p) * EXO_NPAR + slot] = 0.0;
    }
    if (s < ng_draw) {
      const double* src = partial + draw * nblk * (int64_t)ng_draw + s;
      double v = 0.0;
      for (int b = 0; b < nblk; ++b) v += src[(int64_t)b * ng_draw];
      if (s < n_planet * kNG) {
        const int p = s / kNG, k = s % kNG;
        const int map[kNG] = {EXO_P_N, EXO_P_TP, EXO_P_ECC, EXO_P_COSW, EXO_P_SINW,
                              EXO_P_COSI, EXO_P_AOR, EXO_P_ROR, EXO_P_FRATIO, -1, EXO_P_SINI, EXO_P_CLIGHT};
        if (map[k] >= 0) gparams[(draw * n_planet + p) * EXO_NPAR + map[k]] = v;
      } else {
        const int k = s - n_planet * kNG;
        const int nld = secondary ? 6 : 3;
        if (k < nld) gld[draw * nld + k] = v;
        if (k == 6 && flux_dot) flux_dot[draw] = v;
      }
    }
  }
  if (!flux || !vals) return;
  // a thread per value: value and cadence arrays are read contiguously, four loads in flight per thread.  Summed flux of
  // several planets: planet 0 stores, every later planet ADDS with the hardware's fp64 atomic (no load of the target:
  // read-modify-write in the thread made each planet a load and a store round trip, C4 at 64 draws 15.8 us) behind a
  // block barrier -- planets in order, so the sums stay bit-reproducible (a planet's transits and occultations never
  // share a cadence) -- and the next batch of values is loaded before the current one is written.
  const bool per_planet = flags & EXO_FLAG_PER_PLANET;
  const int nthr = (int)blockDim.x;
  // the planets' value counts, all at once (two dependent loads each: one after the other they were the kernel)
  __shared__ int s_total[EXO_MAX_PLANETS];
  if ((int)threadIdx.x < n_planet) {
    int total = 0;
    for (int ev = 0; ev < n_ev; ++ev) {
      const int64_t list = (draw * n_planet + threadIdx.x) * n_ev + ev;
      total += rl.pre_all[list * (rl.r_max + 1) + rl.nrun[list]];
    }
    s_total[threadIdx.x] = total;
  }
  __syncthreads();
  auto total_of = [&](int p) { return s_total[p]; };
  auto load = [&](int p, int cb, int total, double* v, int* i) {
    const int64_t vbase = (draw * n_planet + p) * n_cad;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = cb + (int)threadIdx.x + u * nthr;
      v[u] = e < total ? vals[vbase + e] : 0.0;
      i[u] = e < total ? vcad[vbase + e] : -1;
    }
  };
  // batches (planet, first value) in order; `advance` steps to the next non-empty one
  int pl = 0, cb = -4 * nthr, total = total_of(0);
  auto advance = [&](int& p, int& c, int& tot) {
    c += 4 * nthr;
    while (p < n_planet && c >= tot) {
      ++p; c = 0;
      tot = p < n_planet ? total_of(p) : 0;
    }
  };
  advance(pl, cb, total);
  if (pl >= n_planet) return;
  double v[4];
  int i[4];
  load(pl, cb, total, v, i);
  for (;;) {
    int np = pl, ncb = cb, ntotal = total;
    advance(np, ncb, ntotal);
    const bool more = np < n_planet;
    double nv[4] = {0.0, 0.0, 0.0, 0.0};
    int ni[4] = {-1, -1, -1, -1};
    if (more) load(np, ncb, ntotal, nv, ni);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i[u] < 0) continue;
      if (per_planet) {
        flux[(draw * n_cad + i[u]) * n_planet + pl] = v[u];
      } else {
        double* dst = cm_draws ? flux + (int64_t)i[u] * cm_draws + draw : flux + draw * n_cad + i[u];
        if (pl == 0) *dst = v[u]; else unsafeAtomicAdd(dst, v[u]);
      }
    }
    if (!more) break;
    if (!per_planet && np != pl) __syncthreads();   // planets in order
    pl = np; cb = ncb; total = ntotal;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = nv[u]; i[u] = ni[u]; }
  }
}

__global__ __launch_bounds__(1024) void transit_finish_kernel(
    const double* __restrict__ partial, int nblk, int n_planet, bool secondary, double* __restrict__ gparams,
    double* __restrict__ gld, double* __restrict__ flux_dot, int64_t n_cad, uint32_t flags, int n_ev, RunLists rl,
    const double* __restrict__ vals, const int32_t* __restrict__ vcad, double* __restrict__ flux,
    const double* __restrict__ chi2_part, int n_chi2_part, double* __restrict__ chi2_out,
    Ttv ttv = Ttv{nullptr, nullptr, nullptr, 0}) {
  finish_draw(blockIdx.x, partial, nblk, n_planet, secondary, gparams, gld, flux_dot, n_cad, flags, n_ev, rl, vals, vcad, flux,
              chi2_part, n_chi2_part, chi2_out, ttv,
              ((flags & EXO_FLAG_CADENCE_MAJOR) && !(flags & EXO_FLAG_PER_PLANET)) ? (int64_t)gridDim.x : 0);
}

struct FinishArgs {
  double* gparams;
  double* gld;
  double* flux_dot;
  int fold;        // the runs kernel finishes its draws itself: no transit_finish_kernel launch
  int32_t* done;   // [n_draw] blocks of the draw that are through (zeroed by transit_window_kernel)
};

// CHI2 (one planet, one sample per cadence): gflux is the observed series [n_cad], gsparse its weights ([1] or [n_cad],
// `chi2_nw` says which); the "sum(gflux * flux)" slot of the partials carries sum w ((F - obs)^2 - obs^2) instead.
// TTV (one event per planet): `ttv` holds the timing tables; a trusted list's runs carry their bin (rl.rbin), its
// samples are shifted by the run's shift and d/d(shift) is summed run by run (wave partials in LDS, combined in a fixed
// order into rl.grun: bit-reproducible); the samples of any other list look their bins up and add to gshift atomically.
// Occupancy: THREE blocks per CU (three waves per SIMD: 168 registers, <= 53 KB of LDS per block).  The fp64 work of
// a sample is one long dependency chain (8 cycles per dependent operation against 4 of issue); a third wave per SIMD is
// worth ~1.2x of two.  It fits because (a) this translation unit is built with -mllvm -disable-machine-licm
// (__graft_entry__.py): hoisted out of the cadence loop, the ~60 fp64 constants of the polynomials sat in ~110 vector
// registers for the whole kernel (256 registers + scratch, two waves); re-materialised where they are used the kernel
// needs 168; (b) kSeg = 256 runs per batch keeps the LDS under a third of the CU's.  Variants whose LDS does not fit
// three blocks (timing tables) get the registers of two waves from the compiler; so do the light-delay variants (two
// Kepler solves alive at once: 250 B of scratch at 168 registers, measured slower than two waves without).
#ifndef EXO_RUNS_MIN_WAVES
#define EXO_RUNS_MIN_WAVES 3
#endif
// JAC (round 4; GRAD with a UNIT cotangent, value sweep of exo_transit_flux_fwd_jac_f64): a light curve that is the mean of a GP
// is swept before its cotangent exists, and used to be swept AGAIN for the gradient once it did.  The cotangent enters
// linearly -- gparams = sum over cadences of g x dF / dparams -- so this sweep leaves, next to every solved cadence's value, its
// sixteen derivatives (ten record slots, six limb-darkening coefficients: kJac doubles at `partial`, which is the Jacobian
// array here) and the second sweep becomes a contraction (transit_jac_vjp_kernel).  With an exposure stencil a cadence is
// n_sub Kepler solves and still one row of sixteen: C5 (7 sub-exposures) 238 us -> a few.
constexpr int kJac = 16;
__device__ __forceinline__ int jac_slot(int s) {   // LDS gradient column -> position in the row (-1: not kept)
  return s < G_PAD ? s : (s == G_SINI ? 9 : (s >= kNG && s < kNG + 6 ? 10 + (s - kNG) : -1));
}
template <bool GRAD, bool SECONDARY, bool LDELAY = false, bool CHI2 = false, bool TTV = false, bool JAC = false>
__global__ __launch_bounds__(kBlock, LDELAY ? 2 : EXO_RUNS_MIN_WAVES) void transit_runs_kernel(
    const double* __restrict__ t, int64_t n_cad, const double* __restrict__ texp, int64_t n_texp,
    const double* __restrict__ stencil_dt, const double* __restrict__ stencil_w, int n_sub,
    const double* __restrict__ params, const double* __restrict__ ld, int n_planet, uint32_t flags, int n_ev, RunLists rl,
    const double* __restrict__ gflux, const double* __restrict__ gsparse, double* __restrict__ vals,
    int32_t* __restrict__ vcad, double* __restrict__ fill, double* __restrict__ partial, int64_t chi2_nw = 0,
    Ttv ttv = Ttv{nullptr, nullptr, nullptr, 0}, FinishArgs fin = FinishArgs{nullptr, nullptr, nullptr, 0, nullptr}) {
  __shared__ Shared sh;
  __shared__ Run s_run[kSeg];
  __shared__ int2 s_pre[kSeg + 1];   // positions of a batch's runs among its "inside" items (.x) and its limb items (.y)
  __shared__ int s_all[kSeg + 1];    // position of a run's first cadence in the value array
  __shared__ int s_bin[TTV ? kSeg : 1];
  __shared__ double s_shift[TTV ? kSeg : 1];
  __shared__ double s_grun[(TTV && GRAD) ? kWaves : 1][(TTV && GRAD) ? kSeg : 1];
  __shared__ int s_rounds[2 * EXO_MAX_PLANETS];
  __shared__ double lds_acc[kNG + 7][kBlock];
  const int64_t draw = blockIdx.y;
  const int hb = gridDim.x, bx = blockIdx.x;
  stage_constants(sh, params, ld, stencil_dt, stencil_w, n_sub, n_planet, draw, SECONDARY);
  const bool per_planet = flags & EXO_FLAG_PER_PLANET;
  const int n_lists = n_planet * n_ev;
  // this block's slice of every list, and the rounds of 256 cadences it will take in all
  auto slice = [&](int K, int& k0, int& k1) {
    k0 = (int)((int64_t)K * bx / hb);
    k1 = (int)((int64_t)K * (bx + 1) / hb);
  };
  if ((int)threadIdx.x < n_lists) {
    const int64_t list = draw * n_lists + threadIdx.x;
    int k0, k1;
    slice(rl.nrun[list], k0, k1);
    const int32_t* pall = rl.pre_all + list * (rl.r_max + 1);
    int rounds = 0;
    for (int kb = k0; kb < k1; kb += kSeg) {
      const int ke = (kb + kSeg < k1) ? kb + kSeg : k1;
      rounds += (pall[ke] - pall[kb] + kBlock - 1) / kBlock;
    }
    s_rounds[threadIdx.x] = rounds;
  }
  __syncthreads();
  int total_rounds = 0;
  for (int l = 0; l < n_lists; ++l) total_rounds += s_rounds[l];
  // dense output: this block zeroes its share of the draw's flux, a few pieces per round
  const int64_t npl = per_planet ? n_planet : 1, n_fill = n_cad * npl;
  const int64_t f0 = n_fill * bx / hb, f1 = n_fill * (bx + 1) / hb;
  FillCursor fc(fill ? fill + draw * n_fill + f0 : nullptr, f1 - f0);
  const int per_round =
      __builtin_amdgcn_readfirstlane(total_rounds > 0 ? (fc.pieces_left() + total_rounds - 1) / total_rounds : 0);   // (wave-uniform)

  static_assert(!JAC || (GRAD && !LDELAY && !CHI2 && !TTV), "the Jacobian sweep is the plain value + gradient evaluation");
  const int ng_draw = n_planet * kNG + 7;
  double* __restrict__ pout = (GRAD && !JAC) ? partial + ((int64_t)draw * hb + bx) * ng_draw : nullptr;
  double* __restrict__ jac = JAC ? partial : nullptr;
  const GradAcc acc{GRAD ? &lds_acc[0][threadIdx.x] : nullptr};
  if (GRAD) {
#pragma unroll
    for (int s = 0; s < kNG + 7; ++s) lds_acc[s][threadIdx.x] = 0.0;
  }
  double cld[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cld[k] = uniform((SECONDARY || k < 3) ? sh.c[k] : 0.0);
  const double te = (n_texp == 0) ? 0.0 : texp[0];
  const double sdt0 = uniform(sh.sdt[0]), sw0 = uniform(sh.sw[0]);
  for (int p = 0; p < n_planet; ++p) {
    const PlanetS c(sh.pc[p]);
    if (GRAD && p > 0) {
#pragma unroll
      for (int s = 0; s < kNG; ++s) lds_acc[s][threadIdx.x] = 0.0;
    }
    int64_t vbase = (draw * n_planet + p) * n_cad;   // the planet's values: transits first, occultations behind them
    const TtvRow row(ttv, TTV ? draw * n_planet + p : 0);
    double* __restrict__ grow = (TTV && GRAD) ? ttv.gshift + (draw * n_planet + p) * (int64_t)(ttv.n_edge + 1) : nullptr;
    const TtvGrad tgrad{(GRAD && TTV) ? &lds_acc[0][threadIdx.x] : nullptr, grow, nullptr};
    for (int ev = 0; ev < n_ev; ++ev) {
      const int64_t list = (draw * n_planet + p) * n_ev + ev;
      const int K = rl.nrun[list];
      const Run* __restrict__ runs = rl.runs + list * rl.r_max;
      const int32_t* __restrict__ pin = rl.pre_in + list * (rl.r_max + 1);
      const int32_t* __restrict__ pall = rl.pre_all + list * (rl.r_max + 1);
      int k0, k1;
      slice(K, k0, k1);
      for (int kb = k0; kb < k1; kb += kSeg) {
        const int m = (kb + kSeg < k1) ? kSeg : k1 - kb;
        __syncthreads();   // (the previous batch is done with the tables)
        const int in0 = pin[kb], all0 = pall[kb];
        for (int q = threadIdx.x; q <= m; q += kBlock) {
          const int pi = pin[kb + q] - in0, pa = pall[kb + q];
          s_pre[q] = make_int2(pi, (pa - all0) - pi);
          s_all[q] = pa;
          if (q < m) s_run[q] = runs[kb + q];
          if (TTV && q < m) {
            const int kq = rl.rbin[list * rl.r_max + kb + q];
            s_bin[q] = kq;
            s_shift[q] = kq >= 0 ? row.shift[kq] : 0.0;
            if (GRAD) {
#pragma unroll
              for (int w = 0; w < kWaves; ++w) s_grun[w][q] = 0.0;
            }
          }
        }
        __syncthreads();
        const bool trusted = TTV && s_bin[0] >= 0;   // (all runs of a list or none)
        const int tin = s_pre[m].x, total = tin + s_pre[m].y;
        // dense index j of the batch -> cadence i and position v in the value array: "inside" parts of all
        // runs first, then the limb parts, so that a wave's vote on the arc geometry is nearly unanimous.
        // Which run?  Transits recur: the runs of a list are nearly equally long, so position x runs / items is
        // the run or a neighbour of it; a wave steps its lanes to the right run (a vote per step) and only an uneven
        // list -- gaps in the series, the every-cadence fallback -- pays for a binary search (round 2 paid for one per
        // item: ~100 of the kernel's ~1100 vector instructions per cadence).
        const float g_in = tin > 0 ? (float)m / (float)tin : 0.0f;
        const float g_lim = total > tin ? (float)m / (float)(total - tin) : 0.0f;
        struct Item { int i, v, q; double tv, g, w; };
        auto locate = [&](int j, int& i, int& v, int& qrun) {
          const bool in = j < tin;
          const int jj = in ? j : j - tin;
          int q = (int)((float)jj * (in ? g_in : g_lim));
          q = q < m - 1 ? q : m - 1;
          int2 pa = s_pre[q], pb = s_pre[q + 1];
          int lo = in ? pa.x : pa.y, hi = in ? pb.x : pb.y;
          int tries = 0;
          while (EXO_WAVE_ANY((jj < lo) | (jj >= hi))) {
            if (++tries > 4) {
              q = 0;
#pragma unroll
              for (int step = kSeg / 2; step > 0; step >>= 1) {
                const int c2 = q + step;
                if (c2 < m) {
                  const int2 pc = s_pre[c2];
                  q = (jj >= (in ? pc.x : pc.y)) ? c2 : q;
                }
              }
              pa = s_pre[q];
              lo = in ? pa.x : pa.y;
              break;
            }
            q += (jj < lo) ? -1 : ((jj >= hi) ? 1 : 0);
            pa = s_pre[q]; pb = s_pre[q + 1];
            lo = in ? pa.x : pa.y; hi = in ? pb.x : pb.y;
          }
          const Run r = s_run[q];
          const int off = jj - lo;
          i = in ? r.a + off : ((off < r.a - r.lo) ? r.lo + off : r.b + (off - (r.a - r.lo)));
          v = s_all[q] + (i - r.lo);
          qrun = q;
        };
        auto load_item = [&](int j) -> Item {
          Item it{0, 0, 0, 0.0, 0.0, 0.0};   // lanes past the end of the batch: cadence 0 with a zero cotangent
          if (j < total) locate(j, it.i, it.v, it.q);
          it.tv = t[it.i];
          // the cotangent of the cadence's flux: dense [draw][cadence] (x planet), or -- gsparse -- at the value's own
          // position in the value array (transit_residual_kernel wrote it there)
          if (JAC) {
            it.g = (j < total) ? 1.0 : 0.0;   // unit cotangent: the row of derivatives itself
          } else if (CHI2) {
            if (j < total) { it.g = gflux[it.i]; it.w = gsparse[chi2_nw == 1 ? 0 : it.i]; }
          } else if (GRAD && j < total)
            it.g = gsparse ? gsparse[vbase + it.v]
                           : (per_planet ? gflux[(draw * n_cad + it.i) * n_planet + p]
                                         : ((flags & EXO_FLAG_CADENCE_MAJOR) ? gflux[(int64_t)it.i * gridDim.y + draw]
                                                                             : gflux[draw * n_cad + it.i]));
          return it;
        };
        Item nxt = load_item(threadIdx.x);
        for (int j0 = 0; j0 < total; j0 += kBlock) {
          const int j = j0 + threadIdx.x;
          const bool has = j < total;
          const Item cur = nxt;
          if (j0 + kBlock < total) nxt = load_item(j + kBlock);   // in flight while this round computes
          fc.issue(per_round);
          double f = 0.0;
          const double dsh = TTV ? s_shift[TTV ? cur.q : 0] : 0.0;
          for (int k = 0; k < n_sub; ++k) {
            // (the first sub-exposure's offset and weight sit in scalar registers: without an exposure time there is
            // no LDS read -- and no wait for one -- at the top of a round)
            const double sdt_k = (k == 0) ? sdt0 : sh.sdt[k], sw_k = (k == 0) ? sw0 : sh.sw[k];
            double tt = fma(te, sdt_k, cur.tv);
            int ks = 0;
            if (TTV) {
              double sh_k = dsh;
              if (!trusted) {
                ks = row.bin(tt);
                sh_k = row.shift[ks];
              }
              tt -= sh_k;
            }
            const double gw = cur.g * sw_k;
            const double F = eval_sample<GRAD, SECONDARY, LDELAY, CHI2>(tt, c, cld, CHI2 ? cur.g : gw, acc, cur.w);
            f = fma(sw_k, F, f);
            if (CHI2) {
              const double r = F - cur.g;
              acc.add(kNG + 6, cur.w * (r * r - cur.g * cur.g));
            } else if (GRAD && !JAC) {
              acc.add(kNG + 6, gw * F);
            }
            if (TTV && GRAD && !trusted) tgrad.flush_lane(ks);
          }
          if (TTV && GRAD && trusted) tgrad.flush_runs(cur.q, &s_grun[0][0], kSeg);
          if (JAC) {
            // this cadence's row: the thread's gradient columns hold sum_k w_k dF_k / d(slot); out they go, and back to zero
            double row[kJac];
#pragma unroll
            for (int q = 0; q < kJac; ++q) row[q] = 0.0;
#pragma unroll
            for (int sl = 0; sl < kNG + 6; ++sl) {
              if (jac_slot(sl) >= 0) {
                row[jac_slot(sl)] = lds_acc[sl][threadIdx.x];
                lds_acc[sl][threadIdx.x] = 0.0;
              }
            }
            if (has) {
              double2* __restrict__ dst = reinterpret_cast<double2*>(jac + (vbase + cur.v) * kJac);
#pragma unroll
              for (int q = 0; q < kJac / 2; ++q) dst[q] = make_double2(row[2 * q], row[2 * q + 1]);
            }
          }
          if (vals && has) {
            vals[vbase + cur.v] = f;
            if (vcad) vcad[vbase + cur.v] = cur.i;   // (dense output: where the last kernel puts it)
          }
        }
        if (TTV && GRAD && trusted) {
          __syncthreads();   // the waves' tables of this batch, in wave order
          for (int q = threadIdx.x; q < m; q += kBlock) {
            double v = s_grun[0][q];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) v += s_grun[(TTV && GRAD) ? w : 0][q];
            rl.grun[list * rl.r_max + kb + q] = v;
          }
        }
      }
      vbase += pall[K];
    }
    if (GRAD && TTV) {
      // every sample's t_periastron term went to its bin; the planet's total is in G_PAD
      lds_acc[G_TP][threadIdx.x] = lds_acc[G_PAD][threadIdx.x];
      lds_acc[G_PAD][threadIdx.x] = 0.0;
    }
    if (GRAD && !JAC) reduce_columns(lds_acc, sh.red, 0, kNG, pout + p * kNG);
  }
  if (GRAD && !JAC) reduce_columns(lds_acc, sh.red, kNG, 7, pout + n_planet * kNG);
  fc.issue(1 << 30);   // whatever is left of the fill (all of it for a block without work)
  if (fin.fold) {
    // the block owns its draw (hb = 1): partials -> gradients, values -> their cadences -- what transit_finish_kernel does
    // otherwise, without its launch; a block barrier is all the hand-shake its own stores need
    __syncthreads();
    finish_draw(draw, (GRAD && !JAC) ? partial : nullptr, hb, n_planet, SECONDARY, fin.gparams, fin.gld, fin.flux_dot, n_cad, flags, n_ev,
                rl, vals, vcad, fill, nullptr, 0, nullptr,
                (TTV && GRAD) ? ttv : Ttv{nullptr, nullptr, nullptr, 0});
  }
}

// The second half of the Jacobian route (transit_runs_kernel<.., JAC>): gparams, gld and sum(gflux flux) of a draw from the rows
// the value sweep left -- one block per draw, the planets in turn, every thread its share of the planet's solved cadences
// (the cotangent gathered through the cadence index of the value array), a fixed-order sum over the block: bit-reproducible.
// Output: the draw's partials in the layout of the runs kernel with ONE block per draw (transit_finish_kernel turns them
// into gparams / gld / flux_dot as it does for the sweep's).
__global__ __launch_bounds__(kBlock) void transit_jac_vjp_kernel(int64_t n_cad, int n_planet, int n_ev, uint32_t flags,
                                                                 RunLists rl, const double* __restrict__ vals,
                                                                 const int32_t* __restrict__ vcad,
                                                                 const double* __restrict__ jac,
                                                                 const double* __restrict__ gflux, int64_t n_draw,
                                                                 double* __restrict__ partial) {
  // (EXO_FLAG_SPARSE: gflux is the cotangent of the VALUES, in their layout -- what the sparse GP entries return)
  __shared__ double red[kJac + 1][kBlock];
  const int64_t draw = blockIdx.y;
  const int nb = gridDim.x, bx = blockIdx.x;       // a draw's cadences in nb contiguous shares (as the sweep's blocks share them)
  const int ng_draw = n_planet * kNG + 7;
  double* __restrict__ pout = partial + (draw * nb + bx) * ng_draw;
  const bool cmaj = flags & EXO_FLAG_CADENCE_MAJOR, gsp = flags & EXO_FLAG_SPARSE;
  double keep = 0.0;   // threads 10 .. 15: the running sum over planets of limb-darkening coefficient (thread - 10); thread 16: the dot
  for (int p = 0; p < n_planet; ++p) {
    int n_vals = 0;
    for (int ev = 0; ev < n_ev; ++ev) {
      const int64_t list = (draw * n_planet + p) * n_ev + ev;
      n_vals += rl.pre_all[list * (rl.r_max + 1) + rl.nrun[list]];
    }
    const int64_t vbase = (draw * n_planet + p) * n_cad;
    double acc[kJac + 1];
#pragma unroll
    for (int q = 0; q <= kJac; ++q) acc[q] = 0.0;
    const int v0 = (int)((int64_t)n_vals * bx / nb), v1 = (int)((int64_t)n_vals * (bx + 1) / nb);
    for (int v = v0 + threadIdx.x; v < v1; v += kBlock) {
      double g;
      if (gsp) {
        g = gflux[vbase + v];
      } else {
        const int64_t i = vcad[vbase + v];
        g = cmaj ? gflux[i * n_draw + draw] : gflux[draw * n_cad + i];
      }
      const double2* __restrict__ row = reinterpret_cast<const double2*>(jac + (vbase + v) * kJac);
#pragma unroll
      for (int q = 0; q < kJac / 2; ++q) {
        const double2 r2 = row[q];
        acc[2 * q] = fma(g, r2.x, acc[2 * q]);
        acc[2 * q + 1] = fma(g, r2.y, acc[2 * q + 1]);
      }
      acc[kJac] = fma(g, vals[vbase + v], acc[kJac]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q <= kJac; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    if (threadIdx.x <= kJac) {
      double v = 0.0;
      for (int k = 0; k < kBlock; ++k) v += red[threadIdx.x][k];
      const int q = threadIdx.x;
      if (q < 10) {
        pout[p * kNG + (q < 9 ? q : G_SINI)] = v;
      } else {
        keep += v;
      }
    }
    if (threadIdx.x < kNG && (threadIdx.x == G_PAD || threadIdx.x == G_CL)) pout[p * kNG + threadIdx.x] = 0.0;
  }
  if (threadIdx.x >= 10 && threadIdx.x <= kJac) pout[n_planet * kNG + (threadIdx.x - 10)] = keep;
}

// White-noise likelihood on the sparse output (exo_transit_chi2_vjp_f64), between the value sweep and the gradient
// sweep: for every value (cadence i of a run of list l = (planet, event) of the draw) the draw's TOTAL flux at i --
// its own value plus whatever the draw's other lists hold for that cadence (simultaneous transits: a binary search
// over each other list's runs) -- the residual r = total - obs[i], the cotangent 2 w_i r of the flux at i (written at
// the value's own position, where the gradient sweep reads it) and the draw's chi^2 relative to an empty light curve,
//     sum over solved cadences of  w_i ((total_i - obs_i)^2 - obs_i^2),
// each cadence counted once (by the first list that holds it).  Block partials in a fixed order (bit-reproducible).
constexpr int kResidualBlocks = 16;   // per draw
__global__ __launch_bounds__(kBlock) void transit_residual_kernel(int64_t n_cad, int n_planet, int n_ev, RunLists rl,
                                                                  const double* __restrict__ vals,
                                                                  const int32_t* __restrict__ vcad,
                                                                  const double* __restrict__ obs,
                                                                  const double* __restrict__ ivar, int64_t n_ivar,
                                                                  double* __restrict__ gvals,
                                                                  double* __restrict__ chi2_part) {
  __shared__ double red[kBlock];
  const int64_t draw = blockIdx.y;
  const int nb = gridDim.x, n_lists = n_planet * n_ev;
  double acc = 0.0;
  // value of list l2 at cadence i (0 if none of its runs holds it); `hit` says whether one does
  auto lookup = [&](int l2, int i, bool& hit) -> double {
    const int64_t list = draw * n_lists + l2;
    const int K = rl.nrun[list];
    const Run* __restrict__ runs = rl.runs + list * rl.r_max;
    int lo = 0, hi = K;                       // first run with run.lo > i
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (runs[mid].lo <= i) lo = mid + 1; else hi = mid;
    }
    hit = false;
    if (lo == 0) return 0.0;
    const Run r = runs[lo - 1];
    if (i >= r.hi) return 0.0;
    hit = true;
    const int p2 = l2 / n_ev, ev2 = l2 - p2 * n_ev;
    int64_t base = ((int64_t)draw * n_planet + p2) * n_cad;
    if (ev2 > 0) { const int64_t l0 = list - ev2; base += rl.pre_all[l0 * (rl.r_max + 1) + rl.nrun[l0]]; }
    return vals[base + rl.pre_all[list * (rl.r_max + 1) + lo - 1] + (i - r.lo)];
  };
  for (int l = 0; l < n_lists; ++l) {
    const int64_t list = draw * n_lists + l;
    const int p = l / n_ev, ev = l - p * n_ev;
    int64_t vbase = ((int64_t)draw * n_planet + p) * n_cad;
    if (ev > 0) { const int64_t l0 = list - ev; vbase += rl.pre_all[l0 * (rl.r_max + 1) + rl.nrun[l0]]; }
    const int total = rl.pre_all[list * (rl.r_max + 1) + rl.nrun[list]];
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < total; e += nb * kBlock) {
      const int i = vcad[vbase + e];
      double tot = vals[vbase + e];
      bool first = true;
      for (int l2 = 0; l2 < n_lists; ++l2) {
        if (l2 == l) continue;
        bool hit;
        tot += lookup(l2, i, hit);
        first = first && !(hit && l2 < l);
      }
      const double o = obs[i], w = ivar[n_ivar == 1 ? 0 : i];
      const double r = tot - o;
      gvals[vbase + e] = 2.0 * w * r;
      if (first) acc += w * (r * r - o * o);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int m = kBlock / 2; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) chi2_part[draw * nb + blockIdx.x] = red[0];
}

// (the reference's standalone Ops -- kepler, quad_solution_vector, contact_points -- are exo_ops.hip)

inline int launch_status() { return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH; }

// blocks per draw and tiles per block: enough blocks to fill 256 CUs several
// times over, few enough that each block amortises its prologue / reduction
inline void transit_geometry(int64_t n_cad, int64_t n_draw, int* blocks_per_draw, int* tiles_per_block) {
  const int64_t n_tiles = (n_cad + kTile - 1) / kTile;
  int64_t bpd = (kTargetBlocks + n_draw - 1) / n_draw;
  if (bpd > n_tiles) bpd = n_tiles;
  if (bpd < 1) bpd = 1;
  const int64_t tpb = (n_tiles + bpd - 1) / bpd;
  bpd = (n_tiles + tpb - 1) / tpb;
  *blocks_per_draw = (int)bpd;
  *tiles_per_block = (int)tpb;
}

// heavy blocks take the lists of `merge` consecutive scan blocks: the per-block costs of the
// heavy kernel (constant staging, accumulator reduction, a half-empty last round) are paid
// kHeavyTargetBlocks times rather than kTargetBlocks times, while the scan kernel keeps its finer
// blocks
#ifndef EXO_HEAVY_TARGET_BLOCKS
#define EXO_HEAVY_TARGET_BLOCKS 1024
#endif
inline int heavy_merge(int64_t n_draw, int bpd) {
  int64_t m = (n_draw * bpd + EXO_HEAVY_TARGET_BLOCKS - 1) / EXO_HEAVY_TARGET_BLOCKS;
  if (m > kMaxMerge) m = kMaxMerge;
  if (m > bpd) m = bpd;
  return m < 1 ? 1 : (int)m;
}

// scratch layout shared by forward and reverse: [gradient partials][wave counts][wave lists]
struct Workspace {
  double* partial;
  double* windows;
  int32_t* counts;
  int32_t* list;
  int64_t bytes;
};

inline Workspace carve(void* base, int64_t n_draw, int bpd, int tpb, int n_planet) {
  Workspace w;
  const int64_t n_partial = n_draw * bpd * (int64_t)(n_planet * kNG + 7);
  const int64_t n_slots = n_draw * bpd * (int64_t)kWaves;
  const int64_t n_counts = 2 * n_slots;  // (inside, limb) per wave list
  const int64_t n_list = n_slots * (int64_t)tpb * 128;
  char* p = (char*)base;
  const int64_t n_win = kWin * n_draw * n_planet;
  w.partial = (double*)p;
  w.windows = w.partial + n_partial;
  w.counts = (int32_t*)(p + (n_partial + n_win) * 8);
  w.list = w.counts + ((n_counts + 1) & ~(int64_t)1);
  w.bytes = (n_partial + n_win) * 8 + (((n_counts + 1) & ~(int64_t)1) + n_list) * 4;
  return w;
}

constexpr uint32_t kFlagNoFlux = 0x80000000u;  // internal: scan kernel must not touch flux

// scan kernel dispatch on (secondary, exact fp64 classification requested)
#define EXO_LAUNCH_SCAN_V(VEC, TTV, FLAGS, ...)                                                           \
  do {                                                                                                    \
    const bool sec_ = (FLAGS) & EXO_FLAG_SECONDARY, exact_ = (FLAGS) & EXO_FLAG_EXACT_SCAN;               \
    if (sec_ && exact_) hipLaunchKernelGGL((transit_scan_kernel<true, false, VEC, TTV>), __VA_ARGS__);    \
    else if (sec_) hipLaunchKernelGGL((transit_scan_kernel<true, true, VEC, TTV>), __VA_ARGS__);          \
    else if (exact_) hipLaunchKernelGGL((transit_scan_kernel<false, false, VEC, TTV>), __VA_ARGS__);      \
    else hipLaunchKernelGGL((transit_scan_kernel<false, true, VEC, TTV>), __VA_ARGS__);                   \
  } while (0)
// 16-B loads of t: pairs must not straddle the end (even n_cad) and t must be 16-B aligned.
// The timing-variation path (HAS_TTV) has one variant: per-cadence table lookups dwarf the loads.
#define EXO_LAUNCH_SCAN(N_CAD, T, HAS_TTV, FLAGS, ...)                                         \
  do {                                                                                         \
    if (HAS_TTV)                                                                               \
      EXO_LAUNCH_SCAN_V(false, true, FLAGS, __VA_ARGS__);                                      \
    else if (((N_CAD) & 1) == 0 && (reinterpret_cast<uintptr_t>(T) & 15) == 0)                 \
      EXO_LAUNCH_SCAN_V(true, false, FLAGS, __VA_ARGS__);                                      \
    else                                                                                       \
      EXO_LAUNCH_SCAN_V(false, false, FLAGS, __VA_ARGS__);                                     \
  } while (0)

// the windows of the scan kernel's first test: not needed when the caller asks for the exact
// fp64 scan of every cadence
inline void launch_windows(const double* params, int64_t n_draw, int n_planet, uint32_t flags, double* windows,
                           hipStream_t st) {
  if ((flags & EXO_FLAG_EXACT_SCAN) && !(flags & EXO_FLAG_WINDOW)) return;
  const int64_t n_rec = n_draw * n_planet;
  hipLaunchKernelGGL(transit_window_kernel, dim3((unsigned)((n_rec * kWinLanes + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                     params, n_rec, flags, windows);
}

// scan kernel launch: classify blocks (one per draw and tile run, or one per kScanDraws draws on
// the single-planet path) followed by one fill block per draw and tile run
struct ScanPlan {
  uint32_t flags;      // caller's flags + internal ones
  int64_t n_classify;  // classify blocks
  dim3 grid;
};
inline ScanPlan scan_plan(uint32_t flags, int bpd, int64_t n_draw, int n_planet, int64_t n_texp, bool with_fill) {
  ScanPlan sp;
  const bool stage1 = (flags & EXO_FLAG_WINDOW) || !(flags & EXO_FLAG_EXACT_SCAN);
  const bool grouped = n_planet == 1 && n_texp <= 1 && stage1;
  sp.flags = (flags & 0x0fffffffu) | (grouped ? kFlagGrouped : 0u) | (with_fill ? 0u : kFlagNoFlux);
  sp.n_classify = (grouped ? (n_draw + kScanDraws - 1) / kScanDraws : n_draw) * bpd;
  sp.grid = dim3((unsigned)(sp.n_classify + (with_fill ? n_draw * bpd : 0)));
  return sp;
}

// ---- run-enumeration path -------------------------------------------------------------------------
// heavy blocks per draw.  A round of a block (256 cadences through eval_sample) takes ~8 us whatever its fill, 512
// blocks are resident at once (two per CU), and every (planet, inside / limb) segment of a block ends in a partly
// filled round: one generation of fuller blocks beats two generations of emptier ones (C4 at 64 draws: 11 round
// times at 8 blocks per draw against 18 at 16).
// A draw that is ONE block's work (hb = 1: batches of >= 512 draws) is finished by that block -- no transit_finish_kernel launch.
// (Draws shared by several blocks finished by the last block to arrive -- fence + counter -- were measured in round 3 and
// removed in round 5: the device-scope release each block then needs writes the L2's dirty zero-fill lines back before it
// returns, heavy kernel 54 -> 147 us at 128 draws, 102 -> 225 us on C4 at 64.)
#ifndef EXO_RUNS_TARGET_BLOCKS
#define EXO_RUNS_TARGET_BLOCKS 512
#endif
inline int runs_blocks_per_draw(int64_t n_draw) {
  int64_t hb = (EXO_RUNS_TARGET_BLOCKS + n_draw - 1) / n_draw;
  return (int)(hb < 1 ? 1 : (hb > 64 ? 64 : hb));
}
inline int runs_r_max(int64_t n_cad) { return (int)(n_cad < kRunMax ? (n_cad < 16 ? 16 : n_cad) : kRunMax); }

// scratch layout of the run-enumeration path (sized for two events per planet whatever the flags)
struct RunWs {
  double* partial;
  double* windows;
  int32_t* sorted;
  RunLists rl;
  double* vals;
  int32_t* vcad;   // cadence of every value (dense output, chi^2)
  double* gvals;   // chi^2: cotangent of every value
  double* chi2_part;
  int32_t* done;   // [n_draw] blocks of a draw that are through with it
  int hb, n_sorted;
  int64_t off_nrun, off_runs, off_pre_all, off_vals;   // byte offsets (exo_transit_flux_sparse_layout)
  int64_t bytes;
};
inline RunWs carve_runs(void* base, int64_t n_cad, int64_t n_draw, int n_planet) {
  RunWs w;
  w.hb = runs_blocks_per_draw(n_draw);
  w.n_sorted = (int)((n_cad + kSortBlock - 1) / kSortBlock);
  w.rl.r_max = runs_r_max(n_cad);
  const int64_t n_list = n_draw * n_planet * 2;
  auto up16 = [](int64_t b) { return (b + 15) & ~(int64_t)15; };
  char* p = (char*)base;
  int64_t off = 0;
  w.partial = (double*)(p + off); off = up16(off + 8 * n_draw * w.hb * (int64_t)(n_planet * kNG + 7));
  w.windows = (double*)(p + off); off = up16(off + 8 * (int64_t)kWin * n_draw * n_planet);
  w.sorted = (int32_t*)(p + off); off = up16(off + 4 * (int64_t)w.n_sorted);
  w.off_nrun = off; w.rl.nrun = (int32_t*)(p + off); off = up16(off + 4 * n_list);
  w.off_runs = off; w.rl.runs = (Run*)(p + off); off = up16(off + (int64_t)sizeof(Run) * n_list * w.rl.r_max);
  w.rl.pre_in = (int32_t*)(p + off); off = up16(off + 4 * n_list * (int64_t)(w.rl.r_max + 1));
  w.off_pre_all = off; w.rl.pre_all = (int32_t*)(p + off); off = up16(off + 4 * n_list * (int64_t)(w.rl.r_max + 1));
  w.rl.rbin = (int32_t*)(p + off); off = up16(off + 4 * n_draw * n_planet * (int64_t)w.rl.r_max);
  w.rl.grun = (double*)(p + off); off = up16(off + 8 * n_draw * n_planet * (int64_t)w.rl.r_max);
  w.off_vals = off; w.vals = (double*)(p + off); off = up16(off + 8 * n_draw * n_planet * n_cad);
  w.vcad = (int32_t*)(p + off); off = up16(off + 4 * n_draw * n_planet * n_cad);
  w.gvals = (double*)(p + off); off = up16(off + 8 * n_draw * n_planet * n_cad);
  w.chi2_part = (double*)(p + off); off = up16(off + 8 * n_draw * kResidualBlocks);
  w.done = (int32_t*)(p + off); off = up16(off + 4 * n_draw);
  w.bytes = off;
  return w;
}
// which sweeps take the run-enumeration path (the list path keeps timing tables, per-cadence exposure
// times and the exact fp64 scan that the tests compare against)
inline bool runs_path(bool has_ttv, int64_t n_texp, uint32_t flags) {
  if (has_ttv && (flags & (EXO_FLAG_SECONDARY | EXO_FLAG_LIGHT_DELAY))) return false;
  return n_texp <= 1 && !(flags & EXO_FLAG_EXACT_SCAN);
}

// launches of one sweep on the run-enumeration path; gflux == nullptr: forward only.
// chi2 (obs != nullptr): value sweep into the sparse output, residuals + cotangents on it, gradient sweep reading them.
struct Chi2Args {
  const double* obs;
  const double* ivar;
  int64_t n_ivar;
  double* chi2;
};
template <bool G, bool SEC>
inline void launch_runs_kernel(bool ldelay, dim3 hgrid, hipStream_t st, const double* t, int64_t n_cad, const double* texp,
                               int64_t n_texp, const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                               const double* params, const double* ld, int32_t n_planet, uint32_t flags, int n_ev,
                               const RunLists& rl, const double* gflux, const double* gsparse, double* vals, int32_t* vcad,
                               double* fill, double* partial, const FinishArgs& fin) {
  const Ttv no_ttv{nullptr, nullptr, nullptr, 0};
  if (ldelay)
    hipLaunchKernelGGL((transit_runs_kernel<G, SEC, true>), hgrid, dim3(kBlock), 0, st, t, n_cad, texp, n_texp, stencil_dt,
                       stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, rl, gflux, gsparse, vals, vcad, fill,
                       partial, (int64_t)0, no_ttv, fin);
  else
    hipLaunchKernelGGL((transit_runs_kernel<G, SEC, false>), hgrid, dim3(kBlock), 0, st, t, n_cad, texp, n_texp, stencil_dt,
                       stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, rl, gflux, gsparse, vals, vcad, fill,
                       partial, (int64_t)0, no_ttv, fin);
}
inline int launch_runs_sweep(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                             const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                             int64_t n_draw, int32_t n_planet, uint32_t flags, const double* gflux, double* flux,
                             double* gparams, double* gld, double* flux_dot, const RunWs& w, hipStream_t st,
                             const Chi2Args* chi2 = nullptr, const Ttv* ttv = nullptr, double* jac = nullptr,
                             const double* gvals = nullptr, bool reuse_runs = false, const PackIn* pack = nullptr) {
  // pack: the records are not there yet -- `params` / `ld` are where the fused packing + enumeration launch will put them
  // (requires the fused launch: sorted times on the caller's word, no timing tables)
  // gvals: the cotangent in the VALUE layout of the sparse output (exo_transit_flux_vjp_sparse_f64) instead of gflux;
  // reuse_runs: the workspace still holds the windows and runs of these very records (the forward sweep's): no enumeration
  const bool secondary = flags & EXO_FLAG_SECONDARY, sparse = (flags & EXO_FLAG_SPARSE) || chi2;
  const bool grad = gflux != nullptr || chi2 || gvals != nullptr;
  const int n_ev = secondary ? 2 : 1;
  const dim3 block(kBlock);
  const bool has_ttv = ttv && ttv->edges;
  // sorted times on the caller's word and no fence counters to clear: windows and runs in ONE launch
  const bool fused_enum = (flags & EXO_FLAG_SORTED_TIMES) && !has_ttv;
  if (reuse_runs) {
    // (nothing to launch)
  } else if (!fused_enum) {
    const int64_t n_rec = n_draw * n_planet;
    hipLaunchKernelGGL(transit_window_kernel, dim3((unsigned)((n_rec * kWinLanes + kBlock - 1) / kBlock + w.n_sorted)), block, 0, st,
                       params, n_rec, flags, w.windows, t, n_cad, w.sorted, w.done, n_draw);
  }
  if (pack && (!fused_enum || reuse_runs)) return EXO_ERR_INVALID_ARGUMENT;
  if (reuse_runs) {
  } else if (pack) {
    const PackIn pk = *pack;
    hipLaunchKernelGGL((transit_enum_kernel<true, true>), dim3((unsigned)(n_draw * n_planet * n_ev)), dim3(64), 0, st, t, n_cad, texp,
                       n_texp, stencil_dt, (int)n_sub, flags, (const double*)nullptr, (const int32_t*)nullptr, 0, n_ev, w.rl,
                       params, w.windows, pk);
  } else if (fused_enum)
    hipLaunchKernelGGL(transit_enum_kernel<true>, dim3((unsigned)(n_draw * n_planet * n_ev)), dim3(64), 0, st, t, n_cad, texp,
                       n_texp, stencil_dt, (int)n_sub, flags, (const double*)nullptr, (const int32_t*)nullptr, 0, n_ev, w.rl,
                       params, w.windows);
  else if (has_ttv)
    hipLaunchKernelGGL(transit_enum_ttv_kernel, dim3((unsigned)(n_draw * n_planet)), dim3(64), 0, st, t, n_cad, texp, n_texp,
                       stencil_dt, (int)n_sub, flags, w.windows, w.sorted, w.n_sorted, w.rl, *ttv);
  else
    hipLaunchKernelGGL(transit_enum_kernel<false>, dim3((unsigned)(n_draw * n_planet * n_ev)), dim3(64), 0, st, t, n_cad, texp,
                       n_texp, stencil_dt, (int)n_sub, flags, w.windows, w.sorted, w.n_sorted, n_ev, w.rl);
  if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
  // the values are kept when somebody reads them: the dense output's last kernel, or the caller (sparse)
  double* vals = (flux || sparse) ? w.vals : nullptr;
  double* fill = sparse ? nullptr : flux;
  const dim3 hgrid((unsigned)w.hb, (unsigned)n_draw);
  const bool ldelay = flags & EXO_FLAG_LIGHT_DELAY;
  // a draw that is one block's work is finished by that block (gradients from its partials, values to their cadences):
  // no transit_finish_kernel launch
  // (not with a cadence-major flux: a draw's values land in lines other blocks zero-fill -- after the sweep, then)
  const bool cmaj = flags & EXO_FLAG_CADENCE_MAJOR;
  const bool fold = !(cmaj && flux) && w.hb == 1;
  const FinishArgs fin{gparams, gld, chi2 ? chi2->chi2 : flux_dot, fold ? 1 : 0, w.done}, no_fin{nullptr, nullptr, nullptr, 0, nullptr};
#define EXO_LAUNCH_RUNS(G, GFLUX, GSP, VALS, VCAD, FILL, PARTIAL, FIN)                                                    \
  if (secondary)                                                                                                          \
    launch_runs_kernel<G, true>(ldelay, hgrid, st, t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld,      \
                                n_planet, flags, n_ev, w.rl, GFLUX, GSP, VALS, VCAD, FILL, PARTIAL, FIN);                  \
  else                                                                                                                    \
    launch_runs_kernel<G, false>(ldelay, hgrid, st, t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld,     \
                                 n_planet, flags, n_ev, w.rl, GFLUX, GSP, VALS, VCAD, FILL, PARTIAL, FIN)
  if (chi2 && n_planet == 1 && !secondary && n_sub == 1) {
    // one planet, one sample per cadence: the cotangent of a cadence's flux needs nothing but that flux -- value and
    // gradient in ONE evaluation per solved cadence (the misfit comes out of the "dot" slot of the partials)
    if (has_ttv)
      hipLaunchKernelGGL((transit_runs_kernel<true, false, false, true, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                         stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, chi2->obs, chi2->ivar,
                         nullptr, nullptr, nullptr, w.partial, chi2->n_ivar, *ttv, fin);
    else if (ldelay)
      hipLaunchKernelGGL((transit_runs_kernel<true, false, true, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp, stencil_dt,
                         stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, chi2->obs, chi2->ivar, nullptr,
                         nullptr, nullptr, w.partial, chi2->n_ivar, Ttv{nullptr, nullptr, nullptr, 0}, fin);
    else
      hipLaunchKernelGGL((transit_runs_kernel<true, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp, stencil_dt,
                         stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, chi2->obs, chi2->ivar, nullptr,
                         nullptr, nullptr, w.partial, chi2->n_ivar, Ttv{nullptr, nullptr, nullptr, 0}, fin);
    if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
    if (fold) return EXO_OK;
    hipLaunchKernelGGL(transit_finish_kernel, dim3((unsigned)n_draw), dim3(n_draw <= 256 ? 1024 : kBlock), 0, st, w.partial, w.hb,
                       (int)n_planet, secondary, gparams, gld, chi2->chi2, n_cad, flags, n_ev, w.rl, nullptr, w.vcad, nullptr,
                       nullptr, 0, nullptr, has_ttv ? *ttv : Ttv{nullptr, nullptr, nullptr, 0});
    return launch_status();
  }
  if (has_ttv && chi2) {
    // value sweep into the sparse output, residuals + cotangents on it, gradient sweep reading them (gshift included)
    hipLaunchKernelGGL((transit_runs_kernel<false, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                       stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, nullptr, nullptr, w.vals,
                       w.vcad, nullptr, nullptr, (int64_t)0, *ttv, no_fin);
    hipLaunchKernelGGL(transit_residual_kernel, dim3(kResidualBlocks, (unsigned)n_draw), block, 0, st, n_cad, (int)n_planet,
                       n_ev, w.rl, w.vals, w.vcad, chi2->obs, chi2->ivar, chi2->n_ivar, w.gvals, w.chi2_part);
    hipLaunchKernelGGL((transit_runs_kernel<true, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                       stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, nullptr, w.gvals, nullptr,
                       nullptr, nullptr, w.partial, (int64_t)0, *ttv, no_fin);
  } else if (has_ttv) {
    // (transits only, no light delay: runs_path)
    if (grad)
      hipLaunchKernelGGL((transit_runs_kernel<true, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                         stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, gflux, nullptr, vals,
                         fill ? w.vcad : nullptr, fill, w.partial, (int64_t)0, *ttv, fin);
    else
      hipLaunchKernelGGL((transit_runs_kernel<false, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                         stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, nullptr, nullptr, vals,
                         fill ? w.vcad : nullptr, fill, nullptr, (int64_t)0, *ttv, fin);
  } else if (chi2) {
    EXO_LAUNCH_RUNS(false, nullptr, nullptr, w.vals, w.vcad, nullptr, nullptr, no_fin);
    hipLaunchKernelGGL(transit_residual_kernel, dim3(kResidualBlocks, (unsigned)n_draw), block, 0, st, n_cad, (int)n_planet,
                       n_ev, w.rl, w.vals, w.vcad, chi2->obs, chi2->ivar, chi2->n_ivar, w.gvals, w.chi2_part);
    EXO_LAUNCH_RUNS(true, nullptr, w.gvals, nullptr, nullptr, nullptr, w.partial, no_fin);
  } else if (gvals) {
    // (the values stay as the forward sweep left them: the GP's reverse pass has read them, nobody reads them again)
    EXO_LAUNCH_RUNS(true, nullptr, gvals, nullptr, nullptr, nullptr, w.partial, fin);
  } else if (grad) {
    EXO_LAUNCH_RUNS(true, gflux, nullptr, vals, fill ? w.vcad : nullptr, fill, w.partial, fin);
  } else if (jac) {
    // value sweep that leaves every solved cadence's row of derivatives (transit_runs_kernel<.., JAC>; the cadence index
    // is written whatever the output: the contraction gathers the cotangent through it)
    const Ttv no_ttv{nullptr, nullptr, nullptr, 0};
    if (secondary)
      hipLaunchKernelGGL((transit_runs_kernel<true, true, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                         stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, nullptr, nullptr, vals,
                         w.vcad, fill, jac, (int64_t)0, no_ttv, fin);
    else
      hipLaunchKernelGGL((transit_runs_kernel<true, false, false, false, false, true>), hgrid, block, 0, st, t, n_cad, texp, n_texp,
                         stencil_dt, stencil_w, (int)n_sub, params, ld, (int)n_planet, flags, n_ev, w.rl, nullptr, nullptr, vals,
                         w.vcad, fill, jac, (int64_t)0, no_ttv, fin);
  } else {
    EXO_LAUNCH_RUNS(false, nullptr, nullptr, vals, fill ? w.vcad : nullptr, fill, nullptr, fin);
  }
#undef EXO_LAUNCH_RUNS
  if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
  const bool three_sweeps = chi2 != nullptr;   // (the single-pass likelihood returned above)
  if ((grad || fill) && (!fold || three_sweeps))
    hipLaunchKernelGGL(transit_finish_kernel, dim3((unsigned)n_draw), dim3(n_draw <= 256 ? 1024 : kBlock), 0, st,
                       grad ? w.partial : nullptr, w.hb, (int)n_planet, secondary, gparams, gld, flux_dot, n_cad, flags, n_ev,
                       w.rl, chi2 ? nullptr : vals, w.vcad, fill, chi2 ? w.chi2_part : nullptr, kResidualBlocks,
                       chi2 ? chi2->chi2 : nullptr, (has_ttv && grad) ? *ttv : Ttv{nullptr, nullptr, nullptr, 0});
  return launch_status();
}

// A DENSE flux array kept across steps (exo_transit_sparse_scatter_f64): the summed flux of the cadences in a sparse output's
// runs written into -- or, CLEAR, zeroed in -- a dense [n_draw][n_cad] array that is otherwise left alone.  A step of a sampler
// solves the same few per cent of the cadences as the step before it: clear the last step's, write this one's, and the
// dense result costs the sparse sweep plus two passes over the solved cadences instead of a fill of every cadence (1.2 GB at
// C2).  A block per draw; a wave per run (its cadences are consecutive: coalesced); planets in order with a block barrier,
// the first one storing and the later ones adding with the hardware's fp64 atomic -- the dense sweep's own order, so the
// same bits.
template <bool CLEAR>
__global__ __launch_bounds__(kBlock) void transit_scatter_runs_kernel(RunLists rl, const double* __restrict__ vals, int64_t n_cad,
                                                                      int n_planet, int n_ev, double* __restrict__ flux) {
  const int64_t draw = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_wave = kBlock / 64;
  double* __restrict__ row = flux + draw * n_cad;
  for (int p = 0; p < n_planet; ++p) {
    int64_t vbase = (draw * n_planet + p) * n_cad;      // the planet's values: transits first, occultations behind them
    for (int ev = 0; ev < n_ev; ++ev) {
      const int64_t list = (draw * n_planet + p) * n_ev + ev;
      // (a workspace that is not a sparse output -- never zeroed, never swept -- must not become a wild store: counts and
      // cadences are held to the arrays' bounds)
      int K = rl.nrun[list];
      K = K < 0 ? 0 : (K > rl.r_max ? rl.r_max : K);
      const Run* __restrict__ runs = rl.runs + list * rl.r_max;
      const int32_t* __restrict__ pall = rl.pre_all + list * (rl.r_max + 1);
      for (int k = wave; k < K; k += n_wave) {
        int lo = runs[k].lo, len = runs[k].hi - lo;
        const int pk = pall[k];
        if (lo < 0 || len < 0 || (int64_t)lo + len > n_cad || pk < 0 || (int64_t)pk + len > n_cad) continue;
        const int64_t v0 = vbase + pk;
        for (int i = lane; i < len; i += 64) {
          if (CLEAR) row[lo + i] = 0.0;
          // (planet 0 stores for BOTH events: a planet's transit and occultation lists never share a cadence -- the enumeration
          // keeps two lists only when the windows are disjoint, h0 + h1 < their separation; otherwise event 0 is "every
          // cadence" and event 1 is empty: transit_enum_kernel.  The dense sweep relies on the same invariant, finish_draw.)
          else if (p == 0) row[lo + i] = vals[v0 + i];
          else unsafeAtomicAdd(row + lo + i, vals[v0 + i]);
        }
      }
      const int tot = pall[K];
      vbase += (tot < 0 || tot > n_cad) ? 0 : tot;
    }
    if (!CLEAR && p + 1 < n_planet) __syncthreads();   // planets in order
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The merged sparse model (round 6).  A draw's lists -- (planet, event) -- each hold ascending, disjoint runs of cadences and
// their flux values; lists of DIFFERENT planets may overlap (simultaneous transits), and a list may be the whole series (a
// window that could not be bounded).  For the celerite kernels a draw's mean must be ONE ascending list of disjoint segments
// with one value per cadence: the union of the runs, the values summed over the lists (limb_dark.py:228-230 sums the planets,
// secondary_eclipse.py:67-70 blends transit and occultation -- the blend's weights are in the values already).
//   sparse_merge_segments_kernel   a block per draw: every run's rank among all the draw's runs by binary search in the other
//                                  lists (no sort: each list is sorted), then a prefix-maximum scan of the run ends decides
//                                  where a new segment starts; segment bounds + the prefix sums of their lengths
//   sparse_merge_values_kernel     a thread per merged cadence: its value = the sum over the lists that hold it (binary search)
//   sparse_merge_vjp_kernel        a thread per value of a list: the cotangent of the merged value of its cadence
// All O(solved cadences x lists x log runs): ~3 % of a dense pass.
// ---------------------------------------------------------------------------------------------------------------------
struct MergeWs {
  int32_t* nseg;     // [n_draw]
  int32_t* seg;      // [n_draw][cap_seg][2]      (lo, hi)
  int32_t* off;      // [n_draw][cap_seg + 1]     exclusive prefix sums of hi - lo; [nseg] = the draw's number of values
  int32_t* sorted;   // [n_draw][cap_run][2]      scratch: every run of the draw, by lo
  double* vals;      // [n_draw][n_cad]
  int cap_seg, cap_run;
  int64_t off_nseg, off_seg, off_off, off_vals, bytes;
};
inline MergeWs carve_merge(void* base, int64_t n_cad, int64_t n_draw, int n_planet) {
  MergeWs m;
  const int64_t runs = (int64_t)n_planet * 2 * runs_r_max(n_cad);
  m.cap_run = (int)runs;
  m.cap_seg = (int)(runs < n_cad + 1 ? runs : n_cad + 1);       // (disjoint segments of >= 1 cadence each)
  auto up16 = [](int64_t b) { return (b + 15) & ~(int64_t)15; };
  char* p = (char*)base;
  int64_t off = 0;
  m.off_nseg = off; m.nseg = (int32_t*)(p + off); off = up16(off + 4 * n_draw);
  m.off_seg = off; m.seg = (int32_t*)(p + off); off = up16(off + 8 * n_draw * (int64_t)m.cap_seg);
  m.off_off = off; m.off = (int32_t*)(p + off); off = up16(off + 4 * n_draw * ((int64_t)m.cap_seg + 1));
  m.sorted = (int32_t*)(p + off); off = up16(off + 8 * n_draw * (int64_t)m.cap_run);
  m.off_vals = off; m.vals = (double*)(p + off); off = up16(off + 8 * n_draw * n_cad);
  m.bytes = off;
  return m;
}
inline int merge_blocks_per_draw(int64_t n_draw) {
  const int64_t b = (4096 + n_draw - 1) / (n_draw > 0 ? n_draw : 1);
  return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

// number of runs of a list whose lo is < key (strict) / <= key
__device__ __forceinline__ int runs_lower(const Run* __restrict__ runs, int K, int key, bool or_equal) {
  int lo = 0, hi = K;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int v = runs[mid].lo;
    if (v < key || (or_equal && v == key)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the run of a list that holds cadence n, or -1
__device__ __forceinline__ int runs_find(const Run* __restrict__ runs, int K, int n) {
  const int k = runs_lower(runs, K, n, true) - 1;     // last run with lo <= n
  return (k >= 0 && n < runs[k].hi) ? k : -1;
}

// inclusive scans over a block of kBlock threads through LDS (s: kBlock ints): max / sum
__device__ __forceinline__ int block_scan_max(int v, int* s) {
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {
    const int o = (int)threadIdx.x >= d ? s[threadIdx.x - d] : INT32_MIN;
    __syncthreads();
    if (o > s[threadIdx.x]) s[threadIdx.x] = o;
    __syncthreads();
  }
  return s[threadIdx.x];
}
__device__ __forceinline__ int block_scan_sum(int v, int* s) {
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {
    const int o = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
    __syncthreads();
    s[threadIdx.x] += o;
    __syncthreads();
  }
  return s[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void sparse_merge_segments_kernel(RunLists rl, int n_planet, int n_ev, int64_t n_cad, MergeWs m) {
  const int64_t draw = blockIdx.x;
  const int tid = threadIdx.x, n_lists = n_planet * n_ev;
  __shared__ int s_first[2 * EXO_MAX_PLANETS + 1];   // runs before list l of this draw
  __shared__ int s_scan[kBlock];
  __shared__ int s_flag[kBlock + 1];
  __shared__ int s_carry[2];
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < n_lists; ++l) {
      int K = rl.nrun[draw * n_lists + l];
      K = K < 0 ? 0 : (K > rl.r_max ? rl.r_max : K);
      s_first[l] = acc;
      acc += K;
    }
    s_first[n_lists] = acc;
  }
  __syncthreads();
  const int R = s_first[n_lists];
  int32_t* __restrict__ sorted = m.sorted + draw * (int64_t)m.cap_run * 2;
  int32_t* __restrict__ seg = m.seg + draw * (int64_t)m.cap_seg * 2;
  int32_t* __restrict__ off = m.off + draw * ((int64_t)m.cap_seg + 1);
  const int ncad = (int)n_cad;
  // 1) rank: a run's position among all runs of the draw by (lo, list)
  for (int g = tid; g < R; g += kBlock) {
    int l = 0;
    while (l + 1 < n_lists && g >= s_first[l + 1]) ++l;
    const int k = g - s_first[l];
    const Run* __restrict__ mine = rl.runs + (draw * n_lists + l) * rl.r_max;
    int lo = mine[k].lo, hi = mine[k].hi;
    lo = lo < 0 ? 0 : (lo > ncad ? ncad : lo);
    hi = hi < lo ? lo : (hi > ncad ? ncad : hi);
    int rank = k;
    for (int l2 = 0; l2 < n_lists; ++l2) {
      if (l2 == l) continue;
      const int K2 = s_first[l2 + 1] - s_first[l2];
      rank += runs_lower(rl.runs + (draw * n_lists + l2) * rl.r_max, K2, lo, l2 < l);
    }
    sorted[2 * rank] = lo;
    sorted[2 * rank + 1] = hi;
  }
  if (tid == 0) { s_carry[0] = INT32_MIN; s_carry[1] = 0; }
  __syncthreads();     // (the block's global stores are visible to the block behind the barrier)
  // 2) where segments start: a run starts one iff its lo is not below the largest hi before it (empty runs start nothing)
  for (int base = 0; base < R; base += kBlock) {
    const int i = base + tid;
    const bool valid = i < R;
    const int lo = valid ? sorted[2 * i] : INT32_MAX, hi = valid ? sorted[2 * i + 1] : INT32_MIN;
    const bool live = valid && hi > lo;
    const int carry_max = s_carry[0], carry_seg = s_carry[1];
    const int incl = block_scan_max(live ? hi : INT32_MIN, s_scan);
    int excl = tid > 0 ? s_scan[tid - 1] : INT32_MIN;
    excl = excl > carry_max ? excl : carry_max;
    const bool start = live && lo >= excl;
    __syncthreads();
    const int nstart = block_scan_sum(start ? 1 : 0, s_scan);
    const int sidx = carry_seg + nstart - 1;          // the segment this run belongs to (live runs)
    s_flag[tid] = start ? 1 : 0;
    if (tid == 0) s_flag[kBlock] = 1;
    __syncthreads();
    if (start && sidx < m.cap_seg) seg[2 * sidx] = lo;
    // the end of a segment = the prefix maximum at its last live run; a segment that goes on in the next round is written
    // again there, with a maximum that includes this round's
    if (live && sidx >= 0 && sidx < m.cap_seg) {
      // last live run of its segment within this round: no later run of the round is live without starting a segment ... the
      // prefix maximum is monotone, so EVERY live run may write it as long as the writes are ordered: only the last one does
      bool last = true;
      for (int j = tid + 1; j < kBlock && base + j < R; ++j) {
        if (s_flag[j]) break;                          // the next segment starts: this one ended before it
        const int hj = sorted[2 * (base + j) + 1], lj = sorted[2 * (base + j)];
        if (hj > lj) { last = false; break; }          // a later live run of the same segment
      }
      const int end = incl > carry_max ? incl : carry_max;
      if (last) seg[2 * sidx + 1] = end;
    }
    __syncthreads();
    if (tid == kBlock - 1) {
      s_carry[0] = incl > carry_max ? incl : carry_max;
      s_carry[1] = carry_seg + nstart;
    }
    __syncthreads();
  }
  int S = s_carry[1];
  S = S > m.cap_seg ? m.cap_seg : S;
  // 3) prefix sums of the segment lengths
  if (tid == 0) s_carry[0] = 0;
  __syncthreads();
  for (int base = 0; base < S; base += kBlock) {
    const int i = base + tid;
    const int len = i < S ? seg[2 * i + 1] - seg[2 * i] : 0;
    const int carry = s_carry[0];
    const int incl = block_scan_sum(len, s_scan);
    if (i < S) off[i] = carry + incl - len;
    __syncthreads();
    if (tid == kBlock - 1) s_carry[0] = carry + incl;
    __syncthreads();
  }
  if (tid == 0) {
    off[S] = s_carry[0];
    m.nseg[draw] = S;
  }
}

// position of list l's values in the value array of (draw, planet): occultations behind the transits
__device__ __forceinline__ int64_t list_vbase(const RunLists& rl, int64_t draw, int n_planet, int n_ev, int p, int ev, int64_t n_cad) {
  int64_t vbase = (draw * n_planet + p) * n_cad;
  if (ev > 0) {
    const int64_t l0 = (draw * n_planet + p) * n_ev;
    int K0 = rl.nrun[l0];
    K0 = K0 < 0 ? 0 : (K0 > rl.r_max ? rl.r_max : K0);
    vbase += rl.pre_all[l0 * (rl.r_max + 1) + K0];
  }
  return vbase;
}

__global__ __launch_bounds__(kBlock) void sparse_merge_values_kernel(RunLists rl, const double* __restrict__ vals, int n_planet, int n_ev,
                                                                     int64_t n_cad, MergeWs m) {
  const int64_t draw = blockIdx.y;
  const int n_lists = n_planet * n_ev;
  const int S = m.nseg[draw];
  const int32_t* __restrict__ seg = m.seg + draw * (int64_t)m.cap_seg * 2;
  const int32_t* __restrict__ off = m.off + draw * ((int64_t)m.cap_seg + 1);
  double* __restrict__ out = m.vals + draw * n_cad;
  const int total = off[S];
  for (int pos = blockIdx.x * kBlock + threadIdx.x; pos < total; pos += gridDim.x * kBlock) {
    // the segment of this position: last s with off[s] <= pos
    int lo = 0, hi = S;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    const int sgm = lo - 1;
    const int n = seg[2 * sgm] + (pos - off[sgm]);
    double v = 0.0;
    for (int l = 0; l < n_lists; ++l) {       // in list order: the dense sweep's order of summation
      const int64_t list = draw * n_lists + l;
      int K = rl.nrun[list];
      K = K < 0 ? 0 : (K > rl.r_max ? rl.r_max : K);
      const Run* __restrict__ runs = rl.runs + list * rl.r_max;
      const int k = runs_find(runs, K, n);
      if (k >= 0) {
        const int p = l / n_ev, ev = l - p * n_ev;
        v += vals[list_vbase(rl, draw, n_planet, n_ev, p, ev, n_cad) + rl.pre_all[list * (rl.r_max + 1) + k] + (n - runs[k].lo)];
      }
    }
    out[pos] = v;
  }
}

__global__ __launch_bounds__(kBlock) void sparse_merge_vjp_kernel(RunLists rl, int n_planet, int n_ev, int64_t n_cad, MergeWs m,
                                                                  const double* __restrict__ gm, double* __restrict__ gvals) {
  const int64_t draw = blockIdx.y;
  const int n_lists = n_planet * n_ev;
  const int S = m.nseg[draw];
  const int32_t* __restrict__ seg = m.seg + draw * (int64_t)m.cap_seg * 2;
  const int32_t* __restrict__ off = m.off + draw * ((int64_t)m.cap_seg + 1);
  const double* __restrict__ g = gm + draw * n_cad;
  for (int l = 0; l < n_lists; ++l) {
    const int64_t list = draw * n_lists + l;
    int K = rl.nrun[list];
    K = K < 0 ? 0 : (K > rl.r_max ? rl.r_max : K);
    const Run* __restrict__ runs = rl.runs + list * rl.r_max;
    const int32_t* __restrict__ pall = rl.pre_all + list * (rl.r_max + 1);
    const int p = l / n_ev, ev = l - p * n_ev;
    const int64_t vbase = list_vbase(rl, draw, n_planet, n_ev, p, ev, n_cad);
    const int total = pall[K];
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
      int lo = 0, hi = K;                     // the run of value e: last k with pre_all[k] <= e
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pall[mid] <= e) lo = mid + 1; else hi = mid;
      }
      const int k = lo - 1;
      const int n = runs[k].lo + (e - pall[k]);
      lo = 0; hi = S;                         // the merged segment of cadence n: last s with seg lo <= n
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (seg[2 * mid] <= n) lo = mid + 1; else hi = mid;
      }
      const int sgm = lo - 1;
      gvals[vbase + e] = (sgm >= 0 && n < seg[2 * sgm + 1]) ? g[off[sgm] + (n - seg[2 * sgm])] : 0.0;
    }
  }
}

// every flag bit a sweep knows; anything else is a newer header talking to this library (ABI 10: refused, not ignored --
// a layout flag this build does not know would otherwise come back as a silently different array)
inline bool sweep_flags_ok(uint32_t flags) { return (flags & ~(uint32_t)EXO_FLAG_SWEEP_ALL) == 0; }

inline bool transit_args_ok(int64_t n_cad, int64_t n_texp, int32_t n_sub, int64_t n_draw, int32_t n_planet) {
  return n_cad >= 0 && n_draw >= 0 && n_draw <= 65535 && n_planet >= 1 && n_planet <= EXO_MAX_PLANETS &&
         n_sub >= 1 && n_sub <= EXO_MAX_SUBEXP && (n_texp == 0 || n_texp == 1 || n_texp == n_cad);
}

}  // namespace

extern "C" {

int32_t exo_abi_version(void) { return EXO_ABI_VERSION; }

int64_t exo_transit_flux_workspace_bytes(int64_t n_cad, int64_t n_draw, int32_t n_planet) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1) return -1;
  if (n_cad == 0 || n_draw == 0) return 0;
  int bpd, tpb;
  transit_geometry(n_cad, n_draw, &bpd, &tpb);
  const Workspace w = carve(nullptr, n_draw, bpd, tpb, n_planet);
  const RunWs r = carve_runs(nullptr, n_cad, n_draw, n_planet);
  return w.bytes > r.bytes ? w.bytes : r.bytes;
}

int exo_transit_flux_sparse_layout(int64_t n_cad, int64_t n_draw, int32_t n_planet, int64_t* out) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || !out) return EXO_ERR_INVALID_ARGUMENT;
  const RunWs r = carve_runs(nullptr, n_cad, n_draw, n_planet);
  out[0] = r.off_nrun; out[1] = r.off_runs; out[2] = r.off_pre_all; out[3] = r.off_vals; out[4] = r.rl.r_max;
  return EXO_OK;
}

int exo_transit_sparse_scatter_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                                   uint32_t flags, int32_t clear, double* flux, void* stream) {
  if (n_cad < 0 || n_draw < 0 || n_draw > 65535 || n_planet < 1 || n_planet > EXO_MAX_PLANETS ||
      (flags & ~(uint32_t)EXO_FLAG_SECONDARY))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n_cad == 0 || n_draw == 0) return EXO_OK;
  if (!flux) return EXO_ERR_INVALID_ARGUMENT;
  const RunWs rw = carve_runs(const_cast<void*>(workspace), n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  const int n_ev = (flags & EXO_FLAG_SECONDARY) ? 2 : 1;
  if (clear)
    hipLaunchKernelGGL(transit_scatter_runs_kernel<true>, dim3((unsigned)n_draw), dim3(kBlock), 0, (hipStream_t)stream, rw.rl, rw.vals,
                       n_cad, (int)n_planet, n_ev, flux);
  else
    hipLaunchKernelGGL(transit_scatter_runs_kernel<false>, dim3((unsigned)n_draw), dim3(kBlock), 0, (hipStream_t)stream, rw.rl, rw.vals,
                       n_cad, (int)n_planet, n_ev, flux);
  return launch_status();
}

// forward sweep; ttv.edges == nullptr: no timing variations
static int transit_fwd(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                       const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                       int64_t n_draw, int32_t n_planet, uint32_t flags, const Ttv& ttv, double* flux,
                       void* workspace, int64_t workspace_bytes, void* stream, void* ev_start, void* ev_stop) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || !sweep_flags_ok(flags)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_cad == 0 || n_draw == 0) return EXO_OK;
  if (!t || !params || !ld || (!flux && !(flags & EXO_FLAG_SPARSE)) || (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  const bool has_ttv = ttv.edges != nullptr;
  hipStream_t st = (hipStream_t)stream;
  if ((flags & EXO_FLAG_CADENCE_MAJOR) && (flags & EXO_FLAG_PER_PLANET)) return EXO_ERR_INVALID_ARGUMENT;
  if (runs_path(has_ttv, n_texp, flags)) {
    const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
    if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
    if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
    const int rc = launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet,
                                     flags, nullptr, flux, nullptr, nullptr, nullptr, rw, st, nullptr, &ttv);
    if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
    return rc;
  }
  if (flags & (EXO_FLAG_SPARSE | EXO_FLAG_LIGHT_DELAY | EXO_FLAG_CADENCE_MAJOR)) return EXO_ERR_INVALID_ARGUMENT;   // run-enumeration path only
  int bpd, tpb;
  transit_geometry(n_cad, n_draw, &bpd, &tpb);
  const Workspace w = carve(workspace, n_draw, bpd, tpb, n_planet);
  if (!workspace || workspace_bytes < w.bytes) return EXO_ERR_WORKSPACE;
  const dim3 block(kBlock);
  const bool secondary = flags & EXO_FLAG_SECONDARY;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  launch_windows(params, n_draw, n_planet, flags, w.windows, st);
  const ScanPlan sp = scan_plan(flags, bpd, n_draw, n_planet, n_texp, true);
  EXO_LAUNCH_SCAN(n_cad, t, has_ttv, flags, sp.grid, block, 0, st, t, n_cad, texp, n_texp, stencil_dt, n_sub, params,
                  n_planet, sp.flags, tpb, bpd, n_draw, sp.n_classify, flux, w.counts, w.list, w.windows, ttv);
  if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
  const int merge = heavy_merge(n_draw, bpd);
  const dim3 hgrid((unsigned)((bpd + merge - 1) / merge), (unsigned)n_draw);
  const double* hwin = ((flags & EXO_FLAG_EXACT_SCAN) && !(flags & EXO_FLAG_WINDOW)) ? nullptr : w.windows;
#define EXO_LAUNCH_HEAVY_FWD(SEC, TTV)                                                                             \
  hipLaunchKernelGGL((transit_heavy_kernel<false, SEC, TTV>), hgrid, block, 0, st, t, n_cad, texp, n_texp,         \
                     stencil_dt, stencil_w, n_sub, params, ld, n_planet, flags, tpb, bpd, merge, w.counts, w.list, \
                     nullptr, flux, nullptr, hwin, ttv)
  if (has_ttv) {
    if (secondary) EXO_LAUNCH_HEAVY_FWD(true, true);
    else EXO_LAUNCH_HEAVY_FWD(false, true);
  } else {
    if (secondary) EXO_LAUNCH_HEAVY_FWD(true, false);
    else EXO_LAUNCH_HEAVY_FWD(false, false);
  }
#undef EXO_LAUNCH_HEAVY_FWD
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  return launch_status();
}

// value + VJP sweep; ttv.edges == nullptr: no timing variations
static int transit_vjp(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                       const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                       int64_t n_draw, int32_t n_planet, uint32_t flags, const Ttv& ttv, const double* gflux,
                       double* flux_out, double* gparams, double* gld, double* flux_dot, void* workspace,
                       int64_t workspace_bytes, void* stream, void* ev_start, void* ev_stop) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || !sweep_flags_ok(flags)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!params || !ld || !gparams || !gld || (n_cad > 0 && (!t || !gflux)) ||
      (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  const bool has_ttv = ttv.edges != nullptr;
  const bool secondary = flags & EXO_FLAG_SECONDARY;
  hipStream_t st = (hipStream_t)stream;
  // the bins are accumulated into: start from zero
  if (has_ttv && !exo::zero_fill_async(ttv.gshift, (int64_t)(n_draw * n_planet * (ttv.n_edge + 1)), st))
    return EXO_ERR_LAUNCH;
  if (n_cad == 0) {
    if (!exo::zero_fill_async(gparams, (int64_t)(n_draw * n_planet * EXO_NPAR), st))
      return EXO_ERR_LAUNCH;
    if (flux_dot && !exo::zero_fill_async(flux_dot, (int64_t)(n_draw), st)) return EXO_ERR_LAUNCH;
    return exo::zero_fill_async(gld, (int64_t)(n_draw * (secondary ? 6 : 3)), st)
               ? EXO_OK : EXO_ERR_LAUNCH;
  }
  if (n_planet * kNG + 7 > kBlock) return EXO_ERR_INVALID_ARGUMENT;
  if ((flags & EXO_FLAG_CADENCE_MAJOR) && (flags & EXO_FLAG_PER_PLANET)) return EXO_ERR_INVALID_ARGUMENT;
  if (runs_path(has_ttv, n_texp, flags)) {
    const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
    if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
    if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
    const int rc = launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet,
                                     flags, gflux, flux_out, gparams, gld, flux_dot, rw, st, nullptr, &ttv);
    if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
    return rc;
  }
  if (flags & (EXO_FLAG_SPARSE | EXO_FLAG_LIGHT_DELAY | EXO_FLAG_CADENCE_MAJOR)) return EXO_ERR_INVALID_ARGUMENT;   // run-enumeration path only
  int bpd, tpb;
  transit_geometry(n_cad, n_draw, &bpd, &tpb);
  const Workspace w = carve(workspace, n_draw, bpd, tpb, n_planet);
  if (!workspace || workspace_bytes < w.bytes) return EXO_ERR_WORKSPACE;
  const dim3 block(kBlock);
  // the forward value is a by-product; without a destination the scan kernel skips the fill
  double* flux_dst = flux_out;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  launch_windows(params, n_draw, n_planet, flags, w.windows, st);
  const ScanPlan sp = scan_plan(flags, bpd, n_draw, n_planet, n_texp, flux_dst != nullptr);
  EXO_LAUNCH_SCAN(n_cad, t, has_ttv, flags, sp.grid, block, 0, st, t, n_cad, texp, n_texp, stencil_dt, n_sub, params,
                  n_planet, sp.flags, tpb, bpd, n_draw, sp.n_classify, flux_dst, w.counts, w.list, w.windows, ttv);
  if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
  const int merge = heavy_merge(n_draw, bpd);
  const int nhb = (bpd + merge - 1) / merge;
  const double* hwin = ((flags & EXO_FLAG_EXACT_SCAN) && !(flags & EXO_FLAG_WINDOW)) ? nullptr : w.windows;
  const dim3 hgrid((unsigned)nhb, (unsigned)n_draw);
#define EXO_LAUNCH_HEAVY_VJP(SEC, TTV)                                                                             \
  hipLaunchKernelGGL((transit_heavy_kernel<true, SEC, TTV>), hgrid, block, 0, st, t, n_cad, texp, n_texp,          \
                     stencil_dt, stencil_w, n_sub, params, ld, n_planet, flags, tpb, bpd, merge, w.counts, w.list, \
                     gflux, flux_dst, w.partial, hwin, ttv)
  if (has_ttv) {
    if (secondary) EXO_LAUNCH_HEAVY_VJP(true, true);
    else EXO_LAUNCH_HEAVY_VJP(false, true);
  } else {
    if (secondary) EXO_LAUNCH_HEAVY_VJP(true, false);
    else EXO_LAUNCH_HEAVY_VJP(false, false);
  }
#undef EXO_LAUNCH_HEAVY_VJP
  if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
  hipLaunchKernelGGL(transit_vjp_reduce_kernel, dim3((unsigned)n_draw), dim3(kBlock), 0, st, w.partial, nhb,
                     n_planet, secondary, gparams, gld, flux_dot);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  return launch_status();
}

static bool ttv_args_ok(const double* edges, const double* shift, int32_t n_edge) {
  return edges && shift && n_edge >= 1 && n_edge <= EXO_MAX_TTV_EDGES;
}

int exo_transit_flux_fwd_ev_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                                uint32_t flags, double* flux, void* workspace, int64_t workspace_bytes,
                                void* stream, void* ev_start, void* ev_stop) {
  return transit_fwd(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags,
                     Ttv{nullptr, nullptr, nullptr, 0}, flux, workspace, workspace_bytes, stream, ev_start, ev_stop);
}

int exo_transit_flux_fwd_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                             const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                             const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, double* flux, void* workspace, int64_t workspace_bytes, void* stream) {
  return exo_transit_flux_fwd_ev_f64(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw,
                                     n_planet, flags, flux, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int exo_transit_flux_ttv_fwd_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                 const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                 const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                                 uint32_t flags, const double* ttv_edges, const double* ttv_shift, int32_t n_edge,
                                 double* flux, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!ttv_args_ok(ttv_edges, ttv_shift, n_edge)) return EXO_ERR_INVALID_ARGUMENT;
  return transit_fwd(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags,
                     Ttv{ttv_edges, ttv_shift, nullptr, n_edge}, flux, workspace, workspace_bytes, stream, nullptr,
                     nullptr);
}

int exo_transit_flux_vjp_ev_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                                uint32_t flags, const double* gflux, double* flux_out, double* gparams,
                                double* gld, double* flux_dot, void* workspace, int64_t workspace_bytes,
                                void* stream, void* ev_start, void* ev_stop) {
  return transit_vjp(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags,
                     Ttv{nullptr, nullptr, nullptr, 0}, gflux, flux_out, gparams, gld, flux_dot, workspace,
                     workspace_bytes, stream, ev_start, ev_stop);
}

int exo_transit_flux_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                             const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                             const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, const double* gflux, double* flux_out, double* gparams,
                             double* gld, double* flux_dot, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  return exo_transit_flux_vjp_ev_f64(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw,
                                     n_planet, flags, gflux, flux_out, gparams, gld, flux_dot, workspace,
                                     workspace_bytes, stream, nullptr, nullptr);
}

int exo_transit_flux_ttv_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                 const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                 const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                                 uint32_t flags, const double* ttv_edges, const double* ttv_shift, int32_t n_edge,
                                 const double* gflux, double* flux_out, double* gparams, double* gld,
                                 double* gshift, double* flux_dot, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  if (!ttv_args_ok(ttv_edges, ttv_shift, n_edge) || !gshift) return EXO_ERR_INVALID_ARGUMENT;
  return transit_vjp(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags,
                     Ttv{ttv_edges, ttv_shift, gshift, n_edge}, gflux, flux_out, gparams, gld, flux_dot, workspace,
                     workspace_bytes, stream, nullptr, nullptr);
}

int64_t exo_transit_flux_jac_doubles(int64_t n_cad, int64_t n_draw, int32_t n_planet) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return -1;
  return (int64_t)kJac * n_cad * n_draw * n_planet;
}

int exo_transit_flux_fwd_jac_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                                 const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                                 int64_t n_draw, int32_t n_planet, uint32_t flags, double* flux, double* jac,
                                 int64_t jac_doubles, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || !sweep_flags_ok(flags)) return EXO_ERR_INVALID_ARGUMENT;
  if (flags & (EXO_FLAG_PER_PLANET | EXO_FLAG_EXACT_SCAN | EXO_FLAG_LIGHT_DELAY)) return EXO_ERR_INVALID_ARGUMENT;
  if (!runs_path(false, n_texp, flags)) return EXO_ERR_INVALID_ARGUMENT;   // one exposure time (or none) for all cadences
  if (n_cad == 0 || n_draw == 0) return EXO_OK;
  if (!t || !params || !ld || (!flux && !(flags & EXO_FLAG_SPARSE)) || !jac || (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (jac_doubles < exo_transit_flux_jac_doubles(n_cad, n_draw, n_planet)) return EXO_ERR_WORKSPACE;
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  return launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, nullptr,
                           flux, nullptr, nullptr, nullptr, rw, (hipStream_t)stream, nullptr, nullptr, jac);
}

int exo_transit_flux_jac_vjp_f64(const double* gflux, int64_t n_cad, int64_t n_draw, int32_t n_planet, uint32_t flags,
                                 const double* jac, void* workspace, int64_t workspace_bytes, double* gparams, double* gld,
                                 double* flux_dot, void* stream) {
  if (n_cad < 0 || n_draw < 0 || n_draw > 65535 || n_planet < 1 || n_planet > EXO_MAX_PLANETS || !sweep_flags_ok(flags))
    return EXO_ERR_INVALID_ARGUMENT;
  if (flags & (EXO_FLAG_PER_PLANET | EXO_FLAG_EXACT_SCAN | EXO_FLAG_LIGHT_DELAY)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!gparams || !gld || (n_cad > 0 && (!gflux || !jac))) return EXO_ERR_INVALID_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  const bool secondary = flags & EXO_FLAG_SECONDARY;
  if (n_cad == 0) {
    if (!exo::zero_fill_async(gparams, (int64_t)(n_draw * n_planet * EXO_NPAR), st)) return EXO_ERR_LAUNCH;
    if (flux_dot && !exo::zero_fill_async(flux_dot, (int64_t)(n_draw), st)) return EXO_ERR_LAUNCH;
    return exo::zero_fill_async(gld, (int64_t)(n_draw * (secondary ? 6 : 3)), st) ? EXO_OK : EXO_ERR_LAUNCH;
  }
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);   // the workspace of the forward call: runs, values, cadence index
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  const int n_ev = secondary ? 2 : 1;
  hipLaunchKernelGGL(transit_jac_vjp_kernel, dim3((unsigned)rw.hb, (unsigned)n_draw), dim3(kBlock), 0, st, n_cad, (int)n_planet, n_ev,
                     flags, rw.rl, rw.vals, rw.vcad, jac, gflux, n_draw, rw.partial);
  hipLaunchKernelGGL(transit_finish_kernel, dim3((unsigned)n_draw), dim3(kBlock), 0, st, rw.partial, rw.hb, (int)n_planet, secondary,
                     gparams, gld, flux_dot, n_cad, flags & ~(uint32_t)(EXO_FLAG_CADENCE_MAJOR | EXO_FLAG_SPARSE), n_ev, rw.rl, nullptr, nullptr, nullptr,
                     nullptr, 0, nullptr, Ttv{nullptr, nullptr, nullptr, 0});
  return launch_status();
}

// columns in (as exo_pack_records_cols_f64), records + sweep + (optionally) column cotangents out
int exo_transit_flux_cols_vjp_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                                  const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                                  uint32_t pack_flags, const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                  const double* stencil_dt, const double* stencil_w, int32_t n_sub, int64_t n_draw,
                                  int32_t n_planet, uint32_t flags, const double* gflux, double* flux_out, double* params,
                                  double* ld, double* gparams, double* gld, double* flux_dot, int32_t fold,
                                  const double* gscale, double* const* gcols, double* const* gld_cols, void* workspace,
                                  int64_t workspace_bytes, void* stream, void* ev_start, void* ev_stop) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || !sweep_flags_ok(flags)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!cols || !draw_stride || !planet_stride || !defaults || !ld_cols || !ld_draw_stride || !params || !ld || !gparams || !gld ||
      (fold && (!gcols || !gld_cols)) || (n_cad > 0 && (!t || !gflux)) || (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n_planet * kNG + 7 > kBlock) return EXO_ERR_INVALID_ARGUMENT;
  if ((flags & EXO_FLAG_CADENCE_MAJOR) && (flags & EXO_FLAG_PER_PLANET)) return EXO_ERR_INVALID_ARGUMENT;
  if (((pack_flags ^ flags) & EXO_FLAG_SECONDARY) != 0) return EXO_ERR_INVALID_ARGUMENT;   // (one answer to "occultations?")
  hipStream_t st = (hipStream_t)stream;
  // the launches fuse when the sweep is a run-enumeration sweep on sorted times (the caller's word: EXO_FLAG_SORTED_TIMES); anything
  // else is the three calls one after the other -- same results
  const bool fused = n_cad > 0 && runs_path(false, n_texp, flags) && (flags & EXO_FLAG_SORTED_TIMES);
  if (!fused) {
    int rc = exo_pack_records_cols_f64(cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride, n_draw, n_planet,
                                       pack_flags, params, ld, stream);
    if (rc != EXO_OK) return rc;
    rc = transit_vjp(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags,
                     Ttv{nullptr, nullptr, nullptr, 0}, gflux, flux_out, gparams, gld, flux_dot, workspace, workspace_bytes, stream,
                     ev_start, ev_stop);
    if (rc != EXO_OK || !fold) return rc;
    return exo_pack_records_cols_vjp_f64(cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride, n_draw, n_planet,
                                         pack_flags, gparams, gld, gscale, gcols, gld_cols, stream);
  }
  PackIn pk{};
  const int nld = (pack_flags & EXO_FLAG_SECONDARY) ? 4 : 2;
  for (int k = 0; k < EXO_NIN; ++k) {
    pk.src.ptr[k] = cols[k]; pk.src.ds[k] = draw_stride[k]; pk.src.ps[k] = planet_stride[k]; pk.src.def[k] = defaults[k];
  }
  for (int k = 0; k < 4; ++k) {
    pk.src.ldp[k] = k < nld ? ld_cols[k] : nullptr;
    pk.src.lds[k] = k < nld ? ld_draw_stride[k] : 0;
    if (k < nld && !ld_cols[k]) return EXO_ERR_INVALID_ARGUMENT;
  }
  pk.flags = pack_flags; pk.n_planet = n_planet; pk.params = params; pk.ld = ld;
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  const int rc = launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, gflux,
                                   flux_out, gparams, gld, flux_dot, rw, st, nullptr, nullptr, nullptr, nullptr, false, &pk);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  if (rc != EXO_OK || !fold) return rc;
  return exo_pack_records_cols_vjp_f64(cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride, n_draw, n_planet,
                                       pack_flags, gparams, gld, gscale, gcols, gld_cols, stream);
}

int exo_transit_flux_vjp_sparse_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                                    const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                                    int64_t n_draw, int32_t n_planet, uint32_t flags, const double* gvals, double* gparams,
                                    double* gld, double* flux_dot, void* workspace, int64_t workspace_bytes, int32_t reuse_runs,
                                    void* stream) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || !sweep_flags_ok(flags)) return EXO_ERR_INVALID_ARGUMENT;
  if (!(flags & EXO_FLAG_SPARSE) || (flags & (EXO_FLAG_PER_PLANET | EXO_FLAG_CADENCE_MAJOR | EXO_FLAG_EXACT_SCAN)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (!runs_path(false, n_texp, flags)) return EXO_ERR_INVALID_ARGUMENT;   // one exposure time (or none) for all cadences
  if (n_draw == 0) return EXO_OK;
  if (!params || !ld || !gparams || !gld || (n_cad > 0 && (!t || !gvals)) || (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n_planet * kNG + 7 > kBlock) return EXO_ERR_INVALID_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  if (n_cad == 0) {
    if (!exo::zero_fill_async(gparams, (int64_t)(n_draw * n_planet * EXO_NPAR), st)) return EXO_ERR_LAUNCH;
    if (flux_dot && !exo::zero_fill_async(flux_dot, (int64_t)(n_draw), st)) return EXO_ERR_LAUNCH;
    return exo::zero_fill_async(gld, (int64_t)(n_draw * ((flags & EXO_FLAG_SECONDARY) ? 6 : 3)), st) ? EXO_OK : EXO_ERR_LAUNCH;
  }
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  return launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, nullptr,
                           nullptr, gparams, gld, flux_dot, rw, st, nullptr, nullptr, nullptr, gvals, reuse_runs != 0);
}

int exo_transit_flux_sparse_model(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                                  uint32_t flags, exo_sparse_model* out) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS || !out || n_cad > 0x7fffffff) return EXO_ERR_INVALID_ARGUMENT;
  const int n_ev = (flags & EXO_FLAG_SECONDARY) ? 2 : 1;
  // one list per draw: the runs ARE the segments (several lists -- planets, occultations -- need the merged form)
  if (n_planet * n_ev != 1) return EXO_ERR_INVALID_ARGUMENT;
  const RunWs rw = carve_runs(const_cast<void*>(workspace), n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  // (lists are packed with stride n_ev -- list = (draw * n_planet + planet) * n_ev + event -- so with one list per draw
  // consecutive draws are consecutive rows, although the workspace is SIZED for two lists per record)
  out->nseg = rw.rl.nrun;
  out->seg = reinterpret_cast<const int32_t*>(rw.rl.runs);
  out->off = rw.rl.pre_all;
  out->vals = rw.vals;
  out->seg_step = 4; out->hi_at = 3;
  out->seg_row = (int64_t)n_planet * n_ev * rw.rl.r_max * 4;
  out->off_row = (int64_t)n_planet * n_ev * (rw.rl.r_max + 1);
  out->val_row = (int64_t)n_planet * n_cad;
  out->row_of_draw = nullptr;
  return EXO_OK;
}

// ---- the MERGED sparse model (round 6): several lists per draw -- planets, occultations -- as one ascending list of disjoint
// segments with the SUM of the lists' values: what exo_transit_flux_sparse_model refuses.  See include/exoplanet_amd.h.
int64_t exo_sparse_merge_workspace_bytes(int64_t n_cad, int64_t n_draw, int32_t n_planet) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return -1;
  return carve_merge(nullptr, n_cad, n_draw, n_planet).bytes;
}

int exo_sparse_merge_layout(int64_t n_cad, int64_t n_draw, int32_t n_planet, int64_t* out) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS || !out) return EXO_ERR_INVALID_ARGUMENT;
  const MergeWs m = carve_merge(nullptr, n_cad, n_draw, n_planet);
  out[0] = m.off_nseg; out[1] = m.off_seg; out[2] = m.off_off; out[3] = m.off_vals; out[4] = m.cap_seg;
  return EXO_OK;
}

static int merge_args(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                      uint32_t flags, const void* merge_ws, int64_t merge_ws_bytes, RunWs* rw, MergeWs* m) {
  if (n_cad < 0 || n_cad >= ((int64_t)1 << 31) || n_draw < 0 || n_draw > 65535 || n_planet < 1 || n_planet > EXO_MAX_PLANETS ||
      (flags & ~(uint32_t)EXO_FLAG_SECONDARY))
    return EXO_ERR_INVALID_ARGUMENT;
  *rw = carve_runs(const_cast<void*>(workspace), n_cad, n_draw, n_planet);
  *m = carve_merge(const_cast<void*>(merge_ws), n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw->bytes || !merge_ws || merge_ws_bytes < m->bytes) return EXO_ERR_WORKSPACE;
  return EXO_OK;
}

static void describe_merged(const MergeWs& m, int64_t n_cad, exo_sparse_model* out) {
  out->nseg = m.nseg;
  out->seg = m.seg;
  out->off = m.off;
  out->vals = m.vals;
  out->seg_step = 2; out->hi_at = 1;
  out->seg_row = (int64_t)m.cap_seg * 2;
  out->off_row = (int64_t)m.cap_seg + 1;
  out->val_row = n_cad;
  out->row_of_draw = nullptr;
}

int exo_sparse_model_merged(const void* merge_ws, int64_t merge_ws_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                            exo_sparse_model* out) {
  if (n_cad < 0 || n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS || !out) return EXO_ERR_INVALID_ARGUMENT;
  const MergeWs m = carve_merge(const_cast<void*>(merge_ws), n_cad, n_draw, n_planet);
  if (!merge_ws || merge_ws_bytes < m.bytes) return EXO_ERR_WORKSPACE;
  describe_merged(m, n_cad, out);
  return EXO_OK;
}

int exo_sparse_model_merge_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                               uint32_t flags, void* merge_ws, int64_t merge_ws_bytes, exo_sparse_model* out, void* stream) {
  RunWs rw;
  MergeWs m;
  const int rc = merge_args(workspace, workspace_bytes, n_cad, n_draw, n_planet, flags, merge_ws, merge_ws_bytes, &rw, &m);
  if (rc != EXO_OK) return rc;
  if (out) describe_merged(m, n_cad, out);
  if (n_cad == 0 || n_draw == 0) return EXO_OK;
  const int n_ev = (flags & EXO_FLAG_SECONDARY) ? 2 : 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sparse_merge_segments_kernel, dim3((unsigned)n_draw), dim3(kBlock), 0, st, rw.rl, (int)n_planet, n_ev, n_cad, m);
  hipLaunchKernelGGL(sparse_merge_values_kernel, dim3((unsigned)merge_blocks_per_draw(n_draw), (unsigned)n_draw), dim3(kBlock), 0, st,
                     rw.rl, rw.vals, (int)n_planet, n_ev, n_cad, m);
  return launch_status();
}

int exo_sparse_model_merge_vjp_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                                   uint32_t flags, const void* merge_ws, int64_t merge_ws_bytes, const double* gmvals,
                                   double* gvals, void* stream) {
  RunWs rw;
  MergeWs m;
  const int rc = merge_args(workspace, workspace_bytes, n_cad, n_draw, n_planet, flags, merge_ws, merge_ws_bytes, &rw, &m);
  if (rc != EXO_OK) return rc;
  if (n_cad == 0 || n_draw == 0) return EXO_OK;
  if (!gmvals || !gvals) return EXO_ERR_INVALID_ARGUMENT;
  const int n_ev = (flags & EXO_FLAG_SECONDARY) ? 2 : 1;
  hipLaunchKernelGGL(sparse_merge_vjp_kernel, dim3((unsigned)merge_blocks_per_draw(n_draw), (unsigned)n_draw), dim3(kBlock), 0,
                     (hipStream_t)stream, rw.rl, (int)n_planet, n_ev, n_cad, m, gmvals, gvals);
  return launch_status();
}

int exo_transit_chi2_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                             const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                             int64_t n_draw, int32_t n_planet, uint32_t flags, const double* obs, const double* ivar,
                             int64_t n_ivar, double* chi2, double* gparams, double* gld, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || (n_ivar != 1 && n_ivar != n_cad) || !sweep_flags_ok(flags))
    return EXO_ERR_INVALID_ARGUMENT;
  if (flags & (EXO_FLAG_PER_PLANET | EXO_FLAG_SPARSE | EXO_FLAG_EXACT_SCAN)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!params || !ld || !chi2 || !gparams || !gld || (n_cad > 0 && (!t || !obs || !ivar)) ||
      (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (!runs_path(false, n_texp, flags)) return EXO_ERR_INVALID_ARGUMENT;   // one exposure time (or none) for all cadences
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  const Chi2Args c2{obs, ivar, n_ivar, chi2};
  return launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, nullptr,
                           nullptr, gparams, gld, nullptr, rw, (hipStream_t)stream, &c2);
}

int exo_transit_chi2_ttv_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                                 const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                                 int64_t n_draw, int32_t n_planet, uint32_t flags, const double* ttv_edges,
                                 const double* ttv_shift, int32_t n_edge, const double* obs, const double* ivar,
                                 int64_t n_ivar, double* chi2, double* gparams, double* gld, double* gshift, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (!transit_args_ok(n_cad, n_texp, n_sub, n_draw, n_planet) || (n_ivar != 1 && n_ivar != n_cad) || n_cad < 1 ||
      !sweep_flags_ok(flags))
    return EXO_ERR_INVALID_ARGUMENT;
  if (flags & (EXO_FLAG_PER_PLANET | EXO_FLAG_SPARSE | EXO_FLAG_EXACT_SCAN | EXO_FLAG_SECONDARY | EXO_FLAG_LIGHT_DELAY))
    return EXO_ERR_INVALID_ARGUMENT;
  if (!ttv_args_ok(ttv_edges, ttv_shift, n_edge) || !gshift) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!params || !ld || !chi2 || !gparams || !gld || !t || !obs || !ivar || (n_texp > 0 && (!texp || !stencil_dt || !stencil_w)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (!runs_path(true, n_texp, flags)) return EXO_ERR_INVALID_ARGUMENT;   // one exposure time (or none) for all cadences
  const RunWs rw = carve_runs(workspace, n_cad, n_draw, n_planet);
  if (!workspace || workspace_bytes < rw.bytes) return EXO_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (!exo::zero_fill_async(gshift, (int64_t)(n_draw * n_planet * (n_edge + 1)), st)) return EXO_ERR_LAUNCH;
  const Chi2Args c2{obs, ivar, n_ivar, chi2};
  const Ttv ttv{ttv_edges, ttv_shift, gshift, n_edge};
  return launch_runs_sweep(t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, nullptr,
                           nullptr, gparams, gld, nullptr, rw, st, &c2, &ttv);
}

}  // extern "C"
