// exo_pack.hip -- the O(planets) parameter algebra of KeplerianOrbit.__init__ as
// ONE kernel (and one reverse kernel), for the standard transit parameterisation
//   (period, t0, b, ecc, omega, r, m_star, r_star, m_planet[, sbr]; u1, u2)
// -> the per-(draw, planet) records of the fused light-curve kernels.
//
// Restates (all under /root/reference/src/exoplanet):
//   orbits/keplerian.py:925-928   a = (G (m_star+m_planet) P^2 / 4 pi^2)^(1/3)
//   orbits/keplerian.py:146       n = 2 pi / P
//   orbits/keplerian.py:205-210   E0 = 2 atan2(sqrt(1-e) cos w, sqrt(1+e)(1+sin w)),  M0 = E0 - e sin E0
//   orbits/keplerian.py:212-228   incl_factor, cos i = incl_factor R_star / a * b
//   orbits/keplerian.py:277-281   t_periastron = t0 - M0 / n ;  sin i
//   orbits/keplerian.py:733-763   in-transit window (circular closed form / contact points)
//   orbits/keplerian.py:779-804   _flip for the occultation window
//   light_curves/limb_dark.py:11-18   get_cl
//   light_curves/secondary_eclipse.py:67-68   flux_ratio = sbr k^2
// In torch this is ~170 launch-bound elementwise kernels (forward + autograd) per
// leapfrog step; here it is two.  One (draw, planet) per lane.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"
#include "exo_contact.hpp"
#include "exo_pack_core.hpp"

namespace {

using namespace exo_pack;

template <class Src>
__global__ __launch_bounds__(64) void pack_kernel(Src src, int64_t n_draw, int n_planet, uint32_t flags,
                                                  double* __restrict__ params, double* __restrict__ ld) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n_draw * n_planet) {
    double o[EXO_NPAR];
    pack_record(src, i, i / n_planet, (int)(i % n_planet), flags, o);
#pragma unroll
    for (int k = 0; k < EXO_NPAR; ++k) params[i * EXO_NPAR + k] = o[k];
  }
  if (i < n_draw) pack_ld(src, i, flags, ld + i * ((flags & EXO_FLAG_SECONDARY) ? 6 : 3));
}

// gscale (optional, per draw): the record cotangents are multiplied by it as they are read -- the chain rule through
// a per-draw scalar (L[d] = sum_n gbar[d, n] flux[d, n], cotangent gL[d]) without a pass of its own
template <class Src, class Dst>
__global__ __launch_bounds__(64) void pack_vjp_kernel(Src src, int64_t n_draw, int n_planet, uint32_t flags,
                                                      const double* __restrict__ gparams,
                                                      const double* __restrict__ gld,
                                                      const double* __restrict__ gscale, Dst dst) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n_draw * n_planet)
    pack_vjp_record(src, i, i / n_planet, (int)(i % n_planet), flags, gparams + i * EXO_NPAR, gscale ? gscale[i / n_planet] : 1.0, dst);
  if (i < n_draw)
    pack_vjp_ld(src, i, flags, gld + i * ((flags & EXO_FLAG_SECONDARY) ? 6 : 3), gscale ? gscale[i] : 1.0, dst);
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH; }

}  // namespace

extern "C" {

int exo_pack_records_f64(const double* orbit_in, const double* ld_in, int64_t n_draw, int32_t n_planet,
                         uint32_t flags, double* params, double* ld, void* stream) {
  if (n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!orbit_in || !ld_in || !params || !ld) return EXO_ERR_INVALID_ARGUMENT;
  const int64_t n = n_draw * n_planet;
  hipLaunchKernelGGL(pack_kernel<PackedSrc>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     PackedSrc{orbit_in, ld_in}, n_draw, n_planet, flags, params, ld);
  return launch_status();
}

int exo_pack_records_vjp_f64(const double* orbit_in, const double* ld_in, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, const double* gparams, const double* gld, double* gorbit_in,
                             double* gld_in, void* stream) {
  if (n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!orbit_in || !ld_in || !gparams || !gld || !gorbit_in || !gld_in) return EXO_ERR_INVALID_ARGUMENT;
  const int64_t n = n_draw * n_planet;
  hipLaunchKernelGGL((pack_vjp_kernel<PackedSrc, PackedDst>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, PackedSrc{orbit_in, ld_in}, n_draw, n_planet, flags, gparams, gld,
                     (const double*)nullptr, PackedDst{gorbit_in, gld_in});
  return launch_status();
}

static bool cols_src(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                     const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride, uint32_t flags,
                     ColsSrc* s) {
  if (!cols || !draw_stride || !planet_stride || !defaults || !ld_cols || !ld_draw_stride) return false;
  for (int k = 0; k < EXO_NIN; ++k) {
    s->ptr[k] = cols[k]; s->ds[k] = draw_stride[k]; s->ps[k] = planet_stride[k]; s->def[k] = defaults[k];
  }
  const int nld = (flags & EXO_FLAG_SECONDARY) ? 4 : 2;
  for (int k = 0; k < 4; ++k) {
    s->ldp[k] = k < nld ? ld_cols[k] : nullptr;
    s->lds[k] = k < nld ? ld_draw_stride[k] : 0;
    if (k < nld && !ld_cols[k]) return false;
  }
  return true;
}

int exo_pack_records_cols_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                              const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                              int64_t n_draw, int32_t n_planet, uint32_t flags, double* params, double* ld, void* stream) {
  if (n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  ColsSrc src;
  if (!params || !ld || !cols_src(cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride, flags, &src))
    return EXO_ERR_INVALID_ARGUMENT;
  const int64_t n = n_draw * n_planet;
  hipLaunchKernelGGL(pack_kernel<ColsSrc>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src, n_draw,
                     n_planet, flags, params, ld);
  return launch_status();
}

int exo_pack_records_cols_vjp_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                                  const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                                  int64_t n_draw, int32_t n_planet, uint32_t flags, const double* gparams,
                                  const double* gld, const double* gscale, double* const* gcols, double* const* gld_cols,
                                  void* stream) {
  if (n_draw < 0 || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  ColsSrc src;
  if (!gparams || !gld || !gcols || !gld_cols ||
      !cols_src(cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride, flags, &src))
    return EXO_ERR_INVALID_ARGUMENT;
  ColsDst dst;
  for (int k = 0; k < EXO_NIN; ++k) dst.ptr[k] = gcols[k];
  const int nld = (flags & EXO_FLAG_SECONDARY) ? 4 : 2;
  for (int k = 0; k < 4; ++k) dst.ldp[k] = k < nld ? gld_cols[k] : nullptr;
  const int64_t n = n_draw * n_planet;
  hipLaunchKernelGGL((pack_vjp_kernel<ColsSrc, ColsDst>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, src, n_draw, n_planet, flags, gparams, gld, gscale, dst);
  return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// SHO term -> celerite pair-slot coefficients, one lane per (draw, term): celerite2's SHOTerm parameterisations
// (S0 | sigma, w0 | rho, Q | tau) -> (S0, w0, Q) -> the slot's four doubles and its kind, and the reverse.  In torch
// this is ~35 launch-bound elementwise kernels per term and step (forward + autograd); a three-term kernel spent
// 0.6 ms of a 3.7 ms C5 step in them.
//   w0 = 2 pi / rho;  Q = w0 tau / 2;  S0 = sigma^2 / (w0 Q);   a = S0 w0 Q,  c = w0 / (2 Q)
//   Q <  1/2 (kind 1, two real terms):  f = sqrt(max(1 - 4 Q^2, eps)):  (a (1 + 1/f) / 2, c (1 - f), a (1 - 1/f) / 2, c (1 + f))
//   Q >= 1/2 (kind 0, one complex term): f = sqrt(max(4 Q^2 - 1, eps)):  (a, a / f, c, c f)
// (a clamped f carries no gradient, as torch.clamp)
// ---------------------------------------------------------------------------------------------
namespace {

struct ShoDerived {
  double S0, w0, Q, a, c, f;
  bool over, clamped;
};
__device__ __forceinline__ ShoDerived sho_derive(double amp, double freq, double damp, uint32_t flags, double eps) {
  ShoDerived d;
  d.w0 = (flags & EXO_SHO_RHO) ? 2.0 * kPi / freq : freq;
  d.Q = (flags & EXO_SHO_TAU) ? 0.5 * d.w0 * damp : damp;
  d.S0 = (flags & EXO_SHO_SIGMA) ? amp * amp / (d.w0 * d.Q) : amp;
  d.over = d.Q < 0.5;
  d.a = d.S0 * d.w0 * d.Q;
  d.c = 0.5 * d.w0 / d.Q;
  const double x = d.over ? 1.0 - 4.0 * d.Q * d.Q : 4.0 * d.Q * d.Q - 1.0;
  d.clamped = !(x > eps);
  d.f = sqrt(d.clamped ? eps : x);
  return d;
}

// the terms of a sum, by value in the kernel arguments
struct ShoTerms {
  const double* amp[EXO_SHO_MAX_TERMS];
  const double* freq[EXO_SHO_MAX_TERMS];
  const double* damp[EXO_SHO_MAX_TERMS];
  double* gamp[EXO_SHO_MAX_TERMS];
  double* gfreq[EXO_SHO_MAX_TERMS];
  double* gdamp[EXO_SHO_MAX_TERMS];
  uint32_t flags[EXO_SHO_MAX_TERMS];
  int32_t n_terms;
};

// one term, one element: coefficients + kind; and the reverse
__device__ __forceinline__ void sho_coef_one(double am, double fr, double da, uint32_t flags, double eps, double* o, int32_t* kd) {
  const ShoDerived d = sho_derive(am, fr, da, flags, eps);
  if (d.over) {
    o[0] = 0.5 * d.a * (1.0 + 1.0 / d.f); o[1] = d.c * (1.0 - d.f); o[2] = 0.5 * d.a * (1.0 - 1.0 / d.f); o[3] = d.c * (1.0 + d.f);
  } else {
    o[0] = d.a; o[1] = d.a / d.f; o[2] = d.c; o[3] = d.c * d.f;
  }
  *kd = d.over ? 1 : 0;
}
__device__ __forceinline__ void sho_coef_vjp_one(double am, double fr, double da, uint32_t flags, double eps, const double* g,
                                                 double* g_amp_o, double* g_freq_o, double* g_damp_o) {
  const ShoDerived d = sho_derive(am, fr, da, flags, eps);
  const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
  double ga, gc, gf, dfdQ;
  if (d.over) {
    ga = 0.5 * g0 * (1.0 + 1.0 / d.f) + 0.5 * g2 * (1.0 - 1.0 / d.f);
    gc = g1 * (1.0 - d.f) + g3 * (1.0 + d.f);
    gf = 0.5 * d.a * (g2 - g0) / (d.f * d.f) + d.c * (g3 - g1);
    dfdQ = d.clamped ? 0.0 : -4.0 * d.Q / d.f;
  } else {
    ga = g0 + g1 / d.f;
    gc = g2 + g3 * d.f;
    gf = -g1 * d.a / (d.f * d.f) + g3 * d.c;
    dfdQ = d.clamped ? 0.0 : 4.0 * d.Q / d.f;
  }
  double gS0 = ga * d.w0 * d.Q;
  double gw0 = ga * d.S0 * d.Q + gc * 0.5 / d.Q;
  double gQ = ga * d.S0 * d.w0 - gc * 0.5 * d.w0 / (d.Q * d.Q) + gf * dfdQ;
  double g_amp = gS0;
  if (flags & EXO_SHO_SIGMA) {   // S0 = sigma^2 / (w0 Q)
    g_amp = gS0 * 2.0 * am / (d.w0 * d.Q);
    gw0 -= gS0 * d.S0 / d.w0;
    gQ -= gS0 * d.S0 / d.Q;
  }
  double g_damp = gQ;
  if (flags & EXO_SHO_TAU) {     // Q = w0 tau / 2
    g_damp = gQ * 0.5 * d.w0;
    gw0 += gQ * 0.5 * da;
  }
  *g_amp_o = g_amp;
  *g_freq_o = (flags & EXO_SHO_RHO) ? -gw0 * 2.0 * kPi / (fr * fr) : gw0;
  *g_damp_o = g_damp;
}

__global__ __launch_bounds__(64) void sho_coef_kernel(const double* __restrict__ amp, const double* __restrict__ freq,
                                                      const double* __restrict__ damp, uint32_t flags, double eps, int64_t n,
                                                      double* __restrict__ coef, int32_t* __restrict__ kind) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  sho_coef_one(amp[i], freq[i], damp[i], flags, eps, coef + 4 * i, kind + i);
}

__global__ __launch_bounds__(64) void sho_coef_vjp_kernel(const double* __restrict__ amp, const double* __restrict__ freq,
                                                          const double* __restrict__ damp, uint32_t flags, double eps,
                                                          int64_t n, const double* __restrict__ gcoef,
                                                          double* __restrict__ gamp, double* __restrict__ gfreq,
                                                          double* __restrict__ gdamp) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  sho_coef_vjp_one(amp[i], freq[i], damp[i], flags, eps, gcoef + 4 * i, gamp + i, gfreq + i, gdamp + i);
}

// gcoef == nullptr: forward (coef, kind written); else reverse (the terms' gamp / gfreq / gdamp written)
__global__ __launch_bounds__(64) void sho_coef_multi_kernel(ShoTerms s, double eps, int64_t n, double* __restrict__ coef,
                                                            int32_t* __restrict__ kind, const double* __restrict__ gcoef) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n) return;
  // (the term's pointers are picked with selects over compile-time indices: a struct member indexed at run time is a scratch copy)
  const double *pa = nullptr, *pf = nullptr, *pd = nullptr;
  double *ga = nullptr, *gf = nullptr, *gd = nullptr;
  uint32_t fl = 0u;
#pragma unroll
  for (int q = 0; q < EXO_SHO_MAX_TERMS; ++q) {
    if (q == k) { pa = s.amp[q]; pf = s.freq[q]; pd = s.damp[q]; ga = s.gamp[q]; gf = s.gfreq[q]; gd = s.gdamp[q]; fl = s.flags[q]; }
  }
  const int64_t at = i * s.n_terms + k;
  if (gcoef == nullptr) sho_coef_one(pa[i], pf[i], pd[i], fl, eps, coef + 4 * at, kind + at);
  else sho_coef_vjp_one(pa[i], pf[i], pd[i], fl, eps, gcoef + 4 * at, ga + i, gf + i, gd + i);
}

}  // namespace

extern "C" {

int exo_sho_coefficients_f64(const double* amp, const double* freq, const double* damp, uint32_t flags, double eps,
                             int64_t n, double* coef, int32_t* kind, void* stream) {
  if (n < 0 || (flags & ~(EXO_SHO_SIGMA | EXO_SHO_RHO | EXO_SHO_TAU))) return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  if (!amp || !freq || !damp || !coef || !kind) return EXO_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sho_coef_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, amp, freq, damp,
                     flags, eps, n, coef, kind);
  return launch_status();
}

int exo_sho_coefficients_vjp_f64(const double* amp, const double* freq, const double* damp, uint32_t flags, double eps,
                                 int64_t n, const double* gcoef, double* gamp, double* gfreq, double* gdamp,
                                 void* stream) {
  if (n < 0 || (flags & ~(EXO_SHO_SIGMA | EXO_SHO_RHO | EXO_SHO_TAU))) return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  if (!amp || !freq || !damp || !gcoef || !gamp || !gfreq || !gdamp) return EXO_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sho_coef_vjp_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, amp, freq,
                     damp, flags, eps, n, gcoef, gamp, gfreq, gdamp);
  return launch_status();
}

static int sho_multi(const double* const* amp, const double* const* freq, const double* const* damp, const uint32_t* flags,
                     int32_t n_terms, double eps, int64_t n, double* coef, int32_t* kind, const double* gcoef, double* const* gamp,
                     double* const* gfreq, double* const* gdamp, void* stream) {
  if (n < 0 || n_terms < 1 || n_terms > EXO_SHO_MAX_TERMS || !amp || !freq || !damp || !flags) return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  ShoTerms s{};
  s.n_terms = n_terms;
  for (int k = 0; k < n_terms; ++k) {
    if (!amp[k] || !freq[k] || !damp[k] || (flags[k] & ~(EXO_SHO_SIGMA | EXO_SHO_RHO | EXO_SHO_TAU))) return EXO_ERR_INVALID_ARGUMENT;
    s.amp[k] = amp[k]; s.freq[k] = freq[k]; s.damp[k] = damp[k]; s.flags[k] = flags[k];
    if (gcoef) {
      if (!gamp || !gfreq || !gdamp || !gamp[k] || !gfreq[k] || !gdamp[k]) return EXO_ERR_INVALID_ARGUMENT;
      s.gamp[k] = gamp[k]; s.gfreq[k] = gfreq[k]; s.gdamp[k] = gdamp[k];
    }
  }
  if (!gcoef && (!coef || !kind)) return EXO_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sho_coef_multi_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)n_terms), dim3(64), 0, (hipStream_t)stream, s,
                     eps, n, coef, kind, gcoef);
  return launch_status();
}

int exo_sho_coefficients_multi_f64(const double* const* amp, const double* const* freq, const double* const* damp,
                                   const uint32_t* flags, int32_t n_terms, double eps, int64_t n, double* coef, int32_t* kind,
                                   void* stream) {
  return sho_multi(amp, freq, damp, flags, n_terms, eps, n, coef, kind, nullptr, nullptr, nullptr, nullptr, stream);
}

int exo_sho_coefficients_multi_vjp_f64(const double* const* amp, const double* const* freq, const double* const* damp,
                                       const uint32_t* flags, int32_t n_terms, double eps, int64_t n, const double* gcoef,
                                       double* const* gamp, double* const* gfreq, double* const* gdamp, void* stream) {
  if (!gcoef) return EXO_ERR_INVALID_ARGUMENT;
  return sho_multi(amp, freq, damp, flags, n_terms, eps, n, nullptr, nullptr, gcoef, gamp, gfreq, gdamp, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Timing tables of a TTVOrbit whose transits are all labelled and given as offsets from the linear ephemeris
// (reference: orbits/ttv.py:99-170 with `ttvs`): transit k of planet p of a draw happens at
//     tt_k = t0 + period k + ttv_k,        k = 0 .. n_p - 1,
// the bin edges are the midpoints between neighbours closed by half a period either side (ttv.py:158-166, padded
// with +inf to a common width), and bin j belongs to transit max(0, min(j - 1, n_p - 1)) (ttv.py:167-170); the
// kernels take shift = transit time of the bin - t0 = period k + ttv_k (include/exoplanet_amd.h).  In torch this
// was ~25 small kernels forward and ~30 in the reverse pass per step; here one launch each way.
// One thread per (draw, planet, bin).
// ---------------------------------------------------------------------------------------------
namespace {

struct TtvSrc {
  const double* period; int64_t period_ds, period_ps;
  const double* t0; int64_t t0_ds, t0_ps;
  const double* ttv[EXO_MAX_PLANETS]; int64_t ttv_ds[EXO_MAX_PLANETS]; int32_t n[EXO_MAX_PLANETS];
};
struct TtvGradDst {
  double* gttv[EXO_MAX_PLANETS];   // [n_draw][n_p] each, dense (NULL: not wanted)
  double* gperiod;                 // [n_draw][n_planet] dense (NULL: not wanted)
};

__global__ __launch_bounds__(64) void ttv_tables_kernel(TtvSrc s, int64_t n_draw, int n_planet, int n_edge,
                                                        double* __restrict__ edges, double* __restrict__ shift) {
  const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int per = n_edge + 1;                       // bins of one (draw, planet)
  const int64_t rec = gid / per;
  const int j = (int)(gid - rec * per);
  if (rec >= n_draw * n_planet) return;
  const int64_t d = rec / n_planet;
  const int p = (int)(rec - d * n_planet);
  const int n = s.n[p];
  const double P = s.period[d * s.period_ds + p * s.period_ps], t0 = s.t0[d * s.t0_ds + p * s.t0_ps];
  const double* __restrict__ ttv = s.ttv[p] + d * s.ttv_ds[p];
  auto off = [&](int k) { return fma(P, (double)k, ttv[k]); };   // transit time - t0
  const int k = j - 1 < 0 ? 0 : (j - 1 < n ? j - 1 : n - 1);    // the transit bin j belongs to
  shift[rec * per + j] = off(k);
  if (j < n_edge) {
    double e;
    if (j == 0) e = t0 + off(0) - 0.5 * P;
    else if (j < n) e = 0.5 * ((t0 + off(j - 1)) + (t0 + off(j)));
    else if (j == n) e = t0 + off(n - 1) + 0.5 * P;
    else e = __builtin_inf();
    edges[rec * n_edge + j] = e;
  }
}

// reverse: the cotangent of shift back to the offsets and the periods (the edges carry none: searchsorted, ttv.py:174).
// One wave per (draw, planet): a lane per transit (strided), the period's sum over the wave in a fixed order.
__global__ __launch_bounds__(64) void ttv_tables_vjp_kernel(TtvSrc s, int64_t n_draw, int n_planet, int n_edge,
                                                            const double* __restrict__ gshift, TtvGradDst dst) {
  const int64_t rec = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t d = rec / n_planet;
  const int p = (int)(rec - d * n_planet);
  const int n = s.n[p], per = n_edge + 1;
  const double* __restrict__ g = gshift + rec * per;
  // the bins past the last transit belong to it: their sum, lanes strided, combined below
  double tail = 0.0;
  for (int j = n + 1 + lane; j < per; j += 64) tail += g[j];
  double acc = 0.0;
  for (int k = lane; k < n; k += 64) {
    double v = g[k + 1];
    if (k == 0) v += g[0];
    if (dst.gttv[p]) dst.gttv[p][d * n + k] = v;       // (transit n - 1 gets the tail added by its lane below)
    acc = fma((double)k, v, acc);
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    tail += __shfl_xor(tail, m, 64);
    acc += __shfl_xor(acc, m, 64);
  }
  if (lane == (n - 1) % 64 && dst.gttv[p]) dst.gttv[p][d * n + (n - 1)] += tail;
  if (lane == 0 && dst.gperiod) dst.gperiod[rec] = fma((double)(n - 1), tail, acc);
}

static bool ttv_src(const double* period, int64_t period_ds, int64_t period_ps, const double* t0, int64_t t0_ds, int64_t t0_ps,
                    const double* const* ttv, const int64_t* ttv_ds, const int32_t* n_transit, int32_t n_planet,
                    int32_t n_edge, TtvSrc* s) {
  if (!period || !t0 || !ttv || !ttv_ds || !n_transit || n_planet < 1 || n_planet > EXO_MAX_PLANETS) return false;
  s->period = period; s->period_ds = period_ds; s->period_ps = period_ps;
  s->t0 = t0; s->t0_ds = t0_ds; s->t0_ps = t0_ps;
  int width = 0;
  for (int p = 0; p < n_planet; ++p) {
    if (!ttv[p] || n_transit[p] < 1) return false;
    s->ttv[p] = ttv[p]; s->ttv_ds[p] = ttv_ds[p]; s->n[p] = n_transit[p];
    width = n_transit[p] > width ? n_transit[p] : width;
  }
  return n_edge == width + 1;
}

}  // namespace

extern "C" {

int exo_ttv_tables_f64(const double* period, int64_t period_draw_stride, int64_t period_planet_stride, const double* t0,
                       int64_t t0_draw_stride, int64_t t0_planet_stride, const double* const* ttv,
                       const int64_t* ttv_draw_stride, const int32_t* n_transit, int64_t n_draw, int32_t n_planet,
                       int32_t n_edge, double* edges, double* shift, void* stream) {
  if (n_draw < 0) return EXO_ERR_INVALID_ARGUMENT;
  TtvSrc s;
  if (!ttv_src(period, period_draw_stride, period_planet_stride, t0, t0_draw_stride, t0_planet_stride, ttv, ttv_draw_stride,
               n_transit, n_planet, n_edge, &s))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!edges || !shift) return EXO_ERR_INVALID_ARGUMENT;
  const int64_t n = n_draw * n_planet * (int64_t)(n_edge + 1);
  hipLaunchKernelGGL(ttv_tables_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, s, n_draw,
                     (int)n_planet, (int)n_edge, edges, shift);
  return launch_status();
}

int exo_ttv_tables_vjp_f64(const double* period, int64_t period_draw_stride, int64_t period_planet_stride, const double* t0,
                           int64_t t0_draw_stride, int64_t t0_planet_stride, const double* const* ttv,
                           const int64_t* ttv_draw_stride, const int32_t* n_transit, int64_t n_draw, int32_t n_planet,
                           int32_t n_edge, const double* gshift, double* const* gttv, double* gperiod, void* stream) {
  if (n_draw < 0) return EXO_ERR_INVALID_ARGUMENT;
  TtvSrc s;
  if (!ttv_src(period, period_draw_stride, period_planet_stride, t0, t0_draw_stride, t0_planet_stride, ttv, ttv_draw_stride,
               n_transit, n_planet, n_edge, &s))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!gshift || !gttv) return EXO_ERR_INVALID_ARGUMENT;
  TtvGradDst dst;
  for (int p = 0; p < EXO_MAX_PLANETS; ++p) dst.gttv[p] = p < n_planet ? gttv[p] : nullptr;
  dst.gperiod = gperiod;
  hipLaunchKernelGGL(ttv_tables_vjp_kernel, dim3((unsigned)(n_draw * n_planet)), dim3(64), 0, (hipStream_t)stream, s, n_draw,
                     (int)n_planet, (int)n_edge, gshift, dst);
  return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// NUTS on (chains, parameters) arrays that stay on the device (exoplanet_amd/sampling.py: NUTS.step restated -- that
// torch version is what runs on the CPU and what tests/test_gpu_sampling.py checks this against).  A transition grows
// every chain's trajectory by doublings; a doubling is
//   phase 2  begin:  each chain draws its direction, the sub-tree starts at that end of its trajectory
//   per leaf phase 0: half kick and drift of the moving end     p_half = p + eps g / 2,  q' = q + eps p_half / m
//            (the caller evaluates log-density and gradient at q')
//            phase 1: second half kick, energy error, divergence, multinomial candidate, momentum sums, checkpoint
//            write (even leaves) / generalised turning checks (odd leaves), which chains go on
//   phase 3  merge:  a valid sub-tree joins the trajectory (biased progressive sampling of the proposal, the new end,
//            momentum sum, weights, the trajectory's own turning check)
// -- four launches of these kernels where the torch statement is ~35 per leaf and ~45 per doubling.  One thread per
// chain (a chain's row is a few to a few dozen doubles).  The leaf index lives on the device (phase 0 counts it up),
// so that the captured graph of a leaf needs nothing from the host; the random numbers of a doubling are one
// [2 + leaves][chains] array: row 0 directions, row 1 the merge, row 2 + n leaf n.
// ---------------------------------------------------------------------------------------------
namespace {

struct NutsTree {
  // the sub-tree being built
  double *qe, *pe, *ge;              // [D][n]   its moving end
  double* eps;                       // [D]      signed step size
  uint8_t* on;                       // [D]      still adding leaves (torch.bool)
  const double* H0;                  // [D]
  double *logw, *psum, *sq, *sg, *slp;
  uint8_t *turn, *div;
  double *acc, *accn;
  double *ckp, *cks;                 // [S][D][n] checkpoints: momentum of a sub-sub-tree's first leaf, momentum sum up to it
  const double* mass;                // [D][n]
  double *qn, *ph;                   // [D][n]   scratch: position and half-kicked momentum of the new leaf
  const double *gn, *lpn;            // [D][n], [D]  gradient and log-density at qn
  // the trajectory
  double *ql, *pl, *gl, *qr, *pr, *gr, *tsum;   // [D][n] its ends, the sum of its momenta
  double* logW;                      // [D]
  double *propq, *propg, *proplp;    // the proposal
  uint8_t *active, *diverged, *going;
  double* depth;
  const double* eps_abs;             // [D]
  const double* R;                   // [2 + leaves][D] uniform random numbers of the doubling
  int32_t* leaf;                     // [1] index of the current leaf within the sub-tree
  int64_t D;
  int n, S;
  double max_energy_error;
};
constexpr int kNutsPtrs = 40;

__device__ __forceinline__ double log_add_exp(double x, double y) {
  const double m = fmax(x, y);
  if (!(m > -__builtin_inf())) return m;       // both -inf (or NaN)
  return m + log1p(exp(-fabs(x - y)));
}
// generalised U-turn: the ends no longer move apart along rho = sum of momenta - half of each end
__device__ __forceinline__ bool nuts_turning(const double* pl, const double* pr, const double* sum, const double* mass, int n) {
  double left = 0.0, right = 0.0;
  for (int i = 0; i < n; ++i) {
    const double rho = (sum[i] - 0.5 * (pl[i] + pr[i])) / mass[i];
    left = fma(pl[i], rho, left);
    right = fma(pr[i], rho, right);
  }
  return (left <= 0.0) || (right <= 0.0);
}

__global__ __launch_bounds__(64) void nuts_begin_doubling_kernel(NutsTree a) {
  const int64_t d = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (d == 0) *a.leaf = -1;
  if (d >= a.D) return;
  const bool right = a.R[d] < 0.5;
  a.going[d] = right ? 1 : 0;
  a.eps[d] = right ? a.eps_abs[d] : -a.eps_abs[d];
  const int64_t row = d * a.n;
  const double *q = right ? a.qr : a.ql, *p = right ? a.pr : a.pl, *g = right ? a.gr : a.gl;
  for (int i = 0; i < a.n; ++i) {
    a.qe[row + i] = q[row + i]; a.pe[row + i] = p[row + i]; a.ge[row + i] = g[row + i];
    a.sq[row + i] = q[row + i]; a.sg[row + i] = g[row + i];
    a.psum[row + i] = 0.0;
  }
  a.slp[d] = a.proplp[d];
  a.on[d] = a.active[d];
  a.logw[d] = -__builtin_inf();
  a.turn[d] = 0;
  a.div[d] = 0;
}

__global__ __launch_bounds__(64) void nuts_leaf_begin_kernel(NutsTree a) {
  const int64_t d = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (d == 0) *a.leaf += 1;                      // (nothing in this launch reads it)
  if (d >= a.D) return;
  const double e = a.eps[d];
  for (int i = 0; i < a.n; ++i) {
    const int64_t k = d * a.n + i;
    const double ph = fma(0.5 * e, a.ge[k], a.pe[k]);
    a.ph[k] = ph;
    a.qn[k] = fma(e, ph / a.mass[k], a.qe[k]);
  }
}

__global__ __launch_bounds__(64) void nuts_leaf_update_kernel(NutsTree a) {
  const int64_t d = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (d >= a.D || !a.on[d]) return;              // (a chain that has stopped carries its state along unchanged)
  const int n = a.n, leaf = *a.leaf;
  const int64_t row = d * n, plane = a.D * (int64_t)n;
  const double e = a.eps[d];
  double kin = 0.0;
  for (int i = 0; i < n; ++i) {
    const double pn = fma(0.5 * e, a.gn[row + i], a.ph[row + i]);
    kin += 0.5 * pn * pn / a.mass[row + i];
    a.pe[row + i] = pn;
    a.qe[row + i] = a.qn[row + i];
    a.ge[row + i] = a.gn[row + i];
  }
  const double lp = a.lpn[d];
  double dH = -lp + kin - a.H0[d];
  if (dH != dH) dH = __builtin_inf();
  const bool div = dH > a.max_energy_error;
  a.acc[d] += exp(fmin(-dH, 0.0));
  a.accn[d] += 1.0;
  bool turn = false;
  if (!div) {
    // multinomial sampling within the sub-tree: the new leaf replaces the candidate with probability w / W
    const double new_logw = log_add_exp(a.logw[d], -dH);
    if (log(a.R[(int64_t)(2 + leaf) * a.D + d]) < (-dH - new_logw)) {
      for (int i = 0; i < n; ++i) { a.sq[row + i] = a.qn[row + i]; a.sg[row + i] = a.gn[row + i]; }
      a.slp[d] = lp;
    }
    a.logw[d] = new_logw;
    for (int i = 0; i < n; ++i) a.psum[row + i] += a.pe[row + i];
    // Leaf `leaf` of the sub-tree: even -- its momentum and the sum so far go to checkpoint slot popcount(leaf >> 1);
    // odd -- it closes the sub-sub-trees of 2, 4, ... leaves that end here, one per trailing 1 bit, whose first
    // leaves sit in the slots hi, hi - 1, ...
    const int hi = __popc((unsigned)(leaf >> 1));
    if (leaf & 1) {
      const int ones = __ffs(~(unsigned)leaf) - 1;          // trailing 1 bits
      for (int s = hi; s > hi - ones; --s) {
        const double* __restrict__ cp = a.ckp + s * plane + row;
        const double* __restrict__ cs = a.cks + s * plane + row;
        double left = 0.0, right = 0.0;
        for (int i = 0; i < n; ++i) {
          const double pn = a.pe[row + i];
          const double rho = ((a.psum[row + i] - cs[i] + cp[i]) - 0.5 * (cp[i] + pn)) / a.mass[row + i];
          left = fma(cp[i], rho, left);
          right = fma(pn, rho, right);
        }
        turn = turn || (left <= 0.0) || (right <= 0.0);
      }
    } else {
      for (int i = 0; i < n; ++i) {
        a.ckp[hi * plane + row + i] = a.pe[row + i];
        a.cks[hi * plane + row + i] = a.psum[row + i];
      }
    }
  }
  if (turn) a.turn[d] = 1;
  if (div) a.div[d] = 1;
  a.on[d] = (!div && !turn) ? 1 : 0;
}

__global__ __launch_bounds__(64) void nuts_merge_kernel(NutsTree a) {
  const int64_t d = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (d >= a.D || !a.active[d]) return;
  const int n = a.n;
  const int64_t row = d * n;
  a.depth[d] += 1.0;
  if (a.div[d]) a.diverged[d] = 1;
  if (a.turn[d] || a.div[d]) { a.active[d] = 0; return; }      // the sub-tree is not valid: the trajectory ends as it was
  // biased progressive sampling between the old trajectory and the new half
  if (log(a.R[a.D + d]) < (a.logw[d] - a.logW[d])) {
    for (int i = 0; i < n; ++i) { a.propq[row + i] = a.sq[row + i]; a.propg[row + i] = a.sg[row + i]; }
    a.proplp[d] = a.slp[d];
  }
  double *q = a.going[d] ? a.qr : a.ql, *p = a.going[d] ? a.pr : a.pl, *g = a.going[d] ? a.gr : a.gl;
  for (int i = 0; i < n; ++i) {
    q[row + i] = a.qe[row + i]; p[row + i] = a.pe[row + i]; g[row + i] = a.ge[row + i];
    a.tsum[row + i] += a.psum[row + i];
  }
  a.logW[d] = log_add_exp(a.logW[d], a.logw[d]);
  a.active[d] = nuts_turning(a.pl + row, a.pr + row, a.tsum + row, a.mass + row, n) ? 0 : 1;
}

}  // namespace

extern "C" {

// ptrs: kNutsPtrs device pointers in the order of NutsTree's pointer members
int exo_nuts_f64(const void* const* ptrs, int64_t n_chain, int32_t n_param, int32_t n_slot, double max_energy_error,
                 int32_t phase, void* stream) {
  if (!ptrs || n_chain < 0 || n_param < 1 || n_slot < 1 || phase < 0 || phase > 3) return EXO_ERR_INVALID_ARGUMENT;
  if (n_chain == 0) return EXO_OK;
  for (int k = 0; k < kNutsPtrs; ++k)
    if (!ptrs[k]) return EXO_ERR_INVALID_ARGUMENT;
  NutsTree a;
  int k = 0;
  auto dp = [&]() { return (double*)ptrs[k++]; };
  auto bp = [&]() { return (uint8_t*)ptrs[k++]; };
  a.qe = dp(); a.pe = dp(); a.ge = dp(); a.eps = dp(); a.on = bp(); a.H0 = dp(); a.logw = dp(); a.psum = dp(); a.sq = dp();
  a.sg = dp(); a.slp = dp(); a.turn = bp(); a.div = bp(); a.acc = dp(); a.accn = dp(); a.ckp = dp(); a.cks = dp();
  a.mass = dp(); a.qn = dp(); a.ph = dp(); a.gn = dp(); a.lpn = dp();
  a.ql = dp(); a.pl = dp(); a.gl = dp(); a.qr = dp(); a.pr = dp(); a.gr = dp(); a.tsum = dp(); a.logW = dp();
  a.propq = dp(); a.propg = dp(); a.proplp = dp(); a.active = bp(); a.diverged = bp(); a.going = bp(); a.depth = dp();
  a.eps_abs = dp(); a.R = dp(); a.leaf = (int32_t*)ptrs[k++];
  a.D = n_chain; a.n = n_param; a.S = n_slot; a.max_energy_error = max_energy_error;
  const dim3 grid((unsigned)((n_chain + 63) / 64)), block(64);
  hipStream_t st = (hipStream_t)stream;
  if (phase == 0) hipLaunchKernelGGL(nuts_leaf_begin_kernel, grid, block, 0, st, a);
  else if (phase == 1) hipLaunchKernelGGL(nuts_leaf_update_kernel, grid, block, 0, st, a);
  else if (phase == 2) hipLaunchKernelGGL(nuts_begin_doubling_kernel, grid, block, 0, st, a);
  else hipLaunchKernelGGL(nuts_merge_kernel, grid, block, 0, st, a);
  return launch_status();
}

}  // extern "C"
