// exo_rv.hip -- stellar reflex radial velocity from the same Kepler solve (gfx950).
//
// Reference (all under /root/reference/src/exoplanet/orbits/keplerian.py):
//   :633-677  get_radial_velocity: with K,   K (cos w cos f - sin w sin f + e cos w)   (:660-669)
//                                  circular: K cos f                                    (:658-659)
//             without K:           -conv * z-velocity of the star                       (:671-676)
//   :572-578  _get_velocity, :283-322 _rotate_vector: the z-velocity of the star is
//             -sin(i) K0 m_planet (cos w (cos f + e) - sin w sin f) -- the same function of f with
//             another amplitude, so one kernel serves both forms.
//   :329-334  M = (t - t_periastron) n ; kepler(M, e)
//
// A radial-velocity series is a few hundred to a few thousand epochs: the work is nothing, the
// ~20 launch-bound torch kernels of the composed path (M, Kepler op, rotations, broadcasts, and
// their reverse) were as long as a whole light-curve sweep.  One launch forward, one reverse.
//   rv[d][n][p] = amp[d][p] * (cw cos f - sw sin f + e cw),   f = f(M = (t_n - tp) nn, e)
// Reverse: one block per (draw, planet); lanes stride over the epochs, partial sums in
// registers, one fixed-order LDS reduction (bit-reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"
#include "exo_math.hpp"

namespace {

constexpr int kRvBlock = 256;

struct RvSample {
  double g;      // cw cos f - sw sin f + e cw
  double sinf, cosf;
};

// e outside [0, 1): NaN (the docstring's contract for the Kepler op, keplerian.py:58)
__device__ __forceinline__ RvSample rv_sample(double t, const double* __restrict__ p) {
  const double e = p[EXO_RV_ECC];
  const bool ok = (e >= 0.0) && (e < 1.0);
  const double es = ok ? e : 0.5;
  const exo::KeplerHalf kh = exo::kepler_half((t - p[EXO_RV_TP]) * p[EXO_RV_N], es, sqrt(1.0 - es), sqrt(1.0 + es));
  const double X2 = kh.X * kh.X, Y2 = kh.Y * kh.Y;
  const double iden = 1.0 / (X2 + Y2);
  const double nan = __builtin_nan("");
  RvSample s;
  s.sinf = ok ? 2.0 * kh.X * kh.Y * iden : nan;
  s.cosf = ok ? (X2 - Y2) * iden : nan;
  s.g = p[EXO_RV_COSW] * s.cosf - p[EXO_RV_SINW] * s.sinf + e * p[EXO_RV_COSW];
  return s;
}

__global__ __launch_bounds__(kRvBlock) void rv_fwd_kernel(const double* __restrict__ t, int64_t n_cad,
                                                          const double* __restrict__ params, int64_t n_draw,
                                                          int n_planet, double* __restrict__ rv) {
  const int64_t total = n_draw * n_cad * n_planet;
  const int64_t stride = (int64_t)gridDim.x * kRvBlock;
  for (int64_t i = (int64_t)blockIdx.x * kRvBlock + threadIdx.x; i < total; i += stride) {
    const int p = (int)(i % n_planet);
    const int64_t dn = i / n_planet;
    const int64_t n = dn % n_cad, d = dn / n_cad;
    const double* __restrict__ rec = params + (d * n_planet + p) * EXO_RV_NPAR;
    rv[i] = rec[EXO_RV_AMP] * rv_sample(t[n], rec).g;
  }
}

__global__ __launch_bounds__(kRvBlock) void rv_vjp_kernel(const double* __restrict__ t, int64_t n_cad,
                                                          const double* __restrict__ params, int n_planet,
                                                          const double* __restrict__ grv,
                                                          double* __restrict__ gparams) {
  const int64_t rec_i = blockIdx.x;   // draw * n_planet + planet
  const int64_t d = rec_i / n_planet;
  const int p = (int)(rec_i - d * n_planet);
  const double* __restrict__ rec = params + rec_i * EXO_RV_NPAR;
  const double nn = rec[EXO_RV_N], tp = rec[EXO_RV_TP], e = rec[EXO_RV_ECC], cw = rec[EXO_RV_COSW],
               sw = rec[EXO_RV_SINW], amp = rec[EXO_RV_AMP];
  const double ome2 = 1.0 - e * e;
  const double iome2 = 1.0 / ome2, iome32 = iome2 / sqrt(ome2);
  double acc[EXO_RV_NPAR];
#pragma unroll
  for (int k = 0; k < EXO_RV_NPAR; ++k) acc[k] = 0.0;
  for (int64_t n = threadIdx.x; n < n_cad; n += kRvBlock) {
    const double tn = t[n];
    const RvSample s = rv_sample(tn, rec);
    const double gb = grv[(d * n_cad + n) * n_planet + p];
    // d f / d M = (1 + e cos f)^2 / (1 - e^2)^(3/2),  d f / d e = (2 + e cos f) sin f / (1 - e^2)
    const double q = 1.0 + e * s.cosf;
    const double dfdM = q * q * iome32, dfde = (1.0 + q) * s.sinf * iome2;
    const double dgdf = -(cw * s.sinf + sw * s.cosf);
    const double a = gb * amp;
    acc[EXO_RV_N] += a * dgdf * dfdM * (tn - tp);
    acc[EXO_RV_TP] -= a * dgdf * dfdM * nn;
    acc[EXO_RV_ECC] += a * (dgdf * dfde + cw);
    acc[EXO_RV_COSW] += a * (s.cosf + e);
    acc[EXO_RV_SINW] -= a * s.sinf;
    acc[EXO_RV_AMP] += gb * s.g;
  }
  // fixed-order reduction: thread (slot, c) adds 16 columns, then one thread per slot the 16 partials
  __shared__ double cols[EXO_RV_NPAR][kRvBlock];
  __shared__ double part[EXO_RV_NPAR][16];
#pragma unroll
  for (int k = 0; k < EXO_RV_NPAR; ++k) cols[k][threadIdx.x] = acc[k];
  __syncthreads();
  const int slot = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (slot < EXO_RV_NPAR) {
    double v = 0.0;
    for (int i = 0; i < kRvBlock / 16; ++i) v += cols[slot][c + 16 * i];
    part[slot][c] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < EXO_RV_NPAR) {
    double v = 0.0;
    for (int i = 0; i < 16; ++i) v += part[threadIdx.x][i];
    gparams[rec_i * EXO_RV_NPAR + threadIdx.x] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Position / velocity vectors in the observer frame from the same solve (keplerian.py:380-409 _get_position,
// :572-578 _get_velocity, :283-322 _rotate_vector): what get_{star,planet,relative}_{position,velocity} and
// get_relative_angles (astrometry, :544-570) are made of.  In the orbital plane
//     position:  (u, v) = (1 - e^2) / (1 + e cos f) (cos f, sin f)        velocity:  (u, v) = (-sin f, cos f + e)
//     acceleration (:679-706):  (u, v) = -(1 + e cos f)^2 / (1 - e^2) (cos f, sin f)
// times an amplitude (a_star, a_planet, -a [x parallax au_per_R_sun]; K0 m; (K0 m)^2 / a), then the three rotations
//     x1 = cw u - sw v,  y1 = sw u + cw v;   x2 = x1,  y2 = ci y1,  Z = -si y1;   X = cO x2 - sO y2,  Y = sO x2 + cO y2.
// An astrometric or imaging series is tens of epochs: as for the radial velocities, the composed path's launch-bound
// torch kernels (solve, radius, three rotations, broadcasts, and their reverse: ~40) are the cost; here one launch
// each way.  out[d][n][p][3]; reverse: one block per (draw, planet), fixed-order reduction.
// ---------------------------------------------------------------------------------------------
struct OvSample {
  double sinf, cosf;
  double u, v;          // in-plane vector for unit amplitude
  double x2, y1, y2;    // after the omega and inclination rotations
  double X, Y, Z;       // unit amplitude
};

template <int MODE>
__device__ __forceinline__ OvSample ov_sample(double t, const double* __restrict__ p) {
  const double e = p[EXO_OV_ECC];
  const bool ok = (e >= 0.0) && (e < 1.0);
  const double es = ok ? e : 0.5;
  const exo::KeplerHalf kh = exo::kepler_half((t - p[EXO_OV_TP]) * p[EXO_OV_N], es, sqrt(1.0 - es), sqrt(1.0 + es));
  const double X2 = kh.X * kh.X, Y2 = kh.Y * kh.Y;
  const double iden = 1.0 / (X2 + Y2);
  const double nan = __builtin_nan("");
  OvSample s;
  s.sinf = ok ? 2.0 * kh.X * kh.Y * iden : nan;
  s.cosf = ok ? (X2 - Y2) * iden : nan;
  if (MODE == 1) {
    s.u = -s.sinf; s.v = s.cosf + e;
  } else if (MODE == 2) {
    const double q = 1.0 + e * s.cosf, g = q * q / (1.0 - e * e);
    s.u = -g * s.cosf; s.v = -g * s.sinf;
  } else {
    const double rho = (1.0 - e * e) / (1.0 + e * s.cosf);
    s.u = rho * s.cosf; s.v = rho * s.sinf;
  }
  const double x1 = p[EXO_OV_COSW] * s.u - p[EXO_OV_SINW] * s.v;
  s.y1 = p[EXO_OV_SINW] * s.u + p[EXO_OV_COSW] * s.v;
  s.x2 = x1;
  s.y2 = p[EXO_OV_COSI] * s.y1;
  s.Z = -p[EXO_OV_SINI] * s.y1;
  s.X = p[EXO_OV_COSO] * s.x2 - p[EXO_OV_SINO] * s.y2;
  s.Y = p[EXO_OV_SINO] * s.x2 + p[EXO_OV_COSO] * s.y2;
  return s;
}

template <int MODE>
__global__ __launch_bounds__(kRvBlock) void ov_fwd_kernel(const double* __restrict__ t, int64_t n_cad,
                                                          const double* __restrict__ params, int64_t n_draw,
                                                          int n_planet, double* __restrict__ out) {
  const int64_t total = n_draw * n_cad * n_planet;
  const int64_t stride = (int64_t)gridDim.x * kRvBlock;
  for (int64_t i = (int64_t)blockIdx.x * kRvBlock + threadIdx.x; i < total; i += stride) {
    const int p = (int)(i % n_planet);
    const int64_t dn = i / n_planet;
    const int64_t n = dn % n_cad, d = dn / n_cad;
    const double* __restrict__ rec = params + (d * n_planet + p) * EXO_OV_NPAR;
    const OvSample s = ov_sample<MODE>(t[n], rec);
    const double a = rec[EXO_OV_AMP];
    out[3 * i] = a * s.X; out[3 * i + 1] = a * s.Y; out[3 * i + 2] = a * s.Z;
  }
}

template <int MODE>
__global__ __launch_bounds__(kRvBlock) void ov_vjp_kernel(const double* __restrict__ t, int64_t n_cad,
                                                          const double* __restrict__ params, int n_planet,
                                                          const double* __restrict__ gout,
                                                          double* __restrict__ gparams) {
  const int64_t rec_i = blockIdx.x;   // draw * n_planet + planet
  const int64_t d = rec_i / n_planet;
  const int p = (int)(rec_i - d * n_planet);
  const double* __restrict__ rec = params + rec_i * EXO_OV_NPAR;
  const double nn = rec[EXO_OV_N], tp = rec[EXO_OV_TP], e = rec[EXO_OV_ECC], cw = rec[EXO_OV_COSW], sw = rec[EXO_OV_SINW],
               ci = rec[EXO_OV_COSI], si = rec[EXO_OV_SINI], amp = rec[EXO_OV_AMP], cO = rec[EXO_OV_COSO],
               sO = rec[EXO_OV_SINO];
  const double ome2 = 1.0 - e * e;
  const double iome2 = 1.0 / ome2, iome32 = iome2 / sqrt(ome2);
  double acc[EXO_OV_NPAR];
#pragma unroll
  for (int k = 0; k < EXO_OV_NPAR; ++k) acc[k] = 0.0;
  for (int64_t n = threadIdx.x; n < n_cad; n += kRvBlock) {
    const double tn = t[n];
    const OvSample s = ov_sample<MODE>(tn, rec);
    const double* __restrict__ g = gout + 3 * ((d * n_cad + n) * n_planet + p);
    const double gX0 = g[0], gY0 = g[1], gZ0 = g[2];
    acc[EXO_OV_AMP] += gX0 * s.X + gY0 * s.Y + gZ0 * s.Z;
    const double gX = amp * gX0, gY = amp * gY0, gZ = amp * gZ0;
    acc[EXO_OV_COSO] += gX * s.x2 + gY * s.y2;
    acc[EXO_OV_SINO] += gY * s.x2 - gX * s.y2;
    const double gx2 = gX * cO + gY * sO, gy2 = gY * cO - gX * sO;
    acc[EXO_OV_COSI] += gy2 * s.y1;
    acc[EXO_OV_SINI] -= gZ * s.y1;
    const double gy1 = gy2 * ci - gZ * si, gx1 = gx2;
    acc[EXO_OV_COSW] += gx1 * s.u + gy1 * s.v;
    acc[EXO_OV_SINW] += gy1 * s.u - gx1 * s.v;
    const double gu = gx1 * cw + gy1 * sw, gv = gy1 * cw - gx1 * sw;
    // (u, v) as functions of (f, e), f = f(M, e):  d f / d M = (1 + e cos f)^2 / (1 - e^2)^(3/2),
    // d f / d e = (2 + e cos f) sin f / (1 - e^2)
    const double q = 1.0 + e * s.cosf;
    double gf, ge;
    if (MODE == 1) {
      gf = -gu * s.cosf - gv * s.sinf;
      ge = gv;
    } else if (MODE == 2) {
      const double g = q * q * iome2;
      const double g_f = -2.0 * q * e * s.sinf * iome2;                                 // d g / d f
      const double g_e = 2.0 * q * (s.cosf * ome2 + e * q) * iome2 * iome2;             // d g / d e at fixed f
      gf = -gu * (g_f * s.cosf - g * s.sinf) - gv * (g_f * s.sinf + g * s.cosf);
      ge = -(gu * s.cosf + gv * s.sinf) * g_e;
    } else {
      const double iq = 1.0 / q, rho = ome2 * iq;
      const double rho_f = rho * e * s.sinf * iq;                                // d rho / d f
      const double rho_e = -(2.0 * e + s.cosf * (1.0 + e * e)) * iq * iq;        // d rho / d e at fixed f
      gf = gu * (rho_f * s.cosf - rho * s.sinf) + gv * (rho_f * s.sinf + rho * s.cosf);
      ge = (gu * s.cosf + gv * s.sinf) * rho_e;
    }
    const double dfdM = q * q * iome32, dfde = (1.0 + q) * s.sinf * iome2;
    const double gM = gf * dfdM;
    acc[EXO_OV_N] += gM * (tn - tp);
    acc[EXO_OV_TP] -= gM * nn;
    acc[EXO_OV_ECC] += ge + gf * dfde;
  }
  // fixed-order reduction, as in rv_vjp_kernel
  __shared__ double cols[EXO_OV_NPAR][kRvBlock];
  __shared__ double part[EXO_OV_NPAR][16];
#pragma unroll
  for (int k = 0; k < EXO_OV_NPAR; ++k) cols[k][threadIdx.x] = acc[k];
  __syncthreads();
  const int slot = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (slot < EXO_OV_NPAR) {
    double v = 0.0;
    for (int i = 0; i < kRvBlock / 16; ++i) v += cols[slot][c + 16 * i];
    part[slot][c] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < EXO_OV_NPAR) {
    double v = 0.0;
    for (int i = 0; i < 16; ++i) v += part[threadIdx.x][i];
    gparams[rec_i * EXO_OV_NPAR + threadIdx.x] = v;
  }
}

inline bool rv_args_ok(int64_t n_cad, int64_t n_draw, int32_t n_planet) {
  return n_cad >= 0 && n_draw >= 0 && n_planet >= 1 && n_draw * (int64_t)n_planet <= 0x7fffffff;
}

}  // namespace

extern "C" {

int exo_radial_velocity_fwd_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw,
                                int32_t n_planet, double* rv, void* stream) {
  if (!rv_args_ok(n_cad, n_draw, n_planet)) return EXO_ERR_INVALID_ARGUMENT;
  const int64_t total = n_draw * n_cad * n_planet;
  if (total == 0) return EXO_OK;
  if (!t || !params || !rv) return EXO_ERR_INVALID_ARGUMENT;
  int64_t blocks = (total + kRvBlock - 1) / kRvBlock;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(rv_fwd_kernel, dim3((unsigned)blocks), dim3(kRvBlock), 0, (hipStream_t)stream, t, n_cad, params,
                     n_draw, n_planet, rv);
  return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH;
}

int exo_radial_velocity_vjp_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw,
                                int32_t n_planet, const double* grv, double* gparams, void* stream) {
  if (!rv_args_ok(n_cad, n_draw, n_planet)) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!params || !gparams || (n_cad > 0 && (!t || !grv))) return EXO_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(rv_vjp_kernel, dim3((unsigned)(n_draw * n_planet)), dim3(kRvBlock), 0, (hipStream_t)stream, t,
                     n_cad, params, n_planet, grv, gparams);
  return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH;
}

int exo_orbit_vector_fwd_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, double* out, void* stream) {
  if (!rv_args_ok(n_cad, n_draw, n_planet) || flags > EXO_OV_ACCELERATION) return EXO_ERR_INVALID_ARGUMENT;
  const int64_t total = n_draw * n_cad * n_planet;
  if (total == 0) return EXO_OK;
  if (!t || !params || !out) return EXO_ERR_INVALID_ARGUMENT;
  int64_t blocks = (total + kRvBlock - 1) / kRvBlock;
  if (blocks > 65536) blocks = 65536;
  if (flags == EXO_OV_VELOCITY)
    hipLaunchKernelGGL(ov_fwd_kernel<1>, dim3((unsigned)blocks), dim3(kRvBlock), 0, (hipStream_t)stream, t, n_cad, params,
                       n_draw, n_planet, out);
  else if (flags == EXO_OV_ACCELERATION)
    hipLaunchKernelGGL(ov_fwd_kernel<2>, dim3((unsigned)blocks), dim3(kRvBlock), 0, (hipStream_t)stream, t, n_cad, params,
                       n_draw, n_planet, out);
  else
    hipLaunchKernelGGL(ov_fwd_kernel<0>, dim3((unsigned)blocks), dim3(kRvBlock), 0, (hipStream_t)stream, t, n_cad, params,
                       n_draw, n_planet, out);
  return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH;
}

int exo_orbit_vector_vjp_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, const double* gout, double* gparams, void* stream) {
  if (!rv_args_ok(n_cad, n_draw, n_planet) || flags > EXO_OV_ACCELERATION) return EXO_ERR_INVALID_ARGUMENT;
  if (n_draw == 0) return EXO_OK;
  if (!params || !gparams || (n_cad > 0 && (!t || !gout))) return EXO_ERR_INVALID_ARGUMENT;
  if (flags == EXO_OV_VELOCITY)
    hipLaunchKernelGGL(ov_vjp_kernel<1>, dim3((unsigned)(n_draw * n_planet)), dim3(kRvBlock), 0, (hipStream_t)stream, t,
                       n_cad, params, n_planet, gout, gparams);
  else if (flags == EXO_OV_ACCELERATION)
    hipLaunchKernelGGL(ov_vjp_kernel<2>, dim3((unsigned)(n_draw * n_planet)), dim3(kRvBlock), 0, (hipStream_t)stream, t,
                       n_cad, params, n_planet, gout, gparams);
  else
    hipLaunchKernelGGL(ov_vjp_kernel<0>, dim3((unsigned)(n_draw * n_planet)), dim3(kRvBlock), 0, (hipStream_t)stream, t,
                       n_cad, params, n_planet, gout, gparams);
  return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH;
}

}  // extern "C"
