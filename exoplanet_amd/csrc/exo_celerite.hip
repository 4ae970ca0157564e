// exo_celerite.hip -- celerite (semiseparable) GP log-likelihood, value + VJP,
// batched over posterior draws / chains (gfx950, fp64 VALU, no MFMA).
//
// Replaces, for the reference's user-level model, what celerite2 does behind
//   GaussianProcess.compute() + GaussianProcess.log_likelihood(y)
// (celerite2 is a dependency of the reference, /root/reference/setup.py:36, but
// has no call site in its tree; the algorithm is the published one: Foreman-Mackey
// et al. 2017; Foreman-Mackey 2018):
//
//   factor:      S_n = (P P^T) o (S_{n-1} + d_{n-1} W_{n-1} W_{n-1}^T)
//                d_n = a_n - U_n S_n U_n^T ;  W_n = (V_n - S_n U_n) / d_n
//   solve_lower: F_n = P o (F_{n-1} + W_{n-1} z_{n-1}) ;  z_n = y_n - U_n . F_n
//   loglike    = -1/2 sum (z_n^2 / d_n + log d_n) - N/2 log 2 pi
//
// The recurrence is strictly sequential in time, so parallelism is over draws:
// ONE DRAW PER LANE, the J x J state in registers, U_n / V_n / P_n recomputed in
// the kernel from the term coefficients (a, b, c, d) instead of being streamed as
// N x J arrays.  t and diag are wave-uniform (broadcast loads); the per-draw
// residual rows stream through L1/L2; the saved factorisation (needed by the
// reverse recurrence) is laid out [quantity][cadence][draw] so that a wave's
// stores and loads are coalesced.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"

namespace {

constexpr int kWave = 64;
constexpr double kHalfLog2Pi = 0.91893853320467274178;

template <int J>
struct Coef {
  double a[J], b[J], c[J], d[J];  // per state index; a complex pair repeats its (a,b,c,d)
  double asum;
};

template <int J>
__device__ __forceinline__ void load_coef(Coef<J>& k, const double* __restrict__ coef_real, int n_real,
                                          const double* __restrict__ coef_complex, int n_complex, int64_t draw) {
  k.asum = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (j < n_real) {
      const double* p = coef_real + (draw * n_real + j) * 2;
      k.a[j] = p[0]; k.b[j] = 0.0; k.c[j] = p[1]; k.d[j] = 0.0;
      k.asum += p[0];
    } else {
      const int jc = (j - n_real) >> 1;
      const double* p = coef_complex + (draw * n_complex + jc) * 4;
      k.a[j] = p[0]; k.b[j] = p[1]; k.c[j] = p[2]; k.d[j] = p[3];
      if (((j - n_real) & 1) == 0) k.asum += p[0];
    }
  }
}

// U_n, V_n of SURVEY Appendix B for the cadence time t
template <int J>
__device__ __forceinline__ void make_uv(const Coef<J>& k, int n_real, double t, double* U, double* V) {
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (j < n_real) {
      U[j] = k.a[j];
      V[j] = 1.0;
    } else if (((j - n_real) & 1) == 0 && j + 1 < J) {
      double s, c;
      sincos(k.d[j] * t, &s, &c);
      U[j] = k.a[j] * c + k.b[j] * s;
      U[j + 1] = k.a[j] * s - k.b[j] * c;
      V[j] = c;
      V[j + 1] = s;
    }
  }
}

template <int J>
constexpr int n_state() { return 2 + 2 * J + J * (J + 1) / 2; }

// saved-state addressing: [quantity q][cadence n][draw]
struct StateIdx {
  int64_t n, n_draw;
  __device__ __forceinline__ int64_t at(int q, int64_t i, int64_t draw) const { return ((int64_t)q * n + i) * n_draw + draw; }
};

template <int J, bool SAVE>
__global__ __launch_bounds__(kWave) void celerite_fwd_kernel(
    const double* __restrict__ t, const double* __restrict__ resid, const double* __restrict__ diag,
    int64_t n_diag, int64_t n, const double* __restrict__ coef_real, int n_real,
    const double* __restrict__ coef_complex, int n_complex, int64_t n_draw, double* __restrict__ loglike,
    double* __restrict__ state) {
  const int64_t lane_draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = lane_draw < n_draw;
  const int64_t draw = live ? lane_draw : n_draw - 1;
  Coef<J> k;
  load_coef<J>(k, coef_real, n_real, coef_complex, n_complex, draw);
  const StateIdx six{n, n_draw};
  const double* __restrict__ y = resid + draw * n;
  const double* __restrict__ dg = diag + (n_diag == 1 ? 0 : draw * n);

  double S[J][J], F[J], W[J], U[J], V[J], P[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    F[j] = 0.0;
    P[j] = 1.0;
#pragma unroll
    for (int l = 0; l < J; ++l) S[j][l] = 0.0;
  }
  double tprev = t[0];
  make_uv<J>(k, n_real, tprev, U, V);
  double d = dg[0] + k.asum;
  double z = y[0];
  bool bad = !(d > 0.0);
#pragma unroll
  for (int j = 0; j < J; ++j) W[j] = V[j] / d;
  double acc = z * z / d + log(d);
  double dt_prev = -1.0;
  if (SAVE && live) {
    state[six.at(0, 0, draw)] = d;
    state[six.at(J + 1, 0, draw)] = z;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      state[six.at(1 + j, 0, draw)] = W[j];
      state[six.at(J + 2 + j, 0, draw)] = 0.0;
    }
    int q = 2 * J + 2;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = j; l < J; ++l) state[six.at(q++, 0, draw)] = 0.0;
  }
  double ynext = n > 1 ? y[1] : 0.0;
  for (int64_t i = 1; i < n; ++i) {
    const double ti = t[i];
    const double yi = ynext;
    if (i + 1 < n) ynext = y[i + 1];  // software prefetch of the next residual
    const double dt = ti - tprev;
    tprev = ti;
    if (dt != dt_prev) {  // wave-uniform: evenly sampled series reuse P
#pragma unroll
      for (int j = 0; j < J; ++j) P[j] = exp(-k.c[j] * dt);
      dt_prev = dt;
    }
    // S <- (P P^T) o (S + d W W^T) ; F <- P o (F + W z)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      F[j] = P[j] * fma(W[j], z, F[j]);
      const double dwj = d * W[j];
#pragma unroll
      for (int l = j; l < J; ++l) {
        const double v = P[j] * P[l] * fma(dwj, W[l], S[j][l]);
        S[j][l] = v;
        S[l][j] = v;
      }
    }
    make_uv<J>(k, n_real, ti, U, V);
    double dn = dg[i] + k.asum;
    double zf = yi;
    double u[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) s = fma(S[j][l], U[l], s);
      u[j] = s;
      dn = fma(-U[j], s, dn);
      zf = fma(-U[j], F[j], zf);
    }
    d = dn;
    z = zf;
    bad = bad || !(d > 0.0);
    const double id = 1.0 / d;
#pragma unroll
    for (int j = 0; j < J; ++j) W[j] = (V[j] - u[j]) * id;
    acc += z * z * id + log(d);
    if (SAVE && live) {
      state[six.at(0, i, draw)] = d;
      state[six.at(J + 1, i, draw)] = z;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        state[six.at(1 + j, i, draw)] = W[j];
        state[six.at(J + 2 + j, i, draw)] = F[j];
      }
      int q = 2 * J + 2;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int l = j; l < J; ++l) state[six.at(q++, i, draw)] = S[j][l];
    }
  }
  if (live) {
    // not positive definite -> -inf in band (a sampler rejects the point)
    loglike[draw] = bad ? -INFINITY : fma(-0.5, acc, -(double)n * kHalfLog2Pi);
  }
}

// Reverse recurrence (hand-derived adjoint of the two recurrences above).
template <int J>
__global__ __launch_bounds__(kWave) void celerite_vjp_kernel(
    const double* __restrict__ t, const double* __restrict__ diag, int64_t n_diag, int64_t n,
    const double* __restrict__ coef_real, int n_real, const double* __restrict__ coef_complex, int n_complex,
    int64_t n_draw, const double* __restrict__ gloglike, const double* __restrict__ state,
    double* __restrict__ gresid, double* __restrict__ gdiag, double* __restrict__ gdiag_sum,
    double* __restrict__ gcoef_real, double* __restrict__ gcoef_complex) {
  const int64_t lane_draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = lane_draw < n_draw;
  const int64_t draw = live ? lane_draw : n_draw - 1;
  Coef<J> k;
  load_coef<J>(k, coef_real, n_real, coef_complex, n_complex, draw);
  const StateIdx six{n, n_draw};
  const double gL = gloglike[draw];

  double Sb[J][J], Fb[J], Wb[J];
  double ga[J], gb[J], gc[J], gd[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    Fb[j] = 0.0; Wb[j] = 0.0; ga[j] = 0.0; gb[j] = 0.0; gc[j] = 0.0; gd[j] = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) Sb[j][l] = 0.0;
  }
  double db = 0.0, zb = 0.0, gasum = 0.0;

  // state at the current cadence
  double d_n, z_n, W_n[J], F_n[J], S_n[J][J];
  auto load_state = [&](int64_t i, double& d_, double& z_, double* W_, double* F_, double (*S_)[J]) {
    d_ = state[six.at(0, i, draw)];
    z_ = state[six.at(J + 1, i, draw)];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      W_[j] = state[six.at(1 + j, i, draw)];
      F_[j] = state[six.at(J + 2 + j, i, draw)];
    }
    int q = 2 * J + 2;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = j; l < J; ++l) {
        const double v = state[six.at(q++, i, draw)];
        S_[j][l] = v;
        S_[l][j] = v;
      }
  };
  load_state(n - 1, d_n, z_n, W_n, F_n, S_n);
  double U[J], V[J], P[J];
  double dt_prev = -1.0;
#pragma unroll
  for (int j = 0; j < J; ++j) P[j] = 1.0;

  for (int64_t i = n - 1; i >= 1; --i) {
    double d_p, z_p, W_p[J], F_p[J], S_p[J][J];
    load_state(i - 1, d_p, z_p, W_p, F_p, S_p);
    const double ti = t[i];
    const double dt = ti - t[i - 1];
    if (dt != dt_prev) {
#pragma unroll
      for (int j = 0; j < J; ++j) P[j] = exp(-k.c[j] * dt);
      dt_prev = dt;
    }
    make_uv<J>(k, n_real, ti, U, V);
    const double id = 1.0 / d_n;
    // (5) log-likelihood terms, (4) z_n = y_n - U.F_n
    const double zbar = zb - gL * z_n * id;
    double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id);
    if (live) gresid[draw * n + i] = zbar;
    double Ub[J], Vb[J], ub[J], u[J];
    double wdot = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Ub[j] = -zbar * F_n[j];
      Fb[j] = fma(-zbar, U[j], Fb[j]);
      // (3) W_n = (V_n - u) / d_n
      Vb[j] = Wb[j] * id;
      ub[j] = -Vb[j];
      wdot = fma(Wb[j], W_n[j], wdot);
    }
    dbar -= wdot * id;
    // d_n = a_n - U.u ;  u = S_n U
    if (gdiag && live) gdiag[draw * n + i] = dbar;
    gasum += dbar;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) s = fma(S_n[j][l], U[l], s);
      u[j] = s;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Ub[j] = fma(-dbar, u[j], Ub[j]);
      ub[j] = fma(-dbar, U[j], ub[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        Sb[j][l] = fma(ub[j], U[l], Sb[j][l]);
        s = fma(S_n[l][j], ub[l], s);
      }
      Ub[j] += s;
    }
    // (2) F_n = P o G, G = F_p + W_p z_p   (1) S_n = P P^T o T, T = S_p + d_p W_p W_p^T
    double Pb[J], Gb[J];
    double zb_prev = 0.0, db_prev = 0.0;
    double Wb_prev[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const double G = fma(W_p[j], z_p, F_p[j]);
      Pb[j] = Fb[j] * G;
      Gb[j] = Fb[j] * P[j];
      Wb_prev[j] = Gb[j] * z_p;
      zb_prev = fma(Gb[j], W_p[j], zb_prev);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double wsum = 0.0, psum = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        const double T = fma(d_p * W_p[j], W_p[l], S_p[j][l]);
        const double sym = Sb[j][l] + Sb[l][j];
        psum = fma(sym * T, P[l], psum);
        const double Tsym = sym * P[j] * P[l];   // Tb[j][l] + Tb[l][j]
        wsum = fma(Tsym, W_p[l], wsum);
      }
      Pb[j] += psum;
      Wb_prev[j] = fma(d_p, wsum, Wb_prev[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) {
        const double Tb = Sb[j][l] * P[j] * P[l];
        db_prev = fma(Tb * W_p[j], W_p[l], db_prev);
        Sb[j][l] = Tb;  // becomes the adjoint of S_{n-1}
      }
    // parameter adjoints: P = exp(-c dt), U, V
#pragma unroll
    for (int j = 0; j < J; ++j) {
      gc[j] = fma(-dt * P[j], Pb[j], gc[j]);
      if (j < n_real) {
        ga[j] += Ub[j];
      } else if (((j - n_real) & 1) == 0 && j + 1 < J) {
        const double c = V[j], s = V[j + 1];
        ga[j] += Ub[j] * c + Ub[j + 1] * s;
        gb[j] += Ub[j] * s - Ub[j + 1] * c;
        gd[j] += ti * (Ub[j] * (-k.a[j] * s + k.b[j] * c) + Ub[j + 1] * (k.a[j] * c + k.b[j] * s) - Vb[j] * s +
                       Vb[j + 1] * c);
      }
    }
    // shift to cadence n-1
    db = db_prev;
    zb = zb_prev;
    d_n = d_p;
    z_n = z_p;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Fb[j] = Gb[j];
      Wb[j] = Wb_prev[j];
      W_n[j] = W_p[j];
      F_n[j] = F_p[j];
#pragma unroll
      for (int l = 0; l < J; ++l) S_n[j][l] = S_p[j][l];
    }
  }
  // cadence 0: d_0 = a_0, W_0 = V_0 / d_0, z_0 = y_0
  {
    const double id = 1.0 / d_n;
    const double zbar = zb - gL * z_n * id;
    double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id);
    if (live) gresid[draw * n] = zbar;
    const double t0 = t[0];
    make_uv<J>(k, n_real, t0, U, V);
    double wdot = 0.0;
    double Vb[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Vb[j] = Wb[j] * id;
      wdot = fma(Wb[j], W_n[j], wdot);
    }
    dbar -= wdot * id;
    if (gdiag && live) gdiag[draw * n] = dbar;
    gasum += dbar;
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (j >= n_real && ((j - n_real) & 1) == 0 && j + 1 < J) gd[j] += t0 * (-Vb[j] * V[j + 1] + Vb[j + 1] * V[j]);
  }
  if (!live) return;
  if (gdiag_sum) gdiag_sum[draw] = gasum;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (j < n_real) {
      double* o = gcoef_real + (draw * n_real + j) * 2;
      o[0] = ga[j] + gasum;  // a_n = diag_n + sum a
      o[1] = gc[j];
    } else if (((j - n_real) & 1) == 0 && j + 1 < J) {
      double* o = gcoef_complex + (draw * n_complex + ((j - n_real) >> 1)) * 4;
      o[0] = ga[j] + gasum;
      o[1] = gb[j];
      o[2] = gc[j] + gc[j + 1];
      o[3] = gd[j];
    }
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH; }

inline bool gp_args_ok(int64_t n, int64_t n_diag, int32_t n_real, int32_t n_complex, int64_t n_draw) {
  const int J = n_real + 2 * n_complex;
  return n >= 1 && n_draw >= 1 && n_real >= 0 && n_complex >= 0 && J >= 1 && J <= EXO_GP_MAX_J &&
         (n_diag == 1 || n_diag == n_draw);
}

}  // namespace

extern "C" {

int64_t exo_celerite_state_doubles(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex) {
  const int64_t J = n_real + 2 * (int64_t)n_complex;
  if (n < 0 || n_draw < 0 || J < 1) return -1;
  return n * n_draw * (2 + 2 * J + J * (J + 1) / 2);
}

#define EXO_GP_DISPATCH(J_, CALL) \
  switch (J_) {                   \
    case 1: { constexpr int JJ = 1; CALL; } break; \
    case 2: { constexpr int JJ = 2; CALL; } break; \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    case 7: { constexpr int JJ = 7; CALL; } break; \
    case 8: { constexpr int JJ = 8; CALL; } break; \
    default: return EXO_ERR_INVALID_ARGUMENT;      \
  }

int exo_celerite_loglike_fwd_f64(const double* t, const double* resid, const double* diag, int64_t n_diag,
                                 int64_t n, const double* coef_real, int32_t n_real, const double* coef_complex,
                                 int32_t n_complex, int64_t n_draw, double* loglike, double* state,
                                 int64_t state_doubles, void* stream) {
  if (n_draw == 0) return EXO_OK;
  if (!gp_args_ok(n, n_diag, n_real, n_complex, n_draw) || !t || !resid || !diag || !loglike ||
      (n_real > 0 && !coef_real) || (n_complex > 0 && !coef_complex))
    return EXO_ERR_INVALID_ARGUMENT;
  if (state && state_doubles < exo_celerite_state_doubles(n, n_draw, n_real, n_complex)) return EXO_ERR_WORKSPACE;
  const int J = n_real + 2 * n_complex;
  const dim3 grid((unsigned)((n_draw + kWave - 1) / kWave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  if (state) {
    EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_fwd_kernel<JJ, true>), grid, block, 0, st, t, resid, diag, n_diag,
                                          n, coef_real, n_real, coef_complex, n_complex, n_draw, loglike, state))
  } else {
    EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_fwd_kernel<JJ, false>), grid, block, 0, st, t, resid, diag,
                                          n_diag, n, coef_real, n_real, coef_complex, n_complex, n_draw, loglike,
                                          state))
  }
  return launch_status();
}

int exo_celerite_loglike_vjp_f64(const double* t, const double* diag, int64_t n_diag, int64_t n,
                                 const double* coef_real, int32_t n_real, const double* coef_complex,
                                 int32_t n_complex, int64_t n_draw, const double* gloglike, const double* state,
                                 double* gresid, double* gdiag, double* gdiag_sum, double* gcoef_real,
                                 double* gcoef_complex, void* stream) {
  if (n_draw == 0) return EXO_OK;
  if (!gp_args_ok(n, n_diag, n_real, n_complex, n_draw) || !t || !diag || !gloglike || !state || !gresid ||
      (n_real > 0 && (!coef_real || !gcoef_real)) || (n_complex > 0 && (!coef_complex || !gcoef_complex)))
    return EXO_ERR_INVALID_ARGUMENT;
  const int J = n_real + 2 * n_complex;
  const dim3 grid((unsigned)((n_draw + kWave - 1) / kWave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_vjp_kernel<JJ>), grid, block, 0, st, t, diag, n_diag, n, coef_real,
                                        n_real, coef_complex, n_complex, n_draw, gloglike, state, gresid, gdiag,
                                        gdiag_sum, gcoef_real, gcoef_complex))
  return launch_status();
}

}  // extern "C"
