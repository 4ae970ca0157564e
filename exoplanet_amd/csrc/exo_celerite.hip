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
// The recurrence is strictly sequential in time, so parallelism is over draws and,
// inside a draw, over the J state indices: a draw occupies G = next_pow2(J) adjacent
// lanes (lane j owns row j of S), exchanging values by DPP.  Everything that does
// not depend on the recurrence (U_n, V_n, P_n: sin / cos / exp) is produced by a
// fully parallel pre-pass; the log-determinant accumulates as (mantissa, exponent).
// The saved factorisation (needed by the reverse recurrence) is laid out
// [quantity][cadence][draw x state index] so that a wave's stores and loads are
// coalesced, and is read back through a software prefetch ring.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"

namespace {

constexpr int kWave = 64;
constexpr double kHalfLog2Pi = 0.91893853320467274178;

// A draw is spread over G = next_pow2(J) adjacent lanes: lane j of the group owns
// state index j (row j of the symmetric J x J matrix S, W_j, F_j, U_j, V_j, P_j).
// The sequential chain per cadence then is: one row update (J FMAs), one row-times-
// vector (J FMAs), a log2(G)-step butterfly for the two dot products, one division --
// instead of the whole O(J^2) update on one lane.  A wave carries 64 / G draws.
template <int J>
struct Group {
  static constexpr int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : 8));
};

// In-register lane exchange (DPP) -- an ds_bpermute-based __shfl costs ~100+ cycles of
// latency on the sequential chain; quad_perm / row_shl / row_shr moves cost a few.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114;  // lane i reads lane i+4 / i-4 (same row of 16)

// value held by the lane at distance 4 inside an aligned group of 8 (lane ^ 4)
__device__ __forceinline__ double xor4(double v) {
  const double up = dpp_mov<kRowShl4>(v), dn = dpp_mov<kRowShr4>(v);
  return (threadIdx.x & 4) ? dn : up;
}

// value of `v` held by lane L of this lane's aligned group of G lanes
template <int G, int L>
__device__ __forceinline__ double group_get_c(double v) {
  if (G == 1) return v;
  if (G == 2) return dpp_mov<(L | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6))>(v);
  constexpr int R = L & 3;
  const double q = dpp_mov<R * 0x55>(v);  // every quad broadcasts its own lane R
  if (G == 4) return q;
  // G == 8: pick this quad's broadcast or the other quad's
  const double other = xor4(q);
  return (((threadIdx.x >> 2) & 1) == (L >> 2)) ? q : other;
}

template <int G>
__device__ __forceinline__ double group_get(double v, int l) {
  switch (l) {  // l is a compile-time constant after unrolling; DPP controls are immediates
    case 0: return group_get_c<G, 0>(v);
    case 1: return group_get_c<G, (G > 1 ? 1 : 0)>(v);
    case 2: return group_get_c<G, (G > 2 ? 2 : 0)>(v);
    case 3: return group_get_c<G, (G > 2 ? 3 : 0)>(v);
    case 4: return group_get_c<G, (G > 4 ? 4 : 0)>(v);
    case 5: return group_get_c<G, (G > 4 ? 5 : 0)>(v);
    case 6: return group_get_c<G, (G > 4 ? 6 : 0)>(v);
    default: return group_get_c<G, (G > 4 ? 7 : 0)>(v);
  }
}

// butterfly partner lane ^ M within the group
template <int M>
__device__ __forceinline__ double xor_get(double v) {
  if (M == 1) return dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  if (M == 2) return dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  return xor4(v);
}

// sum over the G lanes of a group, result in every lane
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += xor_get<1>(v);
  if (G >= 4) v += xor_get<2>(v);
  if (G >= 8) v += xor_get<4>(v);
  return v;
}

// per-lane view of the term coefficients: state index j of a real term (a, c) or of
// a complex pair (a, b, c, d); `odd` marks the second index of a pair
struct LaneCoef {
  double a, b, c, d;
  bool real, odd, live;
};

__device__ __forceinline__ LaneCoef lane_coef(const double* __restrict__ coef_real, int n_real,
                                              const double* __restrict__ coef_complex, int n_complex,
                                              int64_t draw, int j, int J) {
  LaneCoef k;
  k.live = j < J;
  k.real = j < n_real;
  k.odd = false;
  k.a = k.b = k.c = k.d = 0.0;
  if (!k.live) return k;
  if (k.real) {
    const double* p = coef_real + (draw * n_real + j) * 2;
    k.a = p[0]; k.c = p[1];
  } else {
    const int jc = (j - n_real) >> 1;
    const double* p = coef_complex + (draw * n_complex + jc) * 4;
    k.a = p[0]; k.b = p[1]; k.c = p[2]; k.d = p[3];
    k.odd = ((j - n_real) & 1) != 0;
  }
  return k;
}

// U_j, V_j of SURVEY Appendix B at time t for this lane's state index
__device__ __forceinline__ void lane_uv(const LaneCoef& k, double t, double* U, double* V, double* cs, double* sn) {
  if (k.real || !k.live) {
    *U = k.live ? k.a : 0.0;
    *V = k.live ? 1.0 : 0.0;
    *cs = 1.0; *sn = 0.0;
    return;
  }
  double s, c;
  sincos(k.d * t, &s, &c);
  *cs = c; *sn = s;
  *U = k.odd ? (k.a * s - k.b * c) : (k.a * c + k.b * s);
  *V = k.odd ? s : c;
}

// saved-state addressing.  Per draw: d, z at [q][n][draw]; per (draw, j): W, F and the
// J entries of row j of S at [q][n][draw * J + j] -- lanes of a wave are consecutive
// (draw, j), so a wave's stores / loads are contiguous.
struct StateIdx {
  int64_t n, n_draw;
  int J;
  __device__ __forceinline__ int64_t scal(int q, int64_t i, int64_t draw) const {  // q = 0 (d), 1 (z)
    return ((int64_t)q * n + i) * n_draw + draw;
  }
  __device__ __forceinline__ int64_t vec(int q, int64_t i, int64_t draw, int j) const {  // q = 0 (W), 1 (F), 2.. (S row)
    return 2 * n * n_draw + (((int64_t)q * n + i) * n_draw + draw) * J + j;
  }
  // U_n, V_n, P_n (q = 0, 1, 2) written by the parallel pre-pass: the sequential kernels
  // never evaluate a sin, cos or exp
  __device__ __forceinline__ int64_t uvp(int q, int64_t i, int64_t draw, int j) const {
    return (2 + (int64_t)(2 + J) * J) * n * n_draw + (((int64_t)q * n + i) * n_draw + draw) * J + j;
  }
};

// Pre-pass, fully parallel over (cadence, draw, state index): everything in the
// recurrences that does not depend on the recurrence itself.
__global__ __launch_bounds__(256) void celerite_prep_kernel(const double* __restrict__ t, int64_t n,
                                                            const double* __restrict__ coef_real, int n_real,
                                                            const double* __restrict__ coef_complex, int n_complex,
                                                            int64_t n_draw, int J, double* __restrict__ state) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = n_draw * J;
  if (e >= n * per) return;
  const int64_t i = e / per, rem = e - i * per;
  const int64_t draw = rem / J;
  const int j = (int)(rem - draw * J);
  const LaneCoef k = lane_coef(coef_real, n_real, coef_complex, n_complex, draw, j, J);
  const StateIdx six{n, n_draw, J};
  double U, V, cs, sn;
  const double ti = t[i];
  lane_uv(k, ti, &U, &V, &cs, &sn);
  state[six.uvp(0, i, draw, j)] = U;
  state[six.uvp(1, i, draw, j)] = V;
  state[six.uvp(2, i, draw, j)] = i > 0 ? exp(-k.c * (ti - t[i - 1])) : 1.0;
}

template <int J, bool SAVE>
__global__ __launch_bounds__(kWave) void celerite_fwd_kernel(
    const double* __restrict__ t, const double* __restrict__ resid, const double* __restrict__ diag,
    int64_t n_diag, int64_t n, const double* __restrict__ coef_real, int n_real,
    const double* __restrict__ coef_complex, int n_complex, int64_t n_draw, double* __restrict__ loglike,
    double* __restrict__ state) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  const LaneCoef k = lane_coef(coef_real, n_real, coef_complex, n_complex, draw, j, J);
  const bool store = SAVE && live_draw && k.live;
  // a_n = diag_n + sum of the a coefficients (first index of each term)
  const double asum = group_sum<G>((k.live && !k.odd) ? k.a : 0.0);
  const StateIdx six{n, n_draw, J};
  const double* __restrict__ y = resid + draw * n;
  const double* __restrict__ dg = diag + (n_diag == 1 ? 0 : draw * n);

  double Srow[J], Wall[J], Uall[J], Pall[J];
#pragma unroll
  for (int l = 0; l < J; ++l) { Srow[l] = 0.0; Pall[l] = 1.0; }
  double Fj = 0.0, Pj = 1.0, Uj, Vj, cs, sn;
  double tprev = t[0];
  const int jj = k.live ? j : 0;
  if (SAVE) {
    Uj = k.live ? state[six.uvp(0, 0, draw, jj)] : 0.0;
    Vj = k.live ? state[six.uvp(1, 0, draw, jj)] : 0.0;
  } else {
    lane_uv(k, tprev, &Uj, &Vj, &cs, &sn);
  }
  double d = dg[0] + asum;
  double z = y[0];
  bool bad = !(d > 0.0);
  double Wj = Vj / d;
#pragma unroll
  for (int l = 0; l < J; ++l) Wall[l] = group_get<G>(Wj, l);
  // sum log d_n = log prod d_n: carry the product as (mantissa, exponent) -- a multiply
  // and a frexp per cadence instead of a ~45-instruction log on the sequential chain
  double acc = z * z / d;
  int lexp;
  double lman = frexp(bad ? 1.0 : d, &lexp);
  int64_t lsum = lexp;
  double dt_prev = -1.0;
  if (store) {
    if (j == 0) { state[six.scal(0, 0, draw)] = d; state[six.scal(1, 0, draw)] = z; }
    state[six.vec(0, 0, draw, j)] = Wj;
    state[six.vec(1, 0, draw, j)] = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) state[six.vec(2 + l, 0, draw, j)] = 0.0;
  }
  // software prefetch ring: the loads of cadence i + kPF are issued while cadence i
  // computes (one wave per SIMD and a serial chain: nothing else hides HBM latency)
  const int64_t vstride = n_draw * J;            // one cadence, in doubles, of a per-(draw, j) quantity
  const int64_t qstride = n * vstride;           // one quantity
  double* __restrict__ p_vec = SAVE ? state + six.vec(0, 0, draw, jj) : nullptr;
  double* __restrict__ p_scal = SAVE ? state + six.scal(0, 0, draw) : nullptr;
  const double* __restrict__ p_uvp = SAVE ? state + six.uvp(0, 0, draw, jj) : nullptr;
  constexpr int kPF = 4;
  double ring_t[kPF], ring_y[kPF], ring_g[kPF], ring_U[kPF], ring_V[kPF], ring_P[kPF];
#pragma unroll
  for (int q = 0; q < kPF; ++q) {
    const int64_t ii = 1 + q < n ? 1 + q : n - 1;
    ring_t[q] = t[ii]; ring_y[q] = y[ii]; ring_g[q] = dg[ii];
    if (SAVE) {
      const double* p = p_uvp + ii * vstride;
      ring_U[q] = k.live ? p[0] : 0.0;
      ring_V[q] = k.live ? p[qstride] : 0.0;
      ring_P[q] = k.live ? p[2 * qstride] : 0.0;
    }
  }
  const double* __restrict__ p_pf = SAVE ? p_uvp + (int64_t)kPF * vstride : nullptr;  // cadence i + kPF, i = 0
  for (int64_t i0 = 1; i0 < n; i0 += kPF) {
#pragma unroll
   for (int q = 0; q < kPF; ++q) {
    const int64_t i = i0 + q;
    if (i >= n) break;
    const double ti = ring_t[q];
    const double yi = ring_y[q];
    const double gi = ring_g[q];
    const double Ui = ring_U[q], Vi = ring_V[q], Pi = ring_P[q];
    {
      const int64_t ii = i + kPF < n ? i + kPF : n - 1;
      ring_t[q] = t[ii]; ring_y[q] = y[ii]; ring_g[q] = dg[ii];
      if (SAVE) {
        p_pf += vstride;                                   // -> cadence i + kPF
        const double* p = i + kPF < n ? p_pf : p_uvp + (n - 1) * vstride;
        ring_U[q] = k.live ? p[0] : 0.0;
        ring_V[q] = k.live ? p[qstride] : 0.0;
        ring_P[q] = k.live ? p[2 * qstride] : 0.0;
      }
    }
    const double dt = ti - tprev;
    tprev = ti;
    if (SAVE) {
      Pj = Pi;
#pragma unroll
      for (int l = 0; l < J; ++l) Pall[l] = group_get<G>(Pj, l);
    } else if (dt != dt_prev) {  // wave-uniform: evenly sampled series reuse P
      Pj = k.live ? exp(-k.c * dt) : 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) Pall[l] = group_get<G>(Pj, l);
      dt_prev = dt;
    }
    // row j of  S <- (P P^T) o (S + d W W^T) ;  F_j <- P_j (F_j + W_j z)
    Fj = Pj * fma(Wj, z, Fj);
    const double dwj = d * Wj;
#pragma unroll
    for (int l = 0; l < J; ++l) Srow[l] = Pj * Pall[l] * fma(dwj, Wall[l], Srow[l]);
    if (SAVE) { Uj = Ui; Vj = Vi; } else { lane_uv(k, ti, &Uj, &Vj, &cs, &sn); }
#pragma unroll
    for (int l = 0; l < J; ++l) Uall[l] = group_get<G>(Uj, l);
    double uj = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) uj = fma(Srow[l], Uall[l], uj);
    // the two dot products of the step share one butterfly
    const double pd = group_sum<G>(Uj * uj), pz = group_sum<G>(Uj * Fj);
    d = gi + asum - pd;
    z = yi - pz;
    bad = bad || !(d > 0.0);
    const double id = 1.0 / d;
    Wj = (Vj - uj) * id;
#pragma unroll
    for (int l = 0; l < J; ++l) Wall[l] = group_get<G>(Wj, l);
    acc = fma(z * z, id, acc);
    lman = frexp(lman * (d > 0.0 ? d : 1.0), &lexp);
    lsum += lexp;
    if (store) {
      // running pointers: one add per cadence instead of a 64-bit index product per access
      p_vec += vstride;
      p_scal += n_draw;
      if (j == 0) { p_scal[0] = d; p_scal[n * n_draw] = z; }
      p_vec[0] = Wj;
      p_vec[qstride] = Fj;
#pragma unroll
      for (int l = 0; l < J; ++l) p_vec[(2 + l) * qstride] = Srow[l];
    }
   }
  }
  if (live_draw && j == 0) {
    // not positive definite -> -inf in band (a sampler rejects the point)
    const double logdet = log(lman) + (double)lsum * 0.69314718055994530942;
    loglike[draw] = bad ? -INFINITY : fma(-0.5, acc + logdet, -(double)n * kHalfLog2Pi);
  }
}

// Reverse recurrence (hand-derived adjoint of the two recurrences above), same lane
// layout: lane j owns row j of the SYMMETRISED adjoint of S (S is symmetric, so only
// the symmetric part of its adjoint matters), Fb_j, Wb_j and the coefficient
// cotangents of its state index.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_vjp_kernel(
    const double* __restrict__ t, const double* __restrict__ diag, int64_t n_diag, int64_t n,
    const double* __restrict__ coef_real, int n_real, const double* __restrict__ coef_complex, int n_complex,
    int64_t n_draw, const double* __restrict__ gloglike, const double* __restrict__ state,
    double* __restrict__ gresid, double* __restrict__ gdiag, double* __restrict__ gdiag_sum,
    double* __restrict__ gcoef_real, double* __restrict__ gcoef_complex) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  const LaneCoef k = lane_coef(coef_real, n_real, coef_complex, n_complex, draw, j, J);
  const int jj = k.live ? j : 0;  // idle lanes read a valid slot and contribute zeros
  const StateIdx six{n, n_draw, J};
  const double gL = gloglike[draw];
  const bool lead = live_draw && j == 0;
  // the other state index of this lane's complex pair (itself for real terms / idle lanes)
  const int partner = (int)threadIdx.x + ((k.live && !k.real) ? (k.odd ? -1 : 1) : 0);

  double Sb[J];  // row j of the symmetric adjoint of S
#pragma unroll
  for (int l = 0; l < J; ++l) Sb[l] = 0.0;
  double Fb = 0.0, Wb = 0.0, db = 0.0, zb = 0.0, gasum = 0.0;
  double ga = 0.0, gb = 0.0, gc = 0.0, gd = 0.0;

  const int64_t vstride = n_draw * J, qstride = n * vstride;
  const double* __restrict__ sc0 = state + six.scal(0, 0, draw);
  const double* __restrict__ ve0 = state + six.vec(0, 0, draw, jj);
  const double* __restrict__ uv0 = state + six.uvp(0, 0, draw, jj);
  auto load = [&](int64_t i, double& d_, double& z_, double& W_, double& F_, double* S_) {
    const double* ps = sc0 + i * n_draw;   // one index product per group of loads
    const double* pv = ve0 + i * vstride;
    d_ = ps[0];
    z_ = ps[n * n_draw];
    W_ = k.live ? pv[0] : 0.0;
    F_ = k.live ? pv[qstride] : 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) S_[l] = k.live ? pv[(2 + l) * qstride] : 0.0;
  };
  double d_n, z_n, W_n, F_n, S_n[J];
  load(n - 1, d_n, z_n, W_n, F_n, S_n);
  double Pj = 1.0, Pall[J];
#pragma unroll
  for (int l = 0; l < J; ++l) Pall[l] = 1.0;
  // software prefetch ring over the saved factorisation: the loads of cadence
  // i - 1 - kPB are in flight while cadence i is processed (serial chain, one wave
  // per SIMD: nothing else hides the HBM latency of this 10 GB-scale stream)
  constexpr int kPB = J <= 2 ? 3 : 1;   // deeper rings cost more registers than they hide at larger J
  double r_d[kPB], r_z[kPB], r_W[kPB], r_F[kPB], r_S[kPB][J], r_t[kPB];
  double r_U[kPB], r_V[kPB], r_Vo[kPB], r_P[kPB];   // U, V, partner's V, P of the cadence being processed
  const int jp = k.live ? (partner - ((int)threadIdx.x - j)) : 0;   // partner's state index
  auto load_uvp = [&](int64_t i, double& U_, double& V_, double& Vo_, double& P_) {
    const double* pu = uv0 + i * vstride;
    U_ = k.live ? pu[0] : 0.0;
    V_ = k.live ? pu[qstride] : 0.0;
    Vo_ = k.live ? pu[qstride + (jp - jj)] : 0.0;
    P_ = k.live ? pu[2 * qstride] : 0.0;
  };
#pragma unroll
  for (int q = 0; q < kPB; ++q) {
    const int64_t ii = n - 2 - q >= 0 ? n - 2 - q : 0;
    load(ii, r_d[q], r_z[q], r_W[q], r_F[q], r_S[q]);
    r_t[q] = t[ii];
    load_uvp(ii + 1 < n ? ii + 1 : n - 1, r_U[q], r_V[q], r_Vo[q], r_P[q]);
  }
  double t_n = t[n - 1];

  for (int64_t i0 = n - 1; i0 >= 1; i0 -= kPB) {
#pragma unroll
   for (int q = 0; q < kPB; ++q) {
    const int64_t i = i0 - q;
    if (i < 1) break;
    const double d_p = r_d[q], z_p = r_z[q], W_p = r_W[q], F_p = r_F[q], t_p = r_t[q];
    double S_p[J];
#pragma unroll
    for (int l = 0; l < J; ++l) S_p[l] = r_S[q][l];
    const double Uj = r_U[q], Vj = r_V[q], Vo = r_Vo[q];
    Pj = r_P[q];
    {
      const int64_t ii = i - 1 - kPB >= 0 ? i - 1 - kPB : 0;
      load(ii, r_d[q], r_z[q], r_W[q], r_F[q], r_S[q]);
      r_t[q] = t[ii];
      load_uvp(ii + 1, r_U[q], r_V[q], r_Vo[q], r_P[q]);
    }
    const double ti = t_n;
    const double dt = ti - t_p;
    t_n = t_p;
#pragma unroll
    for (int l = 0; l < J; ++l) Pall[l] = group_get<G>(Pj, l);
    // cos / sin of this lane's complex pair: its own V and its partner's
    const double cs = k.odd ? Vo : Vj, sn = k.odd ? Vj : Vo;
    double Uall[J], Wpall[J];
#pragma unroll
    for (int l = 0; l < J; ++l) { Uall[l] = group_get<G>(Uj, l); Wpall[l] = group_get<G>(W_p, l); }
    const double id = 1.0 / d_n;
    // (5) log-likelihood terms, (4) z_n = y_n - U.F_n
    const double zbar = zb - gL * z_n * id;
    const double wdot = group_sum<G>(Wb * W_n);
    const double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id) - wdot * id;
    if (lead) {
      gresid[draw * n + i] = zbar;
      if (gdiag) gdiag[draw * n + i] = dbar;
    }
    gasum += dbar;
    double Ub = -zbar * F_n;
    Fb = fma(-zbar, Uj, Fb);
    // (3) W_n = (V_n - u) / d_n ; d_n = a_n - U.u ; u = S_n U
    const double Vb = Wb * id;
    double uj = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) uj = fma(S_n[l], Uall[l], uj);
    Ub = fma(-dbar, uj, Ub);
    const double ubj = -Vb - dbar * Uj;
    double uball[J];
#pragma unroll
    for (int l = 0; l < J; ++l) uball[l] = group_get<G>(ubj, l);
    double acc_u = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      Sb[l] = fma(0.5, fma(ubj, Uall[l], uball[l] * Uj), Sb[l]);  // symmetrised  ub U^T
      acc_u = fma(S_n[l], uball[l], acc_u);                        // (S_n^T ub)_j, S symmetric
    }
    Ub += acc_u;
    // (2) F_n = P o G, G = F_p + W_p z_p   (1) S_n = P P^T o T, T = S_p + d_p W_p W_p^T
    const double Gj = fma(W_p, z_p, F_p);
    double Pb = Fb * Gj;
    const double Gb = Fb * Pj;
    double Wb_prev = Gb * z_p;
    const double zb_prev = group_sum<G>(Gb * W_p);
    double psum = 0.0, wsum = 0.0, dsum = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      const double T = fma(d_p * W_p, Wpall[l], S_p[l]);
      const double Tb = Sb[l] * Pj * Pall[l];       // adjoint of T (symmetric)
      psum = fma(2.0 * Sb[l] * T, Pall[l], psum);
      wsum = fma(2.0 * Tb, Wpall[l], wsum);
      dsum = fma(Tb, Wpall[l], dsum);
      Sb[l] = Tb;                                   // becomes the adjoint of S_{n-1}
    }
    Pb += psum;
    Wb_prev = fma(d_p, wsum, Wb_prev);
    const double db_prev = group_sum<G>(dsum * W_p);
    // parameter adjoints: P = exp(-c dt), U, V.  A complex pair's (a, b, d) cotangents
    // collect on its first lane: fetch the partner lane's Ub, Vb
    gc = fma(-dt * Pj, Pb, gc);
    if (k.live) {
      if (k.real) {
        ga += Ub;
      }
    }
    {
      const double Ub_o = __shfl(Ub, partner, 64), Vb_o = __shfl(Vb, partner, 64);
      if (k.live && !k.real && !k.odd) {
        ga += Ub * cs + Ub_o * sn;
        gb += Ub * sn - Ub_o * cs;
        gd += ti * (Ub * (-k.a * sn + k.b * cs) + Ub_o * (k.a * cs + k.b * sn) - Vb * sn + Vb_o * cs);
      }
    }
    // shift to cadence n-1
    db = db_prev;
    zb = zb_prev;
    Fb = Gb;
    Wb = Wb_prev;
    d_n = d_p; z_n = z_p; W_n = W_p; F_n = F_p;
#pragma unroll
    for (int l = 0; l < J; ++l) S_n[l] = S_p[l];
   }
  }
  // cadence 0: d_0 = a_0, W_0 = V_0 / d_0, z_0 = y_0
  {
    const double id = 1.0 / d_n;
    const double zbar = zb - gL * z_n * id;
    const double wdot = group_sum<G>(Wb * W_n);
    const double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id) - wdot * id;
    if (lead) {
      gresid[draw * n] = zbar;
      if (gdiag) gdiag[draw * n] = dbar;
    }
    gasum += dbar;
    const double t0 = t[0];
    double Uj, Vj, Vo, P0;
    load_uvp(0, Uj, Vj, Vo, P0);
    const double cs = k.odd ? Vo : Vj, sn = k.odd ? Vj : Vo;
    const double Vb = Wb * id;
    const double Vb_o = __shfl(Vb, partner, 64);
    if (k.live && !k.real && !k.odd) gd += t0 * (-Vb * sn + Vb_o * cs);
  }
  // the decay rate of a complex pair is shared by its two state indices
  const double gc_o = __shfl(gc, partner, 64);
  if (!live_draw || !k.live) return;
  if (j == 0 && gdiag_sum) gdiag_sum[draw] = gasum;
  if (k.real) {
    double* o = gcoef_real + (draw * n_real + j) * 2;
    o[0] = ga + gasum;  // a_n = diag_n + sum a
    o[1] = gc;
  } else if (!k.odd) {
    double* o = gcoef_complex + (draw * n_complex + ((j - n_real) >> 1)) * 4;
    o[0] = ga + gasum;
    o[1] = gb;
    o[2] = gc + gc_o;
    o[3] = gd;
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
  return n * n_draw * (2 + 2 * J + J * J + 3 * J);
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
  const int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : 8));
  const int64_t per_wave = kWave / G;
  const dim3 grid((unsigned)((n_draw + per_wave - 1) / per_wave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  if (state) {
    const int64_t n_el = n * n_draw * J;
    hipLaunchKernelGGL(celerite_prep_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st, t, n, coef_real,
                       n_real, coef_complex, n_complex, n_draw, J, state);
    if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
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
  const int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : 8));
  const int64_t per_wave = kWave / G;
  const dim3 grid((unsigned)((n_draw + per_wave - 1) / per_wave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_vjp_kernel<JJ>), grid, block, 0, st, t, diag, n_diag, n, coef_real,
                                        n_real, coef_complex, n_complex, n_draw, gloglike, state, gresid, gdiag,
                                        gdiag_sum, gcoef_real, gcoef_complex))
  return launch_status();
}

}  // extern "C"
