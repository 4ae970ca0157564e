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
// Three sets of kernels.
//  * Sequential (calls without a state buffer, short series, draws the time-parallel path cannot
//    take): parallelism over draws and, inside a draw, over the J state indices -- a draw occupies
//    G = next_pow2(J) adjacent lanes (lane j owns row j of S), exchanging values by DPP; U_n, V_n,
//    P_n come from a fully parallel pre-pass; the saved factorisation is laid out
//    [quantity][cadence][draw x state index] and read back through a software prefetch ring.
//  * Time-parallel, lane-group form (J > 2): the series is cut into chunks whose entering states --
//    and, in reverse, their adjoints -- come from Kalman filtering elements and a short scan over the
//    chunks, after which the same recurrences run inside all chunks at once, a draw on G lanes, the
//    full factorisation saved (see the comment block "Time-parallel path").
//  * Time-parallel, one lane per (draw, chunk) (J <= 2, exo_celerite_core.hpp): every state index in
//    the lane's registers -- no cross-lane traffic, none of the G-fold duplicated scalar work -- and a
//    CHECKPOINTED factorisation: the forward pass stores (F, S) every 4 cadences (10 B per (draw,
//    cadence) at J = 2 instead of 80), the reverse pass recomputes the cadences of a block from its
//    checkpoint in registers.  Draws too ill-conditioned for the scans' trees of element compositions (a score up to
//    1e8) stay on this path by its ROBUST route: Newton iterations on the chunks' entering states (celerite_robust_newton_kernel), the
//    adjoint scan fed from the chunks' own reverse recurrences (celerite_chunk_adj_kernel) -- docs/DESIGN_r1_r4.md 3.11.
// The library keeps no state between calls: how a series is cut is a pure function of the call's
// arguments (gp::chunk_plan), which the forward and the reverse call of a pair share.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/exoplanet_amd.h"
#include "exo_celerite_core.hpp"
#include "exo_celerite_group.hpp"
#include "exo_math.hpp"

namespace {

using namespace gp;

constexpr int kWave = 64;

// A draw is spread over G = next_pow2(J) adjacent lanes: lane j of the group owns
// state index j (row j of the symmetric J x J matrix S, W_j, F_j, U_j, V_j, P_j).
// The sequential chain per cadence then is: one row update (J FMAs), one row-times-
// vector (J FMAs), a log2(G)-step butterfly for the two dot products, one division --
// instead of the whole O(J^2) update on one lane.  A wave carries 64 / G draws.
template <int J>
struct Group {
  static constexpr int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : (J <= 8 ? 8 : 16)));   // (16: one DPP row, J = 9 .. 16)
};

// In-register lane exchange (DPP) -- an ds_bpermute-based __shfl costs ~100+ cycles of
// latency on the sequential chain; quad_perm / row_shl / row_shr moves cost a few.
// (bound_ctrl: a lane whose source lies outside its row reads 0 -- what `old` = 0 gave; with it the destination needs no
// zero written first: two moves fewer per exchanged double, a fifth of a butterfly step)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114;  // lane i reads lane i+4 / i-4 (same row of 16)
constexpr int kRowShl8 = 0x108, kRowShr8 = 0x118;  // ... i+8 / i-8
constexpr int kRowMirror = 0x140, kRowHalfMirror = 0x141;   // lane i reads lane 15 - i of its row / lane 7 - i of its half row

// value held by the lane at distance 4 inside an aligned group of 8 (lane ^ 4)
__device__ __forceinline__ double xor4(double v) {
  const double up = dpp_mov<kRowShl4>(v), dn = dpp_mov<kRowShr4>(v);
  return (threadIdx.x & 4) ? dn : up;
}

// ... at distance 8 inside an aligned group (= DPP row) of 16 (lane ^ 8)
__device__ __forceinline__ double xor8(double v) {
  const double up = dpp_mov<kRowShl8>(v), dn = dpp_mov<kRowShr8>(v);
  return (threadIdx.x & 8) ? dn : up;
}

// value of `v` held by lane L of this lane's aligned group of G lanes
template <int G, int L>
__device__ __forceinline__ double group_get_c(double v) {
  if (G == 1) return v;
  if (G == 2) return dpp_mov<(L | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6))>(v);
  constexpr int R = L & 3;
  const double q = dpp_mov<R * 0x55>(v);  // every quad broadcasts its own lane R
  if (G == 4) return q;
  // G >= 8: pick this quad's broadcast or the other quad's of the octet
  const double other = xor4(q);
  const double o8 = (((threadIdx.x >> 2) & 1) == ((L >> 2) & 1)) ? q : other;   // every octet broadcasts its own lane L & 7
  if (G == 8) return o8;
  // G == 16: this octet's or the other one's of the row
  const double other8 = xor8(o8);
  return (((threadIdx.x >> 3) & 1) == (L >> 3)) ? o8 : other8;
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
    case 7: return group_get_c<G, (G > 4 ? 7 : 0)>(v);
    case 8: return group_get_c<G, (G > 8 ? 8 : 0)>(v);
    case 9: return group_get_c<G, (G > 8 ? 9 : 0)>(v);
    case 10: return group_get_c<G, (G > 8 ? 10 : 0)>(v);
    case 11: return group_get_c<G, (G > 8 ? 11 : 0)>(v);
    case 12: return group_get_c<G, (G > 8 ? 12 : 0)>(v);
    case 13: return group_get_c<G, (G > 8 ? 13 : 0)>(v);
    case 14: return group_get_c<G, (G > 8 ? 14 : 0)>(v);
    default: return group_get_c<G, (G > 8 ? 15 : 0)>(v);
  }
}

// integer flavour of group_get
template <int G>
__device__ __forceinline__ int group_get_int(int v, int l) {
  return __double2loint(group_get<G>(__hiloint2double(0, v), l));
}

// ALL-GATHER within a group: out[l] = the value `v` of lane l of this lane's group, l = 0 .. J - 1 (what the recurrences need
// three times per cadence: U, W and the propagators of every state index).  By DPP broadcasts a value costs ~10 vector
// instructions on a group of eight and ~20 on a row of sixteen (quad broadcast, xor-4 and xor-8 exchanges with their selects, twice
// for the two halves of a double): 200 instructions per gather at J = 10 -- most of what the lane-group kernels of the wide states
// issued (round 6: element kernel 4.3 ms of a 12.7 ms step).  Through LDS it is one ds_write_b64 of the wave's 64 values and J / 2
// ds_read_b128 per lane (the lanes of a group read the same addresses: broadcasts).  A block of these kernels is ONE wave, so the
// barriers below are waits on the LDS counter, not s_barrier round trips.  EXO_GATHER_LDS_MIN_G: from which group width on.
// Measured at the C5 shape, 128 chains (tools/ab_gp.py, same box, alternating): rows of sixteen (J = 10) 12.7 -> 9.5 ms; groups of
// eight too: J = 8 5.08 -> 4.66 ms, two chains redone by the sequential kernels (J = 6 on eight lanes) 182 -> 170 ms.  Pure data
// movement: results bit for bit.  Groups of four and two keep the DPP moves (a quad broadcast is one instruction).
#ifndef EXO_GATHER_LDS_MIN_G
#define EXO_GATHER_LDS_MIN_G 8
#endif
// EXO_GATHER_SYNC = 0: no barrier at all.  A block is one wave, and the LDS operations of ONE wave execute in the order they were
// issued: the reads see the write in front of them, and the next gather's write cannot overtake these reads.  The two
// __syncthreads() this had were each a wait for the LDS counter to drain to zero -- every gather a full write -> read round
// trip with nothing else in flight, 7-9 of them per cadence in the reverse kernel; without them the compiler waits only where
// a value is used, and consecutive gathers overlap.
#ifndef EXO_GATHER_SYNC
#define EXO_GATHER_SYNC 0
#endif
template <int G, int J>
__device__ __forceinline__ void group_gather_lds(double v, double (&out)[J]) {
  __shared__ double s_gather[kWave];
  s_gather[threadIdx.x] = v;
  if (EXO_GATHER_SYNC) __syncthreads(); else __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (the compiler keeps the order too; no instruction)
  const double* __restrict__ base = s_gather + (threadIdx.x & ~(G - 1));
#pragma unroll
  for (int l = 0; l < J; ++l) out[l] = base[l];
  if (EXO_GATHER_SYNC) __syncthreads(); else __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (the next gather overwrites the buffer)
}
// (narrower groups keep the DPP loop exactly as it stood)
#define EXO_GROUP_GATHER(v, out)                                   \
  do {                                                             \
    if constexpr (G >= EXO_GATHER_LDS_MIN_G) {                     \
      group_gather_lds<G, J>(v, out);                              \
    } else {                                                       \
      _Pragma("unroll") for (int l_ = 0; l_ < J; ++l_) (out)[l_] = group_get<G>(v, l_); \
    }                                                              \
  } while (0)

// butterfly partner lane ^ M within the group
template <int M>
__device__ __forceinline__ double xor_get(double v) {
  if (M == 1) return dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  if (M == 2) return dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
  if (M == 4) return xor4(v);
  return xor8(v);
}

// sum over the G lanes of a group, result in every lane
// (after the two quad steps every lane of a quad holds the quad's sum, so the partner of the later steps may be ANY lane of the
// other quad / the other half row: the mirrored lane, one DPP move -- lane ^ 4 and lane ^ 8 take a shift each way and a select.
// The same two numbers are added: bit for bit the butterfly's sums.  Five of these per cadence in the reverse kernel.)
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += xor_get<1>(v);
  if (G >= 4) v += xor_get<2>(v);
  if (G >= 8) v += dpp_mov<kRowHalfMirror>(v);
  if (G >= 16) v += dpp_mov<kRowMirror>(v);
  return v;
}

// saved-state addressing.  Per (cadence, draw): the pair (d, z), 16 B; per (cadence, draw, j): one
// record (W_j, F_j, S_j0 .. S_j,J-1) of R = 2 + J doubles, stored PLANAR within the cadence: piece q
// of every (draw, j) is contiguous -- [cadence][piece][draw x j], pieces of 16 B for even R, of 8 B
// for odd R (whose cadence stride is not 16-B aligned).  Lanes of a wave are consecutive (draw, j), so
// every load / store instruction of a wave covers one contiguous kilobyte; record-major records
// made each instruction touch 16 B of every lane's record -- partial lines at the memory side.
struct StateIdx {
  int64_t n, n_draw;
  int J;
  __device__ __forceinline__ int64_t scal(int q, int64_t i, int64_t draw) const {  // q = 0 (d), 1 (z)
    return (i * n_draw + draw) * 2 + q;
  }
  // address of piece 0 of the record of (cadence i, draw, j); piece q is q * piece() doubles further
  __device__ __forceinline__ int64_t rec(int64_t i, int64_t draw, int j) const {
    const int64_t lanes = n_draw * J, lane = draw * J + j;
    return 2 * n * n_draw + i * lanes * (int64_t)(2 + J) + (((2 + J) % 2 == 0) ? 2 * lane : lane);
  }
  __device__ __forceinline__ int64_t piece() const { return n_draw * J * (int64_t)(((2 + J) % 2 == 0) ? 2 : 1); }
  // U_n, V_n, P_n (q = 0, 1, 2) written by the parallel pre-pass: the sequential kernels
  // never evaluate a sin, cos or exp
  __device__ __forceinline__ int64_t uvp(int q, int64_t i, int64_t draw, int j) const {
    return (2 + (int64_t)(2 + J) * J) * n * n_draw + (((int64_t)q * n + i) * n_draw + draw) * J + j;
  }
};

// one record (W, F, S row): p = piece 0 (StateIdx::rec), ps = StateIdx::piece().  Non-temporal:
// written once per forward pass, read once by the reverse pass, 10 GB-scale.
template <int J>
__device__ __forceinline__ void store_record(double* __restrict__ p, int64_t ps, double W, double F, const double* S) {
  constexpr int R = 2 + J;
  double v[R];
  v[0] = W; v[1] = F;
#pragma unroll
  for (int l = 0; l < J; ++l) v[2 + l] = S[l];
  typedef double v2d __attribute__((ext_vector_type(2)));
  if (R % 2 == 0) {
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      const v2d x = {v[2 * q], v[2 * q + 1]};
      __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(p + q * ps));
    }
  } else {
#pragma unroll
    for (int q = 0; q < R; ++q) __builtin_nontemporal_store(v[q], p + q * ps);
  }
}
template <int J>
__device__ __forceinline__ void load_record(const double* __restrict__ p, int64_t ps, double& W, double& F, double* S) {
  constexpr int R = 2 + J;
  double v[R];
  if (R % 2 == 0) {
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      const double2 x = *reinterpret_cast<const double2*>(p + q * ps);
      v[2 * q] = x.x; v[2 * q + 1] = x.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = p[q * ps];
  }
  W = v[0]; F = v[1];
#pragma unroll
  for (int l = 0; l < J; ++l) S[l] = v[2 + l];
}

// The lane-group CHUNK kernels (celerite_chunk_fwd_kernel / celerite_chunk_vjp_kernel) keep the S row of a record only at every
// lg_span-th cadence of a chunk -- a CHECKPOINT -- and the reverse kernel recomputes the rows in between (one fused multiply-add
// per entry and two all-gathers per cadence): the saved factorisation was 83 doubles per (draw, cadence) at J = 8, 123 at J = 10,
// written at 5.5 and read back at 4.8 TB/s -- both kernels sat on the HBM ceiling at 0.24-0.45 of their issue rate
// (profiles/r06_counters.json, legs j8 / j10).  The records keep their places (pieces 1 .. of the cadences in between are holes
// nobody touches; piece 0 -- (W, F) of every lane -- stays one contiguous run per cadence), so the sequential kernels, which
// write and read every S row of the draws they redo, share the layout unchanged.
// The span is a matter of registers: the reverse kernel holds span rows of J doubles (and span records) on top of its own 170-230,
// and what counts is staying at two waves per SIMD (256 registers): J = 10 with a span of 4 took 268 -- one wave -- and the step went
// from 7.5 to 8.7 ms where J = 8 (240: two waves, down from three) went from 4.04 to 3.59.
#ifndef EXO_LG_SPAN
#define EXO_LG_SPAN 0   // (> 0: that span for every width -- A/B builds)
#endif
template <int J>
constexpr int lg_span() { return EXO_LG_SPAN > 0 ? EXO_LG_SPAN : (J <= 9 ? 4 : (J <= 11 ? 3 : (J <= 14 ? 2 : 1))); }
// (W, F) of a record alone
template <int J>
__device__ __forceinline__ void store_record_wf(double* __restrict__ p, int64_t ps, double W, double F) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  if ((2 + J) % 2 == 0) {
    const v2d x = {W, F};
    __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(p));
  } else {
    __builtin_nontemporal_store(W, p);
    __builtin_nontemporal_store(F, p + ps);
  }
}
template <int J>
__device__ __forceinline__ void load_record_wf(const double* __restrict__ p, int64_t ps, double& W, double& F) {
  if ((2 + J) % 2 == 0) {
    const double2 x = *reinterpret_cast<const double2*>(p);
    W = x.x; F = x.y;
  } else {
    W = p[0]; F = p[ps];
  }
}
// the S row of a record alone
template <int J>
__device__ __forceinline__ void load_record_s(const double* __restrict__ p, int64_t ps, double* S) {
  constexpr int R = 2 + J;
  if (R % 2 == 0) {
#pragma unroll
    for (int q = 1; q < R / 2; ++q) {
      const double2 x = *reinterpret_cast<const double2*>(p + q * ps);
      S[2 * q - 2] = x.x; S[2 * q - 1] = x.y;
    }
  } else {
#pragma unroll
    for (int q = 2; q < R; ++q) S[q - 2] = p[q * ps];
  }
}

// Pre-pass, fully parallel over (cadence, draw, state index): everything in the
// recurrences that does not depend on the recurrence itself.
__global__ __launch_bounds__(256) void celerite_prep_kernel(const double* __restrict__ t, int64_t n,
                                                            Coefs cf,
                                                            int64_t n_draw, int J, double* __restrict__ state) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = n_draw * J;
  if (e >= n * per) return;
  const int64_t i = e / per, rem = e - i * per;
  const int64_t draw = rem / J;
  const int j = (int)(rem - draw * J);
  const LaneCoef k = lane_coef(cf, draw, j, J);
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
    const double* __restrict__ t, Series rs, const double* __restrict__ diag,
    int64_t n_diag, int64_t n, Coefs cf, int64_t n_draw, double* __restrict__ loglike,
    double* __restrict__ state, const double* __restrict__ only_flagged, ChunkGeom cg, int n_slice) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  // after the time-parallel path: redo only the draws it could not take (see DeltaCoef)
  const bool mine = live_draw && (!only_flagged || only_flagged[draw] >= kFlagSeq);
  if (only_flagged) {
    // ... and every other draw's log-likelihood from the chunks' partial sums (n_slice of them per quantity, in the slots of
    // chunks 0 .. n_slice - 1: celerite_chunk_slice_sum_kernel) -- round 5: this launch is there anyway, a kernel of its own for
    // the sums' last step was 7 us of the step
    if (live_draw && !mine && j == 0) {
      const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
      double acc = 0.0, logdet = 0.0, bad = 0.0;
      for (int q = 0; q < n_slice; ++q) {
        acc += state[ws.part(q, 0, draw)];
        logdet += state[ws.part(q, 1, draw)];
        bad += state[ws.part(q, 2, draw)];
      }
      loglike[cf.at(draw)] = (bad > 0.0) ? -INFINITY : fma(-0.5, acc + logdet, -(double)n * kHalfLog2Pi);
    }
    if (__ballot(mine) == 0) return;
  }
  const LaneCoef k = lane_coef(cf, draw, j, J);
  const bool store = SAVE && mine && k.live;
  // a_n = diag_n + sum of the a coefficients (first index of each term)
  const double asum = group_sum<G>((k.live && !k.odd) ? k.a : 0.0);
  const StateIdx six{n, n_draw, J};
  const SeriesRow y(rs, draw, n);
  const double* __restrict__ dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);

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
  EXO_GROUP_GATHER(Wj, Wall);
  // sum log d_n = log prod d_n: carry the product as (mantissa, exponent) -- a multiply
  // and a frexp per cadence instead of a ~45-instruction log on the sequential chain
  double acc = z * z / d;
  int lexp;
  double lman = frexp(bad ? 1.0 : d, &lexp);
  int64_t lsum = lexp;
  double dt_prev = -1.0;
  if (store) {
    if (j == 0) *reinterpret_cast<double2*>(state + six.scal(0, 0, draw)) = double2{d, z};
    store_record<J>(state + six.rec(0, draw, j), six.piece(), Wj, 0.0, Srow);   // S_0 = 0
  }
  // software prefetch ring: the loads of cadence i + kPF are issued while cadence i
  // computes (one wave per SIMD and a serial chain: nothing else hides HBM latency)
  const int64_t vstride = n_draw * J;            // one cadence, in doubles, of a per-(draw, j) quantity (U, V, P)
  const int64_t qstride = n * vstride;           // one quantity (U, V, P)
  const int64_t rstride = vstride * (2 + J);     // one cadence of (W, F, S row) records
  double* __restrict__ p_vec = SAVE ? state + six.rec(0, draw, jj) : nullptr;
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
      EXO_GROUP_GATHER(Pj, Pall);
    } else if (dt != dt_prev) {  // wave-uniform: evenly sampled series reuse P
      Pj = k.live ? exp(-k.c * dt) : 0.0;
      EXO_GROUP_GATHER(Pj, Pall);
      dt_prev = dt;
    }
    // row j of  S <- (P P^T) o (S + d W W^T) ;  F_j <- P_j (F_j + W_j z)
    Fj = Pj * fma(Wj, z, Fj);
    const double dwj = d * Wj;
#pragma unroll
    for (int l = 0; l < J; ++l) Srow[l] = Pj * Pall[l] * fma(dwj, Wall[l], Srow[l]);
    if (SAVE) { Uj = Ui; Vj = Vi; } else { lane_uv(k, ti, &Uj, &Vj, &cs, &sn); }
    EXO_GROUP_GATHER(Uj, Uall);
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
    EXO_GROUP_GATHER(Wj, Wall);
    acc = fma(z * z, id, acc);
    lman = frexp(lman * (d > 0.0 ? d : 1.0), &lexp);
    lsum += lexp;
    if (store) {
      // running pointers: one add per cadence instead of a 64-bit index product per access
      p_vec += rstride;
      p_scal += 2 * n_draw;
      if (j == 0) *reinterpret_cast<double2*>(p_scal) = double2{d, z};
      store_record<J>(p_vec, six.piece(), Wj, Fj, Srow);
    }
   }
  }
  if (mine && j == 0) {
    // not positive definite -> -inf in band (a sampler rejects the point)
    const double logdet = log(lman) + (double)lsum * 0.69314718055994530942;
    loglike[cf.at(draw)] = bad ? -INFINITY : fma(-0.5, acc + logdet, -(double)n * kHalfLog2Pi);
  }
}

// Reverse recurrence (hand-derived adjoint of the two recurrences above), same lane
// layout: lane j owns row j of the SYMMETRISED adjoint of S (S is symmetric, so only
// the symmetric part of its adjoint matters), Fb_j, Wb_j and the coefficient
// cotangents of its state index.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_vjp_kernel(
    const double* __restrict__ t, const double* __restrict__ diag, int64_t n_diag, int64_t n,
    Coefs cf,
    int64_t n_draw, const double* __restrict__ gloglike, const double* __restrict__ state,
    double* __restrict__ gresid, double* __restrict__ gdiag, double* __restrict__ gdiag_sum,
    double* __restrict__ gcoef_real, double* __restrict__ gcoef_complex, const double* __restrict__ only_flagged,
    double gsign, Series rs, ChunkGeom cg) {   // rs: the series this is the cotangent of (its layout is the cotangent's: GradRow)
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = (lane_draw < n_draw) && (!only_flagged || only_flagged[lane_draw < n_draw ? lane_draw : 0] >= kFlagSeq);
  if (only_flagged) {
    // ... and every other draw's coefficient cotangents from the sums over the chunks (totals in chunk 0's slots): gcoef_lane,
    // the combination of this kernel's own tail, on the lanes that would otherwise leave at once
    if (lane_draw < n_draw && !live_draw && j < J)
      gcoef_lane(cf, n, n_draw, state, cg, 1, gdiag_sum, gcoef_real, gcoef_complex, lane_draw, j);
    if (__ballot(live_draw) == 0) return;
  }
  const int64_t draw = lane_draw < n_draw ? lane_draw : n_draw - 1;
  const LaneCoef k = lane_coef(cf, draw, j, J);
  const int jj = k.live ? j : 0;  // idle lanes read a valid slot and contribute zeros
  const StateIdx six{n, n_draw, J};
  const double gL = gloglike[cf.at(draw)];
  const bool lead = live_draw && j == 0;
  GradRow grow(gresid, rs, draw, n);
  // the other state index of this lane's complex pair (itself for real terms / idle lanes)
  const int partner = (int)threadIdx.x + ((k.live && !k.real) ? (k.odd ? -1 : 1) : 0);

  double Sb[J];  // row j of the symmetric adjoint of S
#pragma unroll
  for (int l = 0; l < J; ++l) Sb[l] = 0.0;
  double Fb = 0.0, Wb = 0.0, db = 0.0, zb = 0.0, gasum = 0.0;
  double ga = 0.0, gb = 0.0, gc = 0.0, gd = 0.0;

  const int64_t vstride = n_draw * J, qstride = n * vstride;
  const double* __restrict__ sc0 = state + six.scal(0, 0, draw);
  const double* __restrict__ ve0 = state + six.rec(0, draw, jj);
  const double* __restrict__ uv0 = state + six.uvp(0, 0, draw, jj);
  const int64_t rstride = vstride * (2 + J);   // one cadence of (W, F, S row) records
  auto load = [&](int64_t i, double& d_, double& z_, double& W_, double& F_, double* S_) {
    const double2 dz = *reinterpret_cast<const double2*>(sc0 + i * 2 * n_draw);
    d_ = dz.x; z_ = dz.y;
    load_record<J>(ve0 + i * rstride, six.piece(), W_, F_, S_);   // idle lanes read lane 0's record: harmless, zeroed below
    if (!k.live) {
      W_ = F_ = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) S_[l] = 0.0;
    }
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
    EXO_GROUP_GATHER(Pj, Pall);
    // cos / sin of this lane's complex pair: its own V and its partner's
    const double cs = k.odd ? Vo : Vj, sn = k.odd ? Vj : Vo;
    double Uall[J], Wpall[J];
    EXO_GROUP_GATHER(Uj, Uall);
    EXO_GROUP_GATHER(W_p, Wpall);
    const double id = 1.0 / d_n;
    // (5) log-likelihood terms, (4) z_n = y_n - U.F_n
    const double zbar = zb - gL * z_n * id;
    const double wdot = group_sum<G>(Wb * W_n);
    const double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id) - wdot * id;
    if (lead) {
      grow.store(i, gsign * zbar);
      if (gdiag) gdiag[cf.at(draw) * n + i] = dbar;
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
    EXO_GROUP_GATHER(ubj, uball);
    double acc_u = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      Sb[l] = fma(0.5, fma(ubj, Uall[l], uball[l] * Uj), Sb[l]);  // symmetrised  ub U^T
      acc_u = fma(S_n[l], uball[l], acc_u);                        // (S_n^T ub)_j, S symmetric
    }
    Ub += acc_u;
    // d loglike / d(oscillation rate of a complex pair): - sum over the links of dt x (the phase FLUX across the link), the flux
    // taken from what it is -- Fbar^T G F + < Sbar, G S + S G^T >, G = [[0, -1], [1, 0]] on the pair, the state entering this
    // cadence against its adjoint (phase_flux, exo_celerite_core.hpp) -- at EVERY link: no sum of phase cotangents weighted with
    // their distance from the origin (sum_i (t_i - t_0) g_i lost up to 4e-5 on kernels with aliased oscillation rates -- the
    // sequential recurrences of celerite2's formulation do; tools/gp_seq_dc_check.py).  Lane j of a pair has row j of S and of
    // Sbar; the partner lane (j ^ 1) supplies its row of Sbar and its own cross term.
    double flux = 0.0;
    if (G >= 2) {
      // the pair's other lane: j + 1 for its first index, j - 1 for its second (an odd number of real terms in front leaves
      // the pairs on odd lanes: not lane ^ 1) -- both moves by every lane, then the choice (dpp inside a branch: see grp_argmax8)
      auto other = [&](double v) {
        const double up = dpp_mov<0x101>(v), dn = dpp_mov<0x111>(v);   // row_shl 1: lane i reads i + 1; row_shr 1: i - 1
        return k.odd ? dn : up;
      };
      double cross = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) cross = fma(other(Sb[l]), S_n[l], cross);
      const double cross_o = other(cross), F_o = other(F_n), Fb_o = other(Fb);
      flux = fma(Fb_o, F_n, -Fb * F_o) + 2.0 * (cross - cross_o);
    }
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
      const double Ub_o = __shfl(Ub, partner, 64);
      if (k.live && !k.real && !k.odd) {
        ga += Ub * cs + Ub_o * sn;
        gb += Ub * sn - Ub_o * cs;
        gd = fma(-dt, flux, gd);
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
      grow.store(0, gsign * zbar);
      if (gdiag) gdiag[cf.at(draw) * n] = dbar;
    }
    gasum += dbar;
    // (cadence 0's phase cotangent has no link in front of it -- the phases are counted from t_0, Coefs::origin -- and its
    // U, V carry no other parameter: nothing more to collect)
  }
  // the decay rate of a complex pair is shared by its two state indices
  const double gc_o = __shfl(gc, partner, 64);
  if (!live_draw || !k.live) return;
  if (j == 0 && gdiag_sum) gdiag_sum[cf.at(draw)] = gasum;
  if (k.real) {
    // a real term of its own, or one of the two real terms of a pair slot (kind 1)
    double* o = k.slot < 0 ? gcoef_real + (cf.at(draw) * cf.n_real + j) * 2 : gcoef_complex + cf.at(draw) * cf.n_complex * 4 + k.slot;
    o[0] = ga + gasum;  // a_n = diag_n + sum a
    o[1] = gc;
  } else if (!k.odd) {
    double* o = gcoef_complex + (cf.at(draw) * cf.n_complex + ((j - cf.n_real) >> 1)) * 4;
    o[0] = ga + gasum;
    o[1] = gb;
    o[2] = gc + gc_o;
    o[3] = gd;
  }
}


// ===========================================================================
// Time-parallel path.  The recurrences above are a Kalman filter in disguise: with a
// symmetric Delta_n such that Delta_n U_n = V_n (the state covariance of the process the kernel
// describes, in the rotating frame celerite uses),
//     P_n = Delta_n - S_n   is the covariance of the one-step prediction of the state,
//     F_n                   is its mean,
//     d_n = diag_n + U_n^T P_n U_n,  W_n = P_n U_n / d_n,  z_n   the innovation variance, gain, innovation
// (diag_n = a_n - U_n^T V_n is the measurement variance).  A run of cadences therefore acts on
// (F, P) as a filtering element (A, b, C, eta, J) of Sarkka & Garcia-Fernandez (2021):
//     F' = A (I + P J)^-1 (F + P eta) + b,      P' = A (I + P J)^-1 P A^T + C,
// built by running the filter from (0, 0) next to two sensitivity matrices, and the run's
// log-likelihood as a function of its entering state has the closed form
//     const - 1/2 log det(I + P J) + 1/2 eta^T Y P eta + eta^T Y F - 1/2 F^T J Y F,  Y = (I + P J)^-1.
// So: split the series into C chunks; (A) build every chunk's element in parallel; (B) walk the
// C elements per draw to get the state ENTERING each chunk; (C) run the ordinary recurrences inside
// every chunk in parallel from that state.  The reverse pass needs the adjoint of the entering
// state of every chunk, and the chain rule across chunks only needs the same elements:
//     Fbar  += Y^T (eta - J F) gL + Abar^T Fbar',            Abar = A Y
//     Pbar  += 1/2 (w w^T - J Y) gL + Abar^T Pbar' Abar + sym(Abar^T Fbar' g^T),   g = eta - J Y (F + P eta)
// (B'), after which the ordinary reverse recurrence runs inside every chunk in parallel (C').
// Nothing is differentiated through (A), (B): they only supply boundary values of quantities the
// sequential algorithm also has, so the parameter cotangents are still summed by the ordinary
// reverse recurrence.  Work is ~2x the sequential algorithm's, spread over C x more lanes.
//
// Delta: a real term (a, c): 1 / a.  A complex pair (a, b, c, d) with (cs, sn) = (cos d t, sin d t):
// H Delta0 H, H = [[cs, sn], [sn, -cs]], Delta0 = [[p, q], [q, r]], r = a / (a^2 + b^2),
// q = -b / (a^2 + b^2), p = (a^2 + 2 b^2) / (a (a^2 + b^2)) -- the member of the family
// Delta0 (a, b)^T = (1, 0)^T for which the process noise Delta_{n+1} - phi^2 Delta_n is positive
// semi-definite exactly when the term is a valid kernel (|b d| <= a c).  Draws with a term outside
// a > 0, |b d| <= a c are flagged and handled by the sequential kernels.
// ===========================================================================

// which draws the time-parallel path cannot take (DeltaCoef::valid): kFlagSeq = redo sequentially (the element kernel adds
// its own verdicts on the conditioning: kFlagRobust / kFlagSeq, exo_celerite_core.hpp)
template <int J>
__global__ __launch_bounds__(kWave) void celerite_flag_kernel(Coefs cf,
                                                              int64_t n_draw, double* __restrict__ flag) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  DeltaCoef<J> dc;
  dc.init(cf, draw);
  flag[draw] = dc.valid ? kFlagClean : kFlagSeq;
}

// pre-pass for the flagged draws only (the sequential kernels read U, V, P from `state`; the
// time-parallel kernels recompute them: three fewer arrays across HBM four times)
__global__ __launch_bounds__(256) void celerite_prep_flagged_kernel(const double* __restrict__ t, int64_t n,
                                                                    Coefs cf, int64_t n_draw, int J,
                                                                    double* __restrict__ state,
                                                                    const double* __restrict__ flag) {
  const int64_t draw = blockIdx.y;
  if (flag[draw] < kFlagSeq) return;   // (clean, or finished on the robust route of the time-parallel path)
  const StateIdx six{n, n_draw, J};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * J; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / J;
    const int j = (int)(e - i * J);
    const LaneCoef k = lane_coef(cf, draw, j, J);
    double U, V, cs, sn;
    const double ti = t[i];
    lane_uv(k, ti, &U, &V, &cs, &sn);
    state[six.uvp(0, i, draw, j)] = U;
    state[six.uvp(1, i, draw, j)] = V;
    state[six.uvp(2, i, draw, j)] = i > 0 ? exp(-k.c * (ti - t[i - 1])) : 1.0;
  }
}

// row j of Delta_n for the lane that owns state index j (see DeltaCoef for the formulas)
struct LaneDelta {
  double dp, dq, dr;
  __device__ __forceinline__ explicit LaneDelta(const LaneCoef& k) : dp(0.0), dq(0.0), dr(0.0) {
    if (!k.live) return;
    if (k.real) {
      dp = 1.0 / k.a;
    } else {
      const double h = 1.0 / (k.a * k.a + k.b * k.b);
      dr = k.a * h; dq = -k.b * h; dp = (k.a * k.a + 2.0 * k.b * k.b) * h / k.a;
    }
  }
  // cs, sn: cos / sin (d t_n) of the lane's pair (from lane_uv)
  // Branch-free: a row has its diagonal entry and, for a pair, the entry of the pair's other index -- two values worked out once,
  // then two selects per entry.  (A real or idle lane has dq = dr = 0 and (cs, sn) = (1, 0) from lane_uv: the pair formulas give
  // dp and 0 there.)  As nested ifs inside the unrolled loop this was ~4 divergent branch regions per ENTRY in the element
  // kernel's cadence loop: 47 of them per cadence at J = 10, a third of the 1030 instructions it issued.
  // the two entries of my row that are not zero: the diagonal one, vd, and (a pair) the one at the pair's other index, vo
  __device__ __forceinline__ void two(const LaneCoef& k, double cs, double sn, double* vd, double* vo) const {
    const double v_even = dp * cs * cs + 2.0 * dq * cs * sn + dr * sn * sn;
    const double v_odd = dp * sn * sn - 2.0 * dq * cs * sn + dr * cs * cs;
    const double v_off = (dp - dr) * cs * sn + dq * (sn * sn - cs * cs);
    *vd = k.live ? (k.real ? dp : (k.odd ? v_odd : v_even)) : 0.0;
    *vo = (k.live && !k.real) ? v_off : 0.0;
  }
  // the pair's other index (-1: none)
  static __device__ __forceinline__ int other(const LaneCoef& k, int j) { return (k.live && !k.real) ? (k.odd ? j - 1 : j + 1) : -1; }
  template <int J>
  __device__ __forceinline__ void row(const LaneCoef& k, int j, double cs, double sn, double* D) const {
    double vd, vo;
    two(k, cs, sn, &vd, &vo);
    const int jo = other(k, j);
#pragma unroll
    for (int l = 0; l < J; ++l) D[l] = (l == j) ? vd : ((l == jo) ? vo : 0.0);
  }
};

// The REFERENCE STEP of a chunk for one state index (DrawCoef::step / rot_uv of the one-lane kernels, on a lane).  The lane-group
// kernels used to take sin / cos of d (t - t0) and exp(-c dt) afresh at every cadence -- ~130 of the 500-700 instructions a
// cadence issued at J = 10, the propagators re-evaluated two steps out of three because "evenly sampled" steps differ in their
// last bits.  Here the chunk's first step is the reference: exp(-c dt_ref), (cos, sin)(d dt_ref) once, and a step within 2^-20
// of it takes them corrected to second order in del = dt - dt_ref (exact to 1e-16 when max(|c|, |d|) dt_ref <= 8); the pair's
// (cos, sin) are ROTATED from the neighbouring cadence and taken exactly at every fourth cadence of the chunk (three rotations
// in a row drift by ~3e-16), at gaps, and for terms too fast for the correction.  Two equal steps off the reference in a row
// become the new reference.
#ifndef EXO_LG_REF_STEP
#define EXO_LG_REF_STEP 1
#endif
struct LaneStepper {
  double dt_ref = -1.0, dt_miss = -1.0, ph = 1.0, rc = 1.0, rs = 0.0;
  bool near_ok = false;
  __device__ __forceinline__ void set_ref(const LaneCoef& k, double dt) {
    dt_ref = dt;
    ph = k.live ? exp(-k.c * dt) : 0.0;
    rc = 1.0; rs = 0.0;
    if (k.live && !k.real) exo::sincos_any(k.d * dt, &rs, &rc);
    near_ok = fmax(fabs(k.c), fabs(k.d)) * dt <= 8.0 && dt > 0.0;
  }
  // the propagator of the step dt; true: the step is near the reference (rot / rot_back may follow)
  __device__ __forceinline__ bool step(const LaneCoef& k, double dt, double* P) {
    constexpr double kTol = 9.5367431640625e-07;   // 2^-20
    if (!EXO_LG_REF_STEP) {
      *P = k.live ? exp(-k.c * dt) : 0.0;
      return false;
    }
    if (dt_ref < 0.0) set_ref(k, dt);
    bool near = near_ok && fabs(dt - dt_ref) <= kTol * dt_ref;
    if (!near) {
      if (dt_miss > 0.0 && fabs(dt - dt_miss) <= kTol * dt_miss) {
        set_ref(k, dt);
        near = near_ok;
      }
      dt_miss = dt;
    }
    if (near) {
      const double x = k.c * (dt - dt_ref);
      *P = ph * fma(x, fma(0.5, x, -1.0), 1.0);
    } else {
      *P = k.live ? exp(-k.c * dt) : 0.0;
    }
    return near;
  }
  // the pair's (cos, sin) one near step dt LATER / EARLIER than the cadence they belong to (a real or idle lane's (1, 0) stays)
  __device__ __forceinline__ void rot(const LaneCoef& k, double dt, double& cs, double& sn) const {
    const double c1 = fma(cs, rc, -sn * rs), s1 = fma(sn, rc, cs * rs);
    const double e = k.d * (dt - dt_ref), h = fma(-0.5 * e, e, 1.0);
    cs = fma(-e, s1, c1 * h);
    sn = fma(e, c1, s1 * h);
  }
  __device__ __forceinline__ void rot_back(const LaneCoef& k, double dt, double& cs, double& sn) const {
    const double c1 = fma(cs, rc, sn * rs), s1 = fma(sn, rc, -cs * rs);
    const double e = k.d * (dt - dt_ref), h = fma(-0.5 * e, e, 1.0);
    cs = fma(e, s1, c1 * h);
    sn = fma(-e, c1, s1 * h);
  }
};
// U_j, V_j from the pair's (cos, sin) (lane_uv's last lines)
__device__ __forceinline__ void lane_uv_from(const LaneCoef& k, double cs, double sn, double* U, double* V) {
  *U = k.odd ? (k.a * sn - k.b * cs) : (k.a * cs + k.b * sn);
  *V = k.live ? (k.odd ? sn : cs) : 0.0;
}

// (A) in lane-group form: the element of a (draw, chunk) on the draw's G lanes, lane j owning row j of
// A, Cm, Jm.  Same arithmetic as celerite_elem_kernel; the column sums A^T U travel by a butterfly,
// Cm U / the gain by DPP broadcasts.  A lane carries 3 J + 2 doubles of element state instead of
// 3 J^2 + 2 J (J = 6: the one-lane version needs 256 registers and scratch), and there are G times
// more lanes to hide the loads.  Used for J >= 3, and the only version for J = 7, 8.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_elem_lg_kernel(
    const double* __restrict__ t, Series rs, const double* __restrict__ diag, int64_t n_diag,
    int64_t n, Coefs cf, int64_t n_draw, double* __restrict__ state, ChunkGeom cg, int64_t flag_at) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  const int c = blockIdx.y;
  const bool empty = c * cg.L >= n;   // a fine chunk past the end of the series: the identity element
  const int64_t n0 = empty ? n - 1 : c * cg.L, n1 = empty ? n0 : ((n0 + cg.L < n) ? n0 + cg.L : n);
  const LaneCoef k = lane_coef(cf, draw, j, J);
  const bool live = k.live;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const LaneDelta ld(k);
  const SeriesRow y(rs, draw, n);
  const double* __restrict__ dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  // conditioning score (see celerite_elem_kernel)
  const double asum = group_sum<G>((live && !k.odd) ? fabs(k.a) : 0.0);
  double ba2 = (live && !k.real && !k.odd) ? (k.b * k.b) / (k.a * k.a) : 0.0;
  if (G >= 2) ba2 = fmax(ba2, xor_get<1>(ba2));
  if (G >= 4) ba2 = fmax(ba2, xor_get<2>(ba2));
  if (G >= 8) ba2 = fmax(ba2, xor_get<4>(ba2));
  if (G >= 16) ba2 = fmax(ba2, xor_get<8>(ba2));
  const double rmin = (1.0 + ba2) * asum * (1.0 / EXO_GP_COND_MAX);
  bool ok = true;

  // lane j holds COLUMN j of A (so that (A^T U)_j is a local dot product: no butterfly) and rows j of
  // Cm, Jm
  // (my row of Delta has two entries that are not zero -- LaneDelta::two: the previous cadence's are two doubles, not a row of J:
  // the kernel held 80 + 16 J registers, one wave per SIMD from J = 11 and two from J = 6)
  double Acol[J], Crow[J], Jrow[J], phiall[J];
#pragma unroll
  for (int l = 0; l < J; ++l) { Acol[l] = (live && l == j) ? 1.0 : 0.0; Crow[l] = 0.0; Jrow[l] = 0.0; phiall[l] = 1.0; }
  double bj = 0.0, etaj = 0.0, phi = 1.0;
  double ti = t[n0];
  double Uj, Vj, cs, sn;
  lane_uv(k, ti, &Uj, &Vj, &cs, &sn);
  const int jo = LaneDelta::other(k, j);
  double vd_l, vo_l;
  ld.two(k, cs, sn, &vd_l, &vo_l);
  LaneStepper stp;
#pragma unroll 1
  for (int64_t i = n0; i < n1; ++i) {
    const double yi = y[i], R = dg[i];
    ok = ok && (R >= rmin) && (R < INFINITY);
    double Uall[J];
    EXO_GROUP_GATHER(Uj, Uall);
    double rj = 0.0, cuj = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      rj = fma(Acol[l], Uall[l], rj);            // (A^T U)_j
      cuj = fma(Crow[l], Uall[l], cuj);          // (Cm U)_j
    }
    const double s = R + group_sum<G>(Uj * cuj);
    const double zeta = yi - group_sum<G>(Uj * bj);
    const double is = exo::fast_rcp(s);
    double rall[J];
    EXO_GROUP_GATHER(rj, rall);
    const double ris = rj * is;
    etaj = fma(ris, zeta, etaj);
#pragma unroll
    for (int l = 0; l < J; ++l) Jrow[l] = fma(ris, rall[l], Jrow[l]);
    if (i + 1 < n) {
      const double tn = t[i + 1], dt = tn - ti;
      ti = tn;
      const bool near = stp.step(k, dt, &phi);
      EXO_GROUP_GATHER(phi, phiall);
      if (near && ((i + 1 - n0) & 3) != 0) {
        stp.rot(k, dt, cs, sn);
        lane_uv_from(k, cs, sn, &Uj, &Vj);
      } else {
        lane_uv(k, tn, &Uj, &Vj, &cs, &sn);
      }
      double vd_n, vo_n;
      ld.two(k, cs, sn, &vd_n, &vo_n);
      const double kj = cuj * is;            // the gain of my state index
      bj = phi * fma(kj, zeta, bj);
      // what the updates need of the OTHER indices is their gains k_l = (Cm U)_l / s (gathered once, instead of (Cm U)_l with a
      // multiplication by 1 / s per entry):  A[l][j] = phi_l (A[l][j] - k_l r_j),   Cm[j][l] = phi_j phi_l (Cm[j][l] - (Cm U)_j k_l) + Q
      double kall[J];
      EXO_GROUP_GATHER(kj, kall);
#pragma unroll
      for (int l = 0; l < J; ++l) {
        Acol[l] = phiall[l] * fma(-kall[l], rj, Acol[l]);
        const double dl = (l == j) ? vd_l : ((l == jo) ? vo_l : 0.0), dn = (l == j) ? vd_n : ((l == jo) ? vo_n : 0.0);
        Crow[l] = fma(phi * phiall[l], fma(-cuj, kall[l], Crow[l]) - dl, dn);   // + Q = Dn - phi phi Dl
      }
      vd_l = vd_n; vo_l = vo_n;
    }
  }
  if (!live_draw || !live) return;
  if (!ok && j == 0) state[(flag_at >= 0 ? flag_at : ws.off_flag()) + draw] = kFlagSeq;   // (no robust route on the lane-group path)
  const int E1 = J * J, E2 = J * J + J, E3 = 2 * J * J + J, E4 = 2 * J * J + 2 * J;
#pragma unroll
  for (int l = 0; l < J; ++l) {
    state[ws.elem(c, l * J + j, draw)] = Acol[l];          // A[l][j]
    state[ws.elem(c, E2 + j * J + l, draw)] = Crow[l];
    state[ws.elem(c, E4 + j * J + l, draw)] = Jrow[l];
  }
  state[ws.elem(c, E1 + j, draw)] = bj;
  state[ws.elem(c, E3 + j, draw)] = etaj;
}

// (C) the ordinary recurrences inside every chunk, from the entering state: same lane layout as
// celerite_fwd_kernel (a draw on G lanes), one wave per (64 / G draws, chunk).  With C x more
// waves than the sequential kernel the loads are hidden by occupancy: no prefetch ring.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_chunk_fwd_kernel(
    const double* __restrict__ t, Series rs, const double* __restrict__ diag, int64_t n_diag,
    int64_t n,
    Coefs cf,
    int64_t n_draw, double* __restrict__ state, ChunkGeom cg) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  const int c = blockIdx.y;
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const LaneCoef k = lane_coef(cf, draw, j, J);
  const bool store = live_draw && k.live;
  const double asum = group_sum<G>((k.live && !k.odd) ? k.a : 0.0);
  const StateIdx six{n, n_draw, J};
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const SeriesRow y(rs, draw, n);
  const double* __restrict__ dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  const int jj = k.live ? j : 0;

  double Srow[J], Wall[J], Uall[J], Pall[J];
  {
    // entering state from (B) as (F, P): S = Delta_{n0} - P
    const LaneDelta ld(k);
    double U_, V_, cs, sn;
    lane_uv(k, t[n0], &U_, &V_, &cs, &sn);
    ld.row<J>(k, j, cs, sn, Srow);
#pragma unroll
    for (int l = 0; l < J; ++l) { Srow[l] -= k.live ? state[ws.bnd(1, c, J + jj * J + l, draw)] : 0.0; Wall[l] = 0.0; }
  }
  double Fj = k.live ? state[ws.bnd(1, c, jj, draw)] : 0.0;
  double Wj = 0.0, d = 1.0, z = 0.0;
  bool bad = false;
  double acc = 0.0, lman = 1.0;
  int64_t lsum = 0;
  const int64_t rstride = n_draw * J * (int64_t)(2 + J);   // one cadence of (W, F, S row) records
  double* __restrict__ p_vec = state + six.rec(n0, draw, jj);
  double* __restrict__ p_scal = state + six.scal(0, n0, draw);
  double tprev = t[n0], Pj = 1.0;
  double cs_ = 1.0, sn_ = 0.0;
  LaneStepper stp;
#pragma unroll 1
  for (int64_t i = n0; i < n1; ++i) {
    // U, V, P recomputed, not read: three arrays fewer through HBM (LaneStepper: from the chunk's reference step)
    const double ti = t[i], dt = ti - tprev;
    tprev = ti;
    double Uj, Vj;
    bool near = false;
    if (i > n0) near = stp.step(k, dt, &Pj);
    if (near && ((i - n0) & 3) != 0) {
      stp.rot(k, dt, cs_, sn_);
      lane_uv_from(k, cs_, sn_, &Uj, &Vj);
    } else {
      lane_uv(k, ti, &Uj, &Vj, &cs_, &sn_);
    }
    const double yi = y[i], gi = dg[i];
    if (i > n0) {
      EXO_GROUP_GATHER(Pj, Pall);
      Fj = Pj * fma(Wj, z, Fj);
      const double dwj = d * Wj;
#pragma unroll
      for (int l = 0; l < J; ++l) Srow[l] = Pj * Pall[l] * fma(dwj, Wall[l], Srow[l]);
    }
    EXO_GROUP_GATHER(Uj, Uall);
    double uj = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) uj = fma(Srow[l], Uall[l], uj);
    const double pd = group_sum<G>(Uj * uj), pz = group_sum<G>(Uj * Fj);
    d = gi + asum - pd;
    z = yi - pz;
    bad = bad || !(d > 0.0);
    const double id = exo::fast_rcp(d);
    Wj = (Vj - uj) * id;
    EXO_GROUP_GATHER(Wj, Wall);
    acc = fma(z * z, id, acc);
    int lexp;
    lman = frexp(lman * (d > 0.0 ? d : 1.0), &lexp);
    lsum += lexp;
#ifndef EXO_EXP_NOSTORE
    if (store) {
      if (j == 0) *reinterpret_cast<double2*>(p_scal) = double2{d, z};
      if ((i - n0) % lg_span<J>() == 0) store_record<J>(p_vec, six.piece(), Wj, Fj, Srow);   // a checkpoint: the S row too
      else store_record_wf<J>(p_vec, six.piece(), Wj, Fj);
    }
#endif
    p_vec += rstride; p_scal += 2 * n_draw;
  }
  if (live_draw && j == 0) {
    state[ws.part(c, 0, draw)] = acc;
    state[ws.part(c, 1, draw)] = log(lman) + (double)lsum * 0.69314718055994530942;
    state[ws.part(c, 2, draw)] = bad ? 1.0 : 0.0;
  }
}

// (C') the ordinary reverse recurrence inside every chunk.  It starts from the adjoint of the state
// entering the NEXT chunk (from (B')) with the reverse of the propagation step into that chunk, and
// ends with the measurement half of the chunk's first cadence.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_chunk_vjp_kernel(
    const double* __restrict__ t, int64_t n, Coefs cf, int64_t n_draw, const double* __restrict__ gloglike,
    double* __restrict__ state, ChunkGeom cg, double* __restrict__ gresid, double* __restrict__ gdiag,
    double gsign, Series rs) {
  constexpr int G = Group<J>::G;
  const bool gcm = rs.cm != 0 || rs.sp.nseg != nullptr;   // the cotangent is not a row of consecutive cadences: stored one by one
  const int j = threadIdx.x & (G - 1);
  const int64_t lane_draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  const bool live_draw = lane_draw < n_draw;
  const int64_t draw = live_draw ? lane_draw : n_draw - 1;
  const int c = blockIdx.y;
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const LaneCoef k = lane_coef(cf, draw, j, J);
  const int jj = k.live ? j : 0;
  const StateIdx six{n, n_draw, J};
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const double gL = gloglike[cf.at(draw)];
  const int partner = (int)threadIdx.x + ((k.live && !k.real) ? (k.odd ? -1 : 1) : 0);
  GradRow grow(gresid, rs, draw, n);

  const double gsc = cg.prep ? gL : 1.0;   // (ChunkGeom::prep: the scan's adjoints are those of a cotangent of one)
  double Sb[J];
#pragma unroll
  for (int l = 0; l < J; ++l) Sb[l] = k.live ? gsc * state[ws.bnd(2, c, J + jj * J + l, draw)] : 0.0;
  double Fb = k.live ? gsc * state[ws.bnd(2, c, jj, draw)] : 0.0;
  double Wb = 0.0, db = 0.0, zb = 0.0, gasum = 0.0;
  double ga = 0.0, gb = 0.0, gc = 0.0, gd = 0.0;

  const int64_t vstride = n_draw * J, qstride = n * vstride;
  const double* __restrict__ sc0 = state + six.scal(0, 0, draw);
  const double* __restrict__ ve0 = state + six.rec(0, draw, jj);
  const int64_t rstride = vstride * (2 + J);   // one cadence of (W, F, S row) records
  // (d, z, W, F) of cadence i; the S row of a CHECKPOINT cadence (lg_span: every span-th of a chunk)
  auto load_wf = [&](int64_t i, double& d_, double& z_, double& W_, double& F_) {
    const double2 dz = *reinterpret_cast<const double2*>(sc0 + i * 2 * n_draw);
    d_ = dz.x; z_ = dz.y;
    load_record_wf<J>(ve0 + i * rstride, six.piece(), W_, F_);   // idle lanes read lane 0's record: harmless, zeroed below
    if (!k.live) W_ = F_ = 0.0;
  };
  auto load_s = [&](int64_t i, double* S_) {
    load_record_s<J>(ve0 + i * rstride, six.piece(), S_);
    if (!k.live) {
#pragma unroll
      for (int l = 0; l < J; ++l) S_[l] = 0.0;
    }
  };
  LaneStepper stp, stf;   // (stf: the propagators of the recomputed steps, taken in forward order)
  bool near_last = false;   // the step reversed last -- (i - 1) -> i -- was near the reference: cadence i - 1's (cos, sin) by rot_back
  constexpr int kPer = G >= 8 ? 1 : 8 / G;   // (G = 16: lanes 0 .. 7 of the row keep one cadence each)
  double buf_r[kPer], buf_d[kPer];
  unsigned have = 0u;
#pragma unroll
  for (int q = 0; q < kPer; ++q) buf_r[q] = buf_d[q] = 0.0;
  // reverse of the step (i - 1) -> i :  F_i = P o (F_p + W_p z_p),  S_i = P P^T o (S_p + d_p W_p W_p^T);
  // on entry Sb, Fb are the adjoints of S_i, F_i; on exit those of S_{i-1}, F_{i-1}, and Wb, db, zb
  // those of W_{i-1}, d_{i-1}, z_{i-1}
  auto propagate_adjoint = [&](int64_t i, double d_p, double z_p, double W_p, double F_p, const double* S_p) {
    const double dt = t[i] - t[i - 1];
    double Pj;
    near_last = stp.step(k, dt, &Pj);
    double Pall[J], Wpall[J];
    EXO_GROUP_GATHER(Pj, Pall);
    EXO_GROUP_GATHER(W_p, Wpall);
    const double Gj = fma(W_p, z_p, F_p);
    double Pb = Fb * Gj;
    const double Gb = Fb * Pj;
    double Wb_prev = Gb * z_p;
    const double zb_prev = group_sum<G>(Gb * W_p);
    // (five operations per entry: q = Sb P_l serves the adjoint of T and the propagator's cotangent, and the two sums over
    // Tb W_l -- for Wb and for db -- are one sum; nine as first written, of the ~570 vector instructions a cadence issued at J = 10)
    double psum = 0.0, dsum = 0.0;
    const double dw = d_p * W_p;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      const double T = fma(dw, Wpall[l], S_p[l]);
      const double q = Sb[l] * Pall[l];
      const double Tb = q * Pj;
      psum = fma(q, T, psum);
      dsum = fma(Tb, Wpall[l], dsum);
      Sb[l] = Tb;
    }
    Pb = fma(2.0, psum, Pb);
    Wb_prev = fma(2.0 * d_p, dsum, Wb_prev);
    const double db_prev = group_sum<G>(dsum * W_p);
    gc = fma(-dt * Pj, Pb, gc);
    db = db_prev; zb = zb_prev; Fb = Gb; Wb = Wb_prev;
  };

  // d loglike / d(oscillation rate) as a phase FLUX (exo_celerite_core.hpp, phase_flux): the flux across the chunk's end boundary
  // from the state entering the next chunk (saved at cadence n1) and the adjoint the scan handed over; a pair's first lane keeps it,
  // the partner lane supplies the other row
  double flux = 0.0;
  if (n1 < n) {
    double d1, z1, W1, F1, S1[J];
    load_wf(n1, d1, z1, W1, F1);
    load_s(n1, S1);   // (the next chunk's first cadence: a checkpoint)
    const double Fb_o = __shfl(Fb, partner, 64), F1_o = __shfl(F1, partner, 64);
    double acc = Fb_o * F1 - Fb * F1_o;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      const double Sb_o = __shfl(Sb[l], partner, 64), S1_o = __shfl(S1[l], partner, 64);
      acc = fma(2.0, Sb_o * S1[l] - Sb[l] * S1_o, acc);
    }
    flux = (k.live && !k.real && !k.odd) ? acc : 0.0;
  }
  double cs = 1.0, sn = 0.0;
  // the measurement half of cadence i from its record (d, z, W, F, S row)
  auto measure = [&](int64_t i, double d_n, double z_n, double W_n, double F_n, const double* S_n) {
    // measurement half of cadence i
    const double ti = t[i];
    const double dt_next = (i + 1 < n) ? t[i + 1] - ti : 0.0;
    double Uj, Vj;
    if (i < n1 - 1 && near_last && ((i - n0) & 3) != 0) {   // (cos, sin) of cadence i from cadence i + 1's, one near step back
      stp.rot_back(k, dt_next, cs, sn);
      lane_uv_from(k, cs, sn, &Uj, &Vj);
    } else {
      lane_uv(k, ti, &Uj, &Vj, &cs, &sn);
    }
    double Uall[J];
    EXO_GROUP_GATHER(Uj, Uall);
    const double id = exo::fast_rcp(d_n);   // (seed + two Newton steps, as the one-lane kernels: a third of the IEEE sequence)
    const double zbar = zb - gL * z_n * id;
    const double wdot = group_sum<G>(Wb * W_n);
    const double dbar = db + gL * (0.5 * z_n * z_n * id * id - 0.5 * id) - wdot * id;
    // gresid / gdiag are [draw][cadence]: a store per cadence would touch one 8-B piece of a
    // different 64-B line for every draw of the wave.  The G lanes of a draw (zbar, dbar are the
    // same on all of them) each keep 8 / G consecutive cadences of an aligned block of 8 and the
    // block is written when it is complete (measured: 2.9x write amplification without this).
    if (gcm && live_draw && j == 0) grow.store(i, gsign * zbar);   // cadence-major: the wave's draws are neighbours (sparse: the few values)
    {
      const int owner = (int)(i & 7) / kPer, slot = (int)(i & 7) % kPer;
      if (live_draw && j == owner) {
#pragma unroll
        for (int q = 0; q < kPer; ++q)
          if (slot == q) { buf_r[q] = zbar; buf_d[q] = dbar; }
        have |= 1u << slot;
      }
      if ((i & 7) == 0 || i == n0) {
        const int64_t at = draw * n + (i & ~(int64_t)7) + j * kPer;
#pragma unroll
        for (int q = 0; q < kPer; ++q)
          if ((have >> q) & 1u) {
            if (!gcm) gresid[at + q] = gsign * buf_r[q];
            if (gdiag) gdiag[at + q + (cf.at(draw) - draw) * n] = buf_d[q];
          }
        have = 0u;
      }
    }
    gasum += dbar;
    double Ub = -zbar * F_n;
    Fb = fma(-zbar, Uj, Fb);
    const double Vb = Wb * id;
    double uj = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) uj = fma(S_n[l], Uall[l], uj);
    Ub = fma(-dbar, uj, Ub);
    const double ubj = -Vb - dbar * Uj;
    double uball[J];
    EXO_GROUP_GATHER(ubj, uball);
    double acc_u = 0.0;
    const double ubh = 0.5 * ubj, Uh = 0.5 * Uj;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      Sb[l] = fma(ubh, Uall[l], fma(uball[l], Uh, Sb[l]));   // symmetrised  ub U^T
      acc_u = fma(S_n[l], uball[l], acc_u);
    }
    Ub += acc_u;
    if (k.live && k.real) ga += Ub;
    {
      const double Ub_o = __shfl(Ub, partner, 64), Vb_o = __shfl(Vb, partner, 64);
      if (k.live && !k.real && !k.odd) {
        ga += Ub * cs + Ub_o * sn;
        gb += Ub * sn - Ub_o * cs;
        // the link behind this cadence carries the flux so far; this cadence's phase cotangent leaves it
        gd = fma(-dt_next, flux, gd);
        flux -= Ub * (-k.a * sn + k.b * cs) + Ub_o * (k.a * cs + k.b * sn) - Vb * sn + Vb_o * cs;
      }
    }
  };
  // Blocks of lg_span cadences, last to first.  A block's records are loaded at once (all in flight together), its S rows
  // recomputed forward from the checkpoint at its first cadence -- S_i = P_i P_i^T o (S_(i-1) + d_(i-1) W_(i-1) W_(i-1)^T), the
  // forward kernel's own line -- and its cadences walked backwards: the reverse of the step OUT of cadence i, then the
  // measurement half of cadence i, both from cadence i's record alone.
  constexpr int K = lg_span<J>();
  const int64_t nblk = (n1 - n0 + K - 1) / K;
#pragma unroll 1
  for (int64_t bb = nblk - 1; bb >= 0; --bb) {
    const int64_t b0 = n0 + bb * K;
    const int len = (int)((n1 - b0 < K) ? n1 - b0 : K);   // (block-uniform: c is blockIdx.y)
    double rd[K], rz[K], rW[K], rF[K], Sblk[K][J];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      if (q < len) load_wf(b0 + q, rd[q], rz[q], rW[q], rF[q]);
      else { rd[q] = 1.0; rz[q] = rW[q] = rF[q] = 0.0; }
    }
    load_s(b0, Sblk[0]);
#pragma unroll
    for (int q = 1; q < K; ++q) {
      if (q < len) {
        double Pj;
        stf.step(k, t[b0 + q] - t[b0 + q - 1], &Pj);
        double Pall[J], Wall[J];
        EXO_GROUP_GATHER(Pj, Pall);
        EXO_GROUP_GATHER(rW[q - 1], Wall);
        const double dwj = rd[q - 1] * rW[q - 1];
#pragma unroll
        for (int l = 0; l < J; ++l) Sblk[q][l] = Pj * Pall[l] * fma(dwj, Wall[l], Sblk[q - 1][l]);
      } else {
#pragma unroll
        for (int l = 0; l < J; ++l) Sblk[q][l] = 0.0;
      }
    }
#pragma unroll
    for (int q = K - 1; q >= 0; --q) {
      if (q < len) {
        const int64_t i = b0 + q;
        if (i + 1 < n) propagate_adjoint(i + 1, rd[q], rz[q], rW[q], rF[q], Sblk[q]);   // reverse of the step i -> i + 1
        measure(i, rd[q], rz[q], rW[q], rF[q], Sblk[q]);
      }
    }
  }
  if (live_draw && k.live) {
    state[ws.gpart(c, 4 * j + 0, draw)] = ga;
    state[ws.gpart(c, 4 * j + 1, draw)] = gb;
    state[ws.gpart(c, 4 * j + 2, draw)] = gc;
    state[ws.gpart(c, 4 * j + 3, draw)] = gd;
    if (j == 0) state[ws.gpart(c, 4 * J, draw)] = gasum;
  }
}

// ---------------------------------------------------------------------------------------------
// The scans over the chunks as trees of element compositions (exo_celerite_core.hpp, "The scans (B), (B') as TREES").
// Everything runs one lane per item (celerite_tree_kernel below) except composing two FILTERING elements at J >= 3:
//     M = I + C1 J2,  X1 = M^-1 A1,  x2 = M^-1 (b1 + C1 eta2),  X3 = M^-1 C1,  N = I - J2 X3
//     A = A2 X1        b = A2 x2 + b2        C = A2 X3 A2^T + C2
//     eta = A1^T N (eta2 - J2 b1) + eta1     J = A1^T N J2 A1 + J1
// keeps ~5 J x J matrices alive around the solve -- one lane spills 2 KB at J = 6 and takes 76 us per level whatever
// its size.  Here: one wave per composition (a block), the matrices in LDS padded to 8 x 8, lane (j, l) owning entry
// (j, l) of every product, Gauss-Jordan with partial pivoting for the solve.  LDS-bandwidth-bound (~230 reads of
// 512 B per composition), ~40 us per level at 128 chains x 512 chunks.  Also used where the J > 2 lane-group path
// builds its elements on half chunks and composes them pairwise (ChunkGeom::fine).
// ---------------------------------------------------------------------------------------------
struct ComposeLds {
  double m[13][64];   // A1 C1 J1 A2 C2 J2 M R1 R3 T T2 T3 tmp
  double v[8][8];     // b1 eta1 b2 eta2 r2 v w tmp
};
__global__ __launch_bounds__(kWave) void celerite_compose_lds_kernel(TreeOp op, double* __restrict__ state) {
  __shared__ ComposeLds S;
  const int lane = threadIdx.x, j = lane >> 3, l = lane & 7;
  const int64_t item = blockIdx.x;                       // (index, draw), draws fastest
  const int J = op.J;
  const int64_t n_draw = op.n_draw;
  const int c = (int)(item / n_draw);
  const int64_t draw = item - (int64_t)c * n_draw;
  enum { A1 = 0, C1, J1, A2, C2, J2, MM, R1, R3, TT, T2, T3, TMP };
  enum { B1 = 0, E1, B2, E2, RR2, VV, WW, VT };
  const int E = 3 * J * J + 2 * J;
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J, oJ = 2 * J * J + 2 * J;
  const bool in = j < J && l < J;
  auto src = [&](int pos, int e) -> double {
    const int idx = op.src_rev ? op.src_len - 1 - pos : pos;
    return state[op.src_elem + ((int64_t)idx * E + e) * n_draw + draw];
  };
  // ---- load the two elements; missing ones are the identity
  const int p1 = 2 * c, p2 = 2 * c + 1;
  const bool has1 = p1 < op.src_n, has2 = p2 < op.src_n;
  const double idm = (j == l && in) ? 1.0 : 0.0;
  S.m[A1][lane] = in ? (has1 ? src(p1, oA + j * J + l) : idm) : 0.0;
  S.m[C1][lane] = (in && has1) ? src(p1, oC + j * J + l) : 0.0;
  S.m[J1][lane] = (in && has1) ? src(p1, oJ + j * J + l) : 0.0;
  S.m[A2][lane] = in ? (has2 ? src(p2, oA + j * J + l) : idm) : 0.0;
  S.m[C2][lane] = (in && has2) ? src(p2, oC + j * J + l) : 0.0;
  S.m[J2][lane] = (in && has2) ? src(p2, oJ + j * J + l) : 0.0;
  if (l == 0) {
    S.v[B1][j] = (j < J && has1) ? src(p1, ob + j) : 0.0;
    S.v[E1][j] = (j < J && has1) ? src(p1, oeta + j) : 0.0;
    S.v[B2][j] = (j < J && has2) ? src(p2, ob + j) : 0.0;
    S.v[E2][j] = (j < J && has2) ? src(p2, oeta + j) : 0.0;
  }
  __syncthreads();
  // dst[j][l] = sum_k X[j][k] Y[k][l]  (transposes by index)
  auto mm = [&](int X, bool tx, int Y, bool ty) -> double {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fma(tx ? S.m[X][k * 8 + j] : S.m[X][j * 8 + k], ty ? S.m[Y][l * 8 + k] : S.m[Y][k * 8 + l], acc);
    return acc;
  };
  auto mv = [&](int X, bool tx, int V) -> double {   // (X v)_j, computed by every lane of row j
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fma(tx ? S.m[X][k * 8 + j] : S.m[X][j * 8 + k], S.v[V][k], acc);
    return acc;
  };
  // M = I + C1 J2 ;  right-hand sides R1 = A1, r2 = b1 + C1 eta2, R3 = C1
  {
    const double m = mm(C1, false, J2, false) + (j == l ? 1.0 : 0.0);
    const double r2 = S.v[B1][j] + mv(C1, false, E2);
    S.m[MM][lane] = m;
    S.m[R1][lane] = S.m[A1][lane];
    S.m[R3][lane] = S.m[C1][lane];
    if (l == 0) S.v[RR2][j] = r2;
  }
  __syncthreads();
  // Gauss-Jordan with partial pivoting (padded rows / columns are the identity: never chosen, never changed)
  for (int k = 0; k < J; ++k) {
    int piv = k;
    double best = fabs(S.m[MM][k * 8 + k]);
    for (int i = k + 1; i < J; ++i) {
      const double a = fabs(S.m[MM][i * 8 + k]);
      if (a > best) { best = a; piv = i; }
    }
    // rows k and piv swapped on the way in; row k scaled, its multiples taken off the others
    const int srow = j == k ? piv : (j == piv ? k : j);
    const double ip = 1.0 / S.m[MM][piv * 8 + k];
    const double mk = S.m[MM][piv * 8 + l] * ip, ak = S.m[R1][piv * 8 + l] * ip, ck = S.m[R3][piv * 8 + l] * ip;
    const double rk = S.v[RR2][piv] * ip;
    const double f = j == k ? 0.0 : S.m[MM][srow * 8 + k];
    const double mj = S.m[MM][srow * 8 + l], aj = S.m[R1][srow * 8 + l], cj = S.m[R3][srow * 8 + l], rj = S.v[RR2][srow];
    __syncthreads();
    S.m[MM][lane] = j == k ? mk : fma(-f, mk, mj);
    S.m[R1][lane] = j == k ? ak : fma(-f, ak, aj);
    S.m[R3][lane] = j == k ? ck : fma(-f, ck, cj);
    if (l == 0) S.v[RR2][j] = j == k ? rk : fma(-f, rk, rj);
    __syncthreads();
  }
  // now R1 = X1, r2 = x2, R3 = X3
  const double A_new = mm(A2, false, R1, false);
  const double b_new = mv(A2, false, RR2) + S.v[B2][j];
  S.m[TT][lane] = mm(A2, false, R3, false);                          // A2 X3
  S.m[T2][lane] = (j == l ? 1.0 : 0.0) - mm(J2, false, R3, false);   // N = I - J2 X3
  if (l == 0) S.v[VV][j] = S.v[E2][j] - mv(J2, false, B1);           // eta2 - J2 b1
  __syncthreads();
  const double C_new = mm(TT, false, A2, true) + S.m[C2][lane];
  S.m[T3][lane] = mm(T2, false, J2, false);                          // N J2
  if (l == 0) S.v[WW][j] = mv(T2, false, VV);                        // N (eta2 - J2 b1)
  __syncthreads();
  S.m[TMP][lane] = mm(T3, false, A1, false);                         // N J2 A1
  const double eta_new = mv(A1, true, WW) + S.v[E1][j];
  __syncthreads();
  const double J_new = mm(A1, true, TMP, false) + S.m[J1][lane];
  // symmetrise C and J through LDS
  S.m[TT][lane] = C_new;
  S.m[T2][lane] = J_new;
  __syncthreads();
  const double C_sym = 0.5 * (C_new + S.m[TT][l * 8 + j]), J_sym = 0.5 * (J_new + S.m[T2][l * 8 + j]);
  if (!in) return;
  auto dst = [&](int e) -> double& { return state[op.dst_elem + ((int64_t)c * E + e) * n_draw + draw]; };
  dst(oA + j * J + l) = A_new;
  dst(oC + j * J + l) = C_sym;
  dst(oJ + j * J + l) = J_sym;
  if (l == 0) { dst(ob + j) = b_new; dst(oeta + j) = eta_new; }
}

// The same items, ONE LANE each (tree_item_lane, exo_celerite_core.hpp): (index, draw) pairs with draws fastest, so that
// the [index][quantity][draw] arrays are read and written coalesced.  An item touches ~3 J^2 doubles once; whatever
// spills costs a few loads, and a wave carries 64 items where the LDS kernel above carries one.
template <int J, bool ADJ, bool DOWN>
__global__ __launch_bounds__(kWave) void celerite_tree_kernel(TreeOp op, double* __restrict__ state) {
  const int64_t item = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (item >= (int64_t)op.n_item * op.n_draw) return;
  const int c = (int)(item / op.n_draw);
  tree_item_lane<J, ADJ, DOWN>(op, state, c, item - (int64_t)c * op.n_draw);
}

// WIDE STATES (J = 9 .. 16, round 6): an item of a scan level on a BLOCK of 256 threads, thread (j, l) owning entry (j, l) of every
// J x J matrix, the matrices in LDS padded to 16 x 16.  One lane per item (celerite_tree_kernel) holds an element's 3 J^2 + 2 J
// doubles -- 320 at J = 10, 800 at J = 16 -- in scratch and walks J^3 products through it: 0.4-1.4 ms per level at 128 chains x 128
// chunks, 21 of the 33 ms of a C5-shaped step at J = 10.  Same algebra as tree_compose / tree_apply (exo_celerite_core.hpp), the
// sums in the same order (k ascending); the solve is Gauss-Jordan with partial pivoting on [X | right-hand sides] in place (the
// one-lane code eliminates forward and substitutes back: the same pivots, other roundings at the 1e-16 level).  J is a RUN-TIME
// value here (op.J): one instantiation per (ADJ, DOWN) serves every width.
#ifndef EXO_GP_WIDE_LDS
#define EXO_GP_WIDE_LDS 1     // (0: the one-lane tree kernel for wide states too -- A/B)
#endif
struct WideLds {
  double m[14][256];
  double v[10][16];
  int piv;
};
template <bool ADJ, bool DOWN>
__global__ __launch_bounds__(256) void celerite_tree_wide_kernel(TreeOp op, double* __restrict__ state) {
  __shared__ WideLds S;
  const int tid = threadIdx.x, j = tid >> 4, l = tid & 15;
  const int J = op.J;
  const int64_t nd = op.n_draw;
  const int64_t item = blockIdx.x;                      // (index, draw), draws fastest
  const int c = (int)(item / nd);
  const int64_t draw = item - (int64_t)c * nd;
  const bool in = j < J && l < J;
  const int E = 3 * J * J + 2 * J, Bq = J + J * J;
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J, oJ = 2 * J * J + 2 * J;
  auto src = [&](int pos, int e) -> double {
    const int idx = op.src_rev ? op.src_len - 1 - pos : pos;
    return state[op.src_elem + ((int64_t)idx * E + e) * nd + draw];
  };
  // element `pos` into matrices (a, cm, jm) and vectors (b, eta); the identity past the end
  auto load_elem = [&](int pos, int a, int cm, int jm, int b, int eta) {
    const bool has = pos >= 0 && pos < op.src_n;
    S.m[a][tid] = in ? (has ? src(pos, oA + j * J + l) : (j == l ? 1.0 : 0.0)) : 0.0;
    S.m[cm][tid] = (in && has) ? src(pos, oC + j * J + l) : 0.0;
    S.m[jm][tid] = (in && has) ? src(pos, oJ + j * J + l) : 0.0;
    if (l == 0) {
      S.v[b][j] = (j < J && has) ? src(pos, ob + j) : 0.0;
      S.v[eta][j] = (j < J && has) ? src(pos, oeta + j) : 0.0;
    }
  };
  // sum_k op(X)[j][k] op(Y)[k][l], k ascending from `init`
  auto mm = [&](double init, int X, bool tx, int Y, bool ty) -> double {
    double acc = init;
    for (int k = 0; k < J; ++k)
      acc = fma(tx ? S.m[X][k * 16 + j] : S.m[X][j * 16 + k], ty ? S.m[Y][l * 16 + k] : S.m[Y][k * 16 + l], acc);
    return acc;
  };
  // (op(X) v)_j from `init`, computed by every thread of row j
  auto mv = [&](double init, int X, bool tx, int V) -> double {
    double acc = init;
    for (int k = 0; k < J; ++k) acc = fma(tx ? S.m[X][k * 16 + j] : S.m[X][j * 16 + k], S.v[V][k], acc);
    return acc;
  };
  // Gauss-Jordan with partial pivoting: X (matrix slot) against right-hand sides R1, R3 (matrix slots; R3 < 0: none) and r (vector
  // slot); on exit the right-hand sides hold the solutions.  Every thread of the block calls it.
  auto solve = [&](int X, int R1, int R3, int r) {
    for (int k = 0; k < J; ++k) {
      if (tid == 0) {
        int p = k;
        double best = fabs(S.m[X][k * 16 + k]);
        for (int i = k + 1; i < J; ++i) {
          const double a = fabs(S.m[X][i * 16 + k]);
          if (a > best) { best = a; p = i; }
        }
        S.piv = p;
      }
      __syncthreads();
      const int p = S.piv;
      if (p != k && j == k && l < J) {          // row k's sixteen threads swap rows k and p
        double tmp = S.m[X][k * 16 + l]; S.m[X][k * 16 + l] = S.m[X][p * 16 + l]; S.m[X][p * 16 + l] = tmp;
        tmp = S.m[R1][k * 16 + l]; S.m[R1][k * 16 + l] = S.m[R1][p * 16 + l]; S.m[R1][p * 16 + l] = tmp;
        if (R3 >= 0) { tmp = S.m[R3][k * 16 + l]; S.m[R3][k * 16 + l] = S.m[R3][p * 16 + l]; S.m[R3][p * 16 + l] = tmp; }
        if (l == 0) { tmp = S.v[r][k]; S.v[r][k] = S.v[r][p]; S.v[r][p] = tmp; }
      }
      __syncthreads();
      const double f = (in && j != k) ? S.m[X][j * 16 + k] / S.m[X][k * 16 + k] : 0.0;
      const double xk = in ? S.m[X][k * 16 + l] : 0.0, r1k = in ? S.m[R1][k * 16 + l] : 0.0;
      const double r3k = (in && R3 >= 0) ? S.m[R3][k * 16 + l] : 0.0, rk = S.v[r][k];
      __syncthreads();
      if (in && j != k) {
        S.m[X][tid] = fma(-f, xk, S.m[X][tid]);
        S.m[R1][tid] = fma(-f, r1k, S.m[R1][tid]);
        if (R3 >= 0) S.m[R3][tid] = fma(-f, r3k, S.m[R3][tid]);
        if (l == 0) S.v[r][j] = fma(-f, rk, S.v[r][j]);
      }
      __syncthreads();
    }
    if (in) {
      const double ip = 1.0 / S.m[X][j * 16 + j];
      S.m[R1][tid] *= ip;
      if (R3 >= 0) S.m[R3][tid] *= ip;
      if (l == 0) S.v[r][j] *= ip;
    }
    __syncthreads();
  };
  enum { A1 = 0, C1, J1, A2, C2, J2, MM, R1, R3, T1, T2, T3, T4, T5 };
  enum { B1 = 0, E1, B2, E2, RR, VW, VN, VX };
  if (DOWN) {
    // ---- child states 2c, 2c + 1 from parent state c and child element 2c
    const double mj = (l == 0 && j < J) ? state[op.par_state + ((int64_t)c * Bq + j) * nd + draw] : 0.0;
    const double pjl = in ? state[op.par_state + ((int64_t)c * Bq + J + j * J + l) * nd + draw] : 0.0;
    auto put = [&](int pos, double mval, double pval) {
      const int idx = op.dst_rev ? op.dst_len - 1 - pos : pos;
      double* __restrict__ q = state + op.dst_state + ((int64_t)idx * Bq) * nd + draw;
      if (l == 0 && j < J) q[(int64_t)j * nd] = mval;
      if (in) q[(int64_t)(J + j * J + l) * nd] = op.psign * pval;
    };
    put(2 * c, mj, pjl);
    if (2 * c + 1 >= op.dst_n) return;               // (uniform over the block)
    load_elem(2 * c, A1, C1, J1, B1, E1);
    S.m[T1][tid] = pjl;                               // P
    if (l == 0) S.v[VW][j] = mj;                      // m
    __syncthreads();
    if (ADJ) {
      // x = A^T m ;  T = P A ;  m2 = eta + x ;  P2 = Cm + A^T T + sym(x b^T)
      const double xj = mv(0.0, A1, true, VW);
      const double tv = mm(0.0, T1, false, A1, false);
      if (l == 0) S.v[VX][j] = xj;
      S.m[T2][tid] = tv;
      __syncthreads();
      const double cong = mm(S.m[C1][tid], A1, true, T2, false);
      const double p2 = in ? cong + 0.5 * (S.v[VX][j] * S.v[B1][l] + S.v[B1][j] * S.v[VX][l]) : 0.0;
      S.m[T3][tid] = p2;
      __syncthreads();
      put(2 * c + 1, S.v[E1][j] + S.v[VX][j], 0.5 * (S.m[T3][j * 16 + l] + S.m[T3][l * 16 + j]));
    } else {
      // X = I + P Jm ;  solve X [YP | ym] = [P | m + P eta] ;  m2 = b + A ym ;  P2 = Cm + (A YP) A^T
      const double xv = mm((j == l && in) ? 1.0 : 0.0, T1, false, J1, false);
      const double pe = mv(S.v[VW][j], T1, false, E1);
      S.m[MM][tid] = in ? xv : 0.0;
      S.m[R1][tid] = pjl;
      if (l == 0) S.v[RR][j] = (j < J) ? pe : 0.0;
      __syncthreads();
      solve(MM, R1, -1, RR);
      const double m2 = mv(S.v[B1][j], A1, false, RR);
      const double ay = mm(0.0, A1, false, R1, false);
      S.m[T2][tid] = ay;
      __syncthreads();
      S.m[T3][tid] = mm(S.m[C1][tid], T2, false, A1, true);
      __syncthreads();
      put(2 * c + 1, m2, 0.5 * (S.m[T3][j * 16 + l] + S.m[T3][l * 16 + j]));
    }
    return;
  }
  // ---- UP: dst element c = src elements 2c, 2c + 1 composed (2c acts first)
  load_elem(2 * c, A1, C1, J1, B1, E1);
  load_elem(2 * c + 1, A2, C2, J2, B2, E2);
  __syncthreads();
  double* __restrict__ dstp = state + op.dst_elem + ((int64_t)c * E) * nd + draw;
  auto dst = [&](int e, double val) { dstp[(int64_t)e * nd] = val; };
  if (ADJ) {
    // Abar = Abar1 Abar2 ;  g = g2 + Abar2^T g1 ;  lF = lF2 + Abar2^T lF1 ;  lP = lP2 + Abar2^T lP1 Abar2 + sym(Abar2^T lF1 g2^T)
    const double gj = mv(S.v[B2][j], A2, true, B1);
    const double vj = mv(0.0, A2, true, E1);
    const double a = mm(0.0, A1, false, A2, false);
    S.m[T1][tid] = mm(0.0, C1, false, A2, false);        // T = lP1 Abar2
    if (l == 0) S.v[VX][j] = vj;
    __syncthreads();
    const double cv = mm(in ? S.m[C2][tid] + 0.5 * (S.v[VX][j] * S.v[B2][l] + S.v[B2][j] * S.v[VX][l]) : 0.0, A2, true, T1, false);
    S.m[T2][tid] = cv;
    __syncthreads();
    if (in) {
      dst(oA + j * J + l, a);
      dst(oC + j * J + l, 0.5 * (S.m[T2][j * 16 + l] + S.m[T2][l * 16 + j]));
      dst(oJ + j * J + l, 0.0);
    }
    if (l == 0 && j < J) { dst(ob + j, gj); dst(oeta + j, S.v[E2][j] + vj); }
    return;
  }
  // filtering elements:  M = I + C1 J2 ;  solve M [X1 | X3 | x2] = [A1 | C1 | b1 + C1 eta2]
  {
    const double mval = mm((j == l && in) ? 1.0 : 0.0, C1, false, J2, false);
    const double r2 = mv(S.v[B1][j], C1, false, E2);
    S.m[MM][tid] = in ? mval : 0.0;
    S.m[R1][tid] = S.m[A1][tid];
    S.m[R3][tid] = S.m[C1][tid];
    if (l == 0) S.v[RR][j] = (j < J) ? r2 : 0.0;
    __syncthreads();
  }
  solve(MM, R1, R3, RR);                               // R1 = X1, R3 = X3, RR = x2
  {
    const double bj = mv(S.v[B2][j], A2, false, RR);                   // b = b2 + A2 x2
    const double wj = -mv(-S.v[E2][j], J2, false, B1);                 // w = eta2 - J2 b1  (same order: eta2, then -J2 b1 terms)
    const double a = mm(0.0, A2, false, R1, false);                    // A = A2 X1
    const double ax = mm(0.0, A2, false, R3, false);                   // A2 X3
    const double nv = -mm((j == l && in) ? -1.0 : 0.0, J2, false, R3, false);   // N = I - J2 X3
    S.m[T1][tid] = ax;
    S.m[T2][tid] = in ? nv : 0.0;
    if (l == 0) { S.v[VW][j] = wj; S.v[VN][j] = bj; }
    __syncthreads();
    const double nw = mv(0.0, T2, false, VW);                           // N w
    S.m[T3][tid] = mm(0.0, T2, false, J2, false);                       // N J2
    if (l == 0) S.v[VX][j] = nw;
    __syncthreads();
    S.m[T4][tid] = mm(0.0, T3, false, A1, false);                       // N J2 A1
    __syncthreads();
    const double ej = mv(S.v[E1][j], A1, true, VX);                     // eta = eta1 + A1^T N w
    const double cv = mm(S.m[C2][tid], T1, false, A2, true);            // C = C2 + (A2 X3) A2^T
    const double jv = mm(S.m[J1][tid], A1, true, T4, false);            // J = J1 + A1^T (N J2 A1)
    S.m[T5][tid] = cv;
    S.m[MM][tid] = jv;
    __syncthreads();
    if (in) {
      dst(oA + j * J + l, a);
      dst(oC + j * J + l, 0.5 * (S.m[T5][j * 16 + l] + S.m[T5][l * 16 + j]));
      dst(oJ + j * J + l, 0.5 * (S.m[MM][j * 16 + l] + S.m[MM][l * 16 + j]));
    }
    if (l == 0 && j < J) { dst(ob + j, S.v[VN][j]); dst(oeta + j, ej); }
  }
}

// (B') part 1 for wide states: the adjoint element of chunk c from its filtering element and the state entering it (badj_prep_lane:
// X = I + P Jm, Y = X^-1, ...), a block of 256 threads per (draw, chunk) like celerite_tree_wide_kernel -- one lane per item walks
// 5 KB of scratch (0.86 of a 9.8 ms step at J = 10).  Same sums in the same order; the element is overwritten in place.
__global__ __launch_bounds__(256) void celerite_badj_prep_wide_kernel(const double* __restrict__ gloglike, int64_t n, int64_t n_draw,
                                                                      int J, double* __restrict__ state, ChunkGeom cg,
                                                                      const int32_t* __restrict__ row) {
  __shared__ WideLds S;
  const int tid = threadIdx.x, j = tid >> 4, l = tid & 15;
  const int64_t draw = blockIdx.x;
  const int c = (int)blockIdx.y + 1;
  const bool in = j < J && l < J;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const double gL = gloglike ? gloglike[row ? (int64_t)row[draw] : draw] : 1.0;   // (null: a cotangent of one -- ChunkGeom::prep)
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J, oJ = 2 * J * J + 2 * J;
  enum { MA = 0, MJ, MP, MX, MY, MT };
  enum { VM = 0, VETA, VU, VV, VW, VYV, VD };
  S.m[MA][tid] = in ? state[ws.elem(c, oA + j * J + l, draw)] : 0.0;
  S.m[MJ][tid] = in ? state[ws.elem(c, oJ + j * J + l, draw)] : 0.0;
  S.m[MP][tid] = in ? state[ws.bnd(1, c, J + j * J + l, draw)] : 0.0;
  S.m[MY][tid] = (in && j == l) ? 1.0 : 0.0;
  if (l == 0) {
    S.v[VM][j] = j < J ? state[ws.bnd(1, c, j, draw)] : 0.0;
    S.v[VETA][j] = j < J ? state[ws.elem(c, oeta + j, draw)] : 0.0;
    S.v[VD][j] = 0.0;
  }
  __syncthreads();
  auto mm = [&](double init, int X, bool tx, int Y, bool ty) -> double {
    double acc = init;
    for (int k = 0; k < J; ++k)
      acc = fma(tx ? S.m[X][k * 16 + j] : S.m[X][j * 16 + k], ty ? S.m[Y][l * 16 + k] : S.m[Y][k * 16 + l], acc);
    return acc;
  };
  auto mv = [&](double init, int X, bool tx, int V) -> double {
    double acc = init;
    for (int k = 0; k < J; ++k) acc = fma(tx ? S.m[X][k * 16 + j] : S.m[X][j * 16 + k], S.v[V][k], acc);
    return acc;
  };
  {
    const double xv = mm((j == l && in) ? 1.0 : 0.0, MP, false, MJ, false);      // X = I + P Jm
    const double uj = -mv(-S.v[VETA][j], MJ, false, VM);                        // u = eta - Jm m
    const double vj = mv(S.v[VM][j], MP, false, VETA);                          // v = m + P eta
    S.m[MX][tid] = in ? xv : 0.0;
    if (l == 0) { S.v[VU][j] = uj; S.v[VV][j] = vj; }
    __syncthreads();
  }
  // Y = X^-1: Gauss-Jordan with partial pivoting on [X | I] (celerite_tree_wide_kernel's solve, one right-hand side)
  for (int k = 0; k < J; ++k) {
    if (tid == 0) {
      int p = k;
      double best = fabs(S.m[MX][k * 16 + k]);
      for (int i = k + 1; i < J; ++i) {
        const double a = fabs(S.m[MX][i * 16 + k]);
        if (a > best) { best = a; p = i; }
      }
      S.piv = p;
    }
    __syncthreads();
    const int p = S.piv;
    if (p != k && j == k && l < J) {
      double tmp = S.m[MX][k * 16 + l]; S.m[MX][k * 16 + l] = S.m[MX][p * 16 + l]; S.m[MX][p * 16 + l] = tmp;
      tmp = S.m[MY][k * 16 + l]; S.m[MY][k * 16 + l] = S.m[MY][p * 16 + l]; S.m[MY][p * 16 + l] = tmp;
    }
    __syncthreads();
    const double f = (in && j != k) ? S.m[MX][j * 16 + k] / S.m[MX][k * 16 + k] : 0.0;
    const double xk = in ? S.m[MX][k * 16 + l] : 0.0, yk = in ? S.m[MY][k * 16 + l] : 0.0;
    __syncthreads();
    if (in && j != k) {
      S.m[MX][tid] = fma(-f, xk, S.m[MX][tid]);
      S.m[MY][tid] = fma(-f, yk, S.m[MY][tid]);
    }
    __syncthreads();
  }
  if (in) S.m[MY][tid] *= 1.0 / S.m[MX][j * 16 + j];
  __syncthreads();
  {
    const double wj = mv(0.0, MY, true, VU);                                    // w = Y^T u
    const double yv = mv(0.0, MY, false, VV);                                   // Y v
    const double a = mm(0.0, MA, false, MY, false);                             // A Y
    S.m[MT][tid] = mm(0.0, MJ, false, MY, false);                               // Jm Y
    if (l == 0) { S.v[VW][j] = wj; S.v[VYV][j] = yv; }
    __syncthreads();
    const double gj = -mv(-S.v[VETA][j], MJ, false, VYV);                       // g = eta - Jm (Y v)
    if (in) {
      state[ws.elem(c, oA + j * J + l, draw)] = a;
      state[ws.elem(c, oC + j * J + l, draw)] = 0.5 * gL * (S.v[VW][j] * S.v[VW][l] - 0.5 * (S.m[MT][j * 16 + l] + S.m[MT][l * 16 + j]));
    }
    if (l == 0 && j < J) {
      state[ws.elem(c, ob + j, draw)] = gj;
      state[ws.elem(c, oeta + j, draw)] = gL * S.v[VW][j];
    }
  }
}

// two levels of a scan in one launch, one lane per item of the upper one (tree_item4_*_lane; J <= 2)
#ifndef EXO_GP_TREE4
#define EXO_GP_TREE4 1
#endif
template <int J, bool ADJ, bool DOWN>
__global__ __launch_bounds__(kWave) void celerite_tree4_kernel(TreeOp a, TreeOp b, double* __restrict__ state) {
  const int64_t item = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (item >= (int64_t)b.n_item * b.n_draw) return;
  const int i = (int)(item / b.n_draw);
  const int64_t draw = item - (int64_t)i * b.n_draw;
  if (DOWN) tree_item4_down_lane<J, ADJ>(a, b, state, i, draw);
  else tree_item4_up_lane<J, ADJ>(a, b, state, i, draw);
}

// (A draw's scan in ONE launch -- a block per draw walking the levels with block barriers, all of them or the narrow top ones
// only -- was built in round 4, measured slower inside a replayed graph (C3 3.88 against 3.82 ms, C5 1.90 against 1.91: what a
// narrow level costs is its item's dependent latency, not its launch) and removed in round 5.)
// (Round 5, also measured and not kept: the narrow top as a SERIAL chain -- from the level with <= 128 positions up, one lane per
// draw applying that level's elements one after the other to the seed, elements prefetched four ahead, no compositions above
// it: a step costs ~0.5 us -- the 2 x 2 solve's divisions and a load latency the prefetch only partly hides -- so 128 positions
// are no cheaper than the 14 launches they replace: C3 3.63 ms against 3.57 (64 positions: 3.59, 256: 3.74).)
#ifndef EXO_GP_GROUP_TREES
#define EXO_GP_GROUP_TREES 1
#endif
#ifndef EXO_GP_FINE_GROUP
#define EXO_GP_FINE_GROUP 1   // the pairwise composition of the fine elements (ChunkGeom::fine; J = 7, 8) by celerite_tree_group_kernel too
#endif
constexpr int kScanBlock = 256;
// One level of a scan, items of J >= 3 on groups of eight lanes (tree_item_group): a block is 32 groups = 32 CONSECUTIVE
// DRAWS of one item (draws fastest), so that each of an item's ~100 strided accesses covers, per row, 64 contiguous bytes of
// eight draws (one lane per (item, draw) -- celerite_tree_kernel -- is fully coalesced but spills 2 KB per lane at J = 6; a
// block per draw -- the fused kernel below -- touches 64 lines per instruction and is SLOWER than the 34 launches it replaces).
#ifndef EXO_GROUP_WAVES
#define EXO_GROUP_WAVES 1
#endif
template <int J, bool ADJ, bool DOWN>
__global__ __launch_bounds__(kScanBlock, EXO_GROUP_WAVES) void celerite_tree_group_kernel(TreeOp op, double* state) {
  __shared__ double lds[(kScanBlock / 8) * GroupLds<J>::S];
  const int tid = threadIdx.x;
  const int64_t unit = (int64_t)blockIdx.x * (kScanBlock / 8) + (tid >> 3);
  if (unit >= (int64_t)op.n_item * op.n_draw) return;     // (whole groups leave; nothing below needs a block barrier)
  Grp<J> g;
  g.lds = lds + (tid >> 3) * GroupLds<J>::S;
  g.r = tid & 7;
  g.live = g.r < J;
  const int c = (int)(unit / op.n_draw);
  tree_item_group<J, ADJ, DOWN>(op, state, c, unit - (int64_t)c * op.n_draw, g);
}

// the adjoint elements (B') part 1 on groups of eight lanes, 32 draws per block (badj_prep_group; J = 7, 8)
// From J = 7: one lane per (draw, chunk) holds 512 registers + 1.8 KB of scratch there (137 us at the C5 shape, J = 8: this kernel
// ~35); at J = 6 the one-lane kernel takes all 65 408 units in one round of 40 us and THIS one, four rounds of two waves per SIMD,
// was slower (C5 1.895 -> 1.948 ms).
#ifndef EXO_GP_BADJ_GROUP_MIN_J
#define EXO_GP_BADJ_GROUP_MIN_J 7
#endif
template <int J>
__global__ __launch_bounds__(kScanBlock, EXO_GROUP_WAVES) void celerite_badj_prep_group_kernel(const double* __restrict__ gloglike, int64_t n,
                                                                                             int64_t n_draw, double* state, ChunkGeom cg,
                                                                                             const int32_t* __restrict__ row) {
  __shared__ double lds[(kScanBlock / 8) * GroupLds<J>::S];
  const int tid = threadIdx.x;
  const int64_t unit = (int64_t)blockIdx.x * (kScanBlock / 8) + (tid >> 3);
  if (unit >= (int64_t)(cg.C - 1) * n_draw) return;     // (whole groups leave; nothing below needs a block barrier)
  Grp<J> g;
  g.lds = lds + (tid >> 3) * GroupLds<J>::S;
  g.r = tid & 7;
  g.live = g.r < J;
  const int c = (int)(unit / n_draw) + 1;
  badj_prep_group<J>(gloglike, state, chunk_ws(n, n_draw, J, cg), c, unit - (int64_t)(c - 1) * n_draw, g, row);
}

// the narrow top of a scan as a serial chain on groups of eight lanes, 32 draws per block (tree_serial_group; J >= 3)
template <int J, bool ADJ>
__global__ __launch_bounds__(kScanBlock, EXO_GROUP_WAVES) void celerite_tree_serial_group_kernel(TreeOp op, double* state) {
  __shared__ double lds[(kScanBlock / 8) * GroupLds<J>::S];
  const int tid = threadIdx.x;
  const int64_t draw = (int64_t)blockIdx.x * (kScanBlock / 8) + (tid >> 3);
  if (draw >= op.n_draw) return;     // (whole groups leave; nothing below needs a block barrier)
  Grp<J> g;
  g.lds = lds + (tid >> 3) * GroupLds<J>::S;
  g.r = tid & 7;
  g.live = g.r < J;
  tree_serial_group<J, ADJ>(op, state, draw, g);
}


// the state the forward scan starts from: F = 0, P = Delta(t_0) (S_0 = 0)
template <int J>
__global__ __launch_bounds__(kWave) void celerite_scan_init_kernel(const double* __restrict__ t, Coefs cf, int64_t n_draw,
                                                                   double* __restrict__ dst) {
  constexpr int G = Group<J>::G;
  const int j = threadIdx.x & (G - 1);
  const int64_t draw = ((int64_t)blockIdx.x * kWave + threadIdx.x) / G;
  if (draw >= n_draw) return;
  if (J <= kLaneMaxJ) {
    // the one-lane path's own Delta (DeltaCoef: joint covariance of an over-damped SHO's pair of real terms included);
    // LaneDelta below serves the lane-group kernels (J = 7, 8), where such pairs stay flagged
    if (j == 0) scan_init_lane<J>(t, cf, n_draw, dst, draw);
    return;
  }
  const LaneCoef k = lane_coef(cf, draw, j, J);
  if (!k.live) return;
  const LaneDelta ld(k);
  double U_, V_, cs, sn, Prow[J];
  lane_uv(k, t[0], &U_, &V_, &cs, &sn);
  ld.row<J>(k, j, cs, sn, Prow);
  dst[(int64_t)j * n_draw + draw] = 0.0;
#pragma unroll
  for (int l = 0; l < J; ++l) dst[(int64_t)(J + j * J + l) * n_draw + draw] = Prow[l];
}

// ---------------------------------------------------------------------------------------------
// One lane per draw / per (draw, chunk): thin wrappers over exo_celerite_core.hpp (the same
// functions the host harness of tests/ runs lane by lane).  Lanes of a wave are consecutive draws,
// the chunk is blockIdx.y: every access to the chunk workspace and to the checkpoints is coalesced.
// ---------------------------------------------------------------------------------------------
// Term layouts (exo_celerite_core.hpp, DeltaCoef): NR >= 0 -- the first NR state indices are real
// terms, the rest complex pairs -- is a compile-time constant of the kernel; NR = -1 decides per index
// at run time.  Every variant is a kernel of its own (its own register budget); a wave votes on the
// layout of its draws (layout_vote) and returns at once from the variants it did not vote for, so the
// host may launch several variants of one step when pair kinds are given per draw.
// J = 4 (two SHO terms, a RotationTerm): the all-complex layout as kernels OF ITS OWN (launched beside the run-time-layout ones;
// a wave runs the one it voted for).  Inside one kernel -- as J = 6, 8 have it -- the registers are those of the heavier
// run-time layout: forward 204 against 162 (two waves per SIMD against three), and element / reverse sit 14 / 24 registers above
// the 256 of TWO waves per SIMD, which they are held to here (EXO_J4_WAVES): a 1024-draw batch is 4096 waves -- four rounds of
// resident waves at one per SIMD, two at two.
#ifndef EXO_J4_SPLIT
#define EXO_J4_SPLIT 1
#endif
#ifndef EXO_J4_WAVES
#define EXO_J4_WAVES 2
#endif
constexpr bool split_layouts(int J) { return EXO_J4_SPLIT && J == 4; }
// (J <= 2: four waves per SIMD without the look-ahead load of the next block -- elem_lane's PREFETCH)
#ifndef EXO_ELEM_MIXED_WAVES
#define EXO_ELEM_MIXED_WAVES 4
#endif
// which chunk a block of the one-lane kernels works on.  Dense series: every wave costs the same, blocks in order.  SPARSE
// model: a wave is slow exactly while one of its 64 draws is inside a transit (+ 21 % per block: measured with every block
// forced onto either path), so the chunks that hold a transit are the long ones.  They are dealt FIRST (longest processing
// time first) and the short ones fill the tail of the launch: celerite_sparse_order_kernel writes the order.
// EXO_SPARSE_LPT=0: always in order.
#ifndef EXO_SPARSE_LPT
#define EXO_SPARSE_LPT 1
#endif
template <int SP>
__device__ __forceinline__ int chunk_of_block(const double* __restrict__ state, int64_t n, int64_t n_draw, int J, const ChunkGeom& cg) {
  const int y = (int)blockIdx.y;
  if (SP != 1 || !EXO_SPARSE_LPT) return y;
  return reinterpret_cast<const int32_t*>(state + chunk_ws(n, n_draw, J, cg).off_order())[y];
}
// One block: a chunk is LONG if a segment of one of a few sample draws (spread over the batch: with the draws sorted by
// transit time -- exo_sparse_model.row_of_draw -- the first, the last and those between span the transit times of all) reaches
// into it, widened by a block either side; long chunks first, each class in order of time (a stable partition).
constexpr int kOrderSamples = 8, kOrderSegs = 512;
__global__ __launch_bounds__(256) void celerite_sparse_order_kernel(SparseSegs sp, int64_t n, int64_t n_draw, ChunkGeom cg,
                                                                    int32_t* __restrict__ order) {
  __shared__ int s_long[1024 + 1];
  // the sample draws' segments, staged once: the searches below are dependent loads -- from global memory 8 x ~6 of them per
  // chunk made this kernel 43 us of the C3 step; from LDS ~8 (a draw with more than kOrderSegs segments is searched in place)
  __shared__ int s_lo[kOrderSamples][kOrderSegs], s_hi[kOrderSamples][kOrderSegs], s_n[kOrderSamples];
  {   // (32 threads per sample draw: the eight chains row -> count -> segments run side by side)
    static_assert(kOrderSamples * 32 == 256, "a group of 32 threads per sample");
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int64_t d = n_draw <= kOrderSamples ? (q < n_draw ? q : n_draw - 1) : (n_draw - 1) * q / (kOrderSamples - 1);
    const int64_t row = sp.row(d);
    const int32_t* sg = sp.seg + row * sp.seg_row;
    const int ns = sp.nseg[row];
    if (l == 0) s_n[q] = ns;
    if (ns <= kOrderSegs)
      for (int i = l; i < ns; i += 32) {
        s_lo[q][i] = sg[(int64_t)i * sp.seg_step];
        s_hi[q][i] = sg[(int64_t)i * sp.seg_step + sp.hi_at];
      }
  }
  __syncthreads();
  const int C = cg.C;
  for (int c0 = 0; c0 < C; c0 += 1024) {     // (plans hold at most 1024 chunks; written for any number)
    const int m = C - c0 < 1024 ? C - c0 : 1024;
    for (int i = threadIdx.x; i < m; i += 256) {
      const int c = c0 + i;
      const int64_t lo = (int64_t)c * cg.L - kCkptB, hi = ((int64_t)(c + 1) * cg.L < n ? (int64_t)(c + 1) * cg.L : n) + kCkptB;
      bool lng = false;
      for (int q = 0; q < kOrderSamples && !lng; ++q) {
        const int ns = s_n[q];
        int a = 0, b = ns;                      // the first segment that ends beyond lo
        if (ns <= kOrderSegs) {
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (s_hi[q][mid] <= lo) a = mid + 1; else b = mid;
          }
          lng = a < ns && s_lo[q][a] < hi;
        } else {
          const int64_t d = n_draw <= kOrderSamples ? (q < n_draw ? q : n_draw - 1) : (n_draw - 1) * q / (kOrderSamples - 1);
          const int32_t* sg = sp.seg + sp.row(d) * sp.seg_row;
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (sg[(int64_t)mid * sp.seg_step + sp.hi_at] <= lo) a = mid + 1; else b = mid;
          }
          lng = a < ns && sg[(int64_t)a * sp.seg_step] < hi;
        }
      }
      s_long[i] = lng ? 1 : 0;
    }
    __syncthreads();
    {   // stable partition, long chunks first: every thread four consecutive flags, an exclusive scan of the counts over the block
      __shared__ int s_cnt[256];
      const int i0 = threadIdx.x * 4;
      int mine = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) mine += (i0 + u < m) ? s_long[i0 + u] : 0;
      s_cnt[threadIdx.x] = mine;
      __syncthreads();
      for (int off = 1; off < 256; off <<= 1) {
        const int add = threadIdx.x >= off ? s_cnt[threadIdx.x - off] : 0;
        __syncthreads();
        s_cnt[threadIdx.x] += add;
        __syncthreads();
      }
      const int n_long = s_cnt[255];
      int at_long = c0 + s_cnt[threadIdx.x] - mine;            // long chunks before this thread's
      int at_short = c0 + n_long + (i0 < m ? i0 : m) - (s_cnt[threadIdx.x] - mine);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u < m) {
          if (s_long[i0 + u]) order[at_long++] = c0 + i0 + u; else order[at_short++] = c0 + i0 + u;
        }
    }
    __syncthreads();
  }
}
// the order of the draws of a sparse model by the spacing of their segments (exo_sparse_model_order)
constexpr int kOrderMaxDraws = EXO_SPARSE_ORDER_MAX_DRAWS;
__global__ __launch_bounds__(1024) void celerite_draw_order_kernel(SparseSegs sp, int64_t n_draw, int32_t* __restrict__ order) {
  __shared__ double s_key[kOrderMaxDraws];
  __shared__ int s_idx[kOrderMaxDraws];
  int M = 1;
  while (M < n_draw) M <<= 1;
  for (int i = threadIdx.x; i < M; i += 1024) {
    double key = INFINITY;
    if (i < n_draw) {
      const int32_t* sg = sp.seg + (int64_t)i * sp.seg_row;
      const int ns = sp.nseg[i];
      const double first = ns > 0 ? (double)sg[0] : 0.0;
      const double last = ns > 0 ? (double)sg[(int64_t)(ns - 1) * sp.seg_step] : 0.0;
      key = (last - first) / (double)(ns > 1 ? ns - 1 : 1) + 1e-9 * first;
    }
    s_key[i] = key;
    s_idx[i] = i;
  }
  __syncthreads();
  // (ranking every key against all others -- no barriers -- was tried for small batches: 45 us at 1024 draws against 23: the block
  // is one CU, 5 instructions x n_draw per key)
  if (M <= 1024) {
    // one element per thread, in registers: the steps inside a wave (partner = lane ^ j, j < 64: 45 of the 55 steps at 1024 draws)
    // are shuffles -- no LDS, no barrier -- and only the ten across waves go through LDS (23 -> 12 us)
    const int i = threadIdx.x;
    double key = i < M ? s_key[i] : INFINITY;
    int idx = i < M ? s_idx[i] : i;
    for (int k = 2; k <= M; k <<= 1)
      for (int j = k >> 1; j >= 1; j >>= 1) {
        double kp;
        int ip;
        if (j >= 64) {
          __syncthreads();
          if (i < M) { s_key[i] = key; s_idx[i] = idx; }
          __syncthreads();
          kp = i < M ? s_key[i ^ j] : INFINITY;
          ip = i < M ? s_idx[i ^ j] : i;
        } else {
          kp = __shfl_xor(key, j, 64);
          ip = __shfl_xor(idx, j, 64);
        }
        const bool lower = (i & j) == 0;                         // I am the lower index of the pair
        const bool mine_after = key > kp || (key == kp && idx > ip);
        const bool asc = (i & k) == 0;
        // the pair in ascending order puts the smaller at the lower index; descending the other way round
        const bool take = lower ? (mine_after == asc) : (mine_after != asc);
        if (take) { key = kp; idx = ip; }
      }
    if (i < n_draw) order[i] = idx;
    return;
  }
  for (int k = 2; k <= M; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      for (int i = threadIdx.x; i < M; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const double ka = s_key[i], kb = s_key[p];
          const int ia = s_idx[i], ib = s_idx[p];
          const bool a_after_b = ka > kb || (ka == kb && ia > ib);
          if (a_after_b == ((i & k) == 0)) { s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib; s_idx[p] = ia; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < n_draw; i += 1024) order[i] = s_idx[i];
}
// (A) the filtering element of every (draw, chunk)
template <int J, int NR, int SP>
__global__ __launch_bounds__(kWave, (J <= 2 ? EXO_ELEM_MIXED_WAVES : (split_layouts(J) && NR == 0 ? EXO_J4_WAVES : 1))) void celerite_elem_kernel(const double* __restrict__ t, Series rs,
                                                              const double* __restrict__ diag, int64_t n_diag, int64_t n,
                                                              Coefs cf, int64_t n_draw, double* __restrict__ state,
                                                              ChunkGeom cg, int64_t flag_at) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  const int vote = layout_vote<J>(cf, draw);
  if constexpr (J > 2 && NR == -1 && EXO_GP_WIDE_COMPLEX_LAYOUT && !split_layouts(J)) {   // a wave of all-complex draws takes the compile-time layout
    if (vote == 0) { elem_lane<J, 0, true, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, chunk_of_block<SP>(state, n, n_draw, J, cg), flag_at); return; }
  } else if (vote != NR) return;
  elem_lane<J, NR, (J > 2 || EXO_ELEM_MIXED_WAVES < 4), SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, chunk_of_block<SP>(state, n, n_draw, J, cg), flag_at);
}

// (B), (B'): the scans over the chunks are trees of compositions (celerite_tree_kernel, celerite_compose_lds_kernel above).
// (B') part 1: the adjoint elements, all chunks in parallel (chunk 0's entering adjoint is never needed)
template <int J>
__global__ __launch_bounds__(kWave) void celerite_badj_prep_kernel(const double* __restrict__ gloglike, int64_t n,
                                                                   int64_t n_draw, double* __restrict__ state,
                                                                   ChunkGeom cg, const int32_t* __restrict__ row) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  badj_prep_lane<J>(gloglike, n, n_draw, state, cg, draw, (int)blockIdx.y + 1, row);
}

// (C) / (C') with a checkpointed factorisation, J <= kLaneMaxJ
template <int J, int NR, int SP>
__global__ __launch_bounds__(kWave, (split_layouts(J) && NR == 0 ? 3 : 1)) void celerite_chunk1_fwd_kernel(const double* __restrict__ t, Series rs,
                                                                    const double* __restrict__ diag, int64_t n_diag,
                                                                    int64_t n, Coefs cf, int64_t n_draw,
                                                                    double* __restrict__ state, ChunkGeom cg) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  const int vote = layout_vote<J>(cf, draw);
  if constexpr (J > 2 && NR == -1 && EXO_GP_WIDE_COMPLEX_LAYOUT && !split_layouts(J)) {
    if (vote == 0) { chunk1_fwd_lane<J, 0, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, chunk_of_block<SP>(state, n, n_draw, J, cg), true); return; }
  } else if (vote != NR) return;
  chunk1_fwd_lane<J, NR, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, chunk_of_block<SP>(state, n, n_draw, J, cg), true);
}
// (two waves per SIMD asked for: the J = 2 complex-term variant sits at 254 + 4 registers otherwise -- one wave)
#ifndef EXO_VJP1_WAVES
#define EXO_VJP1_WAVES 2
#endif
#ifndef EXO_VJPP_WAVES
#define EXO_VJPP_WAVES 2
#endif
template <int J, int NR, int SP>
__global__ __launch_bounds__(kWave, (J < EXO_SPAN2_MIN_J ? EXO_VJP1_WAVES : (J <= 2 ? EXO_VJPP_WAVES : (split_layouts(J) && NR == 0 ? EXO_J4_WAVES : 1)))) void celerite_chunk1_vjp_kernel(const double* __restrict__ t, Series rs,
                                                                    const double* __restrict__ diag, int64_t n_diag,
                                                                    int64_t n, Coefs cf, int64_t n_draw,
                                                                    const double* __restrict__ gloglike,
                                                                    double* __restrict__ state, ChunkGeom cg,
                                                                    double* __restrict__ gresid, double* __restrict__ gdiag,
                                                                    double gsign) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  const int vote = layout_vote<J>(cf, draw);
  constexpr bool kBoth = J > 2 && NR == -1 && EXO_GP_WIDE_COMPLEX_LAYOUT && !split_layouts(J);   // (both layouts in this kernel)
  if (!kBoth && vote != NR) return;
  if constexpr (J >= EXO_SPAN2_MIN_J) {   // wide states: packed adjoint, two checkpoints per block, cotangent accumulators in LDS columns
    __shared__ double gacc[4 * J + 1][kWave];
    if constexpr (kBoth) {
      if (vote == 0) {
        chunkp_vjp_lane<J, 0, SP>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, draw, chunk_of_block<SP>(state, n, n_draw, J, cg),
                              &gacc[0][threadIdx.x], kWave);
        return;
      }
    }
    chunkp_vjp_lane<J, NR, SP>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, draw, chunk_of_block<SP>(state, n, n_draw, J, cg),
                           &gacc[0][threadIdx.x], kWave);
  } else
    chunk1_vjp_lane<J, NR, SP>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, draw, chunk_of_block<SP>(state, n, n_draw, J, cg));
}

// ---- the ROBUST route (draws flagged kFlagRobust: exo_celerite_core.hpp, chunk_adj_lane) ---------------------------------
// (B) once more for those draws alone, as a serial chain over the chunks: a draw on one group of eight lanes for J >= 3
// (robust_fwd_chain_group), on one lane below.  Launched after the trees, whose boundary states of these draws it replaces.
template <int J>
__global__ __launch_bounds__(kWave) void celerite_robust_scan_kernel(const double* __restrict__ t, Coefs cf, int64_t n,
                                                                     ChunkGeom cg, int64_t n_draw, double* state) {
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  if constexpr (J >= 3) {
    __shared__ double lds[(kWave / 8) * ChainLds<J>::S];
    const int64_t draw = (int64_t)blockIdx.x * (kWave / 8) + (threadIdx.x >> 3);
    const bool mine = draw < n_draw && state[ws.off_flag() + (draw < n_draw ? draw : 0)] == kFlagRobust;
    if (__ballot(mine) == 0) return;
    if (!mine) return;     // (whole groups leave: nothing below needs a block barrier)
    robust_fwd_chain_group<J>(ws, state, draw, lds + (threadIdx.x >> 3) * ChainLds<J>::S, (int)(threadIdx.x & 7));
  } else {
    const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (draw >= n_draw || state[ws.off_flag() + draw] != kFlagRobust) return;
    bscan_lane<J>(t, cf, n, n_draw, state, cg, draw);
  }
}

// (B) once more for those draws as NEWTON ITERATIONS from the trees' states (exo_celerite_group.hpp, tan_linearise): one block per
// draw -- it leaves at once unless the draw is flagged kFlagRobust -- walks, per iteration, the level-0 items (tangent elements
// of the chunks at the current guess, composed pairwise), the levels of the corrections' scan up and down, and the level-0
// items again (corrections added), a block barrier between levels.  J >= 3; replaces the serial chain above (kept behind
// EXO_GP_ROBUST_NEWTON = 0: 1.34 ms at C5 against ~0.5).
#ifndef EXO_GP_ROBUST_NEWTON
#define EXO_GP_ROBUST_NEWTON 1
#endif
// Iterations: until one's corrections were below EXO_GP_NEWTON_TOL of the states (the next guess is then right to the square
// of that), at most EXO_GP_NEWTON_ITERS (tools/gp_lab_newton.py, 450 chunks, scores of 1e7 .. 1e8: the guess is off by up to
// 4e-2, one iteration leaves 4e-7, two 8e-9 -- the serial chain's own distance from the long-double definition)
#ifndef EXO_GP_NEWTON_ITERS
#define EXO_GP_NEWTON_ITERS 4
#endif
#ifndef EXO_GP_NEWTON_TOL
#define EXO_GP_NEWTON_TOL 1e-8
#endif
#ifndef EXO_GP_NEWTON_BLOCK
#define EXO_GP_NEWTON_BLOCK 256   // (512: 64 items at a time, but 256 registers a lane -- scratch at J = 6 -- and 0.54 ms against 0.43)
#endif
constexpr int kNewtonBlock = EXO_GP_NEWTON_BLOCK;   // threads of a draw's block: 8 per item of a level
template <int J>
__global__ __launch_bounds__(kNewtonBlock) void celerite_robust_newton_kernel(int64_t n, ChunkGeom cg, int64_t n_draw, double* state) {
  __shared__ double lds[(kNewtonBlock / 8) * GroupLds<J>::S];
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const int64_t draw = blockIdx.x;
  if (state[ws.off_flag() + draw] != kFlagRobust) return;     // (the whole block)
  const int tid = threadIdx.x, unit = tid >> 3, n_unit = kNewtonBlock / 8;
  Grp<J> g;
  g.lds = lds + unit * GroupLds<J>::S;
  g.r = tid & 7;
  g.live = g.r < J;
  const int top = ws.tree_top(), n1 = ws.tree_npos(1);
  __shared__ double s_err[kNewtonBlock / kWave];
  bool converged = false;
  for (int it = 0; it < EXO_GP_NEWTON_ITERS; ++it) {
    for (int i = unit; i < n1; i += n_unit) newton_up0<J>(ws, state, i, draw, g);
    __syncthreads();
    for (int f = 1; f + 1 < top; ++f) {
      const TreeOp op = scan_level_op(ws, J, false, f, false);
      for (int c = unit; c < op.n_item; c += n_unit) newton_item<J, false>(op, state, c, draw, g);
      __syncthreads();
    }
    {   // nothing to correct in front of the first chunk
      const int64_t seed = ws.tree_state(top);
      for (int k = tid; k < J + J * J; k += kNewtonBlock) state[seed + (int64_t)k * n_draw + draw] = 0.0;
    }
    __syncthreads();
    for (int f = top - 1; f >= 1; --f) {
      const TreeOp op = scan_level_op(ws, J, false, f, true);
      for (int c = unit; c < op.n_item; c += n_unit) newton_item<J, true>(op, state, c, draw, g);
      __syncthreads();
    }
    double err = 0.0;
    for (int i = unit; i < n1; i += n_unit) err = fmax(err, newton_down0<J>(ws, state, i, draw, g));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) err = fmax(err, __shfl_xor(err, m, 64));
    if ((tid & 63) == 0) s_err[tid >> 6] = err;
    __syncthreads();
    double all = 0.0;
#pragma unroll
    for (int w = 0; w < kNewtonBlock / kWave; ++w) all = fmax(all, s_err[w]);
    __syncthreads();
    if (all < EXO_GP_NEWTON_TOL) { converged = true; break; }     // (the same verdict in every thread)
  }
  // not there after the last iteration, or a NaN on the way (a guess the iterations could not start from): the elements applied
  // one after the other -- the serial chain needs no guess -- by the block's first eight lanes
  if (!converged && tid < 8) {
    static_assert(ChainLds<J>::S <= (kNewtonBlock / 8) * GroupLds<J>::S, "the chain's strip inside the block's LDS");
    robust_fwd_chain_group<J>(ws, state, draw, lds, tid);
  }
}

// (B') part 1 for those draws: the adjoint scan's inputs from the chunks' own reverse recurrences (chunk_adj_lane), written
// over the chunk's element where badj_prep_lane wrote its own -- launched between that kernel and the adjoint trees, which take
// them as they are.  A WAVE per (draw, chunk >= 1): the chunk's reverse sweep in eight pieces, a piece on a group of eight
// lanes (chunk_adj_lane's roles: the state adjoints, the J columns of X, R), the pieces' records multiplied back together
// through LDS (adj_combine_lane).  Nothing but latency -- 1.2 ms for a 127-cadence chunk on one lane at J = 6, 0.7 ms on eight
// lanes by roles, ~0.2 ms in eight pieces.  A block looks at its draws' flags and, almost always, leaves at once.
constexpr int kAdjDraws = 32;   // (8: 14 us of empty blocks for a clean batch of the C3 shape; 64: a batch of nothing but such draws 64 deep)
template <int J>
__global__ __launch_bounds__(kWave) void celerite_chunk_adj_kernel(const double* __restrict__ t, Series rs,
                                                                   const double* __restrict__ diag, int64_t n_diag, int64_t n,
                                                                   Coefs cf, int64_t n_draw, const double* __restrict__ gloglike,
                                                                   double* __restrict__ state, ChunkGeom cg) {
  static_assert(J + 2 <= 8, "roles 0 .. J + 1 on the eight lanes of a group");
  constexpr int kRec = adj_record_doubles<J>(), kPieces = kWave / 8;
  __shared__ double rec[kPieces * kRec];
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  // (a block looks at kAdjDraws consecutive draws: a batch of nothing but such draws works them kAdjDraws deep, a clean batch
  // pays (C - 1) n_draw / kAdjDraws empty blocks -- ~5 us at the C3 shape)
  const int64_t d0 = (int64_t)blockIdx.y * kAdjDraws, dl = d0 + threadIdx.x;
  const bool look = threadIdx.x < kAdjDraws && dl < n_draw;
  unsigned long long todo = __ballot(look && state[ws.off_flag() + (look ? dl : 0)] == kFlagRobust);
  const int c = 1 + (int)blockIdx.x, piece = (int)(threadIdx.x >> 3), role = (int)(threadIdx.x & 7);
  while (todo) {
    const int64_t draw = d0 + (__ffsll((long long)todo) - 1);
    todo &= todo - 1;
    // (every lane has the same draw: the "vote" is that draw's layout -- the compile-time ones cost a quarter fewer instructions)
    if (role <= J + 1)
      with_layout<J>(cf, draw, [&](auto nr) {
        chunk_adj_lane<J, decltype(nr)::value, false>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, draw, c, role, piece,
                                                      kPieces, rec + piece * kRec, 1);
      });
    asm volatile("" ::: "memory");   // (one wave: its LDS operations execute in order; the compiler must keep them so)
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) adj_combine_lane<J>(rec, kRec, 1, kPieces, n, n_draw, state, cg, draw, c);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// sum over the wave in a fixed order (xor butterfly), result in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Sums over the chunks of [chunk][quantity][draw] arrays (the log-likelihood's partials, the coefficient cotangents'), the DRAWS
// ACROSS THE LANES: a row of 64 draws is 512 contiguous bytes.  (Rounds 2-4: a wave per (draw, quantity) with the chunks across its
// lanes -- every lane its own 64-B sector, 8 useful bytes of it: 52 + 20 us of the C3 step at a tenth of the bandwidth.)  Slice
// blockIdx.z of S adds the rows of chunks s, s + S, s + 2 S ... in order and leaves its partial sum in the slot of chunk s -- the
// first row it read; nobody else touches that row -- and the reader adds the S partials in order (slice_total).  S grows until
// the launch has ~2048 waves (chunk_sum_slices): a pure function of the plan, so a draw's sums do not depend on its neighbours.
__global__ __launch_bounds__(kWave) void celerite_chunk_slice_sum_kernel(double* __restrict__ arr, int K, int C, int S,
                                                                         int64_t n_draw) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  const int k = blockIdx.y, s0 = blockIdx.z;
  const int64_t step = (int64_t)S * K * n_draw;
  double* __restrict__ p = arr + ((int64_t)s0 * K + k) * n_draw + draw;
  const double* __restrict__ q = p;
  double v = 0.0;
  int c = s0;
  for (; c + 7 * S < C; c += 8 * S) {     // eight loads in flight, added in order
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = q[(int64_t)u * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += x[u];
    q += 8 * step;
  }
  for (; c < C; c += S) { v += *q; q += step; }
  *p = v;
}

// O(N) companions (exo_celerite_core.hpp): one lane per draw
template <int J>
__global__ __launch_bounds__(kWave) void celerite_dot_tril_kernel(const double* __restrict__ t,
                                                                  const double* __restrict__ diag, int64_t n_diag, int64_t n,
                                                                  Coefs cf, int64_t n_draw, const double* __restrict__ x,
                                                                  double* __restrict__ z) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  dot_tril_lane<J>(t, diag, n_diag, n, cf, x, z, draw);
}
template <int J>
__global__ __launch_bounds__(kWave) void celerite_predict_kernel(const double* __restrict__ t, int64_t n,
                                                                 const double* __restrict__ alpha, Coefs cf, int64_t n_draw,
                                                                 const double* __restrict__ tq, int64_t m,
                                                                 double* __restrict__ mu) {
  const int64_t draw = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (draw >= n_draw) return;
  predict_lane<J>(t, n, alpha, cf, tq, m, mu, draw);
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH; }

inline bool gp_args_ok(int64_t n, int64_t n_diag, int32_t n_real, int32_t n_complex, int64_t n_draw, int32_t n_chunks) {
  const int J = n_real + 2 * n_complex;
  return n >= 1 && n_draw >= 1 && n_real >= 0 && n_complex >= 0 && J >= 1 && J <= EXO_GP_MAX_J &&
         (n_diag == 1 || n_diag == n_draw) && n_chunks >= 0;
}

}  // namespace

extern "C" {

int64_t exo_celerite_state_doubles(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t n_chunks) {
  const int64_t J = n_real + 2 * (int64_t)n_complex;
  if (n_chunks >= 0) n_chunks &= ~EXO_GP_PREPARE_ADJOINT;   // (the flag changes nothing in the workspace)
  if (n < 0 || n_draw < 0 || J < 1 || J > EXO_GP_MAX_J || n_chunks < 0) return -1;
  const int64_t base = seq_state_doubles(n, n_draw, (int)J);
  if (n == 0 || n_draw == 0) return base;
  const ChunkGeom cg = chunk_plan(n, n_draw, (int)J, n_chunks);
  if (cg.C <= 1) return base;
  return base + chunk_ws(n, n_draw, (int)J, cg).total();
}

// The number of chunks the plan takes by default -- for a caller who wants to pass it explicitly (the same value to
// exo_celerite_state_doubles and to both calls of a pair).  sparse != 0: the plan the SPARSE entries do best with -- twice as
// many chunks (of at least 32 cadences) as a dense series gets: the waves of a sparse launch are uneven (slow exactly while a
// draw is inside a transit), and with two rounds of them the short ones fill the tail (C3: 3.68 -> 3.59 ms).  1: sequential.
int32_t exo_celerite_default_chunks(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t sparse) {
  const int64_t J = n_real + 2 * (int64_t)n_complex;
  if (n < 0 || n_draw < 1 || J < 1 || J > EXO_GP_MAX_J) return 1;
  const ChunkGeom cg = chunk_plan(n, n_draw, (int)J, 0);
  if (cg.C <= 1 || !sparse || !cg.lane || J > 2) return cg.C;
  int64_t C = 2 * (int64_t)cg.C;
  if (C > n / 32) C = n / 32;
  return chunk_plan(n, n_draw, (int)J, (int32_t)(C < 2 ? 2 : C)).C;
}

#define EXO_GP_DISPATCH_VOID(J_, CALL) \
  switch (J_) {                        \
    case 1: { constexpr int JJ = 1; CALL; } break; \
    case 2: { constexpr int JJ = 2; CALL; } break; \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    case 7: { constexpr int JJ = 7; CALL; } break; \
    case 8: { constexpr int JJ = 8; CALL; } break; \
    case 9: { constexpr int JJ = 9; CALL; } break; \
    case 10: { constexpr int JJ = 10; CALL; } break; \
    case 11: { constexpr int JJ = 11; CALL; } break; \
    case 12: { constexpr int JJ = 12; CALL; } break; \
    case 13: { constexpr int JJ = 13; CALL; } break; \
    case 14: { constexpr int JJ = 14; CALL; } break; \
    case 15: { constexpr int JJ = 15; CALL; } break; \
    case 16: { constexpr int JJ = 16; CALL; } break; \
    default: break;                                \
  }
// the ONE-LANE tree kernel: state widths 1 .. 8; 9 .. 16 only when the LDS kernel for wide states is switched off (EXO_GP_WIDE_LDS = 0)
#if EXO_GP_WIDE_LDS
#define EXO_GP_DISPATCH_TREE(J_, CALL) \
  switch (J_) {                        \
    case 1: { constexpr int JJ = 1; CALL; } break; \
    case 2: { constexpr int JJ = 2; CALL; } break; \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    case 7: { constexpr int JJ = 7; CALL; } break; \
    case 8: { constexpr int JJ = 8; CALL; } break; \
    default: break;                                \
  }
#else
#define EXO_GP_DISPATCH_TREE(J_, CALL) EXO_GP_DISPATCH_VOID(J_, CALL)
#endif
#define EXO_GP_DISPATCH_GROUP(J_, CALL) \
  switch (J_) {                         \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    case 7: { constexpr int JJ = 7; CALL; } break; \
    case 8: { constexpr int JJ = 8; CALL; } break; \
    default: break;                                \
  }
#define EXO_GP_DISPATCH_LANE(J_, CALL) \
  switch (J_) {                        \
    case 1: { constexpr int JJ = 1; CALL; } break; \
    case 2: { constexpr int JJ = 2; CALL; } break; \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    default: break;                                \
  }
static_assert(kLaneMaxJ <= 6, "EXO_GP_DISPATCH_LANE lists the state widths of the one-lane path");
#define EXO_GP_DISPATCH_NEWTON(J_, CALL) \
  switch (J_) {                          \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    default: break;                                \
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
    case 9: { constexpr int JJ = 9; CALL; } break; \
    case 10: { constexpr int JJ = 10; CALL; } break; \
    case 11: { constexpr int JJ = 11; CALL; } break; \
    case 12: { constexpr int JJ = 12; CALL; } break; \
    case 13: { constexpr int JJ = 13; CALL; } break; \
    case 14: { constexpr int JJ = 14; CALL; } break; \
    case 15: { constexpr int JJ = 15; CALL; } break; \
    case 16: { constexpr int JJ = 16; CALL; } break; \
    default: return EXO_ERR_INVALID_ARGUMENT;      \
  }
// kernels whose wide instantiations exist only when the LDS kernels for wide states are switched off (EXO_GP_WIDE_LDS = 0)
#if EXO_GP_WIDE_LDS
#define EXO_GP_DISPATCH_LE8(J_, CALL) \
  switch (J_) {                       \
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
#else
#define EXO_GP_DISPATCH_LE8(J_, CALL) EXO_GP_DISPATCH(J_, CALL)
#endif
// the SEQUENTIAL kernels (and the O(N) utilities) take state widths up to EXO_GP_MAX_J = 16 (a draw on a DPP row of 16 lanes
// above 8): celerite2, the reference's dependency (setup.py:36), has no limit, and two RotationTerms + an SHO term -- J = 10 --
// is an ordinary stellar-variability model.  The time-parallel path stops at 8 (kChunkMaxJ): wider states run the recurrences
// cadence by cadence, correct and slow.
#define EXO_GP_DISPATCH_SEQ(J_, CALL) \
  switch (J_) {                       \
    case 1: { constexpr int JJ = 1; CALL; } break; \
    case 2: { constexpr int JJ = 2; CALL; } break; \
    case 3: { constexpr int JJ = 3; CALL; } break; \
    case 4: { constexpr int JJ = 4; CALL; } break; \
    case 5: { constexpr int JJ = 5; CALL; } break; \
    case 6: { constexpr int JJ = 6; CALL; } break; \
    case 7: { constexpr int JJ = 7; CALL; } break; \
    case 8: { constexpr int JJ = 8; CALL; } break; \
    case 9: { constexpr int JJ = 9; CALL; } break; \
    case 10: { constexpr int JJ = 10; CALL; } break; \
    case 11: { constexpr int JJ = 11; CALL; } break; \
    case 12: { constexpr int JJ = 12; CALL; } break; \
    case 13: { constexpr int JJ = 13; CALL; } break; \
    case 14: { constexpr int JJ = 14; CALL; } break; \
    case 15: { constexpr int JJ = 15; CALL; } break; \
    case 16: { constexpr int JJ = 16; CALL; } break; \
    default: return EXO_ERR_INVALID_ARGUMENT;      \
  }
static_assert(kLaneMaxJ >= 2, "EXO_GP_LAYOUTS lists compile-time layouts for J <= 2; J > 2 takes run-time flags");
// CALL with `JJ` and `NR` for every layout variant the draws of a call may need (layout_vote):
// J = 1: one real term; J = 2: two real terms or one pair slot -- complex, or, with per-draw kinds,
// either (three launches: waves return at once from the variants they did not vote for); J > 2:
// run-time flags.
// J = 2 with per-draw pair kinds: which draw a lane of the *_mixed_kernel launches works on.  A wave runs ONE layout, the one
// all its draws share, or the run-time layout (half again as many instructions) -- and the step waits for its slowest wave: a
// batch with 1 % of its draws on the other side of Q = 1/2 cost 1.84 x a clean one for the sake of ONE mixed wave.  So the
// lanes take the draws in an order that has no mixed wave: the complex-term draws, padded to whole waves, then the
// two-real-terms draws (a stable partition: neighbours stay neighbours, the [..][draw] arrays stay coalesced), written into the
// workspace by one small kernel per forward call (the reverse call finds it there).
__device__ __forceinline__ int64_t mixed_draw(const double* __restrict__ state, int64_t n, int64_t n_draw, const ChunkGeom& cg) {
  const ChunkWs ws = chunk_ws(n, n_draw, 2, cg);
  const int32_t* __restrict__ perm = reinterpret_cast<const int32_t*>(state + ws.off_perm());
  return perm[(int64_t)blockIdx.x * kWave + threadIdx.x];      // (the launches cover ws.perm_lanes() lanes)
}
__global__ __launch_bounds__(1024) void celerite_kind_partition_kernel(const int32_t* __restrict__ kind, const int32_t* __restrict__ row,
                                                                       int64_t n_draw, int32_t* __restrict__ perm, int64_t lanes) {
  __shared__ int s_cnt[2][16];
  __shared__ int s_base[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int64_t i = tid; i < lanes; i += 1024) perm[i] = -1;
  // the complex-term draws: how many
  int mine = 0;
  for (int64_t d = tid; d < n_draw; d += 1024) mine += kind[row ? row[d] : d] == 0 ? 1 : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
  if (lane == 0) s_cnt[0][wave] = mine;
  if (tid < 2) s_base[tid] = 0;
  __syncthreads();
  int c0 = 0;
  for (int w = 0; w < 16; ++w) c0 += s_cnt[0][w];
  const int64_t off1 = ((int64_t)c0 + 63) / 64 * 64;
  __syncthreads();
  for (int64_t d0 = 0; d0 < n_draw; d0 += 1024) {
    const int64_t d = d0 + tid;
    const int k = d < n_draw ? (kind[row ? row[d] : d] == 0 ? 0 : 1) : -1;
    const unsigned long long m0 = __ballot(k == 0), m1 = __ballot(k == 1), below = (1ull << lane) - 1ull;
    if (lane == 0) { s_cnt[0][wave] = __popcll(m0); s_cnt[1][wave] = __popcll(m1); }
    __syncthreads();
    if (k >= 0) {
      int pos = s_base[k] + __popcll((k == 0 ? m0 : m1) & below);
      for (int w = 0; w < wave; ++w) pos += s_cnt[k][w];
      perm[(k == 0 ? 0 : off1) + pos] = (int32_t)d;
    }
    __syncthreads();
    if (tid < 2) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_cnt[tid][w];
      s_base[tid] += tot;
    }
    __syncthreads();
  }
}

// J = 2 with per-draw pair kinds (a batch of SHO terms that straddles Q = 1/2): the three layout variants in ONE launch.
// blockIdx.z picks the variant; a wave runs the one it voted for and leaves the other two at once.  Launched one after the
// other (round 2), each variant cost its full single-wave latency however few waves took it: a batch with 1 % of its draws
// on the other side of Q = 1/2 paid twice the clean batch's time.  The layouts keep their own code (compile-time NR); the
// kernel's registers are those of the widest variant -- the same waves per SIMD as each alone (145 / 118 / 159, 113 / 100 /
// 124, 217 / 186 / 239 registers for element / forward / reverse).
extern "C++" {   // (templates: the entry points below are extern "C")
template <int SP>
__global__ __launch_bounds__(kWave, EXO_ELEM_MIXED_WAVES) void celerite_elem_mixed_kernel(const double* __restrict__ t, Series rs,
                                                                    const double* __restrict__ diag, int64_t n_diag, int64_t n,
                                                                    Coefs cf, int64_t n_draw, double* __restrict__ state,
                                                                    ChunkGeom cg, int64_t flag_at) {
  const int64_t draw = mixed_draw(state, n, n_draw, cg);
  if (draw < 0) return;
  const int nr = layout_vote<2>(cf, draw);
  const int c = chunk_of_block<SP>(state, n, n_draw, 2, cg);
  if (blockIdx.z == 0) {
    if (nr == 0) elem_lane<2, 0, EXO_ELEM_MIXED_WAVES < 4, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, c, flag_at);
  } else {
    if (nr == 2) elem_lane<2, 2, EXO_ELEM_MIXED_WAVES < 4, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, c, flag_at);
  }   // (no wave is of mixed kinds: mixed_draw)
}
// (four waves per SIMD asked for: the three inlined layouts sit at 129 registers otherwise, and the plan offers four)
#ifndef EXO_FWD_MIXED_WAVES
#define EXO_FWD_MIXED_WAVES 4
#endif
template <int SP>
__global__ __launch_bounds__(kWave, EXO_FWD_MIXED_WAVES) void celerite_chunk1_fwd_mixed_kernel(const double* __restrict__ t, Series rs,
                                                                          const double* __restrict__ diag, int64_t n_diag,
                                                                          int64_t n, Coefs cf, int64_t n_draw,
                                                                          double* __restrict__ state, ChunkGeom cg) {
  const int64_t draw = mixed_draw(state, n, n_draw, cg);
  if (draw < 0) return;
  const int nr = layout_vote<2>(cf, draw);
  const int c = chunk_of_block<SP>(state, n, n_draw, 2, cg);
  if (blockIdx.z == 0) {
    if (nr == 0) chunk1_fwd_lane<2, 0, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, c, true);
  } else {
    if (nr == 2) chunk1_fwd_lane<2, 2, SP>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, draw, c, true);
  }   // (no wave is of mixed kinds: mixed_draw)
}
template <int SP>
__global__ __launch_bounds__(kWave, EXO_VJP1_WAVES) void celerite_chunk1_vjp_mixed_kernel(
    const double* __restrict__ t, Series rs, const double* __restrict__ diag, int64_t n_diag, int64_t n, Coefs cf, int64_t n_draw,
    const double* __restrict__ gloglike, double* __restrict__ state, ChunkGeom cg, double* __restrict__ gresid,
    double* __restrict__ gdiag, double gsign) {
  static_assert(2 < EXO_SPAN2_MIN_J, "the mixed reverse kernel is the J = 2 register-resident form (chunk1_vjp_lane)");
  const int64_t draw = mixed_draw(state, n, n_draw, cg);
  if (draw < 0) return;
  const int nr = layout_vote<2>(cf, draw);
  const int c = chunk_of_block<SP>(state, n, n_draw, 2, cg);
  if (blockIdx.z == 0) {
    if (nr == 0) chunk1_vjp_lane<2, 0, SP>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, draw, c);
  } else {
    if (nr == 2) chunk1_vjp_lane<2, 2, SP>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, draw, c);
  }   // (no wave is of mixed kinds: mixed_draw)
}

}  // extern "C++"

#ifndef EXO_GP_MIXED_ONE_LAUNCH
#define EXO_GP_MIXED_ONE_LAUNCH 1
#endif
#define EXO_GP_LAYOUTS(J_, CF, CALL, MIXED)                                             \
  switch (J_) {                                                                         \
    case 1: { constexpr int JJ = 1, NR = 1; CALL; } break;                              \
    case 2:                                                                             \
      if ((CF).n_real == 2) { constexpr int JJ = 2, NR = 2; CALL; }                     \
      else if (!(CF).kind) { constexpr int JJ = 2, NR = 0; CALL; }                      \
      else if (EXO_GP_MIXED_ONE_LAUNCH) { MIXED; }   /* per-draw kinds: the three variants in one launch */ \
      else {            /* ... or one after the other: a wave returns at once from the variants it did not vote for */ \
        { constexpr int JJ = 2, NR = 0; CALL; }                                         \
        { constexpr int JJ = 2, NR = 2; CALL; }                                         \
        { constexpr int JJ = 2, NR = -1; CALL; }                                        \
      }                                                                                 \
      break;                                                                            \
    case 3: { constexpr int JJ = 3, NR = -1; CALL; } break;                             \
    case 4:                                                                             \
      if (split_layouts(4) && (CF).n_real == 0) { constexpr int JJ = 4, NR = 0; CALL; }   \
      { constexpr int JJ = 4, NR = -1; CALL; }                                          \
      break;                                                                            \
    case 5: { constexpr int JJ = 5, NR = -1; CALL; } break;                             \
    case 6: { constexpr int JJ = 6, NR = -1; CALL; } break;                             \
    case 7: { constexpr int JJ = 7, NR = -1; CALL; } break;                             \
    case 8: { constexpr int JJ = 8, NR = -1; CALL; } break;                             \
    default: return EXO_ERR_INVALID_ARGUMENT;                                           \
  }

// the series' kind is a compile-time constant of the one-lane kernels (SeriesRowT): SP = 1 a sparse model, 0 otherwise
#define EXO_GP_BY_SERIES(RS, NAME)                                                                        \
  {                                                                                                       \
    const int rc_ = (RS).sp.nseg ? NAME(std::integral_constant<int, 1>{}) : NAME(std::integral_constant<int, 0>{}); \
    if (rc_ != EXO_OK) return rc_;                                                                        \
  }

// ---- the adjoint scan beside the forward chunk kernel (EXO_GP_PREPARE_ADJOINT, include/exoplanet_amd.h) -------------------------
// What the reverse call does before its chunk kernel -- adjoint elements (badj_prep), the robust route's own (chunk_adj), the
// scan (B') as a tree: 14-26 short launches, each an item's dependent latency, 0.13-0.17 ms at the C5 shape -- needs nothing of
// the forward chunk kernel: the elements and the entering states (B) are there before it starts, and the adjoint is LINEAR in
// the cotangent of the log-likelihood.  With the flag the forward call works it out for a cotangent of ONE, the scan on a second
// stream while the chunk kernel (one wave per SIMD at J >= 4, bound by its checkpoint stores) runs on the caller's, and the
// reverse chunk kernels scale the adjoint states by the draw's cotangent as they load them (ChunkGeom::prep; a cotangent of
// exactly 1 -- loglike.sum().backward() -- gives the bits of the serial order).  What goes where is decided by registers: a kernel
// that needs more than the forward chunk kernel's waves leave free on a SIMD (badj_prep 482, chunk_adj 512 at J = 6 against 280
// taken of 512) does not run beside it, it waits for it -- and everything queued behind it on its stream with it; those stay on the
// caller's stream, in front of the fork.  chunk_adj reads the forward checkpoints of the draws on the robust route, so those
// draws' recurrences run first and alone (ChunkGeom::which).  Inside a stream capture the second stream joins the capture
// through the two events, so a replayed graph carries the fork; eagerly they order the streams the same way.
#ifndef EXO_GP_SIDE_MAX_DEVICES
#define EXO_GP_SIDE_MAX_DEVICES 64
#endif
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream* side_stream() {
  // one per host thread and device: an event recorded by two threads at once would order the wrong streams
  thread_local SideStream tl[EXO_GP_SIDE_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EXO_GP_SIDE_MAX_DEVICES) return nullptr;
  SideStream& ss = tl[dev];
  if (!ss.s) {
    if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) != hipSuccess) { ss.s = nullptr; (void)hipGetLastError(); return nullptr; }
    if (hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  return (ss.fork && ss.join) ? &ss : nullptr;
}

// gloglike == nullptr: for a cotangent of one (the forward call's; ChunkGeom::prep)
// parts: 1 = the adjoint elements (badj_prep, chunk_adj), 2 = the scan over them, 3 = both
static int celerite_adjoint_scan(const double* t, Series resid, const double* diag, int64_t n_diag, int64_t n, Coefs cf,
                                 int64_t n_draw, const double* gloglike, double* wstate, const ChunkGeom& cg, hipStream_t st,
                                 int parts = 3) {
  const int J = cf.J();
  const dim3 block(kWave);
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const dim3 per_draw((unsigned)((n_draw + kWave - 1) / kWave));
  if (parts & 1) {
    if (EXO_GP_WIDE_LDS && J >= kWideMinJ) {
      hipLaunchKernelGGL(celerite_badj_prep_wide_kernel, dim3((unsigned)n_draw, (unsigned)(cg.C - 1)), dim3(256), 0, st, gloglike, n,
                         n_draw, J, wstate, cg, cf.row);
    } else if (EXO_GP_GROUP_TREES && J >= EXO_GP_BADJ_GROUP_MIN_J && J <= 8) {
      const dim3 ggrid((unsigned)(((int64_t)(cg.C - 1) * n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
      EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_badj_prep_group_kernel<JJ>), ggrid, dim3(kScanBlock), 0, st, gloglike, n, n_draw,
                                                  wstate, cg, cf.row))
    } else {
      EXO_GP_DISPATCH_LE8(J, hipLaunchKernelGGL((celerite_badj_prep_kernel<JJ>), dim3(per_draw.x, (unsigned)(cg.C - 1)), block,
                                                0, st, gloglike, n, n_draw, wstate, cg, cf.row))
    }
    if (cg.lane) {
      // draws flagged kFlagRobust: those inputs once more, from the chunks' own reverse recurrences (chunk_adj_lane)
      EXO_GP_DISPATCH_LANE(J, hipLaunchKernelGGL((celerite_chunk_adj_kernel<JJ>), dim3((unsigned)(cg.C - 1), (unsigned)((n_draw + kAdjDraws - 1) / kAdjDraws)), block, 0,
                                                 st, t, resid, diag, n_diag, n, cf, n_draw, gloglike, wstate, cg))
    }
  }
  if (parts & 2) {
    {
      // (B') as a tree over positions p = C - 1 - chunk: adjoint elements of chunks C - 1 .. 1, zero initial adjoint
      bool ok = true;
      auto launch = [&](const TreeOp& op, bool down) {
                  const dim3 tgrid((unsigned)(((int64_t)op.n_item * n_draw + kWave - 1) / kWave));
                  const dim3 ggrid((unsigned)(((int64_t)op.n_item * n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
                  if (EXO_GP_GROUP_TREES && J >= 3 && J <= 8) {
                    if (down) {
                      EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_group_kernel<JJ, true, true>), ggrid, dim3(kScanBlock), 0, st, op, wstate))
                    } else {
                      EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_group_kernel<JJ, true, false>), ggrid, dim3(kScanBlock), 0, st, op, wstate))
                    }
                  } else if (EXO_GP_WIDE_LDS && J >= kWideMinJ) {
                    const dim3 wgrid((unsigned)((int64_t)op.n_item * n_draw));
                    if (down) hipLaunchKernelGGL((celerite_tree_wide_kernel<true, true>), wgrid, dim3(256), 0, st, op, wstate);
                    else hipLaunchKernelGGL((celerite_tree_wide_kernel<true, false>), wgrid, dim3(256), 0, st, op, wstate);
                  } else if (down) {
                    EXO_GP_DISPATCH_TREE(J, hipLaunchKernelGGL((celerite_tree_kernel<JJ, true, true>), tgrid, block, 0, st, op, wstate))
                  } else {
                    EXO_GP_DISPATCH_TREE(J, hipLaunchKernelGGL((celerite_tree_kernel<JJ, true, false>), tgrid, block, 0, st, op, wstate))
                  }
                };
      auto seed = [&]() {
                  ok = exo::zero_fill_async(wstate + ws.tree_state(ws.tree_top()), (int64_t)ws.B() * n_draw, st);   // (never a memset node: exo_math.hpp)
                };
      if (EXO_GP_TREE4 && J <= 2) {
        tree_scan4(ws, J, true, launch,
                   [&](const TreeOp& a, const TreeOp& b, bool down) {
                     const dim3 tgrid((unsigned)(((int64_t)b.n_item * n_draw + kWave - 1) / kWave));
                     if (J == 1) {
                       if (down) hipLaunchKernelGGL((celerite_tree4_kernel<1, true, true>), tgrid, block, 0, st, a, b, wstate);
                       else hipLaunchKernelGGL((celerite_tree4_kernel<1, true, false>), tgrid, block, 0, st, a, b, wstate);
                     } else {
                       if (down) hipLaunchKernelGGL((celerite_tree4_kernel<2, true, true>), tgrid, block, 0, st, a, b, wstate);
                       else hipLaunchKernelGGL((celerite_tree4_kernel<2, true, false>), tgrid, block, 0, st, a, b, wstate);
                     }
                   },
                   seed);
      } else {
        tree_scan_top(ws, J, true, (EXO_GP_GROUP_TREES && J >= 3 && J <= 8) ? tree_serial_level(ws, J) : ws.tree_top(), launch, seed,
                      [&](const TreeOp& op) {
                        const dim3 sgrid((unsigned)((n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
                        EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_serial_group_kernel<JJ, true>), sgrid, dim3(kScanBlock), 0, st, op, wstate))
                      });
      }
      if (!ok) return EXO_ERR_LAUNCH;
    }
  }
  return launch_status();
}

static int celerite_fwd(const double* t, Series resid, const double* diag, int64_t n_diag, int64_t n, Coefs cf,
                        int64_t n_draw, double* loglike, double* state, int64_t state_doubles, int32_t n_chunks,
                        void* stream) {
  if (n_draw == 0) return EXO_OK;
  const bool want_prep = n_chunks >= 0 && (n_chunks & EXO_GP_PREPARE_ADJOINT) != 0;   // (a flag riding on n_chunks: the same in both calls of a pair)
  if (want_prep) n_chunks &= ~EXO_GP_PREPARE_ADJOINT;
  cf.origin = n > 0 ? t : nullptr;   // phases from the first time stamp (Coefs::origin)
  if (!gp_args_ok(n, n_diag, cf.n_real, cf.n_complex, n_draw, n_chunks) || !t || !resid.y || !diag || !loglike ||
      (cf.n_real > 0 && !cf.real) || (cf.n_complex > 0 && !cf.cplx))
    return EXO_ERR_INVALID_ARGUMENT;
  if (state && state_doubles < exo_celerite_state_doubles(n, n_draw, cf.n_real, cf.n_complex, n_chunks))
    return EXO_ERR_WORKSPACE;
  const int J = cf.J();
  const int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : (J <= 8 ? 8 : 16)));
  const int64_t per_wave = kWave / G;
  const dim3 grid((unsigned)((n_draw + per_wave - 1) / per_wave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  if (state) {
    const ChunkGeom cg = chunk_plan(n, n_draw, J, n_chunks);
    const double* only_flagged = nullptr;
    int n_slice = 0;
    SideStream* ss = nullptr;
    auto fork_side = [&]() -> SideStream* {   // the second stream, ordered behind everything issued on `st` so far
      SideStream* q = side_stream();
      if (q && (hipEventRecord(q->fork, st) != hipSuccess || hipStreamWaitEvent(q->s, q->fork, 0) != hipSuccess)) {
        (void)hipGetLastError();
        q = nullptr;
      }
      return q;
    };
    if (cg.C <= 1) {
      const int64_t n_el = n * n_draw * J;
      hipLaunchKernelGGL(celerite_prep_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st, t, n, cf, n_draw,
                         J, state);
      if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
    } else {
      // time-parallel path: flags (+ pre-pass for flagged draws), elements, entering states,
      // recurrences per chunk, sum of the partials
      const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
      const dim3 per_draw((unsigned)((n_draw + kWave - 1) / kWave));
      EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_flag_kernel<JJ>), per_draw, block, 0, st, cf, n_draw,
                                            state + ws.off_flag()))
      const dim3 egrid(per_draw.x, (unsigned)cg.C), cgrid(grid.x, (unsigned)cg.C);
#ifndef EXO_ELEM_LG_MIN_J
#define EXO_ELEM_LG_MIN_J 7
#endif
      // the elements are built at the finest level (ChunkGeom::fine: 2^fine times more, shorter chunks) and
      // composed pairwise up to the chunks the scans and the chunk kernels work on
      const ChunkGeom cge = cg.fine ? fine_geom(n, n_draw, J, cg, cg.fine) : cg;
      const int64_t flag_at = ws.off_flag();
      const dim3 egrid_f(per_draw.x, (unsigned)cge.C), cgrid_f(grid.x, (unsigned)cge.C);
      if (J == 2 && cf.n_real == 0 && cf.kind && EXO_GP_MIXED_ONE_LAUNCH)   // per-draw pair kinds: no wave of mixed kinds (mixed_draw)
        hipLaunchKernelGGL(celerite_kind_partition_kernel, dim3(1), dim3(1024), 0, st, cf.kind, cf.row, n_draw,
                           reinterpret_cast<int32_t*>(state + ws.off_perm()), ws.perm_lanes());
      if (resid.sp.nseg && cg.lane)   // sparse model: the order in which the one-lane kernels' blocks take the chunks (chunk_of_block)
        hipLaunchKernelGGL(celerite_sparse_order_kernel, dim3(1), dim3(256), 0, st, resid.sp, n, n_draw, cg,
                           reinterpret_cast<int32_t*>(state + ws.off_order()));
      if (J >= EXO_ELEM_LG_MIN_J) {   // the one-lane element kernel is as fast up to J = 6 and does not fit beyond
        EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_elem_lg_kernel<JJ>), cgrid_f, block, 0, st, t, resid, diag, n_diag,
                                              n, cf, n_draw, state, cge, flag_at))
      } else {
        auto launch_elem = [&](auto sp_tag) -> int {
          constexpr int SP = decltype(sp_tag)::value;
          EXO_GP_LAYOUTS(J, cf, hipLaunchKernelGGL((celerite_elem_kernel<JJ, NR, SP>), egrid_f, block, 0, st, t, resid, diag,
                                                   n_diag, n, cf, n_draw, state, cge, flag_at),
                         hipLaunchKernelGGL(celerite_elem_mixed_kernel<SP>, dim3(egrid_f.x + 1, egrid_f.y, 2), block, 0, st, t, resid, diag,
                                            n_diag, n, cf, n_draw, state, cge, flag_at))
          return EXO_OK;
        };
        EXO_GP_BY_SERIES(resid, launch_elem)
      }
      for (int f = cg.fine; f >= 1; --f) {
        TreeOp op{};
        op.J = J; op.n_draw = n_draw;
        op.src_elem = ws.off_fine(f); op.src_n = op.src_len = cg.C << f;
        op.dst_elem = f > 1 ? ws.off_fine(f - 1) : ws.elem(0, 0, 0);
        op.n_item = cg.C << (f - 1);
        if (EXO_GP_GROUP_TREES && EXO_GP_FINE_GROUP && J >= 3 && J <= 8) {
          // the scan trees' own item kernel (eight lanes per composition, 32 draws per block): the wave-per-composition LDS kernel
          // took 227 us for the 32 768 compositions of the C5 shape at J = 8, a level of the tree 26 us for 16 384
          const dim3 ggrid((unsigned)(((int64_t)op.n_item * n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
          EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_group_kernel<JJ, false, false>), ggrid, dim3(kScanBlock), 0, st, op, state))
        } else {
          hipLaunchKernelGGL(celerite_compose_lds_kernel, dim3((unsigned)(op.n_item * n_draw)), block, 0, st, op, state);
        }
      }
      // after the element kernel: it may flag more draws (measurement variance too small)
      hipLaunchKernelGGL(celerite_prep_flagged_kernel, dim3(8, (unsigned)n_draw), dim3(256), 0, st, t, n, cf, n_draw, J,
                         state, state + ws.off_flag());
      {
        // (B) as a tree: compose up to one position, seed it with the initial state, apply back down -- a launch per level
        int rc = EXO_OK;
        auto launch = [&](const TreeOp& op, bool down) {
                    const dim3 tgrid((unsigned)(((int64_t)op.n_item * n_draw + kWave - 1) / kWave));
                    const dim3 ggrid((unsigned)(((int64_t)op.n_item * n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
                    if (EXO_GP_GROUP_TREES && J >= 3 && J <= 8) {
                      if (down) {
                        EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_group_kernel<JJ, false, true>), ggrid, dim3(kScanBlock), 0, st, op, state))
                      } else {
                        EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_group_kernel<JJ, false, false>), ggrid, dim3(kScanBlock), 0, st, op, state))
                      }
                    } else if (EXO_GP_WIDE_LDS && J >= kWideMinJ) {   // a block per item, matrices in LDS (celerite_tree_wide_kernel)
                      const dim3 wgrid((unsigned)((int64_t)op.n_item * n_draw));
                      if (down) hipLaunchKernelGGL((celerite_tree_wide_kernel<false, true>), wgrid, dim3(256), 0, st, op, state);
                      else hipLaunchKernelGGL((celerite_tree_wide_kernel<false, false>), wgrid, dim3(256), 0, st, op, state);
                    } else if (down) {
                      EXO_GP_DISPATCH_TREE(J, hipLaunchKernelGGL((celerite_tree_kernel<JJ, false, true>), tgrid, block, 0, st, op, state))
                    } else if (J >= 3 && J <= 8) {
                      // composing two filtering elements keeps ~5 J x J matrices alive around the solve: one lane per
                      // item spills 2 KB at J = 6 and crawls (76 us per level); a wave per item with the tiles in LDS
                      hipLaunchKernelGGL(celerite_compose_lds_kernel, dim3((unsigned)(op.n_item * n_draw)), block, 0,
                                         st, op, state);
                    } else {
                      EXO_GP_DISPATCH_TREE(J, hipLaunchKernelGGL((celerite_tree_kernel<JJ, false, false>), tgrid, block, 0, st, op, state))
                    }
                  };
        auto seed = [&]() {
                    EXO_GP_DISPATCH_VOID(J, hipLaunchKernelGGL((celerite_scan_init_kernel<JJ>), grid, block, 0, st, t, cf, n_draw,
                                                               state + ws.tree_state(ws.tree_top())))
                  };
        if (EXO_GP_TREE4 && J <= 2) {
          tree_scan4(ws, J, false, launch,
                     [&](const TreeOp& a, const TreeOp& b, bool down) {
                       const dim3 tgrid((unsigned)(((int64_t)b.n_item * n_draw + kWave - 1) / kWave));
                       if (J == 1) {
                         if (down) hipLaunchKernelGGL((celerite_tree4_kernel<1, false, true>), tgrid, block, 0, st, a, b, state);
                         else hipLaunchKernelGGL((celerite_tree4_kernel<1, false, false>), tgrid, block, 0, st, a, b, state);
                       } else {
                         if (down) hipLaunchKernelGGL((celerite_tree4_kernel<2, false, true>), tgrid, block, 0, st, a, b, state);
                         else hipLaunchKernelGGL((celerite_tree4_kernel<2, false, false>), tgrid, block, 0, st, a, b, state);
                       }
                     },
                     seed);
        } else {
          tree_scan_top(ws, J, false, (EXO_GP_GROUP_TREES && J >= 3 && J <= 8) ? tree_serial_level(ws, J) : ws.tree_top(), launch, seed,
                        [&](const TreeOp& op) {
                          const dim3 sgrid((unsigned)((n_draw + kScanBlock / 8 - 1) / (kScanBlock / 8)));
                          EXO_GP_DISPATCH_GROUP(J, hipLaunchKernelGGL((celerite_tree_serial_group_kernel<JJ, false>), sgrid, dim3(kScanBlock), 0, st, op, state))
                        });
        }
        if (rc != EXO_OK) return rc;
      }
      if (cg.lane) {
        // draws flagged kFlagRobust: their entering states once more -- Newton iterations from the trees' (J <= 2: the elements
        // applied one after the other)
        const dim3 rgrid((unsigned)(J >= 3 ? (n_draw + kWave / 8 - 1) / (kWave / 8) : per_draw.x));
        if (EXO_GP_ROBUST_NEWTON && J >= 3) {
          EXO_GP_DISPATCH_NEWTON(J, hipLaunchKernelGGL((celerite_robust_newton_kernel<JJ>), dim3((unsigned)n_draw), dim3(kNewtonBlock), 0, st,
                                                       n, cg, n_draw, state))
        } else {
          EXO_GP_DISPATCH_LANE(J, hipLaunchKernelGGL((celerite_robust_scan_kernel<JJ>), rgrid, block, 0, st, t, cf, n, cg, n_draw,
                                                     state))
        }
        // (fwd_st, fwd_cg: the stream and the draws -- ChunkGeom::which -- of this launch)
        hipStream_t fwd_st = st;
        ChunkGeom fwd_cg = cg;
        auto launch_fwd = [&](auto sp_tag) -> int {
          constexpr int SP = decltype(sp_tag)::value;
          EXO_GP_LAYOUTS(J, cf, hipLaunchKernelGGL((celerite_chunk1_fwd_kernel<(JJ <= kLaneMaxJ ? JJ : 1), NR, SP>), egrid, block, 0,
                                                   fwd_st, t, resid, diag, n_diag, n, cf, n_draw, state, fwd_cg),
                         hipLaunchKernelGGL(celerite_chunk1_fwd_mixed_kernel<SP>, dim3(egrid.x + 1, egrid.y, 2), block, 0, fwd_st, t, resid, diag,
                                            n_diag, n, cf, n_draw, state, fwd_cg))
          return EXO_OK;
        };
        if (want_prep) {
          // the adjoint elements first, on the caller's stream: badj_prep and chunk_adj take a SIMD's whole register file (482 /
          // 512 registers at J = 6), so beside the forward chunk kernel -- a resident wave on every SIMD -- they would only wait
          // for it, the scan behind them.  chunk_adj reads the forward checkpoints of the draws on the robust route: their
          // recurrences run now (ChunkGeom::which = 1; no such draw: two launches that return at once), everybody else's beside
          // the scan.
          int rc;
          fwd_cg.which = 1;
          EXO_GP_BY_SERIES(resid, launch_fwd)
          rc = celerite_adjoint_scan(t, resid, diag, n_diag, n, cf, n_draw, nullptr, state, cg, st, 1);
          if (rc != EXO_OK) return rc;
          fwd_cg.which = 2;
          ss = fork_side();
          rc = celerite_adjoint_scan(t, resid, diag, n_diag, n, cf, n_draw, nullptr, state, cg, ss ? ss->s : st, 2);
          if (rc != EXO_OK) return rc;
        }
        EXO_GP_BY_SERIES(resid, launch_fwd)
      } else {
        if (want_prep) {
          // lane-group kernels: J = 7, 8 -- badj_prep fills the register file: first, on the caller's stream; J >= 9 -- a
          // block per item, 81 registers: with the scan on the second stream
          const bool wide = EXO_GP_WIDE_LDS && J >= kWideMinJ;
          int rc = EXO_OK;
          if (!wide) rc = celerite_adjoint_scan(t, resid, diag, n_diag, n, cf, n_draw, nullptr, state, cg, st, 1);
          if (rc != EXO_OK) return rc;
          ss = fork_side();
          rc = celerite_adjoint_scan(t, resid, diag, n_diag, n, cf, n_draw, nullptr, state, cg, ss ? ss->s : st, wide ? 3 : 2);
          if (rc != EXO_OK) return rc;
        }
        EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_chunk_fwd_kernel<JJ>), cgrid, block, 0, st, t, resid, diag, n_diag,
                                              n, cf, n_draw, state, cg))
      }
      if (ss) {   // join: the sums below (and the caller) wait for the second stream
        if (hipEventRecord(ss->join, ss->s) != hipSuccess || hipStreamWaitEvent(st, ss->join, 0) != hipSuccess) return EXO_ERR_LAUNCH;
      }   // (no second stream to be had: the scan went in line, before the chunk kernel -- the reverse call expects it done)
      {
        const int S = chunk_sum_slices(cg.C, 3, n_draw, 16);   // (the last kernel's lanes add the S partials themselves)
        hipLaunchKernelGGL(celerite_chunk_slice_sum_kernel, dim3(per_draw.x, 3, (unsigned)S), block, 0, st, state + ws.off_part(), 3,
                           cg.C, S, n_draw);
        n_slice = S;   // (the sums' last step: the unflagged lanes of the sequential kernel below)
      }
      if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
      only_flagged = state + ws.off_flag();
    }
    EXO_GP_DISPATCH_SEQ(J, hipLaunchKernelGGL((celerite_fwd_kernel<JJ, true>), grid, block, 0, st, t, resid, diag, n_diag, n,
                                          cf, n_draw, loglike, state, only_flagged, cg, n_slice))
  } else {
    EXO_GP_DISPATCH_SEQ(J, hipLaunchKernelGGL((celerite_fwd_kernel<JJ, false>), grid, block, 0, st, t, resid, diag, n_diag, n,
                                          cf, n_draw, loglike, state, (const double*)nullptr, ChunkGeom{}, 0))
  }
  return launch_status();
}

static int celerite_vjp(const double* t, Series resid, const double* diag, int64_t n_diag, int64_t n, Coefs cf,
                        int64_t n_draw, const double* gloglike, const double* state, int64_t state_doubles,
                        int32_t n_chunks, double* gresid, double gsign, double* gdiag, double* gdiag_sum,
                        double* gcoef_real, double* gcoef_complex, void* stream) {
  if (n_draw == 0) return EXO_OK;
  const bool prepared = n_chunks >= 0 && (n_chunks & EXO_GP_PREPARE_ADJOINT) != 0;   // the forward call ran the adjoint scan (celerite_adjoint_scan)
  if (prepared) n_chunks &= ~EXO_GP_PREPARE_ADJOINT;
  cf.origin = n > 0 ? t : nullptr;   // (the forward call's)
  if (!gp_args_ok(n, n_diag, cf.n_real, cf.n_complex, n_draw, n_chunks) || !t || !resid.y || !diag || !gloglike ||
      !state || !gresid || (cf.n_real > 0 && (!cf.real || !gcoef_real)) ||
      (cf.n_complex > 0 && (!cf.cplx || !gcoef_complex)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (state_doubles < exo_celerite_state_doubles(n, n_draw, cf.n_real, cf.n_complex, n_chunks)) return EXO_ERR_WORKSPACE;
  const int J = cf.J();
  const int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : (J <= 8 ? 8 : 16)));
  const int64_t per_wave = kWave / G;
  const dim3 grid((unsigned)((n_draw + per_wave - 1) / per_wave)), block(kWave);
  hipStream_t st = (hipStream_t)stream;
  ChunkGeom cg = chunk_plan(n, n_draw, J, n_chunks);   // the same plan as the forward call's
  cg.prep = prepared ? 1 : 0;
  const double* only_flagged = nullptr;
  if (cg.C > 1) {
    double* wstate = const_cast<double*>(state);   // the chunk workspace lives behind the saved factorisation
    const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
    const dim3 per_draw((unsigned)((n_draw + kWave - 1) / kWave)), cgrid(grid.x, (unsigned)cg.C),
        egrid(per_draw.x, (unsigned)cg.C);
    if (!prepared) {
      const int rc = celerite_adjoint_scan(t, resid, diag, n_diag, n, cf, n_draw, gloglike, wstate, cg, st);
      if (rc != EXO_OK) return rc;
    }
    if (cg.lane) {
      auto launch_vjp = [&](auto sp_tag) -> int {
        constexpr int SP = decltype(sp_tag)::value;
        EXO_GP_LAYOUTS(J, cf, hipLaunchKernelGGL((celerite_chunk1_vjp_kernel<(JJ <= kLaneMaxJ ? JJ : 1), NR, SP>), egrid, block, 0,
                                                 st, t, resid, diag, n_diag, n, cf, n_draw, gloglike, wstate, cg, gresid,
                                                 gdiag, gsign),
                       hipLaunchKernelGGL(celerite_chunk1_vjp_mixed_kernel<SP>, dim3(egrid.x + 1, egrid.y, 2), block, 0, st, t, resid, diag,
                                          n_diag, n, cf, n_draw, gloglike, wstate, cg, gresid, gdiag, gsign))
        return EXO_OK;
      };
      EXO_GP_BY_SERIES(resid, launch_vjp)
    } else {
      EXO_GP_DISPATCH(J, hipLaunchKernelGGL((celerite_chunk_vjp_kernel<JJ>), cgrid, block, 0, st, t, n, cf, n_draw,
                                            gloglike, wstate, cg, gresid, gdiag, gsign, resid))
    }
    {
      const int K = 4 * J + 1, S = chunk_sum_slices(cg.C, K, n_draw);
      hipLaunchKernelGGL(celerite_chunk_slice_sum_kernel, dim3(per_draw.x, (unsigned)K, (unsigned)S), block, 0, st,
                         wstate + ws.off_gpart(), K, cg.C, S, n_draw);
      // (the S partial sums of the 4 J + 1 quantities once more, into chunk 0's slots: a lane of the last kernel would read
      // S rows per quantity one after the other -- 41 at the C5 shape)
      if (S > 1)
        hipLaunchKernelGGL(celerite_chunk_slice_sum_kernel, dim3(per_draw.x, (unsigned)K, 1), block, 0, st, wstate + ws.off_gpart(), K,
                           S, 1, n_draw);
      // (the combination into gcoef_*: the unflagged lanes of the sequential kernel below)
    }
    if (launch_status() != EXO_OK) return EXO_ERR_LAUNCH;
    only_flagged = state + ws.off_flag();
  }
  EXO_GP_DISPATCH_SEQ(J, hipLaunchKernelGGL((celerite_vjp_kernel<JJ>), grid, block, 0, st, t, diag, n_diag, n, cf, n_draw,
                                        gloglike, state, gresid, gdiag, gdiag_sum, gcoef_real, gcoef_complex,
                                        only_flagged, gsign, resid, cg))
  return launch_status();
}

int exo_celerite_loglike_fwd_f64(const double* t, const double* resid, const double* diag, int64_t n_diag,
                                 int64_t n, const double* coef_real, int32_t n_real, const double* coef_complex,
                                 int32_t n_complex, const int32_t* pair_kind, int64_t n_draw, double* loglike,
                                 double* state, int64_t state_doubles, int32_t n_chunks, void* stream) {
  return celerite_fwd(t, Series{resid, nullptr, 0}, diag, n_diag, n, Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex},
                      n_draw, loglike, state, state_doubles, n_chunks, stream);
}

int exo_celerite_loglike_vjp_f64(const double* t, const double* resid, const double* diag, int64_t n_diag, int64_t n,
                                 const double* coef_real, int32_t n_real, const double* coef_complex,
                                 int32_t n_complex, const int32_t* pair_kind, int64_t n_draw, const double* gloglike,
                                 const double* state, int64_t state_doubles, int32_t n_chunks, double* gresid,
                                 double* gdiag, double* gdiag_sum, double* gcoef_real, double* gcoef_complex,
                                 void* stream) {
  return celerite_vjp(t, Series{resid, nullptr, 0}, diag, n_diag, n, Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex},
                      n_draw, gloglike, state, state_doubles, n_chunks, gresid, 1.0, gdiag, gdiag_sum, gcoef_real,
                      gcoef_complex, stream);
}

int exo_celerite_loglike_obs_fwd_f64(const double* t, const double* obs, const double* model, const double* diag,
                                     int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                     const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                     int64_t n_draw, double* loglike, double* state, int64_t state_doubles,
                                     int32_t n_chunks, void* stream) {
  if (n_draw > 0 && !obs) return EXO_ERR_INVALID_ARGUMENT;
  return celerite_fwd(t, Series{model, obs, 0}, diag, n_diag, n, Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex},
                      n_draw, loglike, state, state_doubles, n_chunks, stream);
}

int exo_celerite_loglike_obs_vjp_f64(const double* t, const double* obs, const double* model, const double* diag,
                                     int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                     const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                     int64_t n_draw, const double* gloglike, const double* state,
                                     int64_t state_doubles, int32_t n_chunks, double* gmodel, double* gdiag,
                                     double* gdiag_sum, double* gcoef_real, double* gcoef_complex, void* stream) {
  if (n_draw > 0 && !obs) return EXO_ERR_INVALID_ARGUMENT;
  return celerite_vjp(t, Series{model, obs, 0}, diag, n_diag, n, Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex},
                      n_draw, gloglike, state, state_doubles, n_chunks, gmodel, -1.0, gdiag, gdiag_sum, gcoef_real,
                      gcoef_complex, stream);
}

int exo_celerite_loglike_obs_fwd_cm_f64(const double* t, const double* obs, const double* model_cm, const double* diag,
                                        int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                        int64_t n_draw, double* loglike, double* state, int64_t state_doubles,
                                        int32_t n_chunks, void* stream) {
  if (n_draw > 0 && !obs) return EXO_ERR_INVALID_ARGUMENT;
  return celerite_fwd(t, Series{model_cm, obs, n_draw}, diag, n_diag, n,
                      Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex}, n_draw, loglike, state, state_doubles,
                      n_chunks, stream);
}

int exo_celerite_loglike_obs_vjp_cm_f64(const double* t, const double* obs, const double* model_cm, const double* diag,
                                        int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                        int64_t n_draw, const double* gloglike, const double* state,
                                        int64_t state_doubles, int32_t n_chunks, double* gmodel_cm, double* gdiag,
                                        double* gdiag_sum, double* gcoef_real, double* gcoef_complex, void* stream) {
  if (n_draw > 0 && !obs) return EXO_ERR_INVALID_ARGUMENT;
  return celerite_vjp(t, Series{model_cm, obs, n_draw}, diag, n_diag, n,
                      Coefs{coef_real, coef_complex, pair_kind, n_real, n_complex}, n_draw, gloglike, state, state_doubles,
                      n_chunks, gmodel_cm, -1.0, gdiag, gdiag_sum, gcoef_real, gcoef_complex, stream);
}

// the model as the light-curve sweep's SPARSE output (exo_sparse_model): segments of cadences + their values
static bool sparse_series(const exo_sparse_model* m, const double* obs, int64_t n, Series* out) {
  if (!m || !obs || !m->nseg || !m->seg || !m->off || !m->vals || m->seg_step < 1 || m->hi_at < 0 || m->hi_at >= m->seg_step ||
      m->seg_row < 0 || m->off_row < 0 || m->val_row < 0 || n > 0x7fffffff)
    return false;
  out->y = m->vals;
  out->obs = obs;
  out->cm = 0;
  out->sp.nseg = m->nseg; out->sp.seg = m->seg; out->sp.off = m->off;
  out->sp.seg_row = m->seg_row; out->sp.off_row = m->off_row; out->sp.val_row = m->val_row;
  out->sp.seg_step = m->seg_step; out->sp.hi_at = m->hi_at;
  out->sp.row_of_draw = m->row_of_draw;
  return true;
}

int exo_celerite_loglike_sparse_fwd_f64(const double* t, const double* obs, const exo_sparse_model* model, const double* diag,
                                        int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                        int64_t n_draw, double* loglike, double* state, int64_t state_doubles,
                                        int32_t n_chunks, void* stream) {
  if (n_draw == 0) return EXO_OK;
  Series rs{};
  if (!sparse_series(model, obs, n, &rs)) return EXO_ERR_INVALID_ARGUMENT;
  Coefs cf{coef_real, coef_complex, pair_kind, n_real, n_complex};
  cf.row = model->row_of_draw;     // (every per-draw array of the call in the caller's order: Coefs::row)
  return celerite_fwd(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, state_doubles, n_chunks, stream);
}

int exo_celerite_loglike_sparse_vjp_f64(const double* t, const double* obs, const exo_sparse_model* model, const double* diag,
                                        int64_t n_diag, int64_t n, const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                        int64_t n_draw, const double* gloglike, const double* state,
                                        int64_t state_doubles, int32_t n_chunks, double* gvals, double* gdiag,
                                        double* gdiag_sum, double* gcoef_real, double* gcoef_complex, void* stream) {
  if (n_draw == 0) return EXO_OK;
  Series rs{};
  if (!sparse_series(model, obs, n, &rs)) return EXO_ERR_INVALID_ARGUMENT;
  Coefs cf{coef_real, coef_complex, pair_kind, n_real, n_complex};
  cf.row = model->row_of_draw;
  return celerite_vjp(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, state_doubles, n_chunks, gvals, -1.0, gdiag, gdiag_sum,
                      gcoef_real, gcoef_complex, stream);
}

// The order in which a sparse model's draws are best handed to the celerite kernels (exo_sparse_model.row_of_draw): ascending
// mean spacing of a draw's segments -- the period, in cadences: draws with neighbouring periods keep their transits together all
// along the series -- the start of the first segment breaking ties (draws with fewer than two segments: by that alone), then the
// draw's index.  One block: keys into LDS, a bitonic sort of (key, index) pairs.  Up to kOrderMaxDraws draws.
int exo_sparse_model_order(const exo_sparse_model* model, int64_t n_draw, int32_t* order, void* stream) {
  if (n_draw == 0) return EXO_OK;
  if (!model || !order || !model->nseg || !model->seg || model->seg_step < 1 || n_draw < 0 || n_draw > kOrderMaxDraws)
    return EXO_ERR_INVALID_ARGUMENT;
  SparseSegs sp{};
  sp.nseg = model->nseg; sp.seg = model->seg; sp.seg_row = model->seg_row; sp.seg_step = model->seg_step; sp.hi_at = model->hi_at;
  hipLaunchKernelGGL(celerite_draw_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sp, n_draw, order);
  return launch_status();
}

int exo_celerite_dot_tril_f64(const double* t, const double* diag, int64_t n_diag, int64_t n, const double* coef_real,
                              int32_t n_real, const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                              int64_t n_draw, const double* x, double* z, void* stream) {
  if (n_draw == 0) return EXO_OK;
  if (!gp_args_ok(n, n_diag, n_real, n_complex, n_draw, 0) || !t || !diag || !x || !z || (n_real > 0 && !coef_real) ||
      (n_complex > 0 && !coef_complex))
    return EXO_ERR_INVALID_ARGUMENT;
  const Coefs cf{coef_real, coef_complex, pair_kind, n_real, n_complex, n > 0 ? t : nullptr};
  const dim3 grid((unsigned)((n_draw + kWave - 1) / kWave)), block(kWave);
  EXO_GP_DISPATCH_SEQ(cf.J(), hipLaunchKernelGGL((celerite_dot_tril_kernel<JJ>), grid, block, 0, (hipStream_t)stream, t, diag,
                                             n_diag, n, cf, n_draw, x, z))
  return launch_status();
}

int exo_celerite_predict_f64(const double* t, int64_t n, const double* alpha, const double* coef_real, int32_t n_real,
                             const double* coef_complex, int32_t n_complex, const int32_t* pair_kind, int64_t n_draw,
                             const double* tq, int64_t m, double* mu, void* stream) {
  if (n_draw == 0 || m == 0) return EXO_OK;
  if (!gp_args_ok(n, 1, n_real, n_complex, n_draw, 0) || m < 0 || !t || !alpha || !tq || !mu ||
      (n_real > 0 && !coef_real) || (n_complex > 0 && !coef_complex))
    return EXO_ERR_INVALID_ARGUMENT;
  const Coefs cf{coef_real, coef_complex, pair_kind, n_real, n_complex, n > 0 ? t : nullptr};
  const dim3 grid((unsigned)((n_draw + kWave - 1) / kWave)), block(kWave);
  EXO_GP_DISPATCH_SEQ(cf.J(), hipLaunchKernelGGL((celerite_predict_kernel<JJ>), grid, block, 0, (hipStream_t)stream, t, n, alpha,
                                             cf, n_draw, tq, m, mu))
  return launch_status();
}

}  // extern "C"

