// exo_ops.hip -- the reference's three standalone Ops as elementwise kernels (gfx950, wave64, fp64):
//   exo_kepler_f64                 ops.kepler(M, ecc) -> (sinf, cosf)            keplerian.py:333,818
//   exo_quad_solution_vector_f64   ops.quad_solution_vector(b, r) -> s [, ds/db, ds/dr]     limb_dark.py:24
//   exo_contact_points_f64         ops.contact_points(a, e, cosw, sinw, cosi, sini, L)      keplerian.py:744-753
// The sweep kernels (exo_transit.hip) inline the same device functions (exo_math.hpp, exo_contact.hpp); these entries are
// what a binding that keeps the reference's Op granularity calls.
//
// Layout of the Kepler kernel (round 6).  32 B per element (M, e in; sin f, cos f out) against ~150 fp64 instructions: at
// 1.5e8 elements that is 0.6 ms of HBM time at the 8 TB/s peak and 0.6 ms of fp64 issue, so the kernel streams only if the
// memory side runs at the wide-access rate AND overlaps the arithmetic.  8-byte accesses reach 0.54-0.70 of the 16-byte
// rate on this chip (MI355X_MICROARCH.md, table of access flavours): a lane takes TWO consecutive elements, one 16-byte load
// per input and one 16-byte store per output, the two solves interleaved by the scheduler (independent chains: the solve is a
// fixed-cost sequence without votes), the next pair's loads issued before this pair's arithmetic.  Unaligned pointers or an
// odd count: the scalar kernel takes them (heads, tails; whole arrays when a pointer is only 8-byte aligned).
// MEASURED (round 6, n = 1.5e8): 3.17 (round 5: one 8-byte access per lane) -> 3.6-3.8 TB/s.  The memory side is not what
// holds it: the circular branch (e = 0: one sincos) runs the same loads and stores at 4.6-5.5 TB/s, which is what a plain
// copy of two arrays in and two out reaches on the same box (4.7-5.2 TB/s: tools/ops_bench.py); the eccentric solve is 197
// vector instructions per element by the counters, 13 of them quarter-rate: ~0.9 ms of issue at the nominal clock, 1.3-1.4 measured.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EXO_SCALAR_CONSTANTS   // polynomial coefficients from scalar registers: pays in these kernels (exo_math.hpp, EXO_K)
#include "../../include/exoplanet_amd.h"
#include "exo_contact.hpp"
#include "exo_math.hpp"

namespace {

constexpr int kBlock = 256;
#ifndef EXO_KEPLER_BLOCKS_PER_CU
#define EXO_KEPLER_BLOCKS_PER_CU 16     // (4 / 8 / 16 blocks per CU: 3.43 / 3.57 / 3.70 TB/s at n = 1.5e8)
#endif

typedef double d2 __attribute__((ext_vector_type(2)));   // a 16-byte access

struct SinCosF {
  double s, c;
};

// (sin f, cos f) of one element; NaN outside 0 <= e < 1 (the reference requires it: keplerian.py:58).  From the half
// angles (sh, ch) = (sin, cos)(E / 2) of the solver:  (1 - e cos E) = (1 - e) ch^2 + (1 + e) sh^2,
// (1 - e cos E) cos f = (1 - e) ch^2 - (1 + e) sh^2,  (1 - e cos E) sin f = 2 sqrt(1 - e^2) sh ch -- ONE square root, of
// (1 - e)(1 + e); no cancellation as e -> 1.
__device__ __forceinline__ SinCosF kepler_one(double M, double e) {
  const bool ok = (e >= 0.0) && (e < 1.0);
  const double es = ok ? e : 0.5;
  const exo::KeplerHalf kh = exo::kepler_half(M, es, 1.0, 1.0);      // X = ch, Y = sh (signed)
  const double ome = 1.0 - es, ope = 1.0 + es;
  const double X2 = ome * (kh.ch * kh.ch), Y2 = ope * (kh.sh * kh.sh);
  const double iden = exo::fast_rcp(X2 + Y2);
  const double sq = exo::fast_sqrt(ome * ope);
  const double nan = __builtin_nan("");
  SinCosF o;
  o.s = ok ? (2.0 * sq) * (kh.sh * kh.ch) * iden : nan;
  o.c = ok ? (X2 - Y2) * iden : nan;
  return o;
}

// 16-byte accesses: PLAIN.  Non-temporal ones (EXO_OPS_NONTEMPORAL) measured no faster for the Kepler kernel (3.55 against 3.57
// TB/s) and much slower for quad_solution_vector on out-of-transit input, the streaming case: 2.57 against 4.19 TB/s.
#ifndef EXO_OPS_NONTEMPORAL
#define EXO_LOAD16(p) (*(p))
#define EXO_STORE16(v, p) (*(p) = (v))
#else
#define EXO_LOAD16(p) __builtin_nontemporal_load(p)
#define EXO_STORE16(v, p) __builtin_nontemporal_store(v, p)
#endif

__global__ __launch_bounds__(kBlock) void kepler_kernel(const double* __restrict__ M, const double* __restrict__ ecc,
                                                        double* __restrict__ sinf, double* __restrict__ cosf, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const SinCosF o = kepler_one(M[i], ecc[i]);
    sinf[i] = o.s;
    cosf[i] = o.c;
  }
}

// pairs: n2 = number of double2 elements.  The trip count is uniform per wave (kepler_half votes on e == 0), the loads of
// the next iteration are issued before the arithmetic of this one.
__global__ __launch_bounds__(kBlock) void kepler_pair_kernel(const d2* __restrict__ M, const d2* __restrict__ ecc,
                                                             d2* __restrict__ sinf, d2* __restrict__ cosf, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t n_round = (n2 + stride - 1) / stride * stride;
  const d2 m_idle = {0.0, 0.0}, e_idle = {0.5, 0.5};
  bool v = i < n2;
  d2 m = v ? EXO_LOAD16(&M[i]) : m_idle;
  d2 e = v ? EXO_LOAD16(&ecc[i]) : e_idle;
  for (; i < n_round; i += stride) {
    const int64_t j = i + stride;
    const bool vn = j < n2;
    const d2 mn = vn ? EXO_LOAD16(&M[j]) : m_idle;
    const d2 en = vn ? EXO_LOAD16(&ecc[j]) : e_idle;
    const SinCosF a = kepler_one(m.x, e.x);
    const SinCosF b = kepler_one(m.y, e.y);
    if (v) {
      const d2 so = {a.s, b.s}, co = {a.c, b.c};
      EXO_STORE16(so, &sinf[i]);
      EXO_STORE16(co, &cosf[i]);
    }
    m = mn; e = en; v = vn;
  }
}

// diagnostic: the fp32 classifier position, so that its error bound can be tested
__global__ __launch_bounds__(kBlock) void orbit_pos_f32_kernel(const double* __restrict__ M,
                                                               const double* __restrict__ ecc,
                                                               double* __restrict__ cx, double* __restrict__ sx,
                                                               int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double e = ecc[i];
  float a, b;
  exo::orbit_pos_f32(M[i], (float)e, (float)(1.0 - e), (float)sqrt((1.0 - e) * (1.0 + e)), &a, &b);
  cx[i] = a;
  sx[i] = b;
}

template <bool GRAD>
__global__ __launch_bounds__(kBlock) void quad_sv_kernel(const double* __restrict__ b,
                                                         const double* __restrict__ r,
                                                         double* __restrict__ s,
                                                         double* __restrict__ dsdb,
                                                         double* __restrict__ dsdr, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  // uniform trip count so the wavefront votes inside quad_sv see whole waves
  const int64_t n_round = (n + stride - 1) / stride * stride;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
    const bool v = i < n;
    const double bs = v ? b[i] : 2.0;
    const double rr = v ? r[i] : 0.1;
    const double sg = bs < 0.0 ? -1.0 : 1.0;
    exo::SV o;
    exo::quad_sv<GRAD>(fabs(bs), rr, o);
    if (v) {
      s[3 * i] = o.s0; s[3 * i + 1] = o.s1; s[3 * i + 2] = o.s2;
      if (GRAD) {
        dsdb[3 * i] = sg * o.db0; dsdb[3 * i + 1] = sg * o.db1; dsdb[3 * i + 2] = sg * o.db2;
        dsdr[3 * i] = o.dr0; dsdr[3 * i + 1] = o.dr1; dsdr[3 * i + 2] = o.dr2;
      }
    }
  }
}

// Two consecutive elements per lane (16-byte aligned b, r, s [, ds/db, ds/dr], n2 = pairs): one 16-byte load per input, and
// the lane's six (value) or eighteen doubles leave as 16-byte stores of CONSECUTIVE addresses (a lane owns 48 contiguous bytes
// of s) -- the one-element kernel writes s with three 8-byte stores of stride 24 per lane.  The two elements go through quad_sv
// one after the other (a rolled loop: the registers of one evaluation), its wave votes seeing whole waves in both passes.
template <bool GRAD>
__global__ __launch_bounds__(kBlock) void quad_sv_pair_kernel(const d2* __restrict__ b, const d2* __restrict__ r,
                                                              d2* __restrict__ s, d2* __restrict__ dsdb,
                                                              d2* __restrict__ dsdr, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_round = (n2 + stride - 1) / stride * stride;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_round; i += stride) {
    const bool v = i < n2;
    const d2 idle_b = {2.0, 2.0}, idle_r = {0.1, 0.1};
    const d2 bb = v ? EXO_LOAD16(&b[i]) : idle_b;
    const d2 rr = v ? EXO_LOAD16(&r[i]) : idle_r;
    double o[2][GRAD ? 9 : 3];
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const double bs = k ? bb.y : bb.x, rk = k ? rr.y : rr.x;
      const double sg = bs < 0.0 ? -1.0 : 1.0;
      exo::SV q;
      exo::quad_sv<GRAD>(fabs(bs), rk, q);
      // (stores at compile-time indices, selected: a run-time index into a register array goes to scratch)
      if (k == 0) {
        o[0][0] = q.s0; o[0][1] = q.s1; o[0][2] = q.s2;
        if (GRAD) { o[0][3] = sg * q.db0; o[0][4] = sg * q.db1; o[0][5] = sg * q.db2; o[0][6] = q.dr0; o[0][7] = q.dr1; o[0][8] = q.dr2; }
      } else {
        o[1][0] = q.s0; o[1][1] = q.s1; o[1][2] = q.s2;
        if (GRAD) { o[1][3] = sg * q.db0; o[1][4] = sg * q.db1; o[1][5] = sg * q.db2; o[1][6] = q.dr0; o[1][7] = q.dr1; o[1][8] = q.dr2; }
      }
    }
    if (v) {
      const d2 s0 = {o[0][0], o[0][1]}, s1 = {o[0][2], o[1][0]}, s2 = {o[1][1], o[1][2]};
      EXO_STORE16(s0, &s[3 * i]); EXO_STORE16(s1, &s[3 * i + 1]); EXO_STORE16(s2, &s[3 * i + 2]);
      if (GRAD) {
        const d2 a0 = {o[0][3], o[0][4]}, a1 = {o[0][5], o[1][3]}, a2 = {o[1][4], o[1][5]};
        EXO_STORE16(a0, &dsdb[3 * i]); EXO_STORE16(a1, &dsdb[3 * i + 1]); EXO_STORE16(a2, &dsdb[3 * i + 2]);
        const d2 c0 = {o[0][6], o[0][7]}, c1 = {o[0][8], o[1][6]}, c2 = {o[1][7], o[1][8]};
        EXO_STORE16(c0, &dsdr[3 * i]); EXO_STORE16(c1, &dsdr[3 * i + 1]); EXO_STORE16(c2, &dsdr[3 * i + 2]);
      }
    }
  }
}

__global__ __launch_bounds__(64) void contact_points_kernel(
    const double* __restrict__ a, const double* __restrict__ e_, const double* __restrict__ cosw,
    const double* __restrict__ sinw, const double* __restrict__ cosi, const double* __restrict__ sini,
    const double* __restrict__ L_, double* __restrict__ Ml, double* __restrict__ Mr,
    int32_t* __restrict__ flag, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  (void)sini;
  double ml, mr;
  const bool bad = exo::contact_solve(a[i], e_[i], cosw[i], sinw[i], cosi[i], L_[i], &ml, &mr);
  Ml[i] = ml;
  Mr[i] = mr;
  flag[i] = bad ? 1 : 0;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EXO_OK : EXO_ERR_LAUNCH; }

inline int elementwise_grid(int64_t n, int blocks_per_cu = 8) {
  const int64_t want = (n + kBlock - 1) / kBlock;
  const int64_t cap = 256 * (int64_t)blocks_per_cu;   // resident blocks, grid-stride the rest
  return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int exo_kepler_f64(const double* M, const double* ecc, double* sinf, double* cosf, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!M || !ecc || !sinf || !cosf))) return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n2 = n / 2;
  if (n2 >= 1024 && aligned16(M) && aligned16(ecc) && aligned16(sinf) && aligned16(cosf)) {
    hipLaunchKernelGGL(kepler_pair_kernel, dim3(elementwise_grid(n2, EXO_KEPLER_BLOCKS_PER_CU)), dim3(kBlock), 0, st,
                       reinterpret_cast<const d2*>(M), reinterpret_cast<const d2*>(ecc), reinterpret_cast<d2*>(sinf),
                       reinterpret_cast<d2*>(cosf), n2);
    if (n & 1)
      hipLaunchKernelGGL(kepler_kernel, dim3(1), dim3(kBlock), 0, st, M + 2 * n2, ecc + 2 * n2, sinf + 2 * n2, cosf + 2 * n2,
                         (int64_t)1);
    return launch_status();
  }
  hipLaunchKernelGGL(kepler_kernel, dim3(elementwise_grid(n)), dim3(kBlock), 0, st, M, ecc, sinf, cosf, n);
  return launch_status();
}

int exo_selftest_orbit_pos_f32(const double* M, const double* ecc, double* cx, double* sx, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!M || !ecc || !cx || !sx))) return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  hipLaunchKernelGGL(orbit_pos_f32_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, M, ecc, cx, sx, n);
  return launch_status();
}

int exo_quad_solution_vector_f64(const double* b, const double* r, double* s, double* dsdb, double* dsdr,
                                 int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!b || !r || !s)) || ((dsdb == nullptr) != (dsdr == nullptr)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(kBlock);
  const int64_t n2 = n / 2;
  int64_t done = 0;
  if (n2 >= 1024 && aligned16(b) && aligned16(r) && aligned16(s) && (!dsdb || (aligned16(dsdb) && aligned16(dsdr)))) {
    const dim3 grid2(elementwise_grid(n2));
    if (dsdb)
      hipLaunchKernelGGL(quad_sv_pair_kernel<true>, grid2, block, 0, st, reinterpret_cast<const d2*>(b), reinterpret_cast<const d2*>(r),
                         reinterpret_cast<d2*>(s), reinterpret_cast<d2*>(dsdb), reinterpret_cast<d2*>(dsdr), n2);
    else
      hipLaunchKernelGGL(quad_sv_pair_kernel<false>, grid2, block, 0, st, reinterpret_cast<const d2*>(b), reinterpret_cast<const d2*>(r),
                         reinterpret_cast<d2*>(s), nullptr, nullptr, n2);
    done = 2 * n2;
    if (done == n) return launch_status();
  }
  const dim3 grid(elementwise_grid(n - done));
  if (dsdb)
    hipLaunchKernelGGL(quad_sv_kernel<true>, grid, block, 0, st, b + done, r + done, s + 3 * done, dsdb + 3 * done, dsdr + 3 * done,
                       n - done);
  else
    hipLaunchKernelGGL(quad_sv_kernel<false>, grid, block, 0, st, b + done, r + done, s + 3 * done, dsdb, dsdr, n - done);
  return launch_status();
}

int exo_contact_points_f64(const double* a, const double* e, const double* cosw, const double* sinw,
                           const double* cosi, const double* sini, const double* L, double* M_left,
                           double* M_right, int32_t* flag, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!a || !e || !cosw || !sinw || !cosi || !sini || !L || !M_left || !M_right || !flag)))
    return EXO_ERR_INVALID_ARGUMENT;
  if (n == 0) return EXO_OK;
  hipLaunchKernelGGL(contact_points_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     a, e, cosw, sinw, cosi, sini, L, M_left, M_right, flag, n);
  return launch_status();
}

}  // extern "C"
