// exo_math.hpp -- fp64 device math for the transit hot path on gfx950 (CDNA4).
//
//   kepler_half()   Kepler's equation -> half-angle true-anomaly components
//   quad_sv()       quadratic limb-darkening solution vector (+ d/db, d/dr)
//
// These replace the arithmetic of the reference's third-party ops
//   ops.kepler               (call site /root/reference/src/exoplanet/orbits/keplerian.py:333)
//   ops.quad_solution_vector (call site /root/reference/src/exoplanet/light_curves/limb_dark.py:24)
// The third-party sources are not available; everything here is derived from
// the mathematical definitions (see docs/DESIGN_r1_r4.md section 3) and checked against
// oracle/mp_reference.py.
//
// One (cadence, sub-exposure, planet) per lane.  No MFMA: there is no dense
// contraction anywhere on this path.  Iterative pieces (the AGM sweeps of the
// elliptic integrals) exit on a wavefront vote so a wave never spins on lanes
// that have already converged and never diverges inside the sweep.
//
// The header is also compiled for the host by tests/ (g++, EXO_HOST_BUILD) to
// check the arithmetic on CPU before it ever runs on a GPU; that build is a
// test harness, not a product path.
#pragma once
#include <math.h>

#ifdef EXO_HOST_BUILD
#define EXO_HD inline
#define EXO_WAVE_ALL(c) (c)
#define EXO_WAVE_ANY(c) (c)
#else
#include <hip/hip_runtime.h>
#define EXO_HD __device__ __forceinline__
// 64-wide wavefront votes (gfx950): uniform branch conditions
#define EXO_WAVE_ALL(c) (__all((int)(c)))
#define EXO_WAVE_ANY(c) (__any((int)(c)))
#endif

// EXO_K(x): the fp64 constant x held in a SCALAR register pair (round 6; opt-in per translation unit: #define
// EXO_SCALAR_CONSTANTS before including this header).  gfx950's VOP3 encoding takes no 64-bit literal, so the compiler
// materialises every polynomial coefficient with two v_mov_b32 and feeds a v_fmac; with EXO_K two s_mov_b32 on the scalar unit
// put the constant into s[92:93] / s[94:95] and the v_fma_f64 reads it from there as its addend (one scalar operand per
// instruction: the constant bus).  `volatile` keeps the moves where they are used: hoisted out of a loop they would cost ~100
// live scalar pairs.  MEASURED (round 6, same box, alternating runs; round 3 had found the same with another form):
//   * standalone Ops (exo_ops.hip, 4-8 waves per SIMD, VALU-issue-bound): quad_solution_vector 1.46 -> 1.70 TB/s (+16 %),
//     kepler 3.43 -> 3.58 TB/s: ON there;
//   * the sweep and the celerite kernels (2-3 waves per SIMD, one long dependent chain per lane): 1236 -> 1063 vector
//     instructions in the sweep's hot loop and 4-6 % SLOWER (sparse sweep 184 -> 194 us, C3 3.36 -> 3.51 ms, C5 1.82 -> 1.95):
//     those kernels are bound by the per-wave chain, every instruction of a wave costs it an issue slot whichever unit
//     executes it, the independent v_movs were filling stall slots for free, and four fewer allocatable scalar registers mean
//     more spills to lanes (14 -> 55 v_readlane in the loop).  OFF there.
#if defined(EXO_HOST_BUILD) || !defined(__HIP_DEVICE_COMPILE__) || !defined(EXO_SCALAR_CONSTANTS)
#define EXO_K(x) (x)
#define EXO_K2(x) (x)
#else
#define EXO_KP_(x, R0, R1)                                                                                             \
  ([]() __attribute__((always_inline)) -> double {                                                                     \
    constexpr unsigned long long u_ = __builtin_bit_cast(unsigned long long, (double)(x));                             \
    double d_;                                                                                                         \
    asm volatile("s_mov_b32 s" #R0 ", %1\n\ts_mov_b32 s" #R1 ", %2"                                                    \
                 : "={s[" #R0 ":" #R1 "]}"(d_)                                                                         \
                 : "i"((unsigned)(u_ & 0xffffffffull)), "i"((unsigned)(u_ >> 32)));                                    \
    return d_;                                                                                                         \
  }())
#define EXO_K(x) EXO_KP_(x, 92, 93)
#define EXO_K2(x) EXO_KP_(x, 94, 95)   // a second pair, for two chains side by side
#endif

namespace exo {

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = 1.57079632679489661923;
constexpr double kTwoPiHi = 6.283185307179586;       // fl(2 pi)
constexpr double kTwoPiLo = 2.4492935982947064e-16;  // 2 pi - fl(2 pi)
constexpr double kTwoThirdsPi = 2.09439510239319549231;

// x - sin(x) for x >= 0 given an independently known sin(x).
// Taylor series below 0.9 (no cancellation), direct above.
EXO_HD double x_minus_sin(double x, double sinx) {
  const double x2 = x * x;
  // x^3/6 (1 - x^2/20 + x^4/840 - ...): coefficients (-1)^k 6/(2k+3)!
  double s = 2.8114572543455207632e-15 * 6.0 * (1.0 / (18.0 * 19.0));   // 6/19!
  s = fma(s, x2, EXO_K(-6.0 * 2.8114572543455207632e-15));                 // -6/17!
  s = fma(s, x2, EXO_K(6.0 * 7.6471637318198164759e-13));                  //  6/15!
  s = fma(s, x2, EXO_K(-6.0 * 1.6059043836821614599e-10));                 // -6/13!
  s = fma(s, x2, EXO_K(6.0 * 2.5052108385441718775e-08));                  //  6/11!
  s = fma(s, x2, EXO_K(-6.0 * 2.7557319223985890653e-06));                 // -6/9!
  s = fma(s, x2, EXO_K(6.0 * 1.9841269841269841270e-04));                  //  6/7!
  s = fma(s, x2, EXO_K(-6.0 * 8.3333333333333333333e-03));                 // -6/5!
  s = fma(s, x2, 1.0);
  s = x * x2 * EXO_K(1.0 / 6.0) * s;
  return (x < 0.9) ? s : (x - sinx);
}

// Approximate-reciprocal division: v_rcp_f64 seed (2^-24.4 measured on gfx950) + one Newton
// step (2^-48.8) + one residual correction of the quotient (error x error: ~1 ulp; no denormal /
// overflow rescue -- callers pass well-scaled operands).  A third of the IEEE sequence.
EXO_HD double fast_div(double x, double y) {
#ifdef EXO_HOST_BUILD
  return x / y;
#else
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  const double q = x * r;
  return fma(fma(-y, q, x), r, q);
#endif
}

// 1/y: seed + two Newton steps (2^-24 -> 2^-49 -> full)
EXO_HD double fast_rcp(double y) {
#ifdef EXO_HOST_BUILD
  return 1.0 / y;
#else
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  return fma(fma(-y, r, 1.0), r, r);
#endif
}

// sqrt from the hardware reciprocal square root (2^-24.2 measured) + one coupled Newton step
// (2^-48) and a residual correction (~1 ulp; x = 0 handled; no denormal rescue).
EXO_HD double fast_sqrt(double x) {
#ifdef EXO_HOST_BUILD
  return sqrt(x);
#else
  const double r0 = __builtin_amdgcn_rsq(x);
  double g = x * r0, h = 0.5 * r0;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return (x > 0.0) ? g : ((x == 0.0) ? 0.0 : __builtin_nan(""));
#endif
}

// sqrt(x) and 1/sqrt(x) together (x > 0): the coupled iteration carries h ~ 1/(2 sqrt x) anyway,
// one more step on h costs two operations -- a division by sqrt(x), or by x, costs seven.
EXO_HD double fast_sqrt_rs(double x, double* rs) {
#ifdef EXO_HOST_BUILD
  const double g = sqrt(x);
  *rs = 1.0 / g;
  return g;
#else
  const double r0 = __builtin_amdgcn_rsq(x);
  double g = x * r0, h = 0.5 * r0;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  e = fma(-h, g, 0.5);
  *rs = 2.0 * fma(h, e, h);
  return g;
#endif
}

// Low-precision building blocks for the fp32 Kepler starter (the starter is only
// good to ~4e-4 by construction, so single-instruction hardware approximations
// -- v_rcp_f32, v_sqrt_f32, v_log_f32 / v_exp_f32 -- are ample).
EXO_HD float fast_rcpf(float y) {
#ifdef EXO_HOST_BUILD
  return 1.0f / y;
#else
  return __builtin_amdgcn_rcpf(y);
#endif
}
EXO_HD float fast_sqrtf(float y) {
#ifdef EXO_HOST_BUILD
  return sqrtf(y);
#else
  return __builtin_amdgcn_sqrtf(y);
#endif
}
EXO_HD float fast_cbrtf(float y) {  // y >= 0
#ifdef EXO_HOST_BUILD
  return cbrtf(y);
#else
  return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(y) * (1.0f / 3.0f));
#endif
}
// reciprocal to ~1e-8 relative (one Newton step on v_rcp_f64): enough for the
// intermediate Halley / quartic steps of the correction
EXO_HD double approx_rcp(double y) {
#ifdef EXO_HOST_BUILD
  return 1.0 / y;
#else
  const double r = __builtin_amdgcn_rcp(y);
  return fma(fma(-y, r, 1.0), r, r);
#endif
}

// sin and cos for 0 <= x <= pi/2 (a few ulp beyond either end is fine): fold to
// [0, pi/4] with a two-term pi/2 and evaluate the Taylor polynomials to x^17 /
// x^16 (truncation < 2e-18).  No range reduction, no slow path: the eccentric
// half-anomaly never leaves this interval.
EXO_HD void sincos_halfpi(double x, double* s, double* c) {
  const bool hi = x > 0.78539816339744830962;
  const double y = hi ? (1.57079632679489655800 - x) + 6.123233995736766036e-17 : x;
  const double y2 = y * y;
  double ps = -2.8114572543455207632e-15;                   // -1/17!
  ps = fma(ps, y2, EXO_K(7.6471637318198164759e-13));       //  1/15!
  double pc = 4.7794773323873852974e-14;                    //  1/16!
  pc = fma(pc, y2, EXO_K2(-1.1470745597729724714e-11));     // -1/14!
  ps = fma(ps, y2, EXO_K(-1.6059043836821614599e-10));      // -1/13!
  pc = fma(pc, y2, EXO_K2(2.0876756987868098979e-09));      //  1/12!
  ps = fma(ps, y2, EXO_K(2.5052108385441718775e-08));       //  1/11!
  pc = fma(pc, y2, EXO_K2(-2.7557319223985890653e-07));     // -1/10!
  ps = fma(ps, y2, EXO_K(-2.7557319223985890653e-06));      // -1/9!
  pc = fma(pc, y2, EXO_K2(2.4801587301587301587e-05));      //  1/8!
  ps = fma(ps, y2, EXO_K(1.9841269841269841270e-04));       //  1/7!
  pc = fma(pc, y2, EXO_K2(-1.3888888888888888889e-03));     // -1/6!
  ps = fma(ps, y2, EXO_K(-8.3333333333333333333e-03));      // -1/5!
  pc = fma(pc, y2, EXO_K2(4.1666666666666666667e-02));      //  1/4!
  ps = fma(ps, y2, EXO_K(1.6666666666666666667e-01));       //  1/3!  (sign applied below)
  pc = fma(pc, y2, -0.5);
  ps = fma(-ps * y2, y, y);                                 // y - y^3/6 + ...  (ps holds +1/6 - y2/120 ...)
  pc = fma(pc, y2, 1.0);
  *s = hi ? pc : ps;
  *c = hi ? ps : pc;
}

// sin and cos of any argument, branch-free: quadrant by a five-term Cody-Waite reduction with
// 24-bit pieces of pi/2 (every product k * piece is exact for |k| < 2^29, i.e. |x| < 8e8; beyond
// that the rounding of x itself, half an ulp ~ 1e-7 rad, dwarfs anything a Payne-Hanek path could
// add), then the Taylor polynomials of sincos_halfpi on [-pi/4, pi/4].  ~45 instructions and no
// slow path: libm's sincos costs twice that and, inlined, the registers of its large-argument code.
EXO_HD void sincos_any(double x, double* s, double* c) {
  const double k = rint(x * EXO_K(0.6366197723675814));
  double y = fma(-k, EXO_K(1.570796251296997), x);
  y = fma(-k, EXO_K(7.549789415861596e-08), y);
  y = fma(-k, EXO_K(5.390302529957765e-15), y);
  y = fma(-k, EXO_K(3.282003415807913e-22), y);
  y = fma(-k, EXO_K(1.270655753080676e-29), y);
  const double y2 = y * y;
  double ps = -2.8114572543455207632e-15;                   // -1/17!
  ps = fma(ps, y2, EXO_K(7.6471637318198164759e-13));
  double pc = 4.7794773323873852974e-14;                    //  1/16!
  pc = fma(pc, y2, EXO_K2(-1.1470745597729724714e-11));
  ps = fma(ps, y2, EXO_K(-1.6059043836821614599e-10));
  pc = fma(pc, y2, EXO_K2(2.0876756987868098979e-09));
  ps = fma(ps, y2, EXO_K(2.5052108385441718775e-08));
  pc = fma(pc, y2, EXO_K2(-2.7557319223985890653e-07));
  ps = fma(ps, y2, EXO_K(-2.7557319223985890653e-06));
  pc = fma(pc, y2, EXO_K2(2.4801587301587301587e-05));
  ps = fma(ps, y2, EXO_K(1.9841269841269841270e-04));
  pc = fma(pc, y2, EXO_K2(-1.3888888888888888889e-03));
  ps = fma(ps, y2, EXO_K(-8.3333333333333333333e-03));
  pc = fma(pc, y2, EXO_K2(4.1666666666666666667e-02));
  ps = fma(ps, y2, EXO_K(1.6666666666666666667e-01));
  pc = fma(pc, y2, -0.5);
  ps = fma(-ps * y2, y, y);
  pc = fma(pc, y2, 1.0);
  // quadrant k mod 4: (s, c) = (ps, pc), (pc, -ps), (-ps, -pc), (-pc, ps)
  const long long q = (long long)k;
  const bool swap = q & 1, neg_s = q & 2, neg_c = (q + 1) & 2;
  const double ss = swap ? pc : ps, cc = swap ? ps : pc;
  *s = neg_s ? -ss : ss;
  *c = neg_c ? -cc : cc;
}

// ---------------------------------------------------------------------------
// Kepler solver.  Markley (1995) cubic starter + one fifth-order correction:
// fixed cost, so there is nothing to vote on.
//   * the starter is only good to ~4e-4 by construction, so it runs in fp32
//     (twice the fp64 rate, single-instruction rcp / sqrt); 1 - e is formed in
//     fp64 first so that e -> 1 survives the narrowing;
//   * the correction runs in fp64 on HALF angles, so that
//     1 - e cos E = X^2 + Y^2 carries no cancellation as e -> 1, with the
//     residual written as (1-e) E + e (E - sin E);
//   * sin / cos of the half angle never leave [0, pi/2]: a branch-free
//     polynomial, and the final (<= 4.4e-4) update is a Taylor rotation.
//
//   in : M (any real), e in [0,1), se = sqrt(1-e), pe = sqrt(1+e)
//   out: X = sqrt(1-e) cos(E/2),  Y = sqrt(1+e) sin(E/2)   (signed)
//        so that  den = X^2+Y^2 = 1 - e cos E,
//                 den cos f = X^2 - Y^2 (= cos E - e),
//                 den sin f = 2 X Y     (= sqrt(1-e^2) sin E)
//        sh, ch = sin(E/2), cos(E/2) (signed) for the reverse pass.
// ---------------------------------------------------------------------------
struct KeplerHalf {
  double X, Y, sh, ch;
};

EXO_HD KeplerHalf kepler_half(double M, double e, double se, double pe) {
  // two-term Cody-Waite reduction to [-pi, pi]
  const double k = rint(M * EXO_K(1.0 / kTwoPiHi));
  double Mr = fma(-k, EXO_K(kTwoPiHi), M);
  Mr = fma(-k, EXO_K(kTwoPiLo), Mr);
  const double sgn = (Mr < 0.0) ? -1.0 : 1.0;
  Mr = fabs(Mr);
  const double ome = 1.0 - e;
  double sh, ch;
  if (EXO_WAVE_ALL(e == 0.0)) {
    // circular orbit (the reference's ecc=None branch, keplerian.py:331-332)
    sincos_halfpi(0.5 * Mr, &sh, &ch);
  } else {
    // --- starter (Markley 1995 eqs. 15-20), fp32
    const float pif = 3.14159265358979f;
    const float Mf = (float)Mr, ef = (float)e, omf = (float)ome;
    const float alpha = fmaf(1.6f * pif * (pif - Mf), fast_rcpf(1.0f + ef), 3.0f * pif * pif) * (1.0f / (pif * pif - 6.0f));
    const float d = fmaf(alpha, ef, 3.0f * omf);
    const float q = fmaf(2.0f * alpha * d, omf, -Mf * Mf);
    const float r = fmaf(3.0f * alpha * d, d - omf, Mf * Mf) * Mf;
    float w = fast_cbrtf(fabsf(r) + fast_sqrtf(fmaxf(fmaf(q * q, q, r * r), 0.0f)));
    w = w * w;
    const float E1 = fmaf(2.0f * r * w, fast_rcpf(fmaf(w, w + q, q * q)), Mf) * fast_rcpf(d);
    const double E = (double)E1;
    // --- one fifth-order correction (eqs. 21-24), fp64
    sincos_halfpi(0.5 * E, &sh, &ch);
    const double sinE = 2.0 * sh * ch;
    const double f0 = fma(ome, E, fma(e, x_minus_sin(E, sinE), -Mr));
    const double f1 = fma(2.0 * e * sh, sh, ome);  // 1 - e cos E
    const double f2 = e * sinE;
    const double f3 = 1.0 - f1;
    // d3, d4 only steer the last denominator: ~1e-8 reciprocals are plenty there
    const double d3 = -f0 * approx_rcp(fma(-0.5 * f0 * f2, approx_rcp(f1), f1));
    const double d4 = -f0 * approx_rcp(fma(d3 * d3 * EXO_K(1.0 / 6.0), f3, fma(0.5 * d3, f2, f1)));
    const double d42 = d4 * d4;
    const double d5 = -fast_div(f0, fma(-d42 * d4 * EXO_K(1.0 / 24.0), f2, fma(d42 * EXO_K(1.0 / 6.0), f3, fma(0.5 * d4, f2, f1))));
    // |d5| <= 4.4e-4 over the whole (M,e) domain: rotate (sh,ch) by d5/2 with
    // a 4th-order Taylor rotation instead of a second sincos
    const double h = 0.5 * d5, h2 = h * h;
    const double sd = h * fma(-h2, EXO_K(1.0 / 6.0), 1.0);
    const double cd = fma(-h2, fma(-h2, EXO_K(1.0 / 24.0), 0.5), 1.0);
    const double sh2 = fma(sh, cd, ch * sd);
    ch = fma(ch, cd, -sh * sd);
    sh = sh2;
  }
  KeplerHalf o;
  o.sh = sgn * sh;
  o.ch = ch;
  o.X = se * ch;
  o.Y = pe * o.sh;
  return o;
}

// ---------------------------------------------------------------------------
// fp32 orbit position for the CONSERVATIVE cadence classifier of the scan
// kernel.  Only the phase M is fp64 (|M| reaches 10^4 radians: fp32 would lose
// the transit); after the reduction to [-pi, pi] everything is fp32 on hardware
// approximations: the Markley starter ALONE (|E1 - E| <= 4.4e-4 by construction,
// no correction step) and one v_sin / v_cos.
// Output: cx = cos E1 - e, sx = sqrt(1-e^2) sin E1 (the orbit position in units
// of -a), with |error| <= 8e-4 (checked on the GPU by
// tests/test_gpu_scan_filter.py); the caller widens its acceptance threshold by
// that bound times a/R, so a cadence is never wrongly discarded.  Nothing
// computed here reaches a result: accepted cadences are re-evaluated in fp64.
// ---------------------------------------------------------------------------
#ifndef EXO_HOST_BUILD
EXO_HD void orbit_pos_f32(double M, float ef, float omf, float sqf, float* cx, float* sx) {
  const double k = rint(M * (1.0 / kTwoPiHi));
  double Mr = fma(-k, kTwoPiHi, M);
  Mr = fma(-k, kTwoPiLo, Mr);
  const float Ms = (float)Mr;
  const float Mf = fabsf(Ms);
  float E = Mf;
  if (!EXO_WAVE_ALL(ef == 0.0f)) {
    const float pif = 3.14159265358979f;
    const float alpha = fmaf(1.6f * pif * (pif - Mf), fast_rcpf(1.0f + ef), 3.0f * pif * pif) * (1.0f / (pif * pif - 6.0f));
    const float d = fmaf(alpha, ef, 3.0f * omf);
    const float q = fmaf(2.0f * alpha * d, omf, -Mf * Mf);
    const float r = fmaf(3.0f * alpha * d, d - omf, Mf * Mf) * Mf;
    float w = fast_cbrtf(fabsf(r) + fast_sqrtf(fmaxf(fmaf(q * q, q, r * r), 0.0f)));
    w = w * w;
    E = fmaf(2.0f * r * w, fast_rcpf(fmaf(w, w + q, q * q)), Mf) * fast_rcpf(d);
  }
  const float x = E * 0.15915494309189535f;  // v_sin / v_cos take revolutions
  const float sE = __builtin_amdgcn_sinf(x), cE = __builtin_amdgcn_cosf(x);
  *cx = cE - ef;
  *sx = (Ms < 0.0f) ? -sqf * sE : sqf * sE;
}
#endif

// ---------------------------------------------------------------------------
// Three complete elliptic integrals on one shared AGM ladder (Bulirsch 1969
// `cel`; sharing the ladder between integrals follows the idea in Agol, Luger &
// Foreman-Mackey 2020):
//   B = int cos^2/Delta,  D = int sin^2/Delta          (p = 1)
//   P = int (aP cos^2 + bP sin^2)/(cos^2 + p sin^2)/Delta
// Delta = sqrt(cos^2 + kc^2 sin^2), all over [0, pi/2].  p > 0 required.
// Every sweep is an exact Landen transformation, so sweeping past convergence
// is harmless: the exit test is a wavefront vote (uniform branch).
// ---------------------------------------------------------------------------
struct Cel3 {
  double B, D, P;
};

EXO_HD Cel3 cel3(double kc, double p, double aP, double bP) {
  // floor: every caller's sin^2 coefficient vanishes with kc^2 (or the result
  // is multiplied by kc^2), so the floor costs O(1e-16 log) at most
  kc = fmax(fabs(kc), EXO_K(1e-8));
  double e = kc, em = 1.0;
  double aB = 1.0, bB = 0.0, aD = 0.0, bD = 1.0;
  double pp = fast_sqrt(p);
  bP = fast_div(bP, pp);
  // For the two p = 1 integrals Bulirsch's p-sequence IS the arithmetic-mean sequence em
  // (p_0 = em_0 = 1 and e = kc em at the top of every sweep, so e / p = kc and p + e / p = em + kc):
  // they need no quotient of their own, and the reciprocals 1/em, 1/pp come from one
  // reciprocal of the product.
#pragma unroll 1
  for (int it = 0; it < 12; ++it) {
    const double rj = fast_rcp(em * pp);
    const double iem = rj * pp, ipp = rj * em;
    const double gP = e * ipp;
    double f = aB;
    aB = fma(bB, iem, aB);
    bB = 2.0 * fma(f, kc, bB);
    f = aD;
    aD = fma(bD, iem, aD);
    bD = 2.0 * fma(f, kc, bD);
    f = aP;
    aP = fma(bP, ipp, aP);
    bP = 2.0 * fma(f, gP, bP);
    pp = gP + pp;
    const double g = em;
    em += kc;
    if (EXO_WAVE_ALL(!(fabs(g - kc) > g * EXO_K(1.0e-8)))) break;
    kc = 2.0 * fast_sqrt(e);
    e = kc * em;
  }
  Cel3 o;
  // B, D: (pi/2) (a em + b) / (em (em + em));  P: (pi/2) (aP em + bP) / (em (em + pp))
  const double rj = fast_rcp(em * em * (em + pp));
  const double q1 = EXO_K(0.5 * kHalfPi) * rj * (em + pp);
  o.B = q1 * fma(aB, em, bB);
  o.D = q1 * fma(aD, em, bD);
  o.P = EXO_K(kHalfPi) * fma(aP, em, bP) * (rj * em);
  return o;
}

// C4 = int_0^{pi/2} cos^4/sqrt(1 - k2 sin^2): Maclaurin series in k2 (k2 < 0.1)
EXO_HD double int_cos4_series(double k2) {
  // c_j = (2j-1)!!/(2j)!! * (2/pi) int sin^{2j} cos^4, j = 17 .. 0 (Horner; the coefficients from scalar registers)
  double s = 4.04623031491638797e-05;
  s = fma(s, k2, EXO_K(4.80048628730208747e-05));
  s = fma(s, k2, EXO_K(5.75458918103226302e-05));
  s = fma(s, k2, EXO_K(6.97940661670976015e-05));
  s = fma(s, k2, EXO_K(8.57825559474889587e-05));
  s = fma(s, k2, EXO_K(1.07056629822466221e-04));
  s = fma(s, k2, EXO_K(1.35996323706422118e-04));
  s = fma(s, k2, EXO_K(1.76394324626016896e-04));
  s = fma(s, k2, EXO_K(2.34540930250659585e-04));
  s = fma(s, k2, EXO_K(3.21377883665263653e-04));
  s = fma(s, k2, EXO_K(4.57070767879486084e-04));
  s = fma(s, k2, EXO_K(6.81549310684204102e-04));
  s = fma(s, k2, EXO_K(1.08146667480468750e-03));
  s = fma(s, k2, EXO_K(1.86920166015625000e-03));
  s = fma(s, k2, EXO_K(3.66210937500000000e-03));
  s = fma(s, k2, EXO_K(8.78906250000000000e-03));
  s = fma(s, k2, EXO_K(3.12500000000000000e-02));
  s = fma(s, k2, EXO_K(3.75000000000000000e-01));
  return EXO_K(kHalfPi) * s;
}

// 8 (k - sin k) - (2k - sin 2k) = 32 int_0^{k/2} sin^4, series for k < 0.4
EXO_HD double i4_series(double k) {
  const double k2 = k * k;
  double s = 524280.0 / 121645100408832000.0;
  s = fma(-k2, s, EXO_K(131064.0 / 355687428096000.0));
  s = fma(-k2, s, EXO_K(32760.0 / 1307674368000.0));
  s = fma(-k2, s, EXO_K(8184.0 / 6227020800.0));
  s = fma(-k2, s, EXO_K(2040.0 / 39916800));
  s = fma(-k2, s, EXO_K(504.0 / 362880));
  s = fma(-k2, s, EXO_K(120.0 / 5040));
  s = fma(-k2, s, EXO_K(24.0 / 120));
  return s * k2 * k2 * k;
}

// atan2(y, x0) and atan2(y, x1) for a common y >= 0 (results in [0, pi]): the two arc half-angles of the solution vector.
// One division each (min / max, so |t| <= 1), atan t = t + t^3 q(t^2) with q of degree 19 in t^2 (interpolated at Chebyshev
// nodes in 60-digit arithmetic: relative error 7.7e-17 with the rounded coefficients, tools/atan_fit.py), the two chains side
// by side.  libm's atan2 costs ~95 vector instructions here (19 coefficients as 38 register moves, an IEEE division with its
// scale / fixup sequence); this is ~60 (~36 with the coefficients in scalar registers, exo_ops.hip).  In the sweep the ~70
// instructions saved per wave bought nothing measurable (it is chain-bound, EXO_K above); kept because it is shorter, has no
// slow path, and quad_solution_vector as an Op gains.
EXO_HD void atan2_pos_pair(double y, double x0, double x1, double* r0, double* r1) {
#ifdef EXO_HOST_BUILD
  *r0 = atan2(y, x0);
  *r1 = atan2(y, x1);
#else
  const double ax0 = fabs(x0), ax1 = fabs(x1);
  const double mx0 = fmax(ax0, y), mx1 = fmax(ax1, y);
  const double t0 = (mx0 > 0.0) ? fast_div(fmin(ax0, y), mx0) : 0.0;
  const double t1 = (mx1 > 0.0) ? fast_div(fmin(ax1, y), mx1) : 0.0;
  const double a0 = t0 * t0, a1 = t1 * t1;
  double p0 = 1.806195461861215e-05, p1 = p0;
  p0 = fma(p0, a0, EXO_K(-0.00019996189377901382));
  p1 = fma(p1, a1, EXO_K2(-0.00019996189377901382));
  p0 = fma(p0, a0, EXO_K(0.0010496035084968515));
  p1 = fma(p1, a1, EXO_K2(0.0010496035084968515));
  p0 = fma(p0, a0, EXO_K(-0.0034958859739163094));
  p1 = fma(p1, a1, EXO_K2(-0.0034958859739163094));
  p0 = fma(p0, a0, EXO_K(0.008368931178450162));
  p1 = fma(p1, a1, EXO_K2(0.008368931178450162));
  p0 = fma(p0, a0, EXO_K(-0.015535152475414177));
  p1 = fma(p1, a1, EXO_K2(-0.015535152475414177));
  p0 = fma(p0, a0, EXO_K(0.023696731580048622));
  p1 = fma(p1, a1, EXO_K2(0.023696731580048622));
  p0 = fma(p0, a0, EXO_K(-0.031277189066996385));
  p1 = fma(p1, a1, EXO_K2(-0.031277189066996385));
  p0 = fma(p0, a0, EXO_K(0.03749486812535247));
  p1 = fma(p1, a1, EXO_K2(0.03749486812535247));
  p0 = fma(p0, a0, EXO_K(-0.04260356632601652));
  p1 = fma(p1, a1, EXO_K2(-0.04260356632601652));
  p0 = fma(p0, a0, EXO_K(0.04737749579527779));
  p1 = fma(p1, a1, EXO_K2(0.04737749579527779));
  p0 = fma(p0, a0, EXO_K(-0.052579733342841106));
  p1 = fma(p1, a1, EXO_K2(-0.052579733342841106));
  p0 = fma(p0, a0, EXO_K(0.05881506877793656));
  p1 = fma(p1, a1, EXO_K2(0.05881506877793656));
  p0 = fma(p0, a0, EXO_K(-0.06666564699289104));
  p1 = fma(p1, a1, EXO_K2(-0.06666564699289104));
  p0 = fma(p0, a0, EXO_K(0.07692298971033217));
  p1 = fma(p1, a1, EXO_K2(0.07692298971033217));
  p0 = fma(p0, a0, EXO_K(-0.09090908590891934));
  p1 = fma(p1, a1, EXO_K2(-0.09090908590891934));
  p0 = fma(p0, a0, EXO_K(0.11111111093490827));
  p1 = fma(p1, a1, EXO_K2(0.11111111093490827));
  p0 = fma(p0, a0, EXO_K(-0.14285714285384132));
  p1 = fma(p1, a1, EXO_K2(-0.14285714285384132));
  p0 = fma(p0, a0, EXO_K(0.1999999999999753));
  p1 = fma(p1, a1, EXO_K2(0.1999999999999753));
  p0 = fma(p0, a0, EXO_K(-0.3333333333333333));
  p1 = fma(p1, a1, EXO_K2(-0.3333333333333333));
  p0 = fma(t0 * a0, p0, t0);
  p1 = fma(t1 * a1, p1, t1);
  const double h0 = (y > ax0) ? kHalfPi - p0 : p0;
  const double h1 = (y > ax1) ? kHalfPi - p1 : p1;
  *r0 = (x0 < 0.0) ? kPi - h0 : h0;
  *r1 = (x1 < 0.0) ? kPi - h1 : h1;
#endif
}

// ---------------------------------------------------------------------------
// Solution vector s = (s0, s1, s2) = int_{visible disk} (1, mu, 4 mu^2 - 2) dA
// for an occultor of radius r at separation b >= 0, and (optionally) ds/db,
// ds/dr.  See docs/DESIGN_r1_r4.md section 3.2 for the derivation:
//   s0 = pi - (two circular segments)
//   s2 = -int_arc (1-rho^2) r (r + b cos phi) dphi      (field rho(1-rho^2) phi^)
//   s1 = 2pi/3 (1 - Theta(r-b)) + 1/3 int_arc (1-rho^2)^{3/2} dtheta
//   ds_n/dr = -r int_arc g_n dphi,  ds_n/db = -r int_arc g_n cos phi dphi
// WITH_GRAD is a compile-time switch so the forward kernel carries no
// derivative arithmetic.
// ---------------------------------------------------------------------------
struct SV {
  double s0, s1, s2;
  double db0, db1, db2;
  double dr0, dr1, dr2;
};

template <bool WITH_GRAD>
EXO_HD void quad_sv(double b, double r, SV& o) {
  o.s0 = kPi; o.s1 = kTwoThirdsPi; o.s2 = 0.0;
  o.db0 = o.db1 = o.db2 = 0.0;
  o.dr0 = o.dr1 = o.dr2 = 0.0;
  if (b != b || r != r) {
    const double nan = b + r;
    o.s0 = o.s1 = o.s2 = nan;
    o.db0 = o.db1 = o.db2 = o.dr0 = o.dr1 = o.dr2 = nan;
    return;
  }
  const bool none = (r <= 0.0) || (b >= 1.0 + r);
  const bool full = !none && (r >= 1.0 + b);
  if (full) { o.s0 = 0.0; o.s1 = 0.0; o.s2 = 0.0; }
  const bool act = !(none || full);
  if (!EXO_WAVE_ANY(act)) return;
  if (!act) { b = 0.5; r = 0.1; }  // benign stand-in so idle lanes stay finite

  const double r2 = r * r, b2 = b * b;
  const bool inside = (b + r <= 1.0);
  // 1-(b-r)^2 and (b+r)^2-1, factored and ordered so that the leading
  // subtraction is exact (Sterbenz) when the larger radius is near 1
  const double x = fmax(b, r), y = fmin(b, r);
  const double A = ((1.0 - x) + y) * (1.0 + (x - y));
  const double Bm = ((x - 1.0) + y) * ((x + y) + 1.0);
  double isqA;
  const double sqA = fast_sqrt_rs(A, &isqA);  // A > 0 on every active lane
  const double br = b * r;
  const double rmb = r - b;

  // ---- arc geometry (partial overlap only)
  double k0 = kPi, u0 = kHalfPi, I2 = 0.25 * kPi, I4 = 0.1875 * kPi, sink0 = 0.0;
  double seg = 0.0;  // lens area (two circular segments)
  if (EXO_WAVE_ANY(!inside)) {
    const double kite = fast_sqrt(fmax(0.0, A * Bm));       // 2 b r sin k0 = 2 b sin k1
    const double c0n = b2 + (r - 1.0) * (r + 1.0);      // 2 b r cos k0
    const double c1n = (1.0 - r) * (1.0 + r) + b2;      // 2 b cos k1
    double pk0, pk1;
    atan2_pos_pair(kite, c0n, c1n, &pk0, &pk1);
    const double i2br = fast_div(0.5, br), i2b = fast_div(0.5, b);
    const double s0k = kite * i2br, c0k = c0n * i2br;
    const double s1k = kite * i2b, c1k = c1n * i2b;
    const double xms_k0 = x_minus_sin(pk0, s0k);
    const double xms_2k0 = x_minus_sin(2.0 * pk0, 2.0 * s0k * c0k);
    const double xms_2k1 = x_minus_sin(2.0 * pk1, 2.0 * s1k * c1k);
    const double pI4 = (pk0 < 0.4) ? i4_series(pk0) : (8.0 * xms_k0 - xms_2k0);
    if (!inside) {
      k0 = pk0; u0 = 0.5 * pk0; sink0 = s0k;
      I2 = 0.25 * xms_k0;
      I4 = (1.0 / 32.0) * pI4;
      seg = 0.5 * (r2 * xms_2k0 + xms_2k1);
    }
  }
  if (inside) seg = EXO_K(kPi) * r2;
  const double q = 1.0 - 2.0 * rmb * rmb;
  const double s0 = EXO_K(kPi) - seg;
  const double s2 = -4.0 * r * (A * rmb * u0 + (2.0 * b * A - 4.0 * br * rmb) * I2 - 8.0 * b2 * r * I4);

  // ---- s1: one shared AGM ladder serves both geometries
  //   inside : modulus m = 4br/A,  kc^2 = -Bm/A,  P = cel(kc, ((b+r)/(b-r))^2, A, -Bm)
  //   partial: modulus k2 = A/4br, kc^2 = 1-k2,   P = cel(kc, 1/(b-r)^2, 1, 0)
  const bool same = (b == r);
  const double rmb_s = same ? 1.0 : rmb;
  const double irmb = fast_div(1.0, rmb_s);
  const double iA = isqA * isqA;
  const double m_in = 4.0 * br * iA;
  const double k2 = inside ? 0.0 : fmin(fast_div(A, 4.0 * br), 1.0);
  const double kc2 = inside ? fmax(-Bm * iA, 0.0) : fmax(1.0 - k2, 0.0);
  const double kc = fast_sqrt(kc2);
  const double bpr = b + r;
  const double pP = inside ? (bpr * irmb) * (bpr * irmb) : irmb * irmb;
  const Cel3 c3 = cel3(kc, pP, inside ? A : 1.0, inside ? -Bm : 0.0);
  const double Ek = fma(kc2, c3.D, c3.B);   // E
  const double Kk = c3.B + c3.D;            // K
  const double theta = (r > b) ? 1.0 : (same ? 0.5 : 0.0);
  double J;
  double pref = 0.0, C2 = 0.0, C4 = 0.0;
  if (inside) {
    const double t3 = (2.0 * (2.0 - m_in) * Ek - kc2 * Kk) * EXO_K(1.0 / 3.0);  // int Delta^3
    J = (sqA * EXO_K(2.0 / 3.0)) * (A * t3 - (r2 - b2) * Ek);
    if (!same) J += EXO_K(2.0 / 3.0) * bpr * irmb * isqA * c3.P;
  } else {
    C2 = c3.B;
    // closed form loses eps/k2^2; switch to the series where that matters
    if (EXO_WAVE_ANY(k2 < 0.1)) {
      const double ser = int_cos4_series(k2);
      C4 = (k2 < 0.1) ? ser : fast_div((3.0 * k2 - 1.0) * c3.B + kc2 * c3.D, 3.0 * k2);
    } else {
      C4 = fast_div((3.0 * k2 - 1.0) * c3.B + kc2 * c3.D, 3.0 * k2);
    }
    pref = 4.0 * sqA * fast_sqrt(k2);
    J = (pref * EXO_K(1.0 / 6.0)) * (A * C4 - (r2 - b2) * C2);
    if (!same) J += (bpr * irmb * EXO_K(1.0 / 6.0)) * pref * c3.P;
  }
  const double s1 = fma(EXO_K(kTwoThirdsPi), 1.0 - theta, J);

  if (act) { o.s0 = s0; o.s1 = s1; o.s2 = s2; }

  if (WITH_GRAD) {
    const double dr0 = -2.0 * r * k0;
    const double db0 = 2.0 * r * sink0;
    const double dr2 = -r * (4.0 * k0 * q - 64.0 * br * I2);
    const double db2 = -4.0 * r * (-2.0 * q * u0 + (4.0 * q + 16.0 * br) * I2 - 32.0 * br * I4);
    double dr1, db1;
    if (inside) {
      dr1 = -4.0 * r * sqA * Ek;
      db1 = (4.0 / 3.0) * r * sqA * fma(-kc2, c3.D, c3.B);  // int cos(2u) Delta
    } else {
      dr1 = -r * pref * C2;
      db1 = r * pref * (C2 * (1.0 - 2.0 * k2) + 2.0 * k2 * C4);
    }
    if (act) {
      o.db0 = db0; o.db1 = db1; o.db2 = db2;
      o.dr0 = dr0; o.dr1 = dr1; o.dr2 = dr2;
    }
  }
}


// Zeros before an accumulation / a scan's seed: a KERNEL, never hipMemsetAsync.  Inside a captured step a memset becomes a
// memset node of the hipGraph, and on ROCm 7.2 (gfx950) such a node stops filling with its value once the process has issued
// enough EAGER device-to-device copies between replays (the runtime's own blit kernels): from then on, every replay, it writes
// garbage -- pointer-like bits -- instead of zeros (tools/graph_node_order.py reproduces it with torch ops alone: correct for
// ~4700 replays, wrong ever after; memcpy nodes and kernel nodes are not affected).  Round 5: the adjoint scan of the celerite
// reverse pass was seeded by such a node; in a NUTS run (eager copies between the leaves) the coefficient gradients of every other
// draw became garbage after ~1000 leaves and the chains' step sizes collapsed (tests/test_gpu_inject_recover.py).
#ifndef EXO_HOST_BUILD
static __global__ __launch_bounds__(256) void zero_fill_kernel(double* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0;
}
inline bool zero_fill_async(double* p, int64_t n, hipStream_t st) {
  if (n <= 0) return true;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
  return hipGetLastError() == hipSuccess;
}
#endif

}  // namespace exo
