// exo_celerite_core.hpp -- the one-lane-per-(draw, chunk) half of the celerite path: term
// coefficients, the chunk workspace, the Kalman filtering elements and the scans over them, and
// the recurrences inside a chunk with a CHECKPOINTED factorisation (forward saves the recurrence
// state every kCkptB cadences; the reverse pass recomputes the cadences of a block from its
// checkpoint in registers).  Nothing here exchanges data between lanes, so the same code compiles
// for the host (g++, EXO_HOST_BUILD: tests/gp_host_harness.cpp runs the whole time-parallel
// pipeline on the CPU against the oracle) and for gfx950, where a lane is a (draw, chunk) and the
// kernels of exo_celerite.hip are thin wrappers.  Algorithm: exo_celerite.hip header comment and
// docs/DESIGN_r1_r4.md 3.4 / 3.5 (celerite2 is a dependency of the reference, /root/reference/setup.py:36;
// the recurrences are the published ones).
#pragma once
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "exo_math.hpp"


#ifdef EXO_HOST_BUILD
#define EXO_HDH inline
#define EXO_RESTRICT __restrict__
#else
#define EXO_HDH __host__ __device__ inline
#define EXO_RESTRICT __restrict__
#endif

namespace gp {

constexpr double kHalfLog2Pi = 0.91893853320467274178;
constexpr int kCkptB = 4;   // cadences per block of the one-lane chunk kernels (series rows move four at a time)
// cadences per CHECKPOINT: the whole block for J <= 2; half a block for wider states -- the reverse kernel keeps the
// recomputed states of a span in registers (J (J + 1) / 2 + 2 J + 2 doubles each), and four of them do not fit at J > 2
// the polish passes (chunk1_fwd_lane): a laboratory result so far -- compiled into the host harness (tests/gp_host_harness.cpp,
// tools/gp_host_lab.py), not into the device kernels: one Jacobi sweep takes the MEDIAN error of ill-conditioned draws down
// 10-100 x but the worst kernels only 3-10 x per four sweeps (docs/DESIGN_r1_r4.md section 3.5)
// the checkpoints of the one-lane forward kernel -- written once, read once by the reverse kernel a millisecond later -- by
// non-temporal stores: C3 3.314 -> 3.260 ms, C5 1.827 -> 1.798 (same box, alternating)
#ifndef EXO_CKPT_NT_STORE
#define EXO_CKPT_NT_STORE 1
#endif
#ifndef EXO_CKPT_NT_LOAD
#define EXO_CKPT_NT_LOAD 0
#endif
#ifndef EXO_GP_POLISH
#define EXO_GP_POLISH 0
#endif
#ifndef EXO_SPAN2_MIN_J
#define EXO_SPAN2_MIN_J 3
#endif
EXO_HDH constexpr int ckpt_span(int J) { return J >= EXO_SPAN2_MIN_J ? 2 : 4; }

// Term coefficients of a batch of draws, celerite2's Term.get_coefficients() form:
//   real  [n_draw][n_real][2]     (a, c)
//   cplx  [n_draw][n_complex][4]  (a, b, c, d)            kind 0 (or kind == nullptr)
//                                 (a1, c1, a2, c2)         kind 1: the slot holds TWO REAL terms
//   kind  [n_draw][n_complex] or nullptr
// A pair slot occupies two state indices either way, so a batch may mix both kinds draw by draw
// (an SHO term is one complex term for Q >= 1/2 and two real ones for Q < 1/2).
struct Coefs {
  const double* real;
  const double* cplx;
  const int32_t* kind;
  int n_real, n_complex;
  // the time the phases d (t - *origin) of the complex terms are counted from (nullptr: 0).  The likelihood sees phase
  // differences only, so any origin gives the same numbers in exact arithmetic; in double precision d t at t ~ 2 457 000 d
  // has lost nine digits before the cosine is taken (round 4: the entry points point this at the series' first time stamp;
  // every kernel of a call reads the same one, the states they exchange live in the same rotating frame)
  const double* origin = nullptr;
  // NULL, or [n_draw]: the kernels' draw d is row row[d] of every per-draw array of the CALL -- coefficients, pair kinds, per-draw
  // diag, loglike, gloglike and the cotangents written back (round 5: exo_sparse_model.row_of_draw; the sparse entries hand the
  // draws over sorted by transit timing without the caller permuting anything).  State and workspace are indexed by d itself.
  const int32_t* row = nullptr;
  EXO_HDH int J() const { return n_real + 2 * n_complex; }
  EXO_HDH int64_t at(int64_t draw) const { return row ? (int64_t)row[draw] : draw; }
};

// per-state-index view: state index j of a real term (a, c) or of a complex pair (a, b, c, d);
// `odd` marks the second index of a complex pair; `slot` = where the cotangents of a real term go
// (-1: its own row of gcoef_real; >= 0: doubles slot, slot + 1 of the draw's pair-slot block)
struct LaneCoef {
  double a, b, c, d;
  double t0;      // Coefs::origin's value
  bool real, odd, live;
  int slot;
};

EXO_HD LaneCoef lane_coef(const Coefs& co, int64_t draw, int j, int J) {
  LaneCoef k;
  k.live = j < J;
  k.real = j < co.n_real;
  k.odd = false;
  k.slot = -1;
  k.a = k.b = k.c = k.d = 0.0;
  k.t0 = co.origin ? *co.origin : 0.0;
  if (!k.live) return k;
  if (k.real) {
    const double* p = co.real + (co.at(draw) * co.n_real + j) * 2;
    k.a = p[0]; k.c = p[1];
  } else {
    const int jc = (j - co.n_real) >> 1, second = (j - co.n_real) & 1;
    const double* p = co.cplx + (co.at(draw) * co.n_complex + jc) * 4;
    // the slot's four doubles are loaded whatever its kind and SELECTED afterwards: the kind varies
    // from lane to lane, and loads through a pointer chosen inside that divergent branch were
    // miscompiled by hipcc 7.2 for gfx950 (address register left undefined for one side)
    const double p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    const int kd = co.kind ? co.kind[co.at(draw) * co.n_complex + jc] : 0;
    const bool two_real = kd != 0;
    k.real = two_real;
    k.a = two_real ? (second ? p2 : p0) : p0;
    k.b = two_real ? 0.0 : p1;
    k.c = two_real ? (second ? p3 : p1) : p2;
    k.d = two_real ? 0.0 : p3;
    k.odd = !two_real && second != 0;
    k.slot = two_real ? jc * 4 + 2 * second : -1;
  }
  return k;
}

// U_j, V_j of SURVEY Appendix B at time t for one state index
EXO_HD void lane_uv(const LaneCoef& k, double t, double* U, double* V, double* cs, double* sn) {
  if (k.real || !k.live) {
    *U = k.live ? k.a : 0.0;
    *V = k.live ? 1.0 : 0.0;
    *cs = 1.0; *sn = 0.0;
    return;
  }
  double s, c;
  exo::sincos_any(k.d * (t - k.t0), &s, &c);   // branch-free, no large-argument path: half the instructions and registers of libm's
  *cs = c; *sn = s;
  *U = k.odd ? (k.a * s - k.b * c) : (k.a * c + k.b * s);
  *V = k.odd ? s : c;
}

// The series the likelihood is evaluated on: y[draw][n] as given, or -- obs != nullptr -- the
// residual obs[n] - y[draw][n] of a per-draw model against one observed series, formed on the
// fly (the subtraction, and the sign flip of its cotangent, never cross HBM as arrays).
// SPARSE model (round 5; obs != nullptr, sp.nseg != nullptr): the per-draw model is zero outside a few SEGMENTS of
// cadences -- a transit light curve: ~3 % of the series -- and only the values inside them exist, in the order of the
// cadences (exo_sparse_model, include/exoplanet_amd.h: the sparse output of the light-curve sweep, or its merged form).
// Segment k of draw d covers cadences [lo, hi) = seg[d * seg_row + k * seg_step + {0, hi_at}], ascending and disjoint; the
// value of cadence n in it is vals[d * val_row + off[d * off_row + k] + (n - lo)].  The dense (draw, cadence) model, 97 % zeros,
// and its cotangent never cross HBM: the cotangent comes back in the same value layout.
struct SparseSegs {
  const int32_t* nseg = nullptr;
  const int32_t* seg = nullptr;
  const int32_t* off = nullptr;
  int64_t seg_row = 0, off_row = 0, val_row = 0;
  int32_t seg_step = 0, hi_at = 0;
  // row_of_draw (optional): draw d of the CALL is row row_of_draw[d] of the model's tables and value array -- a caller that
  // hands the draws over in another order than the light-curve sweep produced them in (sorted by transit time, so that the
  // draws of a wave are inside their transits together: exo_sparse_model) permutes the small per-draw inputs and leaves the
  // model where it is
  const int32_t* row_of_draw = nullptr;
  EXO_HD int64_t row(int64_t draw) const { return row_of_draw ? (int64_t)row_of_draw[draw] : draw; }
};
struct Series {
  const double* y;
  const double* obs;
  int64_t cm;   // 0: y is [draw][cadence]; else CADENCE-MAJOR, [cadence][cm] with cm = n_draw (the reverse pass writes the
                // cotangent of the series in the same layout)
  SparseSegs sp{};   // sp.nseg != nullptr: y is the value array of a sparse model
};
// A lane owns one ROW of a [draw][cadence] array: consecutive lanes are n cadences apart, so every
// load instruction of a wave touches 64 different cache lines, and cadence-by-cadence 8-byte
// accesses fetch each line eight times over (the lines of 16 waves do not stay in a 32 KB L1:
// measured 1.6 ms for the forward chunk kernel of C3, 9.8 GB through L2 for 1.2 GB of data).  The
// chunk kernels therefore move a row kCkptB = 4 cadences at a time, as two 16-byte accesses per
// lane whenever the row is 16-byte aligned there.
// CADENCE-MAJOR arrays ([cadence][draw]: what the light-curve sweep writes under EXO_FLAG_CADENCE_MAJOR) need none of
// that: the lanes of a wave are consecutive draws, every access of a wave is 512 contiguous bytes (C3: the forward
// chunk kernel 0.86 -> 0.64 ms, the reverse one 2.02 -> 1.54 ms against the 16-byte row accesses).
struct alignas(16) D2 {
  double x, y;
};
EXO_HD void row_load4(const double* EXO_RESTRICT row, int64_t i, int len, double* v) {
  if (len == 4 && (reinterpret_cast<uintptr_t>(row + i) & 15) == 0) {
    const D2 a = *reinterpret_cast<const D2*>(row + i), b = *reinterpret_cast<const D2*>(row + i + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = q < len ? row[i + q] : 0.0;
  }
}
EXO_HD void row_store4(double* EXO_RESTRICT row, int64_t i, int len, const double* v) {
  if (len == 4 && (reinterpret_cast<uintptr_t>(row + i) & 15) == 0) {
    *reinterpret_cast<D2*>(row + i) = D2{v[0], v[1]};
    *reinterpret_cast<D2*>(row + i + 2) = D2{v[2], v[3]};
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < len) row[i + q] = v[q];
  }
}

// A lane's place among its draw's segments.  The kernels walk a chunk one way -- the forward recurrences upwards, the reverse
// ones downwards -- so the cursor is one segment and a direction:
//   ASC   k = the first segment that ends beyond the cadence last asked for (k = nseg: none left, lo = hi = "never")
//   DESC  k = the last segment that starts at or before it                (k = -1: none left)
// [lo, hi) its cadences, voff + n the position of cadence n's value.  The first access finds its segment by bisection; after
// that the cursor moves a segment at a time -- three loads, once per transit -- and a block of four cadences outside every
// segment (97 % of them) costs one comparison.
template <bool ASC>
struct SegCursor {
  static constexpr int32_t kFar = 0x7fffffff;
  int32_t k = ASC ? -1 : kFar, lo = 0, hi = 0, voff = 0;
  EXO_HD void fetch(const SparseSegs& sp, int64_t draw) {
    const int32_t nseg = sp.nseg[draw];
    if (k >= 0 && k < nseg) {
      const int32_t* EXO_RESTRICT s = sp.seg + draw * sp.seg_row + (int64_t)k * sp.seg_step;
      lo = s[0];
      hi = s[sp.hi_at];
      voff = sp.off[draw * sp.off_row + k] - lo;
    } else {
      lo = hi = ASC ? kFar : -kFar;
      voff = 0;
    }
  }
  EXO_HD void first(const SparseSegs& sp, int64_t draw, int32_t i) {
    const int32_t* EXO_RESTRICT s = sp.seg + draw * sp.seg_row + (ASC ? sp.hi_at : 0);
    int32_t a = 0, b = sp.nseg[draw];   // ASC: the first segment with hi > i;  DESC: the first with lo > i, minus one
    while (a < b) {
      const int32_t m = (a + b) >> 1;
      if (s[(int64_t)m * sp.seg_step] <= i) a = m + 1; else b = m;
    }
    k = ASC ? a : a - 1;
    fetch(sp, draw);
  }
  // Position the cursor for the block of cadences [b, b + 4); false: no cadence of the block is in any segment (one
  // comparison in the usual case).  Blocks come in the cursor's direction.
  EXO_HD bool enter(const SparseSegs& sp, int64_t draw, int64_t b64) {
    const int32_t b = (int32_t)b64;
    if (ASC) {
      if (k < 0) first(sp, draw, b);
      while (b >= hi) { ++k; fetch(sp, draw); }
      return b + 4 > lo;
    }
    if (k == kFar) first(sp, draw, b + 3);
    while (b + 3 < lo) { --k; fetch(sp, draw); }
    return b < hi;
  }
  // position of cadence n's value, or -1 -- after enter(), for the block's cadences in the cursor's direction (the loop only
  // turns when two segments share the block)
  EXO_HD int32_t idx(const SparseSegs& sp, int64_t draw, int32_t n) {
    if (ASC) {
      while (n >= hi) { ++k; fetch(sp, draw); }
      return n >= lo ? voff + n : -1;
    }
    while (n < lo) { --k; fetch(sp, draw); }
    return n < hi ? voff + n : -1;
  }
  // a single cadence, any time (cadences still in the cursor's direction)
  EXO_HD int32_t at(const SparseSegs& sp, int64_t draw, int64_t i) {
    if (ASC ? (k < 0) : (k == kFar)) first(sp, draw, (int32_t)i);
    return idx(sp, draw, (int32_t)i);
  }
};

// one draw's series: element i at y[i * stride] (stride 1: a row; n_draw: a column of a cadence-major array); or the
// values of a sparse model (SparseSegs), read in ascending (ASC) or descending order of the cadences
// SP: 1 the series is known to be sparse at compile time, 0 known not to be (the hot one-lane kernels are compiled both ways:
// a cursor's registers in the dense kernels cost their J = 2 forms scratch), -1 decided by rs.sp.nseg at run time
template <bool ASC, int SP = -1>
struct SeriesRowT {
  const double* EXO_RESTRICT y;
  const double* EXO_RESTRICT obs;
  int64_t stride;
  const SparseSegs& sp;
  int64_t draw;
  mutable SegCursor<ASC> cur;
  EXO_HD SeriesRowT(const Series& rs, int64_t draw_, int64_t n)
      : y(rs.y + (rs.sp.nseg ? rs.sp.row(draw_) * rs.sp.val_row : (rs.cm ? draw_ : draw_ * n))), obs(rs.obs),
        stride(rs.cm ? rs.cm : 1), sp(rs.sp), draw(rs.sp.nseg ? rs.sp.row(draw_) : draw_) {}   // (draw: the model's row)
  EXO_HD bool sparse() const { return SP == 1 || (SP == -1 && sp.nseg != nullptr); }
  EXO_HD double model_at(int64_t i) const {   // sparse: the model at cadence i
    const int32_t v = cur.at(sp, draw, i);
    return v >= 0 ? y[v] : 0.0;
  }
  EXO_HD double operator[](int64_t i) const {
    if (sparse()) return obs[i] - model_at(i);
    return obs ? obs[i] - y[i * stride] : y[i * stride];
  }
  // cadences i .. i + 3 (the first `len` of them exist)
  EXO_HD void load4(int64_t i, int len, double* v) const {
    if (sparse()) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = 0.0;
      if (cur.enter(sp, draw, i)) {   // (the usual case is the other one: no load)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int q = ASC ? qq : 3 - qq;
          const int32_t at = cur.idx(sp, draw, (int32_t)i + q);
          if (q < len && at >= 0) v[q] = y[at];
        }
      }
    } else if (stride == 1) {
      row_load4(y, i, len, v);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = q < len ? y[(i + q) * stride] : 0.0;
    }
    if (obs) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (q < len ? obs[i + q] : 0.0) - v[q];
    }
  }
};
using SeriesRow = SeriesRowT<true>;
using SeriesRowDesc = SeriesRowT<false>;
// one draw's cotangent of the series, in the layout of the series (sparse: of the values -- cadences outside the segments
// have no value to receive one), written in DESCENDING order of the cadences (every reverse recurrence does)
template <int SP = -1>
struct GradRowT {
  double* EXO_RESTRICT g;
  int64_t stride;
  const SparseSegs& sp;
  int64_t draw;
  SegCursor<false> cur;
  EXO_HD bool sparse() const { return SP == 1 || (SP == -1 && sp.nseg != nullptr); }
  EXO_HD GradRowT(double* gresid, const Series& rs, int64_t draw_, int64_t n)
      : g(gresid + (rs.sp.nseg ? rs.sp.row(draw_) * rs.sp.val_row : (rs.cm ? draw_ : draw_ * n))), stride(rs.cm ? rs.cm : 1),
        sp(rs.sp), draw(rs.sp.nseg ? rs.sp.row(draw_) : draw_) {}
  EXO_HD void store(int64_t i, double v) {
    if (sparse()) {
      const int32_t at = cur.at(sp, draw, i);
      if (at >= 0) g[at] = v;
    } else {
      g[i * stride] = v;
    }
  }
  EXO_HD void store4(int64_t i, int len, const double* v) {
    if (sparse()) {
      if (!cur.enter(sp, draw, i)) return;
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        const int32_t at = cur.idx(sp, draw, (int32_t)i + q);
        if (q < len && at >= 0) g[at] = v[q];
      }
    } else if (stride == 1) {
      row_store4(g, i, len, v);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < len) g[(i + q) * stride] = v[q];
    }
  }
};
using GradRow = GradRowT<>;

struct ChunkGeom {
  int C;        // chunks
  int64_t L;    // cadences per chunk (the last may be shorter); a multiple of kCkptB on the one-lane path
  int64_t base; // first double of the chunk workspace inside `state`
  int lane;     // 1: one-lane chunk kernels with a checkpointed factorisation; 0: lane-group kernels, full factorisation
  int fine;     // levels of pairwise element composition: the elements are BUILT for C << fine chunks of L >> fine
                // cadences (that many times more lanes for the element kernel) and composed back up (0: built directly)
  int tree;     // 1: the scans over the chunks (B), (B') are trees of element compositions (lane-group path), 0: serial
  // Set by the host per launch, not by chunk_plan (EXO_GP_PREPARE_ADJOINT: the adjoint scan runs beside the forward chunk kernel):
  int prep;     // reverse chunk kernels: 1 = the adjoint states in bnd(2, .) were worked out for a cotangent of ONE (inside the
                // forward call): scale them by the draw's cotangent as they are loaded (the adjoint scan is linear in it)
  int which;    // forward chunk kernel: 0 = every draw, 1 = only draws flagged kFlagRobust, 2 = only the others
};

// chunk workspace, all [chunk][quantity][draw] (a lane is a draw: coalesced)
struct ChunkWs {
  int64_t n_draw;
  int J, C;
  int64_t base;
  int64_t n_blk;   // checkpoint blocks (0 on the lane-group path)
  EXO_HDH int E() const { return 3 * J * J + 2 * J; }   // A, b, Cm, eta, Jm
  EXO_HDH int B() const { return J + J * J; }           // vector + matrix
  EXO_HDH int K() const { return J + J * (J + 1) / 2; } // checkpoint: F + packed symmetric S
  EXO_HDH int64_t elem(int c, int e, int64_t draw) const { return base + ((int64_t)c * E() + e) * n_draw + draw; }
  EXO_HDH int64_t off_bnd() const { return base + (int64_t)C * E() * n_draw; }
  // q = 1: (F, P) entering chunk c;  2: adjoint of (F, S) entering chunk c + 1  (0: unused)
  EXO_HDH int64_t bnd(int q, int c, int k, int64_t draw) const {
    return off_bnd() + (((int64_t)q * C + c) * B() + k) * n_draw + draw;
  }
  EXO_HDH int64_t off_part() const { return off_bnd() + (int64_t)3 * C * B() * n_draw; }
  EXO_HDH int64_t part(int c, int k, int64_t draw) const {  // k = 0 acc, 1 logdet, 2 bad
    return off_part() + ((int64_t)c * 3 + k) * n_draw + draw;
  }
  EXO_HDH int64_t off_gpart() const { return off_part() + (int64_t)3 * C * n_draw; }
  EXO_HDH int64_t gpart(int c, int k, int64_t draw) const {  // k = 4 j + {a, b, c, d}; 4 J = gasum
    return off_gpart() + ((int64_t)c * (4 * J + 1) + k) * n_draw + draw;
  }
  EXO_HDH int64_t off_flag() const { return off_gpart() + (int64_t)C * (4 * J + 1) * n_draw; }
  EXO_HDH int64_t off_ckpt() const { return off_flag() + n_draw; }
  // checkpoint g = the state at cadence g ckpt_span(J): k < J: F_k; then packed S
  EXO_HDH int64_t ckpt(int64_t g, int k, int64_t draw) const { return off_ckpt() + (g * K() + k) * n_draw + draw; }
  // elements of the finer levels (ChunkGeom::fine): level f has C << f chunks; level `fine` is built, f = 0 is elem()
  int fine;
  EXO_HDH int64_t off_fine(int f) const {   // f = 1 .. fine
    int64_t o = off_ckpt() + n_blk * K() * n_draw;
    for (int g = fine; g > f; --g) o += ((int64_t)C << g) * E() * n_draw;
    return o;
  }
  EXO_HDH int64_t off_tree() const {
    int64_t o = off_ckpt() + n_blk * K() * n_draw;
    for (int g = 1; g <= fine; ++g) o += ((int64_t)C << g) * E() * n_draw;
    return o;
  }
  // scan tree (ChunkGeom::tree): level f = 1 .. tree_top() has tree_npos(f) positions, each an element and a state
  // (vector + matrix); level 0 is elem() / bnd(); the top level is one position
  int tree;
  EXO_HDH int tree_npos(int f) const {
    int p = C;
    for (int g = 0; g < f; ++g) p = (p + 1) / 2;
    return p;
  }
  EXO_HDH int tree_top() const {
    int f = 0;
    while (tree_npos(f) > 1) ++f;
    return f;
  }
  EXO_HDH int64_t tree_elem(int f) const {   // f >= 1
    int64_t o = off_tree();
    for (int g = 1; g < f; ++g) o += (int64_t)tree_npos(g) * (E() + B()) * n_draw;
    return o;
  }
  EXO_HDH int64_t tree_state(int f) const { return tree_elem(f) + (int64_t)tree_npos(f) * E() * n_draw; }
  EXO_HDH int64_t off_polish() const {
    int64_t o = off_tree();
    if (tree) o = tree_elem(tree_top() + 1);
    return o;
  }
  // polish (round 4): q = 0 the state a chunk's forward recurrences LEAVE for the next chunk (F, packed S at that chunk's
  // first cadence), q = 1 the adjoint of the state a chunk's reverse recurrences were ENTERED with (Fbar, packed Sbar);
  // q = 2, 3: the same from a polish pass (its lanes read pass 0's while they write)
  EXO_HDH int64_t polish(int q, int c, int k, int64_t draw) const {
    return off_polish() + (((int64_t)q * C + c) * K() + k) * n_draw + draw;
  }
  // J = 2 with per-draw pair kinds: the order in which the lanes of the chunk kernels take the draws -- the draws of one kind,
  // padded to whole waves, then those of the other (int32, -1 = no draw): celerite_kind_partition_kernel
  EXO_HDH int64_t off_perm() const { return off_polish() + (EXO_GP_POLISH ? (int64_t)4 * C * K() * n_draw : 0); }
  EXO_HDH int64_t perm_lanes() const { return ((n_draw + 63) / 64 + 1) * 64; }
  // sparse model: the order in which the one-lane kernels' blocks take the chunks (int32 [C]: celerite_sparse_order_kernel)
  EXO_HDH int64_t off_order() const { return off_perm() + (perm_lanes() + 1) / 2; }
  EXO_HDH int64_t total() const { return off_order() + (C + 1) / 2 - base; }
};

// the series and the measurement variance of one block of four cadences [b0, b0 + 4) clipped to n1.
// The chunk kernels load the NEXT block's before they work through the current one: a lane's row
// accesses miss every cache, and with 2-4 waves per SIMD nothing else hides ~1 us of latency per block.
EXO_HD double ckpt_load(const double* p) {
#if defined(__HIP_DEVICE_COMPILE__) && EXO_CKPT_NT_LOAD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
struct BlockIn {
  double y[4], g[4];
};
template <class RowT>
EXO_HD void load_block(const RowT& y, const double* EXO_RESTRICT dg, int64_t n_diag, int64_t b0, int64_t n1, BlockIn& o) {
  const int len = (int)((n1 - b0 < 4) ? (n1 - b0 > 0 ? n1 - b0 : 0) : 4);
  if (len == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { o.y[q] = 0.0; o.g[q] = 1.0; }
    return;
  }
  y.load4(b0, len, o.y);
  if (n_diag == 1) {   // shared by all draws: every lane reads the same address
#pragma unroll
    for (int q = 0; q < 4; ++q) o.g[q] = q < len ? dg[b0 + q] : 1.0;
  } else {
    row_load4(dg, b0, len, o.g);
  }
}

#ifndef EXO_CHUNK_MAX_J
#define EXO_CHUNK_MAX_J 16
#endif
constexpr int kChunkMaxJ = EXO_CHUNK_MAX_J;   // the time-parallel path's widest state.  J = 7, 8: lane-group element / chunk kernels on eight lanes, scans on groups of eight
                                            // lanes; J = 9 .. 16 (round 6): the same kernels on a DPP row of sixteen lanes, the scans one lane per item (kWideMinJ)
constexpr int kWideMinJ = 9;
// conditioning score above which a draw leaves the time-parallel path (elem_lane: how it was calibrated):
// EXO_GP_COND_MAX_J2 for state widths J <= 2, EXO_GP_COND_MAX for wider states
#ifndef EXO_GP_COND_MAX
#define EXO_GP_COND_MAX 3e4
#endif
#ifndef EXO_GP_COND_MAX_J2
#define EXO_GP_COND_MAX_J2 1e7
#endif
// ... and the score up to which such a draw takes the ROBUST route of the time-parallel path (round 4: serial application
// of the elements + the adjoint scan's inputs from the chunks' own recurrences, chunk_adj_lane) instead of the sequential
// kernels: tools/gp_lab_robust.py, 2400 random kernels -- every gradient of every draw under 1e8 within 4.5e-7 of the
// long-double dense definition, 2e-7 worst in [1e8, 1e9), 1e-5 beyond
#ifndef EXO_GP_COND_ROBUST_MAX
#define EXO_GP_COND_ROBUST_MAX 1e8
#endif
// draw flags (ChunkWs::off_flag): how a draw is finished
constexpr double kFlagClean = 0.0;    // scans as trees of element compositions
constexpr double kFlagRobust = 1.0;   // one-lane path only: serial scans, adjoint inputs from the recurrences
constexpr double kFlagSeq = 2.0;      // redone by the sequential kernels
// raise a draw's flag (several chunks' lanes may, with different verdicts: the highest stays)
EXO_HD void flag_raise(double* EXO_RESTRICT flag, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (non-negative doubles order like their bit patterns)
  atomicMax(reinterpret_cast<unsigned long long*>(flag), (unsigned long long)__double_as_longlong(v));
#else
  if (v > *flag) *flag = v;
#endif
}
// One term of the Newton iterations' convergence measure (celerite_robust_newton_kernel): the correction d of an entry of a
// boundary covariance against the scale sc = |P_jj P_ll| of that entry.  A correction or a scale that is not finite -- a tree
// guess or a tangent element gone NaN / Inf -- must NOT read as "converged" (fmax drops a NaN, and NaN > 0 is false: ADVICE r4):
// it reads as +inf, the iterations end unconverged and the draw falls through to the serial chain, which needs no guess.
EXO_HD double newton_err_term(double d, double sc) {
  const double x = sc > 0.0 ? fabs(d) * exo::fast_rcp(sqrt(sc)) : 0.0;
  const bool finite = (x < INFINITY) && (d == d) && (sc < INFINITY) && (fabs(d) < INFINITY);   // (sc < inf is false for a NaN too)
  return finite ? x : INFINITY;
}

#ifndef EXO_LANE_MAX_J
#define EXO_LANE_MAX_J 6
#endif
constexpr int kLaneMaxJ = EXO_LANE_MAX_J;   // one-lane chunk kernels up to this state width (registers: docs/DESIGN_r1_r4.md 4)

// doubles of the full saved factorisation (sequential / lane-group layout) at the head of `state`
EXO_HDH int64_t seq_state_doubles(int64_t n, int64_t n_draw, int J) {
  return n * n_draw * (int64_t)(2 + 2 * J + J * J + 3 * J);
}

// How a series is cut.  n_chunks = 0: the default plan; 1: sequential; > 1: forced.  A pure
// function of its arguments: the forward and the reverse call of a pair compute the same plan.
#ifndef EXO_FINE_LEVELS
#define EXO_FINE_LEVELS 1
#endif
constexpr int kFineLevels = EXO_FINE_LEVELS;   // J > 2: elements built for 2 x finer chunks, composed pairwise once
EXO_HDH ChunkGeom chunk_plan(int64_t n, int64_t n_draw, int J, int32_t n_chunks) {
  ChunkGeom g{1, n, seq_state_doubles(n, n_draw, J), 0, 0, 0};
  if (J > kChunkMaxJ || J < 1 || n < 64) return g;
  const bool lane = J <= kLaneMaxJ;
  int64_t C;
  if (n_chunks > 0) {
    C = n_chunks;
  } else if (lane) {
    // one lane per (draw, chunk): ~4 waves per SIMD (4096 waves of 64 lanes)
    C = (64 * 4096) / n_draw;
    if (C > 512) C = 512;
    if (C < 4) C = 4;
  } else {
    const int G = J <= 1 ? 1 : (J <= 2 ? 2 : (J <= 4 ? 4 : (J <= 8 ? 8 : 16)));
    // ~4 waves per SIMD offered to the lane-group chunk kernels of a group of eight (they fit 2-4): the scans over the chunks
    // are trees, so more chunks cost them little.  A row of sixteen (J >= 9): TWO -- its element and reverse kernels fit two
    // waves per SIMD, so 2048 waves are one full round of them, and its wide scan items (a block of 256 threads each) are
    // not cheap: measured at the C5 shape, J = 10 (tools/wide_chunks.py): 128 chains 6.76 -> 6.33 ms (64 chunks against 128;
    // 48: 7.5, 80: 7.7), 32 chains 2.87 -> 2.40, 512 chains 23.0 -> 22.8; J = 8 at half its chunks: 128 chains -2 %, 512 +7 %.
    C = (64 * (int64_t)(G >= 16 ? 2048 : 4096)) / (n_draw * G);
    if (C > 512) C = 512;
    if (C < 4) C = 1;
  }
  if (C > n / 32) C = n / 32;        // chunks of at least 32 cadences
  if (C < 2) return g;
  // J = 7, 8: the elements on half chunks (composed pairwise back up) only where the batch leaves the element kernel short of one
  // round of waves -- it fits two per SIMD since round 6, and with 2048 waves or more the finer level only adds its composition:
  // measured at the C5 shape, J = 8 (same box, alternating): 128 chains 3.31 -> 3.19 ms without it, 32 chains 1.38 -> 1.31,
  // 512 chains 10.93 -> 10.81; 16 chains (1024 waves) 1.13 -> 1.16: kept there
  const int fine = (lane || J >= kWideMinJ) ? 0 : ((n_draw * 8 * C) / 64 < 2048 ? kFineLevels : 0);
  g.L = (n + C - 1) / C;
  if (lane) g.L = (g.L + kCkptB - 1) / kCkptB * kCkptB;
  else if (fine) g.L = (g.L + (1 << fine) - 1) >> fine << fine;   // whole fine chunks
  g.C = (int)((n + g.L - 1) / g.L);
  if (g.C < 2) { g.C = 1; g.L = n; return g; }
  g.lane = lane ? 1 : 0;
  g.fine = fine;   // (the pairwise composition of fine elements: the scan trees' group kernel, J <= 8)
  g.tree = 1;   // the scans over the chunks as trees of compositions (one lane each)
  return g;
}

EXO_HDH ChunkWs chunk_ws(int64_t n, int64_t n_draw, int J, const ChunkGeom& g) {
  return ChunkWs{n_draw, J, g.C, g.base, g.lane ? (n + ckpt_span(J) - 1) / ckpt_span(J) : 0, g.fine, g.tree};
}
// the geometry the element kernels see when they build level f: C << f chunks of L >> f cadences, elem()
// addressing that level's array; flags stay where the coarse geometry has them (flag_at)
EXO_HDH ChunkGeom fine_geom(int64_t n, int64_t n_draw, int J, const ChunkGeom& g, int f) {
  ChunkGeom q = g;
  q.C = g.C << f;
  q.L = g.L >> f;
  q.base = chunk_ws(n, n_draw, J, g).off_fine(f);
  q.fine = 0;
  q.tree = 0;
  return q;
}

// symmetric J x J in packed upper-triangular storage
template <int J>
struct Sym {
  double v[J * (J + 1) / 2];
  EXO_HD static constexpr int idx(int j, int l) {
    return j <= l ? j * J - j * (j - 1) / 2 + (l - j) : l * J - l * (l - 1) / 2 + (j - l);
  }
  EXO_HD double& operator()(int j, int l) { return v[idx(j, l)]; }
  EXO_HD double operator()(int j, int l) const { return v[idx(j, l)]; }
};

// Delta_n of one draw from its term coefficients and V_n (block diagonal: 1x1 / 2x2 blocks).
// Delta is symmetric with Delta_n U_n = V_n: the state covariance of the process the kernel
// describes, in celerite's rotating frame (exo_celerite.hip, "Time-parallel path").
// NR: the term layout, when it is known at compile time -- the first NR state indices are real terms,
// the rest complex pairs -- or -1: decided per index at run time (pair slots of mixed kinds).  The
// one-lane kernels pick the variant by a wave vote (with_layout): with the layout a constant, the
// per-index selects and branches of the recurrences fold away (a third of their instructions).
template <int J, int NR = -1>
struct DeltaCoef {
  double p[J], q[J], r[J];   // per state index: real term -> p = 1/a; pair -> (p, q, r) on both indices
  bool real[J], first[J];    // first: first index of a complex pair
  bool valid;
  EXO_HD bool is_real(int j) const { return NR < 0 ? real[j] : j < NR; }
  EXO_HD bool is_first(int j) const { return NR < 0 ? first[j] : (j >= NR && ((j - NR) & 1) == 0); }
  // Two real terms that share a pair slot (kind 1: an SHO term with Q < 1/2) and are NOT both positive -- the
  // over-damped oscillator's a2 = S0 w0 Q (1 - 1 / f) / 2 < 0 -- have no covariance of their own (1 / a2 < 0), but the
  // pair has a joint one: the process z' = -diag(c) z + g w (one noise input), observed through (1, 1), has the
  // stationary covariance Pi_ij = g_i g_j / (c_i + c_j) and the kernel sum_i a_i exp(-c_i tau) with a_i = sum_j Pi_ij.
  // With x = Pi_11, y = Pi_22, z = Pi_12:  a1 = x + z, a2 = y + z, z^2 = k x y, k = 4 c1 c2 / (c1 + c2)^2, i.e.
  //     (1 - k) z^2 + k (a1 + a2) z - k a1 a2 = 0,     z = the root that leaves x = a1 - z, y = a2 - z >= 0
  // (the discriminant vanishes exactly for an SHO term: the only admissible member, as for the complex pair), and in
  // celerite's scaling (U = a, V = 1) Delta = [[x / a1^2, z / (a1 a2)], [., y / a2^2]]: Delta U = V, and the process
  // noise Delta - phi Delta phi is positive semi-definite.  Round 2 sent every such draw -- every chain whose Q
  // crosses 1/2 -- to the sequential kernels: ~50x the step time of the whole batch.
  EXO_HD void init(const Coefs& co, int64_t draw, bool allow_coupled = (J <= EXO_LANE_MAX_J)) {
    valid = true;
    double a1 = 0.0, c1 = 0.0;
    bool ok1 = true;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const LaneCoef k = lane_coef(co, draw, j, J);
      real[j] = k.real;
      first[j] = !k.real && !k.odd;
      if (is_real(j)) {
        p[j] = 1.0 / k.a; q[j] = 0.0; r[j] = 0.0;
        const bool ok = (k.a > 0.0) && (k.c >= 0.0) && (k.a < INFINITY) && (k.c < INFINITY);
        const bool pair1 = k.slot >= 0 && (k.slot & 2) == 0, pair2 = k.slot >= 0 && (k.slot & 2) != 0;
        if (pair1) {          // (decided with its partner)
          a1 = k.a; c1 = k.c; ok1 = ok;
        } else if (pair2 && j > 0 && !(ok1 && ok)) {
          const double a2 = k.a, c2 = k.c, cs = c1 + c2, kk = 4.0 * c1 * c2 / (cs * cs), omk = (c1 - c2) * (c1 - c2) / (cs * cs);
          const double sa = a1 + a2, disc = kk * kk * sa * sa + 4.0 * omk * kk * a1 * a2;
          const double z = -(kk * sa + sqrt(fmax(disc, 0.0))) / (2.0 * omk);
          const double x = a1 - z, y = a2 - z;
          const bool okc = allow_coupled && (c1 > 0.0) && (c2 > 0.0) && (omk > 0.0) && (sa > 0.0) && (x >= 0.0) && (y >= 0.0) &&
                           (disc >= -1e-9 * kk * kk * sa * sa) && (x < INFINITY) && (y < INFINITY) && (c1 < INFINITY) && (c2 < INFINITY);
          p[j > 0 ? j - 1 : 0] = x / (a1 * a1);
          q[j > 0 ? j - 1 : 0] = z / (a1 * a2);
          p[j] = y / (a2 * a2);
          valid = valid && okc;
        } else if (pair2) {
          valid = valid && ok1 && ok;
        } else {
          valid = valid && ok;
        }
      } else {
        const double a = k.a, b = k.b, c = k.c, d = k.d;
        const double h = 1.0 / (a * a + b * b);
        r[j] = a * h; q[j] = -b * h; p[j] = (a * a + 2.0 * b * b) * h / a;
        valid = valid && (a > 0.0) && (fabs(b * d) <= a * c * (1.0 + 1e-12)) && (a < INFINITY) && (fabs(b) < INFINITY) &&
                (c < INFINITY) && (fabs(d) < INFINITY);
      }
    }
  }
  // Delta (packed) from V (cos / sin of each pair)
  EXO_HD void eval(const double* V, Sym<J>& D) const {
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) D.v[k] = 0.0;
    // (every store below is unconditional, at a compile-time index, with the layout deciding the VALUE: with run-time
    // layouts the compiler merges the branches' stores into one with a run-time index, and an array indexed at run
    // time lives in scratch memory -- all of it, for the whole kernel: what made the one-lane kernels crawl at J > 2)
    double next_dd = 0.0;   // the pair's second diagonal entry, handed over by its first index
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool re = is_real(j), fi = !re && is_first(j) && j + 1 < J;
      const double cs = V[j], sn = V[j + 1 < J ? j + 1 : j];
      const double dd = p[j] * cs * cs + 2.0 * q[j] * cs * sn + r[j] * sn * sn;
      const double od = (p[j] - r[j]) * cs * sn + q[j] * (sn * sn - cs * cs);
      const double d2 = p[j] * sn * sn - 2.0 * q[j] * cs * sn + r[j] * cs * cs;
      D(j, j) = re ? p[j] : (fi ? dd : next_dd);
      if (j + 1 < J) D(j, j + 1) = fi ? od : (re ? q[j] : 0.0);   // (a real index: 0, or the coupling of a joint pair)
      next_dd = fi ? d2 : 0.0;
    }
  }
};

// U_n, V_n for all J state indices of one draw (one sincos per complex pair); same arithmetic as
// lane_uv, so the values are those the lane-group kernels compute
template <int J, int NR = -1>
struct DrawCoef {
  LaneCoef k[J];
  // the REFERENCE STEP dt_ref of the stretch being walked (see step / rot_uv): a pair's rotation (cos, sin)(d dt_ref),
  // the propagators exp(-c dt_ref); dt_miss: the last step that was not near the reference
  double rc[J], rs[J], ph[J], dt_ref, dt_miss;
  bool near_ok;
  EXO_HD bool is_real(int j) const { return NR < 0 ? k[j].real : j < NR; }
  EXO_HD bool is_first(int j) const { return NR < 0 ? (!k[j].real && !k[j].odd) : (j >= NR && ((j - NR) & 1) == 0); }
  EXO_HD void init(const Coefs& co, int64_t draw) {
#pragma unroll
    for (int j = 0; j < J; ++j) { k[j] = lane_coef(co, draw, j, J); rc[j] = 1.0; rs[j] = 0.0; ph[j] = 1.0; }
    dt_ref = dt_miss = -1.0;
    near_ok = false;
  }
  // (stores unconditional at compile-time indices, see DeltaCoef::eval; a pair's second index gets its values from the
  // first through `nu`, `nv`)
  EXO_HD void uv(double t, double* U, double* V) const {
    double nu = 0.0, nv = 0.0;
    bool carry = false;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double uj = carry ? nu : U[j], vj = carry ? nv : V[j];
      carry = false;
      if (is_real(j)) {
        uj = k[j].a; vj = 1.0;
      } else if (is_first(j)) {
        double sn, cs;
        exo::sincos_any(k[j].d * (t - k[j].t0), &sn, &cs);
        uj = k[j].a * cs + k[j].b * sn; vj = cs;
        nu = k[j].a * sn - k[j].b * cs; nv = sn;
        carry = true;
      }
      U[j] = uj; V[j] = vj;
    }
  }
  // U from V alone (V of a pair is (cos, sin)): the reverse pass keeps V and rebuilds U
  EXO_HD void u_from_v(const double* V, double* U) const {
    double nu = 0.0;
    bool carry = false;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool re = is_real(j), fi = !re && is_first(j) && j + 1 < J;
      const double vn = V[j + 1 < J ? j + 1 : j];
      const double uj = re ? k[j].a : (fi ? k[j].a * V[j] + k[j].b * vn : (carry ? nu : U[j]));
      nu = k[j].a * vn - k[j].b * V[j];
      carry = fi;
      U[j] = uj;
    }
  }
  // Steps of an (almost) evenly sampled stretch.  Time stamps that are "evenly sampled" differ from step to step in
  // their last bits (t_i = i dt rounds each product: two thirds of consecutive steps of such a series are not
  // bitwise equal), and real ones by the barycentric correction (~1e-9 of the step from one cadence to the next), so
  // "same step as before" cannot be a bitwise test.  The first step of a stretch becomes the reference: its
  // propagators exp(-c dt_ref) and pair rotations (cos, sin)(d dt_ref) are evaluated once, and a step within 2^-20
  // of it takes them CORRECTED for the difference del = dt - dt_ref to second order,
  //     exp(-c dt) = ph (1 - x + x^2 / 2),  x = c del;      R(d dt) = R(d del) R(d dt_ref),  R(e) ~ (1 - e^2/2, e),
  // exact to (|c|, |d|) |del| cubed / 6 <= 1e-16 when max(|c|, |d|) dt_ref <= 8 (near_ok).  Any other step (a gap)
  // is evaluated exactly and leaves the reference alone; two such steps of the same size in a row re-reference.
  EXO_HD void set_ref(double dt) {
    dt_ref = dt;
    double big = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      ph[j] = exp(-k[j].c * dt);
      big = fmax(big, fabs(k[j].c));
      double sn = rs[j], cs = rc[j];
      if (!is_real(j) && is_first(j)) {
        exo::sincos_any(k[j].d * dt, &sn, &cs);
        big = fmax(big, fabs(k[j].d));
      }
      rs[j] = sn; rc[j] = cs;
    }
    near_ok = big * dt <= 8.0 && dt > 0.0;
  }
  // propagators of the step dt into phi; true: the step is near the reference (rot_uv may follow)
  // (track = false: a step that was already seen -- the reverse pass asks for a step's propagators again -- does not
  // count towards re-referencing)
  EXO_HD bool step(double dt, double* phi, bool track = true) {
    constexpr double kTol = 9.5367431640625e-07;   // 2^-20
    if (dt_ref < 0.0) set_ref(dt);
    bool near = near_ok && fabs(dt - dt_ref) <= kTol * dt_ref;
    if (!near && track) {
      if (dt_miss > 0.0 && fabs(dt - dt_miss) <= kTol * dt_miss) {
        set_ref(dt);
        near = near_ok;
      }
      dt_miss = dt;
    }
    if (near) {
      const double del = dt - dt_ref;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double x = k[j].c * del;
        phi[j] = ph[j] * fma(x, fma(0.5, x, -1.0), 1.0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) phi[j] = exp(-k[j].c * dt);
    }
    return near;
  }
  // (U, V) one near-reference step dt after the cadence whose V is Vp: the pair's (cos, sin) rotated by d dt
  // (a dozen multiply-adds instead of a 45-instruction sincos).  The kernels restart from the exact values at the
  // first cadence of every block of four (counted from the chunk start): three rotations in a row drift by ~3e-16.
  EXO_HD void rot_uv(double dt, const double* Vp, double* U, double* V) const {
    const double del = dt - dt_ref;
    double nv = 0.0;
    bool carry = false;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool re = is_real(j), fi = !re && is_first(j) && j + 1 < J;
      const double cs = Vp[j], sn = Vp[j + 1 < J ? j + 1 : j];
      const double c1 = fma(cs, rc[j], -sn * rs[j]), s1 = fma(sn, rc[j], cs * rs[j]);
      const double e = k[j].d * del, h = fma(-0.5 * e, e, 1.0);
      const double vj = re ? 1.0 : (fi ? fma(-e, s1, c1 * h) : (carry ? nv : V[j]));
      nv = fma(e, c1, s1 * h);
      carry = fi;
      V[j] = vj;
    }
    u_from_v(V, U);
  }
  EXO_HD double asum() const {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) s += (is_real(j) || is_first(j)) ? k[j].a : 0.0;
    return s;
  }
};

// The term layout of this lane's draw, voted over the wave: f(std::integral_constant<int, NR>) is
// called with NR = the number of leading real terms when every lane of the wave has the layout
// "NR real terms, then complex pairs", and with NR = -1 (run-time flags) otherwise.  J <= 2: the
// layouts are R (J = 1); R R or one complex pair (J = 2; R R also for a pair slot of kind 1).
#ifndef EXO_GP_WIDE_COMPLEX_LAYOUT
#define EXO_GP_WIDE_COMPLEX_LAYOUT 1   // 0: wide states always take the run-time layout (A/B)
#endif
template <int J>
EXO_HD int layout_vote(const Coefs& cf, int64_t draw) {
  if (J == 1) return 1;
  if (J == 2) {
    const bool rr = cf.n_real == 2 || (cf.kind != nullptr && cf.kind[cf.at(draw)] != 0);   // (n_complex = 1: one slot per draw)
    if (EXO_WAVE_ALL(!rr)) return 0;
    if (EXO_WAVE_ALL(rr)) return 2;
  }
  if (EXO_GP_WIDE_COMPLEX_LAYOUT && J > 2 && (J % 2) == 0 && cf.n_real == 0) {
    // wide states made of pair slots only (sums of SHO terms: C5 is three of them): when every slot of every draw of the
    // wave is a COMPLEX term the layout is the compile-time "no real terms" one (round 4) -- the per-index selects and flags of
    // the run-time layout are a quarter of the instructions of the J = 6 kernels
    bool allc = true;
    if (cf.kind != nullptr)
      for (int q = 0; q < J / 2; ++q) allc = allc && (cf.kind[cf.at(draw) * (J / 2) + q] == 0);
    if (EXO_WAVE_ALL(allc)) return 0;
  }
  return -1;
}
// (host harness: the variant as a call; on the device every variant is a kernel of its own -- its own
// register budget -- and a wave returns at once from the variants it did not vote for)
template <int J, typename F>
EXO_HD void with_layout(const Coefs& cf, int64_t draw, F&& f) {
  const int nr = layout_vote<J>(cf, draw);
  if (J == 1) {
    f(std::integral_constant<int, J == 1 ? 1 : -1>{});
  } else if (J == 2 && nr == 0) {
    f(std::integral_constant<int, J == 2 ? 0 : -1>{});
  } else if (J == 2 && nr == 2) {
    f(std::integral_constant<int, J == 2 ? 2 : -1>{});
  } else if (J > 2 && nr == 0) {
    f(std::integral_constant<int, (J > 2) ? 0 : -1>{});
  } else {
    f(std::integral_constant<int, -1>{});
  }
}

// (A) the filtering element of one (draw, chunk)
// (flag_at: where the draw flags live when cg is a fine geometry, -1: cg's own)
// PREFETCH: the next block's series in flight while this one is worked through.  Off for J <= 2 on the device (round 4): its
// sixteen registers are the difference between three and four waves per SIMD there, and a 1024-draw batch is 4096 waves --
// 1.33 rounds of resident waves at three per SIMD, one round at four: element kernel 0.90 -> 0.82 ms at C3.
template <int J, int NR = -1, bool PREFETCH = true, int SP = -1>
EXO_HD void elem_lane(const double* EXO_RESTRICT t, Series rs, const double* EXO_RESTRICT diag, int64_t n_diag,
                      int64_t n, const Coefs& cf, int64_t n_draw, double* EXO_RESTRICT state, const ChunkGeom& cg,
                      int64_t draw, int c, int64_t flag_at = -1) {
  // (a chunk past the end of the series -- fine chunks of a short last chunk -- has no cadence: the identity element)
  const bool empty = c * cg.L >= n;
  const int64_t n0 = empty ? n - 1 : c * cg.L, n1 = empty ? n0 : ((n0 + cg.L < n) ? n0 + cg.L : n);
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  DeltaCoef<J, NR> dc;
  dc.init(cf, draw);
  DrawCoef<J, NR> co;
  co.init(cf, draw);
  const SeriesRowT<true, SP> y(rs, draw, n);
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  double A[J][J], b[J], eta[J];
  Sym<J> Cm, Jm, Dl;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    b[j] = eta[j] = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) A[j][l] = (j == l) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < J * (J + 1) / 2; ++k) Cm.v[k] = Jm.v[k] = 0.0;
  double U[J], V[J], phi[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { U[j] = V[j] = 0.0; phi[j] = 1.0; }
  double ti = t[n0];
  co.uv(ti, U, V);
  dc.eval(V, Dl);
  // Conditioning.  The element is in information form (1 / diag) and lives in celerite's rotating
  // frame, where a complex term's state covariance Delta0 has condition number ~ 4 (b / a)^2: the
  // J x J solves of the scans lose accuracy with  kappa = (1 + max (b/a)^2) sum(a) / min(diag)  -- and with the state
  // width and the spread of the terms' time scales.  Two measurements with the flag off set the thresholds:
  //  * the benchmarked shapes (tools/gp_cond_scale.py: N = 150 000, J = 2, SHO terms from Q = 4 to within 1e-4 of
  //    critical damping on either side, celerite2's Matern-3/2 term, signal / noise variance 4 .. 1e6): every gradient
  //    within 3e-7 of the sequential kernels up to kappa = 5e7; within 1e-6 of the long-double dense definition in
  //    the seven regimes of tests/golden/gp_hard.npz;
  //  * 9 600 draws of RANDOM kernels (tools/gp_cond_bins.py: J = 1 .. 6, decay and oscillation rates from 1e-3 to 30
  //    per sample within one kernel, gaps): worst gradient disagreement per decade of kappa -- J <= 2: 2e-6 (1e5),
  //    8e-6 (1e6), 5e-2 (1e7); J = 3: 1e-7, 4e-5, 2e-4; J = 4 .. 6: 1e-5 .. 7e-5 (1e5), 4e-4 .. 5e-3 (1e6); medians
  //    1e-11 .. 1e-9 throughout: the tail is kernels whose terms differ by four decades in time scale, and it is
  //    heavy for wide states.
  // Hence: a draw is flagged -- until round 4 redone by the sequential kernels; now kept on this path by its robust route up
  // to EXO_GP_COND_ROBUST_MAX (chunk_adj_lane), sequential beyond -- above kappa = 1e7 for J <= 2 (round 2: 1e5; an SHO
  // term within 1e-4 of critical damping, Matern-3/2, a signal 1e6 x the noise stay on this path) and above 1e5 for
  // wider states (as in round 2; 3e4 since round 4, below).  diag = 0 is flagged whatever J.
  // ROUND 4: that tail was ONE gradient -- d loglike / d(oscillation rate of a complex term), whose cancellation across
  // chunk boundaries the absolute-time form could not keep (phase_flux); with the flux form the same scan (tools/gp_cond_bins.py,
  // 11 520 draws, flags off, worst disagreement with the sequential kernels per decade of kappa starting at 1e4 / 1e5 / 1e6 /
  // 1e7) reads  J <= 2: 1e-9, 2e-8, 7e-8, 1e-5;  J = 3: 2e-8, 3e-7, 2e-5, 2e-3;  J = 4: 6e-8, 5e-7, 7e-5;  J = 5, 6: 2e-8,
  // 1e-5 (one draw; median 3e-11), 3e-4 -- what is left is conditioning proper (every gradient, the log-likelihood at 1e-9).
  // Thresholds (margin): J <= 2 stays at 1e7 (7e-8 in the decade below it).  For wider states a cell-by-cell scan with
  // more draws (tools/gp_cond_cell.py: 3200 - 4800 draws per cell) reads 5.7e-7 in [1e4, 1e5) and 4.7e-6 in [1e5, 1e6) even for
  // kernels whose decay rates are all well sampled (c dt <= 3) -- the cut at 1e5 sat within a factor of two of the stated 1e-6: it
  // is 3e4 now.  tests/golden/gp_tail.npz and tests/test_gpu_golden.py pin it.
  double asum = 0.0, ba2 = 0.0, a_first = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    asum += (co.is_real(j) || co.is_first(j)) ? fabs(co.k[j].a) : 0.0;
    if (!co.is_real(j) && co.is_first(j)) ba2 = fmax(ba2, (co.k[j].b * co.k[j].b) / (co.k[j].a * co.k[j].a));
    // a joint pair of real terms (DeltaCoef::init): (|a1| + |a2|) / (a1 + a2) = 1 / f plays the part of b / a = 1 / f
    if (co.is_real(j) && co.k[j].slot >= 0) {
      if ((co.k[j].slot & 2) == 0) {
        a_first = co.k[j].a;
      } else if (!(a_first > 0.0 && co.k[j].a > 0.0)) {
        const double w = (fabs(a_first) + fabs(co.k[j].a)) / (a_first + co.k[j].a);
        ba2 = fmax(ba2, w * w);
      }
    }
  }
  const double rmin = (1.0 + ba2) * asum * (1.0 / (J <= 2 ? EXO_GP_COND_MAX_J2 : EXO_GP_COND_MAX));
  const double rmin_seq = (1.0 + ba2) * asum * (1.0 / EXO_GP_COND_ROBUST_MAX);   // below: not even the robust route
  bool ok = true, ok_robust = true;
  BlockIn cur, nxt;
  load_block(y, dg, n_diag, n0, n1, cur);
#pragma unroll 1
  for (int64_t b0 = n0; b0 < n1; b0 += 4) {
   if (PREFETCH) load_block(y, dg, n_diag, b0 + 4, n1, nxt);   // in flight while this block is worked through
#pragma unroll
   for (int q = 0; q < 4; ++q) {
    const int64_t i = b0 + q;
    if (i < n1) {
    const double yi = cur.y[q], R = cur.g[q];
    ok = ok && (R >= rmin) && (R < INFINITY);
    ok_robust = ok_robust && (R >= rmin_seq) && (R < INFINITY);
    double r[J], cu[J];
    double s = R, zeta = yi;
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double rl = 0.0, cl = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) { rl = fma(A[j][l], U[j], rl); cl = fma(Cm(l, j), U[j], cl); }
      r[l] = rl; cu[l] = cl;
      s = fma(U[l], cl, s);
      zeta = fma(-U[l], b[l], zeta);
    }
    const double is = exo::fast_rcp(s);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      eta[j] = fma(r[j] * is, zeta, eta[j]);
#pragma unroll
      for (int l = j; l < J; ++l) Jm(j, l) = fma(r[j] * is, r[l], Jm(j, l));
    }
    if (i + 1 < n) {
      double Vn[J];
      const double tn = t[i + 1], dt = tn - ti;
      ti = tn;
      const bool near = co.step(dt, phi);
      // exact at the first cadence of every block of four (counted from the chunk start), rotated in between
      if (((i + 1 - n0) & 3) == 0 || !near) co.uv(tn, U, Vn); else co.rot_uv(dt, V, U, Vn);
#pragma unroll
      for (int j = 0; j < J; ++j) V[j] = Vn[j];
      Sym<J> Dn;
      dc.eval(Vn, Dn);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double kj = cu[j] * is;
        b[j] = phi[j] * fma(kj, zeta, b[j]);
#pragma unroll
        for (int l = 0; l < J; ++l) A[j][l] = phi[j] * fma(-kj, r[l], A[j][l]);
#pragma unroll
        for (int l = j; l < J; ++l)
          Cm(j, l) = fma(phi[j] * phi[l], fma(-kj, cu[l], Cm(j, l)) - Dl(j, l), Dn(j, l));  // + Q = Dn - phi phi Dl
      }
      Dl = Dn;
    }
    }
   }
   if (PREFETCH) cur = nxt;
   else load_block(y, dg, n_diag, b0 + 4, n1, cur);
  }
  if (!ok) flag_raise(state + (flag_at >= 0 ? flag_at : ws.off_flag()) + draw, ok_robust ? kFlagRobust : kFlagSeq);
  int e = 0;
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int l = 0; l < J; ++l) state[ws.elem(c, e++, draw)] = A[j][l];
#pragma unroll
  for (int j = 0; j < J; ++j) state[ws.elem(c, e++, draw)] = b[j];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int l = 0; l < J; ++l) state[ws.elem(c, e++, draw)] = Cm(j, l);
#pragma unroll
  for (int j = 0; j < J; ++j) state[ws.elem(c, e++, draw)] = eta[j];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int l = 0; l < J; ++l) state[ws.elem(c, e++, draw)] = Jm(j, l);
}

// solve X Z = B (J x J, NB right-hand sides) in place by Gaussian elimination with partial pivoting
// Loops of the scan-tree items (solve, compose, apply, load / store of an element): unrolled in full for J <= 8 -- the elements
// live in registers, a run-time index would send them to scratch -- and ROLLED for the wide states J = 9 .. 16 (round 6), whose
// 3 J^2 + 2 J doubles per element live in scratch whatever the loops do: unrolled, the 32 wide instantiations of the tree kernel
// (a J^3 product each, several times) took the compiler more than a quarter of an hour; rolled, seconds.
#define EXO_UNROLL_TREE _Pragma("unroll (J <= 8 ? 64 : 1)")
template <int J, int NB>
EXO_HD void solve_inplace(double (&X)[J][J], double (&B)[J][NB]) {
#if defined(EXO_HOST_BUILD) && defined(EXO_SOLVE_LD)
  // (host experiments, tools/gp_host_lab.py: is the rounding of these J x J solves what the conditioning tail is made of?)
  {
    long double Xl[J][J], Bl[J][NB];
    for (int i = 0; i < J; ++i) { for (int l = 0; l < J; ++l) Xl[i][l] = X[i][l]; for (int l = 0; l < NB; ++l) Bl[i][l] = B[i][l]; }
    for (int k = 0; k < J; ++k) {
      int piv = k;
      for (int i = k + 1; i < J; ++i) if (fabsl(Xl[i][k]) > fabsl(Xl[piv][k])) piv = i;
      for (int l = 0; l < J; ++l) { long double tmp = Xl[k][l]; Xl[k][l] = Xl[piv][l]; Xl[piv][l] = tmp; }
      for (int l = 0; l < NB; ++l) { long double tmp = Bl[k][l]; Bl[k][l] = Bl[piv][l]; Bl[piv][l] = tmp; }
      for (int i = 0; i < J; ++i) {
        if (i == k) continue;
        const long double f = Xl[i][k] / Xl[k][k];
        for (int l = 0; l < J; ++l) Xl[i][l] -= f * Xl[k][l];
        for (int l = 0; l < NB; ++l) Bl[i][l] -= f * Bl[k][l];
      }
    }
    for (int i = 0; i < J; ++i) for (int l = 0; l < NB; ++l) B[i][l] = (double)(Bl[i][l] / Xl[i][i]);
    return;
  }
#endif
EXO_UNROLL_TREE
  for (int k = 0; k < J; ++k) {
    int piv = k;
    double best = fabs(X[k][k]);
EXO_UNROLL_TREE
    for (int i = k + 1; i < J; ++i) {
      const bool better = fabs(X[i][k]) > best;
      best = better ? fabs(X[i][k]) : best;
      piv = better ? i : piv;
    }
EXO_UNROLL_TREE
    for (int i = k + 1; i < J; ++i) {
      if (i == piv) {   // swap rows k and i (selects: piv is a run-time value)
EXO_UNROLL_TREE
        for (int l = 0; l < J; ++l) { const double tmp = X[k][l]; X[k][l] = X[i][l]; X[i][l] = tmp; }
EXO_UNROLL_TREE
        for (int l = 0; l < NB; ++l) { const double tmp = B[k][l]; B[k][l] = B[i][l]; B[i][l] = tmp; }
      }
    }
    const double ip = 1.0 / X[k][k];
EXO_UNROLL_TREE
    for (int i = k + 1; i < J; ++i) {
      const double f = X[i][k] * ip;
EXO_UNROLL_TREE
      for (int l = k + 1; l < J; ++l) X[i][l] = fma(-f, X[k][l], X[i][l]);
EXO_UNROLL_TREE
      for (int l = 0; l < NB; ++l) B[i][l] = fma(-f, B[k][l], B[i][l]);
    }
  }
EXO_UNROLL_TREE
  for (int k = J - 1; k >= 0; --k) {
    const double ip = 1.0 / X[k][k];
EXO_UNROLL_TREE
    for (int l = 0; l < NB; ++l) {
      double v = B[k][l];
EXO_UNROLL_TREE
      for (int i = k + 1; i < J; ++i) v = fma(-X[k][i], B[i][l], v);
      B[k][l] = v * ip;
    }
  }
}

template <int J>
struct Elem {
  double A[J][J], b[J], Cm[J][J], eta[J], Jm[J][J];
  EXO_HD void load(const double* EXO_RESTRICT state, const ChunkWs& ws, int c, int64_t draw) {
    int e = 0;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) A[j][l] = state[ws.elem(c, e++, draw)];
#pragma unroll
    for (int j = 0; j < J; ++j) b[j] = state[ws.elem(c, e++, draw)];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) Cm[j][l] = state[ws.elem(c, e++, draw)];
#pragma unroll
    for (int j = 0; j < J; ++j) eta[j] = state[ws.elem(c, e++, draw)];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) Jm[j][l] = state[ws.elem(c, e++, draw)];
  }
};

// a filtering element applied to the state entering it: (m, P) <- the state it leaves
template <int J>
EXO_HD void apply_elem(const Elem<J>& el, double (&m)[J], double (&P)[J][J]) {
  // X = I + P Jm ;  solve X [YP | ym] = [P | m + P eta]
  double X[J][J], Bm[J][J + 1];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    double pe = m[j];
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double x = (j == l) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) x = fma(P[j][k], el.Jm[k][l], x);
      X[j][l] = x;
      Bm[j][l] = P[j][l];
      pe = fma(P[j][l], el.eta[l], pe);
    }
    Bm[j][J] = pe;
  }
  solve_inplace<J, J + 1>(X, Bm);
  // m' = A ym + b ;  P' = A (YP) A^T + Cm  (symmetrised)
  double AY[J][J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    double mj = el.b[j];
#pragma unroll
    for (int l = 0; l < J; ++l) {
      mj = fma(el.A[j][l], Bm[l][J], mj);
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(el.A[j][k], Bm[k][l], v);
      AY[j][l] = v;   // A (YP)
    }
    m[j] = mj;
  }
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double v = el.Cm[j][l];
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(AY[j][k], el.A[l][k], v);
      X[j][l] = v;
    }
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int l = 0; l < J; ++l) P[j][l] = 0.5 * (X[j][l] + X[l][j]);
}

// (B) the state entering every chunk of one draw: C - 1 element applications
template <int J>
EXO_HD void bscan_lane(const double* EXO_RESTRICT t, const Coefs& cf, int64_t n, int64_t n_draw,
                       double* EXO_RESTRICT state, const ChunkGeom& cg, int64_t draw) {
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  DeltaCoef<J> dc;
  dc.init(cf, draw);
  DrawCoef<J> co;
  co.init(cf, draw);
  double m[J], P[J][J];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    m[j] = 0.0;
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) P[j][l] = 0.0;
  }
#pragma unroll 1
  for (int c = 0; c < cg.C; ++c) {
    if (c == 0) {
      double U[J], V[J];
EXO_UNROLL_TREE
      for (int j = 0; j < J; ++j) U[j] = V[j] = 0.0;
      co.uv(t[0], U, V);
      Sym<J> Dl;
      dc.eval(V, Dl);
EXO_UNROLL_TREE
      for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
        for (int l = 0; l < J; ++l) P[j][l] = Dl(j, l);   // S_0 = 0
    }
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      state[ws.bnd(1, c, j, draw)] = m[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) state[ws.bnd(1, c, J + j * J + l, draw)] = P[j][l];
    }
    if (c + 1 == cg.C) break;
    Elem<J> el;
    el.load(state, ws, c, draw);
    apply_elem<J>(el, m, P);
  }
}

// (B'), part 1 -- everything in the chain rule across chunks that does not depend on the adjoint
// coming from later chunks, for one (draw, chunk c >= 1) (the J x J solve lives here):
//   Abar = A Y,  g = eta - Jm Y (F + P eta),  local adjoints  gL w  and  gL/2 (w w^T - Jm Y),
// written over the chunk's element (A <- Abar, b <- g, eta <- local Fbar, Cm <- local Pbar).
template <int J>
EXO_HD void badj_prep_lane(const double* EXO_RESTRICT gloglike, int64_t n, int64_t n_draw, double* EXO_RESTRICT state,
                           const ChunkGeom& cg, int64_t draw, int c, const int32_t* row = nullptr) {
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const double gL = gloglike ? gloglike[row ? (int64_t)row[draw] : draw] : 1.0;   // (null: a cotangent of one -- ChunkGeom::prep)
  Elem<J> el;
  el.load(state, ws, c, draw);
  double m[J], P[J][J];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    m[j] = state[ws.bnd(1, c, j, draw)];
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) P[j][l] = state[ws.bnd(1, c, J + j * J + l, draw)];
  }
  double X[J][J], Y[J][J];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) {
      double x = (j == l) ? 1.0 : 0.0;
EXO_UNROLL_TREE
      for (int k = 0; k < J; ++k) x = fma(P[j][k], el.Jm[k][l], x);
      X[j][l] = x;
      Y[j][l] = (j == l) ? 1.0 : 0.0;
    }
  solve_inplace<J, J>(X, Y);
  double u[J], v[J], w[J], Yv[J];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    double uj = el.eta[j], vj = m[j];
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) { uj = fma(-el.Jm[j][l], m[l], uj); vj = fma(P[j][l], el.eta[l], vj); }
    u[j] = uj; v[j] = vj;
  }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    double wj = 0.0, yv = 0.0;
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) { wj = fma(Y[l][j], u[l], wj); yv = fma(Y[j][l], v[l], yv); }
    w[j] = wj; Yv[j] = yv;
  }
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J;
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    double gj = el.eta[j];
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) {
      gj = fma(-el.Jm[j][l], Yv[l], gj);
      double a = 0.0, jy = 0.0, jyt = 0.0;
EXO_UNROLL_TREE
      for (int k = 0; k < J; ++k) {
        a = fma(el.A[j][k], Y[k][l], a);
        jy = fma(el.Jm[j][k], Y[k][l], jy);
        jyt = fma(el.Jm[l][k], Y[k][j], jyt);
      }
      state[ws.elem(c, oA + j * J + l, draw)] = a;
      state[ws.elem(c, oC + j * J + l, draw)] = 0.5 * gL * (w[j] * w[l] - 0.5 * (jy + jyt));
    }
    state[ws.elem(c, ob + j, draw)] = gj;
    state[ws.elem(c, oeta + j, draw)] = gL * w[j];
  }
}

// (B'), part 2 -- the chain itself, last chunk to first, one draw: two J x J products per chunk
//   Fbar = local + Abar^T Fbar',   Pbar = local + Abar^T Pbar' Abar + sym(Abar^T Fbar' g^T)
template <int J>
EXO_HD void bscan_vjp_lane(int64_t n, int64_t n_draw, double* EXO_RESTRICT state, const ChunkGeom& cg, int64_t draw) {
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J;
  double mb[J], Pb[J][J];   // adjoint of (F, P) entering chunk c + 1
#pragma unroll
  for (int j = 0; j < J; ++j) {
    mb[j] = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) Pb[j][l] = 0.0;
  }
#pragma unroll 1
  for (int c = cg.C - 1; c >= 0; --c) {
    // what chunk c's reverse recurrence starts from: adjoint of (F, S) entering chunk c + 1; Sbar = -Pbar
#pragma unroll
    for (int j = 0; j < J; ++j) {
      state[ws.bnd(2, c, j, draw)] = mb[j];
#pragma unroll
      for (int l = 0; l < J; ++l) state[ws.bnd(2, c, J + j * J + l, draw)] = -Pb[j][l];
    }
    if (c == 0) break;
    double Ab[J][J], g[J], x[J], T[J][J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      g[j] = state[ws.elem(c, ob + j, draw)];
#pragma unroll
      for (int l = 0; l < J; ++l) Ab[j][l] = state[ws.elem(c, oA + j * J + l, draw)];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double xj = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        xj = fma(Ab[l][j], mb[l], xj);
        double tv = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) tv = fma(Pb[j][k], Ab[k][l], tv);
        T[j][l] = tv;
      }
      x[j] = xj;
    }
    double mbn[J], Pbn[J][J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      mbn[j] = state[ws.elem(c, oeta + j, draw)] + x[j];
#pragma unroll
      for (int l = 0; l < J; ++l) {
        double cong = state[ws.elem(c, oC + j * J + l, draw)];
#pragma unroll
        for (int k = 0; k < J; ++k) cong = fma(Ab[k][j], T[k][l], cong);
        Pbn[j][l] = cong + 0.5 * (x[j] * g[l] + g[j] * x[l]);
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      mb[j] = mbn[j];
#pragma unroll
      for (int l = 0; l < J; ++l) Pb[j][l] = 0.5 * (Pbn[j][l] + Pbn[l][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The scans (B), (B') as TREES of element compositions, one lane per composition.
//
// Filtering elements (Sarkka & Garcia-Fernandez 2021, Lemma 8; element 1 acts first):  with M = I + C1 J2,
// X1 = M^-1 A1,  x2 = M^-1 (b1 + C1 eta2),  X3 = M^-1 C1,  N = I - J2 X3
//     A = A2 X1        b = A2 x2 + b2        C = A2 X3 A2^T + C2
//     eta = A1^T N (eta2 - J2 b1) + eta1     J = A1^T N J2 A1 + J1
// and applying an element to a STATE (F, P) is bscan_lane's step.  Adjoint elements (badj_prep_lane: Abar in the A
// slot, g in b, local Fbar in eta, local Pbar in Cm) act on an adjoint state (Fbar, Pbar) as bscan_vjp_lane's step,
//     Fbar' = lF + Abar^T Fbar,   Pbar' = lP + Abar^T Pbar Abar + sym(Abar^T Fbar g^T),
// and compose (1 first) to
//     Abar = Abar1 Abar2    g = g2 + Abar2^T g1    lF = lF2 + Abar2^T lF1    lP = lP2 + Abar2^T lP1 Abar2 + sym(Abar2^T lF1 g2^T).
// Positions p = 0 .. C - 1 (forward: chunk p; adjoint: chunk C - 1 - p); element p takes the state at p to p + 1.
// UP: level f + 1 element i = elements 2i, 2i + 1 of level f composed, until one position is left, which holds the
// initial state; DOWN: state 2i of level f = state i of level f + 1, state 2i + 1 = element 2i of level f applied to it.
// 2 log2 C short launches with (positions x draws) lanes each instead of C dependent steps of one lane per draw.
// An item runs once, not in a loop: its ~3 J^2 doubles may spill without consequence (cf. badj_prep_lane).
// ---------------------------------------------------------------------------------------------
struct TreeOp {
  int J;
  int64_t n_draw;
  int64_t src_elem;    // elements read (UP: the pairs; DOWN: the child level's)
  int src_n;           //   positions p >= src_n hold the identity
  int src_rev;         //   position p is stored at index src_len - 1 - p (adjoint scan, level 0)
  int src_len;
  int64_t dst_elem;    // UP: elements written, [0, n_item)
  int64_t par_state;   // DOWN: parent states, [0, n_item)
  int64_t dst_state;   // DOWN: child states, positions [0, dst_n), stored like src (dst_rev, dst_len)
  int dst_n, dst_rev, dst_len;
  double psign;        // DOWN: the sign the matrix part of the child states is stored with
  int n_item;
};

// element `pos` of a level ([index][A, b, Cm, eta, Jm][draw]); identity past the end
template <int J>
EXO_HD void tree_load_elem(const double* EXO_RESTRICT state, const TreeOp& op, int pos, int64_t draw, Elem<J>& el) {
  const bool has = pos >= 0 && pos < op.src_n;
  const int idx = op.src_rev ? op.src_len - 1 - pos : pos;
  const int64_t E = 3 * J * J + 2 * J;
  const double* EXO_RESTRICT p = state + op.src_elem + ((int64_t)(has ? idx : 0) * E) * op.n_draw + draw;
  int e = 0;
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) { const double v = p[(e++) * op.n_draw]; el.A[j][l] = has ? v : (j == l ? 1.0 : 0.0); }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) { const double v = p[(e++) * op.n_draw]; el.b[j] = has ? v : 0.0; }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) { const double v = p[(e++) * op.n_draw]; el.Cm[j][l] = has ? v : 0.0; }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) { const double v = p[(e++) * op.n_draw]; el.eta[j] = has ? v : 0.0; }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) { const double v = p[(e++) * op.n_draw]; el.Jm[j][l] = has ? v : 0.0; }
}

// an element applied to a state: forward -- bscan_lane's step on (F, P); adjoint -- bscan_vjp_lane's on (Fbar, Pbar)
template <int J, bool ADJ>
EXO_HD void tree_apply(const Elem<J>& el, const double* m, const double (*P)[J], double* m2, double (*Ps)[J]) {
  double P2[J][J];
  if (ADJ) {
    // x = Abar^T Fbar ;  Fbar' = lF + x ;  Pbar' = lP + Abar^T Pbar Abar + sym(x g^T)
    double x[J], T[J][J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double xj = 0.0;
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        xj = fma(el.A[l][j], m[l], xj);
        double tv = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) tv = fma(P[j][k], el.A[k][l], tv);
        T[j][l] = tv;
      }
      x[j] = xj;
    }
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      m2[j] = el.eta[j] + x[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double cong = el.Cm[j][l];
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) cong = fma(el.A[k][j], T[k][l], cong);
        P2[j][l] = cong + 0.5 * (x[j] * el.b[l] + el.b[j] * x[l]);
      }
    }
  } else {
    // X = I + P Jm ;  solve X [YP | ym] = [P | F + P eta] ;  F' = A ym + b ;  P' = A (YP) A^T + Cm
    double X[J][J], Bm[J][J + 1];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double pe = m[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double xv = (j == l) ? 1.0 : 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) xv = fma(P[j][k], el.Jm[k][l], xv);
        X[j][l] = xv;
        Bm[j][l] = P[j][l];
        pe = fma(P[j][l], el.eta[l], pe);
      }
      Bm[j][J] = pe;
    }
    solve_inplace<J, J + 1>(X, Bm);
    double AY[J][J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double mj = el.b[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        mj = fma(el.A[j][l], Bm[l][J], mj);
        double v = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) v = fma(el.A[j][k], Bm[k][l], v);
        AY[j][l] = v;
      }
      m2[j] = mj;
    }
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double v = el.Cm[j][l];
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) v = fma(AY[j][k], el.A[l][k], v);
        P2[j][l] = v;
      }
  }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) Ps[j][l] = 0.5 * (P2[j][l] + P2[l][j]);
}

// two elements composed (e1 acts first), matrices symmetrised as the level arrays hold them: what the next level loads
template <int J, bool ADJ>
EXO_HD void tree_compose(const Elem<J>& e1, const Elem<J>& e2, Elem<J>& out) {
  if (ADJ) {
    double v[J];   // Abar2^T lF1
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double gj = e2.b[j], vj = 0.0;
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        gj = fma(e2.A[l][j], e1.b[l], gj);
        vj = fma(e2.A[l][j], e1.eta[l], vj);
        double a = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) a = fma(e1.A[j][k], e2.A[k][l], a);
        out.A[j][l] = a;
      }
      out.b[j] = gj;
      v[j] = vj;
      out.eta[j] = e2.eta[j] + vj;
    }
    double T[J][J];   // lP1 Abar2
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double tv = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) tv = fma(e1.Cm[j][k], e2.A[k][l], tv);
        T[j][l] = tv;
      }
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double cv = e2.Cm[j][l] + 0.5 * (v[j] * e2.b[l] + e2.b[j] * v[l]);
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) cv = fma(e2.A[k][j], T[k][l], cv);
        out.Cm[j][l] = cv;
        out.Jm[j][l] = 0.0;
      }
  } else {
    double M[J][J], R[J][2 * J + 1];   // right-hand sides [A1 | C1 | b1 + C1 eta2]
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double r2 = e1.b[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double mv = (j == l) ? 1.0 : 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) mv = fma(e1.Cm[j][k], e2.Jm[k][l], mv);
        M[j][l] = mv;
        R[j][l] = e1.A[j][l];
        R[j][J + l] = e1.Cm[j][l];
        r2 = fma(e1.Cm[j][l], e2.eta[l], r2);
      }
      R[j][2 * J] = r2;
    }
    solve_inplace<J, 2 * J + 1>(M, R);   // R = [X1 | X3 | x2]
    double AX3[J][J], Nm[J][J], w[J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double bj = e2.b[j], wj = e2.eta[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        bj = fma(e2.A[j][l], R[l][2 * J], bj);
        wj = fma(-e2.Jm[j][l], e1.b[l], wj);
        double a = 0.0, ax = 0.0, nv = (j == l) ? 1.0 : 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) {
          a = fma(e2.A[j][k], R[k][l], a);
          ax = fma(e2.A[j][k], R[k][J + l], ax);
          nv = fma(-e2.Jm[j][k], R[k][J + l], nv);
        }
        out.A[j][l] = a;
        AX3[j][l] = ax;
        Nm[j][l] = nv;
      }
      out.b[j] = bj;
      w[j] = wj;      // eta2 - J2 b1
    }
    double Nw[J], NJ[J][J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double nw = 0.0;
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        nw = fma(Nm[j][l], w[l], nw);
        double v = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) v = fma(Nm[j][k], e2.Jm[k][l], v);
        NJ[j][l] = v;
      }
      Nw[j] = nw;
    }
    double NJA[J][J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        double v = 0.0;
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) v = fma(NJ[j][k], e1.A[k][l], v);
        NJA[j][l] = v;
      }
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      double ej = e1.eta[j];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) {
        ej = fma(e1.A[l][j], Nw[l], ej);
        double cv = e2.Cm[j][l], jv = e1.Jm[j][l];
EXO_UNROLL_TREE
        for (int k = 0; k < J; ++k) {
          cv = fma(AX3[j][k], e2.A[l][k], cv);
          jv = fma(e1.A[k][j], NJA[k][l], jv);
        }
        out.Cm[j][l] = cv;
        out.Jm[j][l] = jv;
      }
      out.eta[j] = ej;
    }
  }
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = j + 1; l < J; ++l) {
      const double cs = 0.5 * (out.Cm[j][l] + out.Cm[l][j]), js = ADJ ? 0.0 : 0.5 * (out.Jm[j][l] + out.Jm[l][j]);
      out.Cm[j][l] = out.Cm[l][j] = cs;
      out.Jm[j][l] = out.Jm[l][j] = js;
    }
  if (ADJ) {
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) out.Jm[j][j] = 0.0;
  }
}

// element `idx` of a level array written ([index][A, b, Cm, eta, Jm][draw])
template <int J>
EXO_HD void tree_store_elem(double* EXO_RESTRICT state, int64_t dst_elem, int64_t nd, int idx, int64_t draw, const Elem<J>& out) {
  const int E = 3 * J * J + 2 * J;
  double* EXO_RESTRICT q = state + dst_elem + ((int64_t)idx * E) * nd + draw;
  int e = 0;
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) q[(int64_t)(e++) * nd] = out.A[j][l];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) q[(int64_t)(e++) * nd] = out.b[j];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) q[(int64_t)(e++) * nd] = out.Cm[j][l];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) q[(int64_t)(e++) * nd] = out.eta[j];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j)
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) q[(int64_t)(e++) * nd] = out.Jm[j][l];
}

// one item of a level: UP -- dst element c = src elements 2c, 2c + 1 composed; DOWN -- child states 2c, 2c + 1 from
// parent state c and child element 2c
template <int J, bool ADJ, bool DOWN>
EXO_HD void tree_item_lane(const TreeOp& op, double* EXO_RESTRICT state, int c, int64_t draw) {
  const int64_t nd = op.n_draw;
  const int Bq = J + J * J, E = 3 * J * J + 2 * J;
  if (DOWN) {
    double m[J], P[J][J];
EXO_UNROLL_TREE
    for (int j = 0; j < J; ++j) {
      m[j] = state[op.par_state + ((int64_t)c * Bq + j) * nd + draw];
EXO_UNROLL_TREE
      for (int l = 0; l < J; ++l) P[j][l] = state[op.par_state + ((int64_t)c * Bq + J + j * J + l) * nd + draw];
    }
    auto put = [&](int pos, const double* mv, const double (*Pv)[J]) {
      const int idx = op.dst_rev ? op.dst_len - 1 - pos : pos;
      double* EXO_RESTRICT q = state + op.dst_state + ((int64_t)idx * Bq) * nd + draw;
EXO_UNROLL_TREE
      for (int j = 0; j < J; ++j) {
        q[(int64_t)j * nd] = mv[j];
EXO_UNROLL_TREE
        for (int l = 0; l < J; ++l) q[(int64_t)(J + j * J + l) * nd] = op.psign * Pv[j][l];
      }
    };
    put(2 * c, m, P);
    if (2 * c + 1 >= op.dst_n) return;
    Elem<J> el;
    tree_load_elem<J>(state, op, 2 * c, draw, el);
    double m2[J], Ps[J][J];
    tree_apply<J, ADJ>(el, m, P, m2, Ps);
    put(2 * c + 1, m2, Ps);
    return;
  }
  // ---- UP
  Elem<J> e1, e2, out;
  tree_load_elem<J>(state, op, 2 * c, draw, e1);
  tree_load_elem<J>(state, op, 2 * c + 1, draw, e2);
  tree_compose<J, ADJ>(e1, e2, out);
  tree_store_elem<J>(state, op.dst_elem, nd, c, draw, out);
}

// TWO LEVELS IN ONE LAUNCH (round 5, J <= 2).  A narrow level costs its item's dependent latency -- loads, a composition, stores:
// ~5.5 us at C3 whatever its size -- and a scan has 2 log2 C of them.  An item of the radix-4 form takes four elements of level f,
// composes the two pairs (level f + 1, stored: the way down needs them) and those two (level f + 2) with the intermediate
// elements in registers; on the way down a state of level f + 2 becomes the four of level f through those of level f + 1, which
// nobody else reads and which are not stored.  The same compositions and applications on the same numbers as two radix-2
// launches -- the results are bit-identical -- in half the launches.  `a`: the op of the lower level of the pair as scan_level_op
// gives it, `b`: the upper one's.
template <int J, bool ADJ>
EXO_HD void tree_item4_up_lane(const TreeOp& a, const TreeOp& b, double* EXO_RESTRICT state, int i, int64_t draw) {
  const int64_t nd = a.n_draw;
  Elem<J> e0, e1, m0, m1, out;
  tree_load_elem<J>(state, a, 4 * i, draw, e0);
  tree_load_elem<J>(state, a, 4 * i + 1, draw, e1);
  tree_compose<J, ADJ>(e0, e1, m0);
  tree_store_elem<J>(state, a.dst_elem, nd, 2 * i, draw, m0);       // (2 i < a.n_item: i < b.n_item = ceil(a.n_item / 2))
  if (2 * i + 1 < a.n_item) {
    tree_load_elem<J>(state, a, 4 * i + 2, draw, e0);
    tree_load_elem<J>(state, a, 4 * i + 3, draw, e1);
    tree_compose<J, ADJ>(e0, e1, m1);
    tree_store_elem<J>(state, a.dst_elem, nd, 2 * i + 1, draw, m1);
  } else {
    // past the end of level f + 1: the identity, as tree_load_elem hands it to a radix-2 item
#pragma unroll
    for (int j = 0; j < J; ++j) {
      m1.b[j] = m1.eta[j] = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) { m1.A[j][l] = (j == l) ? 1.0 : 0.0; m1.Cm[j][l] = m1.Jm[j][l] = 0.0; }
    }
  }
  tree_compose<J, ADJ>(m0, m1, out);
  tree_store_elem<J>(state, b.dst_elem, nd, i, draw, out);
}

// `b`: the DOWN op of level f + 1 (parents: level f + 2), `a`: that of level f (children written as a says: level 0's order / sign)
template <int J, bool ADJ>
EXO_HD void tree_item4_down_lane(const TreeOp& a, const TreeOp& b, double* EXO_RESTRICT state, int i, int64_t draw) {
  const int64_t nd = a.n_draw;
  const int Bq = J + J * J;
  double m[J], P[J][J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    m[j] = state[b.par_state + ((int64_t)i * Bq + j) * nd + draw];
#pragma unroll
    for (int l = 0; l < J; ++l) P[j][l] = state[b.par_state + ((int64_t)i * Bq + J + j * J + l) * nd + draw];
  }
  auto put = [&](int pos, const double* mv, const double (*Pv)[J]) {
    const int idx = a.dst_rev ? a.dst_len - 1 - pos : pos;
    double* EXO_RESTRICT q = state + a.dst_state + ((int64_t)idx * Bq) * nd + draw;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      q[(int64_t)j * nd] = mv[j];
#pragma unroll
      for (int l = 0; l < J; ++l) q[(int64_t)(J + j * J + l) * nd] = a.psign * Pv[j][l];
    }
  };
  // the two children of a state of level f + 1 at position p
  auto children = [&](int p, const double* mv, const double (*Pv)[J]) {
    put(2 * p, mv, Pv);                      // (2 p < a.dst_n: p < b.dst_n = ceil(a.dst_n / 2))
    if (2 * p + 1 >= a.dst_n) return;
    Elem<J> el;
    tree_load_elem<J>(state, a, 2 * p, draw, el);
    double m2[J], Ps[J][J];
    tree_apply<J, ADJ>(el, mv, Pv, m2, Ps);
    put(2 * p + 1, m2, Ps);
  };
  Elem<J> eb;
  const bool second = 2 * i + 1 < b.dst_n;
  if (second) tree_load_elem<J>(state, b, 2 * i, draw, eb);     // (issued before the first pair's work)
  children(2 * i, m, P);
  if (!second) return;
  double m1[J], P1[J][J];
  tree_apply<J, ADJ>(eb, m, P, m1, P1);
  children(2 * i + 1, m1, P1);
}

// the narrow top of a scan as a serial chain, one lane per draw: the level's elements applied one after the other to the seed
// (`op`: the level's DOWN op with the ONE parent state).  The device runs this on groups of eight lanes for J >= 3
// (exo_celerite_group.hpp, tree_serial_group: where it pays and why); this one-lane form is what the host harness checks the
// schedule with.
template <int J, bool ADJ>
EXO_HD void tree_serial_lane(const TreeOp& op, double* EXO_RESTRICT state, int64_t draw) {
  const int64_t nd = op.n_draw;
  const int Bq = J + J * J;
  double m[J], P[J][J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    m[j] = state[op.par_state + (int64_t)j * nd + draw];
#pragma unroll
    for (int l = 0; l < J; ++l) P[j][l] = state[op.par_state + (int64_t)(J + j * J + l) * nd + draw];
  }
  auto put = [&](int pos, const double* mv, const double (*Pv)[J]) {
    const int idx = op.dst_rev ? op.dst_len - 1 - pos : pos;
    double* EXO_RESTRICT q = state + op.dst_state + ((int64_t)idx * Bq) * nd + draw;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      q[(int64_t)j * nd] = mv[j];
#pragma unroll
      for (int l = 0; l < J; ++l) q[(int64_t)(J + j * J + l) * nd] = op.psign * Pv[j][l];
    }
  };
  put(0, m, P);
  for (int pos = 0; pos + 1 < op.dst_n; ++pos) {
    Elem<J> el;
    tree_load_elem<J>(state, op, pos, draw, el);
    double m2[J], Ps[J][J];
    tree_apply<J, ADJ>(el, m, P, m2, Ps);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      m[j] = m2[j];
#pragma unroll
      for (int l = 0; l < J; ++l) P[j][l] = Ps[j][l];
    }
    put(pos + 1, m, P);
  }
}

// the state the forward scan starts from, (F, P) = (0, Delta(t_0)) (S_0 = 0), as a state record at dst
template <int J>
EXO_HD void scan_init_lane(const double* EXO_RESTRICT t, const Coefs& cf, int64_t n_draw, double* EXO_RESTRICT dst, int64_t draw) {
  DeltaCoef<J> dc;
  dc.init(cf, draw);
  DrawCoef<J> co;
  co.init(cf, draw);
  double U[J], V[J];
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) U[j] = V[j] = 0.0;
  co.uv(t[0], U, V);
  Sym<J> Dl;
  dc.eval(V, Dl);
EXO_UNROLL_TREE
  for (int j = 0; j < J; ++j) {
    dst[(int64_t)j * n_draw + draw] = 0.0;
EXO_UNROLL_TREE
    for (int l = 0; l < J; ++l) dst[(int64_t)(J + j * J + l) * n_draw + draw] = Dl(j, l);
  }
}

// The levels of a scan: `launch(op, down)` is called once per level, in order (UP levels, then -- after `seed()`
// has put the initial state at ws.tree_state(top) -- the DOWN levels).  adj: the adjoint scan (positions reversed,
// states into bnd(2, .) with the sign the chunk kernels expect); else the forward scan (states into bnd(1, .)).
// level f of a scan as a TreeOp: UP -- elements of level f composed into level f + 1 (f = 0 .. top - 2); DOWN -- states of
// level f from those of level f + 1 and the elements of level f (f = top - 1 .. 0).  adj: the adjoint scan (positions
// reversed, states into bnd(2, .) with the sign the chunk kernels expect); else the forward scan (states into bnd(1, .)).
// A pure function of the plan: the host loop below and the one-launch scan kernel (exo_celerite_group.hpp) both call it.
EXO_HDH TreeOp scan_level_op(const ChunkWs& ws, int J, bool adj, int f, bool down) {
  TreeOp op{};
  op.J = J; op.n_draw = ws.n_draw; op.psign = 1.0;
  op.src_elem = f == 0 ? ws.elem(0, 0, 0) : ws.tree_elem(f);
  op.src_n = f == 0 ? ws.C - 1 : ws.tree_npos(f);   // forward: the last chunk's element takes no state anywhere;
  op.src_rev = (adj && f == 0) ? 1 : 0;              // adjoint: elements of chunks C - 1 .. 1 at positions 0 .. C - 2
  op.src_len = ws.tree_npos(f);
  op.n_item = ws.tree_npos(f + 1);
  if (!down) {
    op.dst_elem = ws.tree_elem(f + 1);
  } else {
    op.par_state = ws.tree_state(f + 1);
    op.dst_state = f == 0 ? ws.bnd(adj ? 2 : 1, 0, 0, 0) : ws.tree_state(f);
    op.dst_n = op.dst_len = ws.tree_npos(f);
    op.dst_rev = (adj && f == 0) ? 1 : 0;
    op.psign = (adj && f == 0) ? -1.0 : 1.0;
  }
  return op;
}

// The levels of a scan: `launch(op, down)` is called once per level, in order (UP levels, then -- after `seed()`
// has put the initial state at ws.tree_state(top) -- the DOWN levels).
template <class Launch, class Seed>
EXO_HDH void tree_scan(const ChunkWs& ws, int J, bool adj, Launch&& launch, Seed&& seed) {
  const int top = ws.tree_top();
  for (int f = 0; f + 1 < top; ++f) launch(scan_level_op(ws, J, adj, f, false), false);
  seed();
  for (int f = top - 1; f >= 0; --f) launch(scan_level_op(ws, J, adj, f, true), true);
}

// the same with two levels per launch where a pair is left (tree_item4_*_lane): launch2(lower op, upper op, down)
template <class Launch, class Launch2, class Seed>
EXO_HDH void tree_scan4(const ChunkWs& ws, int J, bool adj, Launch&& launch, Launch2&& launch2, Seed&& seed) {
  const int top = ws.tree_top();
  int f = 0;
  for (; f + 2 < top; f += 2) launch2(scan_level_op(ws, J, adj, f, false), scan_level_op(ws, J, adj, f + 1, false), false);
  for (; f + 1 < top; ++f) launch(scan_level_op(ws, J, adj, f, false), false);
  seed();
  f = top - 1;
  for (; f >= 1; f -= 2) launch2(scan_level_op(ws, J, adj, f - 1, true), scan_level_op(ws, J, adj, f, true), true);
  for (; f >= 0; --f) launch(scan_level_op(ws, J, adj, f, true), true);
}

// the level from which a scan of J >= 3 runs as a serial chain (tree_serial_group): the lowest with at most kSerialTopGroup positions;
// the top level (= no chain) where that would replace fewer than two levels
#ifndef EXO_GP_SERIAL_TOP_GROUP
#define EXO_GP_SERIAL_TOP_GROUP 8     // (measured at C5: 8 -> 1.825 ms, 16 -> 1.85, 32 -> 1.91, none 1.86-1.875)
#endif
EXO_HDH int tree_serial_level(const ChunkWs& ws, int J) {
  const int top = ws.tree_top();
  if (J < 3 || EXO_GP_SERIAL_TOP_GROUP < 2) return top;
  int f = 0;
  while (ws.tree_npos(f) > EXO_GP_SERIAL_TOP_GROUP) ++f;
  return f + 2 <= top ? f : top;
}
// tree_scan with the levels from f0 up replaced by the chain: serial(op)
template <class Launch, class Seed, class Serial>
EXO_HDH void tree_scan_top(const ChunkWs& ws, int J, bool adj, int f0, Launch&& launch, Seed&& seed, Serial&& serial) {
  const int top = ws.tree_top();
  if (f0 >= top) { tree_scan(ws, J, adj, launch, seed); return; }
  for (int f = 0; f < f0; ++f) launch(scan_level_op(ws, J, adj, f, false), false);   // (the elements of level f0)
  seed();
  TreeOp op = scan_level_op(ws, J, adj, f0, true);
  op.par_state = ws.tree_state(top);     // the one state the chain starts from
  serial(op);
  for (int f = f0 - 1; f >= 0; --f) launch(scan_level_op(ws, J, adj, f, true), true);
}

// ---------------------------------------------------------------------------------------------
// (C) / (C') on one lane: the ordinary recurrences of one (draw, chunk), all J state indices in
// the lane's registers, with a checkpointed factorisation.
// ---------------------------------------------------------------------------------------------
// the recurrence state at one cadence, as the reverse pass needs it
template <int J>
struct Step {
  Sym<J> S;
  double F[J], V[J];
  double d, z;
};

// One cadence of the forward recurrences.  On entry (S, F) is the state AFTER the propagation into
// this cadence (for the first cadence of a chunk / of a checkpoint block: the entering state);
// on exit W, d, z are this cadence's; `advance` then propagates (S, F) to the next cadence.
template <int J>
struct Fwd {
  Sym<J> S;
  double F[J], W[J], U[J], V[J];
  double d, z, id;   // id = 1 / d
  EXO_HD void measure(double yi, double diag_plus_asum) {
    double u[J];
    double pd = 0.0, pz = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double uj = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) uj = fma(S(j, l), U[l], uj);
      u[j] = uj;
      pd = fma(U[j], uj, pd);
      pz = fma(U[j], F[j], pz);
    }
    d = diag_plus_asum - pd;
    z = yi - pz;
    id = exo::fast_rcp(d);   // seed + two Newton steps: 1 ulp, a third of the IEEE division sequence
#pragma unroll
    for (int j = 0; j < J; ++j) W[j] = (V[j] - u[j]) * id;
  }
  EXO_HD void advance(const double* phi) {
#pragma unroll
    for (int j = 0; j < J; ++j) F[j] = phi[j] * fma(W[j], z, F[j]);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = j; l < J; ++l) S(j, l) = phi[j] * phi[l] * fma(d * W[j], W[l], S(j, l));
  }
};

// propagators exp(-c_j dt), recomputed only when dt changes (evenly sampled series reuse them)
template <int J>
struct Phi {
  double v[J];
  double dt_prev;
  EXO_HD Phi() : dt_prev(-1.0) {
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = 1.0;
  }
  template <int NR>
  EXO_HD void set(const DrawCoef<J, NR>& co, double dt) {
    if (dt != dt_prev) {
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = exp(-co.k[j].c * dt);
      dt_prev = dt;
    }
  }
};

// (C) forward: acc = sum z^2 / d, log det and the "not positive definite" mark of the chunk go to the
// chunk partials; (F, S) at the first cadence of every checkpoint block goes to the checkpoints.
#ifndef EXO_FWD_PREFETCH_J2
#define EXO_FWD_PREFETCH_J2 1     // (0 = no look-ahead load in the J <= 2 forward sweep: measured, no gain at four waves per SIMD -- 707 against
                                  // 711 us at C3 --, slower at five -- 794: scratch)
#endif
template <int J, int NR = -1, int SP = -1>
EXO_HD void chunk1_fwd_lane(const double* EXO_RESTRICT t, Series rs, const double* EXO_RESTRICT diag, int64_t n_diag,
                            int64_t n, const Coefs& cf, int64_t n_draw, double* EXO_RESTRICT state, const ChunkGeom& cg,
                            int64_t draw, int c, bool save, int polish = 0) {
  // POLISH (round 4).  The scans hand every chunk the state it is entered with, right to kappa x 1e-16 -- which for a
  // conditioning score kappa of 1e5 .. 1e8 is what the gradients' tail was made of, and why such draws were REDONE by the
  // sequential kernels (~50 x the step, for the whole batch).  But the recurrences of a chunk forget the state they
  // were entered with (the filter is a contraction: the better the data, the faster), so the state they LEAVE for the
  // next chunk is far more accurate than the one they were given: pass 0 stores it (ChunkWs::polish), and a second pass
  // over the draws that need it -- polish = 1 -- enters every chunk with what its predecessor left.  One Jacobi sweep of
  // the exact recurrences, all chunks at once; the scans have become the initial guess.
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  if (cg.which && (state[ws.off_flag() + draw] == kFlagRobust) != (cg.which == 1)) return;   // (ChunkGeom::which)
  DrawCoef<J, NR> co;
  co.init(cf, draw);
  const double asum = co.asum();
  const SeriesRowT<true, SP> y(rs, draw, n);
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  Fwd<J> f;
#pragma unroll
  for (int j = 0; j < J; ++j) { f.U[j] = f.V[j] = f.W[j] = 0.0; }
  if (EXO_GP_POLISH && polish && c > 0) {
    co.uv(t[n0], f.U, f.V);
#pragma unroll
    for (int j = 0; j < J; ++j) f.F[j] = state[ws.polish((polish & 1) ? 0 : 2, c - 1, j, draw)];
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) f.S.v[k] = state[ws.polish((polish & 1) ? 0 : 2, c - 1, J + k, draw)];
  } else {
    // entering state from (B) as (F, P): S = Delta_{n0} - P
    DeltaCoef<J, NR> dc;
    dc.init(cf, draw);
    co.uv(t[n0], f.U, f.V);
    Sym<J> Dl;
    dc.eval(f.V, Dl);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      f.F[j] = state[ws.bnd(1, c, j, draw)];
#pragma unroll
      for (int l = j; l < J; ++l) f.S(j, l) = Dl(j, l) - state[ws.bnd(1, c, J + j * J + l, draw)];
    }
  }
  double phi[J];
  double tprev = t[n0];
  double acc = 0.0, lman = 1.0;
  int64_t lsum = 0;
  bool bad = false;
  static_assert(kCkptB == 4, "row_load4 / row_store4 move one checkpoint block");
  BlockIn cur, nxt;
  load_block(y, dg, n_diag, n0, n1, cur);
#pragma unroll 1
  for (int64_t b0 = n0; b0 < n1; b0 += kCkptB) {
    constexpr bool kAhead = J > 2 || EXO_FWD_PREFETCH_J2;
    if (kAhead) load_block(y, dg, n_diag, b0 + kCkptB, n1, nxt);   // in flight while this block is worked through
#pragma unroll
    for (int q = 0; q < kCkptB; ++q) {
      const int64_t i = b0 + q;
      if (i < n1) {
        if (i > n0) {   // (the entering state is already at n0)
          const double ti = t[i], dt = ti - tprev;
          const bool near = co.step(dt, phi);
          tprev = ti;
          f.advance(phi);
          if (q == 0 || !near) {
            co.uv(ti, f.U, f.V);
          } else {
            double Vp[J];
#pragma unroll
            for (int j = 0; j < J; ++j) Vp[j] = f.V[j];
            co.rot_uv(dt, Vp, f.U, f.V);
          }
        }
        if (q % ckpt_span(J) == 0 && save) {   // checkpoint: the state AT the span's first cadence (after the step into it)
          const int64_t g = i / ckpt_span(J);
#if defined(__HIP_DEVICE_COMPILE__) && EXO_CKPT_NT_STORE
#pragma unroll
          for (int j = 0; j < J; ++j) __builtin_nontemporal_store(f.F[j], state + ws.ckpt(g, j, draw));
#pragma unroll
          for (int k = 0; k < J * (J + 1) / 2; ++k) __builtin_nontemporal_store(f.S.v[k], state + ws.ckpt(g, J + k, draw));
#else
#pragma unroll
          for (int j = 0; j < J; ++j) state[ws.ckpt(g, j, draw)] = f.F[j];
#pragma unroll
          for (int k = 0; k < J * (J + 1) / 2; ++k) state[ws.ckpt(g, J + k, draw)] = f.S.v[k];
#endif
        }
        f.measure(cur.y[q], cur.g[q] + asum);
        bad = bad || !(f.d > 0.0);
        acc = fma(f.z * f.z, f.id, acc);
        int lexp;
        lman = frexp(lman * (f.d > 0.0 ? f.d : 1.0), &lexp);
        lsum += lexp;
      }
    }
    if (J > 2 || EXO_FWD_PREFETCH_J2) cur = nxt;
    else load_block(y, dg, n_diag, b0 + kCkptB, n1, cur);
  }
  state[ws.part(c, 0, draw)] = acc;
  state[ws.part(c, 1, draw)] = log(lman) + (double)lsum * 0.69314718055994530942;
  state[ws.part(c, 2, draw)] = bad ? 1.0 : 0.0;
  if (EXO_GP_POLISH && save && n1 < n) {
    // what this chunk leaves for the next one: the step from its last cadence into cadence n1 (ChunkWs::polish; a polish
    // pass reads pass 0's values and writes to the other half of the double buffer `polish & 1` selects -- see the callers)
    const double dt = t[n1] - tprev;
    co.step(dt, phi);
    f.advance(phi);
    const int slot = (polish & 1) ? 2 : 0;   // (two buffers alternate: the lanes of a pass read the pass before while they write)
#pragma unroll
    for (int j = 0; j < J; ++j) state[ws.polish(slot, c, j, draw)] = f.F[j];
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) state[ws.polish(slot, c, J + k, draw)] = f.S.v[k];
  }
}

// d loglike / d(d_pair), the oscillation rate of a complex pair, ACROSS CHUNKS (round 4).  The recurrences carry absolute
// phases d t_i, so the cotangent of d is sum_i t_i g_i, g_i the phase cotangent of cadence i -- and, the likelihood
// depending on phase DIFFERENCES only, sum_i g_i = 0: the sum is a small difference of terms of size t |g|.  In one
// sequential sweep that cancellation is consistent to rounding.  Across chunks it is not: every chunk's reverse sweep
// starts from an adjoint the scans supplied and runs on a state the scans supplied, each right to kappa x 1e-16 on its own,
// and t_c x (that mismatch) stayed behind -- the conditioning tail of tools/gp_cond_bins.py was THIS term (every other
// gradient agreed to 1e-10; the error grew with the number of chunks and with the origin of the time axis: 100 spans away,
// BTJD-style time stamps, it reached 1e-3).  Summed by parts the cotangent is  - sum_i (t_{i+1} - t_i) Phi_i  with the phase
// FLUX across the link (i, i + 1), Phi_i = sum_{i' <= i} g_i': local time differences only.  A chunk's reverse sweep walks
// its links last to first, Phi_{i-1} = Phi_i - g_i, and STARTS from the flux across its end boundary taken not from any
// sum but from what it is: shifting every earlier phase by eps turns the pair's components of the state entering the next
// chunk by eps -- F -> R F, S -> R S R^T -- hence
//     Phi = Fbar^T G F + < Sbar, G S + S G^T >,   G = [[0, -1], [1, 0]] on the pair,
// with the entering state (the next chunk's first checkpoint) and its adjoint (what the adjoint scan hands this chunk).
template <int J, class CoefT, class SbT>
EXO_HD double phase_flux(const CoefT& co, int j, const double* F, const Sym<J>& S, const double* Fb, const SbT& Sb) {
  const int jn = j + 1 < J ? j + 1 : j;
  double acc = Fb[jn] * F[j] - Fb[j] * F[jn];
#pragma unroll
  for (int l = 0; l < J; ++l) acc = fma(2.0, Sb(jn, l) * S(j, l) - Sb(j, l) * S(jn, l), acc);
  (void)co;
  return acc;
}

// (C') reverse.  Hand-derived adjoint of the two recurrences (same algebra as celerite_vjp_kernel
// of exo_celerite.hip, one lane holding every state index): Sb is the SYMMETRISED adjoint of S.
template <int J, int NR = -1>
struct Rev {
  double Sb[J][J], Fb[J], Wb[J];
  double db, zb, gasum;
  double ga[J], gb[J], gc[J], gd[J];
  double flux[J];   // phase flux across the link behind the cadence being reversed (phase_flux), per pair (first index)

  // measurement half of cadence i: adjoints of (d, z, W) of this cadence -> (S, F) of this cadence
  // and the coefficient cotangents through U, V.  W = (V - S U) / d is rebuilt (the step keeps V).
  // Returns zbar (d loglike / d y_i) and dbar (d loglike / d diag_i).
  // dt_next: t_{i+1} - t_i, the link behind this cadence (0: the series ends here)
  // W = (V - S U) / d of a cadence from its saved state, with the pieces the measurement half needs again (U, u = S U, 1 / d).
  // Round 6: computed ONCE per cadence -- for the reverse of the step out of it -- and handed to measure_w; it used to be
  // rebuilt inside measure() as well (u_from_v, a reciprocal, a J x J product: ~8 % of the reverse kernel's instructions).
  struct WOf {
    double U[J], u[J], W[J], id;
  };
  EXO_HD static void w_of(const DrawCoef<J, NR>& co, const Step<J>& s, WOf& w) {
#pragma unroll
    for (int j = 0; j < J; ++j) w.U[j] = 0.0;
    co.u_from_v(s.V, w.U);
    w.id = exo::fast_rcp(s.d);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double uj = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) uj = fma(s.S(j, l), w.U[l], uj);
      w.u[j] = uj;
      w.W[j] = (s.V[j] - uj) * w.id;
    }
  }
  EXO_HD void measure(const DrawCoef<J, NR>& co, const Step<J>& s, double dt_next, double gL, double* W_out, double* zbar_out,
                      double* dbar_out) {
    WOf w;
    w_of(co, s, w);
#pragma unroll
    for (int j = 0; j < J; ++j) W_out[j] = w.W[j];
    measure_w(co, s, w, dt_next, gL, zbar_out, dbar_out);
  }
  EXO_HD void measure_w(const DrawCoef<J, NR>& co, const Step<J>& s, const WOf& w, double dt_next, double gL, double* zbar_out,
                        double* dbar_out) {
    const double* U = w.U;
    const double* u = w.u;
    const double id = w.id;
    double wdot = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) wdot = fma(Wb[j], w.W[j], wdot);
    const double zbar = zb - gL * s.z * id;
    const double dbar = db + gL * (0.5 * s.z * s.z * id * id - 0.5 * id) - wdot * id;
    *zbar_out = zbar;
    *dbar_out = dbar;
    gasum += dbar;
    double Ub[J], Vb[J], ub[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Ub[j] = fma(-dbar, u[j], -zbar * s.F[j]);
      Fb[j] = fma(-zbar, U[j], Fb[j]);
      Vb[j] = Wb[j] * id;
      ub[j] = -Vb[j] - dbar * U[j];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double acc_u = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        Sb[j][l] = fma(0.5, fma(ub[j], U[l], ub[l] * U[j]), Sb[j][l]);   // symmetrised  ub U^T
        acc_u = fma(s.S(j, l), ub[l], acc_u);                            // (S^T ub)_j, S symmetric
      }
      Ub[j] += acc_u;
    }
    // coefficient cotangents: a real term's U = a; a complex pair's (a, b, d) collect on its first index
    // (accumulators updated unconditionally, the layout deciding the increment: see DeltaCoef::eval)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool re = co.is_real(j), fi = !re && co.is_first(j) && j + 1 < J;
      const int jn = j + 1 < J ? j + 1 : j;
      const double cs = s.V[j], sn = s.V[jn];
      const double a = co.k[j].a, b = co.k[j].b;
      ga[j] += re ? Ub[j] : (fi ? Ub[j] * cs + Ub[jn] * sn : 0.0);
      gb[j] += fi ? Ub[j] * sn - Ub[jn] * cs : 0.0;
      // d: the link behind this cadence carries the flux so far; this cadence's phase cotangent leaves it (phase_flux)
      const double gph = fi ? Ub[j] * (-a * sn + b * cs) + Ub[jn] * (a * cs + b * sn) - Vb[j] * sn + Vb[jn] * cs : 0.0;
      gd[j] = fma(-dt_next, flux[j], gd[j]);
      flux[j] -= gph;
    }
  }

  // reverse of the step p -> n (p = n - 1):  F_n = P o (F_p + W_p z_p),  S_n = P P^T o (S_p + d_p W_p W_p^T);
  // on entry Sb, Fb are the adjoints of S_n, F_n; on exit those of S_p, F_p, and Wb, db, zb those of
  // W_p, d_p, z_p.  Wp = W of cadence p.
  EXO_HD void propagate(const Step<J>& p, const double* Wp, const double* phi, double dt) {
    double Gb[J], Pb[J], Wbn[J];
    double zbn = 0.0, dbn = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const double Gj = fma(Wp[j], p.z, p.F[j]);
      Pb[j] = Fb[j] * Gj;
      Gb[j] = Fb[j] * phi[j];
      Wbn[j] = Gb[j] * p.z;
      zbn = fma(Gb[j], Wp[j], zbn);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double psum = 0.0, wsum = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        const double T = fma(p.d * Wp[j], Wp[l], p.S(j, l));
        const double Tb = Sb[j][l] * phi[j] * phi[l];      // adjoint of T (symmetric)
        psum = fma(2.0 * Sb[j][l] * T, phi[l], psum);
        wsum = fma(Tb, Wp[l], wsum);
        Sb[j][l] = Tb;                                     // becomes the adjoint of S_p
      }
      Pb[j] += psum;
      Wbn[j] = fma(2.0 * p.d, wsum, Wbn[j]);
      dbn = fma(wsum, Wp[j], dbn);
      gc[j] = fma(-dt * phi[j], Pb[j], gc[j]);
    }
    db = dbn; zb = zbn;
#pragma unroll
    for (int j = 0; j < J; ++j) { Fb[j] = Gb[j]; Wb[j] = Wbn[j]; }
  }
};

// gsign: +1 writes d loglike / d resid; -1 writes d loglike / d model (obs - model series)
template <int J, int NR = -1, int SP = -1>
EXO_HD void chunk1_vjp_lane(const double* EXO_RESTRICT t, Series rs, const double* EXO_RESTRICT diag, int64_t n_diag,
                            int64_t n, const Coefs& cf, int64_t n_draw, const double* EXO_RESTRICT gloglike,
                            double* EXO_RESTRICT state, const ChunkGeom& cg, double* EXO_RESTRICT gresid,
                            double* EXO_RESTRICT gdiag, double gsign, int64_t draw, int c, int polish = 0) {
  // (polish: the adjoint this chunk is entered with is what the NEXT chunk's reverse recurrences left in pass 0 -- the
  // adjoint of the state entering it -- instead of the adjoint scan's: chunk1_fwd_lane has the reasoning)
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  DrawCoef<J, NR> co;
  co.init(cf, draw);
  const double asum = co.asum();
  const SeriesRowT<false, SP> y(rs, draw, n);
  GradRowT<SP> grow(gresid, rs, draw, n);
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  const double gL = gloglike[cf.at(draw)];
  Rev<J, NR> r;
  const bool from_next = EXO_GP_POLISH && polish && n1 < n;
  const double gsc = cg.prep ? gL : 1.0;   // (ChunkGeom::prep: the scan's adjoints are those of a cotangent of one)
#pragma unroll
  for (int j = 0; j < J; ++j) {
    r.Fb[j] = from_next ? state[ws.polish((polish & 1) ? 1 : 3, c + 1, j, draw)] : gsc * state[ws.bnd(2, c, j, draw)];
    r.Wb[j] = 0.0;
    r.ga[j] = r.gb[j] = r.gc[j] = r.gd[j] = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l)
      r.Sb[j][l] = from_next ? state[ws.polish((polish & 1) ? 1 : 3, c + 1, J + Sym<J>::idx(j, l), draw)]
                             : gsc * state[ws.bnd(2, c, J + j * J + l, draw)];
  }
  r.db = r.zb = r.gasum = 0.0;
  double phi[J];
  {
    // the phase flux across the chunk's end boundary: the state entering the next chunk (its first checkpoint) against the
    // adjoint the scan handed over (phase_flux); nothing flows out of the series' end
    double ckn[J + J * (J + 1) / 2];
#pragma unroll
    for (int k = 0; k < J + J * (J + 1) / 2; ++k) ckn[k] = (n1 < n) ? state[ws.ckpt(n1 / kCkptB, k, draw)] : 0.0;
    Sym<J> Sn;
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) Sn.v[k] = ckn[J + k];
    auto SbAt = [&](int a, int b) { return r.Sb[a][b]; };
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool fi = !co.is_real(j) && co.is_first(j) && j + 1 < J;
      const double fl = phase_flux<J>(co, j, ckn, Sn, r.Fb, SbAt);
      r.flux[j] = fi ? fl : 0.0;
    }
  }
  if (n0 + 1 < n) co.set_ref(t[n0 + 1] - t[n0]);   // the forward kernel's reference: its first step
  // blocks last to first; `pend`: the step from this block's last cadence into cadence `next` (the
  // first cadence of the block after it, or of the next chunk) still has to be reversed
  const int64_t nb = (n1 - n0 + kCkptB - 1) / kCkptB;
  bool pend = n1 < n;
  constexpr int kK = J + J * (J + 1) / 2;
  // inputs of a block: its checkpoint (F, packed S) and its cadences of the series
  BlockIn cur, nxt;
  double ck[kK], ck_nxt[kK];
  auto load_ckpt = [&](int64_t b0, double* dst) {
    const int64_t g = b0 / kCkptB;
#pragma unroll
    for (int k = 0; k < kK; ++k) dst[k] = ckpt_load(state + ws.ckpt(g, k, draw));
  };
  load_block(y, dg, n_diag, n0 + (nb - 1) * kCkptB, n1, cur);
  load_ckpt(n0 + (nb - 1) * kCkptB, ck);
  nxt = cur;
#pragma unroll
  for (int k = 0; k < kK; ++k) ck_nxt[k] = ck[k];
#pragma unroll 1
  for (int64_t bi = nb - 1; bi >= 0; --bi) {
    const int64_t b0 = n0 + bi * kCkptB;
    const int len = (int)((n1 - b0 < kCkptB) ? n1 - b0 : kCkptB);
    if (bi > 0) {   // the block before this one: in flight while this one is worked through
      load_block(y, dg, n_diag, b0 - kCkptB, n1, nxt);
      load_ckpt(b0 - kCkptB, ck_nxt);
    }
    // recompute the block forward from its checkpoint, keeping every cadence's state
    Step<J> st[kCkptB];
    double tt[kCkptB];
    {
      Fwd<J> f;
#pragma unroll
      for (int j = 0; j < J; ++j) { f.F[j] = ck[j]; f.U[j] = f.V[j] = f.W[j] = 0.0; }
#pragma unroll
      for (int k = 0; k < J * (J + 1) / 2; ++k) f.S.v[k] = ck[J + k];
      double tprev = t[b0];
#pragma unroll
      for (int q = 0; q < kCkptB; ++q) {
        if (q < len) {
          const double ti = t[b0 + q];
          tt[q] = ti;
          if (q > 0) {
            const double dt = ti - tprev;
            const bool near = co.step(dt, phi);
            tprev = ti;
            f.advance(phi);
            if (near) {
              double Vp[J];
#pragma unroll
              for (int j = 0; j < J; ++j) Vp[j] = f.V[j];
              co.rot_uv(dt, Vp, f.U, f.V);
            } else {
              co.uv(ti, f.U, f.V);
            }
          } else {
            co.uv(ti, f.U, f.V);
          }
          f.measure(cur.y[q], cur.g[q] + asum);
          st[q].S = f.S;
          st[q].d = f.d; st[q].z = f.z;
#pragma unroll
          for (int j = 0; j < J; ++j) { st[q].F[j] = f.F[j]; st[q].V[j] = f.V[j]; }
        } else {
          tt[q] = 0.0;
          st[q] = st[q > 0 ? q - 1 : 0];
        }
      }
    }
    double zbar[kCkptB], dbar[kCkptB];
    // W (and U, S U, 1 / d) of the cadence being reversed: worked out once, when the step OUT of it is reversed -- or, for the
    // block's last cadence, right here -- and carried into its measurement half (Rev::w_of)
    typename Rev<J, NR>::WOf wq;
#pragma unroll
    for (int q = kCkptB - 1; q >= 0; --q) {
      zbar[q] = dbar[q] = 0.0;
      if (q < len) {
        const int64_t i = b0 + q;
        if (q == len - 1) {
          Rev<J, NR>::w_of(co, st[q], wq);
          if (pend) {
            // the step from cadence i into cadence i + 1 (first of the block / chunk after this one)
            const double dt = t[i + 1] - tt[q];
            co.step(dt, phi, false);
            r.propagate(st[q], wq.W, phi, dt);
          }
        }
        r.measure_w(co, st[q], wq, (q + 1 < len) ? tt[q + 1] - tt[q] : (pend ? t[i + 1] - tt[q] : 0.0), gL, &zbar[q], &dbar[q]);
        if (q > 0) {
          // reverse of the step (i - 1) -> i, with W of cadence i - 1 -- which the next round's measurement half takes over
          Rev<J, NR>::w_of(co, st[q - 1], wq);
          const double dt = tt[q] - tt[q - 1];
          co.step(dt, phi, false);
          r.propagate(st[q - 1], wq.W, phi, dt);
        }
      }
    }
    pend = true;   // the step from the previous block's last cadence into b0
    // gresid / gdiag are [draw][cadence]: the block's cadences are consecutive doubles of one row (gresid cadence-major
    // with the series: GradRow)
#pragma unroll
    for (int q = 0; q < kCkptB; ++q) zbar[q] *= gsign;
    grow.store4(b0, len, zbar);
    if (gdiag) row_store4(gdiag + cf.at(draw) * n, b0, len, dbar);
    cur = nxt;
#pragma unroll
    for (int k = 0; k < kK; ++k) ck[k] = ck_nxt[k];
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    state[ws.gpart(c, 4 * j + 0, draw)] = r.ga[j];
    state[ws.gpart(c, 4 * j + 1, draw)] = r.gb[j];
    state[ws.gpart(c, 4 * j + 2, draw)] = r.gc[j];
    state[ws.gpart(c, 4 * j + 3, draw)] = r.gd[j];
  }
  state[ws.gpart(c, 4 * J, draw)] = r.gasum;
  if (EXO_GP_POLISH && c > 0) {   // the adjoint of the state this chunk was entered with: what a polish pass enters the previous chunk with
    const int slot = (polish & 1) ? 3 : 1;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      state[ws.polish(slot, c, j, draw)] = r.Fb[j];
#pragma unroll
      for (int l = j; l < J; ++l) state[ws.polish(slot, c, J + Sym<J>::idx(j, l), draw)] = 0.5 * (r.Sb[j][l] + r.Sb[l][j]);
    }
  }
}

// (C') reverse for wide states (J > 2): the same adjoint with the (symmetrised) adjoint of S PACKED, and
// (chunkp_vjp_lane) two checkpoints per block.  Hand-derived adjoint of the two recurrences (same algebra as celerite_vjp_kernel
// of exo_celerite.hip, one lane holding every state index): Sb is the SYMMETRISED adjoint of S.
// STATE_ONLY: the adjoints of the recurrence state alone (chunk_adj_lane): no coefficient cotangents, no phase flux
template <int J, int NR = -1, bool STATE_ONLY = false>
struct RevP {
  Sym<J> Sb;   // (symmetrised adjoint of S: packed)
  double Fb[J], Wb[J];
  double db, zb;
  double flux[J];   // phase flux across the link behind the cadence being reversed (phase_flux), per pair (first index)
  // coefficient cotangents: 4 J + 1 accumulators OUTSIDE the register file -- on the device a column of shared
  // memory per lane (slot k at g[k * gs]), on the host a plain array (gs = 1): k = 4 j + {a, b, c, d}; 4 J = sum of dbar
  double* g;
  int gs;
  // (device: the accumulators are columns of LDS, and `+=` there is a read -> add -> write round trip the lane waits for, 25 of them
  // per cadence at J = 6; the LDS unit's own add -- ds_add_f64, nothing returned -- is fire and forget.  A lane's column is its
  // own and LDS operations of a wave execute in order: no atomicity is asked for, only the unit that does the add.)
#ifndef EXO_GACC_LDS_ADD
#define EXO_GACC_LDS_ADD 1
#endif
  EXO_HD void gadd(int k, double v) {
    if (STATE_ONLY) return;
#if defined(__HIP_DEVICE_COMPILE__) && EXO_GACC_LDS_ADD
    (void)__hip_atomic_fetch_add(g + k * gs, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
    g[k * gs] += v;
#endif
  }

  // measurement half of cadence i: adjoints of (d, z, W) of this cadence -> (S, F) of this cadence
  // and the coefficient cotangents through U, V.  W = (V - S U) / d is rebuilt (the step keeps V).
  // Returns zbar (d loglike / d y_i) and dbar (d loglike / d diag_i); U_out (if given): this cadence's U.
  // (Rev::w_of: W of a cadence and the pieces the measurement half needs again, computed once per cadence)
  struct WOf {
    double U[J], u[J], W[J], id;
  };
  EXO_HD static void w_of(const DrawCoef<J, NR>& co, const Step<J>& s, WOf& w) {
#pragma unroll
    for (int j = 0; j < J; ++j) w.U[j] = 0.0;
    co.u_from_v(s.V, w.U);
    w.id = exo::fast_rcp(s.d);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double uj = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) uj = fma(s.S(j, l), w.U[l], uj);
      w.u[j] = uj;
      w.W[j] = (s.V[j] - uj) * w.id;
    }
  }
  EXO_HD void measure(const DrawCoef<J, NR>& co, const Step<J>& s, double dt_next, double gL, double* W_out, double* zbar_out,
                      double* dbar_out, double* U_out = nullptr) {
    WOf w;
    w_of(co, s, w);
#pragma unroll
    for (int j = 0; j < J; ++j) W_out[j] = w.W[j];
    if (U_out) {
#pragma unroll
      for (int j = 0; j < J; ++j) U_out[j] = w.U[j];
    }
    measure_w(co, s, w, dt_next, gL, zbar_out, dbar_out);
  }
  EXO_HD void measure_w(const DrawCoef<J, NR>& co, const Step<J>& s, const WOf& w, double dt_next, double gL, double* zbar_out,
                        double* dbar_out) {
    const double* U = w.U;
    const double* u = w.u;
    const double id = w.id;
    double wdot = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) wdot = fma(Wb[j], w.W[j], wdot);
    const double zbar = zb - gL * s.z * id;
    const double dbar = db + gL * (0.5 * s.z * s.z * id * id - 0.5 * id) - wdot * id;
    *zbar_out = zbar;
    *dbar_out = dbar;
    gadd(4 * J, dbar);
    double Ub[J], Vb[J], ub[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      Ub[j] = fma(-dbar, u[j], -zbar * s.F[j]);
      Fb[j] = fma(-zbar, U[j], Fb[j]);
      Vb[j] = Wb[j] * id;
      ub[j] = -Vb[j] - dbar * U[j];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double acc_u = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        if (l >= j) Sb(j, l) = fma(0.5, fma(ub[j], U[l], ub[l] * U[j]), Sb(j, l));   // symmetrised  ub U^T
        acc_u = fma(s.S(j, l), ub[l], acc_u);                                        // (S^T ub)_j, S symmetric
      }
      Ub[j] += acc_u;
    }
    // coefficient cotangents: a real term's U = a; a complex pair's (a, b, d) collect on its first index
    // (accumulators updated unconditionally, the layout deciding the increment: see DeltaCoef::eval)
    if (STATE_ONLY) return;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool re = co.is_real(j), fi = !re && co.is_first(j) && j + 1 < J;
      const int jn = j + 1 < J ? j + 1 : j;
      const double cs = s.V[j], sn = s.V[jn];
      const double a = co.k[j].a, b = co.k[j].b;
      gadd(4 * j + 0, re ? Ub[j] : (fi ? Ub[j] * cs + Ub[jn] * sn : 0.0));
      gadd(4 * j + 1, fi ? Ub[j] * sn - Ub[jn] * cs : 0.0);
      const double gph = fi ? Ub[j] * (-a * sn + b * cs) + Ub[jn] * (a * cs + b * sn) - Vb[j] * sn + Vb[jn] * cs : 0.0;
      gadd(4 * j + 3, -dt_next * flux[j]);     // (the link behind this cadence carries the flux so far: phase_flux)
      flux[j] -= gph;
    }
  }

  // reverse of the step p -> n (p = n - 1):  F_n = P o (F_p + W_p z_p),  S_n = P P^T o (S_p + d_p W_p W_p^T);
  // on entry Sb, Fb are the adjoints of S_n, F_n; on exit those of S_p, F_p, and Wb, db, zb those of
  // W_p, d_p, z_p.  Wp = W of cadence p.
  EXO_HD void propagate(const Step<J>& p, const double* Wp, const double* phi, double dt) {
    double Gb[J], Pb[J], Wbn[J];
    double zbn = 0.0, dbn = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const double Gj = fma(Wp[j], p.z, p.F[j]);
      Pb[j] = Fb[j] * Gj;
      Gb[j] = Fb[j] * phi[j];
      Wbn[j] = Gb[j] * p.z;
      zbn = fma(Gb[j], Wp[j], zbn);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double psum = 0.0, wsum = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) {
        const double T = fma(p.d * Wp[j], Wp[l], p.S(j, l));
        const double sb = Sb(j, l);                        // (still the adjoint of S_n: updated below, once per pair)
        const double Tb = sb * phi[j] * phi[l];            // adjoint of T (symmetric)
        psum = fma(2.0 * sb * T, phi[l], psum);
        wsum = fma(Tb, Wp[l], wsum);
      }
      Pb[j] += psum;
      Wbn[j] = fma(2.0 * p.d, wsum, Wbn[j]);
      dbn = fma(wsum, Wp[j], dbn);
      gadd(4 * j + 2, -dt * phi[j] * Pb[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = j; l < J; ++l) Sb(j, l) *= phi[j] * phi[l];   // becomes the adjoint of S_p
    db = dbn; zb = zbn;
#pragma unroll
    for (int j = 0; j < J; ++j) { Fb[j] = Gb[j]; Wb[j] = Wbn[j]; }
  }
};

// gsign: +1 writes d loglike / d resid; -1 writes d loglike / d model (obs - model series)
template <int J, int NR = -1, int SP = -1>
EXO_HD void chunkp_vjp_lane(const double* EXO_RESTRICT t, Series rs, const double* EXO_RESTRICT diag, int64_t n_diag,
                            int64_t n, const Coefs& cf, int64_t n_draw, const double* EXO_RESTRICT gloglike,
                            double* EXO_RESTRICT state, const ChunkGeom& cg, double* EXO_RESTRICT gresid,
                            double* EXO_RESTRICT gdiag, double gsign, int64_t draw, int c, double* gacc, int gstride,
                            int polish = 0) {
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  DrawCoef<J, NR> co;
  co.init(cf, draw);
  const double asum = co.asum();
  const SeriesRowT<false, SP> y(rs, draw, n);
  GradRowT<SP> grow(gresid, rs, draw, n);
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  const double gL = gloglike[cf.at(draw)];
  RevP<J, NR> r;
  const bool from_next = EXO_GP_POLISH && polish && n1 < n;   // (polish: entered with what the next chunk's reverse recurrences left, chunk1_fwd_lane)
  const double gsc = cg.prep ? gL : 1.0, hsc = 0.5 * gsc;   // (ChunkGeom::prep: the scan's adjoints are those of a cotangent of one)
#pragma unroll
  for (int j = 0; j < J; ++j) {
    r.Fb[j] = from_next ? state[ws.polish((polish & 1) ? 1 : 3, c + 1, j, draw)] : gsc * state[ws.bnd(2, c, j, draw)];
    r.Wb[j] = 0.0;
#pragma unroll
    for (int l = j; l < J; ++l)   // (the adjoint scan leaves a symmetric matrix)
      r.Sb(j, l) = from_next ? state[ws.polish((polish & 1) ? 1 : 3, c + 1, J + Sym<J>::idx(j, l), draw)]
                             : hsc * (state[ws.bnd(2, c, J + j * J + l, draw)] + state[ws.bnd(2, c, J + l * J + j, draw)]);
  }
  r.db = r.zb = 0.0;
  r.g = gacc; r.gs = gstride;
#pragma unroll
  for (int k = 0; k < 4 * J + 1; ++k) gacc[k * gstride] = 0.0;
  double phi[J];
  {
    // the phase flux across the chunk's end boundary (phase_flux): the next chunk's first checkpoint against the adjoint the
    // scan handed over
    double ckn[J + J * (J + 1) / 2];
#pragma unroll
    for (int k = 0; k < J + J * (J + 1) / 2; ++k) ckn[k] = (n1 < n) ? state[ws.ckpt(n1 / ckpt_span(J), k, draw)] : 0.0;
    Sym<J> Sn;
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) Sn.v[k] = ckn[J + k];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool fi = !co.is_real(j) && co.is_first(j) && j + 1 < J;
      const double fl = phase_flux<J>(co, j, ckn, Sn, r.Fb, r.Sb);
      r.flux[j] = fi ? fl : 0.0;
    }
  }
  if (n0 + 1 < n) co.set_ref(t[n0 + 1] - t[n0]);   // the forward kernel's reference: its first step
  // blocks last to first; `pend`: the step from this block's last cadence into cadence `next` (the
  // first cadence of the block after it, or of the next chunk) still has to be reversed
  const int64_t nb = (n1 - n0 + kCkptB - 1) / kCkptB;
  bool pend = n1 < n;
  constexpr int kK = J + J * (J + 1) / 2;
  constexpr int kSpan = ckpt_span(J), kSub = kCkptB / kSpan;   // cadences per checkpoint, checkpoints per block
  constexpr bool kAhead = kSub == 1;   // the next block's checkpoint is loaded ahead only where the registers allow
  // inputs of a block: its checkpoint(s) (F, packed S) and its cadences of the series
  BlockIn cur, nxt;
  double ck[kK], ck_nxt[kAhead ? kK : 1];
  auto load_ckpt = [&](int64_t i0, double* dst) {   // the checkpoint at cadence i0 (a multiple of kSpan from the chunk start)
    const int64_t g = i0 / kSpan;
#pragma unroll
    for (int k = 0; k < kK; ++k) dst[k] = ckpt_load(state + ws.ckpt(g, k, draw));
  };
  load_block(y, dg, n_diag, n0 + (nb - 1) * kCkptB, n1, cur);
  if (kAhead) {
    load_ckpt(n0 + (nb - 1) * kCkptB, ck);
#pragma unroll
    for (int k = 0; k < (kAhead ? kK : 1); ++k) ck_nxt[k] = ck[k];
  }
  nxt = cur;
#pragma unroll 1
  for (int64_t bi = nb - 1; bi >= 0; --bi) {
    const int64_t b0 = n0 + bi * kCkptB;
    const int len = (int)((n1 - b0 < kCkptB) ? n1 - b0 : kCkptB);
    if (bi > 0) {   // the block before this one: in flight while this one is worked through
      load_block(y, dg, n_diag, b0 - kCkptB, n1, nxt);
      if (kAhead) load_ckpt(b0 - kCkptB, ck_nxt);
    }
    double zbar[kCkptB], dbar[kCkptB];
#pragma unroll
    for (int q = 0; q < kCkptB; ++q) zbar[q] = dbar[q] = 0.0;
    // the block's spans, last to first: recompute a span forward from its checkpoint, keeping every cadence's state,
    // then walk it backwards
#pragma unroll
    for (int h = kSub - 1; h >= 0; --h) {
      const int q0 = h * kSpan;                                  // first cadence of the span, within the block
      const int slen = (len - q0 < kSpan) ? len - q0 : kSpan;    // its cadences that exist (<= 0: none)
      if (slen > 0) {
        if (!kAhead) load_ckpt(b0 + q0, ck);
        Step<J> st[kSpan];
        double tt[kSpan];
        {
          Fwd<J> f;
#pragma unroll
          for (int j = 0; j < J; ++j) { f.F[j] = ck[j]; f.U[j] = f.V[j] = f.W[j] = 0.0; }
#pragma unroll
          for (int k = 0; k < J * (J + 1) / 2; ++k) f.S.v[k] = ck[J + k];
          double tprev = t[b0 + q0];
#pragma unroll
          for (int ql = 0; ql < kSpan; ++ql) {
            if (ql < slen) {
              const double ti = t[b0 + q0 + ql];
              tt[ql] = ti;
              if (ql > 0) {
                const double dt = ti - tprev;
                const bool near = co.step(dt, phi);
                tprev = ti;
                f.advance(phi);
                if (near) {
                  double Vp[J];
#pragma unroll
                  for (int j = 0; j < J; ++j) Vp[j] = f.V[j];
                  co.rot_uv(dt, Vp, f.U, f.V);
                } else {
                  co.uv(ti, f.U, f.V);
                }
              } else {
                co.uv(ti, f.U, f.V);
              }
              f.measure(cur.y[q0 + ql], cur.g[q0 + ql] + asum);
              st[ql].S = f.S;
              st[ql].d = f.d; st[ql].z = f.z;
#pragma unroll
              for (int j = 0; j < J; ++j) { st[ql].F[j] = f.F[j]; st[ql].V[j] = f.V[j]; }
            } else {
              tt[ql] = 0.0;
              st[ql] = st[ql > 0 ? ql - 1 : 0];
            }
          }
        }
        typename RevP<J, NR>::WOf wq;     // (chunk1_vjp_lane: W of a cadence once, carried into its measurement half)
#pragma unroll
        for (int ql = kSpan - 1; ql >= 0; --ql) {
          if (ql < slen) {
            const int q = q0 + ql;
            const int64_t i = b0 + q;
            if (ql == slen - 1) {
              RevP<J, NR>::w_of(co, st[ql], wq);
              // the step from cadence i into cadence i + 1 when that is the first cadence of the span / block / chunk
              // after this one (already walked)
              if (q == len - 1 ? pend : true) {
                const double dt = t[i + 1] - tt[ql];
                co.step(dt, phi, false);
                r.propagate(st[ql], wq.W, phi, dt);
              }
            }
            r.measure_w(co, st[ql], wq, (ql + 1 < slen) ? tt[ql + 1] - tt[ql] : ((i + 1 < n) ? t[i + 1] - tt[ql] : 0.0), gL, &zbar[q],
                        &dbar[q]);
            if (ql > 0) {
              // reverse of the step (i - 1) -> i inside the span
              RevP<J, NR>::w_of(co, st[ql - 1], wq);
              const double dt = tt[ql] - tt[ql - 1];
              co.step(dt, phi, false);
              r.propagate(st[ql - 1], wq.W, phi, dt);
            }
          }
        }
      }
    }
    pend = true;   // the step from the previous block's last cadence into b0
    // gresid / gdiag are [draw][cadence]: the block's cadences are consecutive doubles of one row (gresid cadence-major
    // with the series: GradRow)
#pragma unroll
    for (int q = 0; q < kCkptB; ++q) zbar[q] *= gsign;
    grow.store4(b0, len, zbar);
    if (gdiag) row_store4(gdiag + cf.at(draw) * n, b0, len, dbar);
    cur = nxt;
    if (kAhead) {
#pragma unroll
      for (int k = 0; k < (kAhead ? kK : 1); ++k) ck[k] = ck_nxt[k];
    }
  }
  if (EXO_GP_POLISH && c > 0) {   // the adjoint of the state this chunk was entered with (ChunkWs::polish)
    const int slot = (polish & 1) ? 3 : 1;
#pragma unroll
    for (int j = 0; j < J; ++j) state[ws.polish(slot, c, j, draw)] = r.Fb[j];
#pragma unroll
    for (int k = 0; k < J * (J + 1) / 2; ++k) state[ws.polish(slot, c, J + k, draw)] = r.Sb.v[k];
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    state[ws.gpart(c, 4 * j + 0, draw)] = gacc[(4 * j + 0) * gstride];
    state[ws.gpart(c, 4 * j + 1, draw)] = gacc[(4 * j + 1) * gstride];
    state[ws.gpart(c, 4 * j + 2, draw)] = gacc[(4 * j + 2) * gstride];
    state[ws.gpart(c, 4 * j + 3, draw)] = gacc[(4 * j + 3) * gstride];
  }
  state[ws.gpart(c, 4 * J, draw)] = gacc[4 * J * gstride];
}

// (B'), part 1 by the EXACT RECURRENCES (round 4) -- what badj_prep_lane derives from a chunk's filtering element and the
// state entering it, taken instead from the chunk's own reverse recurrences, for the draws whose conditioning the element
// algebra cannot carry (tools/gp_host_lab.py, tools/gp_lab_robust.py: with the serial forward scan and THIS, every gradient
// of the random-kernel tail is within 5e-7 of the long-double dense definition up to a score of 1e8; badj_prep_lane's local
// terms are accurate as arithmetic but 1e7-fold sensitive to their inputs).  The reverse sweep of a chunk is affine in the
// adjoint it is entered with:  with G_p = Phi_p (I - W_p U_p^T), the closed-loop transition of cadence p and the link behind it,
//     Fbar_in = X Fbar_out + l_F,      Sbar_in = X Sbar_out X^T + sym(X Fbar_out r^T) + l_S,
//     X = G_n0^T ... G_(n1-1)^T,       r = R_n0,   R_p = -(z_p / d_p) U_p + G_p^T R_(p+1),
// and (l_F, l_S) is what the sweep leaves when entered with zero.  One sweep with the state adjoints only (RevP<STATE_ONLY>),
// the columns of X and R riding along, written over the chunk's element exactly where badj_prep_lane writes (A <- X^T,
// b <- -r, eta <- l_F, Cm <- -l_S; P = Delta - S): the adjoint scan -- tree or chain -- takes them unchanged.  c >= 1; reads
// the checkpoints of the forward pass.
// `role`: the sweep of one (draw, chunk) on up to eight lanes -- all of them recompute the forward recurrences of a span (the
// same instructions: no time), role 0 walks the state adjoints back, roles 1 .. J one column of X each, role J + 1 the vector
// R; ALL (the host): everything on one lane.
// `piece` of `n_pieces`: a reverse sweep is nothing but latency (a wave64 instruction every four cycles whatever the number of
// lanes at work), and its affine map is a product of the maps of PIECES of the chunk, each swept on its own from a zero
// adjoint: a chunk's checkpoint blocks are dealt to n_pieces sweeps (one group of eight lanes each on the device) and
// adj_combine_lane multiplies them back together -- X = X_0 X_1 ..., the plain products of the adjoint scan.  Piece records
// go to `out` (stride `os`): X row-major, r, l_F, l_S (J J + 2 J + J J doubles).
template <int J>
constexpr int adj_record_doubles() { return 2 * J * J + 2 * J; }
template <int J, int NR = -1, bool ALL = false>
EXO_HD void chunk_adj_lane(const double* EXO_RESTRICT t, Series rs, const double* EXO_RESTRICT diag, int64_t n_diag, int64_t n,
                           const Coefs& cf, int64_t n_draw, const double* EXO_RESTRICT gloglike, const double* EXO_RESTRICT state,
                           const ChunkGeom& cg, int64_t draw, int c, int role, int piece, int n_pieces, double* out, int os) {
  const int64_t n0 = c * cg.L, n1 = (n0 + cg.L < n) ? n0 + cg.L : n;
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  DrawCoef<J, NR> co;
  co.init(cf, draw);
  const double asum = co.asum();
  const SeriesRowDesc y(rs, draw, n);
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  const double gL = gloglike ? gloglike[cf.at(draw)] : 1.0;   // (null: a cotangent of one -- ChunkGeom::prep)
  constexpr bool all = ALL;
  const bool do_state = all || role == 0, is_R = all || role == J + 1, do_vec = all || role != 0;
  RevP<J, NR, true> r;
  double v[J];   // one column of X (roles 1 .. J), or R
#pragma unroll
  for (int j = 0; j < J; ++j) {
    r.Fb[j] = r.Wb[j] = r.flux[j] = 0.0;
    v[j] = (!is_R && role == j + 1) ? 1.0 : 0.0;
  }
  double Xl[ALL ? J : 1][ALL ? J : 1];   // (ALL: every column of X here)
  if constexpr (ALL) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) Xl[j][l] = (j == l) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int k = 0; k < J * (J + 1) / 2; ++k) r.Sb.v[k] = 0.0;
  r.db = r.zb = 0.0;
  r.g = nullptr; r.gs = 0;
  double phi[J];
  if (n0 + 1 < n) co.set_ref(t[n0 + 1] - t[n0]);
  const int64_t nb = (n1 - n0 + kCkptB - 1) / kCkptB;
  // this piece's blocks [bi_lo, bi_hi): whole checkpoint blocks, dealt evenly
  const int64_t bi_lo = nb * piece / n_pieces, bi_hi = nb * (piece + 1) / n_pieces;
  bool pend = bi_hi < nb || n1 < n;   // (the link behind a piece's last cadence is the piece's to reverse)
  constexpr int kK = J + J * (J + 1) / 2;
  constexpr int kSpan = ckpt_span(J), kSub = kCkptB / kSpan;
  constexpr bool kAheadBlk = J <= 4;   // (the next block's series in flight while this one is worked through: where the registers allow)
  BlockIn cur, nxt;
  double ck[kK];
  if (bi_hi > bi_lo) load_block(y, dg, n_diag, n0 + (bi_hi - 1) * kCkptB, n1, cur);
  nxt = cur;
#pragma unroll 1
  for (int64_t bi = bi_hi - 1; bi >= bi_lo; --bi) {
    const int64_t b0 = n0 + bi * kCkptB;
    const int len = (int)((n1 - b0 < kCkptB) ? n1 - b0 : kCkptB);
    if (kAheadBlk && bi > bi_lo) load_block(y, dg, n_diag, b0 - kCkptB, n1, nxt);
#pragma unroll
    for (int h = kSub - 1; h >= 0; --h) {
      const int q0 = h * kSpan;
      const int slen = (len - q0 < kSpan) ? len - q0 : kSpan;
      if (slen > 0) {
        {
          const int64_t g = (b0 + q0) / kSpan;
#pragma unroll
          for (int k = 0; k < kK; ++k) ck[k] = state[ws.ckpt(g, k, draw)];
        }
        Step<J> st[kSpan];
        double tt[kSpan];
        {
          Fwd<J> f;
#pragma unroll
          for (int j = 0; j < J; ++j) { f.F[j] = ck[j]; f.U[j] = f.V[j] = f.W[j] = 0.0; }
#pragma unroll
          for (int k = 0; k < J * (J + 1) / 2; ++k) f.S.v[k] = ck[J + k];
          double tprev = t[b0 + q0];
#pragma unroll
          for (int ql = 0; ql < kSpan; ++ql) {
            if (ql < slen) {
              const double ti = t[b0 + q0 + ql];
              tt[ql] = ti;
              if (ql > 0) {
                const double dt = ti - tprev;
                const bool near = co.step(dt, phi);
                tprev = ti;
                f.advance(phi);
                if (near) {
                  double Vp[J];
#pragma unroll
                  for (int j = 0; j < J; ++j) Vp[j] = f.V[j];
                  co.rot_uv(dt, Vp, f.U, f.V);
                } else {
                  co.uv(ti, f.U, f.V);
                }
              } else {
                co.uv(ti, f.U, f.V);
              }
              f.measure(cur.y[q0 + ql], cur.g[q0 + ql] + asum);
              st[ql].S = f.S;
              st[ql].d = f.d; st[ql].z = f.z;
#pragma unroll
              for (int j = 0; j < J; ++j) { st[ql].F[j] = f.F[j]; st[ql].V[j] = f.V[j]; }
            } else {
              tt[ql] = 0.0;
              st[ql] = st[ql > 0 ? ql - 1 : 0];
            }
          }
        }
        // W of a cadence from its saved state
        auto w_of = [&](const Step<J>& s, double* U, double* W) {
#pragma unroll
          for (int j = 0; j < J; ++j) U[j] = 0.0;
          co.u_from_v(s.V, U);
          const double id = exo::fast_rcp(s.d);
#pragma unroll
          for (int j = 0; j < J; ++j) {
            double uj = 0.0;
#pragma unroll
            for (int l = 0; l < J; ++l) uj = fma(s.S(j, l), U[l], uj);
            W[j] = (s.V[j] - uj) * id;
          }
        };
#pragma unroll
        for (int ql = kSpan - 1; ql >= 0; --ql) {
          if (ql < slen) {
            const int q = q0 + ql;
            const int64_t i = b0 + q;
            double W[J], U[J];
            const bool link = (i + 1 < n);   // the link behind cadence i; its phi is the one the last co.step call left
            w_of(st[ql], U, W);
            if (ql == slen - 1 && (q == len - 1 ? pend : true)) {
              const double dt = t[i + 1] - tt[ql];
              co.step(dt, phi, false);
              if (do_state) r.propagate(st[ql], W, phi, dt);
            }
            if (do_state) {
              double zbar, dbar, Wm[J];
              r.measure(co, st[ql], 0.0, gL, Wm, &zbar, &dbar);
            }
            if (do_vec) {
              // X <- G^T X,  R <- c U + G^T R   with  G^T x = Phi x - U (W . Phi x)
              const double cz = is_R ? -st[ql].z * exo::fast_rcp(st[ql].d) : 0.0;
              double pv[J], sdot = 0.0;
#pragma unroll
              for (int j = 0; j < J; ++j) { pv[j] = link ? phi[j] * v[j] : (is_R ? 0.0 : v[j]); sdot = fma(W[j], pv[j], sdot); }
#pragma unroll
              for (int j = 0; j < J; ++j) v[j] = fma(cz - sdot, U[j], pv[j]);
              if constexpr (ALL) {
#pragma unroll
                for (int k = 0; k < J; ++k) {
                  sdot = 0.0;
#pragma unroll
                  for (int j = 0; j < J; ++j) { pv[j] = (link ? phi[j] : 1.0) * Xl[j][k]; sdot = fma(W[j], pv[j], sdot); }
#pragma unroll
                  for (int j = 0; j < J; ++j) Xl[j][k] = fma(-sdot, U[j], pv[j]);
                }
              }
            }
            if (ql > 0) {
              double Wp[J];
              w_of(st[ql - 1], U, Wp);
              const double dt = tt[ql] - tt[ql - 1];
              co.step(dt, phi, false);
              if (do_state) r.propagate(st[ql - 1], Wp, phi, dt);
            }
          }
        }
      }
    }
    pend = true;
    if (kAheadBlk) cur = nxt;
    else if (bi > bi_lo) load_block(y, dg, n_diag, b0 - kCkptB, n1, cur);
  }
  // the piece's record: X row-major, r, l_F, l_S
  const int oX = 0, oR = J * J, oF = J * J + J, oS = J * J + 2 * J;
  if (do_state) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      out[(oF + j) * os] = r.Fb[j];
#pragma unroll
      for (int l = 0; l < J; ++l) out[(oS + j * J + l) * os] = r.Sb(j, l);
    }
  }
  if (is_R) {
#pragma unroll
    for (int j = 0; j < J; ++j) out[(oR + j) * os] = v[j];
  } else if (do_vec) {   // column role - 1 of X
#pragma unroll
    for (int j = 0; j < J; ++j) out[(oX + j * J + (role - 1)) * os] = v[j];
  }
  if constexpr (ALL) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int l = 0; l < J; ++l) out[(oX + j * J + l) * os] = Xl[j][l];
  }
}

// the pieces of a chunk's reverse sweep multiplied back together, first piece outermost (it is the last to act on an adjoint
// coming from behind the chunk):  T_a o T_b:  X = X_a X_b,  r = X_a r_b + r_a,  l_F = X_a l_F,b + l_F,a,
// l_S = X_a l_S,b X_a^T + sym((X_a l_F,b) r_a^T) + l_S,a  -- and the result written over the chunk's element where
// badj_prep_lane writes (A <- X^T, b <- -r, eta <- l_F, Cm <- -l_S).  rec: n_pieces records of adj_record_doubles<J>() doubles,
// piece p's entry k at rec[(p * rstride + k) * os].
template <int J>
EXO_HD void adj_combine_lane(const double* rec, int rstride, int os, int n_pieces, int64_t n, int64_t n_draw,
                             double* EXO_RESTRICT state, const ChunkGeom& cg, int64_t draw, int c) {
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const int oX = 0, oR = J * J, oF = J * J + J, oS = J * J + 2 * J;
  double X[J][J], S[J][J], rr[J], lF[J];
  {
    const double* q = rec + (int64_t)(n_pieces - 1) * rstride * os;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      rr[j] = q[(oR + j) * os]; lF[j] = q[(oF + j) * os];
#pragma unroll
      for (int l = 0; l < J; ++l) { X[j][l] = q[(oX + j * J + l) * os]; S[j][l] = q[(oS + j * J + l) * os]; }
    }
  }
#pragma unroll 1
  for (int p = n_pieces - 2; p >= 0; --p) {
    const double* q = rec + (int64_t)p * rstride * os;
    // (X_a is read from its record entry by entry, twice: registers are what this lane is short of)
    double T[J][J], Xn[J][J], xf[J], xr[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double a[J], f = 0.0, g = 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) a[l] = q[(oX + j * J + l) * os];
#pragma unroll
      for (int l = 0; l < J; ++l) {
        f = fma(a[l], lF[l], f);
        g = fma(a[l], rr[l], g);
        double tv = 0.0, xv = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) { tv = fma(a[k], S[k][l], tv); xv = fma(a[k], X[k][l], xv); }
        T[j][l] = tv;                       // X_a l_S,b
        Xn[j][l] = xv;                      // X_a X_b
      }
      xf[j] = f; xr[j] = g;
    }
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double a[J];
#pragma unroll
      for (int k = 0; k < J; ++k) a[k] = q[(oX + l * J + k) * os];
      const double ral = q[(oR + l) * os];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double sv = q[(oS + j * J + l) * os];
#pragma unroll
        for (int k = 0; k < J; ++k) sv = fma(T[j][k], a[k], sv);
        S[j][l] = sv + 0.5 * (xf[j] * ral + q[(oR + j) * os] * xf[l]);
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      rr[j] = xr[j] + q[(oR + j) * os];
      lF[j] = xf[j] + q[(oF + j) * os];
#pragma unroll
      for (int l = 0; l < J; ++l) X[j][l] = Xn[j][l];
    }
  }
  const int oA = 0, ob = J * J, oC = J * J + J, oeta = 2 * J * J + J;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    state[ws.elem(c, ob + j, draw)] = -rr[j];
    state[ws.elem(c, oeta + j, draw)] = lF[j];
#pragma unroll
    for (int l = 0; l < J; ++l) {
      state[ws.elem(c, oA + j * J + l, draw)] = X[l][j];                       // Abar = X^T
      state[ws.elem(c, oC + j * J + l, draw)] = -0.5 * (S[j][l] + S[l][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// O(N) companions of the likelihood (celerite2's GaussianProcess.dot_tril / predict): sequential in
// time, one lane per draw with every state index in its registers.  Not on the per-step path --
// they serve posterior predictions and prior / posterior samples.
// ---------------------------------------------------------------------------------------------
// z = L x with K + diag = L L^T:  L = (I + strict_tril(U W^T o P)) D^(1/2), so with y = sqrt(d) o x
//   F_n = P_{n-1} o (F_{n-1} + W_{n-1} y_{n-1}) ;  z_n = y_n + U_n . F_n
// (the recurrence of solve_lower with the sign flipped); the factorisation (S, d, W) runs beside it.
template <int J>
EXO_HD void dot_tril_lane(const double* EXO_RESTRICT t, const double* EXO_RESTRICT diag, int64_t n_diag, int64_t n,
                          const Coefs& cf, const double* EXO_RESTRICT x, double* EXO_RESTRICT z, int64_t draw) {
  DrawCoef<J> co;
  co.init(cf, draw);
  const double asum = co.asum();
  const double* EXO_RESTRICT dg = diag + (n_diag == 1 ? 0 : cf.at(draw) * n);
  const double* EXO_RESTRICT xr = x + draw * n;
  double* EXO_RESTRICT zr = z + draw * n;
  Fwd<J> f;
  double G[J];   // the F of the dot recurrence (f.F belongs to the unused solve_lower chain)
#pragma unroll
  for (int j = 0; j < J; ++j) { f.F[j] = f.W[j] = f.U[j] = f.V[j] = 0.0; G[j] = 0.0; }
#pragma unroll
  for (int k = 0; k < J * (J + 1) / 2; ++k) f.S.v[k] = 0.0;
  Phi<J> phi;
  double tprev = t[0], yprev = 0.0;
#pragma unroll 1
  for (int64_t i = 0; i < n; ++i) {
    const double ti = t[i];
    if (i > 0) {
      phi.set(co, ti - tprev);
      tprev = ti;
#pragma unroll
      for (int j = 0; j < J; ++j) G[j] = phi.v[j] * fma(f.W[j], yprev, G[j]);
      f.advance(phi.v);
    }
    co.uv(ti, f.U, f.V);
    f.measure(0.0, dg[i] + asum);
    const double y = sqrt(f.d > 0.0 ? f.d : __builtin_nan("")) * xr[i];   // not positive definite: NaN from here on
    double acc = y;
#pragma unroll
    for (int j = 0; j < J; ++j) acc = fma(f.U[j], G[j], acc);
    zr[i] = acc;
    yprev = y;
  }
}

// mu[m] = sum_n k(|tq_m - t_n|) alpha_n for sorted data times t and sorted query times tq, in O(N + M):
//   k(tq - t_n) = sum_j U_j(tq) V_j(t_n) e^{-c_j (tq - t_n)}   for t_n <= tq   (F sweeps forward)
//   k(t_n - tq) = sum_j V_j(tq) U_j(t_n) e^{-c_j (t_n - tq)}   for t_n >  tq   (G sweeps backward)
template <int J>
EXO_HD void predict_lane(const double* EXO_RESTRICT t, int64_t n, const double* EXO_RESTRICT alpha, const Coefs& cf,
                         const double* EXO_RESTRICT tq, int64_t m, double* EXO_RESTRICT mu, int64_t draw) {
  DrawCoef<J> co;
  co.init(cf, draw);
  const double* EXO_RESTRICT al = alpha + draw * n;
  double* EXO_RESTRICT out = mu + draw * m;
  double F[J], U[J], V[J];
#pragma unroll
  for (int j = 0; j < J; ++j) F[j] = U[j] = V[j] = 0.0;
  // forward: data points at or before the query time
  int64_t i = 0;
  double tcur = n > 0 ? t[0] : 0.0;
#pragma unroll 1
  for (int64_t q = 0; q < m; ++q) {
    const double tm = tq[q];
    while (i < n && t[i] <= tm) {
      const double ti = t[i];
      co.uv(ti, U, V);
#pragma unroll
      for (int j = 0; j < J; ++j) F[j] = fma(F[j], exp(-co.k[j].c * (ti - tcur)), V[j] * al[i]);
      tcur = ti;
      ++i;
    }
    co.uv(tm, U, V);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) acc = fma(U[j] * exp(-co.k[j].c * (tm - tcur)), F[j], acc);
    out[q] = (i > 0) ? acc : 0.0;
  }
  // backward: data points after the query time
#pragma unroll
  for (int j = 0; j < J; ++j) F[j] = 0.0;
  i = n - 1;
  tcur = n > 0 ? t[n - 1] : 0.0;
#pragma unroll 1
  for (int64_t q = m - 1; q >= 0; --q) {
    const double tm = tq[q];
    while (i >= 0 && t[i] > tm) {
      const double ti = t[i];
      co.uv(ti, U, V);
#pragma unroll
      for (int j = 0; j < J; ++j) F[j] = fma(F[j], exp(-co.k[j].c * (tcur - ti)), U[j] * al[i]);
      tcur = ti;
      --i;
    }
    if (i < n - 1) {
      co.uv(tm, U, V);
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc = fma(V[j] * exp(-co.k[j].c * (tcur - tm)), F[j], acc);
      out[q] += acc;
    }
  }
}

// how many slices the sums over the chunks are cut into (celerite_chunk_slice_sum_kernel: K quantities, rows of 64 draws): until
// the launch has ~2048 waves, at most `cap` -- a reader adds that many partials per quantity, one after the other -- and no more
// than chunks
EXO_HDH int chunk_sum_slices(int C, int K, int64_t n_draw, int cap = 64) {
  const int64_t waves = ((n_draw + 63) / 64) * K;
  int64_t S = (2048 + waves - 1) / waves;
  if (S > cap) S = cap;
  if (S > C) S = C;
  return (int)(S < 1 ? 1 : S);
}

// coefficient cotangents of one (draw, state index) from the sums over the chunks -- S partial sums per quantity, in the slots of
// chunks 0 .. S - 1 (S = 1: the totals in chunk 0's slots): the same combination as the tail of celerite_vjp_kernel
EXO_HD void gcoef_lane(const Coefs& cf, int64_t n, int64_t n_draw, const double* EXO_RESTRICT state, const ChunkGeom& cg, int S,
                       double* EXO_RESTRICT gdiag_sum, double* EXO_RESTRICT gcoef_real, double* EXO_RESTRICT gcoef_complex,
                       int64_t draw, int j) {
  const int J = cf.J();
  const ChunkWs ws = chunk_ws(n, n_draw, J, cg);
  const LaneCoef k = lane_coef(cf, draw, j, J);
  auto total = [&](int kk) {
    double v = 0.0;
    for (int s = 0; s < S; ++s) v += state[ws.gpart(s, kk, draw)];
    return v;
  };
  const double gasum = total(4 * J);
  if (j == 0 && gdiag_sum) gdiag_sum[cf.at(draw)] = gasum;
  const double ga = total(4 * j), gc = total(4 * j + 2);
  if (k.real) {
    double* o = k.slot < 0 ? gcoef_real + (cf.at(draw) * cf.n_real + j) * 2 : gcoef_complex + cf.at(draw) * cf.n_complex * 4 + k.slot;
    o[0] = ga + gasum;  // a_n = diag_n + sum a
    o[1] = gc;
  } else if (!k.odd) {
    double* o = gcoef_complex + (cf.at(draw) * cf.n_complex + ((j - cf.n_real) >> 1)) * 4;
    o[0] = ga + gasum;
    o[1] = total(4 * j + 1);
    o[2] = gc + total(4 * (j + 1) + 2);   // the decay rate is shared by the pair's two indices
    o[3] = total(4 * j + 3);
  }
}

}  // namespace gp
