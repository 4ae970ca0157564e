// exo_celerite_group.hpp -- the items of the scan trees (exo_celerite_core.hpp, "The scans (B), (B') as TREES") on
// GROUPS OF EIGHT LANES, and the whole scan of a draw in ONE launch (round 4).
//
// Round 2 / 3 ran every level of the two scan trees as a launch of its own: at J = 6, C = 512 chunks that is 8
// composition launches (a WAVE per composition, 8 x 8 tiles in LDS, ~230 LDS reads of 512 B each: 39 us per level) and
// 26 one-lane launches whose items spill (~2 KB of scratch: 7-14 us per level whatever its size) -- 0.61 ms of a 2.8 ms
// step that is nothing but latency.  Here an item -- composing two filtering elements, applying one to a state, and
// the two adjoint counterparts -- is worked by eight adjacent lanes: lane r owns ROW r of every J x J matrix and entry
// r of every vector, in registers; a matrix that is the RIGHT operand of a product is staged in the group's own strip
// of LDS and read back row by row, every lane of the group the same address (a broadcast: 16-B reads, no conflict
// between the eight groups of a wave: their strips start 8 banks apart).  A wave carries eight items, needs no block
// barrier inside an item (LDS operations of one wave execute in order), and one composition costs ~50 LDS instructions
// where the tile kernel spent 230.  The J x J solves are Gauss-Jordan with partial pivoting without moving rows: every
// lane publishes its row, reads the pivot's and eliminates; the rows are brought into order once at the end.
#pragma once
#include "exo_celerite_core.hpp"

namespace gp {

template <int J>
struct GroupLds {
  static constexpr int RS = (J + 1) & ~1;          // row stride: 16-B aligned rows
  static constexpr int W = 3 * J + 1;              // widest Gauss-Jordan row: [M | A1 | C1 | r2]
  static constexpr int WS = (W + 1) & ~1;
  static constexpr int kVec = 5;
  // matrix slots 0 .. 2 live through a solve (the operands staged before it); the rows of the solve share their strip with
  // slots 3 .. 5, which are only written after it
  static constexpr int kGj = (J * WS > 3 * J * RS) ? J * WS : 3 * J * RS;
  static constexpr int raw = 3 * J * RS + kGj + kVec * 8;
  // strips of consecutive groups start 8 banks (4 doubles) apart modulo the 64 banks: S = 4 (mod 32)
  static constexpr int S = ((raw - 4 + 31) / 32) * 32 + 4;
};

template <int CTRL>
__device__ __forceinline__ int grp_dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
// the lane (0 .. 7) holding the largest a >= 0 of the aligned group of eight lanes; of (nearly) equal values the lowest lane.
// One 64-bit key per lane -- the bits of a non-negative double order like the number; its three lowest bits give way to
// 7 - lane -- and a butterfly of maxima: quad_perm [1,0,3,2], [2,3,0,1], then lane ^ 4 (row_shr / row_shl by 4; BOTH moves by
// every lane, then the choice: inside a branch the other quad is switched off, and a DPP read of a switched-off lane returns
// nothing).  a < 0 (a row already used, a lane past the state width): never chosen.
__device__ __forceinline__ int grp_argmax8(double a, int lane8) {
  unsigned long long key = (a >= 0.0) ? (((unsigned long long)__double_as_longlong(a) & ~7ull) | (unsigned)(7 - lane8)) + 8ull : 0ull;
  auto mx = [&](unsigned long long o) { key = o > key ? o : key; };
  auto mv = [](auto tag, unsigned long long k) {
    constexpr int CTRL = decltype(tag)::value;
    return ((unsigned long long)(unsigned)grp_dpp_int<CTRL>((int)(k >> 32)) << 32) | (unsigned)grp_dpp_int<CTRL>((int)k);
  };
  mx(mv(std::integral_constant<int, 0xB1>{}, key));
  mx(mv(std::integral_constant<int, 0x4E>{}, key));
  const unsigned long long dn = mv(std::integral_constant<int, 0x114>{}, key), up = mv(std::integral_constant<int, 0x104>{}, key);
  mx((threadIdx.x & 4) ? dn : up);
  return 7 - (int)(key & 7ull);
}

// the eight lanes of one item
template <int J>
struct Grp {
  using L = GroupLds<J>;
  double* lds;     // this group's strip
  int r;           // row owned (lane & 7); rows >= J idle
  bool live;
  __device__ __forceinline__ double* mat(int s) const { return lds + s * (J * L::RS); }
  __device__ __forceinline__ double* gj() const { return lds + 3 * J * L::RS; }
  __device__ __forceinline__ double* vec(int s) const { return lds + 3 * J * L::RS + L::kGj + s * 8; }
  // LDS operations of one wave execute in program order: what a lane has written is there for the other lanes of its
  // group as soon as the instruction stream says so.  sync() only keeps the COMPILER from moving reads above the
  // writes they depend on (or writes above reads of what they replace); it costs no cycles of its own -- what costs is
  // the round trip write -> read -> arithmetic -> write, so the items below stage as much as they can per exchange.
  // (NOT a wavefront-scope fence: that one also waits for every global load and store in flight -- vmcnt(0) -- at each exchange)
#ifndef EXO_GROUP_FENCE
#define EXO_GROUP_FENCE 0
#endif
  static __device__ __forceinline__ void sync() {
    if (EXO_GROUP_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    else asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }

  // publish my row of a matrix / my entry of a vector (no sync: the caller batches)
  __device__ __forceinline__ void put_rows(int s, const double (&a)[J]) const {
    if (live) {
      double* p = mat(s) + r * L::RS;
#pragma unroll
      for (int l = 0; l < J; ++l) p[l] = a[l];
    }
  }
  __device__ __forceinline__ void put_vec(int s, double v) const {
    if (live) vec(s)[r] = v;
  }
  // row r of (a . B):  sum_k a[k] B[k][l]
  __device__ __forceinline__ void mm(const double (&a)[J], int sB, double (&c)[J]) const {
    const double* B = mat(sB);
#pragma unroll
    for (int l = 0; l < J; ++l) c[l] = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k)
#pragma unroll
      for (int l = 0; l < J; ++l) c[l] = fma(a[k], B[k * L::RS + l], c[l]);
  }
  // row r of (a . B^T):  sum_k a[k] B[l][k]
  __device__ __forceinline__ void mm_t(const double (&a)[J], int sB, double (&c)[J]) const {
    const double* B = mat(sB);
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(a[k], B[l * L::RS + k], v);
      c[l] = v;
    }
  }
  // row r of (A^T . B):  sum_k A[k][r] B[k][l]
  __device__ __forceinline__ void tmm(int sA, int sB, double (&c)[J]) const {
    const double* A = mat(sA);
    const double* B = mat(sB);
    const int rr = live ? r : 0;
#pragma unroll
    for (int l = 0; l < J; ++l) c[l] = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const double a = A[k * L::RS + rr];
#pragma unroll
      for (int l = 0; l < J; ++l) c[l] = fma(a, B[k * L::RS + l], c[l]);
    }
  }
  // a . v  and  (A^T v)_r
  __device__ __forceinline__ double mv(const double (&a)[J], int sV) const {
    const double* v = vec(sV);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) acc = fma(a[k], v[k], acc);
    return acc;
  }
  __device__ __forceinline__ double tmv(int sA, int sV) const {
    const double* A = mat(sA);
    const double* v = vec(sV);
    const int rr = live ? r : 0;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) acc = fma(A[k * L::RS + rr], v[k], acc);
    return acc;
  }
  __device__ __forceinline__ void get_vec(int sV, double (&v)[J]) const {
    const double* p = vec(sV);
#pragma unroll
    for (int k = 0; k < J; ++k) v[k] = p[k];
  }
  // my row of (X + X^T) / 2 for a matrix whose rows were put into slot s (and synced) by the caller
  __device__ __forceinline__ void sym_from(int s, double (&a)[J]) const {
    const double* X = mat(s);
    const int rr = live ? r : 0;
#pragma unroll
    for (int l = 0; l < J; ++l) a[l] = 0.5 * (a[l] + X[l * L::RS + rr]);
  }
  // Solve M Z = R in place (my row of M, my row of the NB right-hand sides; on return my row of Z): Gauss-Jordan with
  // partial pivoting.  ONE exchange per step: every lane publishes its whole row [M | R]; every lane then reads column k
  // (the pivot search, the same answer in all lanes) and the pivot's row, and eliminates.  Rows stay where they are --
  // lane p(k), the pivot of step k, ends up with the row of unknown k -- and one more exchange at the end brings them
  // into order.  (A step's writes follow the previous step's reads in the instruction stream: in order, as LDS is.)
  template <int NB>
  __device__ __forceinline__ void solve(double (&M)[J], double (&R)[NB]) const {
    static_assert(J + NB <= L::WS, "Gauss-Jordan row does not fit its strip");
    // (round 4, late -- the serial chain's step, robust_fwd_chain_group, carried over: the pivot of a step is found in
    // registers, a DPP butterfly of 64-bit keys over the eight lanes, instead of every lane publishing its row and reading a
    // column back; only the pivot's lane scales its row -- by a Newton reciprocal, not an IEEE division -- and publishes it)
    const int lane8 = (int)(threadIdx.x & 7);
    int mine = -1;                 // the unknown my row solves for
    double* prow = gj();
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const double a = (live && mine < 0) ? fabs(M[k]) : -1.0;
      const int piv = grp_argmax8(a, lane8);
      const bool is_piv = lane8 == piv;
      mine = is_piv ? k : mine;
      sync();            // (the previous step's readers are done with the pivot row)
      if (is_piv) {
        const double ip = exo::fast_rcp(M[k]);
#pragma unroll
        for (int l = k; l < J; ++l) { M[l] *= ip; prow[l] = M[l]; }
#pragma unroll
        for (int l = 0; l < NB; ++l) { R[l] *= ip; prow[J + l] = R[l]; }
      }
      sync();
      // every other row loses its multiple of the scaled pivot row (the pivot's own: f = 0, untouched)
      const double f = is_piv ? 0.0 : M[k];
#pragma unroll
      for (int l = k; l < J; ++l) M[l] = fma(-f, prow[l], M[l]);
#pragma unroll
      for (int l = 0; l < NB; ++l) R[l] = fma(-f, prow[J + l], R[l]);
    }
    // rows into order: the lane that solved for unknown k hands its right-hand sides to lane k
    double* rows = gj();
    sync();
    if (live) {
      double* p = rows + (mine < 0 ? r : mine) * L::WS;
#pragma unroll
      for (int l = 0; l < NB; ++l) p[l] = R[l];
    }
    sync();
    {
      const double* p = rows + (live ? r : 0) * L::WS;
#pragma unroll
      for (int l = 0; l < NB; ++l) R[l] = p[l];
    }
    sync();
  }
};

// ---- element / state records of the tree levels, [index][quantity][draw] (tree_load_elem / tree_item_lane's layout):
// my row of every matrix, my entry of every vector
template <int J>
struct ElemRow {
  double A[J], Cm[J], Jm[J], b, eta;
};
template <int J>
__device__ __forceinline__ void group_load_elem(const double* state, const TreeOp& op, int pos, int64_t draw, const Grp<J>& g,
                                                int r, ElemRow<J>& el, bool want_J = true) {
  const bool has = pos >= 0 && pos < op.src_n;
  const int idx = op.src_rev ? op.src_len - 1 - pos : pos;
  const int64_t E = 3 * J * J + 2 * J, nd = op.n_draw;
  const double* p = state + op.src_elem + ((int64_t)(has ? idx : 0) * E) * nd + draw;
  const bool ld = has && g.live;
#pragma unroll
  for (int l = 0; l < J; ++l) {
    el.A[l] = ld ? p[(int64_t)(r * J + l) * nd] : ((g.live && l == r) ? 1.0 : 0.0);
    el.Cm[l] = ld ? p[(int64_t)(J * J + J + r * J + l) * nd] : 0.0;
    el.Jm[l] = (ld && want_J) ? p[(int64_t)(2 * J * J + 2 * J + r * J + l) * nd] : 0.0;
  }
  el.b = ld ? p[(int64_t)(J * J + r) * nd] : 0.0;
  el.eta = ld ? p[(int64_t)(2 * J * J + J + r) * nd] : 0.0;
}
template <int J, bool ADJ>
__device__ __forceinline__ void group_store_elem(double* state, const TreeOp& op, int c, int64_t draw, const Grp<J>& g, int r,
                                                 const double (&A)[J], double b, const double (&Cm)[J], double eta,
                                                 const double (&Jm)[J]) {
  if (!g.live) return;
  const int64_t E = 3 * J * J + 2 * J, nd = op.n_draw;
  double* q = state + op.dst_elem + ((int64_t)c * E) * nd + draw;
#pragma unroll
  for (int l = 0; l < J; ++l) {
    q[(int64_t)(r * J + l) * nd] = A[l];
    q[(int64_t)(J * J + J + r * J + l) * nd] = Cm[l];
    q[(int64_t)(2 * J * J + 2 * J + r * J + l) * nd] = ADJ ? 0.0 : Jm[l];
  }
  q[(int64_t)(J * J + r) * nd] = b;
  q[(int64_t)(2 * J * J + J + r) * nd] = eta;
}
template <int J>
__device__ __forceinline__ void group_put_state(double* state, const TreeOp& op, int pos, int64_t draw, const Grp<J>& g, int r,
                                                double m, const double (&P)[J]) {
  if (!g.live) return;
  const int idx = op.dst_rev ? op.dst_len - 1 - pos : pos;
  const int64_t nd = op.n_draw;
  double* q = state + op.dst_state + ((int64_t)idx * (J + J * J)) * nd + draw;
  q[(int64_t)r * nd] = m;
#pragma unroll
  for (int l = 0; l < J; ++l) q[(int64_t)(J + r * J + l) * nd] = op.psign * P[l];
}

// an element applied to a state on a group's eight lanes (my row of P / P2, my entry of m / m2): tree_apply's arithmetic
template <int J, bool ADJ>
__device__ __forceinline__ void group_apply(const Grp<J>& g, int r, const ElemRow<J>& el, double m, const double (&P)[J], double& m2,
                                            double (&P2)[J]) {
  if (ADJ) {
    // x = Abar^T Fbar ;  Fbar' = lF + x ;  Pbar' = lP + Abar^T Pbar Abar + sym(x g^T)
    g.put_rows(0, el.A);
    g.put_vec(0, m);
    g.put_vec(2, el.b);
    g.sync();
    const double x = g.tmv(0, 0);
    double T[J];
    g.mm(P, 0, T);
    g.put_rows(1, T);
    g.put_vec(1, x);
    g.sync();
    g.tmm(0, 1, P2);
    double xa[J], ba[J];
    g.get_vec(1, xa);
    g.get_vec(2, ba);
    m2 = el.eta + x;
#pragma unroll
    for (int l = 0; l < J; ++l) P2[l] = el.Cm[l] + P2[l] + 0.5 * (x * ba[l] + el.b * xa[l]);
  } else {
    // X = I + P Jm ;  solve X [YP | ym] = [P | F + P eta] ;  F' = A ym + b ;  P' = A (YP) A^T + Cm
    g.put_rows(0, el.Jm);
    g.put_rows(2, el.A);
    g.put_vec(0, el.eta);
    g.sync();
    double X[J], Bm[J + 1];
    g.mm(P, 0, X);
#pragma unroll
    for (int l = 0; l < J; ++l) X[l] = g.live ? X[l] + (l == r ? 1.0 : 0.0) : 0.0;   // (no run-time index: registers)
#pragma unroll
    for (int l = 0; l < J; ++l) Bm[l] = P[l];
    Bm[J] = m + g.mv(P, 0);
    g.template solve<J + 1>(X, Bm);
    double YP[J];
#pragma unroll
    for (int l = 0; l < J; ++l) YP[l] = Bm[l];
    g.put_rows(1, YP);
    g.put_vec(1, Bm[J]);
    g.sync();
    double AY[J];
    g.mm(el.A, 1, AY);
    m2 = el.b + g.mv(el.A, 1);
    g.mm_t(AY, 2, P2);
#pragma unroll
    for (int l = 0; l < J; ++l) P2[l] += el.Cm[l];
  }
  g.put_rows(3, P2);
  g.sync();
  g.sym_from(3, P2);
}

// The NARROW TOP of a scan as a serial chain on a group's eight lanes (round 5, J >= 3).  A level of a J = 6 scan costs its item's
// dependent latency -- ~11 LDS exchanges and a solve: 18 us to compose, 10 to apply, 9 / 7 for the adjoint scan -- twice (up, down),
// whatever its size.  From the level with eight positions up the chain instead: the level's elements applied one after the other
// to the scan's seed (`op`: the level's DOWN op with the one parent state, tree_scan_top), the next element's loads issued before
// the current application; no compositions there.  Measured at C5: an application of the chain is 4.6 us (2.3 for the adjoint
// scan), so eight of them replace three levels' 68 + 40 us with 37 + 18: step 1.87 -> 1.83 ms; from sixteen positions up it is a
// draw, from 32 a loss.  (For J <= 2 the same idea lost outright: a one-lane item is 5.5 us and a step of the chain 0.5 --
// exo_celerite.hip.)  The chain is the better conditioned association of the two.
template <int J, bool ADJ>
__device__ __forceinline__ void tree_serial_group(const TreeOp& op, double* state, int64_t draw, const Grp<J>& g) {
  const int64_t nd = op.n_draw;
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));
  double m, P[J];
  {
    const double* q = state + op.par_state + draw;
    m = g.live ? q[(int64_t)r * nd] : 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) P[l] = g.live ? q[(int64_t)(J + r * J + l) * nd] : 0.0;
  }
  group_put_state<J>(state, op, 0, draw, g, r, m, P);
  ElemRow<J> el, nxt;
  group_load_elem<J>(state, op, 0, draw, g, r, el, !ADJ);
#pragma unroll 1
  for (int pos = 0; pos + 1 < op.dst_n; ++pos) {
    group_load_elem<J>(state, op, pos + 1, draw, g, r, nxt, !ADJ);    // (in flight while this element is applied)
    double m2, P2[J];
    group_apply<J, ADJ>(g, r, el, m, P, m2, P2);
    m = m2;
#pragma unroll
    for (int l = 0; l < J; ++l) P[l] = P2[l];
    group_put_state<J>(state, op, pos + 1, draw, g, r, m, P);
    el = nxt;
  }
}

// one item of a level on eight lanes: the same arithmetic as tree_item_lane<J, ADJ, DOWN> (exo_celerite_core.hpp).
// Everything an exchange can carry is staged at once: composing two filtering elements is 4 exchanges + the 7 of its
// solve, applying one to a state 3 + 7, the adjoint items 3 each.
template <int J, bool ADJ, bool DOWN>
__device__ __forceinline__ void tree_item_group(const TreeOp& op, double* state, int c, int64_t draw, const Grp<J>& g) {
  const int64_t nd = op.n_draw;
  const int Bq = J + J * J;
  // the row index is made opaque per item: every address of the ~100 strided loads / stores of an item is (row-dependent
  // offset) x n_draw, invariant across the loops over items and levels -- hoisted, they were 600 registers and scratch
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));
  if (DOWN) {
    double m = 0.0, P[J];
    {
      const double* q = state + op.par_state + ((int64_t)c * Bq) * nd + draw;
      m = g.live ? q[(int64_t)r * nd] : 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) P[l] = g.live ? q[(int64_t)(J + r * J + l) * nd] : 0.0;
    }
    group_put_state<J>(state, op, 2 * c, draw, g, r, m, P);
    if (2 * c + 1 >= op.dst_n) return;       // (the same for all lanes of the group)
    ElemRow<J> el;
    group_load_elem<J>(state, op, 2 * c, draw, g, r, el, !ADJ);
    double m2, P2[J];
    group_apply<J, ADJ>(g, r, el, m, P, m2, P2);
    group_put_state<J>(state, op, 2 * c + 1, draw, g, r, m2, P2);
    return;
  }
  // ---- UP
  ElemRow<J> e1, e2;
  group_load_elem<J>(state, op, 2 * c, draw, g, r, e1, !ADJ);
  group_load_elem<J>(state, op, 2 * c + 1, draw, g, r, e2, !ADJ);
  double oA[J], oC[J], oJ[J], ob, oeta;
  if (ADJ) {
    // Abar = Abar1 Abar2 ;  g = g2 + Abar2^T g1 ;  lF = lF2 + Abar2^T lF1 ;  lP = lP2 + Abar2^T lP1 Abar2 + sym(Abar2^T lF1 g2^T)
    g.put_rows(0, e2.A);
    g.put_vec(0, e1.b);
    g.put_vec(1, e1.eta);
    g.put_vec(3, e2.b);
    g.sync();
    g.mm(e1.A, 0, oA);
    ob = e2.b + g.tmv(0, 0);
    const double v = g.tmv(0, 1);
    oeta = e2.eta + v;
    double T[J];
    g.mm(e1.Cm, 0, T);
    g.put_rows(1, T);
    g.put_vec(2, v);
    g.sync();
    g.tmm(0, 1, oC);
    double va[J], ba[J];
    g.get_vec(2, va);
    g.get_vec(3, ba);
#pragma unroll
    for (int l = 0; l < J; ++l) {
      oC[l] = e2.Cm[l] + oC[l] + 0.5 * (v * ba[l] + e2.b * va[l]);
      oJ[l] = 0.0;
    }
    g.put_rows(2, oC);
    g.sync();
    g.sym_from(2, oC);
  } else {
    // M = I + C1 J2 ;  M [X1 | X3 | x2] = [A1 | C1 | b1 + C1 eta2] ;  N = I - J2 X3
    // A = A2 X1 ;  b = A2 x2 + b2 ;  C = A2 X3 A2^T + C2 ;  eta = A1^T N (eta2 - J2 b1) + eta1 ;  J = A1^T N J2 A1 + J1
    g.put_rows(0, e2.Jm);
    g.put_rows(1, e2.A);
    g.put_rows(2, e1.A);
    g.put_vec(0, e2.eta);
    g.put_vec(1, e1.b);
    g.sync();
    double M[J], R[2 * J + 1];
    g.mm(e1.Cm, 0, M);
#pragma unroll
    for (int l = 0; l < J; ++l) M[l] = g.live ? M[l] + (l == r ? 1.0 : 0.0) : 0.0;   // (no run-time index: registers)
#pragma unroll
    for (int l = 0; l < J; ++l) { R[l] = e1.A[l]; R[J + l] = e1.Cm[l]; }
    R[2 * J] = e1.b + g.mv(e1.Cm, 0);
    const double vv = e2.eta - g.mv(e2.Jm, 1);      // (eta2 - J2 b1)_r
    g.template solve<2 * J + 1>(M, R);
    double X1[J], X3[J];
#pragma unroll
    for (int l = 0; l < J; ++l) { X1[l] = R[l]; X3[l] = R[J + l]; }
    g.put_rows(3, X1);
    g.put_rows(4, X3);
    g.put_vec(2, R[2 * J]);
    g.put_vec(3, vv);
    g.sync();
    g.mm(e2.A, 3, oA);
    ob = e2.b + g.mv(e2.A, 2);
    double T[J], N[J];
    g.mm(e2.A, 4, T);                               // A2 X3
    g.mm(e2.Jm, 4, N);
#pragma unroll
    for (int l = 0; l < J; ++l) N[l] = ((g.live && l == r) ? 1.0 : 0.0) - N[l];
    g.mm_t(T, 1, oC);
#pragma unroll
    for (int l = 0; l < J; ++l) oC[l] += e2.Cm[l];
    double T3[J], TMP[J];
    g.mm(N, 0, T3);                                 // N J2
    const double w = g.mv(N, 3);                    // N (eta2 - J2 b1)
    g.mm(T3, 2, TMP);                               // N J2 A1
    g.put_rows(5, TMP);
    g.put_vec(4, w);
    g.put_rows(3, oC);
    g.sync();
    oeta = e1.eta + g.tmv(2, 4);
    g.tmm(2, 5, oJ);
#pragma unroll
    for (int l = 0; l < J; ++l) oJ[l] += e1.Jm[l];
    g.sym_from(3, oC);
    g.put_rows(4, oJ);
    g.sync();
    g.sym_from(4, oJ);
  }
  group_store_elem<J, ADJ>(state, op, c, draw, g, r, oA, ob, oC, oeta, oJ);
}

// ---- the ROBUST route (exo_celerite_core.hpp, chunk_adj_lane): the forward scan of ONE draw as a serial chain on one group of
// eight lanes -- C - 1 applications of an element to a state (bscan_lane's arithmetic).  Composing elements is what loses the
// digits of ill-conditioned draws (one level of composition is enough: tools/gp_host_lab.py, LAB_HYBRID_K); applying them one
// after the other does not.  Nothing but latency, so the step is built around its exchanges: the element's own matrices are
// staged in LDS a step AHEAD (two sets of slots), the pivot of a Gauss-Jordan step is found in registers (a DPP butterfly over
// the eight lanes) and only the pivot's lane publishes its row, the solved rows are never brought into order (they are
// written to LDS where they belong), and the symmetrisation is folded into the product that consumes the solution: seven
// write -> read round trips a step (6 + 1) instead of eleven, no IEEE division.
template <int J>
struct ChainLds {
  static constexpr int RS = (J + 1) & ~1;
  static constexpr int WS = (2 * J + 2) & ~1;                 // a Gauss-Jordan row [M | Y | ym]
  // two sets of {Jm, A, eta} (this step's and the next one's), the solved rows [Y | ym], the pivot row
  static constexpr int kSet = 2 * J * RS + 8;
  static constexpr int raw = 2 * kSet + J * WS + WS;
  static constexpr int S = ((raw - 4 + 31) / 32) * 32 + 4;    // strips of consecutive groups 8 banks apart (GroupLds)
};

// my row of element c: plain loads, no predicate (a lane past the state width reads row 0 and never shows what it computes;
// every `cond ? load : constant` is a branch of its own, and the loads of a step -- there to be IN FLIGHT through the solve --
// were waited for one branch after the other)
template <int J>
__device__ __forceinline__ void chain_load_elem(const double* state, const ChunkWs& ws, int c, int64_t draw, int r, ElemRow<J>& el) {
  const int64_t nd = ws.n_draw;
  const double* p = state + ws.elem(c < ws.C ? c : ws.C - 1, 0, draw);
#pragma unroll
  for (int l = 0; l < J; ++l) {
    el.A[l] = p[(int64_t)(r * J + l) * nd];
    el.Cm[l] = p[(int64_t)(J * J + J + r * J + l) * nd];
    el.Jm[l] = p[(int64_t)(2 * J * J + 2 * J + r * J + l) * nd];
  }
  el.b = p[(int64_t)(J * J + r) * nd];
  el.eta = p[(int64_t)(2 * J * J + J + r) * nd];
}

// (B'), part 1 on a group's eight lanes -- badj_prep_lane's arithmetic (exo_celerite_core.hpp): the adjoint element of chunk c from
// its filtering element and the state entering it,
//     X = I + P Jm,  Y = X^-1,  u = eta - Jm m,  v = m + P eta,  w = Y^T u,
//     A <- A Y,  b <- eta - Jm (Y v),  eta <- gL w,  Cm <- gL / 2 (w w^T - sym(Jm Y)),
// written over the element.  One lane per (draw, chunk) keeps ~5 J x J matrices alive around the solve: 482 registers + scratch at
// J = 6 (40 us at the C5 shape, a wave per SIMD, and nothing runs beside it), 512 + 1.8 KB of scratch at J = 8 (137 us).  Here:
// lane r owns row r, three exchanges and the Gauss-Jordan solve of the scan items ([I | v] as right-hand sides).
template <int J>
__device__ __forceinline__ void badj_prep_group(const double* gloglike, double* state, const ChunkWs& ws, int c, int64_t draw,
                                                const Grp<J>& g, const int32_t* row) {
  const int64_t nd = ws.n_draw;
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));   // (tree_item_group: keeps the ~100 strided addresses from being hoisted into registers)
  const double gL = gloglike ? gloglike[row ? (int64_t)row[draw] : draw] : 1.0;   // (null: a cotangent of one -- ChunkGeom::prep)
  ElemRow<J> el;
  chain_load_elem<J>(state, ws, c, draw, r, el);
  double m, P[J];
  {
    const double* q = state + ws.bnd(1, c, 0, draw);
    m = q[(int64_t)r * nd];
#pragma unroll
    for (int l = 0; l < J; ++l) P[l] = q[(int64_t)(J + r * J + l) * nd];
  }
  g.put_rows(0, el.Jm);
  g.put_vec(0, el.eta);
  g.put_vec(1, m);
  g.sync();
  double X[J], R[J + 1];
  g.mm(P, 0, X);
#pragma unroll
  for (int l = 0; l < J; ++l) {
    X[l] = g.live ? X[l] + (l == r ? 1.0 : 0.0) : 0.0;
    R[l] = (g.live && l == r) ? 1.0 : 0.0;
  }
  const double u = el.eta - g.mv(el.Jm, 1);
  R[J] = m + g.mv(P, 0);
  g.template solve<J + 1>(X, R);
  double Y[J];
#pragma unroll
  for (int l = 0; l < J; ++l) Y[l] = R[l];
  g.put_rows(1, Y);
  g.put_vec(2, u);
  g.put_vec(3, R[J]);
  g.sync();
  const double w = g.tmv(1, 2);
  const double gj = el.eta - g.mv(el.Jm, 3);
  double Ab[J], JY[J], wall[J];
  g.mm(el.A, 1, Ab);
  g.mm(el.Jm, 1, JY);
  g.put_rows(2, JY);
  g.put_vec(4, w);
  g.sync();
  g.sym_from(2, JY);
  g.get_vec(4, wall);
  if (!g.live) return;
  double* q = state + ws.elem(c, 0, draw);
#pragma unroll
  for (int l = 0; l < J; ++l) {
    q[(int64_t)(r * J + l) * nd] = Ab[l];
    q[(int64_t)(J * J + J + r * J + l) * nd] = 0.5 * gL * (w * wall[l] - JY[l]);
  }
  q[(int64_t)(J * J + r) * nd] = gj;
  q[(int64_t)(2 * J * J + J + r) * nd] = gL * w;
}

template <int J>
__device__ __forceinline__ void robust_fwd_chain_group(const ChunkWs& ws, double* state, int64_t draw, double* lds, int lane8) {
  using L = ChainLds<J>;
  const int C = ws.C;
  const bool live = lane8 < J;
  int r = live ? lane8 : 0;
  // LDS operations of one wave execute in order: all that is needed between a lane's write and another lane's read is that the
  // COMPILER keeps them in order.  (A wavefront-scope fence does that too -- and waits for every global load in flight,
  // vmcnt(0): the next element's rows, prefetched for exactly that reason, were waited for at the first exchange of a step.)
  auto sync = [] { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  auto set_J = [&](int q) { return lds + q * L::kSet; };
  auto set_A = [&](int q) { return lds + q * L::kSet + J * L::RS; };
  auto set_eta = [&](int q) { return lds + q * L::kSet + 2 * J * L::RS; };
  double* rowsY = lds + 2 * L::kSet;          // [Y | ym], row k = unknown k
  double* prow = rowsY + J * L::WS;           // the pivot's row of the current Gauss-Jordan step
  auto stage = [&](int q, const ElemRow<J>& e) {
    if (live) {
#pragma unroll
      for (int l = 0; l < J; ++l) { set_J(q)[r * L::RS + l] = e.Jm[l]; set_A(q)[r * L::RS + l] = e.A[l]; }
      set_eta(q)[r] = e.eta;
    }
  };
  // the state entering chunk 0 (the trees hand it down untouched)
  double m = state[ws.bnd(1, 0, r, draw)], P[J];
#pragma unroll
  for (int l = 0; l < J; ++l) P[l] = state[ws.bnd(1, 0, J + r * J + l, draw)];
  ElemRow<J> el, nx;
  chain_load_elem<J>(state, ws, 0, draw, r, el);
  stage(0, el);
  sync();
#pragma unroll 1
  for (int c = 0; c + 1 < C; ++c) {
    const int q = c & 1;
    chain_load_elem<J>(state, ws, c + 1, draw, r, nx);
    // X = I + P Jm ;  right-hand sides [P | m + P eta]
    double M[J], R[J + 1];
    {
      const double* Jm = set_J(q);
#pragma unroll
      for (int l = 0; l < J; ++l) M[l] = (live && l == r) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k)
#pragma unroll
        for (int l = 0; l < J; ++l) M[l] = fma(P[k], Jm[k * L::RS + l], M[l]);
      const double* eta = set_eta(q);
      double pe = m;
#pragma unroll
      for (int k = 0; k < J; ++k) pe = fma(P[k], eta[k], pe);
#pragma unroll
      for (int l = 0; l < J; ++l) R[l] = P[l];
      R[J] = pe;
    }
    // Gauss-Jordan with partial pivoting, rows left where they are: `mine` = the unknown this lane's row ends up solving
    int mine = -1;
#pragma unroll
    for (int k = 0; k < J; ++k) {
      const double a = (live && mine < 0) ? fabs(M[k]) : -1.0;
      const int piv = grp_argmax8(a, lane8);   // (a NaN compares false with >= 0: never chosen unless nothing is)
      const bool is_piv = lane8 == piv;
      mine = is_piv ? k : mine;
      sync();            // (the previous step's readers are done with `prow`)
      if (is_piv) {      // the pivot's lane scales its row and publishes it
        const double ip = exo::fast_rcp(M[k]);
#pragma unroll
        for (int l = k; l < J; ++l) { M[l] *= ip; prow[l] = M[l]; }
#pragma unroll
        for (int l = 0; l <= J; ++l) { R[l] *= ip; prow[J + l] = R[l]; }
      }
      sync();
      // every other row loses its multiple of the scaled pivot row (the pivot's own: f = 0, untouched)
      const double f = is_piv ? 0.0 : M[k];
#pragma unroll
      for (int l = k; l < J; ++l) M[l] = fma(-f, prow[l], M[l]);
#pragma unroll
      for (int l = 0; l <= J; ++l) R[l] = fma(-f, prow[J + l], R[l]);
    }
    // the solved rows where they belong: row `mine` of [Y | ym]
    if (live) {
      double* p = rowsY + (mine < 0 ? r : mine) * L::WS;
#pragma unroll
      for (int l = 0; l <= J; ++l) p[l] = R[l];
    }
    // the NEXT step's element into the other set of slots (its loads have been in flight through the solve)
    stage(q ^ 1, nx);
    sync();
    // F' = A ym + b ;  P' = A sym(Y) A^T + Cm
    double AY[J], m2 = el.b, P2[J];
#pragma unroll
    for (int l = 0; l < J; ++l) AY[l] = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) {
      m2 = fma(el.A[k], rowsY[k * L::WS + J], m2);
      const double ha = 0.5 * el.A[k];   // A sym(Y) = (A / 2) Y + (A / 2) Y^T
#pragma unroll
      for (int l = 0; l < J; ++l) AY[l] = fma(ha, rowsY[k * L::WS + l], fma(ha, rowsY[l * L::WS + k], AY[l]));
    }
    {
      const double* A = set_A(q);
#pragma unroll
      for (int l = 0; l < J; ++l) {
        double v = el.Cm[l];
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(AY[k], A[l * L::RS + k], v);
        P2[l] = v;
      }
    }
    m = m2;
#pragma unroll
    for (int l = 0; l < J; ++l) P[l] = P2[l];
    if (live) {   // the state entering chunk c + 1
      state[ws.bnd(1, c + 1, r, draw)] = m;
#pragma unroll
      for (int l = 0; l < J; ++l) state[ws.bnd(1, c + 1, J + r * J + l, draw)] = P[l];
    }
    el = nx;
  }
}

// ---- the ROBUST route's forward scan as NEWTON ITERATIONS (round 4, late).  The states entering the chunks are the fixed
// point of  x_(c+1) = f_c(x_c)  (f_c: element c applied to a state); the trees of compositions hand over a GUESS that is wrong
// by up to 4e-2 for ill-conditioned draws; the serial chain above takes C - 1 dependent applications (1.3 ms at C5).  Newton: every
// f_c and its linearisation at the current guess -- independent, all chunks at once -- and the linear recurrence of the
// corrections
//     dm' = G (dm + dP g) + s,   dP' = G dP G^T + R,      G = A (I + P Jm)^-1,  g = eta - Jm ym,  (s, R) = f_c(x_c) - x_(c+1),
// solved by a scan of PLAIN PRODUCTS (tangent elements compose without a solve: accurate, like the adjoint tree's).  Quadratic:
// tools/gp_lab_newton.py, 450 chunks, scores up to 1e8: the guess 4e-2, one iteration 4e-7, two 8e-9 (of the serial chain's
// gradients).  One block per draw walks the levels with block barriers; the tangent elements and states of levels >= 1 live in
// the forward trees' (finished) arrays in the adjoint elements' slots (A <- G, b <- g, Cm <- R, eta <- s), level 0 is never
// stored: its items linearise on the fly (twice on the way up, once more on the way down).
template <int J>
struct TanRow {
  double G[J], R[J], g, s;
};

// my row of the tangent element of chunk `el` at the state (m, P) against the state (mn, Pn) the guess has behind it
template <int J>
__device__ __forceinline__ void tan_linearise(const Grp<J>& g, int r, const ElemRow<J>& el, double m, const double (&P)[J], double mn,
                                              const double (&Pn)[J], TanRow<J>& o) {
  g.put_rows(0, el.Jm);
  g.put_rows(2, el.A);
  g.put_vec(0, el.eta);
  g.sync();
  double X[J], B[2 * J + 1];
  g.mm(P, 0, X);
#pragma unroll
  for (int l = 0; l < J; ++l) {
    X[l] = g.live ? X[l] + (l == r ? 1.0 : 0.0) : 0.0;
    B[l] = P[l];
    B[J + 1 + l] = (g.live && l == r) ? 1.0 : 0.0;
  }
  B[J] = m + g.mv(P, 0);
  g.template solve<2 * J + 1>(X, B);          // [Y P | Y (m + P eta) | Y],  Y = (I + P Jm)^-1
  double YP[J], Y[J];
#pragma unroll
  for (int l = 0; l < J; ++l) { YP[l] = B[l]; Y[l] = B[J + 1 + l]; }
  g.put_rows(1, YP);
  g.put_rows(3, Y);
  g.put_vec(1, B[J]);
  g.sync();
  double AY[J], P2[J];
  g.mm(el.A, 1, AY);
  g.mm(el.A, 3, o.G);
  const double m2 = el.b + g.mv(el.A, 1);
  o.g = el.eta - g.mv(el.Jm, 1);
  g.mm_t(AY, 2, P2);
#pragma unroll
  for (int l = 0; l < J; ++l) P2[l] += el.Cm[l];
  g.put_rows(4, P2);
  g.sync();
  g.sym_from(4, P2);
  o.s = m2 - mn;
#pragma unroll
  for (int l = 0; l < J; ++l) o.R[l] = P2[l] - Pn[l];
  g.sync();
}
template <int J>
__device__ __forceinline__ void tan_identity(const Grp<J>& g, int r, TanRow<J>& o) {
#pragma unroll
  for (int l = 0; l < J; ++l) { o.G[l] = (g.live && l == r) ? 1.0 : 0.0; o.R[l] = 0.0; }
  o.g = o.s = 0.0;
}
// b o a (a acts first):  G = G_b G_a,  g = g_a + G_a^T g_b,  s = G_b (s_a + R_a g_b) + s_b,  R = G_b R_a G_b^T + R_b
template <int J>
__device__ __forceinline__ void tan_compose(const Grp<J>& g, const TanRow<J>& a, const TanRow<J>& b, TanRow<J>& o) {
  g.put_rows(0, a.G);
  g.put_rows(1, a.R);
  g.put_rows(2, b.G);
  g.put_vec(0, b.g);
  g.sync();
  g.mm(b.G, 0, o.G);
  o.g = a.g + g.tmv(0, 0);
  const double w = a.s + g.mv(a.R, 0);
  double T[J];
  g.mm(b.G, 1, T);
  g.put_vec(1, w);
  g.sync();
  o.s = b.s + g.mv(b.G, 1);
  g.mm_t(T, 2, o.R);
#pragma unroll
  for (int l = 0; l < J; ++l) o.R[l] += b.R[l];
  g.sync();
}
// the correction (dm, dP) carried across one tangent element
template <int J>
__device__ __forceinline__ void tan_apply(const Grp<J>& g, const TanRow<J>& e, double dm, const double (&dP)[J], double& dm2,
                                          double (&dP2)[J]) {
  g.put_rows(0, e.G);
  g.put_vec(0, e.g);
  g.sync();
  const double v = dm + g.mv(dP, 0);
  double T[J];
  g.mm_t(dP, 0, T);                      // dP G^T
  g.put_vec(1, v);
  g.put_rows(1, T);
  g.sync();
  dm2 = e.s + g.mv(e.G, 1);
  g.mm(e.G, 1, dP2);
#pragma unroll
  for (int l = 0; l < J; ++l) dP2[l] += e.R[l];
  g.sync();
}

// state entering chunk c of the guess: my entry / my row (plain loads: chain_load_elem's reasoning)
template <int J>
__device__ __forceinline__ void newton_load_state(const double* state, const ChunkWs& ws, int c, int64_t draw, int r, double& m,
                                                  double (&P)[J]) {
  m = state[ws.bnd(1, c, r, draw)];
#pragma unroll
  for (int l = 0; l < J; ++l) P[l] = state[ws.bnd(1, c, J + r * J + l, draw)];
}
template <int J>
__device__ __forceinline__ void newton_elem_of_chunk(const double* state, const ChunkWs& ws, int c, int64_t draw, const Grp<J>& g,
                                                     int r, TanRow<J>& o) {
  if (c + 1 >= ws.C) { tan_identity<J>(g, r, o); return; }     // (the last chunk's element takes no state anywhere)
  ElemRow<J> el;
  chain_load_elem<J>(state, ws, c, draw, r, el);
  double m, P[J], mn, Pn[J];
  newton_load_state<J>(state, ws, c, draw, r, m, P);
  newton_load_state<J>(state, ws, c + 1, draw, r, mn, Pn);
  tan_linearise<J>(g, r, el, m, P, mn, Pn, o);
}

// one level-0 item on the way UP: the tangent elements of chunks 2 i and 2 i + 1 composed into position i of level 1
template <int J>
__device__ __forceinline__ void newton_up0(const ChunkWs& ws, double* state, int i, int64_t draw, const Grp<J>& g) {
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));
  TanRow<J> a, b, o;
  newton_elem_of_chunk<J>(state, ws, 2 * i, draw, g, r, a);
  newton_elem_of_chunk<J>(state, ws, 2 * i + 1, draw, g, r, b);
  tan_compose<J>(g, a, b, o);
  TreeOp op{};
  op.n_draw = ws.n_draw; op.dst_elem = ws.tree_elem(1);
  group_store_elem<J, true>(state, op, i, draw, g, r, o.G, o.g, o.R, o.s, o.R);
}
// ... and on the way DOWN: the correction entering chunk 2 i (position i of level 1) added to the guess, carried across
// chunk 2 i's tangent element (taken at the OLD guess, as on the way up), added to chunk 2 i + 1's
// Returns the size of the corrections it added: max |dP_jl| / sqrt(P_jj P_ll) over my row (what the iteration count goes by).
template <int J>
__device__ __forceinline__ double newton_down0(const ChunkWs& ws, double* state, int i, int64_t draw, const Grp<J>& g) {
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));
  const int64_t nd = ws.n_draw;
  const int Bq = J + J * J, a = 2 * i;
  double dm, dP[J];
  {
    const double* q = state + ws.tree_state(1) + ((int64_t)i * Bq) * nd + draw;
    dm = q[(int64_t)r * nd];
#pragma unroll
    for (int l = 0; l < J; ++l) dP[l] = q[(int64_t)(J + r * J + l) * nd];
  }
  double m, P[J];
  newton_load_state<J>(state, ws, a, draw, r, m, P);
  // |dP_jl| against sqrt(P_jj P_ll): the diagonal through the group's LDS strip
  auto rel_size = [&](const double (&d)[J], const double (&Q)[J]) {
    double mine = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) mine = (l == r) ? Q[l] : mine;
    g.put_vec(2, mine);
    g.sync();
    double dd[J], e = 0.0;
    g.get_vec(2, dd);
#pragma unroll
    for (int l = 0; l < J; ++l) {
      const double sc = fabs(mine * dd[l]);
      e = fmax(e, newton_err_term(d[l], sc));   // (+inf for anything not finite: never "converged")
    }
    g.sync();
    return g.live ? e : 0.0;
  };
  double err = (a > 0) ? rel_size(dP, P) : 0.0;
  if (a > 0 && g.live && !((dm == dm) && (m == m) && fabs(dm) < INFINITY && fabs(m) < INFINITY)) err = INFINITY;   // the means too
  if (a + 1 < ws.C) {
    TanRow<J> e;
    newton_elem_of_chunk<J>(state, ws, a, draw, g, r, e);
    double dm2, dP2[J], mn, Pn[J];
    tan_apply<J>(g, e, dm, dP, dm2, dP2);
    newton_load_state<J>(state, ws, a + 1, draw, r, mn, Pn);
    err = fmax(err, rel_size(dP2, Pn));
    if (g.live && !((dm2 == dm2) && (mn == mn) && fabs(dm2) < INFINITY && fabs(mn) < INFINITY)) err = INFINITY;
    if (g.live) {
      state[ws.bnd(1, a + 1, r, draw)] = mn + dm2;
#pragma unroll
      for (int l = 0; l < J; ++l) state[ws.bnd(1, a + 1, J + r * J + l, draw)] = Pn[l] + dP2[l];
    }
  }
  if (g.live && a > 0) {
    state[ws.bnd(1, a, r, draw)] = m + dm;
#pragma unroll
    for (int l = 0; l < J; ++l) state[ws.bnd(1, a, J + r * J + l, draw)] = P[l] + dP[l];
  }
  return err;
}
// an item of a level f >= 1 of the corrections' scan: tangent elements composed (UP) / a correction handed down and carried
// across the left child's element (DOWN); records as the trees' (TreeOp: scan_level_op(ws, J, false, f, down))
template <int J, bool DOWN>
__device__ __forceinline__ void newton_item(const TreeOp& op, double* state, int c, int64_t draw, const Grp<J>& g) {
  int r = g.live ? g.r : 0;
  asm volatile("" : "+v"(r));
  const int64_t nd = op.n_draw;
  const int Bq = J + J * J;
  auto load_tan = [&](int pos, TanRow<J>& o) {
    ElemRow<J> el;
    group_load_elem<J>(state, op, pos, draw, g, r, el, false);
#pragma unroll
    for (int l = 0; l < J; ++l) { o.G[l] = el.A[l]; o.R[l] = el.Cm[l]; }
    o.g = el.b; o.s = el.eta;
  };
  if (DOWN) {
    double dm, dP[J];
    {
      const double* q = state + op.par_state + ((int64_t)c * Bq) * nd + draw;
      dm = g.live ? q[(int64_t)r * nd] : 0.0;
#pragma unroll
      for (int l = 0; l < J; ++l) dP[l] = g.live ? q[(int64_t)(J + r * J + l) * nd] : 0.0;
    }
    group_put_state<J>(state, op, 2 * c, draw, g, r, dm, dP);
    if (2 * c + 1 >= op.dst_n) return;
    TanRow<J> e;
    load_tan(2 * c, e);
    double dm2, dP2[J];
    tan_apply<J>(g, e, dm, dP, dm2, dP2);
    group_put_state<J>(state, op, 2 * c + 1, draw, g, r, dm2, dP2);
  } else {
    TanRow<J> a, b, o;
    load_tan(2 * c, a);
    load_tan(2 * c + 1, b);
    tan_compose<J>(g, a, b, o);
    group_store_elem<J, true>(state, op, c, draw, g, r, o.G, o.g, o.R, o.s, o.R);
  }
}

}  // namespace gp
