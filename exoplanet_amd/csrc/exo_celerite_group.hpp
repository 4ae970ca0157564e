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
// where the tile kernel spent 230.  The J x J solves are Gauss-Jordan with partial pivoting without moving rows: the
// pivot lane publishes its row, every other lane eliminates, the rows are brought into order once at the end.
//
// celerite_scan_fused_kernel: one BLOCK per draw walks all UP levels, seeds the top, walks all DOWN levels, a block
// barrier between levels (the levels live in global memory, as before: a block's own stores are visible to it after
// __syncthreads -- one CU, one L1).  34 launches become 2.
#pragma once
#include "exo_celerite_core.hpp"

namespace gp {

template <int J>
struct GroupLds {
  static constexpr int RS = (J + 1) & ~1;          // row stride: 16-B aligned rows
  static constexpr int W = 3 * J + 1;              // widest Gauss-Jordan row: [M | A1 | C1 | r2]
  static constexpr int WS = (W + 1) & ~1;
  static constexpr int kMat = 4, kVec = 5;
  static constexpr int raw = kMat * J * RS + WS + kVec * 8;
  // strips of consecutive groups start 8 banks (4 doubles) apart modulo the 64 banks: S = 4 (mod 32)
  static constexpr int S = ((raw - 4 + 31) / 32) * 32 + 4;
};

// the eight lanes of one item
template <int J>
struct Grp {
  using L = GroupLds<J>;
  double* lds;     // this group's strip
  int r;           // row owned (lane & 7); rows >= J idle
  bool live;
  __device__ __forceinline__ double* mat(int s) const { return lds + s * (J * L::RS); }
  __device__ __forceinline__ double* gj() const { return lds + L::kMat * J * L::RS; }
  __device__ __forceinline__ double* vec(int s) const { return lds + L::kMat * J * L::RS + L::WS + s * 8; }
  // products are kept apart in the instruction stream: left to itself the scheduler starts the LDS reads of three or four
  // of them at once (72 registers each) and the composition item needs 600 registers
  static __device__ __forceinline__ void apart() { __builtin_amdgcn_sched_barrier(0); }
  static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

  // publish my row of a matrix / my entry of a vector
  __device__ __forceinline__ void put_rows(int s, const double (&a)[J]) const {
    fence();
    if (live) {
      double* p = mat(s) + r * L::RS;
#pragma unroll
      for (int l = 0; l < J; ++l) p[l] = a[l];
    }
    fence();
  }
  __device__ __forceinline__ void put_vec(int s, double v) const {
    fence();
    if (live) vec(s)[r] = v;
    fence();
  }
  // row r of (a . B):  sum_k a[k] B[k][l]
  __device__ __forceinline__ void mm(const double (&a)[J], int sB, double (&c)[J]) const {
    apart();
    const double* B = mat(sB);
#pragma unroll
    for (int l = 0; l < J; ++l) c[l] = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k)
#pragma unroll
      for (int l = 0; l < J; ++l) c[l] = fma(a[k], B[k * L::RS + l], c[l]);
    apart();
  }
  // row r of (a . B^T):  sum_k a[k] B[l][k]
  __device__ __forceinline__ void mm_t(const double (&a)[J], int sB, double (&c)[J]) const {
    apart();
    const double* B = mat(sB);
#pragma unroll
    for (int l = 0; l < J; ++l) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(a[k], B[l * L::RS + k], v);
      c[l] = v;
    }
    apart();
  }
  // row r of (A^T . B):  sum_k A[k][r] B[k][l]
  __device__ __forceinline__ void tmm(int sA, int sB, double (&c)[J]) const {
    apart();
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
    apart();
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
  // my row of (X + X^T) / 2 (through matrix slot s)
  __device__ __forceinline__ void symmetrise(int s, double (&a)[J]) const {
    put_rows(s, a);
    const double* X = mat(s);
    const int rr = live ? r : 0;
#pragma unroll
    for (int l = 0; l < J; ++l) a[l] = 0.5 * (a[l] + X[l * L::RS + rr]);
  }
  // Solve M Z = R in place (my row of M, my row of the NB right-hand sides; on return my row of Z): Gauss-Jordan with
  // partial pivoting, rows stay where they are -- lane p(k), the pivot of step k, ends up with the row of unknown k --
  // and one pass through LDS at the end brings them into order.
  template <int NB>
  __device__ __forceinline__ void solve(double (&M)[J], double (&R)[NB]) const {
    static_assert(J + NB <= L::WS, "Gauss-Jordan row does not fit its strip");
    unsigned done = 0u;
    int mine = -1;                 // the unknown my row solves for
    double* col = vec(L::kVec - 1);
    double* row = gj();
#pragma unroll
    for (int k = 0; k < J; ++k) {
      fence();
      if (live) col[r] = M[k];
      fence();
      int piv = 0;
      double best = -1.0;
#pragma unroll
      for (int i = 0; i < J; ++i) {
        const double a = ((done >> i) & 1u) ? -1.0 : fabs(col[i]);
        const bool better = a > best;          // (first of equals: as solve_inplace; a NaN never wins)
        best = better ? a : best;
        piv = better ? i : piv;
      }
      done |= 1u << piv;
      const bool is_piv = live && r == piv;
      if (is_piv) {
        mine = k;
        const double ip = 1.0 / M[k];
#pragma unroll
        for (int l = 0; l < J; ++l) M[l] *= ip;
#pragma unroll
        for (int l = 0; l < NB; ++l) R[l] *= ip;
#pragma unroll
        for (int l = 0; l < J; ++l) row[l] = M[l];
#pragma unroll
        for (int l = 0; l < NB; ++l) row[J + l] = R[l];
      }
      fence();
      const double f = is_piv ? 0.0 : M[k];
#pragma unroll
      for (int l = 0; l < J; ++l) M[l] = fma(-f, row[l], M[l]);
#pragma unroll
      for (int l = 0; l < NB; ++l) R[l] = fma(-f, row[J + l], R[l]);
    }
    // rows into order: the lane that solved for unknown k hands its right-hand sides to lane k (NB <= 2 J + 1 doubles per
    // row: through the matrix slots 0 .. 2 and the Gauss-Jordan strip, a row at a time would alias -- use a strip of
    // NB-wide rows laid over slots 0 .. 3, which the callers have finished with when they solve)
    fence();
    static_assert(J * ((NB + 1) & ~1) <= L::kMat * J * L::RS, "permutation strip");
    constexpr int PS = (NB + 1) & ~1;
    double* perm = mat(0);
    if (live) {
      double* p = perm + (mine < 0 ? r : mine) * PS;
#pragma unroll
      for (int l = 0; l < NB; ++l) p[l] = R[l];
    }
    fence();
    {
      const double* p = perm + (live ? r : 0) * PS;
#pragma unroll
      for (int l = 0; l < NB; ++l) R[l] = p[l];
    }
    fence();
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

// one item of a level on eight lanes: the same arithmetic as tree_item_lane<J, ADJ, DOWN> (exo_celerite_core.hpp)
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
    if (ADJ) {
      // x = Abar^T Fbar ;  Fbar' = lF + x ;  Pbar' = lP + Abar^T Pbar Abar + sym(x g^T)
      g.put_rows(0, el.A);
      g.put_vec(0, m);
      const double x = g.tmv(0, 0);
      double T[J];
      g.mm(P, 0, T);
      g.put_rows(1, T);
      g.put_vec(1, x);
      g.put_vec(2, el.b);
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
      g.put_vec(0, el.eta);
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
      double AY[J];
      g.mm(el.A, 1, AY);
      m2 = el.b + g.mv(el.A, 1);
      g.put_rows(2, el.A);
      g.mm_t(AY, 2, P2);
#pragma unroll
      for (int l = 0; l < J; ++l) P2[l] += el.Cm[l];
    }
    g.symmetrise(3, P2);
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
    g.mm(e1.A, 0, oA);
    ob = e2.b + g.tmv(0, 0);
    const double v = g.tmv(0, 1);
    oeta = e2.eta + v;
    double T[J];
    g.mm(e1.Cm, 0, T);
    g.put_rows(1, T);
    g.put_vec(2, v);
    g.put_vec(3, e2.b);
    g.tmm(0, 1, oC);
    double va[J], ba[J];
    g.get_vec(2, va);
    g.get_vec(3, ba);
#pragma unroll
    for (int l = 0; l < J; ++l) {
      oC[l] = e2.Cm[l] + oC[l] + 0.5 * (v * ba[l] + e2.b * va[l]);
      oJ[l] = 0.0;
    }
    g.symmetrise(2, oC);
  } else {
    // M = I + C1 J2 ;  M [X1 | X3 | x2] = [A1 | C1 | b1 + C1 eta2] ;  N = I - J2 X3
    // A = A2 X1 ;  b = A2 x2 + b2 ;  C = A2 X3 A2^T + C2 ;  eta = A1^T N (eta2 - J2 b1) + eta1 ;  J = A1^T N J2 A1 + J1
    g.put_rows(0, e2.Jm);
    g.put_vec(0, e2.eta);
    g.put_vec(1, e1.b);
    double M[J], R[2 * J + 1];
    g.mm(e1.Cm, 0, M);
#pragma unroll
    for (int l = 0; l < J; ++l) M[l] = g.live ? M[l] + (l == r ? 1.0 : 0.0) : 0.0;   // (no run-time index: registers)
#pragma unroll
    for (int l = 0; l < J; ++l) { R[l] = e1.A[l]; R[J + l] = e1.Cm[l]; }
    R[2 * J] = e1.b + g.mv(e1.Cm, 0);
    const double vv = e2.eta - g.mv(e2.Jm, 1);      // (eta2 - J2 b1)_r
    g.template solve<2 * J + 1>(M, R);              // (slots 0 .. 3 are free again)
    double X1[J], X3[J];
#pragma unroll
    for (int l = 0; l < J; ++l) { X1[l] = R[l]; X3[l] = R[J + l]; }
    g.put_rows(0, X1);
    g.put_rows(1, X3);
    g.put_vec(0, R[2 * J]);
    g.mm(e2.A, 0, oA);
    ob = e2.b + g.mv(e2.A, 0);
    double T[J], N[J];
    g.mm(e2.A, 1, T);                               // A2 X3
    g.mm(e2.Jm, 1, N);
#pragma unroll
    for (int l = 0; l < J; ++l) N[l] = ((g.live && l == r) ? 1.0 : 0.0) - N[l];
    g.put_rows(2, e2.A);
    g.mm_t(T, 2, oC);
#pragma unroll
    for (int l = 0; l < J; ++l) oC[l] += e2.Cm[l];
    g.put_rows(0, e2.Jm);
    g.put_vec(1, vv);
    double T3[J];
    g.mm(N, 0, T3);                                 // N J2
    const double w = g.mv(N, 1);                    // N (eta2 - J2 b1)
    g.put_rows(1, e1.A);
    g.put_vec(2, w);
    double TMP[J];
    g.mm(T3, 1, TMP);                               // N J2 A1
    oeta = e1.eta + g.tmv(1, 2);
    g.put_rows(2, TMP);
    g.tmm(1, 2, oJ);
#pragma unroll
    for (int l = 0; l < J; ++l) oJ[l] += e1.Jm[l];
    g.symmetrise(0, oC);
    g.symmetrise(3, oJ);
  }
  group_store_elem<J, ADJ>(state, op, c, draw, g, r, oA, ob, oC, oeta, oJ);
}

}  // namespace gp
