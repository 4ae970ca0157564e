#include <vector>
#include <cstdio>
#include <cmath>
// Test harness ONLY: compiles the one-lane celerite pipeline (exo_celerite_core.hpp) for the host
// (g++) and runs it lane by lane -- filtering elements, scan, checkpointed chunk recurrences, the
// adjoint scan and the recomputing reverse recurrences -- so that the whole time-parallel algorithm
// is checked against the oracle on a machine without a GPU.  Not part of the product; nothing in
// exoplanet_amd/ loads this.  On the GPU the same functions run one lane per (draw, chunk).
#define EXO_HOST_BUILD 1
#define EXO_GP_POLISH 1   // (the polish passes are a host-side experiment: exo_celerite_core.hpp)
#ifndef EXO_LANE_MAX_J
#define EXO_LANE_MAX_J 6
#endif
#include "../exoplanet_amd/csrc/exo_celerite_core.hpp"

#include <string.h>

namespace {

int g_serial_top = 0;    // > 0: the scans' levels with at most this many positions (and all above) as a serial chain (tree_scan_top)
int g_tree4 = 0;         // 1: two levels of a scan per "launch" where a pair is left (tree_scan4, J <= 2: what the device does)
int g_serial_scan = 0;   // 1: the serial reference scans (bscan_lane, bscan_vjp_lane) instead of the trees
int g_robust_flags = 1;  // draws the element lanes flag kFlagRobust take the robust route (as on the device)
int g_newton = -1;        // experiment: >= 0: that many Newton iterations instead of the serial forward scan (newton_scan)
int g_newton_verbose = 0;
int g_newton_tree = 0;     // the corrections' recurrence by the device's tree of plain products instead of serially
int g_adj_pieces = 8;    // the chunk's reverse sweep in that many pieces (chunk_adj_lane), as the device's eight groups of a wave
int g_adj_roles = 0;     // 1: chunk_adj_lane role by role, as the device's eight lanes run it
int g_adj_tree = 0;      // experiment (with g_robust): chunk_adj_lane's outputs scanned by the adjoint TREE instead of the serial chain
int g_hybrid_k = -1;     // experiment (with g_robust): forward scan = k levels of composition, a serial chain at level k, k levels down
int g_robust = 0;        // 1: serial forward scan + the adjoint scan's inputs from the chunks' own recurrences (chunk_adj_lane)
int g_polish = 0;        // 1: a polish pass over every draw after the chunk recurrences (forward and reverse), as the
                         //    device runs it for the draws whose conditioning asks for it

// EXPERIMENT: the states entering the chunks as the fixed point of  x_(c+1) = f_c(x_c)  (f_c: element c applied), by Newton
// iterations from the trees' (inaccurate) states: every f_c and its linearisation evaluated at the current guess (all chunks
// at once on a device), the linear recurrence of the corrections  dP' = G dP G^T + r,  dm' = G dm + G dP g + s  solved
// (here: serially; on a device a scan of plain products, like the adjoint tree).
template <int J>
void newton_scan(int64_t n, int64_t n_draw, double* state, const gp::ChunkGeom& cg, int64_t d, int iters) {
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  const int C = cg.C;
  std::vector<double> m((size_t)C * J), P((size_t)C * J * J);
  for (int c = 0; c < C; ++c)
    for (int j = 0; j < J; ++j) {
      m[c * J + j] = state[ws.bnd(1, c, j, d)];
      for (int l = 0; l < J; ++l) P[(c * J + j) * J + l] = state[ws.bnd(1, c, J + j * J + l, d)];
    }
  std::vector<double> G((size_t)C * J * J), gv((size_t)C * J), rP((size_t)C * J * J), rm((size_t)C * J);
  double last = 0.0;
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c + 1 < C; ++c) {
      gp::Elem<J> el;
      el.load(state, ws, c, d);
      double X[J][J], B[J][2 * J + 1];
      for (int j = 0; j < J; ++j) {
        double pe = m[c * J + j];
        for (int l = 0; l < J; ++l) {
          double x = (j == l) ? 1.0 : 0.0;
          for (int k = 0; k < J; ++k) x += P[(c * J + j) * J + k] * el.Jm[k][l];
          X[j][l] = x;
          B[j][l] = P[(c * J + j) * J + l];
          B[j][J + 1 + l] = (j == l) ? 1.0 : 0.0;
          pe += P[(c * J + j) * J + l] * el.eta[l];
        }
        B[j][J] = pe;
      }
      gp::solve_inplace<J, 2 * J + 1>(X, B);      // [Y P | Y (m + P eta) | Y]
      double m2[J], P2[J][J], Gc[J][J], AY[J][J];
      for (int j = 0; j < J; ++j) {
        double mj = el.b[j];
        for (int l = 0; l < J; ++l) {
          mj += el.A[j][l] * B[l][J];
          double v = 0.0, gg = 0.0;
          for (int k = 0; k < J; ++k) { v += el.A[j][k] * B[k][l]; gg += el.A[j][k] * B[k][J + 1 + l]; }
          AY[j][l] = v; Gc[j][l] = gg;
        }
        m2[j] = mj;
      }
      for (int j = 0; j < J; ++j)
        for (int l = 0; l < J; ++l) {
          double v = el.Cm[j][l];
          for (int k = 0; k < J; ++k) v += AY[j][k] * el.A[l][k];
          P2[j][l] = v;
        }
      for (int j = 0; j < J; ++j) {
        double gj = el.eta[j];
        for (int l = 0; l < J; ++l) gj -= el.Jm[j][l] * B[l][J];
        gv[c * J + j] = gj;                                   // g = eta - Jm Y (m + P eta)
        rm[(c + 1) * J + j] = m2[j] - m[(c + 1) * J + j];
        for (int l = 0; l < J; ++l) {
          G[(c * J + j) * J + l] = Gc[j][l];
          rP[((c + 1) * J + j) * J + l] = 0.5 * (P2[j][l] + P2[l][j]) - P[((c + 1) * J + j) * J + l];
        }
      }
    }
    if (g_newton_tree) {
      // ... the linear recurrence as the device solves it: tangent elements composed pairwise, level by level, by plain
      // products (exo_celerite_group.hpp, tan_compose), a zero correction in front of the first chunk handed back down
      // (tan_apply).  Element c (G, g, s, R) carries a correction from chunk c to chunk c + 1.
      struct Tan { double G[J][J], g[J], s[J], R[J][J]; };
      auto compose = [](const Tan& a, const Tan& b) {       // b o a
        Tan o;
        for (int j = 0; j < J; ++j) {
          double gg = a.g[j], w = 0.0;
          for (int k = 0; k < J; ++k) gg += a.G[k][j] * b.g[k];
          o.g[j] = gg;
          (void)w;
          for (int l = 0; l < J; ++l) {
            double v = 0.0;
            for (int k = 0; k < J; ++k) v += b.G[j][k] * a.G[k][l];
            o.G[j][l] = v;
          }
        }
        double wv[J], T[J][J];
        for (int j = 0; j < J; ++j) {
          double v = a.s[j];
          for (int l = 0; l < J; ++l) v += a.R[j][l] * b.g[l];
          wv[j] = v;
        }
        for (int j = 0; j < J; ++j) {
          double v = b.s[j];
          for (int k = 0; k < J; ++k) v += b.G[j][k] * wv[k];
          o.s[j] = v;
          for (int l = 0; l < J; ++l) {
            double tv = 0.0;
            for (int k = 0; k < J; ++k) tv += b.G[j][k] * a.R[k][l];
            T[j][l] = tv;
          }
        }
        for (int j = 0; j < J; ++j)
          for (int l = 0; l < J; ++l) {
            double v = b.R[j][l];
            for (int k = 0; k < J; ++k) v += T[j][k] * b.G[l][k];
            o.R[j][l] = v;
          }
        return o;
      };
      auto ident = [] {
        Tan o{};
        for (int j = 0; j < J; ++j) o.G[j][j] = 1.0;
        return o;
      };
      std::vector<std::vector<Tan>> lev(1);
      lev[0].resize(C);
      for (int c = 0; c < C; ++c) {
        if (c + 1 >= C) { lev[0][c] = ident(); continue; }
        Tan& e = lev[0][c];
        for (int j = 0; j < J; ++j) {
          e.g[j] = gv[c * J + j];
          e.s[j] = rm[(c + 1) * J + j];
          for (int l = 0; l < J; ++l) { e.G[j][l] = G[(c * J + j) * J + l]; e.R[j][l] = rP[((c + 1) * J + j) * J + l]; }
        }
      }
      while (lev.back().size() > 1) {
        const std::vector<Tan>& src = lev.back();
        std::vector<Tan> dst((src.size() + 1) / 2);
        for (size_t i = 0; i < dst.size(); ++i) dst[i] = compose(src[2 * i], 2 * i + 1 < src.size() ? src[2 * i + 1] : ident());
        lev.push_back(dst);
      }
      struct St { double dm[J], dP[J][J]; };
      std::vector<St> cur(1, St{}), nxt;
      for (int f = (int)lev.size() - 2; f >= 0; --f) {
        nxt.assign(lev[f].size(), St{});
        for (size_t i = 0; i < cur.size(); ++i) {
          nxt[2 * i] = cur[i];
          if (2 * i + 1 >= lev[f].size()) continue;
          const Tan& e = lev[f][2 * i];
          St o;
          double v[J], T[J][J];
          for (int j = 0; j < J; ++j) {
            double x = cur[i].dm[j];
            for (int l = 0; l < J; ++l) x += cur[i].dP[j][l] * e.g[l];
            v[j] = x;
          }
          for (int j = 0; j < J; ++j) {
            double x = e.s[j];
            for (int k = 0; k < J; ++k) x += e.G[j][k] * v[k];
            o.dm[j] = x;
            for (int l = 0; l < J; ++l) {
              double tv = 0.0;
              for (int k = 0; k < J; ++k) tv += cur[i].dP[j][k] * e.G[l][k];
              T[j][l] = tv;
            }
          }
          for (int j = 0; j < J; ++j)
            for (int l = 0; l < J; ++l) {
              double x = e.R[j][l];
              for (int k = 0; k < J; ++k) x += e.G[j][k] * T[k][l];
              o.dP[j][l] = x;
            }
          nxt[2 * i + 1] = o;
        }
        cur.swap(nxt);
      }
      last = 0.0;
      for (int c = 1; c < C; ++c)
        for (int j = 0; j < J; ++j) {
          m[c * J + j] += cur[c].dm[j];
          for (int l = 0; l < J; ++l) {
            P[(c * J + j) * J + l] += cur[c].dP[j][l];
            const double sc = std::fabs(P[(c * J + j) * J + j] * P[(c * J + l) * J + l]);
            if (sc > 0) last = std::fmax(last, std::fabs(cur[c].dP[j][l]) / std::sqrt(sc));
          }
        }
      if (g_newton_verbose) fprintf(stderr, "newton (tree) draw %lld it %d: max |dP| / sqrt(Pjj Pll) = %.2e\n", (long long)d, it, last);
      if (last < 1e-8) break;
      continue;
    }
    // corrections, chunk 0's state exact
    double dP[J][J] = {}, dm[J] = {};
    last = 0.0;
    for (int c = 0; c + 1 < C; ++c) {
      double T[J][J], nP[J][J], nm[J], dg[J];
      for (int j = 0; j < J; ++j) {
        dg[j] = 0.0;
        for (int l = 0; l < J; ++l) dg[j] += dP[j][l] * gv[c * J + l];
      }
      for (int j = 0; j < J; ++j) {
        double v = rm[(c + 1) * J + j];
        for (int l = 0; l < J; ++l) {
          v += G[(c * J + j) * J + l] * (dm[l] + dg[l]);
          double tv = 0.0;
          for (int k = 0; k < J; ++k) tv += G[(c * J + j) * J + k] * dP[k][l];
          T[j][l] = tv;
        }
        nm[j] = v;
      }
      for (int j = 0; j < J; ++j)
        for (int l = 0; l < J; ++l) {
          double v = rP[((c + 1) * J + j) * J + l];
          for (int k = 0; k < J; ++k) v += T[j][k] * G[(c * J + l) * J + k];
          nP[j][l] = v;
        }
      for (int j = 0; j < J; ++j) {
        dm[j] = nm[j];
        m[(c + 1) * J + j] += dm[j];
        for (int l = 0; l < J; ++l) {
          dP[j][l] = 0.5 * (nP[j][l] + nP[l][j]);
          P[((c + 1) * J + j) * J + l] += dP[j][l];
          const double sc = std::fabs(P[((c + 1) * J + j) * J + j] * P[((c + 1) * J + l) * J + l]);
          if (sc > 0) last = std::fmax(last, std::fabs(dP[j][l]) / std::sqrt(sc));
        }
      }
    }
    if (g_newton_verbose) fprintf(stderr, "newton draw %lld it %d: max |dP| / sqrt(Pjj Pll) = %.2e\n", (long long)d, it, last);
  }
  for (int c = 1; c < C; ++c)
    for (int j = 0; j < J; ++j) {
      state[ws.bnd(1, c, j, d)] = m[c * J + j];
      for (int l = 0; l < J; ++l) state[ws.bnd(1, c, J + j * J + l, d)] = P[(c * J + j) * J + l];
    }
}

template <int J>
void run_fwd(const double* t, gp::Series rs, const double* diag, int64_t n_diag, int64_t n, const gp::Coefs& cf,
             int64_t n_draw, double* loglike, double* state, const gp::ChunkGeom& cg) {
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  for (int64_t d = 0; d < n_draw; ++d) {
    gp::DeltaCoef<J> dc;
    dc.init(cf, d);
    state[ws.off_flag() + d] = dc.valid ? gp::kFlagClean : gp::kFlagSeq;
  }
  for (int c = 0; c < cg.C; ++c)
    for (int64_t d = 0; d < n_draw; ++d)
      gp::with_layout<J>(cf, d, [&](auto nr) {
        gp::elem_lane<J, decltype(nr)::value>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, d, c);
      });
  if (cg.tree && !g_serial_scan && !g_robust) {   // the scans as trees of compositions, level by level, as the device launches them
    auto launch = [&](const gp::TreeOp& op, bool down) {
      for (int c = 0; c < op.n_item; ++c)
        for (int64_t d = 0; d < n_draw; ++d) {
          if (down) gp::tree_item_lane<J, false, true>(op, state, c, d);
          else gp::tree_item_lane<J, false, false>(op, state, c, d);
        }
    };
    auto seed = [&]() {
      for (int64_t d = 0; d < n_draw; ++d) gp::scan_init_lane<J>(t, cf, n_draw, state + ws.tree_state(ws.tree_top()), d);
    };
    if (g_tree4)
      gp::tree_scan4(ws, J, false, launch,
                     [&](const gp::TreeOp& a, const gp::TreeOp& b, bool down) {
                       for (int i = 0; i < b.n_item; ++i)
                         for (int64_t d = 0; d < n_draw; ++d) {
                           if (down) gp::tree_item4_down_lane<J, false>(a, b, state, i, d);
                           else gp::tree_item4_up_lane<J, false>(a, b, state, i, d);
                         }
                     },
                     seed);
    else if (g_serial_top > 0) {
      int f0 = 0;
      while (ws.tree_npos(f0) > g_serial_top) ++f0;
      gp::tree_scan_top(ws, J, false, f0 + 2 <= ws.tree_top() ? f0 : ws.tree_top(), launch, seed,
                        [&](const gp::TreeOp& op) { for (int64_t d = 0; d < n_draw; ++d) gp::tree_serial_lane<J, false>(op, state, d); });
    } else
      gp::tree_scan(ws, J, false, launch, seed);
  } else if (g_robust && g_hybrid_k >= 0 && cg.tree) {
    const int top = ws.tree_top();
    const int k = g_hybrid_k < top ? g_hybrid_k : top - 1;
    for (int64_t d = 0; d < n_draw; ++d) {
      for (int f = 0; f < k; ++f) {
        const gp::TreeOp op = gp::scan_level_op(ws, J, false, f, false);
        for (int c = 0; c < op.n_item; ++c) gp::tree_item_lane<J, false, false>(op, state, c, d);
      }
      // the chain at level k: position p + 1's state from position p's and element p of that level
      gp::TreeOp op{};
      op.J = J; op.n_draw = n_draw;
      op.src_elem = k == 0 ? ws.elem(0, 0, 0) : ws.tree_elem(k);
      op.src_n = k == 0 ? ws.C - 1 : ws.tree_npos(k);
      op.src_len = ws.tree_npos(k);
      const int64_t base = k == 0 ? ws.bnd(1, 0, 0, 0) : ws.tree_state(k);
      const int Bq = J + J * J;
      gp::scan_init_lane<J>(t, cf, n_draw, state + base, d);
      double m[J], P[J][J];
      for (int j = 0; j < J; ++j) {
        m[j] = state[base + (int64_t)j * n_draw + d];
        for (int l = 0; l < J; ++l) P[j][l] = state[base + (int64_t)(J + j * J + l) * n_draw + d];
      }
      for (int pos = 0; pos + 1 < ws.tree_npos(k); ++pos) {
        gp::Elem<J> el;
        gp::tree_load_elem<J>(state, op, pos, d, el);
        gp::apply_elem<J>(el, m, P);
        for (int j = 0; j < J; ++j) {
          state[base + ((int64_t)(pos + 1) * Bq + j) * n_draw + d] = m[j];
          for (int l = 0; l < J; ++l) state[base + ((int64_t)(pos + 1) * Bq + J + j * J + l) * n_draw + d] = P[j][l];
        }
      }
      for (int f = k - 1; f >= 0; --f) {
        const gp::TreeOp dn = gp::scan_level_op(ws, J, false, f, true);
        for (int c = 0; c < dn.n_item; ++c) gp::tree_item_lane<J, false, true>(dn, state, c, d);
      }
    }
  } else {
    for (int64_t d = 0; d < n_draw; ++d) gp::bscan_lane<J>(t, cf, n, n_draw, state, cg, d);
  }
  // the robust route, as the device takes it: draws flagged kFlagRobust get their entering states from the serial scan
  // (experiment, g_newton >= 0: from that many Newton iterations on the chunk-boundary fixed point, started at the trees' states)
  if (cg.tree && !g_serial_scan && !g_robust && g_robust_flags)
    for (int64_t d = 0; d < n_draw; ++d)
      if (state[ws.off_flag() + d] == gp::kFlagRobust) {
        if (g_newton >= 0) newton_scan<J>(n, n_draw, state, cg, d, g_newton);
        else gp::bscan_lane<J>(t, cf, n, n_draw, state, cg, d);
      }
  if (g_polish < 0 && ((-g_polish) & 1)) {
    // experiment: the chunks ONE AFTER THE OTHER, each entered with what its predecessor just left -- the sequential algorithm
    // in chunk-sized steps (exact boundary states): what the accuracy would be if the scans were perfect
    for (int c = 0; c < cg.C; ++c) {
      for (int64_t d = 0; d < n_draw; ++d)
        gp::with_layout<J>(cf, d, [&](auto nr) {
          gp::chunk1_fwd_lane<J, decltype(nr)::value>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, d, c, true, c > 0 ? 1 : 0);
        });
      if (c > 0)
        for (int k = 0; k < ws.K(); ++k)
          for (int64_t d = 0; d < n_draw; ++d) state[ws.polish(0, c, k, d)] = state[ws.polish(2, c, k, d)];
    }
  } else
  for (int pass = 0; pass <= (g_polish > 0 ? g_polish : 0); ++pass)
    for (int c = 0; c < cg.C; ++c)
      for (int64_t d = 0; d < n_draw; ++d)
        gp::with_layout<J>(cf, d, [&](auto nr) {
          gp::chunk1_fwd_lane<J, decltype(nr)::value>(t, rs, diag, n_diag, n, cf, n_draw, state, cg, d, c, true, pass);
        });
  for (int64_t d = 0; d < n_draw; ++d) {
    double acc = 0.0, logdet = 0.0, bad = 0.0;
    for (int c = 0; c < cg.C; ++c) {
      acc += state[ws.part(c, 0, d)];
      logdet += state[ws.part(c, 1, d)];
      bad += state[ws.part(c, 2, d)];
    }
    loglike[d] = (bad > 0.0) ? -INFINITY : fma(-0.5, acc + logdet, -(double)n * gp::kHalfLog2Pi);
  }
}

template <int J>
void run_vjp(const double* t, gp::Series rs, const double* diag, int64_t n_diag, int64_t n, const gp::Coefs& cf,
             int64_t n_draw, const double* gloglike, double* state, const gp::ChunkGeom& cg, double* gresid,
             double* gdiag, double gsign, double* gdiag_sum, double* gcr, double* gcc) {
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  // (B') part 1: from the element (badj_prep_lane), or -- draws flagged kFlagRobust, as on the device; every draw with
  // g_robust -- from the chunk's own reverse recurrences (chunk_adj_lane: the device's eight lanes role by role with g_adj_roles)
  auto robust_draw = [&](int64_t d) {
    return g_robust || (cg.tree && !g_serial_scan && g_robust_flags && state[ws.off_flag() + d] == gp::kFlagRobust);
  };
  for (int c = 1; c < cg.C; ++c)
    for (int64_t d = 0; d < n_draw; ++d) {
      if (robust_draw(d)) {
        gp::with_layout<J>(cf, d, [&](auto nr) {
          constexpr int NR = decltype(nr)::value, kRec = gp::adj_record_doubles<J>();
          double rec[8][kRec];
          const int np = g_adj_pieces < 1 ? 1 : (g_adj_pieces > 8 ? 8 : g_adj_pieces);
          for (int piece = 0; piece < np; ++piece) {
            if (g_adj_roles) {
              for (int role = 0; role <= J + 1; ++role)
                gp::chunk_adj_lane<J, NR, false>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, d, c, role, piece, np, rec[piece], 1);
            } else {
              gp::chunk_adj_lane<J, NR, true>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, d, c, -1, piece, np, rec[piece], 1);
            }
          }
          gp::adj_combine_lane<J>(&rec[0][0], kRec, 1, np, n, n_draw, state, cg, d, c);
        });
      } else {
        gp::badj_prep_lane<J>(gloglike, n, n_draw, state, cg, d, c);
      }
    }
  if (cg.tree && !g_serial_scan && (!g_robust || g_adj_tree)) {   // (the adjoint tree takes the robust draws' inputs as they are)
    auto launch = [&](const gp::TreeOp& op, bool down) {
      for (int c = 0; c < op.n_item; ++c)
        for (int64_t d = 0; d < n_draw; ++d) {
          if (down) gp::tree_item_lane<J, true, true>(op, state, c, d);
          else gp::tree_item_lane<J, true, false>(op, state, c, d);
        }
    };
    auto seed = [&]() {
      double* dst = state + ws.tree_state(ws.tree_top());
      for (int64_t k = 0; k < (int64_t)ws.B() * n_draw; ++k) dst[k] = 0.0;
    };
    if (g_tree4)
      gp::tree_scan4(ws, J, true, launch,
                     [&](const gp::TreeOp& a, const gp::TreeOp& b, bool down) {
                       for (int i = 0; i < b.n_item; ++i)
                         for (int64_t d = 0; d < n_draw; ++d) {
                           if (down) gp::tree_item4_down_lane<J, true>(a, b, state, i, d);
                           else gp::tree_item4_up_lane<J, true>(a, b, state, i, d);
                         }
                     },
                     seed);
    else if (g_serial_top > 0) {
      int f0 = 0;
      while (ws.tree_npos(f0) > g_serial_top) ++f0;
      gp::tree_scan_top(ws, J, true, f0 + 2 <= ws.tree_top() ? f0 : ws.tree_top(), launch, seed,
                        [&](const gp::TreeOp& op) { for (int64_t d = 0; d < n_draw; ++d) gp::tree_serial_lane<J, true>(op, state, d); });
    } else
      gp::tree_scan(ws, J, true, launch, seed);
  } else {
    for (int64_t d = 0; d < n_draw; ++d) gp::bscan_vjp_lane<J>(n, n_draw, state, cg, d);
  }
  const bool seq_adj = g_polish < 0 && ((-g_polish) & 2);   // experiment: exact adjoint boundary states (chunks last to first)
  for (int pass = 0; pass <= (g_polish > 0 ? g_polish : 0); ++pass)
  for (int cc = 0; cc < cg.C; ++cc) {
    const int c = seq_adj ? cg.C - 1 - cc : cc;
    if (seq_adj && c + 1 < cg.C && c + 1 < cg.C - 0 && cc > 1)
      for (int k = 0; k < ws.K(); ++k)
        for (int64_t d = 0; d < n_draw; ++d) state[ws.polish(1, c + 1, k, d)] = state[ws.polish(3, c + 1, k, d)];
    const int lane_pass = seq_adj ? (cc > 0 ? 1 : 0) : pass;
    for (int64_t d = 0; d < n_draw; ++d)
      gp::with_layout<J>(cf, d, [&](auto nr) {
        if constexpr (J > 2) {   // as the device wrapper (celerite_chunk1_vjp_kernel) dispatches
          double gacc[4 * J + 1];
          gp::chunkp_vjp_lane<J, decltype(nr)::value>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag,
                                                      gsign, d, c, gacc, 1, lane_pass);
        } else
          gp::chunk1_vjp_lane<J, decltype(nr)::value>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag,
                                                      gsign, d, c, lane_pass);
      });
  }
  for (int64_t d = 0; d < n_draw; ++d)
    for (int k = 0; k < 4 * J + 1; ++k) {
      double v = 0.0;
      for (int c = 0; c < cg.C; ++c) v += state[ws.gpart(c, k, d)];
      state[ws.gpart(0, k, d)] = v;
    }
  for (int64_t d = 0; d < n_draw; ++d)
    for (int j = 0; j < J; ++j) gp::gcoef_lane(cf, n, n_draw, state, cg, 1, gdiag_sum, gcr, gcc, d, j);
}

}  // namespace

extern "C" {

void harness_set_serial_scan(int v) { g_serial_scan = v; }
void harness_set_tree4(int v) { g_tree4 = v; }
void harness_set_serial_top(int v) { g_serial_top = v; }
void harness_set_polish(int v) { g_polish = v; }
void harness_set_robust(int v) { g_robust = v; }
void harness_set_adj_tree(int v) { g_adj_tree = v; }
void harness_set_adj_roles(int v) { g_adj_roles = v; }
void harness_set_adj_pieces(int v) { g_adj_pieces = v; }
double harness_newton_err_term(double d, double sc) { return gp::newton_err_term(d, sc); }
void harness_set_newton(int v, int verbose) { g_newton = v; g_newton_verbose = verbose; }
void harness_set_newton_tree(int v) { g_newton_tree = v; }
void harness_set_hybrid_k(int v) { g_hybrid_k = v; }
void harness_set_robust_flags(int v) { g_robust_flags = v; }
// (experiments: where the checkpoints -- the states (F, packed S) entering every ckpt_span(J)-th cadence -- live in `state`)
// (experiments: where the filtering elements [chunk][A, b, C, eta, J][draw] live in `state`)
int64_t harness_gp_elem_offset(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t n_chunks, int64_t* C) {
  const int J = n_real + 2 * n_complex;
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  *C = cg.C;
  return ws.elem(0, 0, 0);
}
// (experiments: out = {off_bnd, B, off_polish, K, C, L})
void harness_gp_offsets(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t n_chunks, int64_t* out) {
  const int J = n_real + 2 * n_complex;
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  out[0] = ws.off_bnd(); out[1] = ws.B(); out[2] = ws.off_polish(); out[3] = ws.K(); out[4] = cg.C; out[5] = cg.L;
}
int64_t harness_gp_ckpt_layout(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t n_chunks, int64_t* K,
                               int64_t* span, int64_t* L) {
  const int J = n_real + 2 * n_complex;
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  *K = ws.K(); *span = gp::ckpt_span(J); *L = cg.L;
  return ws.off_ckpt();
}


int64_t harness_gp_state_doubles(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t n_chunks) {
  const int J = n_real + 2 * n_complex;
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  if (cg.C <= 1) return cg.base;
  return cg.base + gp::chunk_ws(n, n_draw, J, cg).total();
}

// the layout of y / gresid for the calls that follow: 0 = [draw][cadence], 1 = cadence-major [cadence][draw]
static int g_cadence_major = 0;
void harness_gp_set_cadence_major(int on) { g_cadence_major = on; }
// ... or a SPARSE model (gp::SparseSegs; y is then the value array and gresid receives the cotangent of the values); nseg == NULL: off
static gp::SparseSegs g_sparse{};
void harness_gp_set_sparse(const int32_t* nseg, const int32_t* seg, const int32_t* off, int64_t seg_row, int64_t off_row,
                           int64_t val_row, int32_t seg_step, int32_t hi_at) {
  g_sparse = gp::SparseSegs{};
  g_sparse.nseg = nseg; g_sparse.seg = seg; g_sparse.off = off;
  g_sparse.seg_row = seg_row; g_sparse.off_row = off_row; g_sparse.val_row = val_row;
  g_sparse.seg_step = seg_step; g_sparse.hi_at = hi_at;
}
static gp::Series harness_series(const double* y, const double* obs, int64_t n_draw) {
  gp::Series rs{y, obs, g_cadence_major ? n_draw : 0};
  if (g_sparse.nseg) { rs.cm = 0; rs.sp = g_sparse; }
  return rs;
}

// returns the number of chunks used (1: the plan is sequential, nothing was computed), -1 on bad J
int harness_gp_fwd(const double* t, const double* y, const double* obs, const double* diag, int64_t n_diag, int64_t n,
                   const double* real, int32_t n_real, const double* cplx, int32_t n_complex, const int32_t* kind,
                   int64_t n_draw, int32_t n_chunks, double* loglike, double* state, double* flags) {
  const gp::Coefs cf{real, cplx, kind, n_real, n_complex, n > 0 ? t : nullptr};   // (as the device entry points)
  const int J = cf.J();
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  if (cg.C <= 1 || !cg.lane) return cg.C <= 1 ? 1 : -1;
  const gp::Series rs = harness_series(y, obs, n_draw);
  switch (J) {
    case 1: run_fwd<1>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    case 2: run_fwd<2>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    case 3: run_fwd<3>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    case 4: run_fwd<4>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    case 5: run_fwd<5>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    case 6: run_fwd<6>(t, rs, diag, n_diag, n, cf, n_draw, loglike, state, cg); break;
    default: return -1;
  }
  const gp::ChunkWs ws = gp::chunk_ws(n, n_draw, J, cg);
  for (int64_t d = 0; d < n_draw; ++d) flags[d] = state[ws.off_flag() + d];
  return cg.C;
}

int harness_gp_vjp(const double* t, const double* y, const double* obs, const double* diag, int64_t n_diag, int64_t n,
                   const double* real, int32_t n_real, const double* cplx, int32_t n_complex, const int32_t* kind,
                   int64_t n_draw, int32_t n_chunks, const double* gloglike, double* state, double* gresid,
                   double* gdiag, double* gdiag_sum, double* gcr, double* gcc) {
  const gp::Coefs cf{real, cplx, kind, n_real, n_complex, n > 0 ? t : nullptr};   // (as the device entry points)
  const int J = cf.J();
  const gp::ChunkGeom cg = gp::chunk_plan(n, n_draw, J, n_chunks);
  if (cg.C <= 1 || !cg.lane) return cg.C <= 1 ? 1 : -1;
  const gp::Series rs = harness_series(y, obs, n_draw);
  const double gsign = obs ? -1.0 : 1.0;
  switch (J) {
    case 1: run_vjp<1>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    case 2: run_vjp<2>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    case 3: run_vjp<3>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    case 4: run_vjp<4>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    case 5: run_vjp<5>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    case 6: run_vjp<6>(t, rs, diag, n_diag, n, cf, n_draw, gloglike, state, cg, gresid, gdiag, gsign, gdiag_sum, gcr, gcc); break;
    default: return -1;
  }
  return cg.C;
}

int harness_gp_dot_tril(const double* t, const double* diag, int64_t n_diag, int64_t n, const double* real, int32_t n_real,
                        const double* cplx, int32_t n_complex, const int32_t* kind, int64_t n_draw, const double* x,
                        double* z) {
  const gp::Coefs cf{real, cplx, kind, n_real, n_complex, n > 0 ? t : nullptr};   // (as the device entry points)
  for (int64_t d = 0; d < n_draw; ++d) switch (cf.J()) {
      case 1: gp::dot_tril_lane<1>(t, diag, n_diag, n, cf, x, z, d); break;
      case 2: gp::dot_tril_lane<2>(t, diag, n_diag, n, cf, x, z, d); break;
      case 3: gp::dot_tril_lane<3>(t, diag, n_diag, n, cf, x, z, d); break;
      case 4: gp::dot_tril_lane<4>(t, diag, n_diag, n, cf, x, z, d); break;
      case 5: gp::dot_tril_lane<5>(t, diag, n_diag, n, cf, x, z, d); break;
      case 6: gp::dot_tril_lane<6>(t, diag, n_diag, n, cf, x, z, d); break;
      default: return -1;
    }
  return 0;
}

int harness_gp_predict(const double* t, int64_t n, const double* alpha, const double* real, int32_t n_real,
                       const double* cplx, int32_t n_complex, const int32_t* kind, int64_t n_draw, const double* tq,
                       int64_t m, double* mu) {
  const gp::Coefs cf{real, cplx, kind, n_real, n_complex, n > 0 ? t : nullptr};   // (as the device entry points)
  for (int64_t d = 0; d < n_draw; ++d) switch (cf.J()) {
      case 1: gp::predict_lane<1>(t, n, alpha, cf, tq, m, mu, d); break;
      case 2: gp::predict_lane<2>(t, n, alpha, cf, tq, m, mu, d); break;
      case 3: gp::predict_lane<3>(t, n, alpha, cf, tq, m, mu, d); break;
      case 4: gp::predict_lane<4>(t, n, alpha, cf, tq, m, mu, d); break;
      case 5: gp::predict_lane<5>(t, n, alpha, cf, tq, m, mu, d); break;
      case 6: gp::predict_lane<6>(t, n, alpha, cf, tq, m, mu, d); break;
      default: return -1;
    }
  return 0;
}

}  // extern "C"
