/* exo_oracle.c -- plain C port of oracle/numpy_port.py (TEST INFRASTRUCTURE ONLY).
 *
 * Scalar, single-threaded CPU restatement of the hot path, used (a) as the
 * checker at sizes the numpy port is too slow for and (b) as bench.py's
 * `cpu_baseline` (kind "port").  Parity status: **parity unpinned** against
 * exoplanet-core / celerite2 (absent, see oracle/__init__.py); pinned against
 * oracle/mp_reference.py through tests/test_oracle.py.
 *
 * Algorithms (same citations as numpy_port.py):
 *   kepler        Markley 1995 starter + 5th-order step; call site
 *                 /root/reference/src/exoplanet/orbits/keplerian.py:333
 *   quad_sv       disk integrals via Green's theorem + Bulirsch cel; call site
 *                 /root/reference/src/exoplanet/light_curves/limb_dark.py:24
 *   transit       glue of keplerian.py:324-334,400-409,283-314,729-769 and
 *                 limb_dark.py:178-252, secondary_eclipse.py:45-70
 *   celerite      Foreman-Mackey et al. 2017 / Foreman-Mackey 2018 recurrences
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.14159265358979323846
#define TWO_PI_HI 6.283185307179586
#define TWO_PI_LO 2.4492935982947064e-16

/* ------------------------------------------------------------------ helpers */
static double x_minus_sin(double x) {
  if (fabs(x) < 0.9) {
    double x2 = x * x;
    double s = 1 - x2 / 272;
    s = 1 - x2 / 210 * s; s = 1 - x2 / 156 * s; s = 1 - x2 / 110 * s;
    s = 1 - x2 / 72 * s;  s = 1 - x2 / 42 * s;  s = 1 - x2 / 20 * s;
    return x * x2 / 6 * s;
  }
  return x - sin(x);
}

/* ------------------------------------------------------------------ kepler */
/* eccentric anomaly in [-pi, pi] for the reduced mean anomaly */
static double kepler_E(double M, double e) {
  double k = rint(M / TWO_PI_HI);
  double Mr = fma(-k, TWO_PI_HI, M);
  Mr = fma(-k, TWO_PI_LO, Mr);
  double sgn = Mr < 0 ? -1.0 : 1.0;
  Mr = fabs(Mr);
  if (e == 0.0) return sgn * Mr;
  double ome = 1 - e;
  double alpha = (3 * PI * PI + 1.6 * PI * (PI - Mr) / (1 + e)) / (PI * PI - 6);
  double d = 3 * ome + alpha * e;
  double q = 2 * alpha * d * ome - Mr * Mr;
  double r = 3 * alpha * d * (d - ome) * Mr + Mr * Mr * Mr;
  double w = pow(fabs(r) + sqrt(q * q * q + r * r), 2.0 / 3.0);
  double E = (2 * r * w / (w * w + w * q + q * q) + Mr) / d;
  double sE = sin(E), cE = cos(E);
  double f0 = ome * E + e * x_minus_sin(E) - Mr;
  double f1 = 1 - e * cE, f2 = e * sE, f3 = 1 - f1, f4 = -f2;
  double d3 = -f0 / (f1 - 0.5 * f0 * f2 / f1);
  double d4 = -f0 / (f1 + 0.5 * d3 * f2 + d3 * d3 * f3 / 6);
  double d5 = -f0 / (f1 + 0.5 * d4 * f2 + d4 * d4 * f3 / 6 + d4 * d4 * d4 * f4 / 24);
  return sgn * (E + d5);
}

void oracle_kepler(const double* M, const double* ecc, double* sinf, double* cosf, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    double e = ecc[i];
    if (!(e >= 0 && e < 1)) { sinf[i] = cosf[i] = NAN; continue; }
    double E = kepler_E(M[i], e);
    double X = sqrt(1 - e) * cos(0.5 * E), Y = sqrt(1 + e) * sin(0.5 * E);
    double den = X * X + Y * Y;
    cosf[i] = (X * X - Y * Y) / den;
    sinf[i] = 2 * X * Y / den;
  }
}

/* ------------------------------------------------------------------ cel */
static double cel(double kc, double p, double a, double b) {
  kc = fabs(kc);
  if (kc < 1e-8) kc = 1e-8;
  double e = kc, em = 1.0;
  if (p > 0) { p = sqrt(p); b /= p; }
  else {
    double f = kc * kc, q = 1 - f, g = 1 - p;
    f -= p; q *= (b - a * p); p = sqrt(f / g); a = (a - b) / g; b = -q / (g * g * p) + a * p;
  }
  for (int it = 0; it < 40; ++it) {
    double f = a;
    a += b / p;
    double g = e / p;
    b += f * g; b += b;
    p = g + p;
    g = em; em += kc;
    if (fabs(g - kc) <= g * 1e-8) break;
    kc = 2 * sqrt(e);
    e = kc * em;
  }
  return 0.5 * PI * (b + a * em) / (em * (em + p));
}

static double int_cos4(double k2, double kc2, double B, double D) {
  if (k2 < 0.1) {
    /* Maclaurin series, coefficients by recurrence */
    double cj = 1.0, Ij = 0.375, pw = 1.0, s = 0.0;
    for (int j = 0; j < 24; ++j) {
      if (j > 0) { cj *= (2.0 * j - 1) / (2.0 * j); Ij *= (2.0 * j - 1) / (2.0 * j + 4); pw *= k2; }
      s += cj * Ij * pw;
    }
    return 0.5 * PI * s;
  }
  return ((3 * k2 - 1) * B + kc2 * D) / (3 * k2);
}

static double i4_num(double k) {
  if (fabs(k) < 0.4) {
    static const double c[8] = {24.0 / 120, 120.0 / 5040, 504.0 / 362880, 2040.0 / 39916800,
                                8184.0 / 6227020800.0, 32760.0 / 1307674368000.0,
                                131064.0 / 355687428096000.0, 524280.0 / 121645100408832000.0};
    double k2 = k * k, s = c[7];
    for (int j = 6; j >= 0; --j) s = c[j] - k2 * s;
    return s * k2 * k2 * k;
  }
  return 8 * x_minus_sin(k) - x_minus_sin(2 * k);
}

/* s[3], dsdb[3], dsdr[3] for b >= 0 */
static void quad_sv_one(double b, double r, double* s, double* db, double* dr, int grad) {
  s[0] = PI; s[1] = 2 * PI / 3; s[2] = 0;
  if (grad) { db[0] = db[1] = db[2] = dr[0] = dr[1] = dr[2] = 0; }
  if (isnan(b) || isnan(r)) {
    s[0] = s[1] = s[2] = NAN;
    if (grad) { db[0] = db[1] = db[2] = dr[0] = dr[1] = dr[2] = NAN; }
    return;
  }
  if (r <= 0 || b >= 1 + r) return;
  if (r >= 1 + b) { s[0] = s[1] = s[2] = 0; return; }
  double r2 = r * r, b2 = b * b;
  int inside = b + r <= 1;
  double x = fmax(b, r), y = fmin(b, r);
  double A = ((1 - x) + y) * (1 + (x - y));
  double Bm = ((x - 1) + y) * ((x + y) + 1);
  double sqA = sqrt(A), br = b * r, rmb = r - b;
  double k0 = PI, u0 = 0.5 * PI, I2 = 0.25 * PI, I4 = 0.1875 * PI, sink0 = 0, seg = PI * r2;
  if (!inside) {
    double kite = sqrt(fmax(0.0, A * Bm));
    k0 = atan2(kite, b2 + (r - 1) * (r + 1));
    double k1 = atan2(kite, (1 - r) * (1 + r) + b2);
    sink0 = kite / (2 * br);
    u0 = 0.5 * k0;
    I2 = 0.25 * x_minus_sin(k0);
    I4 = i4_num(k0) / 32;
    seg = 0.5 * (r2 * x_minus_sin(2 * k0) + x_minus_sin(2 * k1));
  }
  double q = 1 - 2 * rmb * rmb;
  s[0] = PI - seg;
  s[2] = -4 * r * (A * rmb * u0 + (2 * b * A - 4 * br * rmb) * I2 - 8 * b2 * r * I4);
  int same = b == r;
  double theta = r > b ? 1.0 : (same ? 0.5 : 0.0);
  double J, dr1, db1;
  if (inside) {
    double m = 4 * br / A, kc2 = fmax(-Bm / A, 0.0), kc = sqrt(kc2);
    double B = cel(kc, 1, 1, 0), D = cel(kc, 1, 0, 1);
    double E = B + kc2 * D, K = B + D;
    double t3 = (2 * (2 - m) * E - kc2 * K) / 3;
    J = (2 * sqA / 3) * (A * t3 - (r2 - b2) * E);
    if (!same) {
      double pp = (b + r) / rmb;
      J += (2 * (r + b) / (3 * sqA * rmb)) * cel(kc, pp * pp, A, -Bm);
    }
    dr1 = -4 * r * sqA * E;
    db1 = 4 * r * sqA * (B - kc2 * D) / 3;
  } else {
    double k2 = fmin(A / (4 * br), 1.0), kc2 = fmax(1 - k2, 0.0), kc = sqrt(kc2);
    double B = cel(kc, 1, 1, 0), D = cel(kc, 1, 0, 1);
    double C2 = B, C4 = int_cos4(k2, kc2, B, D);
    double pref = 4 * sqA * sqrt(k2);
    J = (pref / 6) * (A * C4 - (r2 - b2) * C2);
    if (!same) J += ((r + b) / (6 * rmb)) * pref * cel(kc, 1 / (rmb * rmb), 1, 0);
    dr1 = -r * pref * C2;
    db1 = r * pref * (C2 * (1 - 2 * k2) + 2 * k2 * C4);
  }
  s[1] = 2 * PI / 3 * (1 - theta) + J;
  if (grad) {
    dr[0] = -2 * r * k0;
    db[0] = 2 * r * sink0;
    dr[2] = -r * (4 * k0 * q - 64 * br * I2);
    db[2] = -4 * r * (-2 * q * u0 + (4 * q + 16 * br) * I2 - 32 * br * I4);
    dr[1] = dr1;
    db[1] = db1;
  }
}

void oracle_quad_sv(const double* b, const double* r, double* s, double* dsdb, double* dsdr, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    double db[3], dr[3];
    quad_sv_one(fabs(b[i]), r[i], s + 3 * i, db, dr, 1);
    double sg = b[i] < 0 ? -1.0 : 1.0;
    for (int j = 0; j < 3; ++j) { dsdb[3 * i + j] = sg * db[j]; dsdr[3 * i + j] = dr[j]; }
  }
}

/* ------------------------------------------------------------------ fused transit */
enum { P_N = 0, P_TP, P_ECC, P_COSW, P_SINW, P_COSI, P_SINI, P_AOR, P_ROR, P_T0, P_PERIOD, P_TS, P_TE,
       P_FRATIO, P_TS2, P_TE2, P_CLIGHT, P_RES1, P_RES2, P_RES3, NPAR };   /* include/exoplanet_amd.h: EXO_NPAR = 20 */
#define FLAG_PER_PLANET 1u
#define FLAG_WINDOW 2u
#define FLAG_SECONDARY 4u

static int in_window(double t, const double* rec, double htexp, int secondary) {
  double P = rec[P_PERIOD], hp = 0.5 * P;
  double x = t - rec[P_T0] + hp;
  double dt = x - P * floor(x / P) - hp;
  int in = dt >= rec[P_TS] - htexp && dt <= rec[P_TE] + htexp;
  if (secondary) {
    double y = t - rec[P_T0];
    y = y - P * floor(y / P);
    in = in || (y >= rec[P_TS2] - htexp && y <= rec[P_TE2] + htexp);
  }
  return in;
}

/* one sample; returns F, accumulates gw * dF/dtheta into g[NPAR], gc[6] when g != NULL */
static double sample(double tt, const double* rec, const double* c, int secondary, double gw, double* g,
                     double* gc) {
  double n = rec[P_N], tp = rec[P_TP], e = rec[P_ECC], cw = rec[P_COSW], sw = rec[P_SINW];
  double ci = rec[P_COSI], si = rec[P_SINI], aor = rec[P_AOR], ror = rec[P_ROR], fr = rec[P_FRATIO];
  if (!(e >= 0 && e < 1)) return NAN;
  double M = (tt - tp) * n;
  double E = kepler_E(M, e);
  double sh = sin(0.5 * E), ch = cos(0.5 * E);
  double se = sqrt(1 - e), pe = sqrt(1 + e);
  double X = se * ch, Y = pe * sh;
  double den = X * X + Y * Y, cx = X * X - Y * Y, sx = 2 * X * Y;
  double xo = -aor * cx, yo = -aor * sx;
  double x1 = cw * xo - sw * yo, y1 = sw * xo + cw * yo;
  double Ys = ci * y1, Z = -si * y1;
  double b2 = x1 * x1 + Ys * Ys, lim = 1 + ror;
  int front = Z > 0, behind = secondary && Z < 0;
  if (!(front || behind) || b2 >= lim * lim) return 0.0;
  double b = sqrt(b2);
  int occ = behind;
  double bq = occ ? b / ror : b, rq = occ ? 1 / ror : ror;
  double s[3], db[3], dr[3];
  quad_sv_one(bq, rq, s, db, dr, g != NULL);
  const double* cc = occ ? c + 3 : c;
  double Fq = s[0] * cc[0] + s[1] * cc[1] + s[2] * cc[2] - 1.0;
  double wq = secondary ? (occ ? fr / (1 + fr) : 1 / (1 + fr)) : 1.0;
  if (g) {
    double gq = gw * wq;
    int o = occ ? 3 : 0;
    for (int j = 0; j < 3; ++j) gc[o + j] += gq * s[j];
    double bbar_q = gq * (db[0] * cc[0] + db[1] * cc[1] + db[2] * cc[2]);
    double rbar_q = gq * (dr[0] * cc[0] + dr[1] * cc[1] + dr[2] * cc[2]);
    double bbar, rorbar;
    if (occ) {
      bbar = bbar_q / ror;
      rorbar = -(bbar_q * b + rbar_q) / (ror * ror);
      g[P_FRATIO] += gw * Fq / ((1 + fr) * (1 + fr));
    } else {
      bbar = bbar_q;
      rorbar = rbar_q;
      if (secondary) g[P_FRATIO] -= gw * Fq / ((1 + fr) * (1 + fr));
    }
    g[P_ROR] += rorbar;
    double ib = b > 0 ? 1 / b : 0.0;
    double x1bar = bbar * x1 * ib, Ysbar = bbar * Ys * ib, y1bar = Ysbar * ci;
    g[P_COSI] += Ysbar * y1;
    double xobar = cw * x1bar + sw * y1bar, yobar = -sw * x1bar + cw * y1bar;
    g[P_COSW] += x1bar * xo + y1bar * yo;
    g[P_SINW] += -x1bar * yo + y1bar * xo;
    g[P_AOR] += -(xobar * cx + yobar * sx);
    double cxbar = -aor * xobar, sxbar = -aor * yobar;
    double sinE = 2 * sh * ch, cosE = ch * ch - sh * sh, sq = se * pe;
    double Ebar = -sinE * cxbar + sq * cosE * sxbar;
    double Mbar = Ebar / den;
    g[P_ECC] += Mbar * sinE - cxbar - e * sinE / sq * sxbar;
    g[P_N] += Mbar * (tt - tp);
    g[P_TP] -= Mbar * n;
  }
  return Fq * wq;
}

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oracle_max_threads(void) { return omp_get_max_threads(); }
#else
void oracle_set_threads(int n) { (void)n; }
int oracle_max_threads(void) { return 1; }
#endif

/* value (+ optional vjp).  gflux == NULL -> forward only. */
void oracle_transit(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* sdt,
                    const double* sw, int32_t n_sub, const double* params, const double* ld, int64_t n_draw,
                    int32_t n_planet, uint32_t flags, const double* gflux, double* flux, double* gparams,
                    double* gld) {
  int secondary = (flags & FLAG_SECONDARY) != 0, per_planet = (flags & FLAG_PER_PLANET) != 0;
  int window = (flags & FLAG_WINDOW) != 0;
  int nld = secondary ? 6 : 3;
  double one = 1.0, zero = 0.0;
  if (n_texp == 0) { sdt = &zero; sw = &one; n_sub = 1; }
  if (gflux) {
    memset(gparams, 0, sizeof(double) * n_draw * n_planet * NPAR);
    memset(gld, 0, sizeof(double) * n_draw * nld);
  }
  /* draws are independent (their gradient rows too): the all-core CPU baseline of bench.py runs one
   * draw per thread -- the analogue of PyMC's one process per chain on every host core.  The
   * thread count is OMP_NUM_THREADS / oracle_set_threads(); one thread = the scalar port.       */
#pragma omp parallel for schedule(dynamic, 1) if (n_draw > 1)
  for (int64_t d = 0; d < n_draw; ++d) {
    const double* c = ld + d * nld;
    for (int64_t i = 0; i < n_cad; ++i) {
      double te = n_texp == 0 ? 0.0 : (n_texp == 1 ? texp[0] : texp[i]);
      double fsum = 0.0;
      for (int p = 0; p < n_planet; ++p) {
        const double* rec = params + (d * n_planet + p) * NPAR;
        double f = 0.0;
        if (!window || in_window(t[i], rec, 0.5 * te, secondary)) {
          double g0 = 0.0;
          if (gflux) g0 = per_planet ? gflux[(d * n_cad + i) * n_planet + p] : gflux[d * n_cad + i];
          for (int k = 0; k < n_sub; ++k) {
            double tt = t[i] + te * sdt[k];
            f += sw[k] * sample(tt, rec, c, secondary, g0 * sw[k],
                                gflux ? gparams + (d * n_planet + p) * NPAR : NULL, gflux ? gld + d * nld : NULL);
          }
        }
        if (per_planet) { if (flux) flux[(d * n_cad + i) * n_planet + p] = f; }
        else fsum += f;
      }
      if (!per_planet && flux) flux[d * n_cad + i] = fsum;
    }
  }
}

/* The same with transit-timing tables (reference: orbits/ttv.py:158-187): every time -- cadence
 * or sub-exposure -- of planet p acts as t - shift[bin(t)], bin(t) = #{edges < t} (searchsorted,
 * left); the caller's windows are tested on the warped mid-exposure time.  edges [n_draw][n_planet]
 * [n_edge] (ascending, +inf padded), shift [..][n_edge + 1]; gshift (with gflux) receives the
 * cotangent of shift: the t_periastron term of each sample, by bin.                            */
static int ttv_bin(const double* edges, int n_edge, double t) {
  int lo = 0, hi = n_edge;             /* first index with edges[i] >= t */
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (edges[mid] < t) lo = mid + 1; else hi = mid;
  }
  return lo;
}

void oracle_transit_ttv(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* sdt,
                        const double* sw, int32_t n_sub, const double* params, const double* ld, int64_t n_draw,
                        int32_t n_planet, uint32_t flags, const double* edges, const double* shift, int32_t n_edge,
                        const double* gflux, double* flux, double* gparams, double* gld, double* gshift) {
  int secondary = (flags & FLAG_SECONDARY) != 0, per_planet = (flags & FLAG_PER_PLANET) != 0;
  int window = (flags & FLAG_WINDOW) != 0;
  int nld = secondary ? 6 : 3;
  double one = 1.0, zero = 0.0;
  if (n_texp == 0) { sdt = &zero; sw = &one; n_sub = 1; }
  if (gflux) {
    memset(gparams, 0, sizeof(double) * n_draw * n_planet * NPAR);
    memset(gld, 0, sizeof(double) * n_draw * nld);
    memset(gshift, 0, sizeof(double) * n_draw * n_planet * (n_edge + 1));
  }
  for (int64_t d = 0; d < n_draw; ++d) {
    const double* c = ld + d * nld;
    for (int64_t i = 0; i < n_cad; ++i) {
      double te = n_texp == 0 ? 0.0 : (n_texp == 1 ? texp[0] : texp[i]);
      double fsum = 0.0;
      for (int p = 0; p < n_planet; ++p) {
        const double* rec = params + (d * n_planet + p) * NPAR;
        const double* ed = edges + (d * n_planet + p) * (int64_t)n_edge;
        const double* sh = shift + (d * n_planet + p) * (int64_t)(n_edge + 1);
        double f = 0.0;
        if (!window || in_window(t[i] - sh[ttv_bin(ed, n_edge, t[i])], rec, 0.5 * te, secondary)) {
          double g0 = 0.0;
          if (gflux) g0 = per_planet ? gflux[(d * n_cad + i) * n_planet + p] : gflux[d * n_cad + i];
          for (int k = 0; k < n_sub; ++k) {
            double tt = t[i] + te * sdt[k];
            int b = ttv_bin(ed, n_edge, tt);
            double gtmp[NPAR] = {0};
            f += sw[k] * sample(tt - sh[b], rec, c, secondary, g0 * sw[k], gflux ? gtmp : NULL,
                                gflux ? gld + d * nld : NULL);
            if (gflux) {
              double* gp = gparams + (d * n_planet + p) * NPAR;
              for (int q = 0; q < NPAR; ++q) gp[q] += gtmp[q];
              gshift[(d * n_planet + p) * (int64_t)(n_edge + 1) + b] += gtmp[P_TP];
            }
          }
        }
        if (per_planet) { if (flux) flux[(d * n_cad + i) * n_planet + p] = f; }
        else fsum += f;
      }
      if (!per_planet && flux) flux[d * n_cad + i] = fsum;
    }
  }
}

/* ------------------------------------------------------------------ celerite
 * log-likelihood and its reverse recurrence for ONE draw (SURVEY Appendix B;
 * adjoint derived as in numpy_port.py / docs/DESIGN_r1_r4.md 3.4).  J <= 8.
 *   coef_real [n_real][2] = (a, c), coef_complex [n_complex][4] = (a, b, c, d)
 * If gresid != NULL also writes gresid[n], gdiag[n], gcoef_real, gcoef_complex
 * for d loglike.  Returns loglike (-inf if not positive definite).           */
#define JMAX 16   /* (round 5: state widths up to 16, as the library's sequential kernels) */
static void make_uv(int J, int n_real, const double* ka, const double* kb, const double* kd, double t,
                    double* U, double* V) {
  for (int j = 0; j < J; ++j) {
    if (j < n_real) { U[j] = ka[j]; V[j] = 1.0; }
    else if (((j - n_real) & 1) == 0) {
      double c = cos(kd[j] * t), s = sin(kd[j] * t);
      U[j] = ka[j] * c + kb[j] * s; U[j + 1] = ka[j] * s - kb[j] * c;
      V[j] = c; V[j + 1] = s;
    }
  }
}

double oracle_celerite(const double* t, const double* y, const double* diag, int64_t n, const double* coef_real,
                       int32_t n_real, const double* coef_complex, int32_t n_complex, double* gresid,
                       double* gdiag, double* gcoef_real, double* gcoef_complex) {
  const int J = n_real + 2 * n_complex;
  double ka[JMAX], kb[JMAX], kc[JMAX], kd[JMAX], asum = 0;
  for (int j = 0; j < J; ++j) {
    if (j < n_real) { ka[j] = coef_real[2 * j]; kb[j] = 0; kc[j] = coef_real[2 * j + 1]; kd[j] = 0; asum += ka[j]; }
    else {
      const double* p = coef_complex + 4 * ((j - n_real) >> 1);
      ka[j] = p[0]; kb[j] = p[1]; kc[j] = p[2]; kd[j] = p[3];
      if (((j - n_real) & 1) == 0) asum += p[0];
    }
  }
  const int NS = 2 + 2 * J + J * J;            /* d, z, W, F, S (full) per cadence */
  double* st = (double*)malloc(sizeof(double) * (size_t)n * NS);
  double S[JMAX][JMAX] = {{0}}, F[JMAX] = {0}, W[JMAX], U[JMAX], V[JMAX], P[JMAX];
  make_uv(J, n_real, ka, kb, kd, t[0], U, V);
  double d = diag[0] + asum, z = y[0], acc;
  int bad = !(d > 0);
  for (int j = 0; j < J; ++j) W[j] = V[j] / d;
  acc = z * z / d + log(d);
#define SAVE(i)                                                        \
  do {                                                                 \
    double* q = st + (size_t)(i) * NS;                                 \
    q[0] = d; q[1] = z;                                                \
    for (int j = 0; j < J; ++j) { q[2 + j] = W[j]; q[2 + J + j] = F[j]; } \
    for (int j = 0; j < J; ++j) for (int l = 0; l < J; ++l) q[2 + 2 * J + j * J + l] = S[j][l]; \
  } while (0)
  SAVE(0);
  for (int64_t i = 1; i < n; ++i) {
    double dt = t[i] - t[i - 1];
    for (int j = 0; j < J; ++j) P[j] = exp(-kc[j] * dt);
    for (int j = 0; j < J; ++j) {
      F[j] = P[j] * (F[j] + W[j] * z);
      for (int l = 0; l < J; ++l) S[j][l] = P[j] * P[l] * (S[j][l] + d * W[j] * W[l]);
    }
    make_uv(J, n_real, ka, kb, kd, t[i], U, V);
    double u[JMAX], dn = diag[i] + asum, zf = y[i];
    for (int j = 0; j < J; ++j) {
      double s = 0;
      for (int l = 0; l < J; ++l) s += S[j][l] * U[l];
      u[j] = s; dn -= U[j] * s; zf -= U[j] * F[j];
    }
    d = dn; z = zf;
    if (!(d > 0)) bad = 1;
    for (int j = 0; j < J; ++j) W[j] = (V[j] - u[j]) / d;
    acc += z * z / d + log(d);
    SAVE(i);
  }
  double ll = bad ? -INFINITY : -0.5 * acc - 0.5 * (double)n * log(2 * PI);
  if (gresid) {
    double Sb[JMAX][JMAX] = {{0}}, Fb[JMAX] = {0}, Wb[JMAX] = {0}, db = 0, zb = 0, gasum = 0;
    double ga[JMAX] = {0}, gb[JMAX] = {0}, gc[JMAX] = {0}, gd[JMAX] = {0};
    for (int64_t i = n - 1; i >= 1; --i) {
      const double* q = st + (size_t)i * NS; const double* p = st + (size_t)(i - 1) * NS;
      double d_n = q[0], z_n = q[1], d_p = p[0], z_p = p[1];
      const double *W_n = q + 2, *F_n = q + 2 + J, *S_n = q + 2 + 2 * J, *W_p = p + 2, *F_p = p + 2 + J, *S_p = p + 2 + 2 * J;
      double dt = t[i] - t[i - 1];
      for (int j = 0; j < J; ++j) P[j] = exp(-kc[j] * dt);
      make_uv(J, n_real, ka, kb, kd, t[i], U, V);
      double zbar = zb - z_n / d_n, dbar = db + 0.5 * z_n * z_n / (d_n * d_n) - 0.5 / d_n;
      gresid[i] = zbar;
      double Ub[JMAX], Vb[JMAX], ub[JMAX], u[JMAX], wdot = 0;
      for (int j = 0; j < J; ++j) {
        Ub[j] = -zbar * F_n[j]; Fb[j] -= zbar * U[j];
        Vb[j] = Wb[j] / d_n; ub[j] = -Vb[j]; wdot += Wb[j] * W_n[j];
      }
      dbar -= wdot / d_n;
      gdiag[i] = dbar; gasum += dbar;
      for (int j = 0; j < J; ++j) { double s = 0; for (int l = 0; l < J; ++l) s += S_n[j * J + l] * U[l]; u[j] = s; }
      for (int j = 0; j < J; ++j) { Ub[j] -= dbar * u[j]; ub[j] -= dbar * U[j]; }
      for (int j = 0; j < J; ++j) {
        double s = 0;
        for (int l = 0; l < J; ++l) { Sb[j][l] += ub[j] * U[l]; s += S_n[l * J + j] * ub[l]; }
        Ub[j] += s;
      }
      double Pb[JMAX], Gb[JMAX], Wbp[JMAX], zbp = 0, dbp = 0;
      for (int j = 0; j < J; ++j) {
        double G = F_p[j] + W_p[j] * z_p;
        Pb[j] = Fb[j] * G; Gb[j] = Fb[j] * P[j]; Wbp[j] = Gb[j] * z_p; zbp += Gb[j] * W_p[j];
      }
      for (int j = 0; j < J; ++j) {
        double wsum = 0, psum = 0;
        for (int l = 0; l < J; ++l) {
          double T = S_p[j * J + l] + d_p * W_p[j] * W_p[l], sym = Sb[j][l] + Sb[l][j];
          psum += sym * T * P[l];
          wsum += sym * P[j] * P[l] * W_p[l];
        }
        Pb[j] += psum; Wbp[j] += d_p * wsum;
      }
      for (int j = 0; j < J; ++j) for (int l = 0; l < J; ++l) {
        double Tb = Sb[j][l] * P[j] * P[l];
        dbp += Tb * W_p[j] * W_p[l]; Sb[j][l] = Tb;
      }
      for (int j = 0; j < J; ++j) {
        gc[j] -= dt * P[j] * Pb[j];
        if (j < n_real) ga[j] += Ub[j];
        else if (((j - n_real) & 1) == 0) {
          double c = V[j], s = V[j + 1];
          ga[j] += Ub[j] * c + Ub[j + 1] * s; gb[j] += Ub[j] * s - Ub[j + 1] * c;
          gd[j] += t[i] * (Ub[j] * (-ka[j] * s + kb[j] * c) + Ub[j + 1] * (ka[j] * c + kb[j] * s) - Vb[j] * s + Vb[j + 1] * c);
        }
      }
      db = dbp; zb = zbp;
      for (int j = 0; j < J; ++j) { Fb[j] = Gb[j]; Wb[j] = Wbp[j]; }
    }
    {
      const double* q = st; double d_n = q[0], z_n = q[1]; const double* W_n = q + 2;
      double zbar = zb - z_n / d_n, dbar = db + 0.5 * z_n * z_n / (d_n * d_n) - 0.5 / d_n, wdot = 0, Vb[JMAX];
      gresid[0] = zbar;
      make_uv(J, n_real, ka, kb, kd, t[0], U, V);
      for (int j = 0; j < J; ++j) { Vb[j] = Wb[j] / d_n; wdot += Wb[j] * W_n[j]; }
      dbar -= wdot / d_n; gdiag[0] = dbar; gasum += dbar;
      for (int j = n_real; j + 1 < J; j += 2) gd[j] += t[0] * (-Vb[j] * V[j + 1] + Vb[j + 1] * V[j]);
    }
    for (int j = 0; j < J; ++j) {
      if (j < n_real) { gcoef_real[2 * j] = ga[j] + gasum; gcoef_real[2 * j + 1] = gc[j]; }
      else if (((j - n_real) & 1) == 0) {
        double* o = gcoef_complex + 4 * ((j - n_real) >> 1);
        o[0] = ga[j] + gasum; o[1] = gb[j]; o[2] = gc[j] + gc[j + 1]; o[3] = gd[j];
      }
    }
  }
  free(st);
  return ll;
}
