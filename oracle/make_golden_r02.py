"""Round-2 golden fixtures (tests/golden/lightcurves_mp.npz, tests/golden/gp_large.npz).
Run:  python -m oracle.make_golden_r02        (a few minutes; numpy + mpmath only)

* lightcurves_mp.npz -- END-TO-END light curves from oracle/mp_lightcurve.py (mpmath, 34 digits: the
  reference's formulas restated directly, sharing no code with the numpy / C ports or the kernels) for
  the BASELINE C4 system (4 planets, per-planet flux, 2048 cadences in four stretches around the
  planets' transits) and the C5 system (Kepler long cadence, exposure stencil x 7, transit +
  occultation, 2048 cadences = 15 orbits), plus C2 with light-travel delay off/on is NOT here (the
  delay has its own oracle path in numpy_port).
* gp_large.npz -- dense-Cholesky Gaussian log-likelihood AND its gradient with respect to y, diag and
  the celerite coefficients for SHO kernels (Q = 0.3, 1/sqrt 2, 3) at N = 500 and N = 2000, irregular
  sampling, in numpy.longdouble (x87 extended, 64-bit mantissa) with a hand-written Cholesky; the same
  code is cross-checked here against mpmath (oracle/mp_reference.py) at N = 60.
DATA only (inputs + expected outputs).
"""
import os

import mpmath as mp
import numpy as np

from . import mp_lightcurve as L
from . import mp_reference as R
from . import numpy_port as P

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
LD = np.longdouble


# ------------------------------------------------------------------------------------------ light curves
from .golden_cases import C4, C5  # noqa: E402,F401  (the tables: numpy only, shared with the tests)


def c4_times():
    """four stretches of 512 two-minute cadences, each centred on one planet's transit time"""
    dt = 2.0 / 1440.0
    return np.sort(np.concatenate([t0 + (np.arange(512) - 256) * dt for t0 in C4["t0"]]))


def golden_lightcurves():
    out = {}
    t = c4_times()
    flux = np.zeros((t.size, 4))
    for p in range(4):
        orbit = L.Orbit(C4["period"][p], t0=C4["t0"][p], b=C4["b"][p], ecc=C4["ecc"][p], omega=C4["omega"][p])
        flux[:, p] = [float(v) for v in L.light_curve(orbit, C4["r"][p], C4["u"], [mp.mpf(float(x)) for x in t])]
    out["c4_t"], out["c4_flux"] = t, flux
    t5 = np.arange(2048) * C5["texp"]
    orbit = L.Orbit(C5["period"], t0=C5["t0"], b=C5["b"], ecc=C5["ecc"], omega=C5["omega"])
    f5 = L.secondary_light_curve(orbit, C5["r"], C5["u_p"], C5["u_s"], C5["sbr"], [mp.mpf(float(x)) for x in t5],
                                 texp=C5["texp"], oversample=C5["oversample"], order=C5["order"])
    out["c5_t"], out["c5_flux"] = t5, np.array([float(v) for v in f5])
    np.savez_compressed(os.path.join(OUT, "lightcurves_mp.npz"), **out)
    return out


# ------------------------------------------------------------------------------------------ GP, long double
def cholesky_ld(K):
    """lower Cholesky factor in long double (LAPACK has none): row-oriented, vectorised inner products"""
    n = K.shape[0]
    Lm = np.zeros_like(K)
    for j in range(n):
        v = K[j:, j] - Lm[j:, :j] @ Lm[j, :j]
        Lm[j, j] = np.sqrt(v[0])
        Lm[j + 1:, j] = v[1:] / Lm[j, j]
    return Lm


def solve_lower_ld(Lm, B):
    X = np.array(B, dtype=LD, copy=True)
    for i in range(Lm.shape[0]):
        X[i] = (X[i] - Lm[i, :i] @ X[:i]) / Lm[i, i]
    return X


def solve_upper_ld(U, B):
    X = np.array(B, dtype=LD, copy=True)
    for i in range(U.shape[0] - 1, -1, -1):
        X[i] = (X[i] - U[i, i + 1:] @ X[i + 1:]) / U[i, i]
    return X


def kernel_ld(tau, co):
    ar, cr, ac, bc, cc, dc = co
    k = np.zeros_like(tau)
    for a, c in zip(ar, cr):
        k += a * np.exp(-c * tau)
    for a, b, c, d in zip(ac, bc, cc, dc):
        k += np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))
    return k


def gp_dense_ld(t, y, diag, co, with_grad=True):
    """log-likelihood and d/d(y, diag, coefficients) = 1/2 alpha^T dK alpha - 1/2 tr(K^-1 dK), long double"""
    t, y, diag = (np.asarray(x, dtype=LD) for x in (t, y, diag))
    co = [np.asarray(c, dtype=LD) for c in co]
    tau = np.abs(t[:, None] - t[None, :])
    K = kernel_ld(tau, co) + np.diag(diag)
    Lm = cholesky_ld(K)
    z = solve_lower_ld(Lm, y)
    alpha = solve_upper_ld(Lm.T.copy(), z)
    ll = -LD(0.5) * (z @ z) - np.sum(np.log(np.diag(Lm))) - LD(0.5) * t.size * np.log(2 * LD(np.pi))
    if not with_grad:
        return float(ll), None
    Linv = solve_lower_ld(Lm, np.eye(t.size, dtype=LD))
    Kinv = Linv.T @ Linv
    Wm = LD(0.5) * (np.outer(alpha, alpha) - Kinv)
    ar, cr, ac, bc, cc, dc = co
    g = {"y": -alpha, "diag": np.diag(Wm).copy()}
    g["ar"] = np.array([np.sum(Wm * np.exp(-c * tau)) for c in cr])
    g["cr"] = np.array([np.sum(Wm * (-tau) * a * np.exp(-c * tau)) for a, c in zip(ar, cr)])
    ga, gb, gc, gd = [], [], [], []
    for a, b, c, d in zip(ac, bc, cc, dc):
        ex, cs, sn = np.exp(-c * tau), np.cos(d * tau), np.sin(d * tau)
        ga.append(np.sum(Wm * ex * cs)); gb.append(np.sum(Wm * ex * sn))
        gc.append(np.sum(Wm * (-tau) * ex * (a * cs + b * sn)))
        gd.append(np.sum(Wm * ex * tau * (-a * sn + b * cs)))
    g["ac"], g["bc"], g["cc"], g["dc"] = (np.array(x, dtype=LD) for x in (ga, gb, gc, gd))
    return float(ll), {k: np.asarray(v, dtype=np.float64) for k, v in g.items()}


def golden_gp():
    rng = np.random.default_rng(9)
    out = {}
    # the long-double code against mpmath at N = 60
    t = np.sort(rng.uniform(0, 10, 60)); y = rng.normal(size=60); diag = 0.1 + 0.1 * rng.uniform(size=60)
    co = P.sho_coefficients(*P.sho_from_sigma_rho(0.8, 3.0, 3.0), 3.0)
    want = R.gp_loglike_dense([mp.mpf(float(v)) for v in t], [mp.mpf(float(v)) for v in y], [mp.mpf(float(v)) for v in diag],
                              *[[mp.mpf(float(v)) for v in c] for c in co])
    got, _ = gp_dense_ld(t, y, diag, co, with_grad=False)
    assert abs(got - float(want)) <= 1e-15 * abs(float(want)), (got, float(want))
    for N in (500, 2000):
        for tag, Q in (("q03", 0.3), ("q07", 1 / np.sqrt(2)), ("q3", 3.0)):
            if N == 2000 and tag != "q07":
                continue
            t = np.sort(rng.uniform(0, 0.05 * N, N)); y = rng.normal(size=N); diag = 0.1 + 0.1 * rng.uniform(size=N)
            co = P.sho_coefficients(*P.sho_from_sigma_rho(0.8, 3.0, Q), Q)
            ll, g = gp_dense_ld(t, y, diag, co)
            key = f"n{N}_{tag}"
            out[f"{key}_t"], out[f"{key}_y"], out[f"{key}_diag"] = t, y, diag
            for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
                out[f"{key}_{nm}"] = np.asarray(c, dtype=np.float64)
                out[f"{key}_g{nm}"] = g[nm]
            out[f"{key}_loglike"] = ll
            out[f"{key}_gy"], out[f"{key}_gdiag"] = g["y"], g["diag"]
            print(key, ll)
    np.savez_compressed(os.path.join(OUT, "gp_large.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    golden_gp()
    golden_lightcurves()
    for f in ("lightcurves_mp.npz", "gp_large.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
