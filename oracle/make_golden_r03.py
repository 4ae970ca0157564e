#!/usr/bin/env python
"""Round-3 fixtures (TEST INFRASTRUCTURE; see oracle/__init__.py): the celerite log-likelihood AND its gradient in the
regimes round 2 sent to the sequential kernels -- terms the time-parallel filter form could not take or conditioned
badly -- from the dense definition in x87 long double (oracle/make_golden_r02.gp_dense_ld: hand-written Cholesky,
itself pinned to mpmath there).  N = 500 irregular cadences each:

  q0505     SHO, Q = 0.505: 1 % above critical damping (b / a = 1 / f = 7)
  q0495     SHO, Q = 0.495: 1 % below -- two real terms, one of NEGATIVE amplitude (a pair slot of kind 1)
  q045      SHO, Q = 0.45
  q02       SHO, Q = 0.2 (deeply over-damped)
  matern    celerite2's Matern32Term(eps = 0.01): one complex term with b / a = w0 / eps = 58
  snr1e6    SHO, Q = 1 / sqrt 2, signal variance 1e6 x the white noise (conditioning score 2e6)
  rotation  celerite2's RotationTerm (two SHO terms, Q0 = 0.02: the second mode at Q = 0.52), J = 4

-> tests/golden/gp_hard.npz.  Run from the repository root:  python oracle/make_golden_r03.py
reference: celerite2 is a dependency of the reference (setup.py:36), not in its tree; the kernels are its published ones
(SURVEY.md Appendix B)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import numpy_port as P  # noqa: E402
from oracle.make_golden_r02 import gp_dense_ld  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rotation_coefficients(sigma, period, Q0, dQ, f):
    """celerite2.terms.RotationTerm: two SHO terms at the rotation period and its first harmonic"""
    amp = sigma ** 2 / (1 + f)
    Q1 = 0.5 + Q0 + dQ
    w1 = 4 * np.pi * Q1 / (period * np.sqrt(4 * Q1 ** 2 - 1))
    S1 = amp / (w1 * Q1)
    Q2 = 0.5 + Q0
    w2 = 8 * np.pi * Q2 / (period * np.sqrt(4 * Q2 ** 2 - 1))
    S2 = f * amp / (w2 * Q2)
    parts = [P.sho_coefficients(S1, w1, Q1), P.sho_coefficients(S2, w2, Q2)]
    return tuple(np.concatenate(x) for x in zip(*parts))


def matern32_coefficients(sigma, rho, eps=0.01):
    """celerite2.terms.Matern32Term"""
    w0 = np.sqrt(3.0) / rho
    S0 = sigma ** 2 / w0
    e = np.zeros(0)
    return e, e, np.array([w0 * S0]), np.array([w0 * w0 * S0 / eps]), np.array([w0]), np.array([eps])


def main():
    rng = np.random.default_rng(31)
    N = 500
    out = {}
    sigma, rho = 0.8, 3.0
    cases = {
        "q0505": (P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, 0.505), 0.505), 0.01 * sigma ** 2),
        "q0495": (P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, 0.495), 0.495), 0.01 * sigma ** 2),
        "q045": (P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, 0.45), 0.45), 0.01 * sigma ** 2),
        "q02": (P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, 0.2), 0.2), 0.1 * sigma ** 2),
        "matern": (matern32_coefficients(sigma, rho), 0.01 * sigma ** 2),
        "snr1e6": (P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, 1 / np.sqrt(2)), 1 / np.sqrt(2)), 1e-6 * sigma ** 2),
        "rotation": (rotation_coefficients(sigma, 4.0, 0.02, 0.5, 0.5), 0.01 * sigma ** 2),
    }
    for key, (co, noise) in cases.items():
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        diag = noise * (1.0 + rng.uniform(size=N))
        K = P.celerite_kernel(np.abs(t[:, None] - t[None, :]), *co) + np.diag(diag)
        y = np.linalg.cholesky(K) @ rng.normal(size=N)          # a draw from the process itself
        ll, g = gp_dense_ld(t, y, diag, co)
        out[f"{key}_t"], out[f"{key}_y"], out[f"{key}_diag"], out[f"{key}_loglike"] = t, y, diag, ll
        for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
            out[f"{key}_{nm}"] = np.asarray(c, dtype=np.float64)
            out[f"{key}_g{nm}"] = g[nm]
        out[f"{key}_gy"], out[f"{key}_gdiag"] = g["y"], g["diag"]
        print(key, ll, [np.asarray(c).size for c in co])
    np.savez_compressed(os.path.join(OUT, "gp_hard.npz"), **out)
    print("gp_hard.npz", os.path.getsize(os.path.join(OUT, "gp_hard.npz")))


if __name__ == "__main__":
    main()
