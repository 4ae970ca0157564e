#!/usr/bin/env python
"""Round-5 fixture (TEST INFRASTRUCTURE; see oracle/__init__.py): the celerite log-likelihood and its gradient for WIDE
kernels -- state widths J = 10, 12 and 16, beyond the 8 the time-parallel path carries -- from the dense definition in x87
long double (oracle/make_golden_r02.gp_dense_ld, pinned to mpmath there):

  rot2_sho   two RotationTerms (celerite2.terms.RotationTerm: two SHO terms each, at a rotation period and its first
             harmonic) + one SHO term: J = 10 -- an ordinary stellar-variability model (two spotted stars of a binary, or a
             star and a contaminant, + granulation)
  rot3       three RotationTerms: J = 12
  mixed16    two RotationTerms + two SHO terms + two real terms (an over-damped SHO, Q = 0.3, as celerite2 writes it) +
             a Matern-3/2 term: J = 16, real terms and pairs side by side

N = 400 irregular cadences with a gap; the series is a draw from the process itself plus noise.  The library runs these on
the sequential kernels (a draw on a DPP row of 16 lanes): tests/test_gpu_golden.py::test_gp_wide_golden holds them to 1e-9 in
the log-likelihood and 1e-6 in every gradient; tests/test_golden_r05.py does the same for the C and numpy ports.

-> tests/golden/gp_wide.npz.  Run from the repository root:  python oracle/make_golden_r05.py
reference: celerite2 >= 0.3.1 is a dependency of the reference (setup.py:36), not in its tree -- and has no limit on the number
of terms; SURVEY.md Appendix B."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import numpy_port as P  # noqa: E402
from oracle.make_golden_r02 import gp_dense_ld  # noqa: E402
from oracle.make_golden_r03 import matern32_coefficients, rotation_coefficients  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N = 400


def cat(*parts):
    return tuple(np.concatenate([np.asarray(p[k], dtype=np.float64).reshape(-1) for p in parts]) for k in range(6))


def sho(sigma, rho, Q):
    return P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, Q), Q)


def main():
    rng = np.random.default_rng(505)
    cases = {
        "rot2_sho": (cat(rotation_coefficients(0.8, 4.0, 0.02, 0.5, 0.5), rotation_coefficients(0.5, 7.3, 1.0, 0.2, 0.3),
                         sho(0.3, 0.4, 1 / np.sqrt(2))), 0.01),
        "rot3": (cat(rotation_coefficients(0.8, 4.0, 0.02, 0.5, 0.5), rotation_coefficients(0.5, 7.3, 1.0, 0.2, 0.3),
                     rotation_coefficients(0.4, 1.9, 3.0, 1.0, 0.8)), 0.02),
        "mixed16": (cat(rotation_coefficients(0.8, 4.0, 0.02, 0.5, 0.5), rotation_coefficients(0.5, 7.3, 1.0, 0.2, 0.3),
                        sho(0.3, 0.4, 1 / np.sqrt(2)), sho(0.6, 11.0, 4.0), sho(0.4, 2.5, 0.3), matern32_coefficients(0.3, 1.3)),
                    0.02),
    }
    out = {}
    for key, (co, noise) in cases.items():
        J = co[0].size + 2 * co[2].size
        t = np.sort(rng.uniform(0, 40.0, N))
        t[N // 2:] += 13.0
        diag = noise * (1.0 + rng.uniform(size=N))
        K = P.celerite_kernel(np.abs(t[:, None] - t[None, :]), *co) + np.diag(diag)
        y = np.linalg.cholesky(K) @ rng.normal(size=N)
        ll, g = gp_dense_ld(t, y, diag, co)
        out[f"{key}_t"], out[f"{key}_y"], out[f"{key}_diag"], out[f"{key}_loglike"] = t, y, diag, ll
        for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
            out[f"{key}_{nm}"] = np.asarray(c, dtype=np.float64)
            out[f"{key}_g{nm}"] = g[nm]
        out[f"{key}_gy"], out[f"{key}_gdiag"] = g["y"], g["diag"]
        print(key, "J =", J, "loglike", ll)
    np.savez_compressed(os.path.join(OUT, "gp_wide.npz"), **out)
    print("gp_wide.npz", os.path.getsize(os.path.join(OUT, "gp_wide.npz")))


if __name__ == "__main__":
    main()
