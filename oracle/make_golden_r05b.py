#!/usr/bin/env python
"""Round-5 fixture, second part (TEST INFRASTRUCTURE; see oracle/__init__.py): the celerite log-likelihood and its gradient where
the time-parallel path hands a draw to the SEQUENTIAL kernels -- conditioning scores beyond the 1e8 its robust route carries, and no
measurement noise at all -- from the dense definition in x87 long double (oracle/make_golden_r02.gp_dense_ld):

  kappa1e9   SHO term (sigma = 1, rho = 3, Q = 2) under error bars of 3e-5 .. 4.5e-5: signal variance / smallest diag = 1.1e9
  kappa1e10  the same under error bars of 1e-5
  diag0      the same kernel, diag = 0 exactly (an interpolating GP): K itself is factorised
  diag0_j4   two SHO terms (J = 4), diag = 0

N = 240 cadences on a jittered grid (spacings 0.08 .. 0.25 d: K without noise stays well inside long double's reach -- condition number
~1e8), a gap in the middle; the series is a draw from the process itself.  tests/test_golden_r05.py (C / numpy ports) and
tests/test_gpu_golden.py::test_gp_edge_golden (the library: default plan and n_chunks = 1) hold them to the tolerances stated there.

-> tests/golden/gp_edge.npz.  Run from the repository root:  python oracle/make_golden_r05b.py
reference: celerite2 >= 0.3.1 (setup.py:36) places no lower bound on diag; SURVEY.md Appendix B."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import numpy_port as P  # noqa: E402
from oracle.make_golden_r02 import gp_dense_ld  # noqa: E402
from oracle.make_golden_r05 import cat, sho  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N = 240


def main():
    rng = np.random.default_rng(606)
    one = sho(1.0, 3.0, 2.0)
    two = cat(sho(1.0, 3.0, 2.0), sho(0.5, 0.9, 0.8))
    cases = {"kappa1e9": (one, 3e-5), "kappa1e10": (one, 1e-5), "diag0": (one, 0.0), "diag0_j4": (two, 0.0)}
    out = {}
    for key, (co, sig) in cases.items():
        co = tuple(np.asarray(c, dtype=np.float64).reshape(-1) for c in co)
        t = np.cumsum(rng.uniform(0.08, 0.25, N))
        t[N // 2:] += 7.0
        diag = (sig * (1.0 + 0.5 * rng.uniform(size=N))) ** 2
        K = P.celerite_kernel(np.abs(t[:, None] - t[None, :]), *co)
        y = np.linalg.cholesky(K + np.diag(diag + 1e-12)) @ rng.normal(size=N)
        ll, g = gp_dense_ld(t, y, diag, co)
        kappa = (co[0].sum() + co[2].sum()) / diag.min() if diag.min() > 0 else np.inf
        out[f"{key}_t"], out[f"{key}_y"], out[f"{key}_diag"], out[f"{key}_loglike"] = t, y, diag, ll
        for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
            out[f"{key}_{nm}"] = c
            out[f"{key}_g{nm}"] = g[nm]
        out[f"{key}_gy"], out[f"{key}_gdiag"] = g["y"], g["diag"]
        print(key, "J =", co[0].size + 2 * co[2].size, "kappa %.3g" % kappa, "cond(K + diag) %.3g" % np.linalg.cond(K + np.diag(diag)), "loglike", ll)
    np.savez_compressed(os.path.join(OUT, "gp_edge.npz"), **out)
    print("gp_edge.npz", os.path.getsize(os.path.join(OUT, "gp_edge.npz")))


if __name__ == "__main__":
    main()
