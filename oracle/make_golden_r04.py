#!/usr/bin/env python
"""Round-4 fixture (TEST INFRASTRUCTURE; see oracle/__init__.py): the celerite log-likelihood and its gradient for RANDOM
kernels of the kind tools/gp_cond_bins.py scans -- J = 2 .. 6 state indices, decay and oscillation rates from 1e-3 to 30 per
sample within one kernel, a gap, irregular sampling -- at conditioning scores kappa = (1 + max (b/a)^2) sum(a) / min(diag)
chosen across the range the time-parallel path keeps (1e3 .. 3e6 for J = 2; 1e3, 2e4 for wider states), just above it (7e4) and
across the range of its robust route (1e6 .. 5e7), from the dense definition in
x87 long double (oracle/make_golden_r02.gp_dense_ld, pinned to mpmath there).  N = 400 cadences each.

What the fixture is for (docs/DESIGN_r1_r4.md section 3.5): the gradient with respect to the oscillation rate d of a
complex term used to carry the whole conditioning tail (exo_celerite_core.hpp, phase_flux) -- and it depended on the ORIGIN of
the time axis.  The dense definition sees time differences only, so the same values hold for t + 2457000 (BJD-style stamps): the
tests evaluate both (the stamps sit on a 2^-20 d grid, so that the shift is exact).

-> tests/golden/gp_tail.npz.  Run from the repository root:  python oracle/make_golden_r04.py
reference: celerite2 is a dependency of the reference (setup.py:36), not in its tree; SURVEY.md Appendix B."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden_r02 import gp_dense_ld  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N = 400


def draw_case(rng, n_real, n_cplx, kappa):
    span = 10 ** rng.uniform(0.5, 2.5)
    t = np.sort(rng.uniform(0, span, N))
    if rng.uniform() < 0.5:
        t[N // 2:] += span * rng.uniform(0.5, 5)
    # time stamps on a grid of 2^-20 d (0.08 s): adding an origin of millions of days is then EXACT in double precision, so
    # the shifted series is the same series
    t = np.round(t * 2.0 ** 20) / 2.0 ** 20
    assert np.all(np.diff(t) > 0)
    dtm = span / N
    cr = np.array([[10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 1.5) / dtm] for _ in range(n_real)]).reshape(n_real, 2)
    cc = np.zeros((n_cplx, 4))
    for j in range(n_cplx):
        a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1.5) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
        b = rng.uniform(-1, 1) * a * c / dd
        if rng.uniform() < 0.5:
            b = np.sign(b) * min(abs(b), a * 10 ** rng.uniform(-1, 1.5))
        cc[j] = [a, b, c, dd]
    amp2 = cr[:, 0].sum() + cc[:, 0].sum()
    ba2 = ((cc[:, 1] / cc[:, 0]) ** 2).max()
    dmin = (1 + ba2) * amp2 / kappa
    diag = dmin * (1 + 0.3 * rng.uniform(size=N))
    diag[int(rng.integers(N))] = dmin
    y = np.sqrt(amp2) * rng.normal(size=N)
    return t, y, diag, cr, cc


def main():
    rng = np.random.default_rng(404)
    out, names = {}, []
    # (wide states: 2e4 sits under the library's threshold of 3e4 -- those draws stay on the time-parallel path --, 7e4 above it: the
    # device redoes them with the sequential kernels, the host-compiled lane pipeline is run on them all the same)
    plan = [(0, 1, k) for k in (1e3, 1e5, 3e6, 3e6)] + [(1, 1, k) for k in (1e3, 2e4)] + [(0, 2, k) for k in (1e3, 2e4, 7e4)] \
        + [(1, 2, k) for k in (1e3, 2e4)] + [(2, 2, k) for k in (1e3, 2e4, 7e4)] + [(0, 3, k) for k in (1e3, 2e4, 7e4)]
    # (round 4, later: the ROBUST route of the time-parallel path takes draws up to a score of 1e8 -- exo_celerite_core.hpp,
    # chunk_adj_lane; appended, so that the cases above keep their random numbers)
    plan += [(0, 1, 5e7)] + [(0, 2, k) for k in (1e6, 1e7, 5e7)] + [(2, 2, k) for k in (1e6, 5e7)] + [(0, 3, k) for k in (1e6, 1e7, 5e7)]
    for i, (n_real, n_cplx, kappa) in enumerate(plan):
        t, y, diag, cr, cc = draw_case(rng, n_real, n_cplx, kappa)
        co = (cr[:, 0], cr[:, 1], cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3])
        ll, g = gp_dense_ld(t, y, diag, co)
        key = f"c{i:02d}"
        names.append(key)
        out[f"{key}_t"], out[f"{key}_y"], out[f"{key}_diag"], out[f"{key}_loglike"] = t, y, diag, ll
        out[f"{key}_kappa"] = kappa
        for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
            out[f"{key}_{nm}"] = np.asarray(c, dtype=np.float64)
            out[f"{key}_g{nm}"] = g[nm]
        out[f"{key}_gy"], out[f"{key}_gdiag"] = g["y"], g["diag"]
        print(key, "J", n_real + 2 * n_cplx, "kappa %.0e" % kappa, ll)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "gp_tail.npz"), **out)
    print("gp_tail.npz", os.path.getsize(os.path.join(OUT, "gp_tail.npz")))


if __name__ == "__main__":
    main()
