"""Generate tests/golden/*.npz from the arbitrary-precision DEFINITIONS
(oracle/mp_reference.py) and, for the end-to-end light curves, from the
mp-validated numpy port.  Run:  python -m oracle.make_golden

No golden vectors exist in the reference's own tests for this path and the
reference cannot be imported here, so these fixtures are
the pin.  They are DATA (inputs + expected outputs); regenerate anywhere with
numpy + mpmath.
"""
import os

import mpmath as mp
import numpy as np

from . import mp_reference as R
from . import numpy_port as P

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def golden_kepler():
    es = [0.0, 1e-8, 0.01, 0.1, 0.3, 0.5, 0.8, 0.9, 0.99, 0.999, 1 - 1e-8]
    Ms = np.concatenate([np.linspace(-4 * np.pi, 4 * np.pi, 65),
                         [1e-12, -1e-12, 1e-6, -1e-6, np.pi - 1e-9, np.pi + 1e-9, 400.123, 1e-3, 0.05]])
    M, e = [x.ravel() for x in np.meshgrid(Ms, es)]
    out = np.array([[float(v) for v in R.kepler(m, ee)] for m, ee in zip(M, e)])
    np.savez_compressed(os.path.join(OUT, "kepler.npz"), M=M, ecc=e, sinf=out[:, 0], cosf=out[:, 1],
                        dsinf_dM=out[:, 2], dcosf_dM=out[:, 3], dsinf_de=out[:, 4], dcosf_de=out[:, 5])


def golden_quad_sv():
    B, Rr = [], []
    b_ref = np.linspace(-1.5, 1.5, 100)          # reference tests/light_curves_test.py:24-25
    B += list(b_ref); Rr += [0.1] * b_ref.size
    for r in [0.01, 0.04221468, 0.1, 0.5, 0.9, 1.0, 1.5, 10.0]:
        bs = np.concatenate([np.linspace(0, r + 1.2, 15), [1e-9, 1e-5]])
        B += list(bs); Rr += [r] * bs.size
    # the five singular points +- 1e-8 (light_curves_test.py:241-254), off the exact loci
    for b0, r0 in [(0.1, 0.9), (0.5, 0.5), (0.0, 0.1), (0.0, 1.0), (1.1, 0.1)]:
        for db, dr in [(1e-8, 0), (-1e-8, 0), (0, 1e-8), (0, -1e-8)]:
            B.append(b0 + db); Rr.append(r0 + dr)
    B, Rr = np.array(B), np.array(Rr)
    s = np.empty((B.size, 3)); db = np.empty_like(s); dr = np.empty_like(s)
    for i, (b, r) in enumerate(zip(B, Rr)):
        s[i] = [float(v) for v in R.quad_sv(b, r)]
        gb, gr = R.quad_sv_grad(b, r)
        db[i] = [float(v) for v in gb]; dr[i] = [float(v) for v in gr]
    np.savez_compressed(os.path.join(OUT, "quad_sv.npz"), b=B, r=Rr, s=s, dsdb=db, dsdr=dr)


def golden_gp():
    rng = np.random.default_rng(4)
    out = {}
    for tag, Q, N in [("q03", 0.3, 10), ("q07", 1 / np.sqrt(2), 40), ("q3", 3.0, 60)]:
        t = np.sort(rng.uniform(0, 10, N)); y = rng.normal(size=N); diag = 0.1 + 0.1 * rng.uniform(size=N)
        co = P.sho_coefficients(*P.sho_from_sigma_rho(0.8, 3.0, Q), Q)
        ll = R.gp_loglike_dense([mp.mpf(v) for v in t], [mp.mpf(v) for v in y], [mp.mpf(v) for v in diag],
                                *[[mp.mpf(float(v)) for v in c] for c in co])
        out[f"{tag}_t"] = t; out[f"{tag}_y"] = y; out[f"{tag}_diag"] = diag
        for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
            out[f"{tag}_{nm}"] = c
        out[f"{tag}_loglike"] = float(ll)
    np.savez_compressed(os.path.join(OUT, "gp_sho.npz"), **out)


from .golden_cases import LIGHTCURVE_CASES, case_time  # noqa: E402,F401  (the tables: numpy only, shared with the tests)


def golden_lightcurves():
    out = {}
    for name, case in LIGHTCURVE_CASES.items():
        okw = {k: (np.array(v, dtype=float) if isinstance(v, list) else v) for k, v in case["orbit"].items()}
        orbit = P.KeplerianOrbit(**okw)
        t = case_time(case["t"])
        for texp in case["texp"]:
            f = P.LimbDarkLightCurve(*case["u"]).get_light_curve(orbit=orbit, r=np.array(case["r"]), t=t, texp=texp,
                                                                 use_in_transit=False)
            out[f"{name}_texp{texp}"] = f
    # secondary eclipse, reference tests/light_curves_test.py:285-311
    t = np.linspace(-6.435, 10.4934, 5000)
    out["secondary"] = P.SecondaryEclipseLightCurve([0.3, 0.2], [0.4, 0.1], 0.3).get_light_curve(
        orbit=P.KeplerianOrbit(period=1.543, t0=-0.123), r=0.08, t=t, use_in_transit=False)
    np.savez_compressed(os.path.join(OUT, "lightcurves.npz"), **out)


def golden_quad_sv_grid():
    """nine points of the reference's grid b = linspace(-1.5, 1.5, 100), r = 0.1 (tests/light_curves_test.py:24-27) for
    tests/test_gpu_reference_suite.py::test_light_curve_against_definition: the GPU box needs no mpmath"""
    b = np.linspace(-1.5, 1.5, 100)
    ks = [0, 17, 33, 45, 49, 50, 60, 83, 99]
    s = np.array([[float(x) for x in R.quad_sv(abs(b[k]), 0.1)] for k in ks])
    np.savez_compressed(os.path.join(OUT, "quad_sv_grid.npz"), k=np.array(ks), b=b[ks], r=np.array(0.1), s=s)


def main():
    os.makedirs(OUT, exist_ok=True)
    golden_kepler()
    golden_quad_sv()
    golden_quad_sv_grid()
    golden_gp()
    golden_lightcurves()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
