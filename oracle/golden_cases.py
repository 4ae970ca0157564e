"""The case tables of the golden fixtures (TEST INFRASTRUCTURE): which systems oracle/make_golden*.py evaluated, shared with the
tests that compare against the fixtures.  numpy only -- the generating scripts need mpmath, the tests on the GPU box must not
(VERDICT r5 item 8)."""
import numpy as np

LIGHTCURVE_CASES = {
    # reference tests/light_curves_test.py:75-102
    "two_planet": dict(orbit=dict(m_star=1.45, r_star=1.5, t0=[0.5, 17.4], period=[10.0, 5.3], ecc=[0.1, 0.8],
                                  omega=[0.5, 1.3], m_planet=[0.3, 0.5]),
                       r=[0.1, 0.01], u=[0.2, 0.3], t=("linspace", -20, 20, 1000), texp=[None, 0.1]),
    # :148-164
    "contact_bug": dict(orbit=dict(period=3.456, ecc=0.6, omega=-1.5), r=[0.1], u=[0.3, 0.2],
                        t=("linspace", -0.1, 0.1, 1000), texp=[0.02]),
    # :167-193
    "small_star": dict(orbit=dict(r_star=0.189, m_star=0.151, period=0.4626413, t0=0.2, b=0.5, ecc=0.1, omega=0.1),
                       r=[0.04221468 * 0.189], u=[0.2, 0.1], t=("linspace", 0, 0.4626413, 500), texp=[None]),
    # BASELINE C1 / C2 at 2048 cadences around a transit
    "c1_circular": dict(orbit=dict(period=3.5, t0=1.0, b=0.3), r=[0.1], u=[0.3, 0.2],
                        t=("arange", 0.8, 2048, 2.0 / 1440.0), texp=[None]),
    "c2_e03": dict(orbit=dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1), r=[0.1], u=[0.3, 0.2],
                   t=("arange", 0.8, 2048, 2.0 / 1440.0), texp=[None]),
}


def case_time(spec):
    if spec[0] == "linspace":
        return np.linspace(spec[1], spec[2], spec[3])
    return spec[1] + np.arange(spec[2]) * spec[3]


C4 = dict(period=[3.5, 7.9, 13.1, 29.7], t0=[1.0, 2.3, 5.1, 11.7], b=[0.3, 0.1, 0.5, 0.2], ecc=[0.05, 0.1, 0.2, 0.3],
          omega=[1.1, -0.4, 2.0, 0.3], r=[0.1, 0.05, 0.07, 0.03], u=(0.3, 0.2))
C5 = dict(period=2.7, t0=0.4, b=0.2, ecc=0.1, omega=0.7, r=0.08, u_p=(0.3, 0.2), u_s=(0.4, 0.1), sbr=0.3,
          texp=29.4 / 1440.0, oversample=7, order=0)


