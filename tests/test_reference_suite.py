"""The reference's own test cases for the path that had no mirror here yet, run against this package (the reference's
assertions, its systems, its tolerances; file:line of each in the docstrings).  CPU: the orbit algebra is torch and runs
without a device.  Where the reference cross-checks against batman (not installed: `pytest.importorskip`), the check is
an independent closed-form sky separation with its own Newton solver -- tests/test_gpu_reference_suite.py."""
import math
import warnings

import numpy as np
import pytest
import torch

import exoplanet_amd as xo
from exoplanet_amd.orbits import constants as C
from exoplanet_amd.orbits.keplerian import _consistent_inputs

M_EARTH_PER_M_SUN = 332946.0487


def _np(x):
    return np.asarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float64)


def _ci(a, period, rho_star, r_star, m_star, m_planet):
    T = lambda v: None if v is None else torch.as_tensor(np.asarray(v, dtype=np.float64))  # noqa: E731
    return _consistent_inputs(T(a), T(period), T(rho_star), T(r_star), T(m_star), T(m_planet))


def test_get_consistent_inputs():
    """tests/orbits/keplerian_test.py:408-501 (the astropy-unit legs given in the package's units: M_sun, g/cm^3)"""
    period0 = np.array([12.567, 45.132])
    r_star0, m_star0 = 1.235, 0.986
    m_planet0 = np.array([1.543, 2.354]) / M_EARTH_PER_M_SUN
    a1, period1, rho_star1, r_star1, m_star1, m_planet1 = _ci(None, period0, None, r_star0, m_star0, m_planet0)
    assert np.allclose(period0, _np(period1))
    assert np.allclose(r_star0, _np(r_star1))
    assert np.allclose(m_star0, _np(m_star1))
    assert np.allclose(m_planet0 * M_EARTH_PER_M_SUN, _np(m_planet1) * M_EARTH_PER_M_SUN)
    first = (period1, rho_star1, r_star1, m_star1, m_planet1)

    def same(got):
        for x, y in zip(first, got):
            assert np.allclose(_np(x), _np(y))

    a2, *rest2 = _ci(a1, None, rho_star1, r_star0, None, m_planet1)
    same(rest2)
    a3, *rest3 = _ci(a2, None, rest2[1], None, rest2[3], rest2[4])
    same(rest3)
    a4, *rest4 = _ci(a3, rest3[0], None, rest3[2], None, rest3[4])
    same(rest4)
    a5, *rest5 = _ci(a3, None, rest3[1], rest3[2], None, rest3[4])        # (rho_star "with_unit(g / cm^3)": the package's unit)
    same(rest5)
    with pytest.raises(ValueError):
        _ci(None, None, None, rest3[2], rest3[3], None)
    with pytest.raises(ValueError):
        _ci(a3, rest3[0], None, rest3[2], rest3[3], None)
    with pytest.raises(ValueError):
        _ci(a3, None, rest3[1], rest3[2], rest3[3], None)


def test_consistent_coords():
    """tests/orbits/keplerian_test.py:377-405: masses of a visual binary from (a, P, m_planet)"""
    au_to_R_sun = 1.0 / C.au_per_R_sun
    a_ang, parallax = 0.324, 24.05
    a = a_ang * 1e3 / parallax             # au
    P = 28.8 * 365.25                      # days
    kappa = 0.45
    Mtot = 4 * math.pi ** 2 * (a * au_to_R_sun) ** 3 / (C.G_grav * P ** 2)
    # (Kepler's third law in solar units says the same to the accuracy of the constants)
    assert abs(Mtot - a ** 3 / (P / 365.25) ** 2) < 2e-4 * Mtot
    M2 = kappa * Mtot
    M1 = Mtot - M2
    orbit = xo.KeplerianOrbit(a=a * au_to_R_sun, period=P, m_planet=M2)
    assert np.allclose(M1, _np(orbit.m_star))
    assert np.allclose(M2, _np(orbit.m_planet))
    assert np.allclose(Mtot, _np(orbit.m_total))


@pytest.mark.parametrize("period", [1.0, [1.0, 2.0]])
@pytest.mark.parametrize("t", [1.0, [1.0], [1.0, 2.0], "grid"])
def test_light_delay_shapes(period, t):
    """tests/orbits/keplerian_test.py:568-608: light delay does not change the shape of a position"""
    orbit = xo.KeplerianOrbit(period=period)
    tt = np.linspace(0, 10, 50) if isinstance(t, str) else t
    x, y, z = orbit.get_planet_position(tt, light_delay=False)
    xr, yr, zr = orbit.get_planet_position(tt, light_delay=True)
    assert tuple(x.shape) == tuple(xr.shape) == tuple(yr.shape) == tuple(zr.shape)
    # (the delay of a solar-mass system's planet at 4 R_sun is a fraction of a minute: the positions barely move)
    assert np.allclose(_np(x), _np(xr), atol=1e-2) and np.allclose(_np(z), _np(zr), atol=1e-2)


def test_duration_without_ror_warning():
    """tests/orbits/keplerian_test.py:646-661"""
    kw = dict(period=10.1235, t0=0.0, b=0.34, duration=0.12, r_star=0.7)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)
        with pytest.raises(UserWarning):
            xo.KeplerianOrbit(**kw)
        xo.KeplerianOrbit(ror=0.06, **kw)


def _check_quad(u, b, depth, ror):
    """tests/light_curves_test.py:257-267"""
    mu = np.sqrt(1 - b ** 2)
    expect = np.sqrt(depth * (1 - u[0] / 3 - u[1] / 6) / (1 - u[0] * (1 - mu) - u[1] * (1 - mu) ** 2))
    assert np.shape(expect) == np.shape(ror)
    assert np.allclose(expect, ror)


def test_approx_transit_depth():
    """tests/light_curves_test.py:270-282"""
    u = np.array([0.3, 0.2])
    lc = xo.LimbDarkLightCurve(u[0], u[1])
    for b, delta in [(np.float64(0.5), np.float64(0.01)), (np.array([0.1, 0.9]), np.array([0.1, 0.5])),
                     (np.array([0.1, 0.9, 0.3]), np.array([0.1, 0.5, 0.0234]))]:
        dv = torch.tensor(delta, dtype=torch.float64, requires_grad=True)
        ror, jac = lc.get_ror_from_approx_transit_depth(dv, torch.as_tensor(b, dtype=torch.float64), jac=True)
        _check_quad(u, b, delta, _np(ror))
        (g,) = torch.autograd.grad(ror.sum(), dv)
        assert np.allclose(_np(g), _np(jac))


def test_vector_limb_darkening_is_deprecated():
    """tests/light_curves_test.py:56-72, the constructor's half (the light curves themselves: the GPU file)"""
    with pytest.warns(DeprecationWarning, match=r"vector of limb darkening"):
        lc = xo.LimbDarkLightCurve(torch.tensor([0.3, 0.2], dtype=torch.float64))
    assert float(lc.u1) == 0.3 and float(lc.u2) == 0.2
    with pytest.warns(DeprecationWarning):
        with pytest.raises(AssertionError):
            xo.LimbDarkLightCurve(torch.tensor([0.3], dtype=torch.float64))


def test_jacobians_all_five_entries():
    """tests/orbits/keplerian_test.py:664-699 (`test_jacobians`), all five entries with the reference's numbers: the
    hand-written Jacobians of the duration parameterisation (a, a_planet, a_star, rho_star) and d cos(i) / d b against
    autograd through the same constructor"""
    from oracle import numpy_port as P

    duration, period, b, ror, r_star = 0.12, 10.1235, 0.34, 0.06, 0.7
    dv = torch.tensor(duration, dtype=torch.float64, requires_grad=True)
    orbit = xo.KeplerianOrbit(period=period, t0=0.0, b=b, duration=dv, r_star=r_star, ror=ror)
    for name in ("a", "a_planet", "a_star", "rho_star"):
        (g,) = torch.autograd.grad(getattr(orbit, name).sum(), dv, retain_graph=True)
        jac = orbit.jacobians["duration"][name]
        assert np.allclose(float(jac.sum()), float(g)), name
    assert float(orbit.jacobians["duration"]["a"].sum()) != 0.0 and float(orbit.jacobians["duration"]["rho_star"].sum()) != 0.0
    # (m_planet = 0: a_star and its Jacobian vanish identically, as in the reference's case)
    bv = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    orbit2 = xo.KeplerianOrbit(period=period, t0=0.0, b=bv, a=orbit.a.detach(), r_star=r_star, ror=ror)
    (g,) = torch.autograd.grad(orbit2.cos_incl.sum(), bv)
    assert np.allclose(float(orbit2.jacobians["b"]["cos_incl"].sum()), float(g))
    # the same against the oracle's restatement of the constructor (central differences)
    h = 1e-6
    for name in ("a", "rho_star"):
        f = lambda d: float(np.sum(getattr(P.KeplerianOrbit(period=period, t0=0.0, b=b, duration=d, r_star=r_star, ror=ror), name)))  # noqa: E731
        fd = (f(duration + h) - f(duration - h)) / (2 * h)
        assert abs(float(orbit.jacobians["duration"][name].sum()) - fd) <= 1e-6 * abs(fd), name
