"""GPU: the "next" rows of SURVEY 8f that ride on the same Kepler op -- velocities /
RV / accelerations, centre of mass, _flip, TTVOrbit, SimpleTransitOrbit -- with the
reference's own self-consistency tests restated
(/root/reference/tests/orbits/keplerian_test.py, ttv_test.py, simple_test.py)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def npy(x):
    return x.detach().cpu().numpy()


ORBIT = dict(m_star=1.3, r_star=1.0, t0=0.5, period=100.0, ecc=0.1, omega=0.5, Omega=1.0, incl=0.25 * np.pi,
             m_planet=0.1)


def _ddt(fn, t):
    """d/dt of each component of fn(t) (tuple of tensors shaped like t) by autograd"""
    out = []
    for i in range(3):
        tt = t.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(fn(tt)[i].sum(), tt)
        out.append(npy(g))
    return np.array(out)


def test_velocity_is_time_derivative_of_position(dev):
    """keplerian_test.py:91-131 (pins d(sinf, cosf)/dM of the Kepler op)"""
    import exoplanet_amd as xo

    t = torch.linspace(0, 100, 1000, dtype=torch.float64, device=dev)
    orbit = xo.KeplerianOrbit(**ORBIT)
    for pos, vel in (("get_star_position", "get_star_velocity"), ("get_planet_position", "get_planet_velocity"),
                     ("get_relative_position", "get_relative_velocity")):
        v = np.array([npy(x) for x in getattr(orbit, vel)(t)])
        assert np.allclose(v, _ddt(getattr(orbit, pos), t))


def test_acceleration_is_time_derivative_of_velocity(dev):
    """keplerian_test.py:157-196"""
    import exoplanet_amd as xo

    t = torch.linspace(0, 100, 1000, dtype=torch.float64, device=dev)
    orbit = xo.KeplerianOrbit(**ORBIT)
    for vel, acc in (("get_star_velocity", "get_star_acceleration"), ("get_planet_velocity", "get_planet_acceleration"),
                     ("get_relative_velocity", "get_relative_acceleration")):
        a = np.array([npy(x) for x in getattr(orbit, acc)(t)])
        assert np.allclose(a, _ddt(getattr(orbit, vel), t))


def test_radial_velocity(dev):
    """keplerian_test.py:134-154: RV = -conv * d z_star/dt, and the K-parameterised form"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits.constants import m_per_s_per_Rsun_per_day

    t = torch.linspace(0, 100, 1000, dtype=torch.float64, device=dev)
    orbit = xo.KeplerianOrbit(**ORBIT)
    rv = npy(orbit.get_radial_velocity(t))
    dz = _ddt(orbit.get_star_position, t)[2]
    assert np.allclose(rv, -m_per_s_per_Rsun_per_day * dz)
    # K given: K (cos(w + f) + e cos w)
    K = 12.5
    got = npy(orbit.get_radial_velocity(t, K=K))
    po = P.KeplerianOrbit(**{k: v for k, v in ORBIT.items()})
    sinf, cosf = po._get_true_anomaly(npy(t))
    want = K * (np.cos(0.5) * cosf[:, 0] - np.sin(0.5) * sinf[:, 0] + 0.1 * np.cos(0.5))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_center_of_mass(dev):
    """keplerian_test.py:61-88"""
    import exoplanet_amd as xo

    t = torch.linspace(0, 100, 1000, dtype=torch.float64, device=dev)
    m_planet = np.array([0.5, 0.1]); m_star = 1.45
    orbit = xo.KeplerianOrbit(m_star=m_star, r_star=1.0, t0=np.array([0.5, 17.4]), period=np.array([100.0, 37.3]),
                              ecc=np.array([0.1, 0.8]), omega=np.array([0.5, 1.3]), Omega=np.array([0.0, 1.0]),
                              incl=np.array([0.25 * np.pi, 0.3 * np.pi]), m_planet=m_planet)
    pc = np.array([npy(x) for x in orbit.get_planet_position(t)])
    sc = np.array([npy(x) for x in orbit.get_star_position(t)])
    com = np.sum((m_planet[None, None, :] * pc + m_star * sc) / (m_star + m_planet)[None, None, :], axis=0)
    assert np.allclose(com, 0.0)


def test_flipped_orbit(dev):
    """keplerian_test.py:199-254: the flipped orbit swaps star and planet (atol 1e-5)"""
    import exoplanet_amd as xo

    t = torch.linspace(0, 100, 1000, dtype=torch.float64, device=dev)
    for kw in (dict(m_star=1.3, m_planet=0.1, period=100.0, t0=0.5, incl=0.25 * np.pi),
               dict(m_star=1.3, m_planet=0.1, period=100.0, t0=0.5, incl=0.25 * np.pi, ecc=0.3, omega=0.5)):
        orbit1 = xo.KeplerianOrbit(r_star=1.1, **kw)
        orbit2 = orbit1._flip(0.7)
        x1, y1, z1 = [npy(v) for v in orbit1.get_star_position(t)]
        x2, y2, z2 = [npy(v) for v in orbit2.get_planet_position(t)]
        assert np.allclose(x1, x2, atol=1e-5) and np.allclose(y1, y2, atol=1e-5) and np.allclose(z1, z2, atol=1e-5)
        x1, y1, z1 = [npy(v) for v in orbit1.get_planet_position(t)]
        x2, y2, z2 = [npy(v) for v in orbit2.get_star_position(t)]
        assert np.allclose(x1, x2, atol=1e-5) and np.allclose(y1, y2, atol=1e-5) and np.allclose(z1, z2, atol=1e-5)


def test_ttv_orbit_without_ttvs_is_keplerian(dev):
    """ttv_test.py:49-83, all seven methods"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit, compute_expected_transit_times

    periods, t0s = np.array([10.5, 56.34]), np.array([45.3, 48.1])
    time = torch.linspace(456.023, 595.23, 5000, dtype=torch.float64, device=dev)
    expected = compute_expected_transit_times(456.023, 595.23, periods, t0s)
    orbit0 = xo.KeplerianOrbit(period=periods, t0=t0s)
    orbit1 = TTVOrbit(period=periods, t0=np.array([t[0] for t in expected]), ttvs=[np.zeros_like(t) for t in expected])
    orbit2 = TTVOrbit(transit_times=expected)
    for arg in ("get_relative_position", "get_star_position", "get_planet_position", "get_relative_velocity",
                "get_star_velocity", "get_planet_velocity", "get_radial_velocity"):
        expect = getattr(orbit0, arg)(time)
        for orb in (orbit1, orbit2):
            calc = getattr(orb, arg)(time)
            if isinstance(expect, tuple):
                for a, b in zip(expect, calc):
                    assert np.allclose(npy(a), npy(b)), arg
            else:
                assert np.allclose(npy(expect), npy(calc)), arg


def test_ttv_light_curve_follows_the_transit_times(dev):
    """a TTV light curve is the Keplerian one re-centred on each labelled transit"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit, compute_expected_transit_times

    rng = np.random.default_rng(3)
    period, t0 = np.array([7.3]), np.array([2.0])
    expected = compute_expected_transit_times(0.0, 60.0, period, t0)
    ttv = [0.05 * rng.normal(size=expected[0].size)]
    orbit = TTVOrbit(period=period, t0=np.array([expected[0][0]]), ttvs=ttv, b=0.2)
    t = np.linspace(0, 60, 20000)
    lc = npy(xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=0.08, t=t))[:, 0]
    ref = xo.LimbDarkLightCurve(0.3, 0.2)
    kep = xo.KeplerianOrbit(period=period, t0=np.array([expected[0][0]]), b=0.2)
    assert lc.min() < -5e-3
    for k, tk in enumerate(expected[0] + ttv[0]):
        sel = np.abs(t - tk) < 0.4
        if sel.sum() < 10:
            continue
        # same shape as the unperturbed transit k shifted by the TTV
        want = npy(ref.get_light_curve(orbit=kep, r=0.08, t=t[sel] - ttv[0][k]))[:, 0]
        np.testing.assert_allclose(lc[sel], want, rtol=0, atol=1e-12)
    # in_transit goes through the warp too: windows == full evaluation
    full = npy(xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=0.08, t=t, use_in_transit=False))[:, 0]
    np.testing.assert_allclose(lc, full, rtol=0, atol=1e-15)


def test_simple_transit_orbit(dev):
    """simple_test.py:51-81: a long-period Keplerian transit ~ the straight-line one (rtol 1e-3);
    plus in_transit == |flux| > 0"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import SimpleTransitOrbit

    period, t0, r = 1000.0, 2.3, 0.04
    t = np.linspace(t0 - 1, t0 + 1, 2000)
    kep = xo.KeplerianOrbit(period=period, t0=t0, b=0.3, r_star=1.1, m_star=0.9)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    y1 = npy(lc.get_light_curve(orbit=kep, r=r, t=t))[:, 0]
    # duration between first and fourth contact from the oracle's window
    po = P.KeplerianOrbit(period=period, t0=t0, b=0.3, r_star=1.1, m_star=0.9)
    idx = po.in_transit(t, r=np.array([r]))
    duration = (t[idx[-1]] - t[idx[0]]) + (t[1] - t[0])
    simple = SimpleTransitOrbit(period=period, t0=t0, b=0.3, duration=duration, r_star=1.1, ror=r / 1.1)
    y2 = npy(lc.get_light_curve(orbit=simple, r=r, t=t))[:, 0]
    assert y1.min() < -1e-3
    assert np.abs(y1 - y2).max() < 2e-3 * np.abs(y1).max() + 2e-5
    inds = npy(simple.in_transit(torch.tensor(t, device=dev), r=r))
    assert set(np.nonzero(y2 != 0)[0]).issubset(set(inds))
