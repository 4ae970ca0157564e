"""GPU: the reference's test cases that need the kernels and had no mirror here yet -- test_sky_coords and test_small_star
(tests/orbits/keplerian_test.py:17-58, 316-349), test_light_curve / test_vector_params (tests/light_curves_test.py:22-39,
56-72).  The reference checks the first two against batman's `_rsky` (absent here, as it may be there:
`pytest.importorskip`); the stand-in is the same definition -- sky-projected separation from the true anomaly, 100 when
the body is behind the star -- with its own Newton solver in numpy, written for these tests only: nothing of the
package, nothing of oracle/, so the comparison is between independent derivations, with the reference's tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda:0")


def rsky(t, t0, period, a, inc, ecc, omega):
    """separation of the centres on the sky in stellar radii (batman's convention: omega of the planet's orbit, transit
    where the true anomaly is pi/2 - omega); 100 where the planet is not in front of the star"""
    f_tr = 0.5 * np.pi - omega
    E_tr = 2.0 * np.arctan(np.sqrt((1 - ecc) / (1 + ecc)) * np.tan(0.5 * f_tr))
    tp = t0 - period * (E_tr - ecc * np.sin(E_tr)) / (2 * np.pi)
    M = 2 * np.pi * (t - tp) / period
    M = np.mod(M + np.pi, 2 * np.pi) - np.pi
    E = M + ecc * np.sin(M)
    for _ in range(60):
        E = E - (E - ecc * np.sin(E) - M) / (1 - ecc * np.cos(E))
    f = 2.0 * np.arctan2(np.sqrt(1 + ecc) * np.sin(0.5 * E), np.sqrt(1 - ecc) * np.cos(0.5 * E))
    d = a * (1 - ecc ** 2) / (1 + ecc * np.cos(f)) * np.sqrt(1 - np.sin(omega + f) ** 2 * np.sin(inc) ** 2)
    return np.where(np.sin(omega + f) * np.sin(inc) <= 0, 100.0, d)


def test_sky_coords(dev):
    """tests/orbits/keplerian_test.py:17-58: 720 orbits x 1000 times"""
    import exoplanet_amd as xo

    t = np.linspace(-100, 100, 1000)
    t0, period, a, e, omega, incl = (x.flatten() for x in np.meshgrid(
        np.linspace(-5.0, 5.0, 2), np.exp(np.linspace(np.log(5.0), np.log(50.0), 3)), np.linspace(50.0, 100.0, 2),
        np.linspace(0.0, 0.9, 5), np.linspace(-np.pi, np.pi, 3), np.arccos(np.linspace(0, 1, 5)[:-1])))
    r_ref = np.stack([rsky(t, t0[i], period[i], a[i], incl[i], e[i], omega[i]) for i in range(len(t0))], axis=1)
    m = r_ref < 100.0
    assert m.sum() > 0
    T = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)  # noqa: E731
    orbit = xo.KeplerianOrbit(period=T(period), a=T(a), t0=T(t0), ecc=T(e), omega=T(omega), incl=T(incl))
    x, y, z = (c.cpu().numpy() for c in orbit.get_relative_position(T(t)))
    r = np.sqrt(x ** 2 + y ** 2)
    assert np.allclose(r_ref[m], r[m], atol=2e-5)     # the in-transit impact parameter
    assert np.all(z[m] > 0)                           # in transit <=> positive z in this parameterisation
    assert np.all((z[~m] < 0) | (r[~m] > 2))          # no transit there: not transiting here


def test_small_star(dev):
    """tests/orbits/keplerian_test.py:316-349: an M dwarf, a / R* ~ 9"""
    import exoplanet_amd as xo

    period, t0, ecc, omega = 0.4626413, 0.2, 0.1, 0.1
    t = np.linspace(0, period, 500)
    T = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)  # noqa: E731
    orbit = xo.KeplerianOrbit(r_star=T(0.189), m_star=T(0.151), period=T(period), t0=T(t0), b=T(0.5), ecc=T(ecc),
                              omega=T(omega))
    # (batman measures in stellar radii, the orbit in solar radii)
    r_ref = rsky(t, t0, period, float(orbit.a) / 0.189, float(orbit.incl), ecc, omega)
    m = r_ref < 100.0
    assert m.sum() > 0
    x, y, z = (c.cpu().numpy().reshape(-1) for c in orbit.get_relative_position(T(t)))
    assert np.allclose(r_ref[m], np.sqrt(x ** 2 + y ** 2)[m] / 0.189, atol=2e-5)


def test_light_curve_against_definition(dev):
    """tests/light_curves_test.py:22-39 checks `_compute_light_curve(b, r)` for b in [-1.5, 1.5] against starry; here the
    same grid against the mpmath definition of the solution vector (oracle/mp_reference.py) dotted with get_cl, and the
    even symmetry in b the reference's grid implies"""
    import os

    import exoplanet_amd as xo
    from oracle import numpy_port as P

    u1, u2 = 0.2, 0.3
    lc = xo.LimbDarkLightCurve(u1, u2)
    b = np.linspace(-1.5, 1.5, 100)
    r = 0.1 + np.zeros_like(b)
    T = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)  # noqa: E731
    got = lc._compute_light_curve(T(b), T(r)).cpu().numpy()
    assert np.allclose(got, got[::-1], atol=1e-15)
    c = P.get_cl(u1, u2)
    # the mpmath values of nine points of the grid: a fixture (tests/golden/quad_sv_grid.npz, made in the build container by
    # oracle.mp_reference.quad_sv at 34 digits: the GPU box needs no mpmath -- VERDICT r5 item 8)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quad_sv_grid.npz"))
    for k, bk, s in zip(g["k"], g["b"], g["s"]):
        assert bk == b[k] and float(g["r"]) == 0.1
        assert abs(got[k] - (float(s @ c) - 1.0)) <= 1e-13, (b[k], got[k])
    assert got.min() < -0.009 and got.max() <= 1e-15


def test_vector_params(dev):
    """tests/light_curves_test.py:56-72"""
    import exoplanet_amd as xo

    u = torch.tensor([0.3, 0.2], dtype=torch.float64, device=dev)
    b = torch.linspace(-1.5, 1.5, 20, dtype=torch.float64, device=dev)
    r = 0.1 + torch.zeros_like(b)
    with pytest.warns(DeprecationWarning, match=r"vector of limb darkening"):
        lc1 = xo.LimbDarkLightCurve(u)._compute_light_curve(b, r)
    lc2 = xo.LimbDarkLightCurve(u[0], u[1])._compute_light_curve(b, r)
    assert torch.allclose(lc1, lc2)
    assert (lc2 < 0).any()
