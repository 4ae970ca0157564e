"""GPU parity of the host-side mirror classes (KeplerianOrbit, LimbDarkLightCurve,
SecondaryEclipseLightCurve) against the oracle's restatement of the reference
glue, on the systems the reference's own tests use
(/root/reference/tests/light_curves_test.py, tests/orbits/keplerian_test.py)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def npy(x):
    return x.detach().cpu().numpy()


TWO_PLANET = dict(m_star=1.45, r_star=1.5, t0=np.array([0.5, 17.4]), period=np.array([10.0, 5.3]),
                  ecc=np.array([0.1, 0.8]), omega=np.array([0.5, 1.3]), m_planet=np.array([0.3, 0.5]))


@pytest.mark.parametrize("texp", [None, 0.1])
@pytest.mark.parametrize("use_in_transit", [None, False])
def test_in_transit_matches_reference_glue(dev, texp, use_in_transit):
    """light_curves_test.py:75-102 system: fused kernel == oracle glue, both window settings."""
    import exoplanet_amd as xo

    t = np.linspace(-20, 20, 1000)
    r = np.array([0.1, 0.01])
    want = P.LimbDarkLightCurve(0.2, 0.3).get_light_curve(orbit=P.KeplerianOrbit(**TWO_PLANET), r=r, t=t, texp=texp,
                                                         use_in_transit=False)
    got = xo.LimbDarkLightCurve(0.2, 0.3).get_light_curve(orbit=xo.KeplerianOrbit(**TWO_PLANET), r=r, t=t, texp=texp,
                                                         use_in_transit=use_in_transit)
    assert got.shape == (1000, 2)
    assert want.min() < -1e-3
    np.testing.assert_allclose(npy(got), want, rtol=0, atol=1e-13)


def test_variable_texp(dev):
    """scalar texp == per-cadence texp (light_curves_test.py:105-145)"""
    import exoplanet_amd as xo

    t = np.linspace(-20, 20, 1000)
    r = np.array([0.1, 0.01])
    orbit = xo.KeplerianOrbit(**TWO_PLANET)
    lc = xo.LimbDarkLightCurve(0.2, 0.3)
    a = lc.get_light_curve(r=r, orbit=orbit, t=t, texp=0.1)
    b = lc.get_light_curve(r=r, orbit=orbit, t=t, texp=0.1 + np.zeros_like(t), use_in_transit=False)
    assert torch.allclose(a, b, rtol=0, atol=1e-15)


def test_contact_bug(dev):
    """light_curves_test.py:148-164"""
    import exoplanet_amd as xo

    orbit = xo.KeplerianOrbit(period=3.456, ecc=0.6, omega=-1.5)
    t = np.linspace(-0.1, 0.1, 1000)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    y1 = lc.get_light_curve(orbit=orbit, r=0.1, t=t, texp=0.02)
    y2 = lc.get_light_curve(orbit=orbit, r=0.1, t=t, texp=0.02, use_in_transit=False)
    assert torch.allclose(y1, y2, rtol=0, atol=1e-15)
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(period=3.456, ecc=0.6, omega=-1.5),
                                                         r=0.1, t=t, texp=0.02, use_in_transit=False)
    np.testing.assert_allclose(npy(y1), want, rtol=0, atol=1e-13)


def test_small_star(dev):
    """M-dwarf system of light_curves_test.py:167-193 (batman cross-check replaced by the oracle)"""
    import exoplanet_amd as xo

    kw = dict(r_star=0.189, m_star=0.151, period=0.4626413, t0=0.2, b=0.5, ecc=0.1, omega=0.1)
    t = np.linspace(0, 0.4626413, 500)
    r_pl = 0.04221468 * 0.189
    got = xo.LimbDarkLightCurve(0.2, 0.1).get_light_curve(r=r_pl, orbit=xo.KeplerianOrbit(**kw), t=t)
    want = P.LimbDarkLightCurve(0.2, 0.1).get_light_curve(r=r_pl, orbit=P.KeplerianOrbit(**kw), t=t,
                                                         use_in_transit=False)
    np.testing.assert_allclose(npy(got), want, rtol=0, atol=1e-13)


def test_compute_light_curve_and_singular_points(dev):
    """continuity at the five singular points (light_curves_test.py:220-254)"""
    import exoplanet_amd as xo

    lc = xo.LimbDarkLightCurve(0.2, 0.3)

    def compare(b, r, be, re):
        f = npy(lc._compute_light_curve(torch.tensor([b - be, b + be, b], dtype=torch.float64, device=dev),
                                        torch.tensor([r - re, r + re, r], dtype=torch.float64, device=dev)))
        assert np.allclose(np.mean(f[:2]), f[2])

    compare(0.1, 0.9, 1e-8, 0.0)
    compare(0.5, 0.5, 1e-8, 0.0)
    compare(0.0, 0.1, 1e-8, 0.0)
    compare(0.0, 1.0, 0.0, 1e-8)
    compare(1.1, 0.1, 1e-8, 0.0)
    b = np.linspace(-1.5, 1.5, 100)
    got = npy(lc._compute_light_curve(torch.tensor(b, device=dev), torch.tensor(0.1 + 0 * b, device=dev)))
    want = P.LimbDarkLightCurve(0.2, 0.3)._compute_light_curve(b, 0.1 + 0 * b)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)


def test_light_curve_grad(dev):
    """verify_grad(lc, [u, b, r]) of light_curves_test.py:42-53 as torch gradcheck"""
    import exoplanet_amd as xo

    u = torch.tensor([0.2, 0.3], dtype=torch.float64, device=dev, requires_grad=True)
    b = torch.tensor(np.linspace(-1.5, 1.5, 20), device=dev, requires_grad=True)
    r = torch.tensor(0.1 + np.zeros(20), device=dev, requires_grad=True)
    fn = lambda u, b, r: xo.LimbDarkLightCurve(u[0], u[1])._compute_light_curve(b, r)  # noqa: E731
    assert torch.autograd.gradcheck(fn, (u, b, r), eps=1e-7, atol=1e-6, rtol=1e-5)


def test_secondary_eclipse(dev):
    """light_curves_test.py:285-311: fused secondary == manual two-orbit blend"""
    import exoplanet_amd as xo

    u1, u2, s, ror = np.array([0.3, 0.2]), np.array([0.4, 0.1]), 0.3, 0.08
    f = ror ** 2 * s
    t = np.linspace(-6.435, 10.4934, 5000)
    orbit1 = xo.KeplerianOrbit(period=1.543, t0=-0.123)
    orbit2 = xo.KeplerianOrbit(period=orbit1.period, t0=orbit1.t0 + 0.5 * orbit1.period, r_star=ror, m_star=1.0)
    y1 = xo.LimbDarkLightCurve(u1[0], u1[1]).get_light_curve(orbit=orbit1, r=ror, t=t)
    y2 = xo.LimbDarkLightCurve(u2[0], u2[1]).get_light_curve(orbit=orbit2, r=1.0, t=t)
    for uit in (None, False):
        y = xo.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=orbit1, r=ror, t=t, use_in_transit=uit)
        assert y.shape == (5000, 1)
        assert np.allclose(npy((y1 + f * y2) / (1 + f)), npy(y), atol=5e-6)
        want = P.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=P.KeplerianOrbit(period=1.543, t0=-0.123),
                                                                      r=ror, t=t, use_in_transit=False)
        np.testing.assert_allclose(npy(y), want, rtol=0, atol=1e-13)
    # eccentric, windows on: occultation is not half a period after transit
    kw = dict(period=2.7, t0=0.4, ecc=0.3, omega=0.7, b=0.2)
    y = xo.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=xo.KeplerianOrbit(**kw), r=ror, t=t, texp=0.02)
    want = P.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=ror, t=t, texp=0.02,
                                                                  use_in_transit=False)
    np.testing.assert_allclose(npy(y), want, rtol=0, atol=1e-13)


def test_relative_position_and_impact_parameter(dev):
    """keplerian_test.py:352-374: b(t0) equals the input impact parameter at e=0.8, z>0"""
    import exoplanet_amd as xo

    kw = dict(period=10.0, t0=0.3, b=0.4, ecc=0.8, omega=0.7, r_star=1.3, m_star=1.1)
    o = xo.KeplerianOrbit(**kw)
    x, y, z = o.get_relative_position(torch.tensor([0.3], dtype=torch.float64, device=dev))
    assert np.allclose(npy(torch.sqrt(x ** 2 + y ** 2)) / 1.3, 0.4)
    assert npy(z) > 0
    t = np.linspace(-5, 25, 400)
    got = o.get_relative_position(torch.tensor(t, device=dev))
    want = P.KeplerianOrbit(**kw).get_relative_position(t)
    for a, b in zip(got, want):
        np.testing.assert_allclose(npy(a), b, rtol=0, atol=1e-11)


def test_light_delay_and_generic_orbit_path(dev):
    """light_delay routes through the composed path (ops.kepler twice, keplerian.py:411-470)"""
    import exoplanet_amd as xo

    kw = dict(period=3.2, t0=0.5, b=0.3, ecc=0.4, omega=-0.6, m_star=1.1, r_star=0.9)
    t = np.linspace(-1, 7, 800)
    got = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(**kw), r=0.08, t=t, light_delay=True)
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=0.08, t=t, light_delay=True)
    assert want.min() < -1e-3
    np.testing.assert_allclose(npy(got), want, rtol=0, atol=1e-12)


def test_in_transit_indices(dev):
    """keplerian_test.py:257-285: indices <=> geometric condition"""
    import exoplanet_amd as xo

    t = np.linspace(-20, 20, 1000)
    r = np.array([0.1, 0.01])
    got = npy(xo.KeplerianOrbit(**TWO_PLANET).in_transit(torch.tensor(t, device=dev), r=r))
    want = P.KeplerianOrbit(**TWO_PLANET).in_transit(t, r=r)
    assert np.array_equal(got, want)


def test_batched_draws_and_user_level_gradients(dev):
    """(D,P) parameters: flux (D,N,P); d/d(period, t0, b, ecc, omega, r, u) vs finite
    differences of the oracle glue."""
    import exoplanet_amd as xo

    rng = np.random.default_rng(7)
    t = np.arange(3000) * (2.0 / 1440.0)
    base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1, u1=0.3, u2=0.2)
    D = 3
    vals = {k: v * (1 + 1e-2 * rng.normal(size=(D, 1))) for k, v in base.items()}
    leaves = {k: torch.tensor(v if k not in ("u1", "u2") else v[:, 0], device=dev, requires_grad=True)
              for k, v in vals.items()}
    g = rng.normal(size=(D, t.size, 1))

    orbit = xo.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"],
                              omega=leaves["omega"])
    flux = xo.LimbDarkLightCurve(leaves["u1"], leaves["u2"]).get_light_curve(orbit=orbit, r=leaves["r"], t=t)
    assert flux.shape == (D, t.size, 1)
    (flux * torch.tensor(g, device=dev)).sum().backward()

    def oracle(d, **over):
        v = {k: float(vals[k][d, 0]) for k in base}
        v.update(over)
        o = P.KeplerianOrbit(period=v["period"], t0=v["t0"], b=v["b"], ecc=v["ecc"], omega=v["omega"])
        return P.LimbDarkLightCurve(v["u1"], v["u2"]).get_light_curve(orbit=o, r=v["r"], t=t, use_in_transit=False)

    for d in range(D):
        np.testing.assert_allclose(npy(flux[d]), oracle(d), rtol=0, atol=1e-13)
        for k in base:
            x0 = float(vals[k][d, 0])
            h = 1e-7 * max(1.0, abs(x0))
            fd = ((oracle(d, **{k: x0 + h}) - oracle(d, **{k: x0 - h})) * g[d]).sum() / (2 * h)
            got = float(leaves[k].grad.reshape(D, -1)[d, 0])
            assert abs(got - fd) <= 2e-5 * max(1.0, abs(fd)), (k, got, fd)


def test_graphed_step_matches_eager(dev):
    """hipGraph replay of a value+gradient step == the eager step, also after the
    parameters change"""
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    rng = np.random.default_rng(5)
    D = 6
    t = torch.tensor(np.arange(4000) * (2.0 / 1440.0) + 0.8, device=dev)
    g = torch.tensor(rng.normal(size=(D, t.numel())), device=dev)
    base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1)
    leaves = [torch.tensor(v * (1 + 1e-2 * rng.normal(size=(D, 1))), device=dev, requires_grad=True) for v in base.values()]

    def step(period, t0, b, ecc, omega, r):
        orbit = xo.KeplerianOrbit(period=period, t0=t0, b=b, ecc=ecc, omega=omega)
        rec, ld, _, flags = orbit.kernel_inputs(r, (0.3, 0.2))
        flux, L = ops.transit_flux_dot(t, rec, ld, g, flags=flags)
        return (flux, L) + torch.autograd.grad(L.sum(), (period, t0, b, ecc, omega, r))

    graphed = xo.GraphedStep(step, *leaves)
    for trial in range(2):
        vals = [x.detach() * (1 + 1e-3 * trial) for x in leaves]
        out = graphed(*vals)
        want = step(*[v.clone().requires_grad_(True) for v in vals])
        for a, b in zip(out, want):
            assert torch.allclose(a, b, rtol=1e-12, atol=1e-14)
        assert out[0].min().item() < -1e-3


def test_total_light_curve_is_the_sum_over_planets(dev):
    """get_light_curve(total=True) == get_light_curve(...).sum(-1) (values and gradients), transit and secondary"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(71)
    D = 4
    t = torch.linspace(0, 20, 4000, dtype=torch.float64, device=dev)
    mk = lambda v: torch.tensor(np.asarray(v)[None, :] * (1 + 1e-3 * rng.normal(size=(D, len(v)))), dtype=torch.float64,  # noqa: E731
                                device=dev, requires_grad=True)
    L = dict(period=mk([3.0, 6.1]), t0=mk([1.5, 1.6]), b=mk([0.2, 0.4]), ecc=mk([0.1, 0.3]), omega=mk([0.5, -1.0]))
    r = mk([0.08, 0.05])
    for lcobj, kw in ((xo.LimbDarkLightCurve(0.3, 0.2), {}), (xo.LimbDarkLightCurve(0.3, 0.2), dict(texp=0.02, oversample=3)),
                      (xo.SecondaryEclipseLightCurve((0.3, 0.2), (0.4, 0.1), 0.3), {})):
        orbit = xo.KeplerianOrbit(**L)
        a = lcobj.get_light_curve(orbit=orbit, r=r, t=t, total=True, **kw)
        b = lcobj.get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, **kw).sum(-1)
        assert a.shape == b.shape == (D, t.numel())
        assert float((a - b).abs().max()) <= 4e-15
        w = torch.randn_like(a)
        ga = torch.autograd.grad((a * w).sum(), list(L.values()) + [r])
        gb = torch.autograd.grad((b * w).sum(), list(L.values()) + [r])
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) <= 1e-10 * float(y.abs().max())
