"""GPU: light-travel delay inside the fused kernels (EXO_FLAG_LIGHT_DELAY; reference
keplerian.py:411-470): flux against the oracle's restatement of _get_retarded_position, gradients
against the composed torch path (two ops.kepler calls + autograd: an independent derivation of the
hand-written reverse sweep through the delay) AND against central differences of the oracle's light curve
(P.LimbDarkLightCurve(...).get_light_curve(light_delay=True): nothing of the product path on that side)."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def T(a, dev, grad=False):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(grad)


PARAMS = dict(period=3.456, t0=1.45, b=0.35, ecc=0.35, omega=-1.3, m_star=1.2, r_star=1.1)


@pytest.mark.parametrize("ecc", [0.35, None])
def test_fused_light_delay_flux_vs_oracle(dev, ecc):
    import exoplanet_amd as xo

    t = np.linspace(0.0, 14.0, 40_001)
    kw = dict(PARAMS)
    if ecc is None:
        kw.pop("ecc"); kw.pop("omega")
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=0.11, t=t, light_delay=True)
    plain = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=0.11, t=t, use_in_transit=False)
    orbit = xo.KeplerianOrbit(**{k: T(v, dev) for k, v in kw.items()})
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    got = lc.get_light_curve(orbit=orbit, r=T(0.11, dev), t=T(t, dev), light_delay=True).cpu().numpy()
    assert want.min() < -5e-3
    assert np.abs(want - plain).max() > 1e-5            # the delay is visible at ingress / egress
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-12)
    # with an exposure time
    want_e = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=0.11, t=t[::7], texp=0.01,
                                                           oversample=5, order=2, light_delay=True)
    got_e = lc.get_light_curve(orbit=orbit, r=T(0.11, dev), t=T(t[::7], dev), texp=0.01, oversample=5, order=2,
                               light_delay=True).cpu().numpy()
    np.testing.assert_allclose(got_e, want_e, rtol=0, atol=2e-12)
    with pytest.raises(NotImplementedError):
        lc.get_light_curve(orbit=orbit, r=T(0.11, dev), t=T(t, dev), light_delay=True, use_in_transit=True)


def test_fused_light_delay_gradients_vs_composed_path(dev):
    import exoplanet_amd as xo

    rng = np.random.default_rng(41)
    t = T(np.linspace(0.0, 14.0, 9_001), dev)
    g = T(rng.normal(size=(9_001, 1)), dev)
    grads = {}
    for mode in ("fused", "composed"):
        leaves = {k: T(v, dev, True) for k, v in PARAMS.items()}
        r, u1, u2 = T(0.11, dev, True), T(0.3, dev, True), T(0.2, dev, True)
        orbit = xo.KeplerianOrbit(**leaves)
        lc = xo.LimbDarkLightCurve(u1, u2)
        if mode == "fused":
            f = lc.get_light_curve(orbit=orbit, r=r, t=t, light_delay=True)
        else:
            f = lc._composed(orbit, r, t, None, None, False, True)
        (f * g).sum().backward()
        grads[mode] = {k: float(v.grad) for k, v in list(leaves.items()) + [("r", r), ("u1", u1), ("u2", u2)]}
        grads[mode]["flux"] = f.detach()
    assert float((grads["fused"]["flux"] - grads["composed"]["flux"]).abs().max()) < 2e-12
    for k in PARAMS.keys() | {"r", "u1", "u2"}:
        a, b = grads["fused"][k], grads["composed"][k]
        assert abs(a - b) <= 2e-8 * abs(b) + 1e-10, (k, a, b)
    # and the delay's own contribution is there: without it the r_star gradient differs
    leaves = {k: T(v, dev, True) for k, v in PARAMS.items()}
    f0 = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(**leaves), r=T(0.11, dev), t=t,
                                                        use_in_transit=False)
    (f0 * g).sum().backward()
    assert abs(float(leaves["t0"].grad) - grads["fused"]["t0"]) > 1e-6 * abs(grads["fused"]["t0"])


def test_fused_light_delay_gradients_vs_oracle_central_differences(dev):
    """the reverse sweep through the delay against the ORACLE: L(theta) = sum(g * flux(theta)) with the
    flux from oracle/numpy_port.py's restatement of keplerian.py:411-470, differentiated by Richardson-extrapolated central
    differences in every parameter.  For the differences to MEAN something L has to be smooth in the time-like parameters: the
    flux has square-root kinks at the contacts, and on a coarse grid with a random cotangent a shift of the transit by h moves
    cadences across them (differences of `period` then disagree with each other at 1e-2).  So two transits are sampled densely
    (1.7 s cadence) with a smooth cotangent -- the sum is a quadrature of a smooth integral, differences at h and h / 2 agree to
    <= 2e-6 (checked on the CPU: t0 the worst) -- and the fused kernel's gradient must agree with them to 5e-6."""
    import exoplanet_amd as xo

    tc = PARAMS["t0"]
    t = np.concatenate([np.linspace(tc - 0.2, tc + 0.2, 20_001), np.linspace(tc + 2 * PARAMS["period"] - 0.2,
                                                                            tc + 2 * PARAMS["period"] + 0.2, 20_001)])
    g = (np.cos(30.0 * t) + 0.5)[:, None]
    base = dict(PARAMS, r=0.11, u1=0.3, u2=0.2)

    def L_oracle(th):
        kw = {k: th[k] for k in PARAMS}
        f = P.LimbDarkLightCurve(th["u1"], th["u2"]).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=th["r"], t=t, light_delay=True)
        return float((f * g).sum())

    def dL(k, h):
        up, dn = dict(base), dict(base)
        up[k] += h; dn[k] -= h
        return (L_oracle(up) - L_oracle(dn)) / (2 * h)

    leaves = {k: T(v, dev, True) for k, v in base.items()}
    orbit = xo.KeplerianOrbit(**{k: leaves[k] for k in PARAMS})
    f = xo.LimbDarkLightCurve(leaves["u1"], leaves["u2"]).get_light_curve(orbit=orbit, r=leaves["r"], t=T(t, dev), light_delay=True)
    (f * T(g, dev)).sum().backward()
    L0 = L_oracle(base)
    assert abs(L0) > 10.0 and abs(float((f.detach().cpu().numpy() * g).sum()) - L0) <= 1e-10 * abs(L0)
    for k, v in base.items():
        h = 2e-5 * max(abs(v), 0.1)       # (at 1e-4 the differences in ecc / m_star disagree with themselves at 7e-6)
        fd = (4.0 * dL(k, h) - dL(k, 2 * h)) / 3.0          # O(h^4)
        got = float(leaves[k].grad)
        assert abs(got - fd) <= 5e-6 * abs(fd) + 1e-8, (k, got, fd)


def test_fused_light_delay_secondary_eclipse_batch(dev):
    """transit + occultation (the flipped orbit's delay has the opposite sign), a batch of draws"""
    import exoplanet_amd as xo

    t = np.linspace(0.0, 9.0, 20_001)
    D = 3
    per = np.array([1.543, 1.6, 1.7])
    got = xo.SecondaryEclipseLightCurve((0.3, 0.2), (0.4, 0.1), 0.3).get_light_curve(
        orbit=xo.KeplerianOrbit(period=T(per[:, None], dev), t0=T(0.23, dev), b=T(0.2, dev), ecc=T(0.2, dev),
                                omega=T(0.7, dev), r_star=T(0.9, dev), m_star=T(1.0, dev)),
        r=T(0.09, dev), t=T(t, dev), light_delay=True).cpu().numpy()
    assert got.shape == (D, t.size, 1)
    for d in range(D):
        want = P.SecondaryEclipseLightCurve((0.3, 0.2), (0.4, 0.1), 0.3).get_light_curve(
            orbit=P.KeplerianOrbit(period=per[d], t0=0.23, b=0.2, ecc=0.2, omega=0.7, r_star=0.9, m_star=1.0), r=0.09, t=t,
            light_delay=True)
        np.testing.assert_allclose(got[d], want, rtol=0, atol=5e-12)
        assert want.min() < -5e-3
