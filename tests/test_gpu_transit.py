"""GPU parity: the fused transit kernel (forward and reverse) vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def make_record(orbit, r, sbr=None, window=False, texp=None):
    """numpy-port orbit -> kernel parameter records [1,P,NPAR] (test-side helper)."""
    Pn = orbit.a.size
    rec = np.zeros((1, Pn, P.NPAR))
    ecc = orbit.ecc if orbit.ecc is not None else np.zeros(Pn)
    cw = orbit.cos_omega if orbit.ecc is not None else np.ones(Pn)
    sw = orbit.sin_omega if orbit.ecc is not None else np.zeros(Pn)
    rec[0, :, P.P_N] = orbit.n
    rec[0, :, P.P_TP] = orbit.t_periastron
    rec[0, :, P.P_ECC] = ecc
    rec[0, :, P.P_COSW] = cw
    rec[0, :, P.P_SINW] = sw
    rec[0, :, P.P_COSI] = orbit.cos_incl
    rec[0, :, P.P_SINI] = orbit.sin_incl
    rec[0, :, P.P_AOR] = orbit.a / orbit.r_star
    rec[0, :, P.P_ROR] = r / orbit.r_star
    rec[0, :, P.P_T0] = orbit.t0
    rec[0, :, P.P_PERIOD] = orbit.period
    rec[0, :, P.P_TS] = -np.inf
    rec[0, :, P.P_TE] = np.inf
    rec[0, :, P.P_TS2] = -np.inf
    rec[0, :, P.P_TE2] = np.inf
    if sbr is not None:
        rec[0, :, P.P_FRATIO] = sbr * (r / orbit.r_star) ** 2
    if window:
        Ml, Mr, flag = P.contact_points(orbit.a, ecc, cw, sw, orbit.cos_incl, orbit.sin_incl, orbit.r_star + r)
        assert np.all(flag == 0)
        hp = 0.5 * orbit.period
        ts = np.mod((Ml - orbit.M0) / orbit.n + hp, orbit.period) - hp
        te = np.mod((Mr - orbit.M0) / orbit.n + hp, orbit.period) - hp
        rec[0, :, P.P_TS] = np.where(ts > 0, ts - orbit.period, ts)
        rec[0, :, P.P_TE] = np.where(te < 0, te + orbit.period, te)
    return rec


CASES = {
    "c2_single_e03": dict(orbit=dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1), r=[0.1], u=(0.3, 0.2),
                          t=np.arange(6000) * (2.0 / 1440.0)),
    "two_planet_ecc": dict(orbit=dict(m_star=1.45, r_star=1.5, t0=[0.5, 17.4], period=[10.0, 5.3], ecc=[0.1, 0.8],
                                      omega=[0.5, 1.3], m_planet=[0.3, 0.5], b=[0.2, 0.5]),
                           r=[0.1, 0.05], u=(0.2, 0.3), t=np.linspace(-20, 20, 5000)),
    "circular": dict(orbit=dict(period=3.5, t0=1.0, b=0.3), r=[0.1], u=(0.3, 0.2), t=np.arange(5000) * (2.0 / 1440.0)),
    "contact_bug": dict(orbit=dict(period=3.456, ecc=0.6, omega=-1.5), r=[0.1], u=(0.3, 0.2),
                        t=np.linspace(-0.1, 0.1, 1000)),
}


def _run(dev, name, texp=None, per_planet=False, window=False, order=0):
    from exoplanet_amd import ops

    case = CASES[name]
    orbit = P.KeplerianOrbit(**{k: (np.array(v, dtype=float) if isinstance(v, list) else v)
                                for k, v in case["orbit"].items()})
    r = np.array(case["r"])
    t = case["t"]
    rec = make_record(orbit, r, window=window)
    c = P.get_cl(*case["u"])[None, :]
    kw = {}
    tk = {}
    if texp is not None:
        sdt, sw = P.exposure_stencil(7, order)
        kw = dict(texp=texp, stencil_dt=sdt, stencil_w=sw)
        tk = dict(texp=T(np.atleast_1d(texp), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    rng = np.random.default_rng(42)
    shape = (1, t.size, r.size) if per_planet else (1, t.size)
    g = rng.normal(size=shape)
    want_f, want_gp, want_gl = P.transit_flux_vjp(t, rec, c, g, per_planet=per_planet, window=window, **kw)
    flags = (ops.FLAG_PER_PLANET if per_planet else 0) | (ops.FLAG_WINDOW if window else 0)
    f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=flags, **tk)
    f2, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=flags, **tk)
    assert want_f.min() < -1e-3
    np.testing.assert_allclose(f.cpu().numpy(), want_f, rtol=0, atol=2e-13)
    assert torch.equal(f, f2)
    slots = list(P.GRAD_SLOTS[:-1])
    scale = np.abs(want_gp[..., slots]).max() + 1e-30
    assert np.abs(gp.cpu().numpy()[..., slots] - want_gp[..., slots]).max() <= 1e-9 * scale + 1e-12
    other = [k for k in range(P.NPAR) if k not in slots]
    assert np.all(gp.cpu().numpy()[..., other] == 0)
    np.testing.assert_allclose(gl.cpu().numpy(), want_gl, rtol=1e-10, atol=1e-10)
    return f


@pytest.mark.parametrize("name", list(CASES))
def test_transit_parity(dev, name):
    _run(dev, name)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_transit_parity_texp(dev, order):
    _run(dev, "two_planet_ecc", texp=0.1, order=order)
    _run(dev, "contact_bug", texp=0.02, order=order)


def test_transit_parity_per_planet(dev):
    _run(dev, "two_planet_ecc", per_planet=True)
    _run(dev, "two_planet_ecc", per_planet=True, texp=0.1)


def test_window_equals_full(dev):
    """use_in_transit=True == False (reference tests/light_curves_test.py:75-102,148-164)."""
    for name, texp in [("two_planet_ecc", None), ("two_planet_ecc", 0.1), ("contact_bug", 0.02), ("c2_single_e03", None)]:
        a = _run(dev, name, texp=texp, window=True)
        b = _run(dev, name, texp=texp, window=False)
        assert torch.allclose(a, b, rtol=0, atol=1e-15)


def test_autograd_matches_vjp(dev):
    from exoplanet_amd import ops

    case = CASES["c2_single_e03"]
    orbit = P.KeplerianOrbit(**case["orbit"])
    rec = make_record(orbit, np.array(case["r"]))
    D = 5
    rng = np.random.default_rng(1)
    recs = np.repeat(rec, D, axis=0)
    recs[:, :, P.P_ROR] *= 1 + 0.05 * rng.normal(size=(D, 1))
    recs[:, :, P.P_ECC] += 0.02 * rng.normal(size=(D, 1))
    c = np.repeat(P.get_cl(0.3, 0.2)[None, :], D, axis=0)
    t = case["t"]
    g = rng.normal(size=(D, t.size))
    pt = T(recs, dev).requires_grad_(True)
    ct = T(c, dev).requires_grad_(True)
    f = ops.transit_flux(T(t, dev), pt, ct)
    (f * T(g, dev)).sum().backward()
    want_f, want_gp, want_gl = P.transit_flux_vjp(t, recs, c, g)
    np.testing.assert_allclose(f.detach().cpu().numpy(), want_f, rtol=0, atol=2e-13)
    slots = list(P.GRAD_SLOTS[:-1])
    np.testing.assert_allclose(pt.grad.cpu().numpy()[..., slots], want_gp[..., slots], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(ct.grad.cpu().numpy(), want_gl, rtol=1e-10, atol=1e-10)


def test_secondary_eclipse_parity(dev):
    from exoplanet_amd import ops

    u1, u2, s, ror = (0.3, 0.2), (0.4, 0.1), 0.3, 0.08
    t = np.linspace(-6.435, 10.4934, 5000)
    for okw in (dict(period=1.543, t0=-0.123), dict(period=2.7, t0=0.4, ecc=0.1, omega=0.7, b=0.2)):
        orbit = P.KeplerianOrbit(**okw)
        want = P.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=orbit, r=ror, t=t, use_in_transit=False)
        rec = make_record(orbit, np.array([ror]), sbr=s)
        c = np.concatenate([P.get_cl(*u1), P.get_cl(*u2)])[None, :]
        f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=ops.FLAG_SECONDARY)
        np.testing.assert_allclose(f.cpu().numpy()[0], want[:, 0], rtol=0, atol=2e-13)
        g = np.random.default_rng(2).normal(size=(1, t.size))
        wf, wgp, wgl = P.transit_flux_vjp(t, rec, c, g, secondary=True)
        np.testing.assert_allclose(wf[0], want[:, 0], rtol=0, atol=1e-14)
        _, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev),
                                                    flags=ops.FLAG_SECONDARY)
        slots = list(P.GRAD_SLOTS)
        scale = np.abs(wgp[..., slots]).max()
        assert np.abs(gp.cpu().numpy()[..., slots] - wgp[..., slots]).max() <= 1e-9 * scale
        np.testing.assert_allclose(gl.cpu().numpy(), wgl, rtol=1e-9, atol=1e-9)


def test_empty_and_ragged(dev):
    from exoplanet_amd import ops

    case = CASES["c2_single_e03"]
    orbit = P.KeplerianOrbit(**case["orbit"])
    rec = make_record(orbit, np.array(case["r"]))
    c = P.get_cl(0.3, 0.2)[None, :]
    f = ops.transit_flux(T(np.zeros(0), dev), T(rec, dev), T(c, dev))
    assert f.shape == (1, 0)
    for n in (1, 63, 64, 65, 511, 513):
        t = 1.0 + np.linspace(-0.2, 0.2, n)
        f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev))
        np.testing.assert_allclose(f.cpu().numpy(), P.transit_flux(t, rec, c), rtol=0, atol=2e-13)
