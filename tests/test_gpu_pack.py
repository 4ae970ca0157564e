"""GPU: the fused record-packing kernel (KeplerianOrbit algebra + get_cl + windows,
forward and reverse) against the torch attribute algebra it replaces."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P

pytestmark = pytest.mark.gpu


def leaves(dev, D, circular, rng):
    base = dict(period=[3.5, 8.1], t0=[1.0, 2.2], b=[0.3, 0.55], r=[0.1, 0.06], m_star=[1.1], r_star=[0.9],
                m_planet=[1e-3, 3e-4], u1=0.3, u2=0.2, sbr=0.3, u1s=0.4, u2s=0.1)
    if not circular:
        base.update(ecc=[0.3, 0.1], omega=[1.1, -0.7])
    out = {}
    for k, v in base.items():
        v = np.atleast_1d(np.asarray(v, dtype=float))
        x = v * (1 + 0.02 * rng.normal(size=(D,) + v.shape)) if k not in ("u1", "u2", "u1s", "u2s", "sbr") \
            else (v * (1 + 0.02 * rng.normal(size=(D, 1))))[:, 0]
        out[k] = torch.tensor(x, dtype=torch.float64, device=dev, requires_grad=True)
    return out


@pytest.mark.parametrize("circular", [False, True])
@pytest.mark.parametrize("secondary", [False, True])
@pytest.mark.parametrize("window", [False, True])
def test_pack_matches_torch_algebra(dev, circular, secondary, window):
    import exoplanet_amd as xo

    rng = np.random.default_rng(11)
    L = leaves(dev, 5, circular, rng)
    okw = dict(period=L["period"], t0=L["t0"], b=L["b"], m_star=L["m_star"], r_star=L["r_star"], m_planet=L["m_planet"])
    if not circular:
        okw.update(ecc=L["ecc"], omega=L["omega"])
    sec = ((L["u1s"], L["u2s"]), L["sbr"]) if secondary else None
    fast = xo.KeplerianOrbit(**okw)
    assert fast._standard
    rec, ld, batch, flags = fast.kernel_inputs(L["r"], (L["u1"], L["u2"]), use_in_transit=window, secondary=sec)
    # the same orbit through a non-standard (but equivalent) parameterisation -> torch algebra
    okw2 = dict(okw)
    if not circular:
        okw2.pop("omega")
        okw2.update(sin_omega=torch.sin(L["omega"]), cos_omega=torch.cos(L["omega"]))
        slow = xo.KeplerianOrbit(**okw2)
    else:
        slow = xo.KeplerianOrbit(**okw)
        slow._standard = False
    assert not slow._standard
    rec2, ld2, batch2, flags2 = slow.kernel_inputs(L["r"], (L["u1"], L["u2"]), use_in_transit=window, secondary=sec)
    assert batch == batch2 == (5,) and flags == flags2
    a, b = rec.detach().cpu().numpy(), rec2.detach().cpu().numpy()
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ld.detach().cpu().numpy(), ld2.detach().cpu().numpy(), rtol=1e-13)
    # reverse: random cotangents on the differentiable slots
    from exoplanet_amd import ops

    w = torch.zeros_like(rec)
    slots = [ops.P_N, ops.P_TP, ops.P_ECC, ops.P_COSW, ops.P_SINW, ops.P_COSI, ops.P_AOR, ops.P_ROR, ops.P_FRATIO]
    w[..., slots] = torch.tensor(rng.normal(size=rec.shape[:2] + (len(slots),)), device=dev)
    wl = torch.tensor(rng.normal(size=ld.shape), device=dev)
    names = [k for k in L if (k not in ("ecc", "omega") or not circular) and (secondary or k not in ("sbr", "u1s", "u2s"))]
    g1 = torch.autograd.grad((rec * w).sum() + (ld * wl).sum(), [L[k] for k in names], allow_unused=True)
    g2 = torch.autograd.grad((rec2 * w).sum() + (ld2 * wl).sum(), [L[k] for k in names], allow_unused=True)
    for k, x, y in zip(names, g1, g2):
        assert (x is None) == (y is None), k
        if x is not None:
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=1e-10, atol=1e-12, err_msg=k)


def test_light_curve_through_both_paths(dev):
    import exoplanet_amd as xo

    t = np.linspace(-3, 30, 3000)
    kw = dict(period=np.array([3.5, 8.1]), t0=np.array([1.0, 2.2]), b=np.array([0.3, 0.55]), ecc=np.array([0.3, 0.1]),
              omega=np.array([1.1, -0.7]), m_star=1.1, r_star=0.9)
    r = np.array([0.1, 0.06])
    fast = xo.KeplerianOrbit(**kw)
    slow = xo.KeplerianOrbit(**kw)
    slow._standard = False
    for texp in (None, 0.05):
        a = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=fast, r=r, t=t, texp=texp)
        b = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=slow, r=r, t=t, texp=texp)
        want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.KeplerianOrbit(**kw), r=r, t=t, texp=texp,
                                                             use_in_transit=False)
        np.testing.assert_allclose(a.cpu().numpy(), want, rtol=0, atol=1e-13)
        np.testing.assert_allclose(b.cpu().numpy(), want, rtol=0, atol=1e-13)


def test_column_form_equals_the_stacked_form_for_mixed_shapes(dev):
    """ops.pack_records_cols: columns of shape (), (P,), (D, 1), (D, P), defaults left out, limb darkening () or (D,):
    the records and every gradient of pack_records on the stacked, broadcast inputs"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(13)
    D, Pn = 4, 3
    mk = lambda v, shape: torch.tensor(np.asarray(v) * (1 + 0.02 * rng.normal(size=shape)), dtype=torch.float64, device=dev,  # noqa: E731
                                       requires_grad=True)
    cols = dict(period=mk([3.5, 8.1, 12.3], (Pn,)), t0=mk(1.0, ()), b=mk(0.3, (D, 1)), ecc=mk([0.1, 0.2, 0.3], (D, Pn)),
                omega=mk([1.1, -0.7, 0.2], (Pn,)), r=mk([0.1, 0.06, 0.04], (D, Pn)), m_star=None, r_star=mk(0.9, ()),
                m_planet=None, sbr=None)
    u1, u2 = mk(0.3, ()), mk(0.2, (D,))
    rec, ld = ops.pack_records_cols(list(cols.values()), [u1, u2], D)
    defaults = dict(m_star=1.0, m_planet=0.0, sbr=0.0)
    full = [(torch.full((), defaults[k], dtype=torch.float64, device=dev) if v is None else v) for k, v in cols.items()]
    full = [x.reshape((1,) * (2 - x.dim()) + tuple(x.shape)).expand(D, Pn) for x in full]
    orbit_in = torch.stack(full, dim=-1).contiguous()
    ld_in = torch.stack([u1.expand(D), u2], dim=-1).contiguous()
    rec2, ld2 = ops.pack_records(orbit_in, ld_in, 0)
    assert rec.shape == (D, Pn, ops.NPAR) and torch.equal(rec, rec2) and torch.equal(ld, ld2)
    wr, wl = torch.randn_like(rec), torch.randn_like(ld)
    wr[~torch.isfinite(rec)] = 0.0
    leaves_ = [v for v in cols.values() if v is not None] + [u1, u2]
    g1 = torch.autograd.grad((torch.nan_to_num(rec, posinf=0.0, neginf=0.0) * wr).sum() + (ld * wl).sum(), leaves_, allow_unused=True)
    g2 = torch.autograd.grad((torch.nan_to_num(rec2, posinf=0.0, neginf=0.0) * wr).sum() + (ld2 * wl).sum(), leaves_, allow_unused=True)
    for x, a, b in zip(leaves_, g1, g2):
        assert a.shape == x.shape
        assert float((a - b).abs().max()) <= 1e-12 * max(float(b.abs().max()), 1e-30)
