"""GPU: product ops / API straight against the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def test_kepler_golden(dev):
    from exoplanet_amd import ops
    from test_oracle import kepler_tol

    g = np.load(os.path.join(GOLD, "kepler.npz"))
    s, c = ops.kepler(T(g["M"], dev), T(g["ecc"], dev))
    tol = kepler_tol(g["M"], g["ecc"], g["cosf"])
    assert np.all(np.abs(s.cpu().numpy() - g["sinf"]) <= tol)
    assert np.all(np.abs(c.cpu().numpy() - g["cosf"]) <= tol)


def test_quad_sv_golden(dev):
    from exoplanet_amd import ops

    g = np.load(os.path.join(GOLD, "quad_sv.npz"))
    b = T(g["b"], dev).requires_grad_(True)
    r = T(g["r"], dev).requires_grad_(True)
    s = ops.quad_solution_vector(b, r)
    assert np.abs(s.detach().cpu().numpy() - g["s"]).max() < 5e-15
    for k in range(3):
        gb, gr = torch.autograd.grad(s[:, k].sum(), (b, r), retain_graph=True)
        gap = np.minimum.reduce([np.abs(np.abs(g["b"]) - np.abs(1 - g["r"])), np.abs(np.abs(g["b"]) - (1 + g["r"])),
                                 np.abs(np.abs(g["b"]) - g["r"]) + 1e-3])
        tol = 5e-14 + 2e-15 / np.sqrt(np.maximum(gap, 1e-16))
        assert np.all(np.abs(gb.cpu().numpy() - g["dsdb"][:, k]) <= tol)
        assert np.all(np.abs(gr.cpu().numpy() - g["dsdr"][:, k]) <= tol)


def test_lightcurves_golden(dev):
    import exoplanet_amd as xo
    from oracle.make_golden import LIGHTCURVE_CASES, case_time

    g = np.load(os.path.join(GOLD, "lightcurves.npz"))
    for name, case in LIGHTCURVE_CASES.items():
        okw = {k: (np.array(v, dtype=float) if isinstance(v, list) else v) for k, v in case["orbit"].items()}
        t = case_time(case["t"])
        for texp in case["texp"]:
            for uit in (None, False):
                got = xo.LimbDarkLightCurve(*case["u"]).get_light_curve(orbit=xo.KeplerianOrbit(**okw), r=np.array(case["r"]),
                                                                       t=t, texp=texp, use_in_transit=uit)
                np.testing.assert_allclose(got.cpu().numpy(), g[f"{name}_texp{texp}"], rtol=0, atol=1e-13)
    t = np.linspace(-6.435, 10.4934, 5000)
    got = xo.SecondaryEclipseLightCurve([0.3, 0.2], [0.4, 0.1], 0.3).get_light_curve(
        orbit=xo.KeplerianOrbit(period=1.543, t0=-0.123), r=0.08, t=t)
    np.testing.assert_allclose(got.cpu().numpy(), g["secondary"], rtol=0, atol=1e-13)


def test_gp_golden(dev):
    from exoplanet_amd.gp import celerite_loglike

    g = np.load(os.path.join(GOLD, "gp_sho.npz"))
    for tag in ("q03", "q07", "q3"):
        real = np.stack([g[f"{tag}_ar"], g[f"{tag}_cr"]], -1)[None]
        cplx = np.stack([g[f"{tag}_ac"], g[f"{tag}_bc"], g[f"{tag}_cc"], g[f"{tag}_dc"]], -1)[None]
        ll = celerite_loglike(T(g[f"{tag}_t"], dev), T(g[f"{tag}_y"][None], dev), T(g[f"{tag}_diag"][None], dev),
                              T(real, dev), T(cplx, dev))
        assert abs(ll.item() - float(g[f"{tag}_loglike"])) < 1e-12 * abs(float(g[f"{tag}_loglike"]))
