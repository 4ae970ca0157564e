"""GPU: the Jacobian route of the light curve (include/exoplanet_amd.h, exo_transit_flux_fwd_jac_f64 / _jac_vjp_f64) -- the value
sweep keeps every solved cadence's row of derivatives, the gradient is a contraction -- against the two-sweep route it
replaces: the same flux bit for bit, the same gradients to rounding (the sums run in another order), for exposure stencils,
several planets, occultations, contact windows, row and cadence-major cotangents; and against the oracle's C port.
reference: src/exoplanet/light_curves/limb_dark.py:99-232 (the light curve as the mean of a GP: its cotangent arrives later)."""
import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P
from test_gpu_runs import system

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


@pytest.mark.parametrize("planets,secondary,window,cmajor", [(1, False, False, False), (3, False, False, True), (1, True, False, True),
                                                              (2, True, True, False)])
def test_jacobian_route_equals_two_sweeps(dev, planets, secondary, window, cmajor):
    from exoplanet_amd import ops

    rng = np.random.default_rng(100 + planets)
    D, N = 11, 30_011
    t = T(np.arange(N) * (10.0 / 1440.0) + 0.25, dev)
    rec, c = system(rng, D, planets, secondary, window=window)
    flags = (ops.FLAG_SECONDARY if secondary else 0) | (ops.FLAG_WINDOW if window else 0) | (ops.FLAG_CADENCE_MAJOR if cmajor else 0)
    dt, w = P.exposure_stencil(7, 0)
    kw = dict(texp=T([0.02], dev), stencil_dt=T(dt, dev), stencil_w=T(w, dev))
    g = T(rng.normal(size=(N, D)), dev).t() if cmajor else T(rng.normal(size=(D, N)), dev)
    out = {}
    for route in (True, False):
        ops._JAC_ROUTE[0] = route
        try:
            rt, ct = T(rec, dev).requires_grad_(True), T(c, dev).requires_grad_(True)
            n_before = ops._JAC_CALLS[0]
            flux = ops.transit_flux(t, rt, ct, flags=flags, **kw)
            assert (ops._JAC_CALLS[0] - n_before) == (1 if route else 0)      # (the route under test is the one that ran)
            gp, gl = torch.autograd.grad(flux, (rt, ct), grad_outputs=g)
            out[route] = (flux.detach().clone(), gp.clone(), gl.clone())
        finally:
            ops._JAC_ROUTE[0] = True
    (fa, gpa, gla), (fb, gpb, glb) = out[True], out[False]
    assert float(fb.min()) < -1e-3
    assert torch.equal(fa, fb)
    assert fa.stride() == fb.stride()
    for a, b in ((gpa, gpb), (gla, glb)):
        assert float((a - b).abs().max()) <= 1e-12 * float(b.abs().max())
    # record slots that carry no gradient read exactly 0 on both routes
    assert torch.equal(gpa == 0, gpb == 0)


def test_jacobian_route_vs_oracle(dev):
    """the C5 geometry (long cadence, 7 sub-exposures, transit + occultation) through the Jacobian route against the C port"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(7)
    D, N = 6, 20_000
    tt = np.arange(N) * (29.4 / 1440.0)
    rec, c = system(rng, D, 1, True)
    dt, w = P.exposure_stencil(7, 0)
    g = rng.normal(size=(D, N))
    want_f, want_gp, want_gl = C.transit(tt, rec, c, g, texp=29.4 / 1440.0, stencil_dt=dt, stencil_w=w, secondary=True)
    rt, ct = T(rec, dev).requires_grad_(True), T(c, dev).requires_grad_(True)
    flux = ops.transit_flux(T(tt, dev), rt, ct, flags=ops.FLAG_SECONDARY, texp=T([29.4 / 1440.0], dev), stencil_dt=T(dt, dev),
                            stencil_w=T(w, dev))
    assert ops._JAC_CALLS[0] > 0
    gp, gl = torch.autograd.grad(flux, (rt, ct), grad_outputs=T(g, dev))
    assert want_f.min() < -1e-3
    assert np.abs(flux.detach().cpu().numpy() - want_f).max() < 1e-12
    sl = list(P.GRAD_SLOTS)
    np.testing.assert_allclose(gp.cpu().numpy()[..., sl], want_gp[..., sl], rtol=1e-9, atol=1e-9 * np.abs(want_gp[..., sl]).max())
    np.testing.assert_allclose(gl.cpu().numpy(), want_gl, rtol=1e-9, atol=1e-9 * np.abs(want_gl).max())
