"""GPU: boundary cases of the C ABI -- maximum planets / sub-exposures / state width,
empty batches, invalid arguments (status codes, no crashes), NaN propagation."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P
from test_gpu_transit import make_record

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def test_max_planets_and_subexposures(dev):
    """EXO_MAX_PLANETS = 16 planets, EXO_MAX_SUBEXP = 63 sub-exposures, against the C oracle"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(2)
    Pn = ops.MAX_PLANETS
    period = 10 ** rng.uniform(0.2, 1.2, Pn)
    orbit = P.KeplerianOrbit(period=period, t0=rng.uniform(0, 3, Pn), b=rng.uniform(0, 0.9, Pn),
                             ecc=rng.uniform(0, 0.5, Pn), omega=rng.uniform(-3, 3, Pn))
    rec = make_record(orbit, rng.uniform(0.02, 0.1, Pn))
    c = P.get_cl(0.3, 0.2)[None]
    t = np.linspace(0, 20, 3001)           # odd length: the 8-byte access path
    sdt, sw = P.exposure_stencil(ops.MAX_SUBEXP, 2)
    assert sdt.size == ops.MAX_SUBEXP
    g = rng.normal(size=(1, t.size, Pn))
    want_f, want_gp, want_gl = C.transit(t, rec, c, g, texp=0.03, stencil_dt=sdt, stencil_w=sw, per_planet=True)
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), texp=T([0.03], dev),
                                               stencil_dt=T(sdt, dev), stencil_w=T(sw, dev), flags=ops.FLAG_PER_PLANET)
    assert (want_f < 0).any(axis=(0, 1)).all()
    np.testing.assert_allclose(f.cpu().numpy(), want_f, rtol=0, atol=2e-13)
    sl = list(P.GRAD_SLOTS[:-1])
    assert np.abs(gp.cpu().numpy()[..., sl] - want_gp[..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
    with pytest.raises(ValueError):
        ops.transit_flux(T(t, dev), T(np.zeros((1, Pn + 1, ops.NPAR)), dev), T(c, dev))


def test_gp_max_state_width(dev):
    """J = 8 (4 complex terms) and J = 7 (1 real + 3 complex) against the C oracle"""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(4)
    N = 600
    t = np.sort(rng.uniform(0, 50, N))
    y = 0.3 * rng.normal(size=N)
    diag = 0.05 + 0.02 * rng.uniform(size=N)
    for n_real, n_cplx in ((0, 4), (1, 3)):
        ar, cr = rng.uniform(0.1, 0.4, n_real), rng.uniform(0.1, 1.0, n_real)
        ac, bc = rng.uniform(0.1, 0.5, n_cplx), rng.uniform(0.0, 0.05, n_cplx)
        cc, dc = rng.uniform(0.05, 0.5, n_cplx), rng.uniform(0.3, 3.0, n_cplx)
        co = (ar, cr, ac, bc, cc, dc)
        want, gw = C.celerite(t, y, diag, co, grad=True)
        real = T(np.stack([ar, cr], -1)[None], dev).requires_grad_(True)
        cplx = T(np.stack([ac, bc, cc, dc], -1)[None], dev).requires_grad_(True)
        yt = T(y[None], dev).requires_grad_(True)
        ll = celerite_loglike(T(t, dev), yt, T(diag[None], dev), real, cplx)
        assert abs(ll.item() - want) < 1e-10 * abs(want)
        ll.sum().backward()
        np.testing.assert_allclose(yt.grad.cpu().numpy()[0], gw["y"], rtol=1e-7, atol=1e-9)
        for k, nm in enumerate(("ac", "bc", "cc", "dc")):
            np.testing.assert_allclose(cplx.grad.cpu().numpy()[0, :, k], gw[nm], rtol=1e-6, atol=1e-9)
        if n_real:
            np.testing.assert_allclose(real.grad.cpu().numpy()[0, :, 0], gw["ar"], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(real.grad.cpu().numpy()[0, :, 1], gw["cr"], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):      # J = 17 > EXO_GP_MAX_J (9 .. 16 run on the sequential kernels: test_gpu_golden.py, gp_wide)
        celerite_loglike(T(t, dev), T(y[None], dev), T(diag[None], dev), T(np.zeros((1, 1, 2)), dev),
                         T(np.zeros((1, 8, 4)), dev))


def test_status_codes_for_invalid_arguments(dev):
    """the C ABI reports, never crashes: null pointers, bad counts, short workspace"""
    from exoplanet_amd import _lib

    lib = _lib.load()
    x = torch.zeros(64, dtype=torch.float64, device=dev)
    p = x.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.exo_kepler_f64(0, p, p, p, 4, st) == 1
    assert lib.exo_kepler_f64(p, p, p, p, -1, st) == 1
    assert lib.exo_kepler_f64(p, p, p, p, 0, st) == 0
    assert lib.exo_quad_solution_vector_f64(p, p, p, p, 0, 4, st) == 1          # only one of dsdb / dsdr
    assert lib.exo_transit_flux_fwd_f64(p, 4, 0, 0, 0, 0, 1, p, p, 1, 0, 0, p, p, 1 << 20, st) == 1   # n_planet = 0
    assert lib.exo_transit_flux_fwd_f64(p, 4, 0, 0, 0, 0, 1, p, p, 1, 1, 0, p, p, 8, st) == 3         # workspace too small
    assert lib.exo_transit_flux_fwd_f64(p, 4, 0, 3, 0, 0, 1, p, p, 1, 1, 0, p, p, 1 << 20, st) == 1   # n_texp not in {0,1,n}
    assert lib.exo_celerite_loglike_fwd_f64(p, p, p, 1, 4, p, 17, p, 0, 0, 1, p, 0, 0, 0, st) == 1     # J > 16 (EXO_GP_MAX_J)
    assert lib.exo_celerite_loglike_fwd_f64(p, p, p, 1, 4, p, 0, p, 1, 0, 1, p, p, 4, 0, st) == 3      # state too small
    assert lib.exo_celerite_loglike_fwd_f64(p, p, p, 1, 4, p, 0, p, 1, 0, 1, p, 0, 0, -1, st) == 1     # n_chunks < 0
    assert lib.exo_pack_records_f64(p, p, 1, 17, 0, p, p, st) == 1
    torch.cuda.synchronize()


def test_nan_parameters_propagate(dev):
    """NaN in -> NaN out, never a trap; e >= 1 is NaN (docstring keplerian.py:58)"""
    from exoplanet_amd import ops

    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec = np.repeat(make_record(orbit, np.array([0.1])), 3, axis=0)
    rec[1, 0, P.P_ECC] = 1.2
    rec[2, 0, P.P_AOR] = np.nan
    c = np.repeat(P.get_cl(0.3, 0.2)[None], 3, 0)
    t = np.linspace(0.5, 1.5, 700)
    f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=ops.FLAG_EXACT_SCAN).cpu().numpy()
    assert np.isfinite(f[0]).all() and f[0].min() < 0
    assert np.isnan(f[1]).any() and np.isnan(f[2]).any()
    f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev)).cpu().numpy()
    assert np.isfinite(f[0]).all() and np.isnan(f[1]).any() and np.isnan(f[2]).any()


def test_many_draws_few_cadences(dev):
    """D = 3000 draws x 100 cadences (one block per draw) and D = 1 x 1 cadence"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(6)
    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec1 = make_record(orbit, np.array([0.1]))
    D = 3000
    rec = np.repeat(rec1, D, axis=0)
    rec[:, 0, P.P_ROR] *= 1 + 0.05 * rng.normal(size=D)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    t = 1.0 + np.linspace(-0.1, 0.1, 100)
    g = rng.normal(size=(D, t.size))
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    wf, wgp, wgl = C.transit(t, rec, c, g)
    np.testing.assert_allclose(f.cpu().numpy(), wf, rtol=0, atol=2e-13)
    sl = list(P.GRAD_SLOTS[:-1])
    assert np.abs(gp.cpu().numpy()[..., sl] - wgp[..., sl]).max() <= 1e-9 * np.abs(wgp[..., sl]).max()
    f1 = ops.transit_flux(T(t[50:51], dev), T(rec1, dev), T(c[:1], dev))
    np.testing.assert_allclose(f1.cpu().numpy(), C.transit(t[50:51], rec1, c[:1])[0], rtol=0, atol=2e-13)
