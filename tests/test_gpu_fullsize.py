"""GPU parity at BASELINE.json's FULL sizes.  The C port of the oracle is fast enough
(tens of ms per 150 k-cadence evaluation) that a few draws are compared directly,
plus size-independent properties (window == full, linearity of the VJP, sharded ==
unsharded)."""
import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P
from test_gpu_transit import make_record

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def _perturbed(rec, D, rng, scale=1e-3):
    recs = np.repeat(rec, D, axis=0)
    for slot in (P.P_ROR, P.P_AOR, P.P_COSI):
        recs[:, :, slot] *= 1 + scale * rng.normal(size=recs.shape[:2])
    recs[:, :, P.P_ECC] = np.clip(recs[:, :, P.P_ECC] * (1 + scale * rng.normal(size=recs.shape[:2])), 0, 0.95)
    return recs


def test_c2_full_size(dev):
    """C2: N = 150 000, e = 0.3; 3 draws directly against the C port, value + VJP; window == full."""
    from exoplanet_amd import ops

    rng = np.random.default_rng(2)
    t = np.arange(150_000) * (2.0 / 1440.0)
    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec = _perturbed(make_record(orbit, np.array([0.1]), window=True), 3, rng)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], 3, 0)
    g = rng.normal(size=(3, t.size))
    want_f, want_gp, want_gl = C.transit(t, rec, c, g)
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    assert (want_f < -1e-3).sum() > 3000
    assert np.abs(f.cpu().numpy() - want_f).max() < 1e-12        # north-star gate is 1e-6
    sl = list(P.GRAD_SLOTS[:-1])
    assert np.abs(gp.cpu().numpy()[..., sl] - want_gp[..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
    np.testing.assert_allclose(gl.cpu().numpy(), want_gl, rtol=1e-9)
    fw, gpw, glw = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=ops.FLAG_WINDOW)
    assert torch.allclose(fw, f, rtol=0, atol=1e-15)
    assert torch.allclose(gpw[..., sl], gp[..., sl], rtol=1e-12, atol=1e-12)
    # linearity of the VJP in the cotangent
    _, gp2, gl2 = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(2.5 * g, dev))
    assert torch.allclose(gp2, 2.5 * gp, rtol=1e-12, atol=1e-12) and torch.allclose(gl2, 2.5 * gl, rtol=1e-12)
    # bit-reproducible reduction
    _, gp3, _ = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    assert torch.equal(gp3, gp)


def test_c4_full_size(dev):
    """C4: 4 planets, N = 200 000; 2 draws against the C port; draws evaluated in two
    shards equal the unsharded batch (the multi-GPU partition)."""
    from exoplanet_amd import ops

    rng = np.random.default_rng(4)
    t = np.arange(200_000) * (2.0 / 1440.0)
    orbit = P.KeplerianOrbit(period=np.array([3.5, 7.9, 13.1, 29.7]), t0=np.array([1.0, 2.3, 5.1, 11.7]),
                             b=np.array([0.3, 0.1, 0.5, 0.2]), ecc=np.array([0.05, 0.1, 0.2, 0.3]),
                             omega=np.array([1.1, -0.4, 2.0, 0.3]))
    rec = _perturbed(make_record(orbit, np.array([0.1, 0.05, 0.07, 0.03])), 6, rng)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], 6, 0)
    g = rng.normal(size=(6, t.size))
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    want_f, want_gp, want_gl = C.transit(t, rec[:2], c[:2], g[:2])
    assert np.abs(f.cpu().numpy()[:2] - want_f).max() < 1e-12
    sl = list(P.GRAD_SLOTS[:-1])
    assert np.abs(gp.cpu().numpy()[:2][..., sl] - want_gp[..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
    for lo, hi in ((0, 3), (3, 6)):
        fs, gps, gls = ops.transit_flux_value_and_vjp(T(t, dev), T(rec[lo:hi], dev), T(c[lo:hi], dev), T(g[lo:hi], dev))
        assert torch.allclose(fs, f[lo:hi], rtol=0, atol=1e-15)
        assert torch.allclose(gps, gp[lo:hi], rtol=1e-12, atol=1e-13)


def test_c5_full_size(dev):
    """C5: Kepler long cadence N = 65 000, texp = 29.4 min oversampled x7, secondary eclipse,
    3-term GP (J = 6): transit part against the C port, GP part against the C port."""
    from exoplanet_amd import ops
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(5)
    t = np.arange(65_000) * (29.4 / 1440.0)
    texp = 29.4 / 1440.0
    orbit = P.KeplerianOrbit(period=2.7, t0=0.4, ecc=0.1, omega=0.7, b=0.2)
    rec = _perturbed(make_record(orbit, np.array([0.08]), sbr=0.3), 2, rng)
    c = np.repeat(np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None], 2, 0)
    sdt, sw = P.exposure_stencil(7, 0)
    g = rng.normal(size=(2, t.size))
    want_f, want_gp, want_gl = C.transit(t, rec, c, g, texp=texp, stencil_dt=sdt, stencil_w=sw, secondary=True)
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), texp=T([texp], dev),
                                               stencil_dt=T(sdt, dev), stencil_w=T(sw, dev), flags=ops.FLAG_SECONDARY)
    assert np.abs(f.cpu().numpy() - want_f).max() < 1e-12
    sl = list(P.GRAD_SLOTS)
    assert np.abs(gp.cpu().numpy()[..., sl] - want_gp[..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
    np.testing.assert_allclose(gl.cpu().numpy(), want_gl, rtol=1e-9, atol=1e-12)
    # GP: SHO(rho=20,Q=2) + SHO(rho=10,Q=1) + SHO(rho=2,Q=1/sqrt2)
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s, r, q), q) for s, r, q in
             ((4e-4, 20.0, 2.0), (3e-4, 10.0, 1.0), (2e-4, 2.0, 1 / np.sqrt(2)))]
    co = tuple(np.concatenate(x) for x in zip(*parts))
    y = want_f[0] + 3e-4 * rng.normal(size=t.size)
    resid = (y - want_f[0])[None]
    diag = np.full((1, t.size), 9e-8)
    want_ll, gw = C.celerite(t, resid[0], diag[0], co, grad=True)
    real = np.zeros((1, 0, 2)); cplx = np.stack(co[2:], -1)[None]
    rt, dt_, ct = T(resid, dev).requires_grad_(True), T(diag, dev), T(cplx, dev).requires_grad_(True)
    ll = celerite_loglike(T(t, dev), rt, dt_, T(real, dev), ct)
    assert abs(ll.item() - want_ll) < 1e-10 * abs(want_ll)
    ll.sum().backward()
    np.testing.assert_allclose(rt.grad.cpu().numpy()[0], gw["y"], rtol=1e-7, atol=1e-6 * np.abs(gw["y"]).max())
    for k, nm in enumerate(("ac", "bc", "cc", "dc")):
        np.testing.assert_allclose(ct.grad.cpu().numpy()[0, :, k], gw[nm], rtol=1e-6)


def test_c3_gp_full_size(dev):
    """C3: N = 150 000, SHO GP (J = 2): log-likelihood and gradient against the C port."""
    from exoplanet_amd.gp import celerite_loglike

    rng = np.random.default_rng(3)
    N = 150_000
    t = np.arange(N) * (2.0 / 1440.0)
    co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3, 5.0, 1 / np.sqrt(2)), 1 / np.sqrt(2))
    y = 1e-3 * rng.normal(size=(2, N))
    diag = np.full((1, N), 2.5e-7)
    cplx = np.repeat(np.stack(co[2:], -1)[None], 2, 0)
    rt = T(y, dev).requires_grad_(True)
    ct = T(cplx, dev).requires_grad_(True)
    ll = celerite_loglike(T(t, dev), rt, T(diag, dev), T(np.zeros((2, 0, 2)), dev), ct)
    ll.sum().backward()
    for d in range(2):
        want_ll, gw = C.celerite(t, y[d], diag[0], co, grad=True)
        assert abs(ll[d].item() - want_ll) < 1e-10 * abs(want_ll)
        np.testing.assert_allclose(rt.grad.cpu().numpy()[d], gw["y"], rtol=1e-7, atol=1e-7 * np.abs(gw["y"]).max())
        for k, nm in enumerate(("ac", "bc", "cc", "dc")):
            np.testing.assert_allclose(ct.grad.cpu().numpy()[d, :, k], gw[nm], rtol=1e-6)


def test_c2_full_size_with_timing_variations(dev):
    """C2 shape with a timing table per draw (60 transits, 150 000 cadences).  The C port has no
    timing tables, so each draw is checked transit by transit: a single-planet TTV light curve is
    the C port's Keplerian one evaluated on the warped clock t - shift[bin(t)]; the per-transit
    cotangents add up to the t_periastron one; shards equal the batch."""
    from exoplanet_amd import ops

    rng = np.random.default_rng(7)
    t = np.arange(150_000) * (2.0 / 1440.0)
    D, n_tr = 6, 60
    recs, edges, shifts = [], [], []
    for d in range(D):
        orbit = P.TTVOrbit(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3]), ecc=np.array([0.3]),
                           omega=np.array([1.1]), ttvs=[0.02 * rng.normal(size=n_tr)])
        recs.append(make_record(orbit, np.array([0.1]))[0])
        e, s = orbit.kernel_tables()
        edges.append(e)
        shifts.append(s)
    rec = _perturbed(np.stack(recs), 1, rng)
    edges, shifts = np.stack(edges), np.stack(shifts)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    g = rng.normal(size=(D, t.size))
    args = (T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    f, gp, gl, gs = ops.transit_flux_value_and_vjp(*args, ttv=(T(edges, dev), T(shifts, dev)))
    sl = list(P.GRAD_SLOTS[:-1])
    for d in range(D):
        warped = t - shifts[d, 0][np.searchsorted(edges[d, 0], t)]
        want_f, want_gp, want_gl = C.transit(warped, rec[d:d + 1], c[d:d + 1], g[d:d + 1])
        assert (want_f < -1e-3).sum() > 3000
        assert np.abs(f[d].cpu().numpy() - want_f[0]).max() < 1e-12
        assert np.abs(gp[d].cpu().numpy()[..., sl] - want_gp[0][..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
        np.testing.assert_allclose(gl[d].cpu().numpy(), want_gl[0], rtol=1e-9)
    assert float(gs.abs().max()) > 0
    np.testing.assert_allclose(gs.sum(-1).cpu().numpy(), gp[..., P.P_TP].cpu().numpy(), rtol=1e-10)
    # draws in two shards == the batch (gshift goes through atomics: last bits may differ)
    for lo, hi in ((0, 4), (4, 6)):
        fs, gps, gls, gss = ops.transit_flux_value_and_vjp(T(t, dev), T(rec[lo:hi], dev), T(c[lo:hi], dev), T(g[lo:hi], dev),
                                                           ttv=(T(edges[lo:hi], dev), T(shifts[lo:hi], dev)))
        assert torch.equal(fs, f[lo:hi])
        # (another batch size is another block decomposition: sums in another order)
        assert torch.allclose(gps, gp[lo:hi], rtol=1e-11, atol=1e-13 * float(gp.abs().max()))
        assert torch.allclose(gls, gl[lo:hi], rtol=1e-11)
        assert torch.allclose(gss, gs[lo:hi], rtol=1e-11, atol=1e-13 * float(gs.abs().max()))


def test_c4_full_size_with_timing_variations(dev):
    """C4 shape (4 planets, 200 000 cadences) with per-draw timing tables and a 7-point exposure
    stencil, directly against the C port (which has the tables too: oracle/c oracle_transit_ttv,
    pinned to the numpy restatement in tests/test_oracle_ttv.py)"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(9)
    t = np.arange(200_000) * (2.0 / 1440.0)
    periods, t0s = np.array([3.5, 7.9, 13.1, 29.7]), np.array([1.0, 2.3, 5.1, 11.7])
    D = 2
    recs, edges, shifts = [], [], []
    for d in range(D):
        ttvs = [0.01 * rng.normal(size=int((t[-1] - a) / p) + 1) for p, a in zip(periods, t0s)]
        orbit = P.TTVOrbit(period=periods, t0=t0s, b=np.array([0.3, 0.1, 0.5, 0.2]), ecc=np.array([0.05, 0.1, 0.2, 0.3]),
                           omega=np.array([1.1, -0.4, 2.0, 0.3]), ttvs=ttvs)
        recs.append(make_record(orbit, np.array([0.1, 0.05, 0.07, 0.03]))[0])
        e, s = orbit.kernel_tables()
        edges.append(e)
        shifts.append(s)
    rec, edges, shifts = np.stack(recs), np.stack(edges), np.stack(shifts)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    g = rng.normal(size=(D, t.size))
    sdt, sw = P.exposure_stencil(7, 0)
    texp = 0.02
    want_f, want_gp, want_gl, want_gs = C.transit_ttv(t, rec, c, (edges, shifts), g, texp=texp, stencil_dt=sdt,
                                                       stencil_w=sw)
    f, gp, gl, gs = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev),
                                                   texp=T(np.array([texp]), dev), stencil_dt=T(sdt, dev),
                                                   stencil_w=T(sw, dev), ttv=(T(edges, dev), T(shifts, dev)))
    assert (want_f < -1e-3).sum() > 5000
    assert np.abs(f.cpu().numpy() - want_f).max() < 1e-12
    sl = list(P.GRAD_SLOTS[:-1])
    assert np.abs(gp.cpu().numpy()[..., sl] - want_gp[..., sl]).max() <= 1e-9 * np.abs(want_gp[..., sl]).max()
    np.testing.assert_allclose(gl.cpu().numpy(), want_gl, rtol=1e-9)
    assert np.abs(gs.cpu().numpy() - want_gs).max() <= 1e-9 * np.abs(want_gs).max()
