"""GPU: the one-call white-noise likelihood on the sparse light curve (exo_transit_chi2_vjp_f64) against the same
quantity assembled from the C port's DENSE light curve and its VJP: single planet, three planets with simultaneous
transits, exposure stencil + secondary eclipse, per-cadence weights, the contact-window flag."""
import numpy as np
import pytest
import torch

from oracle import c_port as C
from oracle import numpy_port as P
from test_gpu_transit import make_record

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def oracle_chi2(t, rec, c, obs, ivar, **kw):
    """chi2 over ALL cadences minus the empty-light-curve constant, and its gradient, from the dense C port"""
    D = rec.shape[0]
    flux, _, _ = C.transit(t, rec, c, None, **kw)
    w = np.broadcast_to(ivar, t.shape)
    chi2 = (w * ((flux - obs) ** 2 - obs ** 2)).sum(-1)
    _, gp, gl = C.transit(t, rec, c, 2.0 * w * (flux - obs), **kw)
    return flux, chi2, gp, gl


CASES = {
    "one_planet": dict(P=1, secondary=False, texp=None, per_cad_ivar=False, window=False),
    "three_planets_overlapping": dict(P=3, secondary=False, texp=None, per_cad_ivar=True, window=False),
    "secondary_texp": dict(P=2, secondary=True, texp=0.02, per_cad_ivar=True, window=False),
    "window": dict(P=2, secondary=False, texp=None, per_cad_ivar=False, window=True),
}


@pytest.mark.parametrize("case", list(CASES))
def test_chi2_matches_dense_oracle(dev, case):
    from exoplanet_amd import ops

    cfg = CASES[case]
    rng = np.random.default_rng(41)
    D, Pn, N = 5, cfg["P"], 6000
    t = np.linspace(0.0, 30.0, N)
    # commensurate periods and aligned t0: planets transit at the same time again and again
    period = np.array([3.0, 6.0, 4.5])[:Pn]
    t0 = np.array([1.5, 1.5, 1.52])[:Pn]
    rec = np.zeros((D, Pn, P.NPAR))
    for d in range(D):
        orbit = P.KeplerianOrbit(period=period * (1 + 1e-4 * rng.normal(size=Pn)), t0=t0 + 1e-3 * rng.normal(size=Pn),
                                 b=rng.uniform(0, 0.8, Pn), ecc=rng.uniform(0, 0.4, Pn), omega=rng.uniform(-3, 3, Pn))
        rec[d] = make_record(orbit, rng.uniform(0.03, 0.12, Pn), sbr=0.4, window=cfg["window"])[0]
    c = np.repeat(np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None], D, 0) * (1 + 1e-2 * rng.normal(size=(D, 6)))
    c = c if cfg["secondary"] else np.ascontiguousarray(c[:, :3])
    kw_o, kw_g = dict(secondary=cfg["secondary"], window=cfg["window"]), {}
    if cfg["texp"] is not None:
        sdt, sw = P.exposure_stencil(5, 0)
        kw_o.update(texp=cfg["texp"], stencil_dt=sdt, stencil_w=sw)
        kw_g = dict(texp=T([cfg["texp"]], dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    truth, _, _ = C.transit(t, rec[:1], c[:1], None, **kw_o)
    obs = truth[0] + 3e-4 * rng.normal(size=N)
    ivar = (1.0 / (3e-4 * (1 + 0.3 * rng.uniform(size=N))) ** 2) if cfg["per_cad_ivar"] else np.array([1.0 / 9e-8])
    flux, want, gp, gl = oracle_chi2(t, rec, c, obs, ivar, **kw_o)
    if Pn > 1:
        both = sum((make != 0).astype(int) for make in [C.transit(t, rec[:, p:p + 1], c, None, **kw_o)[0] for p in range(Pn)])
        assert (both > 1).sum() > 50, "no simultaneous transits in the test system"
    flags = (ops.FLAG_SECONDARY if cfg["secondary"] else 0) | (ops.FLAG_WINDOW if cfg["window"] else 0)
    rt, ct = T(rec, dev).requires_grad_(True), T(c, dev).requires_grad_(True)
    chi2 = ops.transit_chi2(T(t, dev), rt, ct, T(obs, dev), T(ivar, dev), flags=flags, **kw_g)
    got = chi2.detach().cpu().numpy()
    scale = (np.broadcast_to(ivar, t.shape) * obs ** 2).sum()          # the size of the sums the difference is taken from
    assert np.abs(got - want).max() <= 1e-12 * scale
    wgt = T(rng.normal(size=D), dev)
    (chi2 * wgt).sum().backward()
    wn = wgt.cpu().numpy()
    np.testing.assert_allclose(rt.grad.cpu().numpy(), wn[:, None, None] * gp, rtol=2e-9, atol=1e-9 * np.abs(gp).max())
    np.testing.assert_allclose(ct.grad.cpu().numpy(), wn[:, None] * gl, rtol=2e-9, atol=1e-9 * np.abs(gl).max())


def test_white_noise_loglike_and_argument_checks(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(43)
    N, D = 3000, 3
    t = np.linspace(0, 12, N)
    orbit = P.KeplerianOrbit(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3]), ecc=np.array([0.2]), omega=np.array([0.5]))
    rec = np.repeat(make_record(orbit, np.array([0.1])), D, 0) * (1 + 1e-3 * rng.normal(size=(D, 1, P.NPAR)))
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    flux, _, _ = C.transit(t, rec, c, None)
    y = 1.0 + flux[0] + 2e-4 * rng.normal(size=N)
    yerr = 2e-4
    want = -0.5 * (((y - 1.0 - flux) / yerr) ** 2).sum(-1) - N * np.log(yerr * np.sqrt(2 * np.pi))
    got = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), T(y, dev), yerr, mean=1.0)
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max()
    # the data-only terms are cached per (series, error bars, mean): same object again -> same value; the series changed
    # in place, another mean, per-cadence error bars -> recomputed
    yt = T(y, dev)
    a1 = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), yt, yerr, mean=1.0)
    a2 = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), yt, yerr, mean=1.0)
    assert torch.equal(a1, a2) and np.abs(a1.cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max()
    yt.add_(3e-4)
    want2 = -0.5 * (((y + 3e-4 - 1.0 - flux) / yerr) ** 2).sum(-1) - N * np.log(yerr * np.sqrt(2 * np.pi))
    a3 = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), yt, yerr, mean=1.0)
    assert np.abs(a3.cpu().numpy() - want2).max() <= 1e-10 * np.abs(want2).max()
    a4 = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), yt, yerr, mean=1.0 + 3e-4)
    assert np.abs(a4.cpu().numpy() - want).max() <= 1e-9 * np.abs(want).max()
    errs = yerr * (1 + 0.2 * rng.uniform(size=N))
    want5 = -0.5 * (((y + 3e-4 - 1.0 - flux) / errs) ** 2).sum(-1) - np.log(errs * np.sqrt(2 * np.pi)).sum()
    et = T(errs, dev)
    for _ in range(2):
        a5 = ops.white_noise_loglike(T(t, dev), T(rec, dev), T(c, dev), yt, et, mean=1.0)
        assert np.abs(a5.cpu().numpy() - want5).max() <= 1e-10 * np.abs(want5).max()
    with pytest.raises(ValueError):
        ops.transit_chi2(T(t, dev), T(rec, dev), T(c, dev), T(y[:-1], dev), T([1.0], dev))
    with pytest.raises(ValueError):
        ops.transit_chi2(T(t, dev), T(rec, dev), T(c, dev), T(y, dev), T(np.ones(7), dev))
    with pytest.raises(RuntimeError):   # per-cadence exposure times: not on the run-enumeration path
        sdt, sw = P.exposure_stencil(3, 0)
        ops.transit_chi2(T(t, dev), T(rec, dev), T(c, dev), T(y, dev), T([1.0], dev), texp=T(np.full(N, 0.01), dev),
                         stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))


def test_light_curve_level_likelihood_in_a_graph(dev):
    """LimbDarkLightCurve.white_noise_log_likelihood == the same likelihood from get_light_curve + torch, values and
    gradients of every leaf, eagerly and replayed as a hipGraph"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(47)
    D, N = 6, 8000
    t = torch.linspace(0.0, 20.0, N, dtype=torch.float64, device=dev)
    mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    L = dict(period=mk(3.5), t0=mk(1.0), b=mk(0.3), ecc=mk(0.2), omega=mk(1.1), r=mk(0.1))
    names = list(L)
    u = (0.3, 0.2)
    with torch.no_grad():
        orbit = xo.KeplerianOrbit(**{k: v[:1] for k, v in L.items() if k != "r"})
        y = 1.0 + xo.LimbDarkLightCurve(*u).get_light_curve(orbit=orbit, r=L["r"][:1], t=t, texp=0.01).sum(-1).reshape(-1)
        y = y + 2e-4 * torch.randn(N, dtype=torch.float64, device=dev)
    yerr = 2e-4

    def fused(*vals):
        Lv = dict(zip(names, vals))
        orbit = xo.KeplerianOrbit(**{k: v for k, v in Lv.items() if k != "r"})
        ll = xo.LimbDarkLightCurve(*u).white_noise_log_likelihood(orbit=orbit, r=Lv["r"], t=t, y=y, yerr=yerr, mean=1.0, texp=0.01)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    def dense(*vals):
        Lv = dict(zip(names, vals))
        orbit = xo.KeplerianOrbit(**{k: v for k, v in Lv.items() if k != "r"})
        f = xo.LimbDarkLightCurve(*u).get_light_curve(orbit=orbit, r=Lv["r"], t=t, texp=0.01).sum(-1)
        ll = -0.5 * (((y - 1.0 - f) / yerr) ** 2).sum(-1) - N * np.log(yerr * np.sqrt(2 * np.pi))
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    a, b = fused(*L.values()), dense(*L.values())
    assert float((a[0] - b[0]).abs().max()) <= 1e-10 * float(b[0].abs().max())
    for x, z in zip(a[1:], b[1:]):
        assert float((x - z).abs().max()) <= 1e-8 * float(z.abs().max())
    g = xo.GraphedStep(fused, *L.values())
    out = g()
    for x, z in zip(out, a):
        assert float((x - z).abs().max()) <= 1e-12 * float(z.abs().max())


def test_chi2_with_light_delay_and_flux_dot_with_exposure(dev):
    """the one-call likelihood with the fused light-travel delay, and flux_dot with an exposure stencil, against the
    dense light curve of the same public API (values and gradients)"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(53)
    D, N = 4, 6000
    t = torch.linspace(0.0, 15.0, N, dtype=torch.float64, device=dev)
    mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    L = dict(period=mk(3.5), t0=mk(1.0), b=mk(0.3), ecc=mk(0.2), omega=mk(1.1), r=mk(0.1))
    y = 1.0 + 2e-4 * torch.randn(N, dtype=torch.float64, device=dev)
    lcobj = xo.LimbDarkLightCurve(0.3, 0.2)

    def orbit():
        return xo.KeplerianOrbit(**{k: v for k, v in L.items() if k != "r"})

    ll = lcobj.white_noise_log_likelihood(orbit=orbit(), r=L["r"], t=t, y=y, yerr=2e-4, mean=1.0, light_delay=True)
    f = lcobj.get_light_curve(orbit=orbit(), r=L["r"], t=t, light_delay=True, use_in_transit=False, total=True)
    want = -0.5 * (((y - 1.0 - f) / 2e-4) ** 2).sum(-1) - N * np.log(2e-4 * np.sqrt(2 * np.pi))
    assert float((ll - want).abs().max()) <= 1e-10 * float(want.abs().max())
    ga = torch.autograd.grad(ll.sum(), list(L.values()))
    gb = torch.autograd.grad(want.sum(), list(L.values()))
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-8 * float(b.abs().max())
    # flux_dot with exposure-time integration
    from exoplanet_amd.light_curves.limb_dark import exposure_stencil
    sdt, sw = exposure_stencil(5, 0)
    w = torch.randn(D, N, dtype=torch.float64, device=dev)
    T_ = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731
    flux, dot = orbit().flux_dot(L["r"], (0.3, 0.2), t, w, texp=T_([0.02]), stencil=(T_(sdt), T_(sw)))
    f2 = lcobj.get_light_curve(orbit=orbit(), r=L["r"], t=t, texp=0.02, oversample=5, use_in_transit=False, total=True)
    assert float((flux - f2).abs().max()) <= 4e-15
    d2 = (f2 * w).sum(-1)
    assert float((dot - d2).abs().max()) <= 1e-12 * float(d2.abs().max())
    ga = torch.autograd.grad(dot.sum(), list(L.values()))
    gb = torch.autograd.grad(d2.sum(), list(L.values()))
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-9 * float(b.abs().max())


@pytest.mark.parametrize("planets,texp", [(1, None), (2, None), (2, 0.03)])
def test_chi2_with_timing_tables_matches_the_dense_sweep(dev, planets, texp):
    """exo_transit_chi2_ttv_vjp_f64 -- one evaluation per solved cadence for one planet without a stencil, three sweeps
    otherwise -- against chi^2 assembled from the dense timing-table sweep (itself checked against the oracle in
    tests/test_gpu_ttv.py): value, gradients of records / limb darkening / per-transit shifts; the oracle's numpy
    evaluation for the shifts as well"""
    from exoplanet_amd import ops
    from test_gpu_ttv import case_records, single_planet_draws

    rng = np.random.default_rng(43)
    rec, tables = single_planet_draws(6) if planets == 1 else case_records(draws=4)
    D, N = rec.shape[0], 9001
    t = np.linspace(-3.0, 84.0, N)
    c = np.stack([P.get_cl(0.3 + 0.02 * d, 0.2 - 0.01 * d) for d in range(D)])     # (normalised: zero flux off the disk)
    kw_g, kw_o = {}, {}
    if texp is not None:
        sdt, sw = P.exposure_stencil(5, 0)
        kw_g = dict(texp=T([texp], dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
        kw_o = dict(texp=texp, stencil_dt=sdt, stencil_w=sw)
    ttv = (T(tables[0], dev), T(tables[1], dev))
    truth = ops.transit_flux(T(t, dev), T(rec[:1], dev), T(c[:1], dev), ttv=(ttv[0][:1], ttv[1][:1]), **kw_g)[0].cpu().numpy()
    obs = truth + 3e-4 * rng.normal(size=N)
    ivar = 1.0 / (3e-4 * (1 + 0.3 * rng.uniform(size=N))) ** 2
    # dense composition: flux, then chi^2 and its cotangent in torch, VJP through the dense sweep
    flux = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), ttv=ttv, **kw_g)
    w, o = T(ivar, dev), T(obs, dev)
    want = (w * ((flux - o) ** 2 - o ** 2)).sum(-1)
    _, gp, gl, gs = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), 2.0 * w * (flux - o), ttv=ttv, **kw_g)
    rt, ct, st = T(rec, dev).requires_grad_(True), T(c, dev).requires_grad_(True), ttv[1].clone().requires_grad_(True)
    chi2 = ops.transit_chi2(T(t, dev), rt, ct, o, w, ttv=(ttv[0], st), **kw_g)
    scale = float((w * o * o).sum())
    assert float((chi2.detach() - want).abs().max()) <= 1e-12 * scale
    wgt = T(rng.normal(size=D), dev)
    (chi2 * wgt).sum().backward()
    for got, ref, shape in ((rt.grad, gp, (D, 1, 1)), (ct.grad, gl, (D, 1)), (st.grad, gs, (D, 1, 1))):
        ref = wgt.reshape(shape) * ref
        assert float(ref.abs().max()) > 0
        assert float((got - ref).abs().max()) <= 2e-9 * float(ref.abs().max())
    # the shifts' gradient against the oracle's dense evaluation
    f_o, _, _, gs_o = P.transit_flux_vjp(t, rec, c, 2.0 * ivar * (flux.cpu().numpy() - obs), ttv=tables, **kw_o)
    assert np.abs(f_o - flux.cpu().numpy()).max() < 2e-13
    assert np.abs(gs.cpu().numpy() - gs_o).max() <= 1e-8 * np.abs(gs_o).max()
    # twice the same bits (no atomics on this path)
    chi2b = ops.transit_chi2(T(t, dev), rt, ct, o, w, ttv=(ttv[0], st), **kw_g)
    assert torch.equal(chi2b, chi2)


def test_white_noise_log_likelihood_of_a_ttv_orbit(dev):
    """user level: LimbDarkLightCurve.white_noise_log_likelihood with a TTVOrbit = the Gaussian log-likelihood of
    get_light_curve's model, gradients to the per-transit offsets included; a batch of draws"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(44)
    D, N = 5, 6000
    t = torch.linspace(0.0, 40.0, N, dtype=torch.float64, device=dev)
    n_tr = 12
    mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    leaves = dict(period=mk(3.3), t0=mk(0.9), b=mk(0.3), r=mk(0.08))
    ttvs = torch.tensor(0.01 * rng.normal(size=(D, n_tr)), dtype=torch.float64, device=dev, requires_grad=True)
    y = torch.tensor(3e-4 * rng.normal(size=N), dtype=torch.float64, device=dev)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)

    def orbit():
        return xo.orbits.TTVOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ttvs=[ttvs])

    sigma = 3e-4
    ll = lc.white_noise_log_likelihood(orbit=orbit(), r=leaves["r"], t=t, y=y, yerr=sigma)
    assert ll.shape == (D,)
    model = lc.get_light_curve(orbit=orbit(), r=leaves["r"], t=t).sum(-1)
    want = (-0.5 * ((y - model) / sigma) ** 2).sum(-1) - N * np.log(sigma * np.sqrt(2 * np.pi))
    assert torch.allclose(ll, want, rtol=1e-11, atol=1e-7)
    names = list(leaves) + ["ttvs"]
    vals = list(leaves.values()) + [ttvs]
    g1 = torch.autograd.grad(ll.sum(), vals)
    g2 = torch.autograd.grad(want.sum(), vals)
    for n, a, b in zip(names, g1, g2):
        assert float(b.abs().max()) > 0, n
        assert float((a - b).abs().max()) <= 1e-8 * float(b.abs().max()), n


def test_one_system_many_jitter_chains_and_mismatched_error_bars(dev):
    """an UNBATCHED orbit with a (n_draw, 1) error bar ("one system, many jitter chains") returns yerr's draws --
    through both forms of white_noise_log_likelihood (constructor columns and records) -- with the gradient of every
    chain's error bar; a yerr whose draw count matches neither 1 nor the parameter batch is a clear ValueError"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(5)
    N, K = 4000, 7
    t = np.linspace(0.0, 12.0, N)
    td = T(t, dev)
    want_f = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(
        orbit=P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.2, omega=0.4), r=0.1, t=t, use_in_transit=False)[:, 0]
    assert want_f.min() < -1e-3
    y = want_f + 2e-4 * rng.normal(size=N)
    s = 2e-4 * (1 + 0.3 * rng.uniform(size=(K, 1)))
    r2 = ((y - want_f) ** 2).sum()
    want_ll = -0.5 * r2 / s[:, 0] ** 2 - N * np.log(s[:, 0]) - 0.5 * N * np.log(2 * np.pi)
    want_g = r2 / s[:, 0] ** 3 - N / s[:, 0]
    for standard in (True, False):
        yerr = T(s, dev).requires_grad_(True)
        kw = dict(period=3.5, t0=1.0, b=0.3, ecc=0.2, omega=0.4)
        if not standard:                      # a parameterisation the column form does not take: the record form
            kw["rho_star"] = 1.3
        orbit = xo.KeplerianOrbit(**kw)
        if not standard:
            po = P.KeplerianOrbit(**kw)
            f2 = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=po, r=0.1, t=t, use_in_transit=False)[:, 0]
            rr = ((y - f2) ** 2).sum()
            w_ll = -0.5 * rr / s[:, 0] ** 2 - N * np.log(s[:, 0]) - 0.5 * N * np.log(2 * np.pi)
            w_g = rr / s[:, 0] ** 3 - N / s[:, 0]
        else:
            w_ll, w_g = want_ll, want_g
        ll = xo.LimbDarkLightCurve(0.3, 0.2).white_noise_log_likelihood(orbit=orbit, r=0.1, t=td, y=T(y, dev), yerr=yerr)
        assert ll.shape == (K,)
        (g,) = torch.autograd.grad(ll.sum(), yerr)
        assert np.abs(ll.detach().cpu().numpy() - w_ll).max() <= 1e-9 * np.abs(w_ll).max()
        assert np.abs(g.cpu().numpy()[:, 0] - w_g).max() <= 1e-8 * np.abs(w_g).max()
    # a batch of 3 parameter sets with 7 error bars: neither one per draw nor one for all
    orbit3 = xo.KeplerianOrbit(period=T([3.5, 3.6, 3.7], dev).reshape(3, 1), t0=1.0, b=0.3)
    with pytest.raises(ValueError, match="error bars"):
        xo.LimbDarkLightCurve(0.3, 0.2).white_noise_log_likelihood(orbit=orbit3, r=T([[0.1]] * 3, dev), t=td, y=T(y, dev),
                                                                    yerr=T(s, dev))
