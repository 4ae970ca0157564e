"""GPU parity: transit-timing variations in the fused kernels (exo_transit_flux_ttv_*), against
the oracle's record-level evaluation, its restatement of the reference glue
(get_light_curve(orbit=TTVOrbit), ttv.py + limb_dark.py) and the composed torch path."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_transit import T, make_record
from test_oracle_ttv import expected_times, ttv_case

pytestmark = pytest.mark.gpu


def npy(x):
    return x.detach().cpu().numpy()


def check(dev, t, rec, c, tables, texp=None, order=0, oversample=7, window=False, per_planet=False, secondary=False,
          seed=42):
    """forward, one-sweep value + VJP and autograd of the HIP path against the oracle"""
    from exoplanet_amd import ops

    edges, shift = tables
    D, Pn = rec.shape[:2]
    kw, tk = {}, {}
    if texp is not None:
        sdt, sw = P.exposure_stencil(oversample, order)
        kw = dict(texp=texp, stencil_dt=sdt, stencil_w=sw)
        tk = dict(texp=T(np.atleast_1d(texp), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    g = np.random.default_rng(seed).normal(size=(D, t.size, Pn) if per_planet else (D, t.size))
    want_f, want_gp, want_gl, want_gs = P.transit_flux_vjp(t, rec, c, g, per_planet=per_planet, window=window,
                                                           secondary=secondary, ttv=(edges, shift), **kw)
    flags = ((ops.FLAG_PER_PLANET if per_planet else 0) | (ops.FLAG_WINDOW if window else 0)
             | (ops.FLAG_SECONDARY if secondary else 0))
    ttv = (T(edges, dev), T(shift, dev))
    f = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=flags, ttv=ttv, **tk)
    f2, gp, gl, gs = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=flags,
                                                    ttv=ttv, **tk)
    assert want_f.min() < -1e-3
    np.testing.assert_allclose(npy(f), want_f, rtol=0, atol=2e-13)
    assert torch.equal(f, f2)
    slots = list(P.GRAD_SLOTS[:-1]) + ([P.GRAD_SLOTS[-1]] if secondary else [])
    scale = np.abs(want_gp[..., slots]).max() + 1e-30
    assert np.abs(npy(gp)[..., slots] - want_gp[..., slots]).max() <= 1e-9 * scale + 1e-12
    np.testing.assert_allclose(npy(gl), want_gl, rtol=1e-10, atol=1e-10)
    assert np.abs(want_gs).max() > 1e-6
    assert np.abs(npy(gs) - want_gs).max() <= 1e-9 * np.abs(want_gs).max() + 1e-12
    # the bins hold the whole t_periastron cotangent, split by transit
    np.testing.assert_allclose(npy(gs).sum(-1), npy(gp)[..., P.P_TP], rtol=1e-9, atol=1e-9 * scale)
    return f


def case_records(missing=True, draws=1, seed=11):
    """D draws of the two-planet case: each draw has its own timing offsets"""
    recs, edges, shifts = [], [], []
    for d in range(draws):
        kw = ttv_case(seed=seed + d, missing=missing)
        orbit = P.TTVOrbit(**kw)
        recs.append(make_record(orbit, np.array([0.08, 0.05]))[0])
        e, s = orbit.kernel_tables()
        edges.append(e)
        shifts.append(s)
    return np.stack(recs), (np.stack(edges), np.stack(shifts))


@pytest.mark.parametrize("missing", [False, True])
def test_ttv_parity(dev, missing):
    rec, tables = case_records(missing)
    t = np.linspace(0.0, 80.0, 6000)
    check(dev, t, rec, P.get_cl(0.3, 0.2)[None], tables)
    check(dev, t, rec, P.get_cl(0.3, 0.2)[None], tables, per_planet=True)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_ttv_parity_texp(dev, order):
    rec, tables = case_records()
    t = np.linspace(0.0, 80.0, 5001)     # odd count: the scalar-load variant of the kernels
    check(dev, t, rec, P.get_cl(0.3, 0.2)[None], tables, texp=0.1, order=order)
    # one exposure time per cadence
    texp = np.random.default_rng(1).uniform(0.02, 0.2, t.size)
    check(dev, t, rec, P.get_cl(0.3, 0.2)[None], tables, texp=texp, order=order, per_planet=True)


def test_ttv_draws_have_their_own_tables(dev):
    rec, tables = case_records(draws=5)
    c = np.stack([P.get_cl(0.1 + 0.1 * d, 0.2) for d in range(5)])
    t = np.linspace(-10.0, 95.0, 7000)    # also before the first and after the last labelled transit
    check(dev, t, rec, c, tables)
    check(dev, t, rec, c, tables, texp=0.05)


def single_planet_draws(D, seed=21):
    """D draws of one planet, each with its own timing offsets (the scan kernel's grouped path:
    four draws per block, each draw's current bin cached between tiles)"""
    rng = np.random.default_rng(seed)
    recs, edges, shifts = [], [], []
    for d in range(D):
        period, t0 = 6.1 + 0.01 * d, 1.3
        n_tr = int((80.0 - t0) / period) + 1
        orbit = P.TTVOrbit(period=np.array([period]), t0=np.array([t0]), b=np.array([0.3]), ecc=np.array([0.2]),
                           omega=np.array([0.9]), ttvs=[0.04 * rng.normal(size=n_tr)])
        recs.append(make_record(orbit, np.array([0.07 + 0.005 * d]))[0])
        e, s = orbit.kernel_tables()
        edges.append(e)
        shifts.append(s)
    E = max(e.shape[1] for e in edges)
    pad_e = [np.pad(e, ((0, 0), (0, E - e.shape[1])), constant_values=np.inf) for e in edges]
    pad_s = [np.pad(x, ((0, 0), (0, E + 1 - x.shape[1])), mode="edge") for x in shifts]
    return np.stack(recs), (np.stack(pad_e), np.stack(pad_s))


@pytest.mark.parametrize("D", [1, 4, 6])
def test_ttv_single_planet_draw_groups(dev, D):
    rec, tables = single_planet_draws(D)
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    t = np.linspace(-3.0, 84.0, 9000)
    check(dev, t, rec, c, tables)
    check(dev, t, rec, c, tables, texp=0.07, order=1)
    check(dev, t[:-1], rec, c, tables, texp=0.07)                       # odd count
    check(dev, np.random.default_rng(0).permutation(t), rec, c, tables, texp=0.07)


def test_ttv_unsorted_times(dev):
    rec, tables = case_records()
    t = np.random.default_rng(3).permutation(np.linspace(0.0, 80.0, 6000))
    check(dev, t, rec, P.get_cl(0.3, 0.2)[None], tables, texp=0.05)


def test_ttv_window_semantics(dev):
    """use_in_transit: the caller's windows act on the warped mid-exposure time
    (keplerian.py:729-731 through ttv.py:179-187) and give the same light curve"""
    orbit = P.TTVOrbit(**ttv_case())
    r = np.array([0.08, 0.05])
    tables = tuple(x[None] for x in orbit.kernel_tables())
    t = np.linspace(0.0, 80.0, 6000)
    c = P.get_cl(0.3, 0.2)[None]
    for texp in (None, 0.1):
        a = check(dev, t, make_record(orbit, r, window=True), c, tables, texp=texp, window=True, per_planet=True)
        b = check(dev, t, make_record(orbit, r), c, tables, texp=texp, per_planet=True)
        assert torch.allclose(a, b, rtol=0, atol=1e-15)


def test_ttv_secondary(dev):
    """the occultation follows the same warped clock (ops level: the reference's TTVOrbit has no
    flipped orbit, secondary_eclipse.py:52 would raise)"""
    orbit = P.TTVOrbit(**ttv_case(missing=False))
    rec = make_record(orbit, np.array([0.08, 0.05]), sbr=0.4)
    tables = tuple(x[None] for x in orbit.kernel_tables())
    c6 = np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.1, 0.4)])[None]
    t = np.linspace(0.0, 80.0, 6000)
    check(dev, t, rec, c6, tables, secondary=True)
    check(dev, t, rec, c6, tables, secondary=True, texp=0.1)


def test_bin_edge_inside_an_exposure(dev):
    """arbitrary tables: an edge in the middle of a transit, different clocks on either side --
    sub-exposures of one cadence then belong to different bins (ttv.py warps every grid time),
    and their cotangents go to different shift entries"""
    orbit = P.KeplerianOrbit(period=5.0, t0=1.0, b=0.3, ecc=0.2, omega=0.7)
    rec = make_record(orbit, np.array([0.1]))
    edges = np.array([[[-1.5, 1.02, 3.5, 6.03, 8.5, np.inf]]])
    shift = np.array([[[0.0, 0.0, 0.04, 5.0, 5.03, 10.0, 10.0]]])
    t = np.linspace(-2.0, 12.0, 4000)
    c = P.get_cl(0.3, 0.2)[None]
    for texp, order in ((0.06, 0), (0.3, 2)):
        check(dev, t, rec, c, (edges, shift), texp=texp, order=order)
    check(dev, t, rec, c, (edges, shift))


def test_single_bin_table_is_a_time_offset(dev):
    from exoplanet_amd import ops

    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec = make_record(orbit, np.array([0.1]))
    t = np.arange(6000) * (2.0 / 1440.0)
    c = P.get_cl(0.3, 0.2)[None]
    edges = np.full((1, 1, 1), np.inf)
    shift = np.full((1, 1, 2), 0.0123)
    got = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), ttv=(T(edges, dev), T(shift, dev)))
    want = ops.transit_flux(T(t - 0.0123, dev), T(rec, dev), T(c, dev))
    assert float(want.min()) < -5e-3
    np.testing.assert_allclose(npy(got), npy(want), rtol=0, atol=1e-12)


def test_argument_checks(dev):
    from exoplanet_amd import ops

    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3)
    rec = T(make_record(orbit, np.array([0.1])), dev)
    t = T(np.linspace(0, 5, 100), dev)
    c = T(P.get_cl(0.3, 0.2)[None], dev)
    with pytest.raises(ValueError):
        ops.transit_flux(t, rec, c, ttv=(T(np.zeros((1, 2, 3)), dev), T(np.zeros((1, 2, 4)), dev)))
    with pytest.raises(ValueError):
        ops.transit_flux(t, rec, c, ttv=(T(np.zeros((1, 1, 3)), dev), T(np.zeros((1, 1, 3)), dev)))
    with pytest.raises(RuntimeError):
        ops.transit_flux(t, rec, c, ttv=(torch.zeros(1, 1, 3, dtype=torch.float64), T(np.zeros((1, 1, 4)), dev)))


# ------------------------------------------------------------------ the reference-shaped API
def torch_case(dev, kw, requires_grad=False, draws=None, seed=0):
    rng = np.random.default_rng(seed)
    tk = {}
    for k, v in kw.items():
        if k == "transit_inds":
            tk[k] = v
        elif k == "ttvs":
            vals = v if draws is None else [np.stack([x + 0.02 * rng.normal(size=x.shape) for _ in range(draws)])
                                            for x in v]
            tk[k] = [T(x, dev).requires_grad_(requires_grad) for x in vals]
        else:
            tk[k] = T(v, dev)
    return tk


@pytest.mark.parametrize("texp", [None, 0.08])
def test_light_curve_api_matches_reference_glue(dev, texp):
    """LimbDarkLightCurve.get_light_curve(orbit=TTVOrbit): fused kernels == the oracle's
    restatement of limb_dark.py + ttv.py == the composed torch path, values and gradients"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit

    kw = ttv_case()
    r = np.array([0.08, 0.05])
    t = np.linspace(0.0, 80.0, 5000)
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.TTVOrbit(**kw), r=r, t=t, texp=texp)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    g = T(np.random.default_rng(2).normal(size=want.shape), dev)
    grads = []
    for route in ("fused", "composed"):
        tk = torch_case(dev, kw, requires_grad=True)
        period = tk["period"].requires_grad_(True)
        orbit = TTVOrbit(**tk)
        if route == "fused":
            got = lc.get_light_curve(orbit=orbit, r=T(r, dev), t=T(t, dev), texp=texp)
        else:
            stencil = None if texp is None else xo.light_curves.limb_dark.exposure_stencil(7, 0)
            got = lc._composed(orbit, T(r, dev), T(t, dev), texp, stencil, True, False)
        np.testing.assert_allclose(npy(got), want, rtol=0, atol=1e-12)
        grads.append(torch.autograd.grad((got * g).sum(), tk["ttvs"] + [period]))
    for a, b in zip(*grads):
        assert float(b.abs().max()) > 0
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-8, atol=1e-9 * float(b.abs().max()))


def test_light_curve_api_draws(dev):
    """ttvs with a leading draw dimension: every draw equals its own unbatched evaluation"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit

    kw = ttv_case()
    r, t = T(np.array([0.08, 0.05]), dev), T(np.linspace(0.0, 80.0, 5000), dev)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    tk = torch_case(dev, kw, draws=4, seed=9)
    flux = lc.get_light_curve(orbit=TTVOrbit(**tk), r=r, t=t, texp=0.05)
    assert tuple(flux.shape) == (4, 5000, 2)
    for d in range(4):
        one = dict(tk, ttvs=[x[d] for x in tk["ttvs"]])
        want = lc.get_light_curve(orbit=TTVOrbit(**one), r=r, t=t, texp=0.05)
        assert torch.equal(flux[d], want)


def test_transit_times_parameterisation(dev):
    """transit_times given: least-squares ephemeris (ttv.py:91-137); gradient reaches the times"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit

    rng = np.random.default_rng(8)
    expect = expected_times(0.0, 60.0, [7.3, 11.9], [2.0, 5.5])
    times = [e + 0.03 * rng.normal(size=e.size) for e in expect]
    r, t = np.array([0.08, 0.05]), np.linspace(0.0, 60.0, 4000)
    want = P.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=P.TTVOrbit(transit_times=times, b=np.array([0.2, 0.4])),
                                                          r=r, t=t)
    tt = [T(x, dev).requires_grad_(True) for x in times]
    lc = xo.LimbDarkLightCurve(0.3, 0.2)
    got = lc.get_light_curve(orbit=TTVOrbit(transit_times=tt, b=T(np.array([0.2, 0.4]), dev)), r=T(r, dev), t=T(t, dev))
    np.testing.assert_allclose(npy(got), want, rtol=0, atol=1e-12)
    g = T(rng.normal(size=want.shape), dev)
    ga = torch.autograd.grad((got * g).sum(), tt)
    tt2 = [T(x, dev).requires_grad_(True) for x in times]
    comp = lc._composed(TTVOrbit(transit_times=tt2, b=T(np.array([0.2, 0.4]), dev)), T(r, dev), T(t, dev), None, None,
                        True, False)
    gb = torch.autograd.grad((comp * g).sum(), tt2)
    for a, b in zip(ga, gb):
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-8, atol=1e-9 * float(b.abs().max()))


def test_secondary_eclipse_with_ttv_orbit_raises_like_the_reference(dev):
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit

    orbit = TTVOrbit(**torch_case(dev, ttv_case()))
    lc = xo.SecondaryEclipseLightCurve((0.3, 0.2), (0.1, 0.1), 0.3)
    with pytest.raises(ValueError):
        lc.get_light_curve(orbit=orbit, r=T(np.array([0.08, 0.05]), dev), t=T(np.linspace(0, 10, 50), dev))


def test_ttv_step_replays_as_a_hip_graph(dev):
    """TTVOrbit construction + fused light curve + backward inside GraphedStep: nothing on the way
    synchronises with the host or uploads from it (transit_inds are cached on the device)"""
    import exoplanet_amd as xo
    from exoplanet_amd.orbits import TTVOrbit

    kw = ttv_case()
    tk = torch_case(dev, kw, draws=3, seed=4)
    tk.pop("transit_inds")
    inds = kw["transit_inds"]
    t, r = T(np.linspace(0.0, 80.0, 4000), dev), T(np.array([0.08, 0.05]), dev)
    g = T(np.random.default_rng(1).normal(size=(3, 4000, 2)), dev)
    names = ["period", "t0", "b", "ecc", "omega"]
    leaves = [tk[k].clone().requires_grad_(True) for k in names] + [x.clone().requires_grad_(True) for x in tk["ttvs"]]

    def step(*vals):
        orbit = TTVOrbit(**dict(zip(names, vals[:5])), ttvs=list(vals[5:]), transit_inds=inds)
        flux = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t, texp=0.05)
        return (flux.detach(),) + torch.autograd.grad((flux * g).sum(), vals)

    want = step(*leaves)
    graphed = xo.GraphedStep(step, *leaves)
    got = graphed()
    for a, b in zip(got, want):
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-9, atol=1e-12 * float(b.abs().max()))
    # new parameter values through the static inputs
    moved = [x.detach() + 1e-3 for x in leaves[:2]] + [x.detach() for x in leaves[2:]]
    got = graphed(*moved)
    want = step(*[x.clone().requires_grad_(True) for x in moved])
    for a, b in zip(got, want):
        np.testing.assert_allclose(npy(a), npy(b), rtol=1e-9, atol=1e-12 * float(b.abs().max()))


def test_large_irregular_table(dev):
    """4097 bins whose widths grow geometrically: the linear-ephemeris guess is wrong almost
    everywhere and every lookup falls back to the search (17 levels); shifts keep the planet's
    clock continuous only on average"""
    rng = np.random.default_rng(12)
    orbit = P.KeplerianOrbit(period=0.9, t0=0.2, b=0.3, ecc=0.1, omega=0.4)
    rec = make_record(orbit, np.array([0.1]))
    E = 4097
    edges = np.cumsum(1e-3 * 1.0015 ** np.arange(E))[None, None]          # 0.001 .. ~300, ascending
    shift = 0.9 * rng.integers(-3, 4, size=(1, 1, E + 1)) + 0.02 * rng.normal(size=(1, 1, E + 1))
    t = np.sort(rng.uniform(-1.0, edges.max() + 1.0, 6000))
    c = P.get_cl(0.3, 0.2)[None]
    check(dev, t, rec, c, (edges, shift))
    check(dev, t, rec, c, (edges, shift), texp=0.01, order=1)
    # two planets share nothing but the time axis: generic scan path
    orbit2 = P.KeplerianOrbit(period=np.array([0.9, 1.7]), t0=np.array([0.2, 0.5]), b=np.array([0.3, 0.1]))
    rec2 = make_record(orbit2, np.array([0.1, 0.06]))
    edges2 = np.concatenate([edges, edges[:, :, ::-1].max() - edges[:, :, ::-1] + 1e-3], axis=1)
    edges2[0, 1] = np.sort(edges2[0, 1])
    shift2 = np.concatenate([shift, 1.7 * rng.integers(-2, 3, size=(1, 1, E + 1))], axis=1)
    check(dev, t, rec2, c, (edges2, shift2), texp=0.01)


@pytest.mark.parametrize("planets", [1, 2])
def test_ttv_fast_classifier_equals_exact_scan(dev, planets):
    """the run enumeration over the timing bins (one planet, or several: transits only) never drops a cadence on the
    warped clock: EXO_FLAG_EXACT_SCAN (fp64 classification of every cadence, list path) gives the same flux up to the
    wave-vote rounding of the arc evaluation, exactly zero at the same cadences"""
    from exoplanet_amd import ops

    if planets == 1:
        rec, tables = single_planet_draws(6)
    else:
        rec, tables = case_records(draws=3)
    D = rec.shape[0]
    c = T(np.repeat(P.get_cl(0.3, 0.2)[None], D, 0), dev)
    t = T(np.linspace(-3.0, 84.0, 9001), dev)
    ttv = (T(tables[0], dev), T(tables[1], dev))
    g = T(np.random.default_rng(0).normal(size=(D, 9001)), dev)
    sdt, sw = P.exposure_stencil(7, 0)
    for kw in ({}, dict(texp=T(np.array([0.06]), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))):
        fast = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, **kw)
        exact = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, flags=ops.FLAG_EXACT_SCAN, **kw)
        assert float(fast[0].min()) < -1e-3
        assert bool(((fast[0] == 0) == (exact[0] == 0)).all())
        assert float((fast[0] - exact[0]).abs().max()) <= 4e-15
        for a, b in zip(fast[1:3], exact[1:3]):
            assert torch.allclose(a, b, rtol=1e-12, atol=1e-12 * float(b.abs().max()))


def test_shift_gradient_is_bit_reproducible_and_matches_the_list_path(dev):
    """run-enumeration path: d/d(shift) is collected run by run and combined in a fixed order -- the same bits call after
    call -- and agrees with the list path (EXO_FLAG_EXACT_SCAN: fp64 atomics) to rounding; so does a table whose bin
    edge cuts a transit in two (that list falls back to per-sample lookups)"""
    from exoplanet_amd import ops

    rec, tables = case_records(draws=5)
    D = rec.shape[0]
    c = T(np.repeat(P.get_cl(0.3, 0.2)[None], D, 0), dev)
    t = T(np.linspace(-3.0, 84.0, 20001), dev)
    g = T(np.random.default_rng(5).normal(size=(D, 20001)), dev)
    sdt, sw = P.exposure_stencil(5, 1)
    for kw in ({}, dict(texp=T(np.array([0.04]), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))):
        for cut in (False, True):
            edges, shift = tables[0].copy(), tables[1].copy()
            if cut:
                # an edge moved onto a transit: its neighbours' shifts differ, the exposure straddles it
                k = 2
                edges[:, 0, k] = rec[:, 0, ops.P_T0] + shift[:, 0, k] + 0.01
            ttv = (T(edges, dev), T(shift, dev))
            a = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, **kw)
            b = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, **kw)
            ref = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, flags=ops.FLAG_EXACT_SCAN, **kw)
            assert float(a[3].abs().max()) > 0
            if not cut:
                for x, y in zip(a, b):
                    assert torch.equal(x, y)
            assert float((a[0] - ref[0]).abs().max()) <= 4e-15
            for x, y in zip(a[1:], ref[1:]):
                assert float((x - y).abs().max()) <= 1e-11 * float(y.abs().max())
            # the bins' gradients add up to the t_periastron slot's (every sample's clock is t - shift - tp)
            tp_sum = a[1][:, :, ops.P_TP]
            assert float((a[3].sum(-1) - tp_sum).abs().max()) <= 1e-11 * float(tp_sum.abs().max())


def test_one_bin_covering_many_transits(dev):
    """a table that labels only the first transits: the last bin holds every later one (its windows are enumerated
    like those of an unperturbed orbit with one time offset)"""
    rng = np.random.default_rng(8)
    orbit = P.KeplerianOrbit(period=2.1, t0=0.7, b=0.2)
    rec = make_record(orbit, np.array([0.08]))
    tt = 0.7 + 2.1 * np.arange(4) + 0.03 * rng.normal(size=4)
    edges = (0.5 * (tt[1:] + tt[:-1]))[None, None]
    shift = (tt - 0.7)[None, None]
    t = np.linspace(-2.0, 150.0, 12001)
    c = P.get_cl(0.3, 0.2)[None]
    check(dev, t, rec, c, (edges, shift))
    check(dev, t, rec, c, (edges, shift), texp=0.02, order=1)


def test_sparse_output_with_timing_tables(dev):
    """EXO_FLAG_SPARSE through the timing-table entry points: the runs + values are the dense flux at the same cadences,
    zero elsewhere; same gradients, gshift included"""
    from exoplanet_amd import ops

    rec, tables = case_records(draws=4)
    D = rec.shape[0]
    c = T(np.repeat(P.get_cl(0.3, 0.2)[None], D, 0), dev)
    t = T(np.linspace(-3.0, 84.0, 12001), dev)
    g = T(np.random.default_rng(9).normal(size=(D, 12001)), dev)
    ttv = (T(tables[0], dev), T(tables[1], dev))
    sdt, sw = P.exposure_stencil(3, 0)
    for kw in ({}, dict(texp=T(np.array([0.03]), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))):
        f, gp, gl, gs = ops.transit_flux_value_and_vjp(t, T(rec, dev), c, g, ttv=ttv, **kw)
        sp = ops.transit_flux_sparse(t, T(rec, dev), c, ttv=ttv, **kw)
        dense = sp.to_dense(per_planet=False)
        assert np.array_equal(dense, npy(f))
        sp2, gp2, gl2, dot, gs2 = ops.transit_flux_sparse(t, T(rec, dev), c, gflux=g, ttv=ttv, **kw)
        assert np.array_equal(sp2.to_dense(per_planet=False), npy(f))
        assert torch.equal(gp2, gp) and torch.equal(gl2, gl) and torch.equal(gs2, gs)
        np.testing.assert_allclose(npy(dot), (npy(g) * npy(f)).sum(-1), rtol=1e-10, atol=1e-14)
        assert 0 < sp.n_solved() < 0.2 * D * 12001 * 2
    with pytest.raises(ValueError):
        ops.transit_flux_sparse(t, T(rec, dev), c, ttv=ttv, flags=ops.FLAG_SECONDARY)


@pytest.mark.parametrize("batched", [True, False])
def test_fused_table_construction_equals_the_torch_construction(dev, batched):
    """TTVOrbit.kernel_ttv: offsets for every transit go through exo_ttv_tables_f64 (one launch each way); the same
    tables and the same gradients (offsets, period) as the torch construction of ttv.py:99-170"""
    import exoplanet_amd as xo

    rng = np.random.default_rng(71)
    D = 5
    shape = (D, 1) if batched else ()
    mk = lambda v, s: torch.tensor(v * (1 + 1e-3 * rng.normal(size=s)), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    period = mk(np.array([3.3, 7.1]), (D, 2) if batched else (2,))
    t0 = mk(np.array([0.9, 2.2]), (2,))                       # shared by the draws
    b = torch.tensor([0.3, 0.1], dtype=torch.float64, device=dev)
    ttvs = [torch.tensor(0.01 * rng.normal(size=((D, n) if batched else (n,))), dtype=torch.float64, device=dev, requires_grad=True)
            for n in (12, 5)]
    wgt_e = wgt_s = None
    out = {}
    for fused in (True, False):
        orb = xo.orbits.TTVOrbit(period=period, t0=t0, b=b, ttvs=ttvs)
        if not fused:
            orb._fused_tables = lambda: None
        edges, shift = orb.kernel_ttv()
        if wgt_s is None:
            wgt_s = torch.tensor(rng.normal(size=tuple(shift.shape)), dtype=torch.float64, device=dev)
        grads = torch.autograd.grad((shift * wgt_s).sum(), [period] + ttvs, allow_unused=True)
        out[fused] = (edges.detach(), shift.detach(), grads)
        assert not edges.requires_grad
    (e1, s1, g1), (e0, s0, g0) = out[True], out[False]
    assert e1.shape == e0.shape and s1.shape == s0.shape
    assert torch.equal(torch.isinf(e1), torch.isinf(e0))
    fin = torch.isfinite(e0)
    assert float((e1[fin] - e0[fin]).abs().max()) <= 1e-13 * 100.0
    assert float((s1 - s0).abs().max()) <= 1e-13 * 100.0
    for a, b_ in zip(g1, g0):
        assert a.shape == b_.shape
        assert float((a - b_).abs().max()) <= 1e-12 * float(b_.abs().max())
    # and the light curve through both: the same flux
    t = torch.linspace(0.0, 38.0, 4000, dtype=torch.float64, device=dev)
    lcs = []
    for fused in (True, False):
        orb = xo.orbits.TTVOrbit(period=period, t0=t0, b=b, ttvs=ttvs)
        if not fused:
            orb._fused_tables = lambda: None
        lcs.append(xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orb, r=torch.tensor([0.08, 0.05], dtype=torch.float64, device=dev), t=t))
    assert float(lcs[0].min()) < -1e-3 and float((lcs[0] - lcs[1]).abs().max()) <= 1e-12
    # the public attributes are still there, built on first use
    orb = xo.orbits.TTVOrbit(period=period, t0=t0, b=b, ttvs=ttvs)
    assert len(orb.transit_times) == 2 and orb.all_transit_times[0].shape[-1] == 12
