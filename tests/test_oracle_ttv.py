"""CPU: the oracle's TTVOrbit restatement (oracle/numpy_port.py) against the reference's own
TTV tests (/root/reference/tests/orbits/ttv_test.py), its two independent routes to a TTV light
curve against each other, and the host-side timing tables against the oracle's."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_oracle import _records


def expected_times(min_time, max_time, periods, t0s):
    from exoplanet_amd.orbits import compute_expected_transit_times

    return compute_expected_transit_times(min_time, max_time, np.array(periods), np.array(t0s))


def test_compute_expected_transit_times():
    """ttv_test.py:8-20"""
    periods, t0s, lo, hi = [10.5, 56.34], [45.3, 48.1], 456.023, 595.23
    for period, t0, times in zip(periods, t0s, expected_times(lo, hi, periods, t0s)):
        assert np.all(lo <= times) and np.all(times <= hi)
        assert times[0] - period < lo and times[-1] + period > hi


def test_consistency():
    """ttv_test.py:23-46: ttvs <-> transit_times round trip"""
    rng = np.random.default_rng(6934104)
    periods, t0s = [10.5, 56.34], [45.3, 48.1]
    expect = expected_times(456.023, 595.23, periods, t0s)
    ttvs = [0.01 * rng.normal(size=len(t)) for t in expect]
    orbit = P.TTVOrbit(period=np.array(periods), t0=np.array([t[0] for t in expect]), ttvs=ttvs)
    for i in range(2):
        assert np.allclose(orbit.transit_times[i], expect[i] + ttvs[i])
    orbit1 = P.TTVOrbit(transit_times=orbit.transit_times)
    orbit2 = P.TTVOrbit(period=orbit1.period, t0=orbit1.t0, ttvs=orbit1.ttvs)
    for i in range(2):
        assert np.allclose(orbit1.transit_times[i], orbit2.transit_times[i])
        assert np.allclose(orbit1.ttvs[i], orbit2.ttvs[i])


def test_no_ttvs_is_keplerian():
    """ttv_test.py:49-83 (the position the light curve uses)"""
    periods, t0s = np.array([10.5, 56.34]), np.array([45.3, 48.1])
    time = np.linspace(456.023, 595.23, 5000)
    expect = expected_times(456.023, 595.23, periods, t0s)
    orbit0 = P.KeplerianOrbit(period=periods, t0=t0s)
    orbit1 = P.TTVOrbit(period=periods, t0=np.array([t[0] for t in expect]), ttvs=[np.zeros_like(t) for t in expect])
    orbit2 = P.TTVOrbit(transit_times=expect)
    want = orbit0.get_relative_position(time)
    for orb in (orbit1, orbit2):
        for a, b in zip(want, orb.get_relative_position(time)):
            assert np.allclose(a, b)


def ttv_case(seed=11, missing=True):
    rng = np.random.default_rng(seed)
    periods, t0s = np.array([7.3, 11.9]), np.array([2.0, 5.5])
    expect = expected_times(0.0, 80.0, periods, t0s)
    inds = None
    if missing:
        # the second planet's third and fourth transits were not observed
        keep = [np.arange(expect[0].size), np.array([0, 1, 4, 5, 6][:expect[1].size - 2])]
        inds = keep
        expect = [e[k] for e, k in zip(expect, keep)]
    ttvs = [0.05 * rng.normal(size=e.size) for e in expect]
    kw = dict(period=periods, t0=np.array([e[0] for e in expect]) if not missing else t0s, b=np.array([0.2, 0.45]),
              ecc=np.array([0.1, 0.3]), omega=np.array([0.4, -1.0]), ttvs=ttvs, transit_inds=inds)
    return kw


@pytest.mark.parametrize("texp", [None, 0.08])
@pytest.mark.parametrize("window", [False, True])
def test_record_level_equals_class_level(texp, window):
    """oracle: get_light_curve(orbit=TTVOrbit) (reference glue restated) == the record-level
    evaluation with timing tables (what the HIP entry points are checked against)"""
    kw = ttv_case()
    orbit = P.TTVOrbit(**kw)
    r = np.array([0.08, 0.05])
    t = np.linspace(0.0, 80.0, 4000)
    lc = P.LimbDarkLightCurve(0.3, 0.2)
    want = lc.get_light_curve(orbit=orbit, r=r, t=t, texp=texp, use_in_transit=window)
    from test_gpu_transit import make_record

    rec = make_record(orbit, r, window=window)
    edges, shift = orbit.kernel_tables()
    ekw = {}
    if texp is not None:
        sdt, sw = P.exposure_stencil(7, 0)
        ekw = dict(texp=texp, stencil_dt=sdt, stencil_w=sw)
    got = P.transit_flux(t, rec, P.get_cl(0.3, 0.2)[None], per_planet=True, window=window,
                         ttv=(edges[None], shift[None]), **ekw)[0]
    assert want.min() < -3e-3
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


def test_shift_cotangent_by_finite_differences():
    kw = ttv_case(missing=False)
    orbit = P.TTVOrbit(**kw)
    r = np.array([0.08, 0.05])
    t = np.linspace(0.0, 80.0, 3000)
    rec = _records(orbit, r)
    edges, shift = orbit.kernel_tables()
    c = P.get_cl(0.3, 0.2)[None]
    sdt, sw = P.exposure_stencil(5, 1)
    ekw = dict(texp=0.05, stencil_dt=sdt, stencil_w=sw)
    g = np.random.default_rng(0).normal(size=(1, t.size))
    _, _, _, gshift = P.transit_flux_vjp(t, rec, c, g, ttv=(edges[None], shift[None]), **ekw)
    assert np.abs(gshift).max() > 1e-3
    for p, k in [(0, 3), (1, 2), (0, 7)]:
        h = 1e-6
        up, dn = shift.copy(), shift.copy()
        up[p, k] += h
        dn[p, k] -= h
        fd = ((P.transit_flux(t, rec, c, ttv=(edges[None], up[None]), **ekw)
               - P.transit_flux(t, rec, c, ttv=(edges[None], dn[None]), **ekw)) * g).sum() / (2 * h)
        assert abs(fd - gshift[0, p, k]) <= 1e-6 * np.abs(gshift).max()


def test_host_tables_match_oracle():
    """exoplanet_amd.orbits.TTVOrbit.kernel_ttv (torch, host side) == the oracle's histogram"""
    from exoplanet_amd.orbits import TTVOrbit

    kw = ttv_case()
    want_e, want_s = P.TTVOrbit(**kw).kernel_tables()
    cpu = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cpu")  # noqa: E731
    tk = {k: ([cpu(x) for x in v] if k == "ttvs" else v if k == "transit_inds" else cpu(v)) for k, v in kw.items()}
    edges, shift = TTVOrbit(**tk).kernel_ttv()
    np.testing.assert_array_equal(edges.numpy(), want_e)
    np.testing.assert_allclose(shift.numpy(), want_s, rtol=0, atol=1e-13)
    # draws: ttvs carry a leading dimension; each draw's table is the unbatched one
    rng = np.random.default_rng(5)
    draws = [np.stack([v + 0.01 * rng.normal(size=v.shape) for _ in range(3)]) for v in kw["ttvs"]]
    tk["ttvs"] = [cpu(x) for x in draws]
    edges, shift = TTVOrbit(**tk).kernel_ttv()
    assert tuple(edges.shape[:2]) == (3, 2) and shift.shape[-1] == edges.shape[-1] + 1
    for d in range(3):
        kd = dict(kw, ttvs=[x[d] for x in draws])
        e, s = P.TTVOrbit(**kd).kernel_tables()
        np.testing.assert_allclose(edges[d].numpy(), e, rtol=0, atol=1e-13)
        np.testing.assert_allclose(shift[d].numpy(), s, rtol=0, atol=1e-13)


def test_transit_times_parameterisation_tables():
    """transit_times given (ttv.py:91-137): least-squares period / t0, batched the same way"""
    from exoplanet_amd.orbits import TTVOrbit

    rng = np.random.default_rng(8)
    expect = expected_times(0.0, 60.0, [7.3, 11.9], [2.0, 5.5])
    times = [e + 0.03 * rng.normal(size=e.size) for e in expect]
    want = P.TTVOrbit(transit_times=times, b=np.array([0.2, 0.4]))
    got = TTVOrbit(transit_times=[torch.as_tensor(x, device="cpu") for x in times],
                   b=torch.tensor([0.2, 0.4], dtype=torch.float64))
    np.testing.assert_allclose(got.period.numpy(), want.period, rtol=1e-14)
    np.testing.assert_allclose(got.t0.numpy(), want.t0, rtol=1e-13)
    e, s = want.kernel_tables()
    ge, gs = got.kernel_ttv()
    np.testing.assert_allclose(ge.numpy(), e, rtol=0, atol=1e-12)
    np.testing.assert_allclose(gs.numpy(), s, rtol=0, atol=1e-12)


def test_ephemeris_shortcut_and_fallback():
    """TTVOrbit reads (t0, period) straight from its arguments when both are given (no attribute
    algebra on the fused path) and derives them otherwise (t_periastron given): same tables"""
    from exoplanet_amd.orbits import TTVOrbit

    cpu = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cpu")  # noqa: E731
    ttvs = [cpu([0.01, -0.02, 0.015, 0.0]), cpu([0.0, 0.03])]
    a = TTVOrbit(period=cpu([3.0, 7.0]), t0=cpu([1.0, 2.0]), b=cpu([0.1, 0.2]), ecc=cpu([0.1, 0.2]),
                 omega=cpu([0.3, -0.4]), ttvs=ttvs)
    assert not a._ready                                        # nothing materialised yet
    ea, sa = a.kernel_ttv()
    assert not a._ready
    b = TTVOrbit(period=cpu([3.0, 7.0]), t_periastron=a.t_periastron, b=cpu([0.1, 0.2]), ecc=cpu([0.1, 0.2]),
                 omega=cpu([0.3, -0.4]), ttvs=ttvs)
    eb, sb = b.kernel_ttv()
    np.testing.assert_allclose(eb.numpy(), ea.numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(sb.numpy(), sa.numpy(), rtol=0, atol=1e-12)
    # the transit times follow t0 + period * index + ttv (ttv.py:141-147)
    np.testing.assert_allclose(a.transit_times[0].numpy(), 1.0 + 3.0 * np.arange(4) + ttvs[0].numpy(), atol=1e-15)


def test_transit_inds_from_host_or_device_tensor():
    from exoplanet_amd.orbits import TTVOrbit

    cpu = lambda x: torch.as_tensor(x, dtype=torch.float64, device="cpu")  # noqa: E731
    kw = dict(period=cpu([3.0]), t0=cpu([1.0]), b=cpu([0.1]), ttvs=[cpu([0.01, -0.02, 0.015])])
    want = TTVOrbit(transit_inds=[[0, 2, 5]], **kw).kernel_ttv()
    for inds in ([np.array([0, 2, 5])], [torch.tensor([0, 2, 5])]):
        got = TTVOrbit(transit_inds=inds, **kw).kernel_ttv()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert want[0].shape[-1] == 7            # six transits (three of them filled in): seven edges


@pytest.mark.parametrize("window,secondary", [(False, False), (True, False), (False, True)])
def test_c_port_with_timing_tables_equals_numpy(window, secondary):
    """oracle/c (the fast checker for full-size GPU tests) == oracle/numpy_port on timing tables"""
    from oracle import c_port as C
    from test_gpu_transit import make_record

    orbit = P.TTVOrbit(**ttv_case())
    r = np.array([0.08, 0.05])
    rec = make_record(orbit, r, sbr=0.4 if secondary else None, window=window)
    edges, shift = (x[None] for x in orbit.kernel_tables())
    c = np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.1, 0.4)])[None] if secondary else P.get_cl(0.3, 0.2)[None]
    t = np.linspace(0.0, 80.0, 3000)
    sdt, sw = P.exposure_stencil(7, 2)
    kw = dict(texp=0.08, stencil_dt=sdt, stencil_w=sw, window=window, secondary=secondary)
    g = np.random.default_rng(0).normal(size=(1, t.size))
    want = P.transit_flux_vjp(t, rec, c, g, ttv=(edges, shift), **kw)
    got = C.transit_ttv(t, rec, c, (edges, shift), g, **kw)
    assert want[0].min() < -3e-3
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-13)
    for a, b in zip(got[1:], want[1:]):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10 * np.abs(b).max())
