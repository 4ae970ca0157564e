"""GPU: the scan kernel's conservative pre-filters never change a result.
Default path (conjunction windows in mean anomaly, then fp32 classification, fp64 evaluation
of every accepted cadence; single-planet batches take the grouped-draws path) ==
EXO_FLAG_EXACT_SCAN path (fp64 classification) on stress geometries: high
eccentricity, very large and very small a/R, grazing impact parameters, exposure
integration, secondary eclipses, long time baselines."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_transit import make_record

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def random_records(rng, D, Pn):
    rec = np.zeros((D, Pn, P.NPAR))
    for d in range(D):
        period = 10 ** rng.uniform(-0.3, 1.8, Pn)
        ecc = np.where(rng.uniform(size=Pn) < 0.2, 0.0, rng.uniform(0, 0.97, Pn))
        omega = rng.uniform(-np.pi, np.pi, Pn)
        a = 10 ** rng.uniform(0.3, 2.7, Pn)              # a/R from 2 to 500
        b = rng.uniform(0, 1.25, Pn)
        incl_factor = (1 + ecc * np.sin(omega)) / (1 - ecc ** 2)
        cosi = np.clip(incl_factor * b / a, 0, 0.999)
        orbit = P.KeplerianOrbit(period=period, a=a, t0=rng.uniform(0, 5, Pn), incl=np.arccos(cosi), ecc=ecc, omega=omega)
        rec[d] = make_record(orbit, 10 ** rng.uniform(-2, -0.5, Pn), sbr=0.3)[0]
    return rec


@pytest.mark.parametrize("secondary", [False, True])
@pytest.mark.parametrize("texp", [None, 0.02])
@pytest.mark.parametrize("D,Pn", [(24, 3), (53, 1)])   # one planet: classify blocks take four draws each
def test_filter_never_changes_results(dev, secondary, texp, D, Pn):
    from exoplanet_amd import ops

    rng = np.random.default_rng(17 + secondary + 2 * Pn)
    rec = random_records(rng, D, Pn)
    t = np.sort(np.concatenate([np.linspace(0, 60, 30000), 2000 + np.linspace(0, 20, 10000)]))   # |M| up to 1e4 rad
    c = np.repeat(np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None], D, 0)
    c = c if secondary else c[:, :3]
    g = rng.normal(size=(D, t.size))
    kw = {}
    if texp is not None:
        sdt, sw = P.exposure_stencil(5, 1)
        kw = dict(texp=T([texp], dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    base = ops.FLAG_SECONDARY if secondary else 0
    f1, gp1, gl1 = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=base, **kw)
    f2, gp2, gl2 = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev),
                                                  flags=base | ops.FLAG_EXACT_SCAN, **kw)
    assert (f2 != 0).sum().item() > 2000, "stress set has too few in-transit cadences"
    # same fp64 arithmetic for every accepted cadence; the only difference allowed is the
    # last bit: the AGM sweeps of the elliptic integrals exit on a wavefront vote, so the
    # number of (exact, idempotent-to-rounding) extra sweeps depends on a lane's wave-mates
    assert (f1 == 0).eq(f2 == 0).all()
    assert (f1 - f2).abs().max().item() < 2e-15
    # gradients: same terms, possibly summed in a different lane order
    scale = gp2.abs().amax(dim=(0, 1), keepdim=True) + 1e-300
    assert ((gp1 - gp2).abs() / scale).max().item() < 1e-11
    assert torch.allclose(gl1, gl2, rtol=1e-11, atol=1e-12)


def test_filter_matches_oracle_on_extremes(dev):
    """single-draw extremes checked against the C oracle as well"""
    from exoplanet_amd import ops
    from oracle import c_port as C

    t = np.linspace(-2, 12, 20000)
    cases = [dict(period=9.0, a=400.0, t0=1.0, incl=np.arccos(0.5 / 400), ecc=0.9, omega=0.3),
             dict(period=1.1, a=2.2, t0=0.4, incl=1.45, ecc=0.0, omega=0.0),
             dict(period=6.5, a=15.0, t0=2.0, incl=np.arccos(1.09 * (1 - 0.6 ** 2) / (1 + 0.6 * np.sin(1.0)) / 15.0), ecc=0.6, omega=1.0)]
    for okw in cases:
        if okw["ecc"] == 0.0:
            okw = {k: v for k, v in okw.items() if k not in ("ecc", "omega")}
        orbit = P.KeplerianOrbit(**okw)
        rec = make_record(orbit, np.array([0.1]))
        c = P.get_cl(0.3, 0.2)[None]
        want, _, _ = C.transit(t, rec, c)
        got = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev))
        assert want.min() < 0
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-13)


def test_classifier_error_bound(dev):
    """the margin folded into the scan kernel's threshold assumes
    |fp32 position error| <= 8e-4 (Markley starter only): measure it."""
    from exoplanet_amd import _lib

    rng = np.random.default_rng(23)
    n = 2_000_000
    e = np.concatenate([rng.uniform(0, 0.99, n // 2), 1 - 10 ** rng.uniform(-3, 0, n // 4), np.zeros(n // 4)])
    M = np.concatenate([rng.uniform(-np.pi, np.pi, n // 2), rng.uniform(-3e4, 3e4, n // 4),
                        10 ** rng.uniform(-8, 0.5, n // 4) * rng.choice([-1, 1], n // 4)])
    rng.shuffle(M)
    E, _ = P.kepler_E(M, e)
    want_cx, want_sx = np.cos(E) - e, np.sqrt(1 - e * e) * np.sin(E)
    cx = torch.empty(n, dtype=torch.float64, device=dev)
    sx = torch.empty(n, dtype=torch.float64, device=dev)
    lib = _lib.load()
    Mt, et = T(M, dev), T(e, dev)     # keep the inputs alive across the asynchronous launch
    _lib.check(lib.exo_selftest_orbit_pos_f32(Mt.data_ptr(), et.data_ptr(), cx.data_ptr(), sx.data_ptr(),
                                             n, torch.cuda.current_stream().cuda_stream), "selftest")
    torch.cuda.synchronize()
    err = np.maximum(np.abs(cx.cpu().numpy() - want_cx), np.abs(sx.cpu().numpy() - want_sx))
    bound = 8e-4 + 0 * e
    worst = (err / bound).max()
    assert worst < 0.9, f"fp32 position error reaches {worst:.2f} of the assumed bound"


def wide_records(rng, D):
    """single-planet records over the whole range the verdict asked for: e in [0, 0.99], a/R in [1.5, 500],
    r in [1e-3, 1] (log-uniform), b in [0, 1 + r], periastron outside the star"""
    rec = np.zeros((D, 1, P.NPAR))
    d = 0
    while d < D:
        ecc = np.array([0.0 if rng.uniform() < 0.15 else rng.uniform(0, 0.99)])
        a = 10 ** rng.uniform(np.log10(1.5), np.log10(500.0), 1)
        if a[0] * (1 - ecc[0]) < 1.02:
            continue
        r = 10 ** rng.uniform(-3, 0, 1)
        omega = rng.uniform(-np.pi, np.pi, 1)
        b = rng.uniform(0, 1 + r[0], 1)
        incl_factor = (1 + ecc * np.sin(omega)) / (1 - ecc ** 2)
        cosi = np.clip(incl_factor * b / a, 0, 0.9999)
        period = 10 ** rng.uniform(-0.5, 1.5, 1)
        orbit = P.KeplerianOrbit(period=period, a=a, t0=rng.uniform(0, 3, 1), incl=np.arccos(cosi), ecc=ecc, omega=omega)
        rec[d] = make_record(orbit, r, sbr=0.3)[0]
        d += 1
    return rec


@pytest.mark.parametrize("stencil", [False, True])
def test_fp32_classifier_never_discards_over_the_parameter_box(dev, stencil):
    """The fp32 pre-classifier (list path: per-cadence exposure times force it, the run-enumeration path does not use
    one) against the exact fp64 scan AND against the run-enumeration path, 400 random systems over the whole
    parameter box: same cadences non-zero, same values to the last bits, same gradients."""
    from exoplanet_amd import ops

    rng = np.random.default_rng(29 + stencil)
    D = 400
    rec = wide_records(rng, D)
    t = np.sort(np.concatenate([np.linspace(0, 40, 16000), 900 + np.linspace(0, 10, 4000)]))
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    g = rng.normal(size=(D, t.size))
    te = 0.015
    sdt, sw = P.exposure_stencil(3, 0) if stencil else (np.zeros(1), np.ones(1))
    per_cad = dict(texp=T(np.full(t.size, te), dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    scalar = dict(texp=T([te], dev), stencil_dt=T(sdt, dev), stencil_w=T(sw, dev))
    args = (T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
    fast = ops.transit_flux_value_and_vjp(*args, flags=0, **per_cad)
    exact = ops.transit_flux_value_and_vjp(*args, flags=ops.FLAG_EXACT_SCAN, **per_cad)
    runs = ops.transit_flux_value_and_vjp(*args, flags=0, **scalar)
    assert (exact[0] != 0).sum().item() > 20000
    for other in (exact, runs):
        assert (fast[0] == 0).eq(other[0] == 0).all()
        assert float((fast[0] - other[0]).abs().max()) <= 4e-15
        for a, b in zip(fast[1:], other[1:]):
            scale = b.abs().amax(dim=0, keepdim=True) + 1e-300
            assert float(((a - b).abs() / scale).max()) <= 1e-9
