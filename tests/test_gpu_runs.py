"""GPU: the run-enumeration sweep (binary-searched conjunction windows, zero-fill hidden in the heavy
kernel, sparse output) against the list sweep with the exact fp64 scan and the oracle.  The two
paths evaluate an accepted cadence with the same fp64 code; the only difference is which cadences
share a wave, i.e. how many AGM steps the wave votes for: fluxes agree to a few 1e-16."""
import numpy as np
import pytest
import torch

from oracle import numpy_port as P
from test_gpu_transit import make_record

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)


def same_flux(a, b, tol=4e-15):
    """equal up to the wave-vote rounding; exactly zero in the same places"""
    return bool(((a == 0) == (b == 0)).all()) and float((a - b).abs().max()) <= tol


def system(rng, D, planets=1, secondary=False, window=False):
    if planets == 1:
        orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
        r = np.array([0.1])
    else:
        orbit = P.KeplerianOrbit(period=np.array([3.5, 7.9, 13.1]), t0=np.array([1.0, 2.3, 5.1]), b=np.array([0.3, 0.1, 0.5]),
                                 ecc=np.array([0.05, 0.1, 0.2]), omega=np.array([1.1, -0.4, 2.0]))
        r = np.array([0.1, 0.05, 0.07])
    rec = np.repeat(make_record(orbit, r, sbr=0.3 if secondary else None, window=window), D, axis=0)
    for slot in (P.P_ROR, P.P_AOR, P.P_COSI, P.P_TP):
        rec[:, :, slot] *= 1 + 1e-3 * rng.normal(size=rec.shape[:2])
    c = P.get_cl(0.3, 0.2)
    if secondary:
        c = np.concatenate([c, P.get_cl(0.4, 0.1)])
    return rec, np.repeat(c[None], D, 0)


@pytest.mark.parametrize("planets,secondary,texp,per_planet", [(1, False, False, False), (3, False, True, False),
                                                               (1, True, True, True), (3, False, False, True)])
def test_runs_sweep_equals_exact_scan(dev, planets, secondary, texp, per_planet):
    from exoplanet_amd import ops

    rng = np.random.default_rng(31 + planets)
    D, N = 7, 20_011
    t = np.arange(N) * (2.0 / 1440.0) + 0.25
    rec, c = system(rng, D, planets, secondary)
    flags = (ops.FLAG_SECONDARY if secondary else 0) | (ops.FLAG_PER_PLANET if per_planet else 0)
    kw = {}
    if texp:
        dt, w = P.exposure_stencil(5, 1)
        kw = dict(texp=T([0.02], dev), stencil_dt=T(dt, dev), stencil_w=T(w, dev))
    shape = (D, N, planets) if per_planet else (D, N)
    g = rng.normal(size=shape)
    f_ref, gp_ref, gl_ref = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=flags | ops.FLAG_EXACT_SCAN, **kw)
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=flags, **kw)
    assert float(f_ref.min()) < -1e-3
    assert same_flux(f, f_ref)
    for a, b in ((gp, gp_ref), (gl, gl_ref)):
        assert float((a - b).abs().max()) <= 1e-11 * float(b.abs().max())
    # forward-only entry point
    f2 = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=flags, **kw)
    assert same_flux(f2, f_ref)
    # sparse output: the same values at the same cadences, zero elsewhere; same gradients
    sp, gp_s, gl_s, dot = ops.transit_flux_sparse(T(t, dev), T(rec, dev), T(c, dev), gflux=T(g, dev), flags=flags, **kw)
    dense = sp.to_dense(per_planet=per_planet)
    np.testing.assert_allclose(dense, f.cpu().numpy(), rtol=0, atol=4e-15)
    assert np.array_equal(dense == 0, f.cpu().numpy() == 0)
    for x, y in ((gp_s, gp), (gl_s, gl)):
        assert float((x - y).abs().max()) <= 1e-13 * float(y.abs().max())
    np.testing.assert_allclose(dot.cpu().numpy(), (g * f_ref.cpu().numpy()).reshape(D, -1).sum(-1), rtol=1e-10, atol=1e-14)
    assert 0 < sp.n_solved() < 0.2 * D * N * planets * (2 if secondary else 1)


def test_runs_sweep_with_contact_windows(dev):
    """EXO_FLAG_WINDOW: the caller's contact windows decide what is solved (use_in_transit semantics)"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(35)
    D, N = 5, 30_000
    t = np.arange(N) * (2.0 / 1440.0)
    rec, c = system(rng, D, window=True)
    g = rng.normal(size=(D, N))
    a = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=ops.FLAG_WINDOW | ops.FLAG_EXACT_SCAN)
    b = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=ops.FLAG_WINDOW)
    assert same_flux(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert float((x - y).abs().max()) <= 1e-11 * float(x.abs().max())
    want, _, _ = P.transit_flux_vjp(t, rec[:1], c[:1], g[:1])
    np.testing.assert_allclose(b[0][0].cpu().numpy(), want[0], rtol=0, atol=1e-12)


def test_unsorted_times_and_unbounded_windows_fall_back_to_every_cadence(dev):
    from exoplanet_amd import ops

    rng = np.random.default_rng(36)
    D, N = 3, 9_000
    t = np.arange(N) * (2.0 / 1440.0)
    rec, c = system(rng, D)
    g = rng.normal(size=(D, N))
    perm = rng.permutation(N)
    ref = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=ops.FLAG_EXACT_SCAN)
    got = ops.transit_flux_value_and_vjp(T(t[perm], dev), T(rec, dev), T(c, dev), T(g[:, perm], dev))
    assert same_flux(got[0], ref[0][:, torch.as_tensor(perm, device=dev)])
    for x, y in zip(got[1:], ref[1:]):
        assert float((x - y).abs().max()) <= 1e-11 * float(y.abs().max())
    # a / R so small that no window can be bounded (q >= 1), and a NaN record: every cadence is solved
    rec2 = rec.copy()
    rec2[0, 0, P.P_AOR] = 1.05
    rec2[1, 0, P.P_N] = np.nan
    a = ops.transit_flux(T(t, dev), T(rec2, dev), T(c, dev), flags=ops.FLAG_EXACT_SCAN)
    b = ops.transit_flux(T(t, dev), T(rec2, dev), T(c, dev))
    assert same_flux(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
    assert bool(torch.isnan(b[1]).all()) and bool(torch.isfinite(b[2]).all())


def test_short_and_odd_series(dev):
    """series shorter than a window, a single cadence, odd lengths, flux rows at odd offsets"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(37)
    rec, c = system(rng, 4)
    for N in (1, 2, 63, 513):
        t = 0.95 + np.arange(N) * (2.0 / 1440.0)
        g = rng.normal(size=(4, N))
        a = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev), flags=ops.FLAG_EXACT_SCAN)
        b = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(g, dev))
        assert same_flux(a[0], b[0]), N
        for x, y in zip(a[1:], b[1:]):
            assert float((x - y).abs().max()) <= 1e-11 * float(x.abs().max()) + 1e-300


@pytest.mark.parametrize("planets", [1, 2])
def test_folded_finish_equals_the_separate_last_kernel(dev, planets):
    """batches of >= 512 draws: a draw is one block's work and that block finishes it (gradients from its partials,
    values to their cadences); smaller batches share a draw among several blocks and a last kernel finishes it.  The
    same draws through both: the same flux (up to the wave-vote rounding, zero at the same cadences), gradients equal to
    summation order -- dense, per-planet, forward-only, with timing tables."""
    from exoplanet_amd import ops
    from test_gpu_ttv import case_records

    rng = np.random.default_rng(77)
    D, N, k = 640, 3001, 6
    t = np.arange(N) * (30.0 / 1440.0) + 0.1
    rec, c = system(rng, D, 3 if planets == 2 else 1)
    g = rng.normal(size=(D, N))
    sub = slice(100, 100 + k)
    for flags in (0, ops.FLAG_PER_PLANET):
        gg = np.repeat(g[..., None], rec.shape[1], -1) if flags else g
        big = ops.transit_flux_value_and_vjp(T(t, dev), T(rec, dev), T(c, dev), T(gg, dev), flags=flags)
        small = ops.transit_flux_value_and_vjp(T(t, dev), T(rec[sub], dev), T(c[sub], dev), T(gg[sub], dev), flags=flags)
        assert float(big[0].min()) < -1e-3
        assert same_flux(big[0][sub], small[0])
        for a, b in zip(big[1:], small[1:]):
            assert float((a[sub] - b).abs().max()) <= 1e-12 * float(b.abs().max())
        f_big = ops.transit_flux(T(t, dev), T(rec, dev), T(c, dev), flags=flags)
        assert torch.equal(f_big, big[0])
    # timing tables (two planets, their own tables per draw)
    recs, tables = case_records(draws=8)
    reps = 80
    rec_t = np.tile(recs, (reps, 1, 1))
    ed, sh = np.tile(tables[0], (reps, 1, 1)), np.tile(tables[1], (reps, 1, 1))
    Dt = rec_t.shape[0]
    ct = np.repeat(P.get_cl(0.3, 0.2)[None], Dt, 0)
    tt = np.linspace(-3.0, 84.0, 4001)
    gt = rng.normal(size=(Dt, tt.size))
    big = ops.transit_flux_value_and_vjp(T(tt, dev), T(rec_t, dev), T(ct, dev), T(gt, dev), ttv=(T(ed, dev), T(sh, dev)))
    small = ops.transit_flux_value_and_vjp(T(tt, dev), T(rec_t[:8], dev), T(ct[:8], dev), T(gt[:8], dev),
                                           ttv=(T(ed[:8], dev), T(sh[:8], dev)))
    assert Dt >= 512 and float(big[0].min()) < -1e-3
    assert same_flux(big[0][:8], small[0])
    for a, b in zip(big[1:], small[1:]):
        assert float((a[:8] - b).abs().max()) <= 1e-12 * float(b.abs().max())


@pytest.mark.parametrize("planets,secondary,window", [(1, False, False), (3, False, False), (1, True, False), (3, False, True)])
def test_windows_and_runs_in_one_launch(dev, planets, secondary, window, monkeypatch):
    """EXO_FLAG_SORTED_TIMES (set by the torch layer once it has looked at the time array): the enumeration kernel works the
    windows out itself -- the same runs, the same values, the same gradients as with the device's own check and the
    window kernel in front, bit for bit; an unsorted array never gets the flag"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(7 + planets)
    D, N = 9, 20_011
    t = T(np.arange(N) * (2.0 / 1440.0) + 0.25, dev)
    rec, c = system(rng, D, planets, secondary, window=window)
    flags = (ops.FLAG_SECONDARY if secondary else 0) | (ops.FLAG_WINDOW if window else 0)
    dt, w = P.exposure_stencil(5, 1)
    kw = dict(texp=T([0.02], dev), stencil_dt=T(dt, dev), stencil_w=T(w, dev))
    g = T(rng.normal(size=(D, N)), dev)
    out = {}
    for on_device in ("1", "0"):
        monkeypatch.setenv("EXO_CHECK_SORTED_ON_DEVICE", on_device)
        assert ops._sorted_flag(t) == (0 if on_device == "1" else ops.FLAG_SORTED_TIMES)
        sp, gp, gl, dot = ops.transit_flux_sparse(t, T(rec, dev), T(c, dev), gflux=g, flags=flags, **kw)
        f, gp2, gl2 = ops.transit_flux_value_and_vjp(t, T(rec, dev), T(c, dev), g, flags=flags, **kw)
        out[on_device] = (sp.to_dense(), sp.n_solved(), gp.clone(), gl.clone(), dot.clone(), f.clone(), gp2.clone())
    a, b = out["1"], out["0"]
    assert a[1] == b[1] and np.array_equal(a[0], b[0])
    for x, y in zip(a[2:], b[2:]):
        assert torch.equal(x, y)
    monkeypatch.setenv("EXO_CHECK_SORTED_ON_DEVICE", "0")
    tu = t.clone()
    tu[100], tu[101] = t[101], t[100]
    assert ops._sorted_flag(tu) == 0


def test_a_nan_among_the_times_never_gets_the_sorted_flag(dev):
    """the torch layer's look at the time array says "unordered" for a NaN, as the device's own check does: the sweep then
    solves every cadence, and the finite cadences keep their flux"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(3)
    D, N = 4, 5003
    tt = np.arange(N) * (2.0 / 1440.0) + 0.25
    rec, c = system(rng, D)
    f_ref = ops.transit_flux(T(tt, dev), T(rec, dev), T(c, dev))
    tn = tt.copy()
    tn[777] = np.nan
    t = T(tn, dev)
    assert ops._sorted_flag(t) == 0 and ops.known_sorted(t) and not ops.known_sorted(t, nan_ok=False)
    f = ops.transit_flux(t, T(rec, dev), T(c, dev))
    keep = np.ones(N, bool)
    keep[777] = False
    assert same_flux(f[:, keep], f_ref[:, keep])
    assert float(f_ref.min()) < -1e-3


def test_sorted_flag_policy_and_the_device_guard(dev):
    """(i) inside a hipGraph capture the flag is only set for a series the caller vouched for -- a captured
    step whose time buffer is REFILLED with an unsorted series before a replay still gives the right light curve (the
    device's own check runs at every replay); (ii) with the flag forced on an unordered series, the launch's own coarse
    look (65 evenly spaced cadences) sends the sweep to "every cadence": right results, not undefined ones; (iii) a
    tensor without a version counter never gets the flag; (iv) unknown flag bits are refused"""
    import exoplanet_amd as xo
    from exoplanet_amd import ops, _lib

    rng = np.random.default_rng(11)
    D, N = 6, 30_000
    tt = np.arange(N) * (2.0 / 1440.0) + 0.25
    perm = rng.permutation(N)
    rec, c = system(rng, D)
    recd, cd = T(rec, dev), T(c, dev)
    want = ops.transit_flux(T(tt, dev), recd, cd, flags=ops.FLAG_EXACT_SCAN)       # list path: no searches at all
    assert float(want.min()) < -1e-3
    # (i) capture on a sorted buffer nobody vouched for, refill it with a permutation, replay
    tbuf = T(tt, dev)
    assert ops._sorted_flag(tbuf) == ops.FLAG_SORTED_TIMES          # eager: looked at, remembered
    step = xo.GraphedStep(lambda r: ops.transit_flux(tbuf, r, cd), recd)
    assert same_flux(step().clone(), want)
    tbuf.data.copy_(T(tt[perm], dev))                               # behind the version counter's back
    got = step().clone()
    assert same_flux(got, want[:, torch.as_tensor(perm, device=dev)])
    # a vouched-for series carries the flag into a capture
    tv = ops.vouch_sorted(T(tt, dev))
    seen = []
    g2 = xo.GraphedStep(lambda r: (seen.append(ops._sorted_flag(tv)), ops.transit_flux(tv, r, cd))[1], recd)
    assert seen[-1] == ops.FLAG_SORTED_TIMES and same_flux(g2().clone(), want)
    ops.release_sorted(tv)
    with pytest.raises(ValueError):
        ops.vouch_sorted(T(tt[perm], dev))
    # (ii) the flag forced onto an unordered series: the launch's own look at it degrades the sweep, results stay right
    tp = T(tt[perm], dev)
    assert ops._sorted_flag(tp) == 0
    forced = ops.transit_flux(tp, recd, cd, flags=ops.FLAG_SORTED_TIMES)
    assert same_flux(forced, want[:, torch.as_tensor(perm, device=dev)])
    # (iii) inference tensors have no version counter
    with torch.inference_mode():
        ti = T(tt, dev) + 0.0
    assert ops._sorted_flag(ti) == 0
    # (iv) a flag bit this build does not know
    lib = _lib.load()
    flux = torch.empty(D, N, dtype=torch.float64, device=dev)
    nbytes = lib.exo_transit_flux_workspace_bytes(N, D, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = lib.exo_transit_flux_fwd_f64(tbuf.data_ptr(), N, 0, 0, 0, 0, 1, recd.data_ptr(), cd.data_ptr(), D, 1, 1 << 12,
                                      flux.data_ptr(), ws.data_ptr(), nbytes, 0)
    assert rc == 1      # EXO_ERR_INVALID_ARGUMENT
    assert lib.exo_abi_version() == _lib.ABI_VERSION
