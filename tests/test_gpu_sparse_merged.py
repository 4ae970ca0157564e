"""GPU: the MERGED sparse model (round 6; exo_sparse_model_merge_f64 / _merged / _merge_vjp_f64, ops.MergedSparseLightCurve) --
several lists per draw (planets whose transits overlap, transit + occultation) merged on the device into one ascending list of
disjoint segments with summed values, as the mean of a celerite GP -- against the dense cadence-major route on the same
inputs: the same arithmetic on the same numbers (log-likelihood to 1e-13, gradients to 1e-12 of the largest), the merged
values against the dense light curve bit for bit, the segments against a host-side union of the runs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _T(x, dev, grad=True):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=torch.float64, device=dev, requires_grad=grad)


def _system(dev, D, kind, seed):
    """leaves of a batch of D draws: `two` = two planets whose transits overlap now and then (periods 3.5 and 5.25: every third
    transit of the first meets every second of the other), `sec` = one planet with occultations, `two_sec` = both"""
    rng = np.random.default_rng(seed)
    j = lambda v: np.asarray(v) * (1 + 1e-3 * rng.normal(size=(D, len(np.atleast_1d(v)))))  # noqa: E731
    if kind in ("two", "two_sec", "three"):
        per = [3.5, 5.25] if kind != "three" else [3.5, 5.25, 1.75]
        n = len(per)
        L = dict(period=_T(j(per), dev), t0=_T(j([1.0] * n), dev), b=_T(j([0.3, 0.5, 0.1][:n]), dev),
                 ecc=_T(j([0.2, 0.1, 0.05][:n]), dev), omega=_T(j([1.1, -0.4, 0.6][:n]), dev))
        r = _T(j([0.1, 0.06, 0.04][:n]), dev)
    else:
        L = dict(period=_T(j([2.7]), dev), t0=_T(j([0.4]), dev), b=_T(j([0.2]), dev), ecc=_T(j([0.1]), dev), omega=_T(j([0.7]), dev))
        r = _T(j([0.08]), dev)
    u1 = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
    u2 = torch.full((D,), 0.2, dtype=torch.float64, device=dev, requires_grad=True)
    sbr = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True) if "sec" in kind else None
    return L, r, u1, u2, sbr


def _light_curve(xo, L, r, u1, u2, sbr, t, texp, **kw):
    orbit = xo.KeplerianOrbit(**L)
    if sbr is not None:
        star = xo.SecondaryEclipseLightCurve((u1, u2), (0.4, 0.1), sbr)
    else:
        star = xo.LimbDarkLightCurve(u1, u2)
    return star.get_light_curve(orbit=orbit, r=r, t=t, texp=texp, total=True, **kw)


def _kernel(xo, dev, D, n_terms, seed):
    rng = np.random.default_rng(seed)
    v = lambda x: torch.tensor(x * (1 + 0.05 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    ps = [dict(sigma=v(4e-4), rho=v(20.0), Q=v(2.0)), dict(sigma=v(3e-4), rho=v(10.0), Q=v(1.0)),
          dict(sigma=v(2e-4), rho=v(2.0), Q=v(0.7071)), dict(sigma=v(2e-4), rho=v(0.7), Q=v(3.0))][:n_terms]
    kern = xo.gp.terms.SHOTerm(**ps[0])
    for p in ps[1:]:
        kern = kern + xo.gp.terms.SHOTerm(**p)
    return kern, [x for p in ps for x in p.values()]


def _both_routes(xo, dev, D, N, kind, n_terms, texp=None, cadence=2.0 / 1440.0, seed=1, yerr=5e-4):
    from exoplanet_amd import _lib

    t = torch.arange(N, dtype=torch.float64, device=dev) * cadence
    y = torch.tensor(yerr * np.random.default_rng(seed + 7).normal(size=N), dtype=torch.float64, device=dev)
    saved = os.environ.get("EXO_GP_CHUNKS")
    if not saved:      # the same chunk plan on both routes: the same arithmetic on the same numbers
        os.environ["EXO_GP_CHUNKS"] = str(int(_lib.load().exo_celerite_default_chunks(N, D, 0, n_terms, 1)))
    out = {}
    try:
        for sparse in (False, True):
            L, r, u1, u2, sbr = _system(dev, D, kind, seed)
            kern, kl = _kernel(xo, dev, D, n_terms, seed + 1)
            lc = _light_curve(xo, L, r, u1, u2, sbr, t, texp, **(dict(sparse=True) if sparse else dict(cadence_major=True)))
            if sparse:
                assert isinstance(lc, xo.ops.MergedSparseLightCurve), type(lc)
            else:
                assert torch.is_tensor(lc)
            ll = xo.gp.GaussianProcess(kern, t=t, yerr=yerr, mean=lc).log_likelihood(y)
            leaves = list(L.values()) + [r, u1, u2] + ([sbr] if sbr is not None else []) + kl
            w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=dev)
            grads = torch.autograd.grad((ll * w).sum(), leaves)
            out[sparse] = (ll.detach().clone(), [g.detach().clone() for g in grads], lc)
    finally:
        if not saved:
            del os.environ["EXO_GP_CHUNKS"]
    return out


def _compare(out, ll_rtol=1e-13, g_rtol=1e-12):
    ll_d, g_d, _ = out[False]
    ll_s, g_s, _ = out[True]
    assert torch.isfinite(ll_d).all() and torch.isfinite(ll_s).all()
    assert float(((ll_s - ll_d).abs() / ll_d.abs()).max()) <= ll_rtol
    for a, b in zip(g_d, g_s):
        assert torch.isfinite(b).all()
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= g_rtol * scale + 1e-300, (float((a - b).abs().max()), scale)
    assert float(g_s[0].abs().max()) > 0 and float(g_s[5].abs().max()) > 0


@pytest.mark.parametrize("kind,n_terms,D,N,texp", [
    ("two", 1, 70, 20_011, None),                 # J = 2: overlapping transits of two planets
    ("three", 2, 33, 12_000, None),               # J = 4, three planets
    ("sec", 3, 24, 9_000, None),                  # J = 6: transit + occultation
    ("sec", 3, 16, 8_000, 29.4 / 1440.0),         # ... with the exposure stencil (the Jacobian route): the C5 shape
    ("two_sec", 1, 20, 10_000, None),             # four lists per draw
    ("two", 4, 9, 7_000, None),                   # J = 8: the lane-group path
])
def test_merged_sparse_mean_equals_dense_mean(dev, kind, n_terms, D, N, texp):
    import exoplanet_amd as xo

    cadence = 29.4 / 1440.0 if texp else 2.0 / 1440.0
    _compare(_both_routes(xo, dev, D, N, kind, n_terms, texp=texp, cadence=cadence), g_rtol=1e-11 if n_terms == 4 else 1e-12)


def _host_union(lay, d, P, n_ev):
    """segments of draw d from the sweep's runs: the union of the intervals (touching ones stay separate)"""
    runs = []
    for p in range(P):
        for ev in range(n_ev):
            K = int(lay.nrun[d, p, ev])
            r = lay.runs[d, p, ev, :K].cpu().numpy()
            runs += [(int(a[0]), int(a[3])) for a in r if a[3] > a[0]]
    runs.sort()
    segs, end = [], None
    for lo, hi in runs:
        if end is None or lo >= end:
            segs.append([lo, hi])
        else:
            segs[-1][1] = max(segs[-1][1], hi)
        end = segs[-1][1]
    return segs


@pytest.mark.parametrize("kind", ["two", "two_sec", "three"])
def test_merged_segments_and_values(dev, kind):
    """the merged segments are the union of the lists' runs; the merged values, laid out as a dense array, are the dense light
    curve bit for bit (a cadence's planets summed in planet order, as the dense sweep does); dense() is differentiable"""
    import exoplanet_amd as xo

    D, N = 7, 30_000
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    L, r, u1, u2, sbr = _system(dev, D, kind, 5)
    want = _light_curve(xo, L, r, u1, u2, sbr, t, None)
    sp = _light_curve(xo, L, r, u1, u2, sbr, t, None, sparse=True)
    assert isinstance(sp, xo.ops.MergedSparseLightCurve)
    got = sp.dense()
    assert torch.equal(got, want) and bool((got != 0).any())
    nseg, seg, off = sp.segments()
    # the UNMERGED sparse output of the same sweep (what _fused hands to the merge): its lists, for the host-side union
    sec = None if sbr is None else ((torch.as_tensor(0.4, device=dev, dtype=torch.float64),
                                     torch.as_tensor(0.1, device=dev, dtype=torch.float64)), sbr)
    rec, ld, _, flags = xo.KeplerianOrbit(**L).kernel_inputs(r, (u1, u2), use_in_transit=True, secondary=sec)
    box = []
    vals = xo.ops._TransitFluxSparse.apply(t, None, None, None, rec.detach(), ld.detach(), int(flags), box, False)
    P = r.shape[1]
    n_ev = 2 if sbr is not None else 1
    lay = xo.ops.SparseLightCurve(vals, box[0], N, D, P, int(flags) | xo.ops.FLAG_SPARSE).layout()
    n_over = 0
    for d in range(D):
        want_segs = _host_union(lay, d, P, n_ev)
        S = int(nseg[d])
        got_segs = seg[d, :S].cpu().numpy().tolist()
        assert got_segs == want_segs, (d, got_segs[:5], want_segs[:5])
        lens = np.array([b - a for a, b in want_segs])
        assert off[d, :S + 1].cpu().numpy().tolist() == [0] + np.cumsum(lens).tolist()
        n_runs = sum(int(lay.nrun[d, p, ev]) for p in range(P) for ev in range(n_ev))
        n_over += n_runs - S
    assert n_over > 0, "the system was meant to have overlapping windows"
    g = torch.randn(D, N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    leaves = list(L.values()) + [r, u1, u2] + ([sbr] if sbr is not None else [])
    ga = torch.autograd.grad((want * g).sum(), leaves)
    gb = torch.autograd.grad((got * g).sum(), leaves)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())


def test_merged_model_with_unbounded_windows(dev):
    """a list that degenerates to "every cadence" (more conjunction windows than a run list holds) next to an ordinary one: the
    merged model is one segment per run of the whole-series list, the other planet's values added inside"""
    import exoplanet_amd as xo

    N, D = 300, 4
    t = torch.arange(N, dtype=torch.float64, device=dev) * 1.0
    res = {}
    for sparse in (False, True):
        rng = np.random.default_rng(3)
        j = lambda v: np.asarray(v) * (1 + 1e-3 * rng.normal(size=(D, 2)))  # noqa: E731
        L = dict(period=_T(j([0.04, 37.0]), dev), t0=_T(j([1.0, 5.0]), dev), b=_T(j([0.3, 0.2]), dev), ecc=_T(j([0.1, 0.1]), dev),
                 omega=_T(j([1.1, 0.3]), dev), a=_T(j([3.0, 40.0]), dev))
        r = _T(j([0.1, 0.08]), dev)
        u1 = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
        u2 = torch.full((D,), 0.2, dtype=torch.float64, device=dev, requires_grad=True)
        lc = xo.LimbDarkLightCurve(u1, u2).get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True,
                                                           **(dict(sparse=True) if sparse else dict(cadence_major=True)))
        kern, kl = _kernel(xo, dev, D, 1, 5)
        y = torch.tensor(1e-3 * np.random.default_rng(1).normal(size=N), dtype=torch.float64, device=dev)
        ll = xo.gp.GaussianProcess(kern, t=t, yerr=1e-3, mean=lc).log_likelihood(y)
        g = torch.autograd.grad(ll.sum(), list(L.values()) + [r, u1, u2] + kl)
        res[sparse] = (ll.detach().clone(), [x.clone() for x in g], lc)
    sp = res[True][2]
    assert isinstance(sp, xo.ops.MergedSparseLightCurve)
    nseg, seg, off = sp.segments()
    covered = int(sum(int(off[d, int(nseg[d])]) for d in range(D)))
    assert covered == D * N, "expected the every-cadence list to cover the series"
    _compare(res, g_rtol=1e-11)


def test_merged_step_replayed_as_a_hip_graph(dev):
    """the C5-shaped step (occultations, exposure stencil, J = 6) on the merged sparse mean, captured and replayed with
    changing leaves: a replay computes what an eager call does"""
    import exoplanet_amd as xo

    D, N = 32, 6_000
    texp = 29.4 / 1440.0
    t = xo.ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * texp)
    y = torch.tensor(3e-4 * np.random.default_rng(8).normal(size=N), dtype=torch.float64, device=dev)
    L, r, u1, u2, sbr = _system(dev, D, "sec", 11)
    kern_leaves = _kernel(xo, dev, D, 3, 12)[1]
    leaves = list(L.values()) + [r, u1, u2, sbr] + kern_leaves
    names = list(L)

    def fn(*vals):
        Lv = dict(zip(names, vals[:5]))
        r_, u1_, u2_, sbr_ = vals[5:9]
        ks = vals[9:]
        T = xo.gp.terms
        kern = (T.SHOTerm(sigma=ks[0], rho=ks[1], Q=ks[2]) + T.SHOTerm(sigma=ks[3], rho=ks[4], Q=ks[5])
                + T.SHOTerm(sigma=ks[6], rho=ks[7], Q=ks[8]))
        lc = xo.SecondaryEclipseLightCurve((u1_, u2_), (0.4, 0.1), sbr_).get_light_curve(
            orbit=xo.KeplerianOrbit(**Lv), r=r_, t=t, texp=texp, total=True, sparse=True)
        assert isinstance(lc, xo.ops.MergedSparseLightCurve)
        ll = xo.gp.GaussianProcess(kern, t=t, yerr=3e-4, mean=lc).log_likelihood(y)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    graph = xo.GraphedStep(fn, *leaves)
    for step in range(3):
        with torch.no_grad():
            leaves[0].mul_(1.0 + 1e-4 * (step + 1))      # the period moves: other runs, other segments
            leaves[5].mul_(1.0 + 1e-3)
        got = [x.clone() for x in graph()]
        want = fn(*leaves)
        torch.cuda.synchronize()
        assert float(((got[0] - want[0]).abs() / want[0].abs()).max()) <= 1e-13
        for a, b in zip(got[1:], want[1:]):
            assert float((a - b).abs().max()) <= 1e-11 * float(b.abs().max()) + 1e-300


def test_merged_model_without_any_run(dev):
    """two planets that never cross the disk inside the series (first transits beyond its end): no run in any list -- zero
    segments per draw, the likelihood that of the observed series alone, zero gradients for the orbits, the same as the dense route"""
    import exoplanet_amd as xo

    N, D = 4_000, 6
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)          # 5.6 days
    y = torch.tensor(5e-4 * np.random.default_rng(2).normal(size=N), dtype=torch.float64, device=dev)
    res = {}
    for sparse in (False, True):
        rng = np.random.default_rng(9)
        j = lambda v: np.asarray(v) * (1 + 1e-3 * rng.normal(size=(D, 2)))  # noqa: E731
        L = dict(period=_T(j([50.0, 80.0]), dev), t0=_T(j([20.0, 33.0]), dev), b=_T(j([0.3, 0.2]), dev))
        r = _T(j([0.1, 0.08]), dev)
        u1 = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
        u2 = torch.full((D,), 0.2, dtype=torch.float64, device=dev, requires_grad=True)
        lc = xo.LimbDarkLightCurve(u1, u2).get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True,
                                                           **(dict(sparse=True) if sparse else dict(cadence_major=True)))
        kern, kl = _kernel(xo, dev, D, 1, 5)
        ll = xo.gp.GaussianProcess(kern, t=t, yerr=5e-4, mean=lc).log_likelihood(y)
        g = torch.autograd.grad(ll.sum(), list(L.values()) + [r] + kl, allow_unused=True)
        res[sparse] = (ll.detach().clone(), g, lc)
    sp = res[True][2]
    assert isinstance(sp, xo.ops.MergedSparseLightCurve)
    nseg, _, off = sp.segments()
    assert int(nseg.abs().sum()) == 0 and int(off[:, 0].abs().sum()) == 0
    assert float((sp.dense() != 0).sum()) == 0
    assert float(((res[True][0] - res[False][0]).abs() / res[False][0].abs()).max()) <= 1e-13
    for a, b in zip(res[False][1], res[True][1]):
        if a is None or b is None:
            continue
        assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max()) + 1e-300


def test_merged_route_does_not_depend_on_what_its_buffers_held(dev):
    """the merge workspace (segments, offsets, merged values), the cotangent handed back to the lists and every celerite buffer
    filled with zeros / NaN / 1e300 / -3 before the call: log-likelihood and all gradients bit-identical (a replayed hipGraph
    hands every step the previous step's contents)"""
    import exoplanet_amd as xo
    from exoplanet_amd.gp import celerite as C

    D, N = 24, 6_000
    texp = 29.4 / 1440.0
    t = torch.arange(N, dtype=torch.float64, device=dev) * texp
    y = torch.tensor(3e-4 * np.random.default_rng(4).normal(size=N), dtype=torch.float64, device=dev)
    out = []
    try:
        for fill in (0.0, float("nan"), 1e300, -3.0):
            C._POISON[0] = fill
            xo.ops._POISON[0] = fill
            L, r, u1, u2, sbr = _system(dev, D, "two_sec", 21)
            kern, kl = _kernel(xo, dev, D, 3, 22)
            lc = _light_curve(xo, L, r, u1, u2, sbr, t, texp, sparse=True)
            assert isinstance(lc, xo.ops.MergedSparseLightCurve)
            ll = xo.gp.GaussianProcess(kern, t=t, yerr=3e-4, mean=lc).log_likelihood(y)
            g = torch.autograd.grad(ll.sum(), list(L.values()) + [r, u1, u2, sbr] + kl)
            torch.cuda.synchronize()
            out.append((ll.detach().clone(), [x.clone() for x in g]))
    finally:
        C._POISON[0] = None
        xo.ops._POISON[0] = None
    assert torch.isfinite(out[0][0]).all()
    for ll, g in out[1:]:
        assert torch.equal(ll, out[0][0])
        for a, b in zip(g, out[0][1]):
            assert torch.equal(a, b)
