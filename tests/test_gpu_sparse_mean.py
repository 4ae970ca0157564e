"""GPU: the light curve as a SPARSE mean of a celerite GP (round 5; exo_sparse_model, exo_celerite_loglike_sparse_*_f64,
exo_transit_flux_vjp_sparse_f64, EXO_FLAG_SPARSE on the Jacobian pair) -- segments of cadences + their values travel
between the two ops, the dense (draw, cadence) model and its cotangent never exist -- against the dense cadence-major route
on the same inputs: the same arithmetic on the same numbers, so the log-likelihood agrees to the last bits (1e-13 stated)
and every gradient to 1e-12 of the largest.  Time-parallel path (J = 1 .. 6, kinds mixed per draw), the lane-group path
(J = 7, 8), sequential kernels, flagged draws, exposure stencils (the Jacobian route), unbounded windows (every cadence a
segment), and the fallbacks for what the sparse form does not take."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda:0")


def _orbit_leaves(dev, D, seed, ecc=0.3, period=3.5, spread=1e-3):
    rng = np.random.default_rng(seed)
    base = dict(period=period, t0=1.0, b=0.3, ecc=ecc, omega=1.1)
    L = {k: torch.tensor(v * (1 + spread * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)
         for k, v in base.items()}
    r = torch.tensor(0.1 * (1 + spread * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)
    u1 = torch.full((D,), 0.3, dtype=torch.float64, device=dev, requires_grad=True)
    u2 = torch.full((D,), 0.2, dtype=torch.float64, device=dev, requires_grad=True)
    return L, r, u1, u2


def _kernel(xo, dev, D, which, seed):
    rng = np.random.default_rng(seed)
    T = xo.gp.terms
    v = lambda x: torch.tensor(x * (1 + 0.05 * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    sho = lambda s, rho, q: dict(sigma=v(s), rho=v(rho), Q=v(q))  # noqa: E731
    if which == "real":       # J = 1: an over-damped... no: a single real term through Matern's limit is J = 2; use RealTerm
        p = dict(a=v(1e-6), c=v(0.7))
        return T.RealTerm(**p), list(p.values())
    if which == "sho":        # J = 2
        p = sho(1e-3, 5.0, 0.7071)
        return T.SHOTerm(**p), list(p.values())
    if which == "mixed":      # J = 2, draws on both sides of Q = 1/2 (pair kinds per draw)
        q = torch.tensor(np.where(np.arange(D) % 3 == 0, 0.3, 2.0), dtype=torch.float64, device=dev, requires_grad=True)
        p = dict(sigma=v(1e-3), rho=v(3.0), Q=q)
        return T.SHOTerm(**p), list(p.values())
    ps = [sho(4e-4, 20.0, 2.0), sho(3e-4, 10.0, 1.0), sho(2e-4, 2.0, 0.7071), sho(2e-4, 0.7, 3.0)]
    n = {"sho2": 2, "sho3": 3, "sho4": 4}[which]     # J = 4, 6, 8
    kern = T.SHOTerm(**ps[0])
    for p in ps[1:n]:
        kern = kern + T.SHOTerm(**p)
    leaves = [x for p in ps[:n] for x in p.values()]
    if which == "sho3r":
        pass
    return kern, leaves


def _both_routes(xo, dev, D, N, which, texp=None, seed=1, yerr=5e-4, cadence=2.0 / 1440.0, ecc=0.3, same_plan=True,
                 use_in_transit=None, extra_real=False):
    """log-likelihood and all leaf gradients through the dense cadence-major mean and through the sparse mean.  same_plan:
    both routes cut the series into the same chunks (the sparse route's default plan has twice the dense one's for J <= 2,
    exo_celerite_default_chunks: other chunks, other roundings -- 1e-13 in the log-likelihood) so that the comparison is of
    the same arithmetic on the same numbers"""
    import os

    from exoplanet_amd import _lib

    saved = os.environ.get("EXO_GP_CHUNKS")
    if same_plan and not saved:
        kern0, _ = _kernel(xo, dev, D, which, seed + 1)
        ar, cr, pairs, _ = kern0.pair_coefficients()
        n_real, n_cplx = ar.shape[-1] + (1 if extra_real else 0), pairs.shape[-2]
        os.environ["EXO_GP_CHUNKS"] = str(int(_lib.load().exo_celerite_default_chunks(N, D, n_real, n_cplx, 1)))
    try:
        return _both_routes_inner(xo, dev, D, N, which, texp, seed, yerr, cadence, ecc, use_in_transit, extra_real)
    finally:
        if same_plan and not saved:
            del os.environ["EXO_GP_CHUNKS"]


def _both_routes_inner(xo, dev, D, N, which, texp, seed, yerr, cadence, ecc, use_in_transit, extra_real):
    t = torch.arange(N, dtype=torch.float64, device=dev) * cadence
    rng = np.random.default_rng(seed + 7)
    y = torch.tensor(yerr * rng.normal(size=N), dtype=torch.float64, device=dev)
    out = {}
    for sparse in (False, True):
        L, r, u1, u2 = _orbit_leaves(dev, D, seed, ecc=ecc)
        kern, kl = _kernel(xo, dev, D, which, seed + 1)
        if extra_real:      # an odd state width: J + 1
            a = torch.full((D,), 1e-7, dtype=torch.float64, device=dev, requires_grad=True)
            c = torch.full((D,), 0.4, dtype=torch.float64, device=dev, requires_grad=True)
            kern = kern + xo.gp.terms.RealTerm(a=a, c=c)
            kl = kl + [a, c]
        orbit = xo.KeplerianOrbit(**L)
        kw = dict(sparse=True) if sparse else dict(cadence_major=True)
        lc = xo.LimbDarkLightCurve(u1, u2).get_light_curve(orbit=orbit, r=r, t=t, texp=texp, total=True,
                                                           use_in_transit=use_in_transit, **kw)
        if sparse:
            assert isinstance(lc, xo.ops.SparseLightCurve), type(lc)
        gp = xo.gp.GaussianProcess(kern, t=t, yerr=yerr, mean=lc)
        ll = gp.log_likelihood(y)
        leaves = list(L.values()) + [r, u1, u2] + kl
        w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=dev)
        grads = torch.autograd.grad((ll * w).sum(), leaves)
        out[sparse] = (ll.detach().clone(), [g.detach().clone() for g in grads], lc)
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
    # the gradients are not trivially zero: the light curve's leaves feel the GP
    assert float(g_s[0].abs().max()) > 0 and float(g_s[5].abs().max()) > 0


@pytest.mark.parametrize("which,D,N", [("sho", 70, 20_011), ("mixed", 66, 9_001), ("real", 5, 6_000), ("sho2", 40, 8_003),
                                       ("sho3", 9, 12_000), ("sho", 600, 30_000)])
def test_sparse_mean_equals_dense_mean(dev, which, D, N):
    import exoplanet_amd as xo

    _compare(_both_routes(xo, dev, D, N, which))


def test_sparse_mean_default_plans(dev):
    """each route on its own default plan (the sparse one cuts a J <= 2 series into twice as many chunks): the time-parallel
    recurrences are exact whatever the chunks -- the results differ by roundings only"""
    import exoplanet_amd as xo

    _compare(_both_routes(xo, dev, 600, 30_000, "sho", same_plan=False), ll_rtol=1e-11, g_rtol=1e-8)
    _compare(_both_routes(xo, dev, 66, 9_001, "mixed", same_plan=False), ll_rtol=1e-11, g_rtol=1e-8)


@pytest.mark.parametrize("which,extra_real", [("sho3", True), ("sho4", False), ("sho4", True)])
def test_sparse_mean_on_the_lane_group_path(dev, which, extra_real):
    """J = 7, 8: a draw on eight lanes, the full factorisation saved (celerite_elem_lg / chunk_fwd / chunk_vjp kernels); J = 9
    (round 6): the same kernels on a DPP row of sixteen lanes, the scans on celerite_tree_wide_kernel"""
    import exoplanet_amd as xo

    _compare(_both_routes(xo, dev, 11, 7_000, which, extra_real=extra_real), g_rtol=1e-11)


def test_sparse_mean_with_exposure_stencil_takes_the_jacobian_route(dev):
    """texp: seven sub-exposures per cadence -- the value sweep leaves the rows of derivatives next to the sparse values and
    the reverse pass is their contraction with the cotangent of the values (EXO_FLAG_SPARSE on the Jacobian pair)"""
    import exoplanet_amd as xo

    before = xo.ops._JAC_CALLS[0]
    out = _both_routes(xo, dev, 33, 9_000, "sho", texp=29.4 / 1440.0, cadence=29.4 / 1440.0)
    assert xo.ops._JAC_CALLS[0] >= before + 2
    _compare(out)


def test_sparse_mean_sequential_kernels_and_short_series(dev, monkeypatch):
    """EXO_GP_CHUNKS=1: the sequential recurrences (celerite_fwd / _vjp kernels) read the segments too; and a series too
    short for the time-parallel plan"""
    import exoplanet_amd as xo

    monkeypatch.setenv("EXO_GP_CHUNKS", "1")
    _compare(_both_routes(xo, dev, 7, 5_000, "sho"))
    _compare(_both_routes(xo, dev, 3, 4_000, "sho3"), g_rtol=1e-11)
    monkeypatch.delenv("EXO_GP_CHUNKS")
    _compare(_both_routes(xo, dev, 4, 50, "sho", cadence=0.05))


def test_sparse_mean_with_flagged_draws(dev):
    """error bars so small that the conditioning score of every draw is beyond the time-parallel path's robust route: the
    draws are redone by the sequential kernels, launched behind the chunk kernels with a mask -- on the sparse model too"""
    import exoplanet_amd as xo

    out = _both_routes(xo, dev, 6, 6_000, "sho", yerr=1e-9)
    _compare(out, ll_rtol=1e-12, g_rtol=1e-10)


def test_sparse_mean_every_cadence_a_segment(dev):
    """more conjunction windows in the series than a run list holds (a 58-minute orbit sampled once a day): the list
    degenerates to "every cadence" -- segments that tile the whole series, every cadence solved"""
    import exoplanet_amd as xo

    N, D = 300, 5
    t = torch.arange(N, dtype=torch.float64, device=dev) * 1.0
    res = {}
    for sparse in (False, True):
        L, r, u1, u2 = _orbit_leaves(dev, D, 3, period=0.04, ecc=0.1)
        orbit = xo.KeplerianOrbit(**L, a=torch.full((D, 1), 3.0, dtype=torch.float64, device=dev))
        kw = dict(sparse=True) if sparse else dict(cadence_major=True)
        lc = xo.LimbDarkLightCurve(u1, u2).get_light_curve(orbit=orbit, r=r, t=t, total=True, **kw)
        kern, kl = _kernel(xo, dev, D, "sho", 5)
        y = torch.tensor(1e-3 * np.random.default_rng(1).normal(size=N), dtype=torch.float64, device=dev)
        ll = xo.gp.GaussianProcess(kern, t=t, yerr=1e-3, mean=lc).log_likelihood(y)
        g = torch.autograd.grad(ll.sum(), list(L.values()) + [r, u1, u2] + kl)
        res[sparse] = (ll.detach().clone(), [x.clone() for x in g], lc)
    sp = res[True][2]
    lay = sp.layout()
    covered = int(torch.gather(lay.pre_all.long(), -1, lay.nrun.long().unsqueeze(-1)).sum().item())
    assert covered == D * N, "expected the every-cadence fallback"
    dense = sp.dense()
    assert int((dense != 0).sum()) >= 3 * D          # (some cadences do fall into a transit)
    _compare(res, g_rtol=1e-11)


def test_sparse_light_curve_dense_and_fallbacks(dev):
    """dense() is the ordinary light curve (values and gradients); what the sparse form does not take -- several planets,
    occultations -- comes back as the cadence-major dense array; a batched y goes through dense()"""
    import exoplanet_amd as xo

    D, N = 12, 9_000
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    L, r, u1, u2 = _orbit_leaves(dev, D, 9)
    star = xo.LimbDarkLightCurve(u1, u2)
    want = star.get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True)
    sp = star.get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True, sparse=True)
    assert isinstance(sp, xo.ops.SparseLightCurve)
    got = sp.dense()
    assert torch.equal(got, want) and bool((got != 0).any())
    g = torch.randn(D, N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    leaves = list(L.values()) + [r, u1, u2]
    ga = torch.autograd.grad((want * g).sum(), leaves)
    gb = torch.autograd.grad((got * g).sum(), leaves)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())
    # a batch of observed series: not the fused form -- the dense array, same numbers as the dense mean
    kern, _ = _kernel(xo, dev, D, "sho", 2)
    yb = torch.tensor(5e-4 * np.random.default_rng(4).normal(size=(D, N)), dtype=torch.float64, device=dev)
    ll_sp = xo.gp.GaussianProcess(kern, t=t, yerr=5e-4, mean=sp).log_likelihood(yb)
    ll_de = xo.gp.GaussianProcess(kern, t=t, yerr=5e-4, mean=want).log_likelihood(yb)
    assert float(((ll_sp - ll_de).abs() / ll_de.abs()).max()) <= 1e-13
    # two planets, occultations: several lists per draw -- merged on the device since round 6 (tests/test_gpu_sparse_merged.py);
    # with the merge switched off they come back as the dense cadence-major array, as before
    L2 = {k: torch.cat([v, v * 1.7], 1).detach().requires_grad_(True) for k, v in L.items()}
    r2 = torch.cat([r, 0.5 * r], 1).detach().requires_grad_(True)
    sec = xo.SecondaryEclipseLightCurve((0.3, 0.2), (0.4, 0.1), torch.full((D,), 0.3, dtype=torch.float64, device=dev))
    lc2 = star.get_light_curve(orbit=xo.KeplerianOrbit(**L2), r=r2, t=t, total=True, sparse=True)
    lc3 = sec.get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True, sparse=True)
    assert isinstance(lc2, xo.ops.MergedSparseLightCurve) and isinstance(lc3, xo.ops.MergedSparseLightCurve)
    assert torch.equal(lc2.dense(), star.get_light_curve(orbit=xo.KeplerianOrbit(**L2), r=r2, t=t, total=True))
    xo.ops._MULTI_LIST_MEAN[0] = False
    try:
        lc2 = star.get_light_curve(orbit=xo.KeplerianOrbit(**L2), r=r2, t=t, total=True, sparse=True)
        lc3 = sec.get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True, sparse=True)
    finally:
        xo.ops._MULTI_LIST_MEAN[0] = True
    assert torch.is_tensor(lc2) and xo.ops.is_cadence_major(lc2)
    assert torch.is_tensor(lc3) and xo.ops.is_cadence_major(lc3)
    # arithmetic on a sparse light curve gives the ordinary tensor
    assert torch.equal(1.0 + sp, 1.0 + want) and torch.equal(sp * 2.0, want * 2.0) and torch.equal(-sp, -want)


def test_sparse_step_replayed_as_a_hip_graph(dev):
    """the sparse C3-shaped step captured and replayed with changing leaves: the replay computes what an eager call does"""
    import exoplanet_amd as xo

    D, N = 64, 12_000
    t = xo.ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0))
    y = torch.tensor(5e-4 * np.random.default_rng(8).normal(size=N), dtype=torch.float64, device=dev)
    L, r, u1, u2 = _orbit_leaves(dev, D, 21)
    kern_l = [torch.full((D,), v, dtype=torch.float64, device=dev, requires_grad=True) for v in (1e-3, 5.0, 0.7071)]
    leaves = list(L.values()) + [r, u1, u2] + kern_l
    names = list(L)

    def step(*vals):
        Lv = dict(zip(names, vals[:5]))
        lc = xo.LimbDarkLightCurve(vals[6], vals[7]).get_light_curve(orbit=xo.KeplerianOrbit(**Lv), r=vals[5], t=t, total=True,
                                                                    sparse=True)
        gp = xo.gp.GaussianProcess(xo.gp.terms.SHOTerm(sigma=vals[8], rho=vals[9], Q=vals[10]), t=t, yerr=5e-4, mean=lc)
        ll = gp.log_likelihood(y)
        return (ll.detach(),) + torch.autograd.grad(ll.sum(), vals)

    eager0 = [x.clone() for x in step(*leaves)]
    static = [x.detach().clone().requires_grad_(True) for x in leaves]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step(*static)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step(*static)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager0, outs):
        assert torch.equal(a, b)
    with torch.no_grad():
        static[0].mul_(1.0003)          # other periods: the transits move, the segments with them
        static[5].mul_(1.01)
    graph.replay()
    torch.cuda.synchronize()
    moved = [x.detach().clone().requires_grad_(True) for x in static]
    eager1 = step(*moved)
    for a, b in zip(eager1, outs):
        assert torch.equal(a, b)
    xo.ops.release_sorted(t)


def test_draw_order_of_the_library_and_results_do_not_depend_on_it(dev):
    """exo_sparse_model_order (one launch) gives the order torch's stable argsort of the same keys gives; and the order in which
    the kernels take the draws (exo_sparse_model.row_of_draw, ABI 12: every per-draw array of the call is indexed through it,
    nothing is permuted by the caller) changes no number: per-draw diag, mixed pair kinds, periods 10 % apart -- sorted, unsorted
    and a random order give bit-identical log-likelihoods and gradients."""
    import exoplanet_amd as xo
    from exoplanet_amd.gp import celerite as C

    D, N = 200, 24_000
    rng = np.random.default_rng(11)
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    y = torch.tensor(5e-4 * rng.normal(size=N), dtype=torch.float64, device=dev)

    def run(order_fn):
        rs = np.random.default_rng(12)
        mk = lambda x, s: torch.tensor(x * (1 + s * rs.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
        L = dict(period=mk(3.0, 0.03), t0=mk(1.0, 0.05), b=mk(0.3, 0.05))
        r = mk(0.1, 0.02)
        kern, kl = _kernel(xo, dev, D, "mixed", 13)
        yerr = torch.tensor(5e-4 * (1 + 0.2 * rs.random((D, N))), dtype=torch.float64, device=dev, requires_grad=True)   # a per-draw diag
        lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(**L), r=r, t=t, total=True, sparse=True)
        assert isinstance(lc, xo.ops.SparseLightCurve)
        saved = C._transit_order
        seen = {}
        def order(sp):
            seen["lib"] = saved(sp)
            return order_fn(sp, seen["lib"])
        C._transit_order = order
        try:
            ll = xo.gp.GaussianProcess(kern, t=t, yerr=yerr, mean=lc).log_likelihood(y)
            w = torch.linspace(0.5, 1.5, D, dtype=torch.float64, device=dev)
            g = torch.autograd.grad((ll * w).sum(), list(L.values()) + [r, yerr] + kl)
        finally:
            C._transit_order = saved
        return ll.detach().clone(), [x.detach().clone() for x in g], seen["lib"], lc

    ll0, g0, lib_order, lc = run(lambda sp, lib: lib)
    # the library's order against torch's stable argsort of the same keys
    lay = lc.layout()
    nrun = lay.nrun.reshape(D).long()
    lo = lay.runs.reshape(D, lay.r_max, 4)[:, :, 0]
    first = lo[:, 0].double()
    last = lo.gather(1, (nrun - 1).clamp_min(0).unsqueeze(1)).squeeze(1).double()
    key = (last - first) / (nrun - 1).clamp_min(1).double() + 1e-9 * first
    assert torch.equal(lib_order.long(), torch.argsort(key, stable=True))
    assert not torch.equal(lib_order.long(), torch.arange(D, device=dev))          # (a real permutation: the periods differ)
    shuffled = torch.tensor(np.random.default_rng(5).permutation(D), dtype=torch.int32, device=dev)
    for fn in (lambda sp, lib: torch.arange(D, dtype=torch.int32, device=dev), lambda sp, lib: shuffled):
        ll1, g1, _, _ = run(fn)
        assert torch.equal(ll1, ll0)
        for a, b in zip(g1, g0):
            assert torch.equal(a, b)


@pytest.mark.parametrize("D", [1, 5, 33, 64, 65, 1000, 1024, 1025, 2048, 4096, 5000])
def test_draw_order_sizes(dev, D):
    """exo_sparse_model_order (a bitonic network over the next power of two) and the torch fallback past its 4096 draws, against a
    stable argsort of the keys; periods in a few groups, so that there are ties to break"""
    import exoplanet_amd as xo
    from exoplanet_amd.gp import celerite as C

    N = 4000
    rng = np.random.default_rng(D)
    t = torch.arange(N, dtype=torch.float64, device=dev) * (10.0 / 1440.0)
    period = torch.tensor(rng.choice([2.5, 2.7, 3.1, 3.3, 40.0], size=(D, 1)), dtype=torch.float64, device=dev)
    t0 = torch.tensor(rng.choice([0.9, 1.0, 1.1], size=(D, 1)), dtype=torch.float64, device=dev)
    b = torch.full((D, 1), 0.3, dtype=torch.float64, device=dev)
    lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(period=period, t0=t0, b=b), r=0.1, t=t, total=True, sparse=True)
    assert isinstance(lc, xo.ops.SparseLightCurve)
    got = C._transit_order(lc)
    lay = lc.layout()
    nrun = lay.nrun.reshape(D).long()
    lo = lay.runs.reshape(D, lay.r_max, 4)[:, :, 0]
    first = lo[:, 0].double()
    last = lo.gather(1, (nrun - 1).clamp_min(0).unsqueeze(1)).squeeze(1).double()
    key = (last - first) / (nrun - 1).clamp_min(1).double() + 1e-9 * first
    assert D < 33 or int((nrun < 2).sum()) > 0            # (the 40-day period: one transit in the series)
    assert torch.equal(got.long(), torch.argsort(key, stable=True))


def test_sorted_draws_on_every_route(dev, monkeypatch):
    """more than 64 draws (the kernels take them through exo_sparse_model.row_of_draw, sorted by period) on the routes behind the
    main one: the sequential kernels alone (EXO_GP_CHUNKS=1), draws flagged on the device and redone by the sequential kernels
    behind the chunk kernels, J = 6 (scan trees on lane groups with their serial top), the lane-group chunk kernels (J = 8)"""
    import exoplanet_amd as xo

    _compare(_both_routes(xo, dev, 70, 6_000, "sho", yerr=1e-9), ll_rtol=1e-12, g_rtol=1e-10)
    _compare(_both_routes(xo, dev, 70, 9_000, "sho3"))
    _compare(_both_routes(xo, dev, 66, 6_000, "sho4"))
    monkeypatch.setenv("EXO_GP_CHUNKS", "1")
    _compare(_both_routes(xo, dev, 70, 5_000, "sho"))
    _compare(_both_routes(xo, dev, 67, 3_000, "mixed"))
