"""GPU: nothing the celerite entry points return may depend on what their workspace (the state buffer the reverse pass
re-reads) or their output buffers held BEFORE the call.  An eager call gets those from torch's allocator -- whatever a
freed tensor left there; a captured step replayed as a hipGraph gets the previous replay's contents, so a slot read before
it is written makes a step depend on the step before it (round 5: a sampler run whose odd chains collapsed).  Every buffer
is filled with NaN, with 1e300 and with zeros before the call: log-likelihood and all gradients must be bit-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda:0")


def _terms(xo, dev, D, which, rng):
    T = xo.gp.terms
    v = lambda x, s=0.05: torch.tensor(x * (1 + s * rng.normal(size=D)), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    if which == "real":
        p = [dict(a=v(1e-6), c=v(0.7))]
        return T.RealTerm(**p[0]), list(p[0].values())
    if which == "sampler":      # the C3-model sampler test's kernel: sigma per chain, rho and Q the same for all
        one = torch.ones(D, dtype=torch.float64, device=dev)
        s = v(8e-4, 0.3)
        return T.SHOTerm(sigma=s, rho=1.5 * one, Q=0.7071 * one), [s]
    if which == "mixed":
        q = torch.tensor(np.where(np.arange(D) % 3 == 0, 0.3, 2.0), dtype=torch.float64, device=dev, requires_grad=True)
        p = dict(sigma=v(1e-3), rho=v(3.0), Q=q)
        return T.SHOTerm(**p), list(p.values())
    ps = [dict(sigma=v(4e-4), rho=v(20.0), Q=v(2.0)), dict(sigma=v(3e-4), rho=v(10.0), Q=v(1.0)),
          dict(sigma=v(2e-4), rho=v(2.0), Q=v(0.7071)), dict(sigma=v(2e-4), rho=v(0.7), Q=v(3.0)),
          dict(sigma=v(1e-4), rho=v(0.3), Q=v(1.5))]
    n = {"sho2": 2, "sho3": 3, "sho4": 4, "sho5": 5}[which]       # J = 4, 6, 8 (lane groups of eight), 10 (a DPP row of sixteen: the wide path of round 6)
    kern = T.SHOTerm(**ps[0])
    for p in ps[1:n]:
        kern = kern + T.SHOTerm(**p)
    return kern, [x for p in ps[:n] for x in p.values()]


def _step(xo, dev, D, N, which, route, seed, sigma_scale=None):
    """(loglike, [gradients]) of one value + gradient step of a transit + GP model"""
    from exoplanet_amd import ops

    rng = np.random.default_rng(seed)
    t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0))
    leaf = lambda x, s=1e-3: torch.tensor(x * (1 + s * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev, requires_grad=True)  # noqa: E731
    P = dict(period=leaf(3.5), t0=leaf(1.0), b=leaf(0.3))
    r = leaf(0.1)
    kern, kl = _terms(xo, dev, D, which, rng)
    if sigma_scale is not None:     # some draws far off (flagged: robust route / sequential kernels)
        with torch.no_grad():
            kl[0][: len(sigma_scale)] *= torch.tensor(sigma_scale, dtype=torch.float64, device=dev)
    orbit = xo.orbits.KeplerianOrbit(**P)
    star = xo.LimbDarkLightCurve(0.3, 0.2)
    y = torch.tensor(5e-4 * rng.normal(size=N), dtype=torch.float64, device=dev)
    if route == "dense":
        lc = star.get_light_curve(orbit=orbit, r=r, t=t, total=True)
    elif route == "cm":
        lc = star.get_light_curve(orbit=orbit, r=r, t=t, total=True, cadence_major=True)
    else:
        lc = star.get_light_curve(orbit=orbit, r=r, t=t, total=True, sparse=True)
    yerr = torch.tensor(5e-4 * (1 + 0.1 * rng.random(N)), dtype=torch.float64, device=dev, requires_grad=True)
    gp = xo.gp.GaussianProcess(kern, t=t, yerr=yerr, mean=lc)
    ll = gp.log_likelihood(y)
    leaves = list(P.values()) + [r, yerr] + kl
    g = torch.autograd.grad(ll.sum(), leaves)
    torch.cuda.synchronize()
    ops.release_sorted(t)
    return ll.detach().cpu().numpy(), [x.cpu().numpy() for x in g]


CASES = [("real", "cm", 128, 4000), ("sampler", "cm", 128, 4320), ("sampler", "sparse", 128, 4320), ("sampler", "dense", 100, 3000),
         ("mixed", "cm", 128, 4000), ("mixed", "sparse", 70, 3000), ("sho2", "cm", 64, 3000), ("sho3", "sparse", 64, 3000),
         ("sho4", "cm", 40, 3000), ("sho5", "cm", 16, 1500)]


@pytest.mark.parametrize("which,route,D,N", CASES)
def test_results_do_not_depend_on_what_the_buffers_held(dev, which, route, D, N):
    import exoplanet_amd as xo
    from exoplanet_amd.gp import celerite as C

    out = {}
    try:
        for fill in (0.0, float("nan"), 1e300, -3.0):
            C._POISON[0] = fill
            out[fill if fill == fill else "nan"] = _step(xo, dev, D, N, which, route, seed=5)
    finally:
        C._POISON[0] = None
    ll0, g0 = out[0.0]
    assert np.all(np.isfinite(ll0)) and all(np.all(np.isfinite(g)) for g in g0)
    for key, (ll, g) in out.items():
        assert np.array_equal(ll, ll0), f"log-likelihood depends on the buffers' previous contents (fill {key})"
        for i, (a, b) in enumerate(zip(g, g0)):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            assert bad.size == 0, f"gradient {i} depends on the buffers' previous contents (fill {key}): first at {bad[:4].tolist()}"


@pytest.mark.parametrize("which", ["sampler", "mixed", "sho2"])
def test_flagged_draws_too(dev, which):
    """draws the forward pass flags (eight decades of amplitude apart from their neighbours: the robust route, the
    sequential kernels behind it) next to ordinary ones"""
    import exoplanet_amd as xo
    from exoplanet_amd.gp import celerite as C

    scale = [1e4, 1.0, 1e-4, 1.0, 1e3, 1e3, 1.0, 30.0, 1.0, 1e5]
    out = []
    try:
        for fill in (0.0, float("nan"), 1e300):
            C._POISON[0] = fill
            out.append(_step(xo, dev, 128, 4000, which, "cm", seed=9, sigma_scale=scale))
    finally:
        C._POISON[0] = None
    ll0, g0 = out[0]
    for ll, g in out[1:]:
        assert np.array_equal(ll, ll0, equal_nan=True)
        for i, (a, b) in enumerate(zip(g, g0)):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            assert bad.size == 0, f"gradient {i}: first at {bad[:4].tolist()}"
