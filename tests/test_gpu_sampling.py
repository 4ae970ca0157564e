"""GPU: the HMC driver on the real hot path -- C2-like transit likelihood of a batch of chains --
with the leapfrog trajectory replayed as one hipGraph: graph replay == eager trajectory, and the
chains move towards the parameters the data were generated with."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hmc_on_the_transit_likelihood_graph_vs_eager(dev):
    import exoplanet_amd as xo
    from exoplanet_amd.sampling import HMC

    rng = np.random.default_rng(51)
    N, D = 8000, 16
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    truth = dict(period=3.5, t0=1.0, b=0.3, r=0.1)
    with torch.no_grad():
        f0 = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(
            orbit=xo.KeplerianOrbit(period=torch.tensor(truth["period"], device=dev, dtype=torch.float64),
                                    t0=torch.tensor(truth["t0"], device=dev, dtype=torch.float64),
                                    b=torch.tensor(truth["b"], device=dev, dtype=torch.float64)),
            r=torch.tensor(truth["r"], device=dev, dtype=torch.float64), t=t)[:, 0]
    sigma = 5e-4
    y = f0 + sigma * torch.as_tensor(rng.normal(size=N), device=dev)

    def logp(t0, r):          # (D, 1) each: white-noise likelihood of every chain's light curve
        orbit = xo.KeplerianOrbit(period=3.5, t0=t0, b=0.3)
        f = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t).sum(-1)
        return -0.5 * (((y - f) / sigma) ** 2).sum(-1)

    def start():
        g = np.random.default_rng(52)
        return [torch.tensor(1.0 + 2e-3 * g.normal(size=(D, 1)), dtype=torch.float64, device=dev),
                torch.tensor(0.1 * (1 + 0.05 * g.normal(size=(D, 1))), dtype=torch.float64, device=dev)]

    mass = [1.0 / (2e-4) ** 2, 1.0 / (5e-4) ** 2]
    out = {}
    for mode in (True, False):
        gen = torch.Generator(device=dev).manual_seed(9)
        hmc = HMC(logp, start(), step_size=0.25, n_leapfrog=4, mass=mass, graph=mode, generator=gen)
        assert (hmc._graph is not None) == mode
        lp_first = None
        for it in range(25):
            hmc.step()
            if it == 0:
                lp_first = hmc.last_logp.clone()
        out[mode] = (hmc.params[0].clone(), hmc.params[1].clone(), hmc.last_logp.clone(), hmc.accept_rate().clone(), lp_first)
    for a, b in zip(out[True], out[False]):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-12)
    t0s, rs, lp, rate, lp_first = out[True]
    assert float(rate.mean()) > 0.5
    assert float(lp.mean()) > float(lp_first.mean())                     # the chains climb
    # (25 short trajectories from an over-dispersed start: the chains are still drifting in -- this test is about graph ==
    # eager; WHAT the samplers sample is held to closed-form moments in the three tests at the end of this file)
    assert abs(float(t0s.mean()) - truth["t0"]) < 1e-3 and abs(float(rs.mean()) - truth["r"]) < 8e-3


def test_nuts_on_the_white_noise_likelihood_graph_vs_eager(dev):
    """NUTS over the fused white-noise likelihood (exo_transit_chi2_vjp_f64): the leapfrog step replayed as a hipGraph
    gives the chains the eager step gives, and they settle on the generating parameters"""
    import exoplanet_amd as xo
    from exoplanet_amd.sampling import NUTS

    rng = np.random.default_rng(61)
    N, D = 8000, 24
    t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
    T = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)   # noqa: E731
    with torch.no_grad():
        f0 = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=xo.KeplerianOrbit(period=T(3.5), t0=T(1.0), b=T(0.3)),
                                                             r=T(0.1), t=t)[:, 0]
    sigma = 5e-4
    y = f0 + sigma * torch.as_tensor(rng.normal(size=N), device=dev)
    lc = xo.LimbDarkLightCurve(0.3, 0.2)

    def logp(t0, r):
        orbit = xo.KeplerianOrbit(period=3.5, t0=t0, b=0.3)
        return lc.white_noise_log_likelihood(orbit=orbit, r=r, t=t, y=y, yerr=sigma)

    def start():
        g = np.random.default_rng(62)
        return [torch.tensor(1.0 + 2e-3 * g.normal(size=(D, 1)), dtype=torch.float64, device=dev),
                torch.tensor(0.1 * (1 + 0.05 * g.normal(size=(D, 1))), dtype=torch.float64, device=dev)]

    mass = [1.0 / (2e-4) ** 2, 1.0 / (5e-4) ** 2]
    out = {}
    for mode in (True, False):
        nuts = NUTS(logp, start(), step_size=0.3, max_depth=5, mass=mass, graph=mode,
                    generator=torch.Generator(device=dev).manual_seed(9))
        assert (nuts._graph is not None) == mode
        for it in range(20):
            nuts.step()
        out[mode] = (nuts.params[0].clone(), nuts.params[1].clone(), nuts.last_logp.clone(), nuts.mean_depth().clone())
    for a, b in zip(out[True], out[False]):
        assert torch.allclose(a, b, rtol=1e-8, atol=1e-11)
    t0s, rs, lp, depth = out[True]
    assert 1.0 <= float(depth.mean()) <= 5.0
    assert abs(float(t0s.mean()) - 1.0) < 1e-3 and abs(float(rs.mean()) - 0.1) < 8e-3      # (see the note above)


def test_native_tree_building_equals_the_torch_statement(dev):
    """exo_nuts_f64 (begin / leaf / merge kernels, leaf index on the device) against the torch statement of the same
    tree building in NUTS: the same chains, depths, divergences and acceptance statistics transition after transition
    on a correlated Gaussian with a wall (both read the same random numbers; every decision is the same)"""
    from exoplanet_amd.sampling import NUTS

    D = 97
    A = torch.tensor([[1.0, 0.6, 0.0], [0.6, 2.0, -0.3], [0.0, -0.3, 0.5]], dtype=torch.float64, device=dev)

    def logp(x, y):
        z = torch.cat([x, y], dim=1)
        lp = -0.5 * ((z @ A) * z).sum(-1)
        return torch.where(z[:, 0] > 2.5, torch.full_like(lp, -float("inf")), lp)      # a wall: divergences

    def make(native, graph):
        x = torch.linspace(-1, 1, 2 * D, dtype=torch.float64, device=dev).reshape(D, 2).clone()
        y = torch.linspace(0.5, -0.5, D, dtype=torch.float64, device=dev).reshape(D, 1).clone()
        s = NUTS(logp, [x, y], step_size=0.45, max_depth=5, graph=graph, mass=[torch.tensor([1.0, 2.0], device=dev), 0.7],
                 generator=torch.Generator(device=dev).manual_seed(5))
        if not native:
            s._native = None
        return s

    a, b, c = make(True, False), make(False, False), make(True, True)
    assert a._native is not None and c._graph is not None
    depths = []
    for it in range(25):
        da, db, dc = a.step(), b.step(), c.step()
        assert torch.equal(da, db) and torch.equal(da, dc), it
        assert torch.equal(a.last_diverged, b.last_diverged)
        for s2 in (b, c):
            for x, y in zip(a.params, s2.params):
                assert torch.allclose(x, y, rtol=1e-10, atol=1e-12), it
            assert torch.allclose(a.last_accept_prob, s2.last_accept_prob, rtol=1e-10, atol=1e-12)
            assert torch.allclose(a.last_logp, s2.last_logp, rtol=1e-10, atol=1e-10)
        depths.append(float(da.mean()))
    assert a.n_leapfrog == b.n_leapfrog == c.n_leapfrog
    assert 1.5 < np.mean(depths) < 5 and float(a.n_divergent.sum()) > 0 and float(da.max()) >= 3


# ---------------------------------------------------------------------------------------------------
# the DISTRIBUTION the native path samples: closed-form moments, Monte-Carlo-error bounds
# ---------------------------------------------------------------------------------------------------
_COV = np.array([[1.0, 0.6, -0.3], [0.6, 2.0, 0.5], [-0.3, 0.5, 0.7]])
_MU = np.array([0.5, -1.0, 2.0])


def _gauss_logp(dev):
    A = torch.as_tensor(np.linalg.inv(_COV), device=dev)
    mu = torch.as_tensor(_MU, device=dev)

    def logp(x, y):            # two parameter blocks, as the transit models have: (D, 2) and (D, 1)
        z = torch.cat([x, y], dim=1) - mu
        return -0.5 * ((z @ A) * z).sum(-1)
    return logp


def _moment_checks(draws, n_chain_steps, accept, target):
    """draws (steps, chains, 3).  Means within 4 sigma of their Monte-Carlo error, covariance entries within 10 % of
    sqrt(S_ii S_jj), acceptance statistic near its target.  The MC error uses the effective sample size measured from
    the chains themselves (lag autocorrelations of every chain, averaged)."""
    S, D, d = draws.shape
    flat = draws.reshape(-1, d)
    x = draws - draws.mean(0, keepdims=True).mean(1, keepdims=True)
    var = (x * x).mean((0, 1))
    tau = np.ones(d)
    for lag in range(1, 40):
        rho = (x[lag:] * x[:-lag]).mean((0, 1)) / var
        if np.all(rho < 0.05):
            break
        tau += 2 * np.maximum(rho, 0)
    n_eff = S * D / tau
    sd = np.sqrt(np.diag(_COV))
    err = np.abs(flat.mean(0) - _MU)
    assert np.all(err < 4 * sd / np.sqrt(n_eff)), (err, 4 * sd / np.sqrt(n_eff), tau)
    C = np.cov(flat.T)
    scale = np.sqrt(np.outer(np.diag(_COV), np.diag(_COV)))
    assert np.abs(C - _COV).max() / 1.0 <= 0.1 * scale.max() and np.all(np.abs(C - _COV) <= 0.1 * scale), (C, _COV)
    # (dual averaging keeps the AVERAGED step size, which sits below the last iterate: the sampling-phase statistic ends
    # up between the target and ~0.1 above it, as in Stan / PyMC)
    assert target - 0.1 < accept < target + 0.15, accept
    return tau


def test_native_nuts_samples_the_right_distribution(dev):
    """256 chains x 500 post-warm-up NUTS transitions on a 3-D correlated Gaussian with the native tree kernels
    (exo_nuts_f64) and the leaf replayed as a hipGraph: first and second moments against the closed form within
    Monte-Carlo error, energy-error acceptance statistic at its target"""
    from exoplanet_amd.sampling import NUTS

    D = 256
    g = torch.Generator(device=dev).manual_seed(123)
    x = torch.randn(D, 2, dtype=torch.float64, device=dev, generator=g)
    y = torch.randn(D, 1, dtype=torch.float64, device=dev, generator=g)
    smp = NUTS(_gauss_logp(dev), [x, y], step_size=0.3, max_depth=6, generator=g)
    assert smp._native is not None and smp._graph is not None
    smp.warmup(200, target_accept=0.8)
    draws, acc = [], []
    for _ in range(500):
        smp.step()
        draws.append(torch.cat([smp.params[0], smp.params[1]], dim=1).clone())
        acc.append(smp.last_accept_prob.clone())
    draws = torch.stack(draws).cpu().numpy()
    tau = _moment_checks(draws, None, float(torch.stack(acc).mean()), 0.8)
    assert np.all(tau < 4.0), tau                  # NUTS on a Gaussian: nearly independent draws
    assert float(smp.n_divergent.sum()) == 0
    assert 1.5 < float(smp.mean_depth().mean()) < 5.0


def test_native_hmc_samples_the_right_distribution(dev):
    """the same target through HMC (trajectory as one hipGraph, per-chain step sizes from the warm-up)"""
    from exoplanet_amd.sampling import HMC

    D = 256
    g = torch.Generator(device=dev).manual_seed(321)
    x = torch.randn(D, 2, dtype=torch.float64, device=dev, generator=g)
    y = torch.randn(D, 1, dtype=torch.float64, device=dev, generator=g)
    smp = HMC(_gauss_logp(dev), [x, y], step_size=0.3, n_leapfrog=4, generator=g)
    assert smp._graph is not None
    smp.warmup(200, target_accept=0.8)
    draws, acc = [], []
    for _ in range(600):
        smp.step()
        draws.append(torch.cat([smp.params[0], smp.params[1]], dim=1).clone())
        acc.append(smp.last_accept_prob.clone())
    draws = torch.stack(draws).cpu().numpy()
    _moment_checks(draws, None, float(torch.stack(acc).mean()), 0.8)


def test_native_nuts_with_a_wall_samples_the_truncated_distribution(dev):
    """the wall case: the Gaussian truncated at z0 < mu0 + 1.2 sigma0 (log-density -inf beyond: divergences).  The
    marginal of z0 is a truncated normal with closed-form mean and variance; nothing ever sits beyond the wall"""
    from scipy.stats import truncnorm
    from exoplanet_amd.sampling import NUTS

    D = 256
    base = _gauss_logp(dev)
    wall = _MU[0] + 1.2 * np.sqrt(_COV[0, 0])

    def logp(x, y):
        lp = base(x, y)
        return torch.where(x[:, 0] > wall, torch.full_like(lp, -float("inf")), lp)

    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(D, 2, dtype=torch.float64, device=dev, generator=g) * 0.3
    y = torch.randn(D, 1, dtype=torch.float64, device=dev, generator=g) * 0.3
    smp = NUTS(logp, [x, y], step_size=0.25, max_depth=6, generator=g)
    assert smp._native is not None
    smp.warmup(200, target_accept=0.8)
    z0 = []
    for _ in range(600):
        smp.step()
        z0.append(smp.params[0][:, 0].clone())
    z0 = torch.stack(z0).cpu().numpy()
    assert z0.max() <= wall
    tn = truncnorm(-np.inf, 1.2, loc=_MU[0], scale=np.sqrt(_COV[0, 0]))
    n_eff = z0.size / 4.0
    assert abs(z0.mean() - tn.mean()) < 4 * tn.std() / np.sqrt(n_eff), (z0.mean(), tn.mean())
    assert abs(z0.std() / tn.std() - 1) < 0.05
    assert float(smp.n_divergent.sum()) > 0
