"""End-to-end use of the sampler on the path itself: inject a transit into synthetic data, sample the
posterior with the native batched NUTS (128 chains, tree kernels exo_nuts_f64, the leaf replayed as a hipGraph) through
(a) `LimbDarkLightCurve.white_noise_log_likelihood` and (b) the C3 model -- light curve as the mean of a celerite SHO
`GaussianProcess` -- and require the truth back: every posterior mean within 3 posterior standard deviations of the
injected value, between- / within-chain variance ratio R-hat <= 1.05, no divergences.

The reference runs this as `pm.sample` on the same model (docs/tutorials/data-and-models.md:458-509); the likelihood
underneath is the one the parity tests hold to the oracle -- this test is about the pieces working TOGETHER: constructor
columns -> packing kernel -> sweep -> likelihood -> autograd -> leapfrog -> tree kernels, replayed thousands of times.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 20_000
CAD = 2.0 / 1440.0
TRUTH = dict(t0=1.0, r=0.1, b=0.5)
SCALE = dict(t0=1e-4, r=1e-3, b=0.05)          # the sampler works in units of these around TRUTH (a user standardises too)
FIXED = dict(period=3.5, ecc=0.3, omega=1.1)


def rhat(x):
    """x [draw][chain]: sqrt(((n - 1) / n W + B / n) / W)"""
    n = x.shape[0]
    W = x.var(axis=0, ddof=1).mean()
    B = n * x.mean(axis=0).var(ddof=1)
    return float(np.sqrt(((n - 1) / n * W + B / n) / W))


def orbit_of(xo, q, dev):
    t0 = TRUTH["t0"] + SCALE["t0"] * q[:, 0:1]
    r = TRUTH["r"] + SCALE["r"] * q[:, 1:2]
    b = TRUTH["b"] + SCALE["b"] * q[:, 2:3]
    one = torch.ones_like(t0)
    orbit = xo.KeplerianOrbit(period=FIXED["period"] * one, t0=t0, b=b, ecc=FIXED["ecc"] * one, omega=FIXED["omega"] * one)
    return orbit, r, b


def truth_curve(xo, t, dev):
    with torch.no_grad():
        orbit, r, _ = orbit_of(xo, torch.zeros(1, 3, dtype=torch.float64, device=dev), dev)
        return xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t).sum(-1).reshape(-1)


def run_nuts(xo, logp, q0, gen, n_warm, n_draw, max_depth=5):
    smp = xo.NUTS(logp, [q0], step_size=0.05, max_depth=max_depth, generator=gen)
    assert smp._native is not None and smp._graph is not None
    smp.warmup(n_warm, target_accept=0.8, adapt_mass=True)
    draws = []
    for _ in range(n_draw):
        smp.step()
        draws.append(smp.params[0].clone())
    return torch.stack(draws).cpu().numpy(), smp


def check(draws, names, truth_q, smp):
    assert np.isfinite(draws).all()
    assert float(smp.n_divergent.sum()) == 0
    for k, name in enumerate(names):
        x = draws[:, :, k]
        mean, sd = x.mean(), x.std()
        assert abs(mean - truth_q[k]) < 3.0 * sd, (name, mean, sd, truth_q[k])
        assert rhat(x) <= 1.05, (name, rhat(x))
        assert sd > 0


def test_nuts_recovers_an_injected_transit_white_noise(dev):
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D, yerr = 128, 5e-4
    t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * CAD)     # fixed for the run: the caller's word
    gen = torch.Generator(device=dev).manual_seed(2024)
    f = truth_curve(xo, t, dev)
    assert float(f.min()) < -5e-3 and int((f < 0).sum()) > 400
    y = f + yerr * torch.randn(N, dtype=torch.float64, device=dev, generator=gen)

    def logp(q):
        orbit, r, b = orbit_of(xo, q, dev)
        ll = xo.LimbDarkLightCurve(0.3, 0.2).white_noise_log_likelihood(orbit=orbit, r=r, t=t, y=y, yerr=yerr)
        inside = ((b > 0.0) & (b < 1.0)).reshape(-1)
        return torch.where(inside, ll.reshape(-1), torch.full_like(ll.reshape(-1), -float("inf")))

    q0 = 2.0 * torch.randn(D, 3, dtype=torch.float64, device=dev, generator=gen)       # over-dispersed starts
    draws, smp = run_nuts(xo, logp, q0, gen, 200, 250)
    check(draws, ("t0", "r", "b"), (0.0, 0.0, 0.0), smp)
    # the posterior is as narrow as the data say: the radius to better than 2 %, mid-transit time to under a minute
    assert draws[:, :, 1].std() * SCALE["r"] < 0.02 * TRUTH["r"]
    assert draws[:, :, 0].std() * SCALE["t0"] < 1.0 / 1440.0


@pytest.mark.parametrize("mean,seed", [("cadence_major", 99), ("cadence_major", 102), ("sparse", 99), ("sparse", 104)])
def test_nuts_recovers_transit_and_gp_amplitude_c3_model(dev, mean, seed):
    """mean: how the light curve reaches the GP -- the dense cadence-major array, or the sparse model of round 5 (segments + values;
    the draws' order worked out inside the captured leaf).  Seeds 102 and 104 are the ones on which, in round 5, a memset node of the
    captured leaf went bad after ~1000 replays and the odd chains' step sizes collapsed (exo_math.hpp, zero_fill_async)."""
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D, yerr, sigma, rho, Q = 128, 3e-4, 8e-4, 1.5, 0.7071
    t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * CAD)
    gen = torch.Generator(device=dev).manual_seed(seed)
    f = truth_curve(xo, t, dev)
    T = xo.gp.terms
    with torch.no_grad():
        noise = xo.gp.GaussianProcess(T.SHOTerm(sigma=sigma, rho=rho, Q=Q), t=t, yerr=yerr).sample(generator=gen).reshape(-1)
    assert 0.3 * sigma < float(noise.std()) < 3 * sigma
    y = f + noise
    ones = torch.ones(D, dtype=torch.float64, device=dev)

    def logp(q):
        orbit, r, b = orbit_of(xo, torch.cat([q[:, :2], torch.zeros_like(q[:, :1])], dim=1), dev)      # b fixed at its truth
        lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t, total=True, **{mean: True})
        if mean == "sparse":
            assert isinstance(lc, ops.SparseLightCurve)
        s = sigma * torch.exp(0.1 * q[:, 2])
        gp = xo.gp.GaussianProcess(T.SHOTerm(sigma=s, rho=rho * ones, Q=Q * ones), t=t, yerr=yerr, mean=lc)
        return gp.log_likelihood(y) + 0.1 * q[:, 2]          # (flat prior on sigma: the Jacobian of the log scale)

    q0 = 1.5 * torch.randn(D, 3, dtype=torch.float64, device=dev, generator=gen)
    draws, smp = run_nuts(xo, logp, q0, gen, 200, 250)
    check(draws, ("t0", "r", "log sigma"), (0.0, 0.0, 0.0), smp)
    # the GP's amplitude is measured, not just carried along: a 27-day series with rho = 1.5 d pins it to a few tens of %
    assert draws[:, :, 2].std() * 0.1 < 0.5


def test_nuts_recovers_transit_and_amplitude_with_a_wide_kernel(dev):
    """round 6: the noise model is two RotationTerms + an SHO term -- J = 10, on the time-parallel path for wide states (a draw on a
    DPP row of sixteen lanes, the scans on celerite_tree_wide_kernel) -- sampled for 128 chains with the native NUTS, the leaf
    replayed as a hipGraph a few thousand times: transit time, radius and the first term's amplitude come back"""
    import exoplanet_amd as xo
    from exoplanet_amd import ops

    D, yerr = 128, 3e-4
    t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * CAD)
    gen = torch.Generator(device=dev).manual_seed(311)
    f = truth_curve(xo, t, dev)
    T = xo.gp.terms
    sig1 = 8e-4

    def kernel(s1, like):
        one = torch.ones_like(like)
        return (T.RotationTerm(sigma=s1, period=2.3 * one, Q0=1.5 * one, dQ=0.4 * one, f=0.6 * one)
                + T.RotationTerm(sigma=3e-4 * one, period=5.9 * one, Q0=2.5 * one, dQ=0.7 * one, f=0.3 * one)
                + T.SHOTerm(sigma=2e-4 * one, rho=0.6 * one, Q=0.7071 * one))

    with torch.no_grad():
        one1 = torch.ones(1, dtype=torch.float64, device=dev)
        noise = xo.gp.GaussianProcess(kernel(sig1 * one1, one1), t=t, yerr=yerr).sample(generator=gen).reshape(-1)
    assert 0.3 * sig1 < float(noise.std()) < 3 * sig1
    y = f + noise

    def logp(q):
        orbit, r, b = orbit_of(xo, torch.cat([q[:, :2], torch.zeros_like(q[:, :1])], dim=1), dev)      # b fixed at its truth
        lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t, total=True, cadence_major=True)
        s = sig1 * torch.exp(0.1 * q[:, 2])
        gp = xo.gp.GaussianProcess(kernel(s, s), t=t, yerr=yerr, mean=lc)
        assert gp.kernel.pair_coefficients()[2].shape[-2] == 5          # five pair slots: J = 10
        return gp.log_likelihood(y) + 0.1 * q[:, 2]

    q0 = 1.5 * torch.randn(D, 3, dtype=torch.float64, device=dev, generator=gen)
    draws, smp = run_nuts(xo, logp, q0, gen, 150, 200)
    check(draws, ("t0", "r", "log sigma"), (0.0, 0.0, 0.0), smp)
    assert draws[:, :, 2].std() * 0.1 < 0.5
