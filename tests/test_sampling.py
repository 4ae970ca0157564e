"""CPU: the batched HMC / NUTS drivers (exoplanet_amd/sampling.py) are device-agnostic; their logic -- leapfrog,
per-chain Metropolis step or tree building, in-place update of the chains, warm-up -- is checked here on targets with
known moments.  On a GPU the same classes replay their leapfrog steps as hipGraphs (tests/test_gpu_sampling.py)."""
import numpy as np
import torch

from exoplanet_amd.sampling import HMC, NUTS, _ckpt_range


def test_hmc_samples_a_gaussian():
    torch.manual_seed(0)
    D, d = 64, 3
    mu = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    sd = torch.tensor([0.5, 2.0, 1.0], dtype=torch.float64)

    def logp(x, s):          # two parameter blocks: a vector and a scalar per chain
        return (-0.5 * ((x - mu) / sd) ** 2).sum(-1) - 0.5 * (s[:, 0] / 3.0) ** 2

    x = torch.zeros(D, d, dtype=torch.float64)
    s = torch.zeros(D, 1, dtype=torch.float64)
    g = torch.Generator().manual_seed(1)
    hmc = HMC(logp, [x, s], step_size=0.35, n_leapfrog=6, mass=[1.0 / sd**2, 1.0 / 9.0], generator=g)
    xs, ss = [], []
    for it in range(400):
        acc = hmc.step()
        assert acc.shape == (D,) and acc.dtype == torch.bool
        if it >= 50:
            xs.append(hmc.params[0].clone()); ss.append(hmc.params[1].clone())
    xs, ss = torch.stack(xs).reshape(-1, d), torch.stack(ss).reshape(-1)
    rate = float(hmc.accept_rate().mean())
    assert 0.6 < rate <= 1.0
    n_eff = xs.shape[0] / 10
    assert torch.all((xs.mean(0) - mu).abs() < 5 * sd / np.sqrt(n_eff))
    assert torch.all((xs.std(0) / sd - 1).abs() < 0.1)
    assert abs(float(ss.std()) / 3.0 - 1) < 0.1
    # the chains' own tensors were updated in place
    assert hmc.params[0].data_ptr() == x.data_ptr() or torch.equal(hmc.params[0], x)


def test_hmc_rejects_invalid_proposals_and_conserves_energy():
    D = 8
    q = torch.zeros(D, 1, dtype=torch.float64)

    def logp(x):             # a wall at |x| > 1.5: -inf beyond it
        lp = -0.5 * x[:, 0] ** 2
        return torch.where(x[:, 0].abs() > 1.5, torch.full_like(lp, -float("inf")), lp)

    g = torch.Generator().manual_seed(3)
    hmc = HMC(logp, [q], step_size=0.2, n_leapfrog=5, generator=g)
    for _ in range(200):
        hmc.step()
        assert bool((hmc.params[0].abs() <= 1.5).all())
    # tiny steps conserve the Hamiltonian: every proposal accepted
    hmc2 = HMC(lambda x: -0.5 * (x ** 2).sum(-1), [torch.zeros(D, 2, dtype=torch.float64)], step_size=1e-3, n_leapfrog=3,
               generator=torch.Generator().manual_seed(4))
    assert all(bool(hmc2.step().all()) for _ in range(20))


def test_warmup_adapts_step_sizes_and_masses():
    """dual averaging brings every chain's acceptance rate to the target; the pooled mass adaptation finds the scales
    of a badly scaled Gaussian (1e-2 .. 1e2) from a unit mass matrix"""
    torch.manual_seed(5)
    D = 96
    sd = torch.tensor([1e-2, 1.0, 1e2], dtype=torch.float64)
    logp = lambda x: (-0.5 * (x / sd) ** 2).sum(-1)  # noqa: E731
    x = torch.randn(D, 3, dtype=torch.float64) * sd
    hmc = HMC(logp, [x], step_size=1.0, n_leapfrog=8, generator=torch.Generator().manual_seed(6))
    eps = hmc.warmup(600, target_accept=0.8, adapt_mass=True)
    assert eps.shape == (D,) and bool((eps > 0).all())
    got = (1.0 / hmc.mass[0][0]).sqrt()
    assert torch.all((got / sd - 1).abs() < 0.3), got
    xs = []
    for _ in range(300):
        hmc.step()
        xs.append(hmc.params[0].clone())
    rate = hmc.accept_rate()
    assert 0.65 < float(rate.mean()) < 0.95 and float(rate.min()) > 0.4
    xs = torch.stack(xs).reshape(-1, 3)
    assert torch.all((xs.std(0) / sd - 1).abs() < 0.15)
    # without mass adaptation: step sizes only (the stiffest direction sets them)
    hmc2 = HMC(logp, [torch.randn(D, 3, dtype=torch.float64) * sd], step_size=1.0, n_leapfrog=4,
               generator=torch.Generator().manual_seed(7))
    eps2 = hmc2.warmup(300)
    assert float(eps2.median()) < 0.05
    for _ in range(100):
        hmc2.step()
    assert 0.6 < float(hmc2.accept_rate().mean()) < 0.97


def test_nuts_checkpoint_slots():
    """leaf n (odd) closes the sub-trees of 2, 4, ... leaves that end at it -- one per trailing 1 bit -- whose first
    leaves sit in consecutive checkpoint slots; brute force over every aligned sub-tree of a 64-leaf tree"""
    first_slot = {}                      # first leaf (even) -> slot it was stored in
    for n in range(64):
        if n % 2 == 0:
            first_slot[n] = bin(n >> 1).count("1")
            continue
        lo, hi = _ckpt_range(n)
        want = []
        size = 2
        while (n + 1) % size == 0 and size <= 64:
            want.append(first_slot[n + 1 - size])      # slots are reused: the value must be that sub-tree's first leaf's
            size *= 2
        assert list(range(hi, lo - 1, -1)) == want, (n, lo, hi, want)
    assert _ckpt_range(7) == (0, 2) and _ckpt_range(11) == (1, 2) and _ckpt_range(1) == (0, 0)


def test_nuts_samples_correlated_and_skewed_targets():
    """a correlated Gaussian (rho = 0.9, unit masses) and log of a Gamma(2.5) variate: covariance, mean, variance and
    third central moment against the closed forms (digamma, trigamma, tetragamma)"""
    from scipy.special import digamma, polygamma

    D, rho, k = 384, 0.9, 2.5

    def logp(x, y):
        a, b = x[:, 0], x[:, 1]
        return -(a * a - 2 * rho * a * b + b * b) / (2 * (1 - rho * rho)) + (k * y[:, 0] - torch.exp(y[:, 0]))

    torch.manual_seed(0)
    x, y = torch.randn(D, 2, dtype=torch.float64), torch.zeros(D, 1, dtype=torch.float64)
    nuts = NUTS(logp, [x, y], step_size=0.1, max_depth=6, generator=torch.Generator().manual_seed(3))
    eps = nuts.warmup(80)
    assert eps.shape == (D,) and 0.1 < float(eps.mean()) < 1.0
    dx, dy = [], []
    for _ in range(250):
        depth = nuts.step()
        assert depth.shape == (D,) and 1 <= float(depth.min()) and float(depth.max()) <= 6
        dx.append(nuts.params[0].clone()); dy.append(nuts.params[1].clone())
    assert float(nuts.n_divergent.sum()) == 0
    assert 0.6 < float(nuts.last_accept_prob.mean()) <= 1.0
    assert 1.5 < float(nuts.mean_depth().mean()) < 5
    dx, dy = torch.stack(dx).reshape(-1, 2), torch.stack(dy).reshape(-1)
    cov = torch.cov(dx.T)
    assert abs(float(cov[0, 0]) - 1) < 0.05 and abs(float(cov[1, 1]) - 1) < 0.05 and abs(float(cov[0, 1]) - rho) < 0.05
    assert abs(float(dy.mean()) - digamma(k)) < 0.02
    assert abs(float(dy.var()) / polygamma(1, k) - 1) < 0.05
    assert abs(float(((dy - dy.mean()) ** 3).mean()) / polygamma(2, k) - 1) < 0.15
    # the chains' own tensors were updated in place
    assert torch.equal(nuts.params[0], x)


def test_nuts_walls_divergences_and_depth_limit():
    D = 16

    def logp(x):             # a wall at |x| > 1.5: -inf beyond it (such leaves diverge and end the tree)
        lp = -0.5 * x[:, 0] ** 2
        return torch.where(x[:, 0].abs() > 1.5, torch.full_like(lp, -float("inf")), lp)

    nuts = NUTS(logp, [torch.zeros(D, 1, dtype=torch.float64)], step_size=0.3, max_depth=5,
                generator=torch.Generator().manual_seed(3))
    for _ in range(60):
        nuts.step()
        assert bool((nuts.params[0].abs() <= 1.5).all()) and bool(torch.isfinite(nuts.last_logp).all())
    assert float(nuts.n_divergent.sum()) > 0
    # tiny steps on a wide Gaussian: no U-turn within the depth limit -- every tree has exactly max_depth doublings,
    # 2**max_depth - 1 leapfrog steps, and (energy conserved) every leaf is accepted
    nuts2 = NUTS(lambda x: -0.5 * (x ** 2).sum(-1), [torch.zeros(D, 2, dtype=torch.float64)], step_size=1e-3, max_depth=4,
                 generator=torch.Generator().manual_seed(4))
    d = nuts2.step()
    assert bool((d == 4).all()) and nuts2.n_leapfrog == 15 and float(nuts2.last_accept_prob.min()) > 0.999
    # same seed, same chain
    a = NUTS(lambda x: -0.5 * (x ** 2).sum(-1), [torch.ones(D, 2, dtype=torch.float64)], step_size=0.4, max_depth=6,
             generator=torch.Generator().manual_seed(8))
    b = NUTS(lambda x: -0.5 * (x ** 2).sum(-1), [torch.ones(D, 2, dtype=torch.float64)], step_size=0.4, max_depth=6,
             generator=torch.Generator().manual_seed(8))
    for _ in range(5):
        a.step(); b.step()
    assert torch.equal(a.params[0], b.params[0])


def test_nuts_chain_outside_the_support_stays_put():
    D = 6

    def logp(x):
        lp = -0.5 * x[:, 0] ** 2
        return torch.where(x[:, 0] > 2.0, torch.full_like(lp, -float("inf")), lp)

    x = torch.zeros(D, 1, dtype=torch.float64)
    x[0, 0] = 5.0                                # outside: log-density -inf from the start
    nuts = NUTS(logp, [x], step_size=0.5, max_depth=4, generator=torch.Generator().manual_seed(2))
    for _ in range(10):
        d = nuts.step()
        assert float(nuts.params[0][0, 0]) == 5.0 and float(d[0]) == 0
        assert bool(torch.isfinite(nuts.params[0]).all()) and bool((nuts.params[0][1:, 0] <= 2.0).all())
    assert float(nuts.params[0][1:].abs().max()) > 0       # the others move


def test_warmup_leaves_chains_outside_the_support_alone():
    """a chain that starts outside the support never has a valid leaf; its acceptance statistic says nothing
    about its step size, so dual averaging must not drive that step size to zero"""
    D = 6

    def logp(x):
        lp = -0.5 * (x ** 2).sum(-1)
        return torch.where(x[:, 0] > 5.0, torch.full_like(lp, -float("inf")), lp)

    for cls, kw in ((NUTS, dict(max_depth=4)), (HMC, dict(n_leapfrog=4))):
        x = torch.zeros(D, 2, dtype=torch.float64)
        x[0, 0] = 9.0                                     # outside the support
        smp = cls(logp, [x], step_size=0.5, generator=torch.Generator().manual_seed(2), **kw)
        eps = smp.warmup(60, target_accept=0.8)
        if cls is NUTS:
            assert float(smp.params[0][0, 0]) == 9.0     # it stays where it is (HMC may step back inside: -inf -> finite is accepted)
        assert torch.isfinite(eps).all() and float(eps[0]) > 1e-3, eps      # and keeps a usable step size
        assert torch.all(eps[1:] > 0.1) and torch.all(eps[1:] < 5.0)
