"""CPU: the batched HMC driver (exoplanet_amd/sampling.py) is device-agnostic; its logic -- leapfrog,
per-chain Metropolis step, in-place update of the chains -- is checked here on a Gaussian target.
On a GPU the same class replays the trajectory as a hipGraph (tests/test_gpu_sampling.py)."""
import numpy as np
import torch

from exoplanet_amd.sampling import HMC


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
