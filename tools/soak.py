#!/usr/bin/env python
"""Soak of the sampler on the C3 model (the test's model, tests/test_gpu_inject_recover.py) for many more transitions than the test:
python tools/soak.py [transitions] [cadence_major|sparse|wide]  -> R-hat per parameter, divergences, spread of the last third against
the first.  `wide`: the J = 10 noise model of round 6."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402
from test_gpu_inject_recover import CAD, N, orbit_of, rhat, truth_curve  # noqa: E402

n_draw = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
mean = sys.argv[2] if len(sys.argv) > 2 else "sparse"
dev = torch.device("cuda:0")
D, yerr, sigma, rho, Q = 128, 3e-4, 8e-4, 1.5, 0.7071
t = ops.vouch_sorted(torch.arange(N, dtype=torch.float64, device=dev) * CAD)
gen = torch.Generator(device=dev).manual_seed(99)
f = truth_curve(xo, t, dev)
T = xo.gp.terms
ones = torch.ones(D, dtype=torch.float64, device=dev)


def kernel(s, like):
    one = torch.ones_like(like)
    if mean != "wide":
        return T.SHOTerm(sigma=s, rho=rho * one, Q=Q * one)
    return (T.RotationTerm(sigma=s, period=2.3 * one, Q0=1.5 * one, dQ=0.4 * one, f=0.6 * one)
            + T.RotationTerm(sigma=3e-4 * one, period=5.9 * one, Q0=2.5 * one, dQ=0.7 * one, f=0.3 * one)
            + T.SHOTerm(sigma=2e-4 * one, rho=0.6 * one, Q=0.7071 * one))


with torch.no_grad():
    one1 = torch.ones(1, dtype=torch.float64, device=dev)
    noise = xo.gp.GaussianProcess(kernel(sigma * one1, one1), t=t, yerr=yerr).sample(generator=gen).reshape(-1)
y = f + noise
route = "cadence_major" if mean == "wide" else mean


def logp(q):
    orbit, r, b = orbit_of(xo, torch.cat([q[:, :2], torch.zeros_like(q[:, :1])], dim=1), dev)
    lc = xo.LimbDarkLightCurve(0.3, 0.2).get_light_curve(orbit=orbit, r=r, t=t, total=True, **{route: True})
    s = sigma * torch.exp(0.1 * q[:, 2])
    return xo.gp.GaussianProcess(kernel(s, s), t=t, yerr=yerr, mean=lc).log_likelihood(y) + 0.1 * q[:, 2]


q0 = 1.5 * torch.randn(D, 3, dtype=torch.float64, device=dev, generator=gen)
smp = xo.NUTS(logp, [q0], step_size=0.05, max_depth=5, generator=gen)
smp.warmup(200, target_accept=0.8, adapt_mass=True)
draws = []
for _ in range(n_draw):
    smp.step()
    draws.append(smp.params[0].clone())
x = torch.stack(draws).cpu().numpy()
third = n_draw // 3
print(mean, "transitions", n_draw, "finite", bool(np.isfinite(x).all()), "divergent", float(smp.n_divergent.sum()),
      "R-hat", [round(rhat(x[:, :, k]), 4) for k in range(3)],
      "sd last third / first third", [round(float(x[-third:, :, k].std() / x[:third, :, k].std()), 3) for k in range(3)])
