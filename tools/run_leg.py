#!/usr/bin/env python
"""run one op-level leg of the C2 sweep a few times (for rocprofv3 --pmc passes): tools/run_leg.py dense|sparse|chi2|value [draws] [iters], or kepler|quadsv [elements] [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

leg = sys.argv[1]
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
if leg in ("kepler", "quadsv"):
    # the reference's standalone Ops on n = D elements (bench.py extra_ops: the same inputs)
    n = D
    gen = torch.Generator(device=dev).manual_seed(11)
    with torch.no_grad():
        if leg == "kepler":
            M = (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5) * 800.0
            e = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 0.9
            for _ in range(iters):
                ops.kepler(M, e)
        else:
            b = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 1.3
            r = torch.full((n,), 0.1, dtype=torch.float64, device=dev)
            for _ in range(iters):
                ops.quad_solution_vector(b, r)
            for _ in range(iters):
                ops.quad_solution_vector_derivs(b, r)
    torch.cuda.synchronize()
    sys.exit(0)
N = 150_000
t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
rng = np.random.default_rng(100)
base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1)
L = {k: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev) for k, v in base.items()}
orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"])
rec, ld, _, flags = orbit.kernel_inputs(L["r"], (0.3, 0.2), use_in_transit=False)
rec, ld = rec.detach().contiguous(), ld.detach().contiguous()
g = torch.randn(D, N, dtype=torch.float64, device=dev)
obs = 1e-4 * torch.randn(N, dtype=torch.float64, device=dev)
w = torch.tensor([1e8], dtype=torch.float64, device=dev)
fn = {"dense": lambda: ops.transit_flux_value_and_vjp(t, rec, ld, g, flags=flags),
      "sparse": lambda: ops.transit_flux_sparse(t, rec, ld, gflux=g, flags=flags),
      "chi2": lambda: ops.transit_chi2(t, rec, ld, obs, w, flags=flags),
      "value": lambda: ops.transit_flux_sparse(t, rec, ld, flags=flags)}[leg]
with torch.no_grad():
    for _ in range(iters):
        fn()
torch.cuda.synchronize()
