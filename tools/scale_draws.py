#!/usr/bin/env python
"""chi2 (white-noise misfit + gradient) sweep of C2 against the number of draws: how much of the time at 1024 draws is
the second, partly filled generation of blocks (768 resident slots)?  tools/scale_draws.py [draws ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = 150_000
t = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 / 1440.0)
obs = 1e-4 * torch.randn(N, dtype=torch.float64, device=dev)
w = torch.tensor([1e8], dtype=torch.float64, device=dev)


def med(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)])) * 1e3


for D in [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1024, 1536, 2048, 3072]:
    rng = np.random.default_rng(100)
    base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1)
    L = {k: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev) for k, v in base.items()}
    orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"])
    rec, ld, _, flags = orbit.kernel_inputs(L["r"], (0.3, 0.2), use_in_transit=False)
    rec, ld = rec.detach().contiguous(), ld.detach().contiguous()
    us = med(lambda: ops.transit_chi2(t, rec, ld, obs, w, flags=flags))
    print(f"draws {D:5d}  chi2 {us:7.1f} us  {us / D * 1e3:6.1f} ns per draw")
