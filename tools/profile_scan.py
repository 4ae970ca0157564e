"""scan-kernel A/B: calls with and without the flux fill, to be run under rocprofv3 --kernel-trace"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import exoplanet_amd as xo
from exoplanet_amd import ops

dev = torch.device("cuda:0")
D, N = 1024, 150000
rng = np.random.default_rng(0)
t = torch.tensor(np.arange(N) * (2.0 / 1440.0), device=dev)
g = torch.randn(D, N, dtype=torch.float64, device=dev)
def mk(v): return torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), device=dev, requires_grad=True)
orbit = xo.KeplerianOrbit(period=mk(3.5), t0=mk(1.0), b=mk(0.3), ecc=mk(0.3), omega=mk(1.1))
rec, ld, _, flags = orbit.kernel_inputs(mk(0.1), (0.3, 0.2))
rec = rec.detach().requires_grad_(True)
for _ in range(5):
    flux = ops.transit_flux(t, rec, ld.detach(), flags=flags)      # fill + classify, heavy<false>
    flux.backward(g)                                               # classify only, heavy<true>
torch.cuda.synchronize()
