"""one case of bench.py's c3_gp_conditioning leg, launched eagerly a few times (for rocprofv3 --kernel-trace --stats):
python tools/gp_cond_one.py near|clean|two"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench, exoplanet_amd as xo
from exoplanet_amd import ops
dev = torch.device('cuda:0'); D = 1024; T = xo.gp.terms
t = ops.vouch_sorted(torch.arange(bench.N_CAD, dtype=torch.float64, device=dev) * bench.CADENCE)
y = torch.as_tensor(5e-4 * np.random.default_rng(3).normal(size=bench.N_CAD), device=dev)
model = torch.zeros(bench.N_CAD, D, dtype=torch.float64, device=dev).t().requires_grad_(True)
full = lambda v: torch.full((D,), v, dtype=torch.float64, device=dev, requires_grad=True)
Q = np.full(D, 0.7071)
if sys.argv[1] == "near":
    Q[:10:2] = 0.505; Q[1:10:2] = 0.495
h = [full(1e-3), full(5.0), torch.tensor(Q, device=dev, requires_grad=True)]
two = sys.argv[1] == "two"      # (two SHO terms: J = 4)
if two:
    h = [full(1e-3), full(5.0), full(7e-4), full(2.5)]
for _ in range(4):
    kern = (T.SHOTerm(sigma=h[0], rho=h[1], Q=1.2) + T.SHOTerm(sigma=h[2], rho=h[3], Q=1.5)) if two else T.SHOTerm(sigma=h[0], rho=h[1], Q=h[2])
    gp = xo.gp.GaussianProcess(kern, t=t, yerr=5e-4, mean=model)
    ll = gp.log_likelihood(y)
    torch.autograd.grad(ll.sum(), [model] + h)
torch.cuda.synchronize()
