import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from exoplanet_amd.gp import celerite_loglike
from oracle import numpy_port as P
from test_gpu_transit import make_record
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
rng = np.random.default_rng(1)
if which == "c3":
    N, D = 150_000, 1024
    t = np.arange(N) * (2.0 / 1440.0)
    co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3, 5.0, 1 / np.sqrt(2)), 1 / np.sqrt(2))
    diagv = 2.5e-7
else:
    N, D = 65_000, 128
    t = np.arange(N) * (29.4 / 1440.0)
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s, r, q), q) for s, r, q in ((4e-4, 20.0, 2.0), (3e-4, 10.0, 1.0), (2e-4, 2.0, 1 / np.sqrt(2)))]
    co = tuple(np.concatenate(x) for x in zip(*parts))
    diagv = 9e-8
cplx = T(np.repeat(np.stack(co[2:], -1)[None], D, 0)).requires_grad_(True)
real = T(np.zeros((D, 0, 2)))
resid = (1e-3 * torch.randn(D, N, dtype=torch.float64, device=dev)).requires_grad_(True)
diag = T(np.full((1, N), diagv))
tt = T(t)
for _ in range(3):
    ll = celerite_loglike(tt, resid, diag, real, cplx)
    torch.autograd.grad(ll.sum(), (resid, cplx))
torch.cuda.synchronize()
