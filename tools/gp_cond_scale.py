"""time-parallel vs sequential celerite kernels at the BENCHMARKED sizes as the conditioning score grows (run with a library
built with -DEXO_GP_COND_MAX=1e30, so that nothing is flagged): C3 shape (N = 150 000, J = 2) with an SHO term from
well under- to nearly critically damped and a Matern-3/2 term, signal / noise from 1 to 1e6; C5 shape (N = 65 000, J = 6).
Prints, per case, kappa and the worst relative disagreement of the log-likelihood and of every gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from exoplanet_amd.gp import celerite_loglike
from oracle import numpy_port as P

dev = torch.device("cuda:0")
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)  # noqa: E731


def run(t, y, diag, cr, cc, kind, chunks):
    yt, dt, crt, cct = T(y, True), T(diag, True), T(cr, True), T(cc, True)
    kd = None if kind is None else torch.as_tensor(kind, dtype=torch.int32, device=dev)
    ll = celerite_loglike(T(t), yt, dt, crt, cct, pair_kind=kd, n_chunks=chunks)
    ll.sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, cct.grad)]


def compare(tag, t, y, diag, cc, kind=None):
    D = y.shape[0]
    cr = np.zeros((D, 0, 2))
    if diag.shape[0] == 1:
        diag = np.repeat(diag, D, 0)
    want = run(t, y, diag, cr, cc, kind, 1)
    got = run(t, y, diag, cr, cc, kind, 0)
    for d in range(D):
        a = np.where(kind[d] != 0, cc[d, :, 0] + cc[d, :, 2], cc[d, :, 0]).sum() if kind is not None else cc[d, :, 0].sum()
        ba2 = 0.0
        for j in range(cc.shape[1]):
            if kind is not None and kind[d, j]:
                ba2 = max(ba2, ((abs(cc[d, j, 0]) + abs(cc[d, j, 2])) / (cc[d, j, 0] + cc[d, j, 2])) ** 2)
            else:
                ba2 = max(ba2, (cc[d, j, 1] / cc[d, j, 0]) ** 2)
        kappa = (1 + ba2) * a / diag[d if diag.shape[0] > 1 else 0].min()
        e = [abs(got[0][d] - want[0][d]) / abs(want[0][d])]
        for g, w in zip(got[1:], want[1:]):
            e.append(np.abs(g[d] - w[d]).max() / (np.abs(w[d]).max() + 1e-300))
        print(f"{tag} kappa {kappa:9.2e}  ll {e[0]:8.1e}  gy {e[1]:8.1e}  gdiag {e[2]:8.1e}  gcoef {e[3]:8.1e}")


rng = np.random.default_rng(0)
N = 150_000
t = np.arange(N) * (2.0 / 1440.0)
sigma, rho = 1e-3, 5.0
for snr2 in (4.0, 1e2, 1e4, 1e6):
    diag = np.full((1, N), sigma ** 2 / snr2)
    Qs = [4.0, 0.7071, 0.52, 0.505, 0.5005, 0.49995 + 1e-9, 0.495, 0.45, 0.3]
    cc = np.zeros((len(Qs), 1, 4)); kind = np.zeros((len(Qs), 1), dtype=np.int32)
    for d, Q in enumerate(Qs):
        co = P.sho_coefficients(*P.sho_from_sigma_rho(sigma, rho, Q), Q)
        if co[0].size:
            cc[d, 0] = [co[0][0], co[1][0], co[0][1], co[1][1]]; kind[d, 0] = 1
        else:
            cc[d, 0] = [co[2][0], co[3][0], co[4][0], co[5][0]]
    y = np.sqrt(sigma ** 2 + diag[0, 0]) * rng.normal(size=(len(Qs), N))
    print(f"--- C3 shape, SHO Q = {Qs}, signal/noise variance {snr2:g}")
    compare("sho", t, y, diag, cc, kind)
    # celerite2's Matern32Term(sigma, rho, eps = 0.01)
    w0 = np.sqrt(3.0) / rho
    S0 = sigma ** 2 / w0
    cm = np.array([[[w0 * S0, w0 * w0 * S0 / 0.01, w0, 0.01]], [[w0 * S0, w0 * w0 * S0 / 0.001, w0, 0.001]]])
    compare("matern32 eps=.01,.001", t, y[:2], diag, cm)
