"""One cell of the conditioning scan, many draws: random kernels of J = 3 .. 6 whose terms are all WELL SAMPLED (decay rate c dt <= cmax
per sample, oscillation rates as tools/gp_cond_bins.py draws them) at conditioning scores kappa in [klo, khi); time-parallel path
(library built with the flags off) against the sequential kernels.  Is there room above kappa = 1e5 for such kernels?
usage: EXOPLANET_AMD_LIB=<noflag .so> python tools/gp_cond_cell.py <seed> <cases> <klo> <khi> <cmax>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from exoplanet_amd.gp import celerite_loglike

dev = torch.device("cuda:0")
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)  # noqa: E731


def run(t, y, diag, cr, cc, chunks):
    yt, dt, crt, cct = T(y, True), T(diag, True), T(cr, True), T(cc, True)
    ll = celerite_loglike(T(t), yt, dt, crt, cct, n_chunks=chunks)
    torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll)).sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, crt.grad, cct.grad)]


seed, n_cases = int(sys.argv[1]), int(sys.argv[2])
klo, khi, cmax = float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
rng = np.random.default_rng(seed)
errs, Js = [], []
for case in range(n_cases):
    n_real = int(rng.integers(0, 4))
    n_cplx = int(rng.integers(1, (6 - n_real) // 2 + 1))
    if n_real + 2 * n_cplx < 3:
        n_cplx += 1
    N = int(rng.integers(600, 6000))
    D = 32
    span = 10 ** rng.uniform(0, 3)
    t = np.sort(rng.uniform(0, span, N)) + 10 ** rng.uniform(0, 3.5)
    if rng.uniform() < 0.3:
        t[N // 2:] += span * rng.uniform(0.5, 20)
    dtm = span / N
    lc = np.log10(cmax)
    cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
    for d in range(D):
        for j in range(n_real):
            cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, lc) / dtm]
        for j in range(n_cplx):
            a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, lc) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
            b = rng.uniform(-1, 1) * a * c / dd
            if rng.uniform() < 0.5:
                b = np.sign(b) * min(abs(b), a * 10 ** rng.uniform(-1, 3))
            cc[d, j] = [a, b, c, dd]
    amp2 = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
    ba2 = ((cc[..., 1] / cc[..., 0]) ** 2).max(-1)
    kap = 10 ** rng.uniform(np.log10(klo), np.log10(khi), size=D)
    dmin = (1 + ba2) * amp2 / kap
    diag = dmin[:, None] * (1 + 0.3 * rng.uniform(size=(D, N)))
    diag[np.arange(D), rng.integers(0, N, D)] = dmin
    y = np.sqrt(amp2)[:, None] * rng.normal(size=(D, N))
    want = run(t, y, diag, cr, cc, 1)
    got = run(t, y, diag, cr, cc, 0)
    for d in range(D):
        if not np.isfinite(want[0][d]):
            continue
        e = 0.0
        for g, w in zip(got[1:], want[1:]):
            if w.size:
                e = max(e, np.abs(g[d] - w[d]).max() / (np.abs(w[d]).max() + 1e-300))
        errs.append(e); Js.append(n_real + 2 * n_cplx)
errs, Js = np.array(errs), np.array(Js)
print("kappa [%g, %g), c dt <= %g: %d draws; worst %.1e, 99.9 %% %.1e, 99 %% %.1e, median %.1e" % (
    klo, khi, cmax, len(errs), errs.max(), np.quantile(errs, 0.999), np.quantile(errs, 0.99), np.median(errs)))
for j in (3, 4, 5, 6):
    m = Js == j
    if m.any():
        print("  J = %d: %d draws, worst %.1e" % (j, m.sum(), errs[m].max()))
