"""Which draws does the time-parallel celerite path lose accuracy on?  Random kernels as tools/gp_stress.py draws them (J = 1..6,
time scales far below to far above the sampling, gaps, noise 1e-6..1 of the signal), library built with -DEXO_GP_COND_MAX=1e30
(nothing flagged): per draw the worst relative gradient disagreement with the sequential kernels, binned by the two factors of
the conditioning score -- max (b/a)^2 over the complex terms and sum(a) / min(diag) -- and by max c dt.
usage: EXOPLANET_AMD_LIB=<noflag .so> python tools/gp_cond_bins.py <seed> <cases>"""
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


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rows = []
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    n_real = int(rng.integers(0, 4))
    n_cplx = int(rng.integers(1, (6 - n_real) // 2 + 1))
    N = int(rng.integers(300, 4000))
    D = 32
    span = 10 ** rng.uniform(0, 3)
    t = np.sort(rng.uniform(0, span, N))
    if rng.uniform() < 0.3:
        t[N // 2:] += span * rng.uniform(0.5, 20)
    dtm = span / N
    cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
    for d in range(D):
        for j in range(n_real):
            cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 2) / dtm]
        for j in range(n_cplx):
            a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1.5) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
            b = rng.uniform(-1, 1) * a * c / dd
            if rng.uniform() < 0.5:
                b = np.sign(b) * min(abs(b), a * 10 ** rng.uniform(-1, 3))
            cc[d, j] = [a, b, c, dd]
    amp2 = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
    diag = (10 ** rng.uniform(-8, 0, size=(D, 1)) * amp2[:, None]) * (1 + 0.3 * rng.uniform(size=(D, N)))
    y = np.sqrt(amp2)[:, None] * rng.normal(size=(D, N))
    want = run(t, y, diag, cr, cc, 1)
    got = run(t, y, diag, cr, cc, 0)
    ba2 = ((cc[..., 1] / cc[..., 0]) ** 2).max(-1)
    snr = amp2 / diag.min(-1)
    cdt = np.maximum(cc[..., 2].max(-1), cr[..., 1].max(-1) if n_real else 0.0) * dtm
    for d in range(D):
        if not np.isfinite(want[0][d]):
            continue
        e = 0.0
        for g, w in zip(got[1:], want[1:]):
            if w.size:
                e = max(e, np.abs(g[d] - w[d]).max() / (np.abs(w[d]).max() + 1e-300))
        dmed = np.median(np.diff(t))
        cs = np.concatenate([cc[d, :, 2], cr[d, :, 1]]) if n_real else cc[d, :, 2]
        rows.append((ba2[d], snr[d], cdt[d], e, n_real + 2 * n_cplx, cs.min() * dtm, cc[d, :, 3].max() * dtm, cc[d, :, 3].min() * dtm,
                     np.diff(t).max() / dtm, abs(got[0][d] - want[0][d]) / abs(want[0][d])))
rows = np.array(rows)
print("draws", len(rows))
print("rows: log10 max (b/a)^2; columns: log10 sum(a) / min(diag);  entries: max grad error (count)")
bb = [-3, 0, 1, 2, 3, 4, 5, 6, 8]
ss = [0, 2, 3, 4, 5, 6, 7, 8, 9]
print("%10s" % "" + "".join("%16s" % f"[{ss[i]},{ss[i + 1]})" for i in range(len(ss) - 1)))
for i in range(len(bb) - 1):
    line = "%10s" % f"[{bb[i]},{bb[i + 1]})"
    for j in range(len(ss) - 1):
        m = (rows[:, 0] >= 10.0 ** bb[i]) & (rows[:, 0] < 10.0 ** bb[i + 1]) & (rows[:, 1] >= 10.0 ** ss[j]) & (rows[:, 1] < 10.0 ** ss[j + 1])
        line += "%16s" % (f"{rows[m, 3].max():.0e} ({m.sum()})" if m.any() else "-")
    print(line)
for lo, hi in ((0, 1), (1, 3), (3, 10), (10, 100)):
    m = (rows[:, 2] >= lo) & (rows[:, 2] < hi)
    if m.any():
        print(f"max c dt in [{lo},{hi}): n={m.sum()} max err {rows[m, 3].max():.1e} median {np.median(rows[m, 3]):.1e}")
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gp_cond_bins.npy"), rows)
