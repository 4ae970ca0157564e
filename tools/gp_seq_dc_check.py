"""GPU: the sequential kernels' and the default path's gradients against the long-double dense definition on the draws of
tools/gp_lab_robust.py (seed 2) where the sequential recurrences of the C port were off by 1e-6 .. 2e-5 in d/d(oscillation rate)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np, torch
from oracle import c_port as C
from oracle.make_golden_r02 import gp_dense_ld
from exoplanet_amd.gp import celerite_loglike

WANT = {(118, 0), (85, 3), (284, 5), (202, 6), (1, 4)}
NAMES = ("y", "diag", "ar", "cr", "ac", "bc", "cc", "dc")
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)


def cases(seed, n_cases):   # (tools/gp_host_lab.py's generator)
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        n_real = int(rng.integers(0, 4)); n_cplx = int(rng.integers(1, (6 - n_real) // 2 + 1))
        N = int(rng.integers(300, 1500)); D = 8
        span = 10 ** rng.uniform(0, 3); t = np.sort(rng.uniform(0, span, N))
        if rng.uniform() < 0.3: t[N // 2:] += span * rng.uniform(0.5, 20)
        dtm = span / N
        cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
        for d in range(D):
            for j in range(n_real): cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 2) / dtm]
            for j in range(n_cplx):
                a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1.5) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
                b = rng.uniform(-1, 1) * a * c / dd
                if rng.uniform() < 0.5: b = np.sign(b) * min(abs(b), a * 10 ** rng.uniform(-1, 3))
                cc[d, j] = [a, b, c, dd]
        amp2 = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
        diag = (10 ** rng.uniform(-8, 0, size=(D, 1)) * amp2[:, None]) * (1 + 0.3 * rng.uniform(size=(D, N)))
        y = np.sqrt(amp2)[:, None] * rng.normal(size=(D, N))
        if n_real + 2 * n_cplx > 6: continue
        yield case, t, y, diag, cr, cc


def err(got, want):
    return max(np.abs(got[k] - want[k]).max() / (np.abs(want[k]).max() + 1e-300) for k in NAMES if want[k].size)


for ci, (case, t, y, diag, cr, cc) in enumerate(cases(2, 300)):
    for d in range(8):
        if (ci, d) not in WANT: continue
        co = (cr[d, :, 0], cr[d, :, 1], cc[d, :, 0], cc[d, :, 1], cc[d, :, 2], cc[d, :, 3])
        ll, gt = gp_dense_ld(t, y[d], diag[d], co)
        truth = {k: np.asarray(gt[k], dtype=float) for k in NAMES}
        _, wg = C.celerite(t, y[d], diag[d], co, grad=True)
        out = {}
        for nm, chunks in (("sequential kernels", 1), ("default path", None)):
            yt, dt, rt, ct = (T(a[d:d + 1]).requires_grad_(True) for a in (y, diag, cr, cc))
            celerite_loglike(T(t), yt, dt, rt, ct, n_chunks=chunks).sum().backward()
            g = {"y": yt.grad[0], "diag": dt.grad[0], "ar": rt.grad[0, :, 0], "cr": rt.grad[0, :, 1], "ac": ct.grad[0, :, 0],
                 "bc": ct.grad[0, :, 1], "cc": ct.grad[0, :, 2], "dc": ct.grad[0, :, 3]}
            out[nm] = err({k: v.cpu().numpy() for k, v in g.items()}, truth)
        print("case %d draw %d N %d J %d | vs long double: C port %.1e, sequential kernels %.1e, default path %.1e" % (
            ci, d, t.size, cr.shape[1] + 2 * cc.shape[1], err(wg, truth), out["sequential kernels"], out["default path"]))
