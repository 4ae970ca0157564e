"""Randomised stress of the time-parallel celerite path against the sequential kernels:
random term mixes (J = 1..6), time scales from far below to far above the sampling, gaps,
per-draw noise levels spanning six decades.  Prints the worst disagreements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exoplanet_amd.gp import celerite_loglike

dev = torch.device("cuda:0")
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)


def run(t, y, diag, cr, cc, chunks):
    if chunks is None:
        os.environ.pop("EXO_GP_CHUNKS", None)
    else:
        os.environ["EXO_GP_CHUNKS"] = str(chunks)
    yt, dt, crt, cct = T(y, True), T(diag, True), T(cr, True), T(cc, True)
    ll = celerite_loglike(T(t), yt, dt, crt, cct)
    w = torch.linspace(0.5, 1.5, ll.numel(), dtype=torch.float64, device=dev)
    (torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll)) * w).sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, crt.grad, cct.grad)]


rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {"ll": 0.0, "g": 0.0}
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(n_cases):
    n_real = int(rng.integers(0, 4))
    n_cplx = int(rng.integers(0 if n_real else 1, (6 - n_real) // 2 + 1))
    N = int(rng.integers(70, 4000))
    D = int(rng.integers(1, 40))
    span = 10 ** rng.uniform(0, 3)
    t = np.sort(rng.uniform(0, span, N))
    if rng.uniform() < 0.3:
        t[N // 2:] += span * rng.uniform(0.5, 20)          # a long gap
    dtm = span / N
    cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
    for d in range(D):
        for j in range(n_real):
            cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 2) / dtm]
        for j in range(n_cplx):
            a = 10 ** rng.uniform(-2, 1)
            c = 10 ** rng.uniform(-3, 1.5) / dtm
            dd = 10 ** rng.uniform(-2, 1.5) / dtm
            b = rng.uniform(-1, 1) * a * c / dd
            cc[d, j] = [a, b, c, dd]
    amp = np.sqrt(cr[..., 0].sum(-1) + cc[..., 0].sum(-1))
    diag = (10 ** rng.uniform(-6, 0, size=(D, 1)) * amp[:, None] ** 2) * (1 + 0.3 * rng.uniform(size=(D, N)))
    y = amp[:, None] * rng.normal(size=(D, N))
    want = run(t, y, diag, cr, cc, 1)   # n_chunks = 1: sequential recurrences
    for chunks in (None, int(rng.integers(2, 60))):
        got = run(t, y, diag, cr, cc, chunks)
        ok = np.isfinite(want[0])
        assert (np.isfinite(got[0]) == ok).all(), (case, chunks, "finiteness differs")
        e_ll = np.max(np.abs(got[0][ok] - want[0][ok]) / np.abs(want[0][ok])) if ok.any() else 0.0
        e_g = 0.0
        for g, w in zip(got[1:], want[1:]):
            if w.size:
                gg, ww = g[ok], w[ok]
                if ww.size:
                    scale = np.abs(ww).reshape(ww.shape[0], -1).max(1) + 1e-300
                    e_g = max(e_g, float((np.abs(gg - ww).reshape(ww.shape[0], -1).max(1) / scale).max()))
        worst["ll"] = max(worst["ll"], e_ll); worst["g"] = max(worst["g"], e_g)
        if e_ll > 1e-10 or e_g > 1e-6:
            print(f"case {case} chunks {chunks}: J=({n_real},{n_cplx}) N={N} D={D} span={span:.3g} ll err {e_ll:.2e} grad err {e_g:.2e}")
print("worst loglike rel err %.2e, worst grad rel err %.2e over %d cases" % (worst["ll"], worst["g"], n_cases))
