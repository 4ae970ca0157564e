"""chunked-vs-sequential gradient error per draw against a conditioning score: calibrates the
device-side flag threshold (kCondMax).  KAPPA_PLAIN=1: kappa = sum(a) / min(diag) (the whitened
basis); default: with the (1 + max (b/a)^2) factor the rotating-frame version needed.
EXO_GP_KCOND (if the library is built to read it) is not used: run with a library whose kCondMax
is large to see the unflagged errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exoplanet_amd.gp import celerite_loglike
dev = torch.device("cuda:0")
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)
def run(t, y, diag, cr, cc, chunks):
    if chunks is None: os.environ.pop("EXO_GP_CHUNKS", None)
    else: os.environ["EXO_GP_CHUNKS"] = str(chunks)
    yt, dt, crt, cct = T(y, True), T(diag, True), T(cr, True), T(cc, True)
    ll = celerite_loglike(T(t), yt, dt, crt, cct)
    torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll)).sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, crt.grad, cct.grad)]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rows = []
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n_real = int(rng.integers(0, 3)); n_cplx = int(rng.integers(1, 3))
    N = int(rng.integers(200, 3000)); D = 32
    span = 10 ** rng.uniform(0, 2); t = np.sort(rng.uniform(0, span, N)); dtm = span / N
    cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
    for d in range(D):
        for j in range(n_real): cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 1) / dtm]
        for j in range(n_cplx):
            a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1) / dtm; dd = 10 ** rng.uniform(-2, 1) / dtm
            ba = rng.choice([-1, 1]) * 10 ** rng.uniform(-2, 2.5)
            ba = np.sign(ba) * min(abs(ba), c / dd)            # valid kernel: |b d| <= a c
            cc[d, j] = [a, ba * a, c, dd]
    amp2 = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
    diag = (10 ** rng.uniform(-7, 0, size=(D, 1)) * amp2[:, None]) * (1 + 0.3 * rng.uniform(size=(D, N)))
    y = np.sqrt(amp2)[:, None] * rng.normal(size=(D, N))
    want = run(t, y, diag, cr, cc, 1)   # n_chunks = 1: sequential recurrences; got = run(t, y, diag, cr, cc, None)
    kappa = amp2 / diag.min(-1) * (1.0 if os.environ.get("KAPPA_PLAIN") else (1 + ((cc[..., 1] / cc[..., 0]) ** 2).max(-1)))
    for d in range(D):
        if not np.isfinite(want[0][d]): continue
        e = 0.0
        for g, w in zip(got[1:], want[1:]):
            if w.size: e = max(e, np.abs(g[d] - w[d]).max() / (np.abs(w[d]).max() + 1e-300))
        rows.append((kappa[d], e, abs(got[0][d] - want[0][d]) / abs(want[0][d])))
rows = np.array(rows)
for lo in range(0, 14):
    m = (rows[:, 0] >= 10.0 ** lo) & (rows[:, 0] < 10.0 ** (lo + 1))
    if m.any(): print(f"kappa 1e{lo}..1e{lo+1}: n={m.sum():4d}  max grad err {rows[m,1].max():.1e}  median {np.median(rows[m,1]):.1e}  max ll err {rows[m,2].max():.1e}")
