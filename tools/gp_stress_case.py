"""one case of tools/gp_stress.py against the dense oracle, draw by draw"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import numpy_port as P
import importlib.util
spec = importlib.util.spec_from_file_location("gs", os.path.join(os.path.dirname(__file__), "gp_stress.py"))

seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
from exoplanet_amd.gp import celerite_loglike
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)
def run(t, y, diag, cr, cc, chunks):
    if chunks is None: os.environ.pop("EXO_GP_CHUNKS", None)
    else: os.environ["EXO_GP_CHUNKS"] = str(chunks)
    yt, dt, crt, cct = T(y, True), T(diag, True), T(cr, True), T(cc, True)
    ll = celerite_loglike(T(t), yt, dt, crt, cct)
    ll.sum().backward()
    return [x.detach().cpu().numpy() for x in (ll, yt.grad, dt.grad, crt.grad, cct.grad)]
for case in range(target + 1):
    n_real = int(rng.integers(0, 4))
    n_cplx = int(rng.integers(0 if n_real else 1, (6 - n_real) // 2 + 1))
    N = int(rng.integers(70, 4000)); D = int(rng.integers(1, 40))
    span = 10 ** rng.uniform(0, 3)
    t = np.sort(rng.uniform(0, span, N))
    if rng.uniform() < 0.3: t[N // 2:] += span * rng.uniform(0.5, 20)
    dtm = span / N
    cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
    for d in range(D):
        for j in range(n_real): cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 2) / dtm]
        for j in range(n_cplx):
            a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1.5) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
            b = rng.uniform(-1, 1) * a * c / dd
            cc[d, j] = [a, b, c, dd]
    amp = np.sqrt(cr[..., 0].sum(-1) + cc[..., 0].sum(-1))
    diag = (10 ** rng.uniform(-6, 0, size=(D, 1)) * amp[:, None] ** 2) * (1 + 0.3 * rng.uniform(size=(D, N)))
    y = amp[:, None] * rng.normal(size=(D, N))
    rng.integers(2, 60)
seq = run(t, y, diag, cr, cc, 1)
chk = run(t, y, diag, cr, cc, None)
print("case", target, "J", n_real, n_cplx, "N", N, "D", D, "dtm", dtm)
for d in range(D):
    co = (cr[d, :, 0], cr[d, :, 1], cc[d, :, 0], cc[d, :, 1], cc[d, :, 2], cc[d, :, 3])
    ref, g = P.gp_loglike_dense(t, y[d], diag[d], co)
    def err(x):
        gy = np.abs(x[1][d] - g["y"]).max() / np.abs(g["y"]).max()
        gc = max([abs(x[4][d][j, k] - g[nm][j]) / (abs(g[nm][j]) + 1e-300) for j in range(n_cplx) for k, nm in enumerate(("ac", "bc", "cc", "dc"))] or [0])
        return abs(x[0][d] - ref) / abs(ref), gy, gc
    es, ec = err(seq), err(chk)
    if max(es + ec) > 1e-7:
        print(f"draw {d}: noise/amp^2 {diag[d].mean()/amp[d]**2:.1e} c dt {cc[d,:,2]*dtm} d dt {cc[d,:,3]*dtm} b/a {cc[d,:,1]/cc[d,:,0]}\n   sequential (ll, gy, gcoef) {es[0]:.1e} {es[1]:.1e} {es[2]:.1e}   chunked {ec[0]:.1e} {ec[1]:.1e} {ec[2]:.1e}")
