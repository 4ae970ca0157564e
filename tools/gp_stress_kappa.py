import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
from exoplanet_amd.gp import celerite_loglike
T = lambda a, g=False: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(g)
def run(t, y, diag, cr, cc, chunks):
    if chunks is None: os.environ.pop("EXO_GP_CHUNKS", None)
    else: os.environ["EXO_GP_CHUNKS"] = str(chunks)
    ll = celerite_loglike(T(t), T(y), T(diag), T(cr), T(cc))
    return ll.cpu().numpy()
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
kap = (1 + (cc[:, :, 1] / cc[:, :, 0]).max(1) ** 2) * (cr[..., 0].sum(-1) + cc[..., 0].sum(-1)) / diag.min(1)
err = np.abs(seq - chk) / np.abs(seq)
for d in np.argsort(-err)[:6]:
    print("draw", d, "err %.1e kappa %.1e d*dt %.1f c*dt %.3f" % (err[d], kap[d], cc[d, 0, 3] * dtm, cc[d, 0, 2] * dtm))
