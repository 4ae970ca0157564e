import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exoplanet_amd import _lib
from oracle import numpy_port as P
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(23)
for name, e, M in [("e0", np.zeros(100000), rng.uniform(-np.pi, np.pi, 100000)),
                   ("mid", rng.uniform(0, 0.9, 100000), rng.uniform(-np.pi, np.pi, 100000)),
                   ("bigM", rng.uniform(0, 0.9, 100000), rng.uniform(-3e4, 3e4, 100000)),
                   ("hi_e", 1 - 10 ** rng.uniform(-3, -1, 100000), rng.uniform(-np.pi, np.pi, 100000)),
                   ("smallM", rng.uniform(0, 0.99, 100000), 10 ** rng.uniform(-8, 0.5, 100000) * rng.choice([-1, 1], 100000))]:
    n = e.size
    E, _ = P.kepler_E(M, e)
    wcx, wsx = np.cos(E) - e, np.sqrt(1 - e * e) * np.sin(E)
    cx = torch.empty(n, dtype=torch.float64, device=dev); sx = torch.empty(n, dtype=torch.float64, device=dev)
    lib = _lib.load()
    Mt, et = T(M), T(e)
    lib.exo_selftest_orbit_pos_f32(Mt.data_ptr(), et.data_ptr(), cx.data_ptr(), sx.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    err = np.maximum(np.abs(cx.cpu().numpy() - wcx), np.abs(sx.cpu().numpy() - wsx))
    bound = 4e-6 + 2e-6 / (1 - e)
    i = np.argmax(err / bound)
    print(name, "worst ratio", (err / bound).max(), "M", M[i], "e", e[i], "got", cx[i].item(), sx[i].item(), "want", wcx[i], wsx[i], "nan", np.isnan(err).sum())
