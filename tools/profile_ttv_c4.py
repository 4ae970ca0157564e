"""C4 shape (4 planets, 200 000 cadences, 64 draws) with and without per-draw timing tables:
op-level value + gradient times.   python tools/profile_ttv_c4.py [draws]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P
from test_gpu_transit import make_record
from tools.bench_configs import timeit

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(4)
t4 = np.arange(200_000) * (2.0 / 1440.0)
periods, t0s = np.array([3.5, 7.9, 13.1, 29.7]), np.array([1.0, 2.3, 5.1, 11.7])
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
recs, edges, shifts = [], [], []
for d in range(D):
    ttvs = [0.01 * rng.normal(size=int((t4[-1] - a) / p) + 1) for p, a in zip(periods, t0s)]
    orbit = P.TTVOrbit(period=periods, t0=t0s, b=np.array([0.3, 0.1, 0.5, 0.2]), ecc=np.array([0.05, 0.1, 0.2, 0.3]),
                       omega=np.array([1.1, -0.4, 2.0, 0.3]), ttvs=ttvs)
    recs.append(make_record(orbit, np.array([0.1, 0.05, 0.07, 0.03]))[0])
    e, s = orbit.kernel_tables()
    edges.append(e); shifts.append(s)
rec = np.stack(recs) * (1 + 1e-4 * rng.normal(size=(D, 4, P.NPAR)))
c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
r4, c4, g4, tt = T(rec), T(c), torch.randn(D, t4.size, dtype=torch.float64, device=dev), T(t4)
ttv = (T(np.stack(edges)), T(np.stack(shifts)))
plain = timeit(lambda: ops.transit_flux_value_and_vjp(tt, r4, c4, g4), 20)
with_ttv = timeit(lambda: ops.transit_flux_value_and_vjp(tt, r4, c4, g4, ttv=ttv), 20)
print(json.dumps({"draws": D, "planets": 4, "n_cad": t4.size, "transits": [int(x.shape[0]) for x in ttvs],
                  "value_grad_ms": {"keplerian": plain * 1e3, "ttv": with_ttv * 1e3},
                  "ttv_evals_per_s": D / with_ttv}))
