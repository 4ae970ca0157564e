"""C4 leg (4 planets, 200 000 cadences, 64 draws) under rocprofv3: kernel times of the fused sweep"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P
from test_gpu_transit import make_record
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(4)
t4 = np.arange(200_000) * (2.0 / 1440.0)
orbit4 = P.KeplerianOrbit(period=np.array([3.5, 7.9, 13.1, 29.7]), t0=np.array([1.0, 2.3, 5.1, 11.7]),
                          b=np.array([0.3, 0.1, 0.5, 0.2]), ecc=np.array([0.05, 0.1, 0.2, 0.3]),
                          omega=np.array([1.1, -0.4, 2.0, 0.3]))
rec4 = make_record(orbit4, np.array([0.1, 0.05, 0.07, 0.03]), window=True)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rec = np.repeat(rec4, D, 0) * (1 + 1e-4 * rng.normal(size=(D,) + rec4.shape[1:]))
c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
r4, c4, g4, tt = T(rec), T(c), torch.randn(D, t4.size, dtype=torch.float64, device=dev), T(t4)
for _ in range(5):
    ops.transit_flux_value_and_vjp(tt, r4, c4, g4)
    ops.transit_flux_value_and_vjp(tt, r4, c4, g4, flags=ops.FLAG_WINDOW)
torch.cuda.synchronize()
