import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P
from test_gpu_scan_filter import random_records
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(17)
D, Pn = 24, 3
rec = random_records(rng, D, Pn)
t = np.sort(np.concatenate([np.linspace(0, 60, 30000), 2000 + np.linspace(0, 20, 10000)]))
c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
for d in range(D):
    for p in range(Pn):
        r1 = rec[d:d+1, p:p+1]
        f1 = ops.transit_flux(T(t), T(r1), T(c[:1]))
        f2 = ops.transit_flux(T(t), T(r1), T(c[:1]), flags=ops.FLAG_EXACT_SCAN)
        bad = (f1 != f2)
        if bad.any():
            idx = bad.nonzero()[:, 1].cpu().numpy()
            print(f"draw {d} planet {p}: {bad.sum().item()} mismatches; e={r1[0,0,P.P_ECC]:.4f} aor={r1[0,0,P.P_AOR]:.2f} ror={r1[0,0,P.P_ROR]:.3f} ci={r1[0,0,P.P_COSI]:.5f} n={r1[0,0,P.P_N]:.3f}",
                  "t range", t[idx].min(), t[idx].max(), "fast", f1[0, idx[:3]].cpu().numpy(), "exact", f2[0, idx[:3]].cpu().numpy())
