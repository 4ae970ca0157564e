import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exoplanet_amd import _lib
from oracle import numpy_port as P
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
for e0 in (0.0, 0.3, 0.9):
    M = np.array([0.1, 0.5, 1.0, 1.5, 1.6, 2.0, 2.5, 3.0, 3.1, 3.14, -0.5, -3.0, 6.0, 7.0])
    e = np.full(M.size, e0)
    E, _ = P.kepler_E(M, e)
    cx = torch.empty(M.size, dtype=torch.float64, device=dev); sx = torch.empty_like(cx)
    lib = _lib.load()
    lib.exo_selftest_orbit_pos_f32(T(M).data_ptr(), T(e).data_ptr(), cx.data_ptr(), sx.data_ptr(), M.size, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("e", e0)
    for m, a, b, c, d in zip(M, cx.cpu().numpy(), sx.cpu().numpy(), np.cos(E) - e, np.sqrt(1 - e * e) * np.sin(E)):
        print(f"  M={m:6.2f} got ({a:+.6f},{b:+.6f}) want ({c:+.6f},{d:+.6f})")
