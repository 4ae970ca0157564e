"""Ad-hoc kernel timing for development (not the bench contract; see bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P

def main(D=64, N=150000, iters=20):
    dev = torch.device("cuda:0")
    orbit = P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1)
    rec = np.zeros((D, 1, P.NPAR))
    rng = np.random.default_rng(2)
    rec[:, 0, P.P_N] = orbit.n[0]; rec[:, 0, P.P_TP] = orbit.t_periastron[0]; rec[:, 0, P.P_ECC] = 0.3
    rec[:, 0, P.P_COSW] = np.cos(1.1); rec[:, 0, P.P_SINW] = np.sin(1.1)
    rec[:, 0, P.P_COSI] = orbit.cos_incl[0]; rec[:, 0, P.P_SINI] = orbit.sin_incl[0]
    rec[:, 0, P.P_AOR] = orbit.a[0]; rec[:, 0, P.P_ROR] = 0.1 * (1 + 1e-3 * rng.normal(size=D))
    rec[:, 0, P.P_T0] = 1.0; rec[:, 0, P.P_PERIOD] = 3.5; rec[:, 0, P.P_TS] = -np.inf; rec[:, 0, P.P_TE] = np.inf
    c = np.repeat(P.get_cl(0.3, 0.2)[None], D, 0)
    T = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    t = T(np.arange(N) * (2.0 / 1440.0)); rec_t = T(rec); c_t = T(c); g = torch.randn(D, N, dtype=torch.float64, device=dev)
    for name, fn in [("fwd", lambda: ops.transit_flux(t, rec_t, c_t)),
                     ("value+vjp", lambda: ops.transit_flux_value_and_vjp(t, rec_t, c_t, g))]:
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
        print(f"{name}: D={D} N={N} {dt*1e3:.3f} ms/step  {D/dt:.1f} evals/s  {24*D*N/dt/1e9:.1f} GB/s(alg,24B)")

if __name__ == "__main__":
    main(D=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
