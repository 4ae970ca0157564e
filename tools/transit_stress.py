"""Randomised stress of the fused light-curve kernels against the oracle's C port: random batch
shapes, planet counts, geometries (e to 0.97, a/R from 2 to 500, grazing), exposure integration,
secondary eclipses, per-planet output, caller windows.  Prints the worst disagreements."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from exoplanet_amd import ops
from oracle import numpy_port as P, c_port as C
from test_gpu_transit import make_record

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = dict(flux=0.0, gp=0.0, gl=0.0)
for case in range(n_cases):
    D, Pn = int(rng.integers(1, 12)), int(rng.integers(1, 5))
    N = int(rng.integers(50, 6000))
    secondary, per_planet, window = rng.uniform() < 0.3, rng.uniform() < 0.4, rng.uniform() < 0.3
    use_texp = rng.uniform() < 0.4
    span = 10 ** rng.uniform(0.5, 2.5)
    t = np.sort(rng.uniform(0, span, N)) + (2.45e6 if rng.uniform() < 0.2 else 0.0)
    rec = np.zeros((D, Pn, P.NPAR))
    for d in range(D):
        period = 10 ** rng.uniform(-0.3, 1.5, Pn)
        ecc = np.where(rng.uniform(size=Pn) < 0.2, 0.0, rng.uniform(0, 0.6 if window else 0.97, Pn))
        omega = rng.uniform(-np.pi, np.pi, Pn)
        a = 10 ** rng.uniform(0.8 if window else 0.3, 2.7, Pn)
        b = rng.uniform(0, 0.9 if window else 1.25, Pn)   # (the test helper wants solvable contact points)
        cosi = np.clip((1 + ecc * np.sin(omega)) / (1 - ecc ** 2) * b / a, 0, 0.999)
        orbit = P.KeplerianOrbit(period=period, a=a, t0=t[0] + rng.uniform(0, 5, Pn), incl=np.arccos(cosi), ecc=ecc, omega=omega)
        rr = 10 ** rng.uniform(-2, -0.5, Pn)
        try:
            rec[d] = make_record(orbit, rr, sbr=0.3, window=window)[0]
        except AssertionError:      # contact points not solvable: the product gives such records an infinite window
            rec[d] = make_record(orbit, rr, sbr=0.3, window=False)[0]
            if window:
                rec[d][:, P.P_TS] = -np.inf; rec[d][:, P.P_TE] = np.inf
                rec[d][:, P.P_TS2] = -np.inf; rec[d][:, P.P_TE2] = np.inf
    c = np.repeat(np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None], D, 0)
    c = c if secondary else c[:, :3]
    shape = (D, N, Pn) if per_planet else (D, N)
    g = rng.normal(size=shape)
    kw, ckw = {}, {}
    if use_texp:
        sdt, sw = P.exposure_stencil(int(rng.choice([3, 5, 7])), int(rng.integers(0, 3)))
        te = 10 ** rng.uniform(-3, -1.3)
        kw = dict(texp=T([te]), stencil_dt=T(sdt), stencil_w=T(sw))
        ckw = dict(texp=te, stencil_dt=sdt, stencil_w=sw)
    flags = (ops.FLAG_SECONDARY if secondary else 0) | (ops.FLAG_PER_PLANET if per_planet else 0) | (ops.FLAG_WINDOW if window else 0)
    f, gp, gl = ops.transit_flux_value_and_vjp(T(t), T(rec), T(c), T(g), flags=flags, **kw)
    wf, wgp, wgl = C.transit(t, rec, c, g, per_planet=per_planet, window=window, secondary=secondary, **ckw)
    ef = float(np.abs(f.cpu().numpy() - wf).max())
    sc = np.abs(wgp).max(axis=(0, 1), keepdims=True) + 1e-300
    egp = float((np.abs(gp.cpu().numpy() - wgp) / sc).max())
    egl = float(np.abs(gl.cpu().numpy() - wgl).max() / (np.abs(wgl).max() + 1e-300))
    worst["flux"] = max(worst["flux"], ef); worst["gp"] = max(worst["gp"], egp); worst["gl"] = max(worst["gl"], egl)
    if ef > 1e-12 or egp > 1e-8 or egl > 1e-8:
        print(f"case {case}: D={D} P={Pn} N={N} sec={secondary} pp={per_planet} win={window} texp={use_texp}: flux {ef:.1e} gparams {egp:.1e} gld {egl:.1e}")
print("worst |dflux| %.2e, gparams rel %.2e, gld rel %.2e over %d cases" % (worst["flux"], worst["gp"], worst["gl"], n_cases))
