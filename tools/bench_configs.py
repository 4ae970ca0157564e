"""First numbers for BASELINE.json configs C1..C5.
Not the driver's bench contract (that is bench.py = C2); run on the GPU box:
    python tools/bench_configs.py > gpurun_out/configs.json
Each entry: evaluations/s (value+grad) on 1 GPU at the config's per-GPU draw count,
and the 1-core CPU port (oracle/c) beside it."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402
from exoplanet_amd.gp import celerite_loglike  # noqa: E402
from oracle import c_port as C  # noqa: E402
from oracle import numpy_port as P  # noqa: E402
from test_gpu_transit import make_record  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def cpu_time(fn, min_s=3.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn(); n += 1
    return (time.perf_counter() - t0) / n


def perturb(rec, D, rng):
    recs = np.repeat(rec, D, axis=0)
    for slot in (P.P_ROR, P.P_AOR, P.P_COSI):
        recs[:, :, slot] *= 1 + 1e-3 * rng.normal(size=recs.shape[:2])
    return recs


def main():
    out = {}
    rng = np.random.default_rng(1)
    # ---- C1 (CPU reference case only)
    t = np.arange(10_000) * (2.0 / 1440.0)
    rec = make_record(P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3), np.array([0.1]))
    c = P.get_cl(0.3, 0.2)[None]
    g = rng.normal(size=(1, t.size))
    out["C1"] = {"cpu_1core_evals_per_s": 1 / cpu_time(lambda: C.transit(t, rec, c, g)), "n_cad": 10_000}
    # ---- C2 (bench.py is authoritative; op-level here for the table)
    t = np.arange(150_000) * (2.0 / 1440.0)
    rec1 = make_record(P.KeplerianOrbit(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1), np.array([0.1]), window=True)
    D = 1024
    recs = perturb(rec1, D, rng)
    cs = np.repeat(c, D, 0)
    tt, rt, ct, gt = T(t), T(recs), T(cs), torch.randn(D, t.size, dtype=torch.float64, device=dev)
    dt_all = timeit(lambda: ops.transit_flux_value_and_vjp(tt, rt, ct, gt), 20)
    dt_win = timeit(lambda: ops.transit_flux_value_and_vjp(tt, rt, ct, gt, flags=ops.FLAG_WINDOW), 20)
    g1 = rng.normal(size=(1, t.size))
    out["C2"] = {"draws_per_gpu": D, "gpu_evals_per_s_every_cadence": D / dt_all,
                 "gpu_evals_per_s_in_transit_only": D / dt_win,
                 "alg_GBps_every_cadence": 24 * D * t.size / dt_all / 1e9,
                 "alg_GBps_in_transit_only": 24 * D * t.size / dt_win / 1e9,
                 "cpu_1core_evals_per_s_every_cadence": 1 / cpu_time(lambda: C.transit(t, rec1, c, g1)),
                 "cpu_1core_evals_per_s_in_transit_only": 1 / cpu_time(lambda: C.transit(t, rec1, c, g1, window=True))}
    # ---- C3 = C2 + SHO GP: flux -> resid -> loglike, backward through both kernels
    D3 = 1024
    co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3, 5.0, 1 / np.sqrt(2)), 1 / np.sqrt(2))
    cplx = T(np.repeat(np.stack(co[2:], -1)[None], D3, 0))
    real = T(np.zeros((D3, 0, 2)))
    y = T(5e-4 * rng.normal(size=t.size))
    diag = T(np.full((1, t.size), 2.5e-7))

    def c3():
        r = rt[:D3].clone().requires_grad_(True)
        cc = cplx.clone().requires_grad_(True)
        f = ops.transit_flux(tt, r, ct[:D3], flags=ops.FLAG_WINDOW)
        ll = celerite_loglike(tt, f, diag, real, cc, obs=y)      # obs - model formed inside the GP kernels
        torch.autograd.grad(ll.sum(), (r, cc))

    dt3 = timeit(c3, 5)
    f1, _, _ = C.transit(t, rec1, c, None, window=True)
    yy = 5e-4 * rng.normal(size=t.size)

    def c3_cpu():
        f, _, _ = C.transit(t, rec1, c, None, window=True)
        ll, gw = C.celerite(t, yy - f[0], np.full(t.size, 2.5e-7), co, grad=True)
        C.transit(t, rec1, c, -gw["y"][None], window=True, want_flux=False)

    J = 2
    out["C3"] = {"draws_per_gpu": D3, "gpu_evals_per_s": D3 / dt3, "ms_per_step": 1e3 * dt3,
                 "alg_GBps": (48 + 16 * (1 + J + J * J)) * D3 * t.size / dt3 / 1e9,
                 "cpu_1core_evals_per_s": 1 / cpu_time(c3_cpu)}
    # ---- C4: 4 planets, N = 200 000, 64 draws per GPU
    t4 = np.arange(200_000) * (2.0 / 1440.0)
    orbit4 = P.KeplerianOrbit(period=np.array([3.5, 7.9, 13.1, 29.7]), t0=np.array([1.0, 2.3, 5.1, 11.7]),
                              b=np.array([0.3, 0.1, 0.5, 0.2]), ecc=np.array([0.05, 0.1, 0.2, 0.3]),
                              omega=np.array([1.1, -0.4, 2.0, 0.3]))
    rec4 = make_record(orbit4, np.array([0.1, 0.05, 0.07, 0.03]), window=True)
    D4 = 64
    r4, c4, g4 = T(perturb(rec4, D4, rng)), T(np.repeat(c, D4, 0)), torch.randn(D4, t4.size, dtype=torch.float64, device=dev)
    t4t = T(t4)
    dt4 = timeit(lambda: ops.transit_flux_value_and_vjp(t4t, r4, c4, g4), 20)
    dt4w = timeit(lambda: ops.transit_flux_value_and_vjp(t4t, r4, c4, g4, flags=ops.FLAG_WINDOW), 20)
    g4c = rng.normal(size=(1, t4.size))
    out["C4"] = {"draws_per_gpu": D4, "gpu_evals_per_s_every_cadence": D4 / dt4, "gpu_evals_per_s_in_transit_only": D4 / dt4w,
                 "alg_GBps_every_cadence": 24 * D4 * t4.size / dt4 / 1e9,
                 "cpu_1core_evals_per_s_every_cadence": 1 / cpu_time(lambda: C.transit(t4, rec4, c, g4c)),
                 "cpu_1core_evals_per_s_in_transit_only": 1 / cpu_time(lambda: C.transit(t4, rec4, c, g4c, window=True))}
    # ---- C5: long cadence, secondary eclipse, 3-term GP, 128 chains per GPU
    t5 = np.arange(65_000) * (29.4 / 1440.0)
    texp = 29.4 / 1440.0
    orbit5 = P.KeplerianOrbit(period=2.7, t0=0.4, ecc=0.1, omega=0.7, b=0.2)
    D5 = 128
    xo5 = xo.KeplerianOrbit(period=2.7, t0=0.4, ecc=0.1, omega=0.7, b=0.2)
    rec5_t, _ = xo5.kernel_records(0.08, use_in_transit=True, secondary_sbr=0.3)
    rec5 = rec5_t.cpu().numpy()
    r5 = T(perturb(rec5, D5, rng))
    c6 = np.concatenate([P.get_cl(0.3, 0.2), P.get_cl(0.4, 0.1)])[None]
    c5 = T(np.repeat(c6, D5, 0))
    sdt, sw = P.exposure_stencil(7, 0)
    parts = [P.sho_coefficients(*P.sho_from_sigma_rho(s, r, q), q) for s, r, q in
             ((4e-4, 20.0, 2.0), (3e-4, 10.0, 1.0), (2e-4, 2.0, 1 / np.sqrt(2)))]
    co5 = tuple(np.concatenate(x) for x in zip(*parts))
    cplx5 = T(np.repeat(np.stack(co5[2:], -1)[None], D5, 0))
    real5 = T(np.zeros((D5, 0, 2)))
    t5t, y5, diag5 = T(t5), T(3e-4 * rng.normal(size=t5.size)), T(np.full((1, t5.size), 9e-8))
    kw5 = dict(texp=T([texp]), stencil_dt=T(sdt), stencil_w=T(sw), flags=ops.FLAG_SECONDARY | ops.FLAG_WINDOW)

    def c5f():
        r = r5.clone().requires_grad_(True)
        cc = cplx5.clone().requires_grad_(True)
        f = ops.transit_flux(t5t, r, c5, **kw5)
        ll = celerite_loglike(t5t, f, diag5, real5, cc, obs=y5)
        torch.autograd.grad(ll.sum(), (r, cc))

    dt5 = timeit(c5f, 5)
    yy5 = 3e-4 * rng.normal(size=t5.size)

    def c5_cpu():
        f, _, _ = C.transit(t5, rec5, c6, None, texp=texp, stencil_dt=sdt, stencil_w=sw, secondary=True, window=True)
        ll, gw = C.celerite(t5, yy5 - f[0], np.full(t5.size, 9e-8), co5, grad=True)
        C.transit(t5, rec5, c6, -gw["y"][None], texp=texp, stencil_dt=sdt, stencil_w=sw, secondary=True, window=True,
                  want_flux=False)

    J = 6
    out["C5"] = {"draws_per_gpu": D5, "gpu_evals_per_s": D5 / dt5, "ms_per_step": 1e3 * dt5,
                 "alg_GBps": (48 + 16 * (1 + J + J * J)) * D5 * t5.size / dt5 / 1e9,
                 "cpu_1core_evals_per_s": 1 / cpu_time(c5_cpu)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
