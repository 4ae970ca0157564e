#!/usr/bin/env python
"""Op-level timing of the light-curve sweeps for the library selected by EXOPLANET_AMD_LIB (A/B of kernel variants):
C2 at 1024 / 128 draws (dense value + VJP, sparse value + VJP, white-noise misfit + gradient, value only), C4 at 64
draws, C5's light curve at 128 chains.  Prints one JSON line: medians in microseconds + checksums of the results (so
that a variant that changes a result shows)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
CAD = 2.0 / 1440.0


def med(fn, iters=60, warm=5):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)])) * 1e3


def c2_inputs(D, seed=100):
    rng = np.random.default_rng(seed)
    base = dict(period=3.5, t0=1.0, b=0.3, ecc=0.3, omega=1.1, r=0.1)
    L = {k: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D, 1))), dtype=torch.float64, device=dev) for k, v in base.items()}
    orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"])
    rec, ld, _, flags = orbit.kernel_inputs(L["r"], (0.3, 0.2), use_in_transit=False)
    return rec.detach().contiguous(), ld.detach().contiguous(), flags


out = {"lib": os.environ.get("EXOPLANET_AMD_LIB", "product")}
QUICK = "--quick" in sys.argv      # C2 at 1024 draws only (for alternating A/B/A/B runs: boxes and clocks drift)
N = 150_000
t = torch.arange(N, dtype=torch.float64, device=dev) * CAD
for D in ((1024,) if QUICK else (1024, 128)):
    rec, ld, flags = c2_inputs(D)
    g = torch.randn(D, N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    obs = 1e-4 * torch.randn(N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    w = torch.tensor([1e8], dtype=torch.float64, device=dev)
    tag = f"c2_{D}"
    out[tag + "_dense_vjp"] = med(lambda: ops.transit_flux_value_and_vjp(t, rec, ld, g, flags=flags))
    out[tag + "_sparse_vjp"] = med(lambda: ops.transit_flux_sparse(t, rec, ld, gflux=g, flags=flags))
    out[tag + "_chi2"] = med(lambda: ops.transit_chi2(t, rec, ld, obs, w, flags=flags))
    with torch.no_grad():
        out[tag + "_value"] = med(lambda: ops.transit_flux(t, rec, ld, flags=flags))
        out[tag + "_sparse_value"] = med(lambda: ops.transit_flux_sparse(t, rec, ld, flags=flags))
    f, gp, gl = ops.transit_flux_value_and_vjp(t, rec, ld, g, flags=flags)
    out[tag + "_check"] = [float(f.sum()), float(gp.abs().sum()), float(gl.abs().sum()), float(ops.transit_chi2(t, rec, ld, obs, w, flags=flags).sum())]
    del g, f
torch.cuda.empty_cache()
if QUICK:
    print(json.dumps(out))
    sys.exit(0)
# C4: 4 planets, 200 000 cadences, 64 draws
rng = np.random.default_rng(4)
n4, D4 = 200_000, 64
t4 = torch.arange(n4, dtype=torch.float64, device=dev) * CAD
base = dict(period=[3.5, 7.9, 13.1, 29.7], t0=[1.0, 2.3, 5.1, 11.7], b=[0.3, 0.1, 0.5, 0.2], ecc=[0.05, 0.1, 0.2, 0.3],
            omega=[1.1, -0.4, 2.0, 0.3], r=[0.1, 0.05, 0.07, 0.03])
L = {k: torch.tensor(np.asarray(v)[None, :] * (1 + 1e-3 * rng.normal(size=(D4, 4))), dtype=torch.float64, device=dev) for k, v in base.items()}
orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"])
rec4, ld4, _, fl4 = orbit.kernel_inputs(L["r"], (0.3, 0.2), use_in_transit=False)
rec4, ld4 = rec4.detach().contiguous(), ld4.detach().contiguous()
g4 = torch.randn(D4, n4, dtype=torch.float64, device=dev)
out["c4_64_dense_vjp"] = med(lambda: ops.transit_flux_value_and_vjp(t4, rec4, ld4, g4, flags=fl4))
f, gp, gl = ops.transit_flux_value_and_vjp(t4, rec4, ld4, g4, flags=fl4)
out["c4_64_check"] = [float(f.sum()), float(gp.abs().sum())]
# C5's light curve: 65 000 long cadences, exposure stencil x 7, secondary eclipse, 128 chains
rng = np.random.default_rng(5)
n5, D5, texp = 65_000, 128, 29.4 / 1440.0
t5 = torch.arange(n5, dtype=torch.float64, device=dev) * texp
mk = lambda v: torch.tensor(v * (1 + 1e-3 * rng.normal(size=(D5, 1))), dtype=torch.float64, device=dev)  # noqa: E731
L = {k: mk(v) for k, v in dict(period=2.7, t0=0.4, b=0.2, ecc=0.1, omega=0.7, r=0.08).items()}
orbit = xo.KeplerianOrbit(period=L["period"], t0=L["t0"], b=L["b"], ecc=L["ecc"], omega=L["omega"])
sbr = torch.full((D5,), 0.3, dtype=torch.float64, device=dev)
rec5, ld5, _, fl5 = orbit.kernel_inputs(L["r"], (0.3, 0.2), use_in_transit=False, secondary=((0.4, 0.1), sbr))
rec5, ld5 = rec5.detach().contiguous(), ld5.detach().contiguous()
from exoplanet_amd.light_curves.limb_dark import exposure_stencil  # noqa: E402
sdt, sw = exposure_stencil(7, 0)
sdt, sw = torch.tensor(sdt, device=dev), torch.tensor(sw, device=dev)
g5 = torch.randn(D5, n5, dtype=torch.float64, device=dev)
te = torch.tensor([texp], dtype=torch.float64, device=dev)
out["c5_128_lc_vjp"] = med(lambda: ops.transit_flux_value_and_vjp(t5, rec5, ld5, g5, texp=te, stencil_dt=sdt, stencil_w=sw, flags=fl5))
f, gp, gl = ops.transit_flux_value_and_vjp(t5, rec5, ld5, g5, texp=te, stencil_dt=sdt, stencil_w=sw, flags=fl5)
out["c5_128_check"] = [float(f.sum()), float(gp.abs().sum())]
print(json.dumps(out))
