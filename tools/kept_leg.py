"""bench.py's extras leg c2_dense_kept_across_steps on its own: python tools/kept_leg.py"""
import json, sys, torch
sys.path.insert(0, '.')
import bench, exoplanet_amd as xo
from exoplanet_amd import ops
import numpy as np
dev = torch.device('cuda:0'); D = 1024
wl = bench.workload_c2(xo, ops, dev, D)
leaves, t, gbar = wl.data["leaves"], wl.data["t"], wl.data["gbar"]
with torch.no_grad():
    orbit = xo.KeplerianOrbit(**{k: leaves[k].detach() for k in ("period", "t0", "b", "ecc", "omega")})
    params, ld, _, _ = orbit.kernel_inputs(leaves["r"].detach(), (leaves["u1"].detach(), leaves["u2"].detach()))
    params, ld = params.contiguous(), ld.contiguous()
keeper = ops.KeptDenseFlux(t, D, 1)
for name, fn in (("dense_sweep", lambda pr: ops.transit_flux_value_and_vjp(t, pr, ld, gbar)[1:]),
                 ("sparse_sweep", lambda pr: ops.transit_flux_sparse(t, pr, ld, gbar)[1:3]),
                 ("kept_dense", lambda pr: keeper.step(pr, ld, gbar)[1:3])):
    q, how = bench.graphed(xo, fn, [params], dev, 50)
    print(name, round(q["median_ms"], 4), how)
print("equal", bool(torch.equal(keeper.flux, ops.transit_flux_value_and_vjp(t, params, ld, gbar)[0])))
