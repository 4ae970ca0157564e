"""how the light-curve sweep's time depends on the number of solved cadences: the C2 batch with caller windows
(EXO_FLAG_WINDOW) scaled about their centres, value + VJP, dense and sparse output
    python tools/window_scaling.py [own]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import exoplanet_amd as xo
from exoplanet_amd import ops

dev = torch.device("cuda:0")
D = 1024
leaves = bench.make_leaves(D, 100, dev)
t = torch.arange(bench.N_CAD, dtype=torch.float64, device=dev) * bench.CADENCE
g = torch.randn(D, bench.N_CAD, dtype=torch.float64, device=dev)
with torch.no_grad():
    orb = xo.orbits.KeplerianOrbit(period=leaves["period"], t0=leaves["t0"], b=leaves["b"], ecc=leaves["ecc"], omega=leaves["omega"])
    rec, ld, _, fl = orb.kernel_inputs(leaves["r"], (leaves["u1"], leaves["u2"]), use_in_transit=True)
own = len(sys.argv) > 1 and sys.argv[1] == "own"   # the library's own windows instead (one pass)
if own:
    fl &= ~ops.FLAG_WINDOW
for s in ((1.0,) if own else (0.5, 0.8, 0.9, 1.0, 1.1, 1.25, 1.5, 2.0)):
    r = rec.clone()
    mid, half = 0.5 * (rec[..., ops.P_TS] + rec[..., ops.P_TE]), 0.5 * (rec[..., ops.P_TE] - rec[..., ops.P_TS])
    r[..., ops.P_TS], r[..., ops.P_TE] = mid - s * half, mid + s * half
    for name, flags in (("dense", fl), ("sparse", fl | ops.FLAG_SPARSE)):
        def run():
            if flags & ops.FLAG_SPARSE:
                return ops.transit_flux_sparse(t, r, ld, gflux=g, flags=flags & ~ops.FLAG_SPARSE)
            return ops.transit_flux_value_and_vjp(t, r, ld, g, flags=flags)
        out = run()
        n = out[0].n_solved() if flags & ops.FLAG_SPARSE else int((out[0] != 0).sum())
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40):
            run()
        b.record(); torch.cuda.synchronize()
        print("scale %.2f %-6s %.4f ms  cadences %d" % (s, name, a.elapsed_time(b) / 40, n))
