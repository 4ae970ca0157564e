"""a whole user-level step of bench.py (extras: c5, c3 or ttv) eagerly, 5 times, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import exoplanet_amd as xo
bench.graphed = lambda xo_, fn, inputs, dev, iters: ([fn(*inputs) for _ in range(5)] and ({"median_ms": 1.0}, "eager"))
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "ttv":
    from exoplanet_amd import ops
    t = torch.arange(bench.N_CAD, dtype=torch.float64, device=dev) * bench.CADENCE
    gbar = torch.randn(1024, bench.N_CAD, dtype=torch.float64, device=dev)
    print(bench.extra_ttv(xo, ops, bench.make_leaves(1024, 100, dev), t, gbar, dev, 1024))
    sys.exit(0)
print(bench.extra_c5(xo, torch.device("cuda:0")) if len(sys.argv) < 2 or sys.argv[1] == "c5" else bench.extra_c3(xo, bench.make_leaves(1024, 100, torch.device("cuda:0")), torch.arange(bench.N_CAD, dtype=torch.float64, device="cuda:0") * bench.CADENCE, torch.device("cuda:0"), 1024))
