#!/usr/bin/env python
"""which torch ops (and from where) launch device kernels in one eager evaluation of a bench.py step:
tools/ops_of_step.py likelihood|c2|sparse|c3|c4|c5 [draws]"""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "likelihood"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if which == "likelihood":
    t = torch.arange(bench.N_CAD, dtype=torch.float64, device=dev) * bench.CADENCE
    obs = torch.as_tensor(1e-4 * np.random.default_rng(3).normal(size=bench.N_CAD), device=dev)
    leaves = bench.make_leaves(D, 100, dev)
    fn, vals = bench.likelihood_step_fn(xo, list(leaves), t, obs, 1e-4), list(leaves.values())
else:
    w = bench.WORKLOADS[which](xo, ops, dev, D)
    fn, vals = w.fn, w.leaves
for _ in range(3):
    fn(*vals)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fn(*vals)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        frames = [f for f in (e.stack or []) if "/exoplanet_amd/" in f or "bench.py" in f]
        rows.append((e.name, [k.name[:60] for k in e.kernels], frames[:2]))
for name, ks, fr in rows:
    print(f"{name:28s} {ks[0]:62s} {' <- '.join(x.split('/')[-1] for x in fr)}")
print(len(rows), "ops with kernels")
