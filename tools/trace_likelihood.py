#!/usr/bin/env python
"""the white-noise-likelihood step of bench.py (C2 at 1024 draws: packing -> one-pass misfit + gradient sweep -> packing VJP),
replayed as a hipGraph a few times: run under rocprofv3 --kernel-trace --stats to see every kernel of the step"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402

dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t = torch.arange(bench.N_CAD, dtype=torch.float64, device=dev) * bench.CADENCE
obs = torch.as_tensor(1e-4 * np.random.default_rng(3).normal(size=bench.N_CAD), device=dev)
leaves = bench.make_leaves(D, 100, dev)
names = list(leaves)
fn = bench.likelihood_step_fn(xo, names, t, obs, 1e-4)
q, how = bench.graphed(xo, fn, list(leaves.values()), dev, iters)
print(how, q)
