#!/usr/bin/env python
"""the C2 step of bench.py (dense output, value + gradient) with use_in_transit on or off, replayed as a hipGraph a few
times: run under rocprofv3 --kernel-trace --stats to compare the kernels of the two.  tools/trace_step.py 0|1 [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
uit = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = bench.workload_c2(xo, ops, dev, 1024)
t, gbar = w.data["t"], w.data["gbar"]
names = w.names
fn = lambda *v: bench.step(xo, ops, dict(zip(names, v)), t, gbar, use_in_transit=uit)  # noqa: E731
q, how = bench.graphed(xo, fn, w.leaves, dev, iters)
print(uit, how, q)
