#!/usr/bin/env python
"""a few eager C5-shape steps with a chosen kernel (for rocprofv3): python tools/wide_step.py rot2_sho|sho4|sho3 [chains] [steps]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

kernel = sys.argv[1] if len(sys.argv) > 1 else "rot2_sho"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
w = bench.workload_c5(xo, ops, dev, D, kernel=kernel)
for _ in range(steps):
    w.fn(*w.leaves)
torch.cuda.synchronize()
