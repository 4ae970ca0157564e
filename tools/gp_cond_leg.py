#!/usr/bin/env python
"""the c3_gp_conditioning legs of bench.py alone (clean / 1 % near-critical / Matern): tools/gp_cond_leg.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

r = bench.extra_gp_conditioning(xo, ops, torch.device("cuda:0"), 1024)
print(json.dumps({k: round(v["median_ms"], 3) for k, v in r.items() if isinstance(v, dict) and "median_ms" in v}))
