#!/usr/bin/env python
"""GP-heavy legs for an A/B of library builds: python tools/ab_gp.py  (EXOPLANET_AMD_LIB selects the build)"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for name, kw in (("c5", {}), ("c5_j8", dict(kernel="sho4")), ("c5_j10", dict(kernel="rot2_sho")), ("c5b", dict(bright=2)),
                 ("c5_kappa1e9", dict(bright=2, bright_factor=10 ** 4.5))):
    q = bench.extra_config(xo, ops, dev, "c5", 128, 5, **kw)
    out[name] = round(q["median_ms"], 3)
print(os.environ.get("EXOPLANET_AMD_LIB", "product").split("/")[-1], out)
