#!/usr/bin/env python
"""the C5 shape with a wide kernel against the number of chunks: python tools/wide_chunks.py [kernel] [chains] [chunks ...]"""
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
dev = torch.device("cuda:0")
for c in (sys.argv[3:] or ["0", "32", "64", "128", "256", "512"]):
    if c == "0":
        os.environ.pop("EXO_GP_CHUNKS", None)
    else:
        os.environ["EXO_GP_CHUNKS"] = c
    q = bench.extra_config(xo, ops, dev, "c5", D, 5, kernel=kernel)
    print(kernel, "chains", D, "chunks", c, "%.3f ms" % q["median_ms"], flush=True)
