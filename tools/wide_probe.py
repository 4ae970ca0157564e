#!/usr/bin/env python
"""time-parallel path for wide states (J = 9 .. 16): golden check + timing at the C5 shape.  python tools/wide_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402
from exoplanet_amd.gp import celerite_loglike  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731
g = np.load(os.path.join(R, "tests", "golden", "gp_wide.npz"))
for key in ("rot2_sho", "rot3", "mixed16"):
    co = [g[f"{key}_{nm}"] for nm in ("ar", "cr", "ac", "bc", "cc", "dc")]
    want = float(g[f"{key}_loglike"])
    D = 5
    real = np.repeat(np.stack(co[:2], -1)[None], D, 0)
    cplx = np.repeat(np.stack(co[2:], -1)[None], D, 0)
    for chunks in (0, 1):
        os.environ["EXO_GP_CHUNKS"] = str(chunks) if chunks else ""
        if not chunks:
            del os.environ["EXO_GP_CHUNKS"]
        yt = T(np.repeat(g[f"{key}_y"][None], D, 0)).requires_grad_(True)
        ct = T(cplx).requires_grad_(True)
        rt = T(real).requires_grad_(True)
        ll = celerite_loglike(T(g[f"{key}_t"]), yt, T(np.repeat(g[f"{key}_diag"][None], D, 0)), rt, ct)
        ll.sum().backward()
        gy = yt.grad.cpu().numpy()[0]
        ref = g[f"{key}_gy"]
        gc = ct.grad.cpu().numpy()[0]
        refc = np.stack([g[f"{key}_g{nm}"] for nm in ("ac", "bc", "cc", "dc")], -1)
        from exoplanet_amd import _lib
        plan = int(_lib.load().exo_celerite_default_chunks(g[f"{key}_t"].size, D, real.shape[1], cplx.shape[1], 0))
        print(key, "plan", plan, "J =", real.shape[1] + 2 * cplx.shape[1], "N =", g[f"{key}_t"].size, "chunks" , chunks or "default",
              "ll rel err %.2e" % (abs(float(ll[0]) - want) / abs(want)),
              "gy rel %.2e" % (np.abs(gy - ref).max() / np.abs(ref).max()),
              "gcoef rel %.2e" % (np.abs(gc - refc) / np.abs(refc).max(0)).max())
os.environ.pop("EXO_GP_CHUNKS", None)
for kernel in ("sho3", "sho4", "rot2_sho"):
    q = bench.extra_config(xo, ops, dev, "c5", 128, 5, kernel=kernel)
    print(kernel, "C5 shape, 128 chains: %.3f ms" % q["median_ms"], q["launch"])
