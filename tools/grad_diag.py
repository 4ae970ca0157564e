#!/usr/bin/env python
"""Which side carries the 1e-6-level per-draw differences of the C2 leaf gradients?  HIP step vs oracle/c (analytic VJP chained to
the leaves by finite-difference Jacobians, tests/test_gpu_timed_config.chain_to_leaves) vs a THIRD estimate: Richardson central
differences of the oracle's own L = sum(g * flux) in the leaf, for the worst draws.  python tools/grad_diag.py [draws]"""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import bench  # noqa: E402
import exoplanet_amd as xo  # noqa: E402
from exoplanet_amd import ops  # noqa: E402
from oracle import c_port as C  # noqa: E402
from oracle import numpy_port as P  # noqa: E402
from test_gpu_timed_config import chain_to_leaves, npy, records  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
C.lib().oracle_set_threads(16)
wl = bench.workload_c2(xo, ops, dev, D, rank=0)
lv = dict(zip(wl.names, wl.leaves))
vals = {k: npy(lv[k]) for k in wl.names if k not in ("u1", "u2")}
u1, u2 = npy(lv["u1"]), npy(lv["u2"])
rec, c = records(vals), np.stack(P.get_cl(u1, u2), -1)
t, g = npy(wl.data["t"]), npy(wl.data["gbar"])
want_f, want_gp, want_gl = C.transit(t, rec, c, g)
leaf = chain_to_leaves(vals, want_gp, list(P.GRAD_SLOTS[:-1]), gld=want_gl, u=(u1, u2))
out = wl.fn(*wl.leaves)
torch.cuda.synchronize()
got = dict(zip(wl.names, [npy(x) for x in out[2:]]))


def L_oracle(v, d):
    r1 = records({k: x[d:d + 1] for k, x in v.items()})
    f = C.transit(t, r1, c[d:d + 1], None, want_flux=True)[0]
    return float((g[d] * f[0]).sum())


for k in wl.names:
    if k in ("u1", "u2"):
        continue
    w, q = leaf[k].reshape(D), got[k].reshape(D)
    scale = np.abs(w).max()
    rel = np.abs(q - w) / np.maximum(np.abs(w), 1e-3 * scale)
    i = int(np.argmax(rel))
    # third estimate for draw i
    x0 = vals[k][i, 0]
    est = []
    for hs in (1e-5, 3e-5):
        h = hs * max(abs(x0), 1e-2)
        def dL(hh):
            vp = {kk: x.copy() for kk, x in vals.items()}; vm = {kk: x.copy() for kk, x in vals.items()}
            vp[k][i, 0] += hh; vm[k][i, 0] -= hh
            return (L_oracle(vp, i) - L_oracle(vm, i)) / (2 * hh)
        est.append((4 * dL(h) - dL(2 * h)) / 3)
    print(f"{k:7s} batch-max err/scale {np.abs(q - w).max() / scale:.2e}  worst per-draw rel {rel[i]:.2e} at draw {i} (|w|/scale {abs(w[i]) / scale:.2e}): "
          f"hip {q[i]:+.12e}  oracle-chain {w[i]:+.12e}  oracle-FD {est[0]:+.12e} / {est[1]:+.12e}")
