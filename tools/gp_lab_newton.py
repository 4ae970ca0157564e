"""CPU: Newton iterations on the chunk-boundary fixed point (tests/gp_host_harness.cpp, newton_scan), started at the trees' states,
against the serial forward chain of the robust route: per number of iterations, the worst disagreement of any gradient over the
draws the element lanes flag kFlagRobust.  usage: python tools/gp_lab_newton.py <seed> <cases> [L]"""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
import gp_host_lab as L
import test_gp_host as H

seed, n_cases = int(sys.argv[1]), int(sys.argv[2])
import subprocess
out = os.path.join(R, "tests", "_build", "lab_newton.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, os.path.join(R, "tests", "gp_host_harness.cpp")], check=True)
lib = ctypes.CDLL(out); lib.harness_gp_state_doubles.restype = ctypes.c_int64
lib.harness_set_newton_tree(int(os.environ.get('LAB_NEWTON_TREE', '0')))   # 1: the corrections' recurrence by the tree of plain products
KS = (0, 1, 2, 3, 4, 6)
worst = {k: [] for k in KS}
n_chunks_of = lambda t: max(2, t.size // int(sys.argv[3])) if len(sys.argv) > 3 else 0
for t, y, diag, cr, cc, dtm in L.cases(seed, n_cases):
    D = y.shape[0]
    lib.harness_set_newton(-1, 0)
    ll0, flags, Cu, g0 = H.run(lib, t, y, diag, cr, cc, gll=np.ones(D), n_chunks=n_chunks_of(t))
    rob = np.nonzero(flags == 1)[0]
    if not rob.size:
        continue
    for k in KS:
        lib.harness_set_newton(k, 0)
        ll, _, _, g = H.run(lib, t, y, diag, cr, cc, gll=np.ones(D), n_chunks=n_chunks_of(t))
        for d in rob:
            e = abs(ll[d] - ll0[d]) / abs(ll0[d])
            for nm in ("y", "diag", "real", "cplx"):
                a, b = g[nm][d], g0[nm][d]
                if b.size:
                    e = max(e, np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
            kap = (1 + ((cc[d, :, 1] / cc[d, :, 0]) ** 2).max()) * (cr[d, :, 0].sum() + cc[d, :, 0].sum()) / diag[d].min()
            worst[k].append((e if np.isfinite(e) else 1e300, kap, Cu))
lib.harness_set_newton(-1, 0)
for k in KS:
    a = np.array(worst[k])
    line = "newton %d: draws %d" % (k, len(a))
    for lo in (4, 5, 6, 7):
        m = (a[:, 1] >= 10.0 ** lo) & (a[:, 1] < 10.0 ** (lo + 1))
        if m.any():
            line += "  1e%d: worst %.0e median %.0e (%d)" % (lo, a[m, 0].max(), np.median(a[m, 0]), m.sum())
    print(line)
