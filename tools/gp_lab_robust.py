"""CPU: the robust route (serial forward scan + chunk_adj_lane, tests/gp_host_harness.cpp harness_set_robust) on random kernels:
the draws that disagree most with the sequential recurrences (C port), each against the long-double dense definition -- who is off.
usage: python tools/gp_lab_robust.py <seed> <cases> [threshold]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
import gp_host_lab as L
from oracle import c_port as C
from oracle.make_golden_r02 import gp_dense_ld
import test_gp_host as H

seed, n_cases = int(sys.argv[1]), int(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-7
lib = L.build("robust", [])
lib.harness_set_robust(1)
NAMES = ("y", "diag", "ar", "cr", "ac", "bc", "cc", "dc")


def lane_grads(g, d):
    return {"y": g["y"][d], "diag": g["diag"][d], "ar": g["real"][d, :, 0], "cr": g["real"][d, :, 1], "ac": g["cplx"][d, :, 0],
            "bc": g["cplx"][d, :, 1], "cc": g["cplx"][d, :, 2], "dc": g["cplx"][d, :, 3]}


def err(a, b):
    return {k: np.abs(a[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-300) for k in NAMES if b[k].size}


out = []
for ci, (t, y, diag, cr, cc, dtm) in enumerate(L.cases(seed, n_cases)):
    D = y.shape[0]
    ll, flags, Cu, g = H.run(lib, t, y, diag, cr, cc, gll=np.ones(D), n_chunks=0)
    for d in range(D):
        co = (cr[d, :, 0], cr[d, :, 1], cc[d, :, 0], cc[d, :, 1], cc[d, :, 2], cc[d, :, 3])
        wl, wg = C.celerite(t, y[d], diag[d], co, grad=True)
        if not np.isfinite(wl):
            continue
        e = err(lane_grads(g, d), wg)
        kap = (1 + ((cc[d, :, 1] / cc[d, :, 0]) ** 2).max()) * (cr[d, :, 0].sum() + cc[d, :, 0].sum()) / diag[d].min()
        if max(e.values()) > thr and kap < float(os.environ.get('LAB_KAPPA_MAX', '1e8')):
            out.append((max(e.values()), kap, ci, d, t, y[d], diag[d], co, lane_grads(g, d), wg, Cu))
out.sort(key=lambda r: -r[0])
print("draws over %.0e: %d" % (thr, len(out)))
for e, kap, ci, d, t, y, diag, co, lg, wg, Cu in out[:int(os.environ.get('LAB_SHOW', '12'))]:
    t0 = time.time()
    if t.size > 1600:
        print("case %d draw %d N %d kappa %.1e J %d: lanes vs C port %.1e (too long for the dense check)" % (ci, d, t.size, kap, len(co[0]) + 2 * len(co[2]), e))
        continue
    ll, gt = gp_dense_ld(t, y, diag, co)
    truth = {k: np.asarray(gt[k], dtype=float) for k in NAMES}
    el, es = err(lg, truth), err(wg, truth)
    print("case %d draw %d N %d C %d kappa %.1e J %d | lanes vs C port %.1e | vs long double: lanes %.1e (%s), C port %.1e (%s)  [%.0f s]" % (
        ci, d, t.size, Cu, kap, len(co[0]) + 2 * len(co[2]), e, max(el.values()), max(el, key=el.get), max(es.values()), max(es, key=es.get), time.time() - t0))
