"""CPU: "trust but verify" -- can a draw whose scans went wrong be recognised AFTER the chunk passes, instead of being predicted from
its conditioning score?  Per draw: the largest mismatch between the state a chunk's recurrences LEAVE and the state the scan
handed the next chunk (forward), and between the adjoint a chunk's reverse recurrences END with and the adjoint the adjoint scan
handed the previous chunk; against the draw's worst gradient error (vs the C port).  python tools/gp_lab_mismatch.py <seed> <cases>"""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np

import gp_host_lab as L
from oracle import c_port as C

lib = L.build("mismatch", [])
lib.harness_gp_ckpt_layout.restype = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)


def run_all(t, y, diag, real, cplx):
    lib.harness_set_polish(0); lib.harness_set_serial_scan(0)
    D, n = y.shape
    n_real, n_complex = real.shape[1], cplx.shape[1]
    ns = lib.harness_gp_state_doubles(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0)
    state = np.full(ns + 8, np.nan); ll = np.empty(D); flags = np.empty(D)
    p = lambda a: a.ctypes.data_as(_dp)
    lib.harness_gp_set_cadence_major(0)
    args = (p(t), p(y), None, p(diag), ctypes.c_int64(diag.shape[0]), ctypes.c_int64(n), p(real), n_real, p(cplx), n_complex, None,
            ctypes.c_int64(D), 0)
    lib.harness_gp_fwd(*args, p(ll), p(state), p(flags))
    g = {"y": np.empty((D, n)), "diag": np.empty((D, n)), "diag_sum": np.empty(D), "real": np.empty_like(real), "cplx": np.empty_like(cplx)}
    lib.harness_gp_vjp(*args, p(np.ones(D)), p(state), p(g["y"]), p(g["diag"]), p(g["diag_sum"]), p(g["real"]), p(g["cplx"]))
    out = (ctypes.c_int64 * 6)(); lib.harness_gp_offsets(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, out)
    K = ctypes.c_int64(); span = ctypes.c_int64(); Lc = ctypes.c_int64()
    off = lib.harness_gp_ckpt_layout(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, ctypes.byref(K), ctypes.byref(span), ctypes.byref(Lc))
    return state, list(out), off, span.value, ll, g


rows = []
for t, y, diag, cr, cc, dtm in L.cases(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    t, y, diag, cr, cc = (np.ascontiguousarray(a) for a in (t, y, diag, cr, cc))
    D, n = y.shape
    J = cr.shape[1] + 2 * cc.shape[1]
    s, (ob, B, op, K, Cn, Lc), off, span, ll, g = run_all(t, y, diag, cr, cc)
    S3 = s.reshape(-1)
    idx = np.array([[(j * J - j * (j - 1) // 2 + (l - j)) if j <= l else (l * J - l * (l - 1) // 2 + (j - l)) for l in range(J)] for j in range(J)])
    for d in range(D):
        co = (cr[d, :, 0], cr[d, :, 1], cc[d, :, 0], cc[d, :, 1], cc[d, :, 2], cc[d, :, 3])
        wl, wg = C.celerite(t, y[d], diag[d], co, grad=True)
        if not np.isfinite(wl):
            continue
        e = 0.0
        for got, want in ((g["y"][d], wg["y"]), (g["diag"][d], wg["diag"]), (g["real"][d, :, 0], wg["ar"]), (g["real"][d, :, 1], wg["cr"]),
                          (g["cplx"][d, :, 0], wg["ac"]), (g["cplx"][d, :, 1], wg["bc"]), (g["cplx"][d, :, 2], wg["cc"]), (g["cplx"][d, :, 3], wg["dc"])):
            if want.size:
                e = max(e, np.abs(got - want).max() / (np.abs(want).max() + 1e-300))
        mf = ma = 0.0
        wnum = wden = 0.0
        for c in range(Cn - 1):
            ex = np.array([s[op + ((0 * Cn + c) * K + k) * D + d] for k in range(K)])              # what chunk c leaves
            gck = ((c + 1) * Lc) // span
            en = np.array([s[off + (gck * K + k) * D + d] for k in range(K)])                     # what the scan handed chunk c + 1
            mf = max(mf, np.abs(ex[:J] - en[:J]).max() / (np.abs(en[:J]).max() + 1e-300), np.abs(ex[J:] - en[J:]).max() / (np.abs(en[J:]).max() + 1e-300))
            ax = np.array([s[op + ((1 * Cn + c + 1) * K + k) * D + d] for k in range(K)])          # adjoint chunk c + 1's reverse sweep ends with
            Fb = np.array([s[ob + ((2 * Cn + c) * B + k) * D + d] for k in range(J)])              # adjoint the scan handed chunk c
            Sb = np.array([[s[ob + ((2 * Cn + c) * B + J + j * J + l) * D + d] for l in range(J)] for j in range(J)])
            Sbs = 0.5 * (Sb + Sb.T)
            axS = ax[J:][idx]
            ma = max(ma, np.abs(ax[:J] - Fb).max() / (np.abs(Fb).max() + 1e-300), np.abs(axS - Sbs).max() / (np.abs(Sbs).max() + 1e-300))
            # adjoint-weighted: what the forward mismatch at this boundary is worth in log-likelihood, term by term, against the
            # same sum of the state itself (Fbar, Sbar: the adjoint of the state entering chunk c + 1)
            enS = en[J:][idx]; exS = ex[J:][idx]
            wnum += (np.abs(Fb) * np.abs(ex[:J] - en[:J])).sum() + (np.abs(Sbs) * np.abs(exS - enS)).sum()
            wden += (np.abs(Fb) * np.abs(en[:J])).sum() + (np.abs(Sbs) * np.abs(enS)).sum()
            # and the adjoint mismatch weighted by the state
            wnum += (np.abs(ax[:J] - Fb) * np.abs(en[:J])).sum() + (np.abs(axS - Sbs) * np.abs(enS)).sum()
        ba2 = ((cc[d, :, 1] / cc[d, :, 0]) ** 2).max(); snr = (cr[d, :, 0].sum() + cc[d, :, 0].sum()) / diag[d].min()
        rows.append(((1 + ba2) * snr, e, mf, ma, J, abs(ll[d] - wl) / abs(wl), wnum / (wden + 1e-300)))
rows = np.array(rows)
np.save(os.path.join(R, "gpurun_out", "lab_mismatch.npy"), rows)
print("draws", len(rows))
m = np.maximum(rows[:, 2], rows[:, 3])
for lo, hi in ((0, 1e-12), (1e-12, 1e-11), (1e-11, 1e-10), (1e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-4), (1e-4, 1)):
    sel = (m >= lo) & (m < hi)
    if sel.any():
        print("mismatch [%g, %g): n = %4d  worst gradient error %.1e  median %.1e   (kappa up to %.0e)" % (lo, hi, sel.sum(), rows[sel, 1].max(), np.median(rows[sel, 1]), rows[sel, 0].max()))
