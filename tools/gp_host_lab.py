"""CPU laboratory for the conditioning tail of the time-parallel celerite path (no GPU): the host-compiled lane pipeline
(tests/gp_host_harness.cpp) built with the flags given on the command line, run on random kernels drawn as
tools/gp_cond_bins.py draws them, against the oracle's sequential recurrences (oracle/c).
usage: python tools/gp_host_lab.py <name> <seed> <cases> [-DFLAG ...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import c_port as C
import test_gp_host as H


def build(name, flags):
    out = os.path.join(ROOT, "tests", "_build", f"lab_{name}.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DEXO_GP_COND_MAX=1e30", "-DEXO_GP_COND_MAX_J2=1e30"] + flags
                   + ["-o", out, os.path.join(ROOT, "tests", "gp_host_harness.cpp")], check=True)
    lib = ctypes.CDLL(out)
    lib.harness_gp_state_doubles.restype = ctypes.c_int64
    return lib


def cases(seed, n_cases, jmax=6, nmax=1500):
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        n_real = int(rng.integers(0, 4))
        n_cplx = int(rng.integers(1, (6 - n_real) // 2 + 1))
        N = int(rng.integers(int(os.environ.get("LAB_NMIN", "300")), int(os.environ.get("LAB_NMAX", str(nmax)))))
        D = 8
        span = 10 ** rng.uniform(0, 3)
        t = np.sort(rng.uniform(0, span, N))
        if rng.uniform() < 0.3:
            t[N // 2:] += span * rng.uniform(0.5, 20)
        dtm = span / N
        cr = np.zeros((D, n_real, 2)); cc = np.zeros((D, n_cplx, 4))
        for d in range(D):
            for j in range(n_real):
                cr[d, j] = [10 ** rng.uniform(-2, 1), 10 ** rng.uniform(-3, 2) / dtm]
            for j in range(n_cplx):
                a = 10 ** rng.uniform(-2, 1); c = 10 ** rng.uniform(-3, 1.5) / dtm; dd = 10 ** rng.uniform(-2, 1.5) / dtm
                b = rng.uniform(-1, 1) * a * c / dd
                if rng.uniform() < 0.5:
                    b = np.sign(b) * min(abs(b), a * 10 ** rng.uniform(-1, 3))
                cc[d, j] = [a, b, c, dd]
        amp2 = cr[..., 0].sum(-1) + cc[..., 0].sum(-1)
        diag = (10 ** rng.uniform(-8, 0, size=(D, 1)) * amp2[:, None]) * (1 + 0.3 * rng.uniform(size=(D, N)))
        y = np.sqrt(amp2)[:, None] * rng.normal(size=(D, N))
        if n_real + 2 * n_cplx > jmax:
            continue
        yield t, y, diag, cr, cc, dtm


def main():
    name, seed, n_cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lib = build(name, sys.argv[4:])
    lib.harness_set_serial_scan(int(os.environ.get("LAB_SERIAL", "0")))
    lib.harness_set_polish(int(os.environ.get("LAB_POLISH", "0")))
    lib.harness_set_robust(int(os.environ.get("LAB_ROBUST", "0")))
    lib.harness_set_adj_tree(int(os.environ.get("LAB_ADJ_TREE", "0")))
    lib.harness_set_hybrid_k(int(os.environ.get("LAB_HYBRID_K", "-1")))
    rows = []
    for t, y, diag, cr, cc, dtm in cases(seed, n_cases):
        D = y.shape[0]
        gll = np.ones(D)
        n_chunks = int(os.environ.get("LAB_CHUNKS", "0"))
        if "LAB_L" in os.environ:
            n_chunks = max(2, t.size // int(os.environ["LAB_L"]))
        ll, flags, C_used, g = H.run(lib, t, y, diag, cr, cc, gll=gll, n_chunks=n_chunks)
        for d in range(D):
            co = (cr[d, :, 0], cr[d, :, 1], cc[d, :, 0], cc[d, :, 1], cc[d, :, 2], cc[d, :, 3])
            wl, wg = C.celerite(t, y[d], diag[d], co, grad=True)
            if not np.isfinite(wl):
                continue
            e = 0.0
            for got, want in ((g["y"][d], wg["y"]), (g["diag"][d], wg["diag"]), (g["real"][d, :, 0], wg["ar"]), (g["real"][d, :, 1], wg["cr"]),
                              (g["cplx"][d, :, 0], wg["ac"]), (g["cplx"][d, :, 1], wg["bc"]), (g["cplx"][d, :, 2], wg["cc"]),
                              (g["cplx"][d, :, 3], wg["dc"])):
                if want.size:
                    e = max(e, np.abs(got - want).max() / (np.abs(want).max() + 1e-300))
            ba2 = ((cc[d, :, 1] / cc[d, :, 0]) ** 2).max()
            snr = (cr[d, :, 0].sum() + cc[d, :, 0].sum()) / diag[d].min()
            rows.append((ba2, snr, e, cr.shape[1] + 2 * cc.shape[1], abs(ll[d] - wl) / abs(wl), C_used))
    rows = np.array(rows)
    kap = (1 + rows[:, 0]) * rows[:, 1]
    J = rows[:, 3].astype(int)
    print(name, "draws", len(rows))
    for nm, sel in (("J<=2", J <= 2), ("J=3,4", (J == 3) | (J == 4)), ("J=5,6", J >= 5)):
        line = "%7s" % nm
        for dlo in range(3, 9):
            m = sel & (kap >= 10.0 ** dlo) & (kap < 10.0 ** (dlo + 1))
            line += "  1e%d: %s" % (dlo, f"{rows[m, 2].max():.0e}/{np.median(rows[m, 2]):.0e}({m.sum()})" if m.any() else "-")
        print(line)
    np.save(os.path.join(ROOT, "gpurun_out", f"lab_{name}.npy"), rows)


if __name__ == "__main__":
    main()
