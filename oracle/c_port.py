"""ctypes wrapper of oracle/c (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "exo_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "c")], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def kepler(M, e):
    M = _c(M); e = _c(e) + np.zeros_like(M)
    e = _c(e)
    s = np.empty_like(M); c = np.empty_like(M)
    lib().oracle_kepler(_p(M), _p(e), _p(s), _p(c), ctypes.c_int64(M.size))
    return s, c


def quad_solution_vector(b, r):
    b = _c(b); r = _c(r) + np.zeros_like(b)
    r = _c(r)
    s = np.empty(b.shape + (3,)); db = np.empty_like(s); dr = np.empty_like(s)
    lib().oracle_quad_sv(_p(b), _p(r), _p(s), _p(db), _p(dr), ctypes.c_int64(b.size))
    return s, db, dr


def transit(t, params, ld, gflux=None, texp=None, stencil_dt=None, stencil_w=None, per_planet=False,
            window=False, secondary=False, want_flux=True):
    """-> flux (or None), gparams, gld (None without gflux)."""
    t = _c(t); params = _c(params); ld = _c(ld)
    D, P, _ = params.shape
    N = t.size
    flags = (1 if per_planet else 0) | (2 if window else 0) | (4 if secondary else 0)
    if texp is None:
        tex = sdt = sw = None; n_texp = 0; n_sub = 1
    else:
        tex = _c(np.atleast_1d(texp)); sdt = _c(stencil_dt); sw = _c(stencil_w)
        n_texp = tex.size; n_sub = sdt.size
    shape = (D, N, P) if per_planet else (D, N)
    flux = np.empty(shape) if want_flux else None
    if gflux is not None:
        gflux = _c(gflux)
        assert gflux.shape == shape
        gp = np.empty_like(params); gl = np.empty_like(ld)
    else:
        gp = gl = None
    lib().oracle_transit(_p(t), ctypes.c_int64(N), _p(tex), ctypes.c_int64(n_texp), _p(sdt), _p(sw),
                         ctypes.c_int32(n_sub), _p(params), _p(ld), ctypes.c_int64(D), ctypes.c_int32(P),
                         ctypes.c_uint32(flags), _p(gflux), _p(flux), _p(gp), _p(gl))
    return flux, gp, gl


def transit_ttv(t, params, ld, ttv, gflux=None, texp=None, stencil_dt=None, stencil_w=None, per_planet=False,
                window=False, secondary=False):
    """transit() with timing tables ttv = (edges [D,P,E], shift [D,P,E+1]) (ttv.py:158-187)
    -> flux, gparams, gld, gshift (the last three None without gflux)."""
    t = _c(t); params = _c(params); ld = _c(ld)
    edges, shift = _c(ttv[0]), _c(ttv[1])
    D, P, _ = params.shape
    N = t.size
    E = edges.shape[-1]
    assert edges.shape == (D, P, E) and shift.shape == (D, P, E + 1)
    flags = (1 if per_planet else 0) | (2 if window else 0) | (4 if secondary else 0)
    if texp is None:
        tex = sdt = sw = None; n_texp = 0; n_sub = 1
    else:
        tex = _c(np.atleast_1d(texp)); sdt = _c(stencil_dt); sw = _c(stencil_w)
        n_texp = tex.size; n_sub = sdt.size
    shape = (D, N, P) if per_planet else (D, N)
    flux = np.empty(shape)
    if gflux is not None:
        gflux = _c(gflux)
        assert gflux.shape == shape
        gp = np.empty_like(params); gl = np.empty_like(ld); gs = np.empty_like(shift)
    else:
        gp = gl = gs = None
    lib().oracle_transit_ttv(_p(t), ctypes.c_int64(N), _p(tex), ctypes.c_int64(n_texp), _p(sdt), _p(sw),
                             ctypes.c_int32(n_sub), _p(params), _p(ld), ctypes.c_int64(D), ctypes.c_int32(P),
                             ctypes.c_uint32(flags), _p(edges), _p(shift), ctypes.c_int32(E), _p(gflux), _p(flux),
                             _p(gp), _p(gl), _p(gs))
    return flux, gp, gl, gs


def celerite(t, y, diag, coeffs, grad=False):
    """log-likelihood of one draw (and, with grad, d/d(y, diag, ar, cr, ac, bc, cc, dc))."""
    ar, cr, ac, bc, cc, dc = [_c(x) for x in coeffs]
    t = _c(t); y = _c(y); diag = _c(diag)
    real = _c(np.stack([ar, cr], -1)) if ar.size else np.zeros((0, 2))
    cplx = _c(np.stack([ac, bc, cc, dc], -1)) if ac.size else np.zeros((0, 4))
    fn = lib().oracle_celerite
    fn.restype = ctypes.c_double
    if not grad:
        return fn(_p(t), _p(y), _p(diag), ctypes.c_int64(t.size), _p(real), ctypes.c_int32(ar.size), _p(cplx),
                  ctypes.c_int32(ac.size), None, None, None, None)
    gy = np.empty_like(y); gd = np.empty_like(y); gr = np.empty_like(real); gc = np.empty_like(cplx)
    ll = fn(_p(t), _p(y), _p(diag), ctypes.c_int64(t.size), _p(real), ctypes.c_int32(ar.size), _p(cplx),
            ctypes.c_int32(ac.size), _p(gy), _p(gd), _p(gr), _p(gc))
    g = {"y": gy, "diag": gd, "ar": gr[:, 0], "cr": gr[:, 1], "ac": gc[:, 0], "bc": gc[:, 1], "cc": gc[:, 2],
         "dc": gc[:, 3]}
    return ll, g
