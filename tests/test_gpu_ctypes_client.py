"""The reference-side binding of INTEGRATION.md, executed.

The reference's Ops hand host numpy arrays to a native driver (src/exoplanet/compat.py:27,56; call sites
orbits/keplerian.py:333,744-753, light_curves/limb_dark.py:24).  INTEGRATION.md sections 1-2 show the ctypes binding a
maintainer would add on top of include/exoplanet_amd.h: this test runs the code block of section 1 VERBATIM (only the
library path is made absolute) and the bodies of the three `perform` methods of section 2 in a subprocess that never
imports torch -- no torch allocations, raw hipMalloc / hipMemcpy, NULL stream -- on the committed golden vectors."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLIENT = r'''
import re, sys, os
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
sec1 = text[text.index("## 1. ctypes loader"):text.index("## 2. Drop-in")]
code = re.search(r"```python\n(.*?)```", sec1, re.S).group(1)
code = code.replace('"libexoplanet_amd.so"', repr(os.path.join(ROOT, "exoplanet_amd", "lib", "libexoplanet_amd.so")))
ns = {}
exec(code, ns)                                     # section 1, verbatim
lib, hip, dmalloc, h2d, d2h, vp = (ns[k] for k in ("lib", "hip", "dmalloc", "h2d", "d2h", "vp"))
sec2 = text[text.index("## 2. Drop-in"):text.index("## 3. The fast binding")]
helper = re.search(r"(def contact_points_host\(.*?\n    return Ml, Mr, flag\n)", sec2, re.S).group(1)
exec(helper, ns)                                   # section 2's contact-point helper, verbatim


def kepler_perform(M, ecc):                        # body of Kepler.perform (section 2)
    M, ecc = (np.ascontiguousarray(x, dtype=np.float64) for x in (M, ecc))
    n = M.size
    dM, de, ds, dc = (dmalloc(8 * n) for _ in range(4))
    h2d(dM, M); h2d(de, ecc)
    assert lib.exo_kepler_f64(dM, de, ds, dc, n, None) == 0     # NULL stream
    s, c = np.empty_like(M), np.empty_like(M)
    d2h(s, ds); d2h(c, dc)
    for p in (dM, de, ds, dc): hip.hipFree(p)
    return s, c


def quad_perform(b, r):                            # body of QuadSolutionVector.perform (section 2)
    b, r = (np.ascontiguousarray(x, dtype=np.float64) for x in (b, r))
    n = b.size
    db_, dr_ = dmalloc(8 * n), dmalloc(8 * n)
    ds, dsb, dsr = (dmalloc(24 * n) for _ in range(3))
    h2d(db_, b); h2d(dr_, r)
    assert lib.exo_quad_solution_vector_f64(db_, dr_, ds, dsb, dsr, n, None) == 0
    outs = []
    for p in (ds, dsb, dsr):
        a = np.empty(b.shape + (3,)); d2h(a, p); outs.append(a)
    for p in (db_, dr_, ds, dsb, dsr): hip.hipFree(p)
    return outs


g = np.load(os.path.join(ROOT, "tests", "golden", "kepler.npz"))
s, c = kepler_perform(g["M"], g["ecc"])
e = g["ecc"]
cond = 1 + (1 + e * g["cosf"]) ** 2 / (1 - e * e) ** 1.5 * np.abs(np.remainder(g["M"] + np.pi, 2 * np.pi) - np.pi)
err = np.maximum(np.abs(s - g["sinf"]), np.abs(c - g["cosf"]))
assert np.all(err <= 8 * 2.3e-16 * cond), float((err / cond).max())
q = np.load(os.path.join(ROOT, "tests", "golden", "quad_sv.npz"))
sv, dsdb, dsdr = quad_perform(q["b"], q["r"])
assert np.abs(sv - q["s"]).max() < 5e-15, float(np.abs(sv - q["s"]).max())
gap = np.minimum.reduce([np.abs(np.abs(q["b"]) - np.abs(1 - q["r"])), np.abs(np.abs(q["b"]) - (1 + q["r"])),
                         np.abs(np.abs(q["b"]) - q["r"]) + 1e-3])
tol = (5e-14 + 2e-15 / np.sqrt(np.maximum(gap, 1e-16)))[:, None]      # (the derivatives are singular at the contact points)
assert np.all(np.abs(dsdb - q["dsdb"]) <= tol) and np.all(np.abs(dsdr - q["dsdr"]) <= tol)
from oracle import numpy_port as P
rng = np.random.default_rng(9)
n = 300
a = rng.uniform(3, 60, n); e = rng.uniform(0, 0.9, n); w = rng.uniform(-np.pi, np.pi, n)
cosi = (1 + e * np.sin(w)) / (1 - e * e) * rng.uniform(0, 1.4, n) / a
sini = np.sqrt(np.clip(1 - cosi ** 2, 0, None))
L = 1 + rng.uniform(0.01, 0.2, n)
Ml, Mr, flag = ns["contact_points_host"](a, e, np.cos(w), np.sin(w), cosi, sini, L)
ml, mr, f0 = P.contact_points(a, e, np.cos(w), np.sin(w), cosi, sini, L)
assert np.array_equal(flag, f0)
good = f0 == 0
assert good.sum() > 100
assert np.abs(Ml[good] - ml[good]).max() < 1e-12 and np.abs(Mr[good] - mr[good]).max() < 1e-12
assert "torch" not in sys.modules, "the client must not need torch"
print("CTYPES_CLIENT_OK", int(good.sum()))
'''


@pytest.mark.gpu
def test_integration_md_ctypes_client_runs_without_torch(dev, tmp_path):
    script = tmp_path / "client.py"
    script.write_text(CLIENT)
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "CTYPES_CLIENT_OK" in res.stdout


def test_integration_md_perform_bodies_match_the_document():
    """the perform bodies the client runs ARE the ones printed in INTEGRATION.md section 2 (whitespace aside)"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    norm = lambda s: re.sub(r"\s+", " ", s)  # noqa: E731
    doc = norm(text)
    for line in ("assert lib.exo_kepler_f64(dM, de, ds, dc, n, None) == 0",
                 "assert lib.exo_quad_solution_vector_f64(db_, dr_, ds, dsb, dsr, n, None) == 0",
                 "assert lib.exo_contact_points_f64(*din, dl, dr_, dflag, n, None) == 0",
                 "dM, de, ds, dc = (dmalloc(8 * n) for _ in range(4))",
                 "ds, dsb, dsr = (dmalloc(24 * n) for _ in range(3))"):
        assert norm(line) in doc, line
        if "contact_points" not in line:       # (the contact-point helper is executed from the document itself)
            assert norm(line) in norm(CLIENT), line


SPARSE_CLIENT = r"""
# INTEGRATION.md section 4.1 as a torch-free client: the light curve as a SPARSE mean of a celerite GP, raw hipMalloc / hipMemcpy,
# NULL stream -- against the dense cadence-major route through the same raw ABI.
import re, sys, os, ctypes
import numpy as np
ROOT = sys.argv[1]
text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
sec1 = text[text.index("## 1. ctypes loader"):text.index("## 2. Drop-in")]
code = re.search(r"```python\n(.*?)```", sec1, re.S).group(1)
code = code.replace('"libexoplanet_amd.so"', repr(os.path.join(ROOT, "exoplanet_amd", "lib", "libexoplanet_amd.so")))
ns = {}
exec(code, ns)
lib, hip, dmalloc, h2d, d2h, vp = (ns[k] for k in ("lib", "hip", "dmalloc", "h2d", "d2h", "vp"))
sec4 = text[text.index("### 4.1 The light curve as a SPARSE mean"):text.index("### 4.2")]
struct_src = re.search(r"(class SparseModel\(ctypes.Structure\):.*?\n\n)", sec4, re.S).group(1)
exec("from ctypes import *\n" + struct_src, ns)                    # the struct of section 4.1, verbatim
SparseModel = ns["SparseModel"]
i64, i32, u32, cint = ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_int
lib.exo_transit_flux_workspace_bytes.restype = i64
lib.exo_transit_flux_workspace_bytes.argtypes = [i64, i64, i32]
lib.exo_celerite_state_doubles.restype = i64
lib.exo_celerite_state_doubles.argtypes = [i64, i64, i32, i32, i32]
lib.exo_celerite_default_chunks.restype = i32
lib.exo_celerite_default_chunks.argtypes = [i64, i64, i32, i32, i32]
lib.exo_pack_records_f64.argtypes = [vp, vp, i64, i32, u32, vp, vp, vp]
lib.exo_transit_flux_fwd_f64.argtypes = [vp, i64, vp, i64, vp, vp, i32, vp, vp, i64, i32, u32, vp, vp, i64, vp]
lib.exo_transit_flux_vjp_f64.argtypes = [vp, i64, vp, i64, vp, vp, i32, vp, vp, i64, i32, u32, vp, vp, vp, vp, vp, vp, i64, vp]
lib.exo_transit_flux_sparse_model.argtypes = [vp, i64, i64, i64, i32, u32, vp]
lib.exo_sparse_model_order.argtypes = [vp, i64, vp, vp]
lib.exo_transit_flux_vjp_sparse_f64.argtypes = [vp, i64, vp, i64, vp, vp, i32, vp, vp, i64, i32, u32, vp, vp, vp, vp, vp, i64, i32, vp]
gp_fwd = [vp, vp, vp, vp, i64, i64, vp, i32, vp, i32, vp, i64, vp, vp, i64, i32, vp]
gp_vjp = [vp, vp, vp, vp, i64, i64, vp, i32, vp, i32, vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]
lib.exo_celerite_loglike_sparse_fwd_f64.argtypes = gp_fwd
lib.exo_celerite_loglike_obs_fwd_cm_f64.argtypes = gp_fwd
lib.exo_celerite_loglike_sparse_vjp_f64.argtypes = gp_vjp
lib.exo_celerite_loglike_obs_vjp_cm_f64.argtypes = gp_vjp
SPARSE, CM, SORTED = 32, 128, 256

rng = np.random.default_rng(4)
D, n = 96, 6000
t = np.arange(n) * (10.0 / 1440.0)
orbit_in = np.zeros((D, 1, 10))
orbit_in[:, 0, 0] = 3.0 * (1 + 0.05 * rng.normal(size=D))     # period
orbit_in[:, 0, 1] = 1.0 + 0.05 * rng.normal(size=D)           # t0
orbit_in[:, 0, 2] = 0.3                                        # b
orbit_in[:, 0, 5] = 0.1 * (1 + 0.02 * rng.normal(size=D))     # r
orbit_in[:, 0, 6] = 1.0; orbit_in[:, 0, 7] = 1.0               # m_star, r_star
ld_in = np.tile([0.3, 0.2], (D, 1))
y = 5e-4 * rng.normal(size=n)
diag = np.full((1, n), 2.5e-7)
cc = np.stack([np.full(D, 1e-6), np.full(D, 5e-8), 0.4 * (1 + 0.1 * rng.normal(size=D)), np.full(D, 2.0)], -1).reshape(D, 1, 4)
gll = np.linspace(0.5, 1.5, D)

def dev(a):
    a = np.ascontiguousarray(a); p = dmalloc(max(a.nbytes, 8)); h2d(p, a); return p
def host(p, shape, dtype=np.float64):
    a = np.empty(shape, dtype=dtype); d2h(a, p); return a

dt, dy, ddiag, dcc, dgll = dev(t), dev(y), dev(diag), dev(cc), dev(gll)
params, ld = dmalloc(8 * D * 20), dmalloc(8 * D * 3)
assert lib.exo_pack_records_f64(dev(orbit_in), dev(ld_in), D, 1, 8, params, ld, None) == 0       # EXO_PACK_CIRCULAR
ws_bytes = lib.exo_transit_flux_workspace_bytes(n, D, 1)

def zeros(nbytes):
    p = dmalloc(nbytes); assert hip.hipMemset(p, 0, ctypes.c_size_t(nbytes)) == 0; return p

out = {}
for route in ("dense", "sparse"):
    ws = zeros(ws_bytes)
    sparse = route == "sparse"
    C = lib.exo_celerite_default_chunks(n, D, 0, 1, 1)          # the same plan on both routes: the same arithmetic
    n_state = lib.exo_celerite_state_doubles(n, D, 0, 1, C)
    state = dmalloc(8 * n_state)
    ll, gcc, gparams, gld = dmalloc(8 * D), dmalloc(8 * D * 4), dmalloc(8 * D * 20), dmalloc(8 * D * 3)
    if sparse:
        F = SPARSE | SORTED
        assert lib.exo_transit_flux_fwd_f64(dt, n, None, 0, None, None, 1, params, ld, D, 1, F, None, ws, ws_bytes, None) == 0
        m = SparseModel()
        assert lib.exo_transit_flux_sparse_model(ws, ws_bytes, n, D, 1, F, ctypes.byref(m)) == 0
        order = dmalloc(4 * D)
        assert lib.exo_sparse_model_order(ctypes.byref(m), D, order, None) == 0
        m.row_of_draw = order
        assert lib.exo_celerite_loglike_sparse_fwd_f64(dt, dy, ctypes.addressof(m), ddiag, 1, n, None, 0, dcc, 1, None, D, ll, state, n_state, C, None) == 0
        gvals = dmalloc(8 * D * n)
        assert lib.exo_celerite_loglike_sparse_vjp_f64(dt, dy, ctypes.addressof(m), ddiag, 1, n, None, 0, dcc, 1, None, D, dgll, state, n_state, C,
                                                       gvals, None, None, None, gcc, None) == 0
        assert lib.exo_transit_flux_vjp_sparse_f64(dt, n, None, 0, None, None, 1, params, ld, D, 1, F, gvals, gparams, gld, None, ws, ws_bytes, 1, None) == 0
        out["order"] = host(order, (D,), np.int32)
    else:
        F = CM | SORTED
        flux = dmalloc(8 * D * n)
        assert lib.exo_transit_flux_fwd_f64(dt, n, None, 0, None, None, 1, params, ld, D, 1, F, flux, ws, ws_bytes, None) == 0
        assert lib.exo_celerite_loglike_obs_fwd_cm_f64(dt, dy, flux, ddiag, 1, n, None, 0, dcc, 1, None, D, ll, state, n_state, C, None) == 0
        gflux = dmalloc(8 * D * n)
        assert lib.exo_celerite_loglike_obs_vjp_cm_f64(dt, dy, flux, ddiag, 1, n, None, 0, dcc, 1, None, D, dgll, state, n_state, C,
                                                       gflux, None, None, None, gcc, None) == 0
        assert lib.exo_transit_flux_vjp_f64(dt, n, None, 0, None, None, 1, params, ld, D, 1, F, gflux, None, gparams, gld, None, ws, ws_bytes, None) == 0
    assert hip.hipDeviceSynchronize() == 0
    out[route] = (host(ll, (D,)), host(gcc, (D, 4)), host(gparams, (D, 20)), host(gld, (D, 3)))

(ll_d, gcc_d, gp_d, gld_d), (ll_s, gcc_s, gp_s, gld_s) = out["dense"], out["sparse"]
assert np.all(np.isfinite(ll_d)) and np.abs(gp_d).max() > 0 and np.abs(gcc_d).max() > 0
assert np.abs(ll_s - ll_d).max() <= 1e-13 * np.abs(ll_d).max(), np.abs(ll_s - ll_d).max()
for a, b in ((gcc_s, gcc_d), (gp_s, gp_d), (gld_s, gld_d)):
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
assert sorted(out["order"].tolist()) == list(range(D)) and out["order"].tolist() != list(range(D))    # a real permutation
assert "torch" not in sys.modules, "the client must not need torch"
print("SPARSE_CLIENT_OK")
"""


@pytest.mark.gpu
def test_integration_md_sparse_mean_client_runs_without_torch(dev, tmp_path):
    """INTEGRATION.md section 4.1 (ABI 11 / 12): sweep -> exo_sparse_model -> exo_sparse_model_order -> sparse celerite pair -> sparse
    reverse sweep, a client that never imports torch, held to the dense cadence-major route through the same raw entry points"""
    script = tmp_path / "sparse_client.py"
    script.write_text(SPARSE_CLIENT)
    res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "SPARSE_CLIENT_OK" in res.stdout
