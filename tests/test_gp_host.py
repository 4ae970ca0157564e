"""CPU: the one-lane celerite pipeline of exo_celerite_core.hpp -- filtering elements, scan over
the chunks, CHECKPOINTED forward recurrences, adjoint scan, recomputing reverse recurrences --
compiled for the host (tests/gp_host_harness.cpp) and run lane by lane, against the oracle: the
dense-Cholesky likelihood and its analytic gradient (oracle/numpy_port.py) for short series, the C
port's sequential recurrences for long ones.  On the GPU the same functions run one lane per
(draw, chunk); tests/test_gpu_gp*.py check that build through the C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import c_port as C
from oracle import numpy_port as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "gp_host_harness.so")
    srcs = [os.path.join(ROOT, "tests", "gp_host_harness.cpp"),
            os.path.join(ROOT, "exoplanet_amd", "csrc", "exo_celerite_core.hpp"),
            os.path.join(ROOT, "exoplanet_amd", "csrc", "exo_math.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, srcs[0]], check=True)
    lib = ctypes.CDLL(so)
    lib.harness_gp_state_doubles.restype = ctypes.c_int64
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def run(lib, t, y, diag, real, cplx, kind=None, obs=None, n_chunks=0, gll=None, cadence_major=False):
    """-> loglike (D,), flags (D,), and with gll: dict of gradients.  cadence_major: the kernels are handed y (and write its
    cotangent) as [cadence][draw] arrays (Series::cm)"""
    t = np.ascontiguousarray(t, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    diag = np.ascontiguousarray(diag, dtype=np.float64)
    real = np.ascontiguousarray(real, dtype=np.float64)
    cplx = np.ascontiguousarray(cplx, dtype=np.float64)
    D, n = y.shape
    if cadence_major:
        y = np.ascontiguousarray(y.T)
    lib.harness_gp_set_cadence_major(1 if cadence_major else 0)
    n_real, n_complex = real.shape[1], cplx.shape[1]
    kind_p = None
    if kind is not None:
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        kind_p = kind.ctypes.data_as(_ip)
    if obs is not None:
        obs = np.ascontiguousarray(obs, dtype=np.float64)
    ns = lib.harness_gp_state_doubles(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, n_chunks)
    state = np.full(ns + 8, np.nan)
    ll, flags = np.empty(D), np.empty(D)
    args = (_p(t), _p(y), _p(obs), _p(diag), ctypes.c_int64(diag.shape[0]), ctypes.c_int64(n), _p(real), n_real, _p(cplx),
            n_complex, kind_p, ctypes.c_int64(D), n_chunks)
    C_used = lib.harness_gp_fwd(*args, _p(ll), _p(state), _p(flags))
    assert C_used > 1, C_used
    if gll is None:
        return ll, flags, C_used
    gll = np.ascontiguousarray(gll, dtype=np.float64)
    g = {"y": np.empty((D, n)), "diag": np.empty((D, n)), "diag_sum": np.empty(D), "real": np.empty_like(real),
         "cplx": np.empty_like(cplx)}
    rc = lib.harness_gp_vjp(*args, _p(gll), _p(state), _p(g["y"]), _p(g["diag"]), _p(g["diag_sum"]), _p(g["real"]),
                            _p(g["cplx"]))
    assert rc == C_used
    if cadence_major:
        g["y"] = np.ascontiguousarray(g["y"].reshape(n, D).T)
    return ll, flags, C_used, g


def rand_terms(rng, n_real, n_complex):
    """valid celerite terms in the filter form (a > 0, |b d| <= a c), moderate conditioning"""
    ar = rng.uniform(0.3, 1.5, n_real)
    cr = rng.uniform(0.05, 2.0, n_real)
    ac = rng.uniform(0.3, 1.5, n_complex)
    cc = rng.uniform(0.05, 1.0, n_complex)
    dc = rng.uniform(0.3, 4.0, n_complex)
    bc = rng.uniform(-0.9, 0.9, n_complex) * ac * cc / dc
    return ar, cr, ac, bc, cc, dc


@pytest.mark.parametrize("n_real,n_complex", [(1, 0), (2, 0), (0, 1), (1, 1), (0, 2), (3, 0), (2, 1), (1, 2), (0, 3), (2, 2)])
def test_lane_pipeline_vs_dense(harness, n_real, n_complex):
    rng = np.random.default_rng(10 * n_real + n_complex)
    n, D = 203, 3          # not a multiple of the checkpoint block or of the chunk length
    t = np.sort(rng.uniform(0, 40, n))
    t[50:60] += 3.0        # a gap
    t = np.sort(t)
    diag = rng.uniform(0.05, 0.3, (D, n))
    y = rng.normal(size=(D, n))
    terms = [rand_terms(rng, n_real, n_complex) for _ in range(D)]
    real = np.stack([np.stack([c[0], c[1]], -1) for c in terms]).reshape(D, n_real, 2)
    cplx = np.stack([np.stack([c[2], c[3], c[4], c[5]], -1) for c in terms]).reshape(D, n_complex, 4)
    gll = rng.normal(size=D)
    for n_chunks in (0, 5):
        ll, flags, C_used, g = run(harness, t, y, diag, real, cplx, n_chunks=n_chunks, gll=gll)
        assert np.all(flags == 0)
        assert C_used == (5 if n_chunks else C_used)
        for d in range(D):
            want, gw = P.gp_loglike_dense(t, y[d], diag[d], terms[d])
            assert abs(ll[d] - want) <= 1e-11 * abs(want)
            np.testing.assert_allclose(g["y"][d], gll[d] * gw["y"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(g["diag"][d], gll[d] * gw["diag"], rtol=1e-8, atol=1e-10)
            assert abs(g["diag_sum"][d] - gll[d] * gw["diag"].sum()) <= 1e-8 * np.abs(gw["diag"]).sum()
            sc = np.abs(gll[d])
            if n_real:
                np.testing.assert_allclose(g["real"][d, :, 0], gll[d] * gw["ar"], rtol=1e-7, atol=1e-8 * sc)
                np.testing.assert_allclose(g["real"][d, :, 1], gll[d] * gw["cr"], rtol=1e-7, atol=1e-8 * sc)
            if n_complex:
                for q, key in enumerate(("ac", "bc", "cc", "dc")):
                    np.testing.assert_allclose(g["cplx"][d, :, q], gll[d] * gw[key], rtol=1e-7, atol=1e-7 * sc)


def test_lane_pipeline_obs_series_and_shared_diag(harness):
    """obs - model formed on the fly: same likelihood, the gradient comes back with respect to the MODEL"""
    rng = np.random.default_rng(3)
    n, D = 160, 2
    t = np.sort(rng.uniform(0, 20, n))
    obs = rng.normal(size=n)
    model = 0.3 * rng.normal(size=(D, n))
    diag = rng.uniform(0.05, 0.2, (1, n))
    terms = [rand_terms(rng, 0, 1) for _ in range(D)]
    real = np.zeros((D, 0, 2))
    cplx = np.stack([np.stack(c[2:], -1) for c in terms])
    gll = np.array([1.0, -0.5])
    ll, flags, _, g = run(harness, t, model, diag, real, cplx, obs=obs, gll=gll)
    ll2, _, _, g2 = run(harness, t, obs[None] - model, np.repeat(diag, D, 0), real, cplx, gll=gll)
    np.testing.assert_allclose(ll, ll2, rtol=1e-14)
    np.testing.assert_allclose(g["y"], -g2["y"], rtol=1e-12, atol=1e-14)
    for d in range(D):
        want, _ = P.gp_loglike_dense(t, obs - model[d], diag[0], terms[d])
        assert abs(ll[d] - want) <= 1e-11 * abs(want)


def test_lane_pipeline_mixed_pair_kinds(harness):
    """a pair slot is one complex term or two real terms, draw by draw (SHO terms either side of
    Q = 1/2): kind 1 draws equal the same terms given as real terms, cotangents land in the slot"""
    rng = np.random.default_rng(5)
    n, D = 180, 4
    t = np.sort(rng.uniform(0, 30, n))
    y = rng.normal(size=(D, n))
    diag = rng.uniform(0.05, 0.2, (D, n))
    kind = np.array([[0], [1], [1], [0]], dtype=np.int32)
    cplx = np.zeros((D, 1, 4))
    terms = []
    for d in range(D):
        if kind[d, 0]:
            ar, cr, _, _, _, _ = rand_terms(rng, 2, 0)
            cplx[d, 0] = [ar[0], cr[0], ar[1], cr[1]]
            terms.append((ar, cr, np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0)))
        else:
            c = rand_terms(rng, 0, 1)
            cplx[d, 0] = [c[2][0], c[3][0], c[4][0], c[5][0]]
            terms.append(c)
    gll = rng.normal(size=D)
    ll, flags, _, g = run(harness, t, y, diag, np.zeros((D, 0, 2)), cplx, kind=kind, gll=gll)
    assert np.all(flags == 0)
    for d in range(D):
        want, gw = P.gp_loglike_dense(t, y[d], diag[d], terms[d])
        assert abs(ll[d] - want) <= 1e-11 * abs(want)
        np.testing.assert_allclose(g["y"][d], gll[d] * gw["y"], rtol=1e-8, atol=1e-10)
        if kind[d, 0]:
            got = g["cplx"][d, 0]
            np.testing.assert_allclose(got[[0, 2]], gll[d] * gw["ar"], rtol=1e-7, atol=1e-8)
            np.testing.assert_allclose(got[[1, 3]], gll[d] * gw["cr"], rtol=1e-7, atol=1e-8)
        else:
            for q, key in enumerate(("ac", "bc", "cc", "dc")):
                np.testing.assert_allclose(g["cplx"][d, 0, q], gll[d] * gw[key][0], rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("n_extra", [0, 1, 2])
def test_lane_pipeline_takes_overdamped_sho_pairs(harness, n_extra):
    """An SHO term with Q < 1/2 is two real terms, one of NEGATIVE amplitude: as a pair slot of kind 1 the time-parallel
    path takes it with the pair's joint state covariance (DeltaCoef::init) -- not flagged, log-likelihood and every
    gradient equal to the dense definition -- from deep over-damping to within 1 % of critical damping, alone (J = 2)
    and next to other terms (J = 4, 6: an under-damped SHO, a plain real pair)."""
    rng = np.random.default_rng(40 + n_extra)
    n = 230
    t = np.sort(rng.uniform(0, 40, n))
    t[100:] += 2.5
    Qs = [0.05, 0.2, 0.35, 0.45, 0.49, 0.495]
    D = len(Qs)
    y = rng.normal(size=(D, n))
    diag = rng.uniform(0.05, 0.3, (D, n))
    n_slot = 1 + n_extra
    cplx = np.zeros((D, n_slot, 4))
    kind = np.zeros((D, n_slot), dtype=np.int32)
    terms = []
    for d, Q in enumerate(Qs):
        ar, cr, *_ = P.sho_coefficients(rng.uniform(0.5, 1.5), rng.uniform(0.8, 2.0), Q)
        assert ar.size == 2 and ar.min() < 0          # the over-damped pair
        cplx[d, 0] = [ar[0], cr[0], ar[1], cr[1]]
        kind[d, 0] = 1
        tr = [ar, cr, np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0)]
        if n_extra >= 1:                               # an under-damped SHO in the next slot
            c = P.sho_coefficients(0.7, 1.1, 2.0)
            cplx[d, 1] = [c[2][0], c[3][0], c[4][0], c[5][0]]
            tr = [tr[0], tr[1]] + [c[k] for k in (2, 3, 4, 5)]
        if n_extra >= 2:                               # two positive real terms sharing a slot (independent, as before)
            a2, c2, *_ = rand_terms(rng, 2, 0)
            cplx[d, 2] = [a2[0], c2[0], a2[1], c2[1]]
            kind[d, 2] = 1
            tr[0], tr[1] = np.concatenate([tr[0], a2]), np.concatenate([tr[1], c2])
        terms.append(tuple(tr))
    gll = rng.normal(size=D)
    for n_chunks in (0, 6):
        ll, flags, _, g = run(harness, t, y, diag, np.zeros((D, 0, 2)), cplx, kind=kind, n_chunks=n_chunks, gll=gll)
        assert np.all(flags == 0), flags
        for d in range(D):
            want, gw = P.gp_loglike_dense(t, y[d], diag[d], terms[d])
            assert abs(ll[d] - want) <= 1e-10 * abs(want), (Qs[d], ll[d], want)
            tol = 1e-6 if Qs[d] > 0.48 else 1e-7      # (within 2 % of critical damping a1 ~ -a2 ~ 1 / f: conditioning)
            np.testing.assert_allclose(g["y"][d], gll[d] * gw["y"], rtol=tol, atol=1e-9)
            got = g["cplx"][d, 0]
            sc = np.abs(gll[d] * gw["ar"][:2]).max()
            np.testing.assert_allclose(got[[0, 2]], gll[d] * gw["ar"][:2], rtol=tol, atol=tol * sc)
            np.testing.assert_allclose(got[[1, 3]], gll[d] * gw["cr"][:2], rtol=tol, atol=tol * np.abs(gll[d] * gw["cr"][:2]).max())


def test_lane_pipeline_flags_what_it_cannot_take(harness):
    """the negative-amplitude real term of an over-damped SHO is outside the filter form: flagged
    (the library then redoes the draw with the sequential kernels)"""
    rng = np.random.default_rng(6)
    n = 128
    t = np.sort(rng.uniform(0, 10, n))
    ar, cr, *_ = P.sho_coefficients(1.0, 1.3, 0.3)
    real = np.stack([ar, cr], -1)[None]
    ll, flags, _ = run(harness, t, rng.normal(size=(1, n)), np.full((1, n), 0.1), real, np.zeros((1, 0, 4)))
    assert flags[0] == 2.0     # (kFlagSeq)


def test_lane_pipeline_long_series_vs_c_port(harness):
    """C3-like: evenly sampled 20 000 cadences, SHO term, default plan -- against the C port's
    sequential recurrences and their reverse pass"""
    rng = np.random.default_rng(8)
    n, D = 20_000, 2
    t = np.arange(n) * (2.0 / 1440.0)
    co = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3, 5.0, 1 / np.sqrt(2)), 1 / np.sqrt(2))
    y = 5e-4 * rng.normal(size=(D, n))
    diag = np.full((1, n), 2.5e-7)
    cplx = np.repeat(np.stack(co[2:], -1)[None], D, 0) * (1 + 1e-3 * rng.normal(size=(D, 1, 4)))
    cplx[:, :, 1] = np.minimum(np.abs(cplx[:, :, 1]), cplx[:, :, 0] * cplx[:, :, 2] / cplx[:, :, 3])
    gll = np.ones(D)
    ll, flags, C_used, g = run(harness, t, y, diag, np.zeros((D, 0, 2)), cplx, gll=gll)
    assert np.all(flags == 0) and C_used > 100
    for d in range(D):
        z = np.zeros(0)
        coeffs = (z, z, cplx[d, :, 0], cplx[d, :, 1], cplx[d, :, 2], cplx[d, :, 3])
        want, gw = C.celerite(t, y[d], diag[0], coeffs, grad=True)
        assert abs(ll[d] - want) <= 1e-12 * abs(want)
        np.testing.assert_allclose(g["y"][d], gw["y"], rtol=1e-7, atol=1e-9 * np.abs(gw["y"]).max())
        for q, key in enumerate(("ac", "bc", "cc", "dc")):
            np.testing.assert_allclose(g["cplx"][d, 0, q], gw[key][0], rtol=2e-6)


def test_lane_pipeline_reference_step_regimes(harness):
    """the reference-step shortcut of the one-lane kernels (DrawCoef::step / rot_uv): steps that differ in their last
    bits, barycentric-like drift of the step (1e-7 per cadence), gaps, a change of cadence half way (20 s -> 2 min),
    and a term too fast for the shortcut (d dt > 8) -- all against the C port's exact recurrences"""
    rng = np.random.default_rng(18)
    n, D = 6_000, 2
    dt = np.full(n, 2.0 / 1440.0) * (1 + 1e-7 * np.sin(np.arange(n) / 300.0))
    dt[n // 2:] *= 6.0                          # cadence change
    dt[[700, 701, 2500, 4100]] += (0.3, 0.02, 1.7, 0.5)   # gaps
    t = 1325.0 + np.cumsum(dt)
    base = P.sho_coefficients(*P.sho_from_sigma_rho(1e-3, 0.7, 3.0), 3.0)
    fast = P.sho_coefficients(*P.sho_from_sigma_rho(5e-4, 2e-4, 2.0), 2.0)      # d dt ~ 30: exact path
    y = 5e-4 * rng.normal(size=(D, n))
    diag = np.full((1, n), 2.5e-7)
    gll = np.ones(D)
    for co in (base, fast):
        cplx = np.repeat(np.stack(co[2:], -1)[None], D, 0) * (1 + 1e-3 * rng.normal(size=(D, 1, 4)))
        cplx[:, :, 1] = np.minimum(np.abs(cplx[:, :, 1]), cplx[:, :, 0] * cplx[:, :, 2] / cplx[:, :, 3])
        ll, flags, C_used, g = run(harness, t, y, diag, np.zeros((D, 0, 2)), cplx, gll=gll, n_chunks=24)
        assert np.all(flags == 0)
        for d in range(D):
            z = np.zeros(0)
            coeffs = (z, z, cplx[d, :, 0], cplx[d, :, 1], cplx[d, :, 2], cplx[d, :, 3])
            want, gw = C.celerite(t, y[d], diag[0], coeffs, grad=True)
            assert abs(ll[d] - want) <= 1e-12 * abs(want)
            np.testing.assert_allclose(g["y"][d], gw["y"], rtol=1e-7, atol=1e-9 * np.abs(gw["y"]).max())
            for q, key in enumerate(("ac", "bc", "cc", "dc")):
                np.testing.assert_allclose(g["cplx"][d, 0, q], gw[key][0], rtol=2e-6)


def test_tree_scans_equal_serial_scans(harness):
    """the scans over the chunks as trees of compositions (what the device launches) against the serial
    element-by-element scans, forward and adjoint, odd and even chunk counts, J = 1 .. 6"""
    rng = np.random.default_rng(21)
    for n_real, n_complex, n_chunks in ((1, 0, 5), (0, 1, 8), (2, 1, 7), (0, 3, 13), (1, 2, 2)):
        n, D = 900, 3
        t = np.sort(rng.uniform(0, 30, n))
        y = rng.normal(size=(D, n))
        diag = 0.1 + 0.1 * rng.uniform(size=(D, n))
        real = np.stack([10 ** rng.uniform(-1, 0, (D, n_real)), 10 ** rng.uniform(-1, 0.5, (D, n_real))], -1)
        a = 10 ** rng.uniform(-1, 0, (D, n_complex)); c = 10 ** rng.uniform(-1, 0.3, (D, n_complex))
        d = 10 ** rng.uniform(-0.5, 0.8, (D, n_complex)); b = rng.uniform(-0.9, 0.9, (D, n_complex)) * a * c / d
        cplx = np.stack([a, b, c, d], -1)
        gll = rng.normal(size=D)
        res = []
        for serial in (0, 1):
            harness.harness_set_serial_scan(serial)
            res.append(run(harness, t, y, diag, real, cplx, gll=gll, n_chunks=n_chunks))
        harness.harness_set_serial_scan(0)
        (ll0, _, _, g0), (ll1, _, _, g1) = res
        assert np.abs(ll0 - ll1).max() <= 1e-12 * np.abs(ll1).max()
        for k in g0:
            if g1[k].size:
                assert np.abs(g0[k] - g1[k]).max() <= 1e-10 * (np.abs(g1[k]).max() + 1e-300), (n_real, n_complex, k)


def test_two_levels_per_launch_is_the_same_tree(harness):
    """tree_scan4 (round 5: an item takes four elements, the intermediate level in registers; the way down skips storing the
    intermediate states) performs the compositions and applications of two radix-2 levels on the same numbers: log-likelihood and
    every gradient BIT-identical, chunk counts with every remainder of the pairing (odd / even numbers of levels, ragged last items)"""
    rng = np.random.default_rng(77)
    for n_real, n_complex, n_chunks in ((1, 0, 5), (0, 1, 8), (0, 1, 9), (2, 0, 16), (0, 1, 17), (0, 1, 31), (1, 0, 33), (0, 1, 64), (0, 1, 100), (2, 1, 13)):
        n, D = 40 * n_chunks, 3
        t = np.sort(rng.uniform(0, 30, n))
        y = rng.normal(size=(D, n))
        diag = 0.1 + 0.1 * rng.uniform(size=(D, n))
        real = np.stack([10 ** rng.uniform(-1, 0, (D, n_real)), 10 ** rng.uniform(-1, 0.5, (D, n_real))], -1)
        a = 10 ** rng.uniform(-1, 0, (D, n_complex)); c = 10 ** rng.uniform(-1, 0.3, (D, n_complex))
        d = 10 ** rng.uniform(-0.5, 0.8, (D, n_complex)); b = rng.uniform(-0.9, 0.9, (D, n_complex)) * a * c / d
        cplx = np.stack([a, b, c, d], -1)
        gll = rng.normal(size=D)
        res = []
        for four in (0, 1):
            harness.harness_set_tree4(four)
            res.append(run(harness, t, y, diag, real, cplx, gll=gll, n_chunks=n_chunks))
        harness.harness_set_tree4(0)
        (ll0, _, _, g0), (ll1, _, _, g1) = res
        assert np.array_equal(ll0, ll1), (n_real, n_complex, n_chunks)
        for k in g0:
            assert np.array_equal(g0[k], g1[k]), (n_real, n_complex, n_chunks, k)


def test_serial_top_of_the_scans_equals_the_trees(harness):
    """tree_scan_top (round 5, what the device does for J >= 3): the levels with at most 16 positions -- and everything above -- as a
    serial chain of applications from the seed, no compositions there; against the full trees, forward and adjoint, chunk counts that
    put the cut at different levels, J = 1 .. 6: the same mathematics in another association"""
    rng = np.random.default_rng(78)
    for n_real, n_complex, n_chunks, top in ((1, 0, 40, 16), (0, 1, 64, 16), (2, 1, 37, 4), (0, 3, 70, 16), (1, 2, 130, 16), (0, 3, 33, 2)):
        n, D = 30 * n_chunks, 3
        t = np.sort(rng.uniform(0, 30, n))
        y = rng.normal(size=(D, n))
        diag = 0.1 + 0.1 * rng.uniform(size=(D, n))
        real = np.stack([10 ** rng.uniform(-1, 0, (D, n_real)), 10 ** rng.uniform(-1, 0.5, (D, n_real))], -1)
        a = 10 ** rng.uniform(-1, 0, (D, n_complex)); c = 10 ** rng.uniform(-1, 0.3, (D, n_complex))
        d = 10 ** rng.uniform(-0.5, 0.8, (D, n_complex)); b = rng.uniform(-0.9, 0.9, (D, n_complex)) * a * c / d
        cplx = np.stack([a, b, c, d], -1)
        gll = rng.normal(size=D)
        res = []
        for st in (0, top):
            harness.harness_set_serial_top(st)
            res.append(run(harness, t, y, diag, real, cplx, gll=gll, n_chunks=n_chunks))
        harness.harness_set_serial_top(0)
        (ll0, _, _, g0), (ll1, _, _, g1) = res
        assert np.abs(ll0 - ll1).max() <= 1e-12 * np.abs(ll1).max()
        for k in g0:
            if g1[k].size:
                assert np.abs(g0[k] - g1[k]).max() <= 1e-10 * (np.abs(g1[k]).max() + 1e-300), (n_real, n_complex, n_chunks, k)


def _kernel(tau, c):
    return P.celerite_kernel(tau, *c)


@pytest.mark.parametrize("n_real,n_complex", [(1, 0), (0, 1), (1, 1), (0, 3)])
def test_dot_tril_and_predict_vs_dense(harness, n_real, n_complex):
    """z = L x (K + diag = L L^T) and the conditional mean kernel product K(tq, t) alpha, O(N)
    recurrences against dense linear algebra"""
    rng = np.random.default_rng(50 + n_real + 3 * n_complex)
    n, D, m = 150, 2, 97
    t = np.sort(rng.uniform(0, 30, n))
    diag = rng.uniform(0.05, 0.3, (D, n))
    x = rng.normal(size=(D, n))
    alpha = rng.normal(size=(D, n))
    tq = np.sort(np.concatenate([rng.uniform(-3, 33, m - 3), t[[0, 40, n - 1]]]))      # outside the data, and ON data points
    terms = [rand_terms(rng, n_real, n_complex) for _ in range(D)]
    real = np.ascontiguousarray(np.stack([np.stack([c[0], c[1]], -1) for c in terms]).reshape(D, n_real, 2))
    cplx = np.ascontiguousarray(np.stack([np.stack([c[2], c[3], c[4], c[5]], -1) for c in terms]).reshape(D, n_complex, 4))
    z = np.empty((D, n))
    mu = np.empty((D, m))
    i64 = ctypes.c_int64
    assert harness.harness_gp_dot_tril(_p(t), _p(diag), i64(D), i64(n), _p(real), n_real, _p(cplx), n_complex, None, i64(D),
                                       _p(np.ascontiguousarray(x)), _p(z)) == 0
    assert harness.harness_gp_predict(_p(t), i64(n), _p(np.ascontiguousarray(alpha)), _p(real), n_real, _p(cplx), n_complex,
                                      None, i64(D), _p(np.ascontiguousarray(tq)), i64(m), _p(mu)) == 0
    for d in range(D):
        K = _kernel(t[:, None] - t[None, :], terms[d]) + np.diag(diag[d])
        L = np.linalg.cholesky(K)
        np.testing.assert_allclose(z[d], L @ x[d], rtol=1e-9, atol=1e-11)
        Ks = _kernel(tq[:, None] - t[None, :], terms[d])
        np.testing.assert_allclose(mu[d], Ks @ alpha[d], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("n_real,n_complex", [(0, 1), (2, 1), (0, 3)])
def test_cadence_major_series_is_the_same_arithmetic(harness, n_real, n_complex):
    """the series as a [cadence][draw] array (what the light-curve sweep writes under EXO_FLAG_CADENCE_MAJOR): same bits
    as the [draw][cadence] rows, with and without an observed series subtracted on the fly"""
    rng = np.random.default_rng(77 + n_real)
    n, D = 331, 5
    t = np.sort(rng.uniform(0, 60, n))
    diag = rng.uniform(0.05, 0.3, (1, n))
    y = rng.normal(size=(D, n))
    obs = rng.normal(size=n)
    terms = [rand_terms(rng, n_real, n_complex) for _ in range(D)]
    real = np.stack([np.stack([c[0], c[1]], -1) for c in terms]).reshape(D, n_real, 2)
    cplx = np.stack([np.stack([c[2], c[3], c[4], c[5]], -1) for c in terms]).reshape(D, n_complex, 4)
    gll = rng.normal(size=D)
    for o in (None, obs):
        a = run(harness, t, y, diag, real, cplx, obs=o, gll=gll)
        b = run(harness, t, y, diag, real, cplx, obs=o, gll=gll, cadence_major=True)
        assert np.array_equal(a[0], b[0])
        for k in a[3]:
            assert np.array_equal(a[3][k], b[3][k]), k


def _sparse_case(rng, D, n, n_seg_max, touching=False):
    """random ascending disjoint segments per draw, their values, and the dense model they stand for"""
    seg_step, hi_at, cap = 4, 3, n_seg_max + 2
    nseg = np.zeros(D, dtype=np.int32)
    seg = np.full((D, cap, seg_step), -7, dtype=np.int32)
    off = np.zeros((D, cap + 1), dtype=np.int32)
    vals = np.full((D, n), np.nan)            # positions no segment covers stay NaN: nobody may read them
    dense = np.zeros((D, n))
    for d in range(D):
        k = int(rng.integers(0, n_seg_max + 1))
        cuts = np.sort(rng.choice(np.arange(0, n + 1), size=2 * k, replace=False)) if k else np.zeros(0, dtype=int)
        pos = int(rng.integers(0, 5))         # (the value array need not start at 0)
        for s in range(k):
            lo, hi = int(cuts[2 * s]), int(cuts[2 * s + 1])
            if touching and s > 0:
                lo = int(seg[d, s - 1, hi_at])   # segments that touch: the cursor steps across without a gap
                hi = max(hi, lo)                 # (possibly empty)
            seg[d, s, 0], seg[d, s, hi_at] = lo, hi
            off[d, s] = pos
            v = 0.3 * rng.normal(size=hi - lo)
            vals[d, pos:pos + hi - lo] = v
            dense[d, lo:hi] = v
            pos += hi - lo + int(rng.integers(0, 3))
        nseg[d] = k
    return nseg, seg, off, vals, dense, (cap * seg_step, cap + 1, n, seg_step, hi_at)


@pytest.mark.parametrize("n_real,n_complex,n,touching", [(0, 1, 203, False), (1, 0, 160, True), (0, 3, 331, False), (2, 1, 97, True)])
def test_lane_pipeline_sparse_model_equals_dense_model(harness, n_real, n_complex, n, touching):
    """round 5: the model as segments of cadences + values (gp::SparseSegs, the light-curve sweep's sparse output) through the
    element, forward and reverse lanes -- same log-likelihood, bit for bit, as the dense model it stands for, and the
    cotangent of every value equal to the dense cotangent at its cadence; positions outside the segments untouched"""
    rng = np.random.default_rng(100 * n_real + 10 * n_complex + n)
    D = 5
    t = np.sort(rng.uniform(0, 30, n))
    obs = rng.normal(size=n)
    diag = rng.uniform(0.05, 0.2, (1, n))
    terms = [rand_terms(rng, n_real, n_complex) for _ in range(D)]
    real = np.stack([np.stack([c[0], c[1]], -1) for c in terms]).reshape(D, n_real, 2)
    cplx = np.stack([np.stack([c[2], c[3], c[4], c[5]], -1) for c in terms]).reshape(D, n_complex, 4)
    gll = rng.normal(size=D)
    nseg, seg, off, vals, dense, (seg_row, off_row, val_row, seg_step, hi_at) = _sparse_case(rng, D, n, 6, touching)
    assert nseg.min() == 0 or True
    for n_chunks in (0, 7):
        harness.harness_gp_set_sparse(None, None, None, ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), 0, 0)
        ll_d, flags, C_used, g_d = run(harness, t, dense, diag, real, cplx, obs=obs, n_chunks=n_chunks, gll=gll)
        harness.harness_gp_set_sparse(nseg.ctypes.data_as(_ip), seg.ctypes.data_as(_ip), off.ctypes.data_as(_ip),
                                      ctypes.c_int64(seg_row), ctypes.c_int64(off_row), ctypes.c_int64(val_row), seg_step, hi_at)
        try:
            # (run() hands gresid a (D, n) buffer: here it is the cotangent of the VALUE array, val_row = n)
            ll_s, flags_s, C_s, g_s = run(harness, t, np.nan_to_num(vals, nan=1e300), diag, real, cplx, obs=obs,
                                          n_chunks=n_chunks, gll=gll)
        finally:
            harness.harness_gp_set_sparse(None, None, None, ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), 0, 0)
        assert C_s == C_used
        np.testing.assert_array_equal(ll_s, ll_d)
        for k in ("real", "cplx", "diag", "diag_sum"):
            np.testing.assert_array_equal(g_s[k], g_d[k])
        for d in range(D):
            for s in range(nseg[d]):
                lo, hi = seg[d, s, 0], seg[d, s, hi_at]
                np.testing.assert_array_equal(g_s["y"][d, off[d, s]:off[d, s] + hi - lo], g_d["y"][d, lo:hi])


def test_newton_convergence_measure_never_reads_a_nan_as_converged(harness):
    """the robust route's Newton iterations decide convergence from max |dP| / sqrt(P_jj P_ll); a NaN correction
    (dropped by fmax) or a NaN state (NaN > 0 is false) used to read as 0 -- "converged at iteration 0" -- and skip the serial
    chain.  The term now reports +inf for anything that is not finite (gp::newton_err_term, used by newton_down0)."""
    f = harness.harness_newton_err_term
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double, ctypes.c_double]
    assert abs(f(1e-9, 4.0) - 5e-10) < 1e-18
    assert f(0.0, 0.0) == 0.0 and f(1e-3, 0.0) == 0.0          # (a zero scale: the entry carries no information)
    for d, sc in ((np.nan, 1.0), (1.0, np.nan), (np.inf, 1.0), (-np.inf, 2.0), (1.0, np.inf), (np.nan, 0.0), (np.nan, np.nan)):
        assert f(d, sc) == np.inf, (d, sc)
    tol = 1e-8
    assert not (f(np.nan, 1.0) < tol) and (f(1e-12, 1.0) < tol)
