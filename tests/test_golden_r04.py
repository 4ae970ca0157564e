"""CPU: the random-kernel conditioning tail (tests/golden/gp_tail.npz, oracle/make_golden_r04.py: long-double dense
log-likelihood + gradients of 17 random kernels, J = 2 .. 6, conditioning scores 1e3 .. 3e6) against
* the C port's sequential recurrences (what celerite2's own algorithm delivers there), and
* the HOST-COMPILED time-parallel lane pipeline -- the code the GPU runs -- at the series' own time stamps AND with the
  origin 2 457 000 days away (BJD-style stamps; the dense definition sees time differences only): every gradient to 1e-6;
  the draws under the library's thresholds unflagged, the three above them flagged (and still right on the lanes).  The gradient with respect to a complex term's oscillation rate used to be the whole tail
  and grew with the origin (exo_celerite_core.hpp, phase_flux): 1e-3 at 100 spans.

reference: celerite2 is a dependency of the reference (setup.py:36), not in its tree; BASELINE.md section 3 states the
tolerance (gradients 1e-6 relative)."""
import os

import numpy as np
import pytest

from oracle import c_port as C
from test_gp_host import harness, run  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = np.load(os.path.join(GOLD, "gp_tail.npz"))
KEYS = [str(k) for k in G["names"]]
NAMES = ("ar", "cr", "ac", "bc", "cc", "dc")


def case(key):
    co = tuple(G[f"{key}_{nm}"] for nm in NAMES)
    return G[f"{key}_t"], G[f"{key}_y"], G[f"{key}_diag"], co, float(G[f"{key}_loglike"])


def worst(got, key):
    e = 0.0
    for nm, v in got.items():
        w = G[f"{key}_g{nm}"]
        if w.size:
            e = max(e, float(np.abs(v - w).max() / np.abs(w).max()))
    return e


@pytest.mark.parametrize("key", KEYS)
def test_gp_tail_c_port_vs_long_double(key):
    t, y, diag, co, want = case(key)
    ll, gr = C.celerite(t, y, diag, co, grad=True)
    assert abs(ll - want) <= 1e-9 * abs(want)
    assert worst({"y": gr["y"], "diag": gr["diag"], **{nm: gr[nm] for nm in NAMES}}, key) <= 1e-6


@pytest.mark.parametrize("origin", [0.0, 2457000.0])
@pytest.mark.parametrize("key", KEYS)
def test_gp_tail_host_compiled_lane_pipeline(harness, key, origin):  # noqa: F811
    t, y, diag, co, want = case(key)
    ar, cr, ac, bc, cc, dc = co
    real = np.stack([ar, cr], -1)[None]
    cplx = np.stack([ac, bc, cc, dc], -1)[None]
    ll, flags, C_used, gr = run(harness, t + origin, y[None], diag[None], real, cplx, gll=np.ones(1))
    J = real.shape[1] + 2 * cplx.shape[1]
    kept = float(G[f"{key}_kappa"]) <= (1e7 if J <= 2 else 3e4)        # (exo_celerite_core.hpp: EXO_GP_COND_MAX_J2, EXO_GP_COND_MAX)
    # above the thresholds and up to a score of 1e8 (EXO_GP_COND_ROBUST_MAX): kFlagRobust -- the serial scans and the adjoint
    # inputs from the chunks' own recurrences, still on the lanes
    assert flags[0] == (0 if kept else 1) and C_used >= 8
    assert abs(ll[0] - want) <= 1e-9 * abs(want)
    got = {"y": gr["y"][0], "diag": gr["diag"][0], "ar": gr["real"][0, :, 0], "cr": gr["real"][0, :, 1]}
    got.update({nm: gr["cplx"][0, :, q] for q, nm in enumerate(("ac", "bc", "cc", "dc"))})
    assert np.array_equal((t + origin) - origin, t)       # (the fixture's stamps sit on a 2^-20 d grid: the shift is exact)
    assert worst(got, key) <= 1e-6, worst(got, key)


def test_robust_route_is_what_holds_the_ill_conditioned_draws(harness):  # noqa: F811
    """two J = 6 kernels at scores of 5e7 and 1e7: with the trees of element compositions (robust routing switched off) their
    gradients are off by 3e-5 and 2e-6; with the serial scans and the adjoint inputs from the chunks' own recurrences
    (chunk_adj_lane) they are within 1e-8 of the long-double definition"""
    res = {}
    for on in (0, 1):
        harness.harness_set_robust_flags(on)
        try:
            for key in ("c22", "c24"):
                t, y, diag, co, want = case(key)
                ar, cr, ac, bc, cc, dc = co
                real = np.stack([ar, cr], -1)[None]
                cplx = np.stack([ac, bc, cc, dc], -1)[None]
                ll, flags, C_used, gr = run(harness, t, y[None], diag[None], real, cplx, gll=np.ones(1))
                assert flags[0] == 1 and (not on or abs(ll[0] - want) <= 1e-9 * abs(want))
                got = {"y": gr["y"][0], "diag": gr["diag"][0], "ar": gr["real"][0, :, 0], "cr": gr["real"][0, :, 1]}
                got.update({nm: gr["cplx"][0, :, q] for q, nm in enumerate(("ac", "bc", "cc", "dc"))})
                res[on, key] = worst(got, key)
        finally:
            harness.harness_set_robust_flags(1)
    assert res[0, "c22"] > 1e-6 and res[0, "c24"] > 1e-6, res
    assert res[1, "c22"] <= 1e-8 and res[1, "c24"] <= 1e-8, res


def test_reverse_sweep_in_pieces_and_by_roles(harness):  # noqa: F811
    """chunk_adj_lane as the device runs it -- a chunk's reverse sweep dealt to eight pieces, each piece role by role (the state
    adjoints, the columns of X, R), the pieces' affine maps multiplied back together (adj_combine_lane) -- against the one sweep
    on one lane: the same numbers to rounding, and every gradient at the long-double definition's"""
    res = {}
    try:
        for pieces, roles in ((1, 0), (8, 1), (3, 1)):
            harness.harness_set_adj_pieces(pieces)
            harness.harness_set_adj_roles(roles)
            for key in ("c17", "c20", "c25"):
                t, y, diag, co, want = case(key)
                ar, cr, ac, bc, cc, dc = co
                real = np.stack([ar, cr], -1)[None]
                cplx = np.stack([ac, bc, cc, dc], -1)[None]
                ll, flags, C_used, gr = run(harness, t, y[None], diag[None], real, cplx, gll=np.ones(1))
                assert flags[0] == 1
                got = {"y": gr["y"][0], "diag": gr["diag"][0], "ar": gr["real"][0, :, 0], "cr": gr["real"][0, :, 1]}
                got.update({nm: gr["cplx"][0, :, q] for q, nm in enumerate(("ac", "bc", "cc", "dc"))})
                assert worst(got, key) <= 1e-8
                res[pieces, roles, key] = np.concatenate([got[k].ravel() for k in sorted(got)])
    finally:
        harness.harness_set_adj_pieces(8)
        harness.harness_set_adj_roles(0)
    for key in ("c17", "c20", "c25"):
        ref = res[1, 0, key]
        for other in ((8, 1), (3, 1)):
            assert np.abs(res[other + (key,)] - ref).max() <= 1e-9 * np.abs(ref).max()


def test_newton_iterations_reach_the_serial_chain(harness):  # noqa: F811
    """the device's forward scan for ill-conditioned draws (exo_celerite_group.hpp, tan_linearise): Newton iterations on the fixed
    point x_(c+1) = f_c(x_c) of the states entering the chunks, started at the trees' states -- here with the corrections'
    linear recurrence solved serially (tests/gp_host_harness.cpp, newton_scan).  Zero iterations = the trees' states: well off
    on these kernels; two iterations: the serial chain's gradients to 1e-8 and the long-double definition's to 1e-8"""
    res = {}
    try:
        for iters in (-1, 0, 2, 104):
            # (104: up to four iterations, stopped by the size of their corrections, the corrections' recurrence solved by
            # the device's TREE of plain products -- tan_compose / tan_apply -- instead of serially)
            harness.harness_set_newton_tree(1 if iters >= 100 else 0)
            harness.harness_set_newton(iters % 100 if iters >= 0 else iters, 0)
            for key in ("c22", "c24"):
                t, y, diag, co, want = case(key)
                ar, cr, ac, bc, cc, dc = co
                real = np.stack([ar, cr], -1)[None]
                cplx = np.stack([ac, bc, cc, dc], -1)[None]
                ll, flags, C_used, gr = run(harness, t, y[None], diag[None], real, cplx, gll=np.ones(1))
                got = {"y": gr["y"][0], "diag": gr["diag"][0], "ar": gr["real"][0, :, 0], "cr": gr["real"][0, :, 1]}
                got.update({nm: gr["cplx"][0, :, q] for q, nm in enumerate(("ac", "bc", "cc", "dc"))})
                res[iters, key] = (worst(got, key), np.concatenate([got[k].ravel() for k in sorted(got)]))
    finally:
        harness.harness_set_newton(-1, 0)
        harness.harness_set_newton_tree(0)
    for key in ("c22", "c24"):
        assert res[104, key][0] <= 1e-8
        assert np.abs(res[104, key][1] - res[-1, key][1]).max() <= 1e-8 * np.abs(res[-1, key][1]).max()
        assert res[0, key][0] > 30 * res[2, key][0]        # the trees' states (with the robust route's adjoint side): 5e-7 / 2e-8
        assert res[2, key][0] <= 1e-8
        chain, newton = res[-1, key][1], res[2, key][1]
        assert np.abs(newton - chain).max() <= 1e-8 * np.abs(chain).max()
