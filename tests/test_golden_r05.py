"""CPU: tests/golden/gp_wide.npz (oracle/make_golden_r05.py) -- celerite kernels of state width J = 10, 12, 16, the
log-likelihood and every gradient from the dense definition in long double -- against the sequential recurrences of the
oracle's numpy and C ports: the fixture and the ports agree, so what the GPU's sequential kernels are held
to in tests/test_gpu_golden.py::test_gp_wide_golden is the published algorithm AND the dense definition.  The generating script
is committed; the fixture is data (inputs, expected outputs)."""
import os

import numpy as np
import pytest

from oracle import c_port as C
from oracle import numpy_port as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WIDE = ["rot2_sho", "rot3", "mixed16"]


@pytest.mark.parametrize("key", WIDE)
def test_gp_wide_ports_vs_long_double(key):
    g = np.load(os.path.join(GOLD, "gp_wide.npz"))
    co = tuple(g[f"{key}_{nm}"] for nm in ("ar", "cr", "ac", "bc", "cc", "dc"))
    J = co[0].size + 2 * co[2].size
    assert J == {"rot2_sho": 10, "rot3": 12, "mixed16": 16}[key]
    t, y, diag, want = g[f"{key}_t"], g[f"{key}_y"], g[f"{key}_diag"], float(g[f"{key}_loglike"])
    assert abs(P.celerite_loglike(t, y, diag, co) - want) <= 1e-10 * abs(want)
    ll, gr = C.celerite(t, y, diag, co, grad=True)
    assert abs(ll - want) <= 1e-10 * abs(want)
    np.testing.assert_allclose(gr["y"], g[f"{key}_gy"], rtol=1e-6, atol=1e-6 * np.abs(g[f"{key}_gy"]).max())
    for nm, c in zip(("ar", "cr", "ac", "bc", "cc", "dc"), co):
        if c.size:
            np.testing.assert_allclose(gr[nm], g[f"{key}_g{nm}"], rtol=1e-6, atol=1e-6 * np.abs(g[f"{key}_g{nm}"]).max())


EDGE = ["kappa1e9", "kappa1e10", "diag0", "diag0_j4"]


@pytest.mark.parametrize("key", EDGE)
def test_gp_edge_ports_vs_long_double(key):
    """tests/golden/gp_edge.npz (oracle/make_golden_r05b.py): conditioning scores of 1e9 and 1e10 and diag = 0 exactly -- the draws
    the library's time-parallel path hands to its sequential kernels -- on the oracle's sequential recurrences (numpy and C ports)
    against the dense definition in long double: log-likelihood to 1e-9, every gradient to 1e-6 of its largest entry"""
    g = np.load(os.path.join(GOLD, "gp_edge.npz"))
    co = tuple(g[f"{key}_{nm}"] for nm in ("ar", "cr", "ac", "bc", "cc", "dc"))
    t, y, diag, want = g[f"{key}_t"], g[f"{key}_y"], g[f"{key}_diag"], float(g[f"{key}_loglike"])
    if key.startswith("diag0"):
        assert np.all(diag == 0.0)
    else:
        assert (co[0].sum() + co[2].sum()) / diag.min() > 1e9
    assert abs(P.celerite_loglike(t, y, diag, co) - want) <= 1e-9 * abs(want)
    ll, gr = C.celerite(t, y, diag, co, grad=True)
    assert abs(ll - want) <= 1e-9 * abs(want)
    for nm in ("y", "diag", "ar", "cr", "ac", "bc", "cc", "dc"):
        w = g[f"{key}_g{nm}"]
        if w.size:
            assert np.abs(gr[nm] - w).max() <= 1e-6 * np.abs(w).max(), (nm, np.abs(gr[nm] - w).max(), np.abs(w).max())
