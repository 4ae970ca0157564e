"""CPU: the C-ABI library builds (hipcc cross-compiles without a GPU), loads, and
exports every symbol include/exoplanet_amd.h declares.  No kernel is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from exoplanet_amd import _lib

    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "exoplanet_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(exo_[a-z0-9_]+)\s*\(", text)))


def test_header_and_bindings_agree(lib):
    from exoplanet_amd import _lib

    assert declared_symbols() == _lib.exported_symbols()


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(os.path.join(ROOT, "exoplanet_amd", "lib", "libexoplanet_amd.so"))
    for name in declared_symbols():
        assert hasattr(raw, name), name


def test_abi_version_and_host_side_helpers(lib):
    from exoplanet_amd import _lib

    assert lib.exo_abi_version() == _lib.ABI_VERSION
    # pure host arithmetic: scratch sizes
    assert lib.exo_transit_flux_workspace_bytes(150000, 256, 1) > 0
    assert lib.exo_transit_flux_workspace_bytes(-1, 1, 1) == -1
    # saved factorisation (+ the chunk workspace of the time-parallel path, when it is taken)
    base = 100 * 3 * (2 + 4 + 4 + 6)
    assert lib.exo_celerite_state_doubles(100, 3, 0, 1, 0) >= base
    assert lib.exo_celerite_state_doubles(100, 3, 0, 1, 1) == base          # n_chunks = 1: sequential recurrences
    # the plan is a pure function of the arguments (no environment, no state): same answer twice, and
    # a forced chunk count changes it
    a = lib.exo_celerite_state_doubles(150000, 1024, 0, 1, 0)
    assert a == lib.exo_celerite_state_doubles(150000, 1024, 0, 1, 0) > 150000 * 1024 * (2 + 4 + 4 + 6)
    assert lib.exo_celerite_state_doubles(150000, 1024, 0, 1, 64) != a
    assert lib.exo_celerite_state_doubles(100, 3, 0, 0, 0) == -1
    assert lib.exo_celerite_state_doubles(100, 3, 0, 1, -2) == -1


def test_header_layout_constants_match_python(lib):
    from exoplanet_amd import ops

    text = open(os.path.join(ROOT, "include", "exoplanet_amd.h")).read()
    consts = dict(re.findall(r"#define\s+(EXO_[A-Z0-9_]+)\s+(\d+)u?\b", text))
    assert int(consts["EXO_NPAR"]) == ops.NPAR
    for name in ("N", "TP", "ECC", "COSW", "SINW", "COSI", "SINI", "AOR", "ROR", "T0", "PERIOD", "TS", "TE", "FRATIO",
                 "TS2", "TE2"):
        assert int(consts[f"EXO_P_{name}"]) == getattr(ops, f"P_{name}")
    assert int(consts["EXO_FLAG_PER_PLANET"]) == ops.FLAG_PER_PLANET
    assert int(consts["EXO_FLAG_WINDOW"]) == ops.FLAG_WINDOW
    assert int(consts["EXO_FLAG_SECONDARY"]) == ops.FLAG_SECONDARY
    assert int(consts["EXO_FLAG_EXACT_SCAN"]) == ops.FLAG_EXACT_SCAN
    assert int(consts["EXO_PACK_CIRCULAR"]) == ops.PACK_CIRCULAR
    assert int(consts["EXO_MAX_PLANETS"]) == ops.MAX_PLANETS
    assert int(consts["EXO_MAX_SUBEXP"]) == ops.MAX_SUBEXP
    from exoplanet_amd import _lib
    from exoplanet_amd.gp import celerite

    assert int(consts["EXO_SPARSE_ORDER_MAX_DRAWS"]) == celerite.lib_max_order_draws()
    assert int(consts["EXO_GP_MAX_J"]) == celerite.MAX_J
    assert int(re.search(r"#define\s+EXO_GP_PREPARE_ADJOINT\s+0x([0-9a-fA-F]+)", text).group(1), 16) == celerite.PREPARE_ADJOINT
    # exo_sparse_model_order rejects what it cannot sort, on the host, before any launch
    import ctypes

    assert lib.exo_sparse_model_order(None, 8, None, None) != 0
    m = _lib.SparseModel()
    assert lib.exo_sparse_model_order(ctypes.addressof(m), 8, None, None) != 0       # (no arrays behind the struct)


def test_rv_layout_constants_and_argument_checks(lib):
    """EXO_RV_* match the Python side; the newer entry points reject bad arguments on the host,
    before any launch (no GPU needed for that)"""
    from exoplanet_amd import ops

    text = open(os.path.join(ROOT, "include", "exoplanet_amd.h")).read()
    consts = dict(re.findall(r"#define\s+(EXO_[A-Z0-9_]+)\s+(\d+)u?\b", text))
    assert int(consts["EXO_RV_NPAR"]) == ops.RV_NPAR
    for name in ("N", "TP", "ECC", "COSW", "SINW", "AMP"):
        assert int(consts[f"EXO_RV_{name}"]) == getattr(ops, f"RV_{name}")
    INVALID = 1
    # radial velocity: negative sizes, no planets, missing pointers
    assert lib.exo_radial_velocity_fwd_f64(None, -1, None, 1, 1, None, None) == INVALID
    assert lib.exo_radial_velocity_fwd_f64(None, 10, None, 1, 0, None, None) == INVALID
    assert lib.exo_radial_velocity_fwd_f64(None, 10, None, 1, 1, None, None) == INVALID
    assert lib.exo_radial_velocity_fwd_f64(None, 0, None, 3, 2, None, None) == 0          # nothing to do
    assert lib.exo_radial_velocity_vjp_f64(None, 10, None, 1, 1, None, None, None) == INVALID
    # timing tables: both tables and a positive edge count are required
    args = [None, 10, None, 0, None, None, 1, None, None, 1, 1, 0]
    assert lib.exo_transit_flux_ttv_fwd_f64(*args, None, None, 4, None, None, 0, None) == INVALID
    assert lib.exo_transit_flux_ttv_fwd_f64(*args, 8, 8, 0, None, None, 0, None) == INVALID
    assert lib.exo_transit_flux_ttv_fwd_f64(*args, 8, 8, int(consts["EXO_MAX_TTV_EDGES"]) + 1, None, None, 0, None) == INVALID
    assert lib.exo_transit_flux_ttv_vjp_f64(*args, 8, 8, 4, None, None, None, None, None, None, None, 0, None) == INVALID
    # observed-minus-model likelihood: the observed series is required
    assert lib.exo_celerite_loglike_obs_fwd_f64(8, None, 8, 8, 1, 100, None, 0, 8, 1, None, 2, 8, None, 0, 0, None) == INVALID


def test_missing_extension_fails_loudly(monkeypatch):
    from exoplanet_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libexoplanet_amd.so")
    with pytest.raises(_lib.ExtensionMissingError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_ops_reject_host_tensors():
    import torch
    from exoplanet_amd import ops

    x = torch.zeros(4, dtype=torch.float64)
    for fn in (lambda: ops.kepler(x, x), lambda: ops.quad_solution_vector(x, x),
               lambda: ops.transit_flux(x, torch.zeros(1, 1, 16, dtype=torch.float64), torch.zeros(1, 3, dtype=torch.float64))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn()


def test_round_two_entry_points_reject_bad_arguments(lib):
    """the timing-table likelihood, the table construction and the sampler kernels check their arguments on the host,
    before any launch"""
    import ctypes

    from exoplanet_amd import ops

    INVALID = 1
    # white-noise likelihood with timing tables: flags it cannot take, missing tables, missing gshift, empty series
    base = [8, 10, None, 0, None, None, 1, 8, 8, 1, 1]
    for bad in (ops.FLAG_SECONDARY, ops.FLAG_LIGHT_DELAY, ops.FLAG_SPARSE, ops.FLAG_PER_PLANET, ops.FLAG_EXACT_SCAN):
        assert lib.exo_transit_chi2_ttv_vjp_f64(*base, bad, 8, 8, 3, 8, 8, 1, 8, 8, 8, 8, 8, 1 << 30, None) == INVALID
    assert lib.exo_transit_chi2_ttv_vjp_f64(*base, 0, None, 8, 3, 8, 8, 1, 8, 8, 8, 8, 8, 1 << 30, None) == INVALID
    assert lib.exo_transit_chi2_ttv_vjp_f64(*base, 0, 8, 8, 3, 8, 8, 1, 8, 8, 8, None, 8, 1 << 30, None) == INVALID
    empty = [8, 0, None, 0, None, None, 1, 8, 8, 1, 1]
    assert lib.exo_transit_chi2_ttv_vjp_f64(*empty, 0, 8, 8, 3, 8, 8, 1, 8, 8, 8, 8, 8, 1 << 30, None) == INVALID
    assert lib.exo_transit_chi2_ttv_vjp_f64(*base, 0, 8, 8, 3, 8, 8, 5, 8, 8, 8, 8, 8, 1 << 30, None) == INVALID   # n_ivar
    # table construction: n_edge must be the widest row + 1, every planet needs offsets and at least one transit
    ptr = (ctypes.c_void_p * 2)(8, 8)
    ds = (ctypes.c_int64 * 2)(0, 0)
    cnt = (ctypes.c_int32 * 2)(5, 3)
    assert lib.exo_ttv_tables_f64(8, 0, 1, 8, 0, 1, ptr, ds, cnt, 4, 2, 5, 8, 8, None) == INVALID          # n_edge != 6
    assert lib.exo_ttv_tables_f64(8, 0, 1, 8, 0, 1, ptr, ds, cnt, 0, 2, 6, None, None, None) == 0          # no draws
    assert lib.exo_ttv_tables_f64(8, 0, 1, 8, 0, 1, ptr, ds, cnt, 4, 2, 6, None, 8, None) == INVALID        # no output
    cnt0 = (ctypes.c_int32 * 2)(5, 0)
    assert lib.exo_ttv_tables_f64(8, 0, 1, 8, 0, 1, ptr, ds, cnt0, 4, 2, 6, 8, 8, None) == INVALID
    hole = (ctypes.c_void_p * 2)(8, None)
    assert lib.exo_ttv_tables_vjp_f64(8, 0, 1, 8, 0, 1, hole, ds, cnt, 4, 2, 6, 8, ptr, 8, None) == INVALID
    assert lib.exo_ttv_tables_vjp_f64(8, 0, 1, 8, 0, 1, ptr, ds, cnt, 4, 2, 6, None, ptr, 8, None) == INVALID
    # sampler kernels: forty pointers, a known phase
    full = (ctypes.c_void_p * 40)(*([8] * 40))
    assert lib.exo_nuts_f64(None, 4, 3, 2, 1000.0, 0, None) == INVALID
    assert lib.exo_nuts_f64(full, 4, 3, 2, 1000.0, 4, None) == INVALID
    assert lib.exo_nuts_f64(full, 4, 0, 2, 1000.0, 0, None) == INVALID
    assert lib.exo_nuts_f64(full, 0, 3, 2, 1000.0, 0, None) == 0                                           # no chains
    gap = (ctypes.c_void_p * 40)(*([8] * 39 + [None]))
    assert lib.exo_nuts_f64(gap, 4, 3, 2, 1000.0, 1, None) == INVALID


def test_no_memset_or_copy_nodes_in_the_library():
    """Every entry point may be captured into a hipGraph (exoplanet_amd/graph.py).  hipMemsetAsync would become a memset node
    there, and on ROCm 7.2 a memset node fills with garbage once the process has issued enough eager device-to-device copies
    between replays (round 5: exo_math.hpp, zero_fill_async; tools/graph_node_order.py reproduces it).  Zeros come from a kernel;
    nothing in csrc/ enqueues anything but kernels (copy nodes measured fine, kept out all the same)."""
    import glob

    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "exoplanet_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.h*")):
        code = "\n".join(line.split("//")[0] for line in open(path).read().splitlines())
        for call in ("hipMemsetAsync", "hipMemcpyAsync", "hipMemset(", "hipMemcpy("):
            assert call not in code, f"{os.path.basename(path)} calls {call}"
