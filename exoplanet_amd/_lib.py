"""ctypes binding of the C ABI in include/exoplanet_amd.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
--offload-arch=gfx950).  There is NO fallback: if the library is missing or an
op is handed a host tensor, the call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXOPLANET_AMD_LIB selects another in-tree build of the same ABI (A/B measurements)
LIB_PATH = os.environ.get("EXOPLANET_AMD_LIB") or os.path.join(_HERE, "lib", "libexoplanet_amd.so")
ABI_VERSION = 14

_c_dp = ctypes.c_void_p  # device pointers travel as integers
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_u32 = ctypes.c_uint32

_SIGNATURES = {
    "exo_abi_version": (_i32, []),
    "exo_kepler_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    "exo_selftest_orbit_pos_f32": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    "exo_quad_solution_vector_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    "exo_contact_points_f64": (ctypes.c_int, [_c_dp] * 10 + [_i64, _c_dp]),
    "exo_transit_flux_fwd_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _i64, _c_dp],
    ),
    "exo_transit_flux_fwd_ev_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _i64, _c_dp,
         _c_dp, _c_dp],
    ),
    "exo_transit_flux_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "exo_transit_flux_sparse_layout": (ctypes.c_int, [_i64, _i64, _i32, ctypes.POINTER(_i64)]),
    "exo_transit_flux_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp],
    ),
    "exo_transit_flux_vjp_ev_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp],
    ),
    "exo_transit_flux_ttv_fwd_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _i32,
         _c_dp, _c_dp, _i64, _c_dp],
    ),
    "exo_transit_flux_ttv_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _i32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp],
    ),
    # cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride (host), pack_flags, t, n_cad, texp, n_texp, stencil_dt,
    # stencil_w, n_sub, n_draw, n_planet, flags, gflux, flux_out, params, ld, gparams, gld, flux_dot, fold, gscale, gcols, gld_cols
    # (host), workspace, workspace_bytes, stream, ev_start, ev_stop
    "exo_transit_flux_cols_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _u32, _c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _i64, _i32, _u32, _c_dp, _c_dp,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp],
    ),
    "exo_celerite_state_doubles": (_i64, [_i64, _i64, _i32, _i32, _i32]),
    "exo_celerite_default_chunks": (_i32, [_i64, _i64, _i32, _i32, _i32]),
    "exo_sparse_model_order": (ctypes.c_int, [_c_dp, _i64, _c_dp, _c_dp]),
    # t, resid, diag, n_diag, n, coef_real, n_real, coef_complex, n_complex, pair_kind, n_draw, loglike, state,
    # state_doubles, n_chunks, stream
    "exo_celerite_loglike_fwd_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32, _c_dp],
    ),
    # t, resid, diag, n_diag, n, coef_real, n_real, coef_complex, n_complex, pair_kind, n_draw, gloglike, state,
    # state_doubles, n_chunks, gresid, gdiag, gdiag_sum, gcoef_real, gcoef_complex, stream
    "exo_celerite_loglike_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp],
    ),
    "exo_celerite_loglike_obs_fwd_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32, _c_dp],
    ),
    "exo_celerite_loglike_obs_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp],
    ),
    # the same pair with model / gmodel CADENCE-MAJOR ([n][n_draw])
    "exo_celerite_loglike_obs_fwd_cm_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32, _c_dp],
    ),
    "exo_celerite_loglike_obs_vjp_cm_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp],
    ),
    # t, obs, model (host struct exo_sparse_model*), diag, ... as the obs pair; gvals instead of gmodel
    "exo_celerite_loglike_sparse_fwd_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32, _c_dp],
    ),
    "exo_celerite_loglike_sparse_vjp_f64": (
        ctypes.c_int,
        [_c_dp, _c_dp, _c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp, _i64, _i32,
         _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp],
    ),
    # workspace, workspace_bytes, n_cad, n_draw, n_planet, flags, out (host struct)
    "exo_transit_flux_sparse_model": (ctypes.c_int, [_c_dp, _i64, _i64, _i64, _i32, _u32, _c_dp]),
    "exo_sparse_merge_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "exo_sparse_merge_layout": (ctypes.c_int, [_i64, _i64, _i32, ctypes.POINTER(_i64)]),
    "exo_sparse_model_merge_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _i64, _i32, _u32, _c_dp, _i64, _c_dp, _c_dp]),
    "exo_sparse_model_merged": (ctypes.c_int, [_c_dp, _i64, _i64, _i64, _i32, _c_dp]),
    "exo_sparse_model_merge_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _i64, _i32, _u32, _c_dp, _i64, _c_dp, _c_dp, _c_dp]),
    # t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, gvals, gparams, gld, flux_dot,
    # workspace, workspace_bytes, reuse_runs, stream
    "exo_transit_flux_vjp_sparse_f64": (
        ctypes.c_int,
        [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64,
         _i32, _c_dp],
    ),
    "exo_celerite_dot_tril_f64": (ctypes.c_int, [_c_dp, _c_dp, _i64, _i64, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _c_dp,
                                                 _c_dp]),
    "exo_celerite_predict_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _i32, _c_dp, _i64, _c_dp, _i64, _c_dp,
                                                _c_dp]),
    "exo_radial_velocity_fwd_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _i32, _c_dp, _c_dp]),
    "exo_radial_velocity_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _i32, _c_dp, _c_dp, _c_dp]),
    "exo_orbit_vector_fwd_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp]),
    "exo_orbit_vector_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _c_dp]),
    # t, n_cad, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, n_draw, n_planet, flags, obs, ivar, n_ivar,
    # chi2, gparams, gld, workspace, workspace_bytes, stream
    "exo_transit_flux_jac_doubles": (_i64, [_i64, _i64, _i32]),
    "exo_transit_flux_fwd_jac_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32,
                                                    _c_dp, _c_dp, _i64, _c_dp, _i64, _c_dp]),
    "exo_transit_flux_jac_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _i32, _u32, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _c_dp]),
    "exo_transit_chi2_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32,
                                                _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _c_dp]),
    # ..., flags, ttv_edges, ttv_shift, n_edge, obs, ivar, n_ivar, chi2, gparams, gld, gshift, workspace, workspace_bytes, stream
    "exo_transit_chi2_ttv_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _c_dp, _i64, _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _i32, _u32,
                                                    _c_dp, _c_dp, _i32, _c_dp, _c_dp, _i64, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp,
                                                    _i64, _c_dp]),
    # period, ds, ps, t0, ds, ps, ttv (host), ttv_ds (host), n_transit (host), n_draw, n_planet, n_edge, edges, shift, stream
    "exo_ttv_tables_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _c_dp, _i64, _i64, _c_dp, _c_dp, _c_dp, _i64, _i32, _i32, _c_dp,
                                          _c_dp, _c_dp]),
    "exo_ttv_tables_vjp_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _c_dp, _i64, _i64, _c_dp, _c_dp, _c_dp, _i64, _i32, _i32, _c_dp,
                                              _c_dp, _c_dp, _c_dp]),
    "exo_nuts_f64": (ctypes.c_int, [_c_dp, _i64, _i32, _i32, ctypes.c_double, _i32, _c_dp]),
    "exo_sho_coefficients_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _u32, ctypes.c_double, _i64, _c_dp, _c_dp, _c_dp]),
    "exo_sho_coefficients_vjp_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _u32, ctypes.c_double, _i64, _c_dp, _c_dp, _c_dp,
                                                    _c_dp, _c_dp]),
    # amp, freq, damp, flags (host arrays), n_terms, eps, n, ...
    "exo_sho_coefficients_multi_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i32, ctypes.c_double, _i64, _c_dp, _c_dp, _c_dp]),
    "exo_sho_coefficients_multi_vjp_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _i32, ctypes.c_double, _i64, _c_dp, _c_dp,
                                                          _c_dp, _c_dp, _c_dp]),
    "exo_transit_sparse_scatter_f64": (ctypes.c_int, [_c_dp, _i64, _i64, _i64, _i32, ctypes.c_uint32, _i32, _c_dp, _c_dp]),
    "exo_pack_records_f64": (ctypes.c_int, [_c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _c_dp]),
    "exo_pack_records_vjp_f64": (ctypes.c_int, [_c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    # cols, draw_stride, planet_stride, defaults, ld_cols, ld_draw_stride (host arrays), n_draw, n_planet, flags, ...
    "exo_pack_records_cols_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp, _c_dp,
                                                 _c_dp]),
    "exo_pack_records_cols_vjp_f64": (ctypes.c_int, [_c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _i64, _i32, _u32, _c_dp,
                                                     _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
}

class SparseModel(ctypes.Structure):
    """exo_sparse_model of include/exoplanet_amd.h (a HOST struct of device pointers and strides)"""
    _fields_ = [("nseg", ctypes.c_void_p), ("seg", ctypes.c_void_p), ("off", ctypes.c_void_p), ("vals", ctypes.c_void_p),
                ("seg_row", _i64), ("off_row", _i64), ("val_row", _i64), ("seg_step", _i32), ("hi_at", _i32),
                ("row_of_draw", ctypes.c_void_p)]


_ERRORS = {1: "invalid argument", 2: "kernel launch failed", 3: "workspace too small"}

_lib = None


class ExtensionMissingError(RuntimeError):
    pass


def exported_symbols():
    """Names include/exoplanet_amd.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGNATURES)


def load():
    """dlopen the HIP library; raises ExtensionMissingError loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissingError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  exoplanet_amd has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ExtensionMissingError(f"{LIB_PATH} does not export {name}: stale build?") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.exo_abi_version()
    if got != ABI_VERSION:
        raise ExtensionMissingError(f"ABI mismatch: library reports {got}, bindings expect {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what}: {_ERRORS.get(status, f'error {status}')}")
