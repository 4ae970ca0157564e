"""Device ops behind the reference's ``exoplanet.compat.ops`` names.

Drop-in for the three callables the reference imports from exoplanet_core
(/root/reference/src/exoplanet/compat.py:27,56):

    ops.kepler(M, ecc) -> (sinf, cosf)                 keplerian.py:333,818
    ops.quad_solution_vector(b, r) -> s[..., 3]        limb_dark.py:24
    ops.contact_points(a, e, cosw, sinw, cosi, sini, L)
        -> (M_left, M_right, flag)                     keplerian.py:744-753

plus the fused ops that replace whole sub-graphs: ``transit_flux`` (everything between
the time array and the flux array; ``ttv=`` for a TTVOrbit's timing tables),
``transit_flux_dot`` / ``transit_flux_value_and_vjp`` (value and gradient in one sweep),
``pack_records`` (the constructor algebra) and ``radial_velocity``.  All take float64 ROCm tensors and
run hand-written HIP kernels through the C ABI (include/exoplanet_amd.h); each
has value + gradient (torch.autograd).  There is no CPU / eager fallback.
"""
import os

import torch

from . import _lib

NPAR = 20
(P_N, P_TP, P_ECC, P_COSW, P_SINW, P_COSI, P_SINI, P_AOR, P_ROR, P_T0, P_PERIOD, P_TS, P_TE,
 P_FRATIO, P_TS2, P_TE2, P_CLIGHT) = range(17)
FLAG_PER_PLANET = 1
FLAG_WINDOW = 2
FLAG_SECONDARY = 4
PACK_CIRCULAR = 8
FLAG_EXACT_SCAN = 16
FLAG_SPARSE = 32
FLAG_LIGHT_DELAY = 64
FLAG_CADENCE_MAJOR = 128   # summed dense flux / its cotangent as [cadence][draw] arrays (include/exoplanet_amd.h)
FLAG_SORTED_TIMES = 256    # t has been checked to be non-decreasing (known_sorted): windows and runs in one launch
NIN = 10
(IN_PERIOD, IN_T0, IN_B, IN_ECC, IN_OMEGA, IN_R, IN_MSTAR, IN_RSTAR, IN_MPLANET, IN_SBR) = range(10)
MAX_PLANETS = 16
MAX_SUBEXP = 63


_SORTED = {}
_VOUCHED = {}


def _series_key(t):
    return (t.device.index, t.data_ptr(), t.numel(), t.stride(0) if t.dim() else 0)


def vouch_sorted(t):
    """The caller's word that the device tensor ``t`` is non-decreasing (no NaN) and STAYS so for as long as this buffer is
    a time array -- whatever is written into it later, by whatever means.  (Keyed by device, address and extent; at most
    sixteen series are held: a seventeenth releases the oldest, with a RuntimeWarning.)  The sweeps then carry FLAG_SORTED_TIMES
    (windows and runs in one launch) also where the torch layer cannot see for itself: inside a hipGraph capture, whose
    launches keep their flags at every replay.  The array is looked at once, now (one host synchronisation; not during a
    capture); ValueError if it is not sorted.  A sampler's time array is the use: fixed for the whole run.
    ``release_sorted(t)`` takes the word back."""
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError("vouch_sorted: a device tensor, please")
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("vouch_sorted looks at the array (a host synchronisation): call it before the capture")
    if t.numel() > 1 and not bool((t[1:] >= t[:-1]).all()):
        raise ValueError("vouch_sorted: the times are not non-decreasing (or hold a NaN)")
    key = _series_key(t)
    _VOUCHED.pop(key, None)
    if len(_VOUCHED) >= 16:
        # the table holds sixteen series (an entry keeps its time array alive): the OLDEST word is taken back, with a warning --
        # that series falls back to the sweep's own device check inside later captures (slower, never wrong)
        import warnings

        oldest = next(iter(_VOUCHED))
        _VOUCHED.pop(oldest)
        warnings.warn("vouch_sorted: more than 16 vouched series; the oldest one's word has been released (release_sorted "
                      "what you no longer use)", RuntimeWarning, stacklevel=2)
    _VOUCHED[key] = t        # (keeps the storage alive: the address cannot pass to another series meanwhile)
    return t


def release_sorted(t):
    _VOUCHED.pop(_series_key(t), None)


def known_sorted(t, unknown=True, nan_ok=True):
    """True if ``t`` is non-decreasing.  A device tensor is looked at once (one host synchronisation) and remembered
    by storage and version -- the entry keeps the tensor alive, so its address cannot be handed to another series
    meanwhile: a sampler evaluates on the same time array every step, and a step that is being captured into a hipGraph
    must not synchronise (``unknown``: the answer then, for a tensor never looked at).  ``nan_ok=False``: every
    neighbouring pair must compare as ordered -- a NaN among the times makes the answer False (what the sweep's own
    device check says: it then solves every cadence instead of searching).  A tensor without a version counter
    (inference mode) is ``unknown``.  The version counter sees torch's own in-place writes only: a buffer that is
    written behind torch's back (``t.data.copy_``, DLPack / ctypes consumers) must not rely on this cache -- see
    ``vouch_sorted`` and INTEGRATION.md."""
    if not t.is_cuda:
        return bool((t[1:] >= t[:-1]).all()) if not nan_ok else not bool((t[1:] < t[:-1]).any())
    # (by storage address, extent and version: `t.detach()` is a new object on the same storage with the same version
    # counter; the entry holds a tensor on that storage, so the address is not reused while it is here)
    try:
        key = _series_key(t) + (t._version,)
    except RuntimeError:                 # inference tensors do not track versions
        return unknown
    hit = _SORTED.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            return unknown   # cannot look during a capture; the warm-up runs before it did
        both = torch.stack([~(t[1:] < t[:-1]).any(), (t[1:] >= t[:-1]).all()]).cpu()     # one synchronisation
        if len(_SORTED) >= 8:      # (an entry keeps its time array alive: a handful of series, not dozens -- ADVICE r2)
            _SORTED.clear()
        hit = _SORTED[key] = (bool(both[0]), bool(both[1]), t)
    return hit[0] if nan_ok else hit[1]


def _sorted_flag(t):
    """FLAG_SORTED_TIMES when the sweep may skip its own check of ``t``: never on a guess, never with a NaN among the times,
    and -- ADVICE r3 -- never baked into a captured launch on the strength of the version cache (a replay runs on whatever
    the buffer holds THEN; the cache saw what it held at capture): inside a capture only a series the caller vouched for
    (``vouch_sorted``) gets the flag, everything else keeps the device's own check, every replay."""
    if os.environ.get("EXO_CHECK_SORTED_ON_DEVICE") == "1":     # (A/B: the sweep's own check, every call)
        return 0
    if t.numel() <= 1:
        return 0
    if _series_key(t) in _VOUCHED:
        return FLAG_SORTED_TIMES
    if torch.cuda.is_current_stream_capturing():
        return 0
    return FLAG_SORTED_TIMES if known_sorted(t, unknown=False, nan_ok=False) else 0


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(x, name):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(
            f"{name} lives on {x.device}: exoplanet_amd ops run only on a ROCm device "
            "(HIP kernels through libexoplanet_amd.so); there is no CPU fallback"
        )
    if x.dtype != torch.float64:
        raise TypeError(f"{name} must be float64 (the reference casts everything to float64, utils.py:18)")
    return x.contiguous()


def _ptr(x):
    return 0 if x is None else x.data_ptr()


def is_cadence_major(x):
    """a (D, N) float64 device tensor laid out [cadence][draw] -- the transposed view of a contiguous (N, D) array: what
    the light-curve sweep returns under FLAG_CADENCE_MAJOR and what the celerite kernels read (and write the cotangent
    of) with every wave's accesses contiguous, a lane being a draw"""
    return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.shape[0] > 1
            and x.stride() == (1, x.shape[0]))


# ------------------------------------------------------------------------------
# kepler
# ------------------------------------------------------------------------------
class _Kepler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, ecc):
        M = _dev(M, "M")
        ecc = _dev(ecc, "ecc")
        if M.shape != ecc.shape:
            raise ValueError("kepler: M and ecc must have the same shape (the caller broadcasts, keplerian.py:333)")
        sinf = torch.empty_like(M)
        cosf = torch.empty_like(M)
        lib = _lib.load()
        with torch.cuda.device(M.device):
            _lib.check(lib.exo_kepler_f64(_ptr(M), _ptr(ecc), _ptr(sinf), _ptr(cosf), M.numel(), _stream(M)),
                       "exo_kepler_f64")
        ctx.save_for_backward(sinf, cosf, ecc)
        return sinf, cosf

    @staticmethod
    def backward(ctx, gs, gc):
        sinf, cosf, e = ctx.saved_tensors
        # df/dM = (1+e cosf)^2/(1-e^2)^{3/2},  df/de = (2+e cosf) sinf/(1-e^2)
        ome2 = 1 - e * e
        gf = gs * cosf - gc * sinf
        dfdM = (1 + e * cosf) ** 2 / ome2 ** 1.5
        dfde = (2 + e * cosf) * sinf / ome2
        return gf * dfdM, gf * dfde


def kepler(M, ecc):
    """sin and cos of the true anomaly; same shape as ``M`` (= shape of ``ecc``)."""
    return _Kepler.apply(M, ecc)


# ------------------------------------------------------------------------------
# quad_solution_vector
# ------------------------------------------------------------------------------
class _QuadSV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, r):
        b = _dev(b, "b")
        r = _dev(r, "r")
        if b.shape != r.shape:
            raise ValueError("quad_solution_vector: b and r must have the same shape")
        need = any(ctx.needs_input_grad)   # (not .requires_grad: forward runs under no_grad and .contiguous() may copy)
        s = torch.empty(b.shape + (3,), dtype=torch.float64, device=b.device)
        dsdb = torch.empty_like(s) if need else None
        dsdr = torch.empty_like(s) if need else None
        lib = _lib.load()
        with torch.cuda.device(b.device):
            _lib.check(
                lib.exo_quad_solution_vector_f64(_ptr(b), _ptr(r), _ptr(s), _ptr(dsdb), _ptr(dsdr), b.numel(),
                                                 _stream(b)),
                "exo_quad_solution_vector_f64",
            )
        if need:
            ctx.save_for_backward(dsdb, dsdr)
        return s

    @staticmethod
    def backward(ctx, gs):
        dsdb, dsdr = ctx.saved_tensors
        return (gs * dsdb).sum(-1), (gs * dsdr).sum(-1)


def quad_solution_vector(b, r):
    """s[..., 3] for separations ``b`` (|b| is used) and radius ratios ``r``."""
    return _QuadSV.apply(b, r)


@torch.no_grad()
def quad_solution_vector_derivs(b, r):
    """(s, ds/db, ds/dr), each [..., 3], as plain tensors -- for callers that carry their own
    differentiation (the PyTensor Op of compat_pytensor.py)."""
    b = _dev(b, "b")
    r = _dev(r, "r")
    if b.shape != r.shape:
        raise ValueError("quad_solution_vector: b and r must have the same shape")
    s = torch.empty(b.shape + (3,), dtype=torch.float64, device=b.device)
    dsdb, dsdr = torch.empty_like(s), torch.empty_like(s)
    lib = _lib.load()
    with torch.cuda.device(b.device):
        _lib.check(
            lib.exo_quad_solution_vector_f64(_ptr(b), _ptr(r), _ptr(s), _ptr(dsdb), _ptr(dsdr), b.numel(), _stream(b)),
            "exo_quad_solution_vector_f64",
        )
    return s, dsdb, dsdr


# ------------------------------------------------------------------------------
# contact_points (no gradient: it only selects cadences, keplerian.py:769-775)
# ------------------------------------------------------------------------------
@torch.no_grad()
def contact_points(a, e, cosw, sinw, cosi, sini, L):
    args = torch.broadcast_tensors(*[x.detach() for x in (a, e, cosw, sinw, cosi, sini, L)])
    args = [_dev(x, n) for x, n in zip(args, ("a", "e", "cosw", "sinw", "cosi", "sini", "L"))]
    ref = args[0]
    Ml = torch.empty_like(ref)
    Mr = torch.empty_like(ref)
    flag = torch.empty(ref.shape, dtype=torch.int32, device=ref.device)
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        _lib.check(
            lib.exo_contact_points_f64(*[_ptr(x) for x in args], _ptr(Ml), _ptr(Mr), _ptr(flag), ref.numel(),
                                       _stream(ref)),
            "exo_contact_points_f64",
        )
    return Ml, Mr, flag


# ------------------------------------------------------------------------------
# fused transit flux
# ------------------------------------------------------------------------------
def _transit_args(t, texp, stencil_dt, stencil_w, params, ld, flags):
    t = _dev(t, "t")
    params = _dev(params, "params")
    ld = _dev(ld, "ld")
    if t.dim() != 1:
        raise ValueError("t must be 1-D (n_cad,)")
    if params.dim() != 3 or params.shape[-1] != NPAR:
        raise ValueError(f"params must be (n_draw, n_planet, {NPAR})")
    D, P, _ = params.shape
    nld = 6 if flags & FLAG_SECONDARY else 3
    if ld.shape != (D, nld):
        raise ValueError(f"ld must be (n_draw, {nld})")
    if not 1 <= P <= MAX_PLANETS:
        raise ValueError(f"1 <= n_planet <= {MAX_PLANETS}")
    if texp is None:
        n_texp, n_sub = 0, 1
        stencil_dt = stencil_w = None
    else:
        texp = _dev(texp, "texp").reshape(-1)
        stencil_dt = _dev(stencil_dt, "stencil_dt")
        stencil_w = _dev(stencil_w, "stencil_w")
        n_texp, n_sub = texp.numel(), stencil_dt.numel()
        if n_texp not in (1, t.numel()):
            raise ValueError("texp must be a scalar or have one entry per cadence")
        if stencil_w.numel() != n_sub or not 1 <= n_sub <= MAX_SUBEXP:
            raise ValueError(f"stencil must have 1..{MAX_SUBEXP} points")
    return t, texp, n_texp, stencil_dt, stencil_w, n_sub, params, ld, D, P


def _ttv_args(ttv, D, P):
    """(edges (D, P, E), shift (D, P, E + 1)) of a timing-variation table (include/exoplanet_amd.h)"""
    if ttv is None:
        return None, None, 0
    edges, shift = ttv
    edges = _dev(edges, "ttv edges")
    shift = _dev(shift, "ttv shift")
    if edges.dim() != 3 or tuple(edges.shape[:2]) != (D, P) or edges.shape[2] < 1:
        raise ValueError("ttv edges must be (n_draw, n_planet, n_edge >= 1)")
    if tuple(shift.shape) != (D, P, edges.shape[2] + 1):
        raise ValueError("ttv shift must be (n_draw, n_planet, n_edge + 1)")
    return edges, shift, int(edges.shape[2])


# the Jacobian route of _TransitFlux: on from this many samples per cadence (with one sample a cadence the rows cost as much
# HBM traffic as the second sweep costs arithmetic: C3 gains nothing), up to this many bytes of rows (EXO_JAC_ROUTE=0: A/B)
_JAC_ROUTE = [os.environ.get("EXO_JAC_ROUTE", "1") != "0"]
_JAC_MIN_SUB = 2
_JAC_MAX_BYTES = 8 << 30


_JAC_DECISIONS = {}     # (device index, bytes of rows) -> bool: the first answer stands (warm-up, capture and replay agree)


def _jac_fits(nbytes, device):
    """the rows of derivatives are worst-case sized (16 doubles per (draw, planet, cadence), a few per cent touched): take the
    route only if they fit the cap AND half of what the device can still give (a smaller-memory device must fall
    back to the two-sweep route rather than run out).  "Can still give" = the driver's free bytes + what torch's caching
    allocator holds but has not handed out -- a warm sampler's reserved pool would otherwise read as a full device and flip
    the route between steps.  The decision is remembered per (device, size): a capture takes what its warm-up took."""
    if nbytes > _JAC_MAX_BYTES:
        return False
    key = (torch.device(device).index, int(nbytes))
    hit = _JAC_DECISIONS.get(key)
    if hit is not None:
        return hit
    try:
        if torch.cuda.is_current_stream_capturing():
            return True          # (no warm-up call asked before this capture: nothing to agree with; not remembered)
        free, _ = torch.cuda.mem_get_info(device)
        free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        fits = 2 * nbytes <= free
    except Exception:
        fits = True
    _JAC_DECISIONS[key] = fits
    return fits


_JAC_CALLS = [0]        # forward sweeps that took the route (tests look at it)


class _TransitFlux(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, texp, stencil_dt, stencil_w, params, ld, flags, ttv_edges, ttv_shift, grad_mode=True):
        t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(
            t, texp, stencil_dt, stencil_w, params, ld, flags)
        flags |= _sorted_flag(t)
        edges, shift, n_edge = _ttv_args(None if ttv_edges is None else (ttv_edges, ttv_shift), D, P)
        N = t.numel()
        shape = (D, N, P) if flags & FLAG_PER_PLANET else (D, N)
        if flags & FLAG_CADENCE_MAJOR:
            if flags & FLAG_PER_PLANET:
                raise ValueError("FLAG_CADENCE_MAJOR is a layout of the summed flux")
            flux = torch.empty((N, D), dtype=torch.float64, device=t.device).t()     # (D, N), draws innermost
        else:
            flux = torch.empty(shape, dtype=torch.float64, device=t.device)
        lib = _lib.load()
        nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
        ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
        # the Jacobian route (include/exoplanet_amd.h, exo_transit_flux_fwd_jac_f64): when the gradient will be asked for and a
        # cadence is several samples, the value sweep keeps every solved cadence's row of derivatives and backward() is a
        # contraction instead of a second sweep
        n_jac = lib.exo_transit_flux_jac_doubles(N, D, P)
        # (the memory question last: it is a driver call -- ADVICE r5)
        use_jac = (_JAC_ROUTE[0] and grad_mode and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5])
                   and n_sub >= _JAC_MIN_SUB and not n_edge and n_texp <= 1
                   and not flags & (FLAG_PER_PLANET | FLAG_SPARSE | FLAG_EXACT_SCAN | FLAG_LIGHT_DELAY)
                   and _jac_fits(8 * n_jac, t.device))
        ctx.jac = None
        if use_jac:
            _JAC_CALLS[0] += 1
            jac = torch.empty(n_jac, dtype=torch.float64, device=t.device)
            with torch.cuda.device(t.device):
                _lib.check(lib.exo_transit_flux_fwd_jac_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                            _ptr(ld), D, P, flags, _ptr(flux), _ptr(jac), n_jac, _ptr(ws), nbytes,
                                                            _stream(t)), "exo_transit_flux_fwd_jac_f64")
            ctx.jac = (jac, ws, nbytes)
            ctx.save_for_backward(t, texp, sdt, sw, params, ld, edges, shift)
            ctx.meta = (n_texp, n_sub, D, P, flags)
            return flux
        with torch.cuda.device(t.device):
            if n_edge:
                _lib.check(
                    lib.exo_transit_flux_ttv_fwd_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                     _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                     n_edge, _ptr(flux), _ptr(ws), nbytes, _stream(t)),
                    "exo_transit_flux_ttv_fwd_f64",
                )
            else:
                _lib.check(
                    lib.exo_transit_flux_fwd_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                 _ptr(params), _ptr(ld), D, P, flags, _ptr(flux), _ptr(ws), nbytes,
                                                 _stream(t)),
                    "exo_transit_flux_fwd_f64",
                )
        ctx.save_for_backward(t, texp, sdt, sw, params, ld, edges, shift)
        ctx.meta = (n_texp, n_sub, D, P, flags)
        return flux

    @staticmethod
    def backward(ctx, gflux):
        t, texp, sdt, sw, params, ld, edges, shift = ctx.saved_tensors
        n_texp, n_sub, D, P, flags = ctx.meta
        if ctx.jac is not None:
            jac, ws, nbytes = ctx.jac
            N = t.numel()
            if is_cadence_major(gflux):
                flags |= FLAG_CADENCE_MAJOR
            else:
                flags &= ~FLAG_CADENCE_MAJOR
                gflux = _dev(gflux, "gflux")
            gparams, gld = torch.empty_like(params), torch.empty_like(ld)
            with torch.cuda.device(t.device):
                _lib.check(_lib.load().exo_transit_flux_jac_vjp_f64(_ptr(gflux), N, D, P, flags, _ptr(jac), _ptr(ws), nbytes,
                                                                    _ptr(gparams), _ptr(gld), 0, _stream(t)),
                           "exo_transit_flux_jac_vjp_f64")
            return None, None, None, None, gparams, gld, None, None, None, None
        ttv = None if edges is None else (edges, shift)
        _, gparams, gld, _, gshift = _vjp(t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P, flags, gflux, False,
                                          ttv=ttv)
        return None, None, None, None, gparams, gld, None, None, gshift, None


def _vjp(t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P, flags, gflux, want_flux, events=(None, None), ttv=None):
    N = t.numel()
    flags |= _sorted_flag(t)
    shape = (D, N, P) if flags & FLAG_PER_PLANET else (D, N)
    if isinstance(gflux, torch.Tensor) and tuple(gflux.shape) != shape:
        raise ValueError(f"gflux must have shape {shape}")
    edges, shift, n_edge = _ttv_args(ttv, D, P)
    # the cotangent as it comes: cadence-major (the celerite kernels' gradient of a cadence-major model) or rows.  Only the
    # run-enumeration sweeps read it cadence-major (include/exoplanet_amd.h): anything else gets rows
    cm_sweep = (n_texp <= 1 and not flags & (FLAG_EXACT_SCAN | FLAG_PER_PLANET)
                and not (n_edge and flags & (FLAG_SECONDARY | FLAG_LIGHT_DELAY)))
    if is_cadence_major(gflux) and cm_sweep:
        flags |= FLAG_CADENCE_MAJOR
    elif flags & FLAG_CADENCE_MAJOR and want_flux and cm_sweep:
        gflux = _dev(gflux, "gflux").t().contiguous().t()
    else:
        flags &= ~FLAG_CADENCE_MAJOR
        gflux = _dev(gflux, "gflux")
    lib = _lib.load()
    nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
    ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
    gparams = torch.empty_like(params)
    gld = torch.empty_like(ld)
    dot = torch.empty(D, dtype=torch.float64, device=t.device)
    flux = None
    if want_flux and not flags & FLAG_SPARSE:
        flux = (torch.empty((N, D), dtype=torch.float64, device=t.device).t() if flags & FLAG_CADENCE_MAJOR
                else torch.empty(shape, dtype=torch.float64, device=t.device))
    gshift = torch.empty_like(shift) if n_edge else None
    with torch.cuda.device(t.device):
        if n_edge:
            _lib.check(
                lib.exo_transit_flux_ttv_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                 _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                 n_edge, _ptr(gflux), _ptr(flux), _ptr(gparams), _ptr(gld),
                                                 _ptr(gshift), _ptr(dot), _ptr(ws), nbytes, _stream(t)),
                "exo_transit_flux_ttv_vjp_f64",
            )
        else:
            _lib.check(
                lib.exo_transit_flux_vjp_ev_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                _ptr(params), _ptr(ld), D, P, flags, _ptr(gflux), _ptr(flux),
                                                _ptr(gparams), _ptr(gld), _ptr(dot), _ptr(ws), nbytes, _stream(t),
                                                events[0], events[1]),
                "exo_transit_flux_vjp_f64",
            )
    if flags & FLAG_SPARSE:
        flux = _sparse_from_ws(ws, N, D, P, flags)
    return flux, gparams, gld, dot, gshift


def _sparse_from_ws(ws, N, D, P, flags):
    """the sweep's output lives in its workspace (include/exoplanet_amd.h, EXO_FLAG_SPARSE)"""
    import ctypes

    lay = (ctypes.c_int64 * 5)()
    _lib.check(_lib.load().exo_transit_flux_sparse_layout(N, D, P, lay), "exo_transit_flux_sparse_layout")
    return SparseFlux(ws, list(lay), N, D, P, 2 if flags & FLAG_SECONDARY else 1)


def transit_flux(t, params, ld, texp=None, stencil_dt=None, stencil_w=None, flags=0, ttv=None):
    """Fused light curve for ``n_draw`` parameter sets.

    t (n_cad,), params (n_draw, n_planet, 20) (slot meaning: include/exoplanet_amd.h),
    ld (n_draw, 3|6).  Returns (n_draw, n_cad) or, with FLAG_PER_PLANET,
    (n_draw, n_cad, n_planet).  Differentiable w.r.t. ``params`` and ``ld``.
    ``ttv = (edges (n_draw, n_planet, E), shift (n_draw, n_planet, E + 1))``: transit-timing
    variations (every time is measured from the transit of its bin); differentiable
    w.r.t. ``shift`` as well.
    """
    edges, shift = (None, None) if ttv is None else ttv
    return _TransitFlux.apply(t, texp, stencil_dt, stencil_w, params, ld, int(flags),
                              None if edges is None else edges.detach(), shift, torch.is_grad_enabled())


@torch.no_grad()
def transit_flux_value_and_vjp(t, params, ld, gflux, texp=None, stencil_dt=None, stencil_w=None, flags=0,
                               events=(None, None), ttv=None):
    """One sweep over t: flux AND the cotangents of (params, ld) for a given
    ``gflux`` -- 24 B per (draw, cadence).  Returns (flux, gparams, gld), plus the
    cotangent of the shift table when ``ttv`` is given.
    ``events``: optional (hipEvent_t, hipEvent_t) handles (ints) recorded around
    the dominant kernel (profiling hook of the C ABI)."""
    flags = int(flags)
    t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(t, texp, stencil_dt, stencil_w, params, ld,
                                                                      flags)
    out = _vjp(t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P, flags, gflux, True, events, ttv=ttv)
    return out[:3] if ttv is None else out[:3] + (out[4],)


class _TransitFluxDot(torch.autograd.Function):
    """(flux, L) with L[d] = sum_n gflux[d, n] * flux[d, n]: because the cotangent
    is known up front, forward runs the one-sweep value+vjp kernel and stores
    the parameter cotangents; backward only scales them by dL."""

    @staticmethod
    def forward(ctx, t, texp, stencil_dt, stencil_w, params, ld, gflux, flags, events, ttv_edges, ttv_shift):
        t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(
            t, texp, stencil_dt, stencil_w, params, ld, flags)
        ttv = None if ttv_edges is None else (ttv_edges, ttv_shift)
        flux, gparams, gld, dot, gshift = _vjp(t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P, flags, gflux, True,
                                               events, ttv=ttv)
        ctx.save_for_backward(gparams, gld, gshift)
        ctx.set_materialize_grads(False)  # never build a (D, N) zero cotangent for the detached flux
        if isinstance(flux, SparseFlux):
            flux = flux._ws      # the workspace travels as a (non-differentiable) output; the wrapper rebuilds the views
        ctx.mark_non_differentiable(flux)
        return flux, dot

    @staticmethod
    def backward(ctx, _gflux_unused, gdot):
        gparams, gld, gshift = ctx.saved_tensors
        if gdot is None:
            return (None,) * 11
        return (None, None, None, None, gdot[:, None, None] * gparams, gdot[:, None] * gld, None, None, None, None,
                None if gshift is None else gdot[:, None, None] * gshift)


def transit_flux_dot(t, params, ld, gflux, texp=None, stencil_dt=None, stencil_w=None, flags=0,
                     events=(None, None), ttv=None):
    """One-sweep value + gradient for a cotangent known in advance: returns
    ``(flux, L)`` with ``L[d] = (gflux[d] * flux[d]).sum()``; ``L`` is
    differentiable w.r.t. ``params`` and ``ld`` (and the ``ttv`` shift table);
    flux itself is returned detached.  With ``flags | FLAG_SPARSE`` no dense flux array is written:
    the first return value is a :class:`SparseFlux` (runs of cadences + their values)."""
    edges, shift = (None, None) if ttv is None else ttv
    out = _TransitFluxDot.apply(t, texp, stencil_dt, stencil_w, params, ld, gflux, int(flags), events,
                                None if edges is None else edges.detach(), shift)
    if int(flags) & FLAG_SPARSE:
        return _sparse_from_ws(out[0], t.numel(), params.shape[0], params.shape[1], int(flags)), out[1]
    return out


class _TransitChi2(torch.autograd.Function):
    """chi2[d] = sum_n w_n ((f[d, n] - obs[n])^2 - obs[n]^2) over the cadences solved for draw d, and -- computed in the
    same call, like transit_flux_dot -- its gradient with respect to params and ld (exo_transit_chi2_vjp_f64)."""

    @staticmethod
    def forward(ctx, t, texp, stencil_dt, stencil_w, params, ld, obs, ivar, flags, ttv_edges, ttv_shift):
        if ctx.needs_input_grad[6] or ctx.needs_input_grad[7]:
            # (the kernels never hold the per-cadence residuals of every draw; see white_noise_loglike for what is offered)
            raise NotImplementedError(
                "transit_chi2 is not differentiable with respect to obs / ivar: per-draw error bars go through "
                "white_noise_loglike(yerr=(n_draw, 1) tensor), anything else through get_light_curve(total=True) and torch")
        t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(t, texp, stencil_dt, stencil_w, params, ld, flags)
        flags |= _sorted_flag(t)
        edges, shift, n_edge = _ttv_args(None if ttv_edges is None else (ttv_edges, ttv_shift), D, P)
        N = t.numel()
        obs = _dev(obs, "obs")
        ivar = _dev(ivar, "ivar").reshape(-1)
        if tuple(obs.shape) != (N,):
            raise ValueError("obs must have shape (n_cad,)")
        if ivar.numel() not in (1, N):
            raise ValueError("ivar must be a scalar or have one entry per cadence")
        lib = _lib.load()
        nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
        ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
        chi2 = torch.empty(D, dtype=torch.float64, device=t.device)
        gparams = torch.empty_like(params)
        gld = torch.empty_like(ld)
        gshift = torch.empty_like(shift) if n_edge else None
        with torch.cuda.device(t.device):
            if n_edge:
                _lib.check(lib.exo_transit_chi2_ttv_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                            _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                            n_edge, _ptr(obs), _ptr(ivar), ivar.numel(), _ptr(chi2),
                                                            _ptr(gparams), _ptr(gld), _ptr(gshift), _ptr(ws), nbytes,
                                                            _stream(t)), "exo_transit_chi2_ttv_vjp_f64")
                ctx.save_for_backward(gparams, gld, gshift)
            else:
                _lib.check(lib.exo_transit_chi2_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                        _ptr(ld), D, P, flags, _ptr(obs), _ptr(ivar), ivar.numel(), _ptr(chi2),
                                                        _ptr(gparams), _ptr(gld), _ptr(ws), nbytes, _stream(t)),
                           "exo_transit_chi2_vjp_f64")
                ctx.save_for_backward(gparams, gld)
        return chi2

    @staticmethod
    def backward(ctx, gchi2):
        gparams, gld, *rest = ctx.saved_tensors
        gshift = gchi2[:, None, None] * rest[0] if rest else None
        return (None, None, None, None, gchi2[:, None, None] * gparams, gchi2[:, None] * gld, None, None, None, None, gshift)


_CONSTS = {}
_WN_CACHE = {}     # white_noise_loglike: (series, error bars, mean) -> residuals, weights, constants


def _const(value, device):
    """a one-element float64 device tensor holding ``value`` (cached per device: created once, outside any capture)"""
    key = (float(value), str(device))
    if key not in _CONSTS:
        if len(_CONSTS) >= 256:      # a handful of error bars in practice: never let a sweep over values grow it
            _CONSTS.clear()
        _CONSTS[key] = torch.tensor([float(value)], dtype=torch.float64, device=device)
    return _CONSTS[key]


def transit_chi2(t, params, ld, obs, ivar, texp=None, stencil_dt=None, stencil_w=None, flags=0, ttv=None):
    """White-noise misfit of ONE observed series ``obs`` (n_cad,) against the light curves of ``n_draw`` parameter
    sets, relative to an empty light curve, without any (n_draw, n_cad) array:
    ``chi2[d] = sum_n ivar_n ((flux[d, n] - obs[n])**2 - obs[n]**2)`` -- the sum runs over the cadences in which a
    planet of draw d can overlap the disk, everything else cancels.  Differentiable w.r.t. ``params`` and ``ld``
    (the gradient is computed in the same call).  ``ivar``: scalar or (n_cad,).  The Gaussian log-likelihood is
    ``-0.5 * (chi2 + (ivar * obs**2).sum()) + 0.5 * log(ivar / 2 pi).sum()``: :func:`white_noise_loglike`.
    ``ttv = (edges, shift)``: timing tables (transits only); differentiable w.r.t. ``shift`` as well."""
    edges, shift = (None, None) if ttv is None else ttv
    return _TransitChi2.apply(t, texp, stencil_dt, stencil_w, params, ld, obs, ivar, int(flags), edges, shift)


def _white_noise_terms(y, yerr, mean):
    """(y - mean, weights, sum w (y - mean)^2, sum log(w / 2 pi), and the likelihood's constant term (the last minus the
    one before, halved)) of a white-noise likelihood"""
    # what depends on the data alone (residual series, weights, the two constants of the likelihood) is computed once per
    # (y, yerr, mean): a sampler calls this thousands of times with the same series
    key = None
    if isinstance(mean, (int, float)) and not y.requires_grad and (isinstance(yerr, (int, float)) or not yerr.requires_grad):
        # (keyed by the tensor OBJECTS, which the entry keeps alive -- an address alone could be handed to another series)
        key = (id(y), y._version, float(mean),
               float(yerr) if isinstance(yerr, (int, float)) else (id(yerr), yerr._version))
    hit = _WN_CACHE.get(key) if key is not None else None
    if hit is not None and not (hit[4] is y and (isinstance(yerr, (int, float)) or hit[5] is yerr)):
        hit = None
    if hit is None:
        if isinstance(yerr, (int, float)):
            ivar = _const(1.0 / (float(yerr) * float(yerr)), y.device)      # cached: no upload inside a hipGraph capture
        else:
            yerr = _dev(yerr, "yerr")
            ivar = (1.0 / (yerr * yerr)).reshape(-1)
        obs = y - mean
        n = y.numel()
        const = (ivar * obs * obs).sum() if ivar.numel() == n else ivar[0] * (obs * obs).sum()
        lognorm = torch.log(ivar / (2.0 * torch.pi)).sum() * (1.0 if ivar.numel() == n else float(n))
        hit = (obs, ivar, const, lognorm, y, yerr, 0.5 * (lognorm - const))
        if key is not None:
            if len(_WN_CACHE) >= 4:      # (an entry keeps y, its residuals and weights alive: a few series at most)
                _WN_CACHE.clear()
            _WN_CACHE[key] = hit
    return hit[:4] + (hit[6],)


def per_draw_yerr(yerr, n_cad):
    """``yerr`` as a per-draw error bar: a tensor that broadcasts against a (n_draw, n_cad) light curve along the
    draws only -- 0-d, or last dimension 1 -- flattened to (n_draw | 1,); None for anything else (a number, a
    per-cadence vector)."""
    if not isinstance(yerr, torch.Tensor):
        return None
    if yerr.dim() == 0 or yerr.shape[-1] == 1 or (yerr.numel() == 1 and n_cad != 1):
        return yerr.reshape(-1)
    return None


def _check_data_terms(y, yerr, mean):
    """the data-side terms of the white-noise likelihood carry no gradient through the fused kernels: refuse rather than
    return a partial gradient (jitter per chain: a (n_draw, 1) ``yerr``; anything else: the dense light curve + torch)"""
    for name, x in (("y", y), ("yerr", yerr), ("mean", mean)):
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError(
                f"white-noise likelihood: `{name}` requires grad, and the fused kernels differentiate only the orbit / "
                "limb-darkening parameters (plus per-draw error bars, yerr of shape (n_draw, 1)); use "
                "LimbDarkLightCurve.white_noise_log_likelihood (it falls back to the dense light curve) or "
                "get_light_curve(total=True) with torch")


def _scaled_loglike(unit_fn, y, yerr_d, mean):
    """log-likelihood for per-draw error bars from the unit-weight misfit: with w_d = 1 / yerr_d^2,
    ll_d = -w_d / 2 (chi2_1[d] + sum obs^2) + n / 2 log(w_d / 2 pi) -- differentiable in yerr_d through torch, in
    everything else through the kernels' own gradient; ``unit_fn(obs, one)`` returns chi2 at unit weights"""
    _check_data_terms(y, None, mean)
    obs, one, s2, _, _ = _white_noise_terms(y, 1.0, mean)
    w = 1.0 / (yerr_d * yerr_d)
    n = float(y.numel())
    chi2 = unit_fn(obs, one)
    if chi2.numel() != 1 and yerr_d.numel() not in (1, chi2.numel()):
        raise ValueError(f"white-noise likelihood: yerr holds {yerr_d.numel()} per-draw error bars, the parameters "
                         f"{chi2.numel()} draws -- one error bar, one per draw, or one system with many error bars")
    # (one system, many error bars -- a jitter chain per entry of yerr: the result has yerr's draws)
    return -0.5 * w * (chi2 + s2) + 0.5 * n * torch.log(w / (2.0 * torch.pi))


def white_noise_loglike(t, params, ld, y, yerr, mean=0.0, texp=None, stencil_dt=None, stencil_w=None, flags=0, ttv=None):
    """Gaussian log-likelihood (n_draw,) of the observed series ``y`` with independent errors ``yerr`` (a number, a
    per-cadence vector, or PER DRAW: a 0-d / (n_draw, 1) tensor, differentiable -- a jitter term sampled per chain)
    given ``mean + light curve`` -- the reference's ``pm.Normal("obs", mu=mean + lc, sigma=yerr, observed=y)`` for a
    batch of parameter sets, value and gradient in one call (:func:`transit_chi2`).  ``y``, a per-cadence ``yerr`` and
    ``mean`` are data here: a tensor among them that requires grad is refused, not silently dropped."""
    y = _dev(y, "y")
    kw = dict(texp=texp, stencil_dt=stencil_dt, stencil_w=stencil_w, flags=flags, ttv=ttv)
    yd = per_draw_yerr(yerr, y.numel())
    if yd is not None:
        return _scaled_loglike(lambda obs, one: transit_chi2(t, params, ld, obs, one, **kw), y, _dev(yd, "yerr"), mean)
    _check_data_terms(y, yerr, mean)
    obs, ivar, const, lognorm, _ = _white_noise_terms(y, yerr, mean)
    chi2 = transit_chi2(t, params, ld, obs, ivar, **kw)
    return -0.5 * (chi2 + const) + 0.5 * lognorm


class SparseFlux:
    """Output of a sparse sweep (include/exoplanet_amd.h, EXO_FLAG_SPARSE): for every list
    (draw, planet, event) the runs of cadences in which the planet can overlap the disk, and the
    flux of exactly those cadences; every other cadence has flux 0.  Tensors are views into the
    sweep's workspace (kept alive here)."""

    def __init__(self, ws, layout, n_cad, D, P, n_ev):
        off_nrun, off_runs, off_pre, off_vals, r_max = layout
        i32 = ws.view(torch.int32)
        n_list = D * P * n_ev
        self.n_cad, self.n_draw, self.n_planet, self.n_ev, self.r_max = n_cad, D, P, n_ev, r_max
        self.nrun = i32[off_nrun // 4: off_nrun // 4 + n_list].view(D, P, n_ev)
        self.runs = i32[off_runs // 4: off_runs // 4 + n_list * r_max * 4].view(D, P, n_ev, r_max, 4)
        self.pre_all = i32[off_pre // 4: off_pre // 4 + n_list * (r_max + 1)].view(D, P, n_ev, r_max + 1)
        self.vals = ws[off_vals // 8: off_vals // 8 + D * P * n_cad].view(D, P, n_cad)
        self._ws = ws

    def n_solved(self):
        """cadences in runs, summed over all lists (one host synchronisation)"""
        k = self.nrun.long().unsqueeze(-1)
        return int(torch.gather(self.pre_all.long(), -1, k).sum().item())

    def to_dense(self, per_planet=False):
        """(D, N) summed flux, or (D, N, P): rebuilt on the host (tests / plots)"""
        nrun, runs = self.nrun.cpu().numpy(), self.runs.cpu().numpy()
        pre, vals = self.pre_all.cpu().numpy(), self.vals.cpu().numpy()
        import numpy as np

        out = np.zeros((self.n_draw, self.n_cad, self.n_planet))
        for d in range(self.n_draw):
            for p in range(self.n_planet):
                base = 0
                for e in range(self.n_ev):
                    for k in range(nrun[d, p, e]):
                        lo, _, _, hi = runs[d, p, e, k]
                        out[d, lo:hi, p] = vals[d, p, base + pre[d, p, e, k]: base + pre[d, p, e, k] + hi - lo]
                    base += pre[d, p, e, nrun[d, p, e]]
        return out if per_planet else out.sum(-1)


@torch.no_grad()
def transit_flux_sparse(t, params, ld, gflux=None, texp=None, stencil_dt=None, stencil_w=None, flags=0, ttv=None, ws=None):
    """Sparse sweep: no dense flux array is written.  Returns a :class:`SparseFlux`, and with
    ``gflux`` (dense cotangent, read only at the solved cadences) also (gparams, gld, dot[, gshift]).
    Needs sorted times, a scalar (or no) exposure time and no FLAG_EXACT_SCAN; ``ttv = (edges, shift)``:
    timing tables (transits only)."""
    flags = int(flags) | FLAG_SPARSE
    t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(t, texp, stencil_dt, stencil_w, params, ld, flags)
    flags |= _sorted_flag(t)
    edges, shift, n_edge = _ttv_args(ttv, D, P)
    if n_edge and flags & FLAG_SECONDARY:
        raise ValueError("the sparse sweep takes timing tables for transits only")
    N = t.numel()
    lib = _lib.load()
    nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
    if ws is None:
        ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
    elif ws.dtype != torch.float64 or ws.numel() * 8 < nbytes or not ws.is_contiguous() or ws.device != t.device:
        raise ValueError("ws: a contiguous float64 tensor of at least exo_transit_flux_workspace_bytes on the device of t")
    import ctypes

    lay = (ctypes.c_int64 * 5)()
    _lib.check(lib.exo_transit_flux_sparse_layout(N, D, P, lay), "exo_transit_flux_sparse_layout")
    n_ev = 2 if flags & FLAG_SECONDARY else 1
    with torch.cuda.device(t.device):
        if gflux is None:
            if n_edge:
                _lib.check(lib.exo_transit_flux_ttv_fwd_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                            _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                            n_edge, 0, _ptr(ws), nbytes, _stream(t)),
                           "exo_transit_flux_ttv_fwd_f64")
            else:
                _lib.check(lib.exo_transit_flux_fwd_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                        _ptr(ld), D, P, flags, 0, _ptr(ws), nbytes, _stream(t)),
                           "exo_transit_flux_fwd_f64")
            return SparseFlux(ws, list(lay), N, D, P, n_ev)
        gflux = _dev(gflux, "gflux")
        gparams, gld = torch.empty_like(params), torch.empty_like(ld)
        dot = torch.empty(D, dtype=torch.float64, device=t.device)
        if n_edge:
            gshift = torch.empty_like(shift)
            _lib.check(lib.exo_transit_flux_ttv_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                        _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                        n_edge, _ptr(gflux), 0, _ptr(gparams), _ptr(gld), _ptr(gshift),
                                                        _ptr(dot), _ptr(ws), nbytes, _stream(t)),
                       "exo_transit_flux_ttv_vjp_f64")
            return SparseFlux(ws, list(lay), N, D, P, n_ev), gparams, gld, dot, gshift
        _lib.check(lib.exo_transit_flux_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                _ptr(ld), D, P, flags, _ptr(gflux), 0, _ptr(gparams), _ptr(gld), _ptr(dot),
                                                _ptr(ws), nbytes, _stream(t)), "exo_transit_flux_vjp_f64")
    return SparseFlux(ws, list(lay), N, D, P, n_ev), gparams, gld, dot


# ------------------------------------------------------------------------------
# the light curve as a SPARSE model of a GP (round 5): segments of cadences + their values, differentiable
# ------------------------------------------------------------------------------
_SPARSE_MEAN = [os.environ.get("EXO_SPARSE_MEAN", "1") != "0"]     # (0: A/B -- get_light_curve(sparse=True) returns the dense array)


_MULTI_LIST_MEAN = [os.environ.get("EXO_SPARSE_MULTI_LIST", "1") != "0"]    # (0: several lists per draw keep the dense mean -- A/B)


def sparse_mean_supported(n_texp, n_edge, flags, P):
    """can this light curve travel to the celerite kernels as segments + values?  A run-enumeration sweep of the summed flux
    without timing tables or light delay.  One list per draw (one planet, no occultations): the sweep's runs ARE the segments
    (exo_transit_flux_sparse_model); several lists -- planets, occultations -- are merged on the device first
    (exo_sparse_model_merge_f64, round 6)"""
    n_ev = 2 if flags & FLAG_SECONDARY else 1
    return (_SPARSE_MEAN[0] and (P * n_ev == 1 or _MULTI_LIST_MEAN[0]) and n_texp <= 1 and not n_edge
            and not flags & (FLAG_PER_PLANET | FLAG_EXACT_SCAN | FLAG_LIGHT_DELAY | FLAG_CADENCE_MAJOR))


class _TransitFluxSparse(torch.autograd.Function):
    """the EXO_FLAG_SPARSE sweep as a differentiable op: returns (values (D, P * N) -- a view of the value array inside the
    sweep's workspace --, workspace).  The cotangent of `values` comes back in the same layout (the sparse celerite entry
    writes it at the positions of the values) and goes to the reverse sweep as it is -- or, on the Jacobian route, to the
    contraction: no dense (draw, cadence) array in either direction."""

    @staticmethod
    def forward(ctx, t, texp, stencil_dt, stencil_w, params, ld, flags, box, grad_mode):
        import ctypes

        t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(t, texp, stencil_dt, stencil_w, params, ld, flags)
        flags = (flags | FLAG_SPARSE | _sorted_flag(t)) & ~FLAG_CADENCE_MAJOR
        N = t.numel()
        lib = _lib.load()
        nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
        ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
        lay = (ctypes.c_int64 * 5)()
        _lib.check(lib.exo_transit_flux_sparse_layout(N, D, P, lay), "exo_transit_flux_sparse_layout")
        need = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        n_jac = lib.exo_transit_flux_jac_doubles(N, D, P)
        # (grad_mode: torch.is_grad_enabled() of the CALLER -- inside forward() it is always off; under no_grad the rows of
        # derivatives would be written for nothing: ADVICE r4)
        # the same exclusions as _TransitFlux (per-cadence exposure times, per-planet / exact-scan / light-delay sweeps have no
        # Jacobian form: without them the library refuses -- ADVICE r5: fall back to the two-sweep route instead)
        use_jac = (_JAC_ROUTE[0] and need and grad_mode and n_sub >= _JAC_MIN_SUB and n_texp <= 1
                   and not flags & (FLAG_PER_PLANET | FLAG_EXACT_SCAN | FLAG_LIGHT_DELAY)
                   and _jac_fits(8 * n_jac, t.device))
        ctx.jac = None
        with torch.cuda.device(t.device):
            if use_jac:
                _JAC_CALLS[0] += 1
                jac = torch.empty(n_jac, dtype=torch.float64, device=t.device)
                _lib.check(lib.exo_transit_flux_fwd_jac_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                            _ptr(ld), D, P, flags, 0, _ptr(jac), n_jac, _ptr(ws), nbytes,
                                                            _stream(t)), "exo_transit_flux_fwd_jac_f64")
                ctx.jac = jac
            else:
                _lib.check(lib.exo_transit_flux_fwd_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                        _ptr(ld), D, P, flags, 0, _ptr(ws), nbytes, _stream(t)),
                           "exo_transit_flux_fwd_f64")
        vals = ws[lay[3] // 8: lay[3] // 8 + D * P * N].view(D, P * N)
        ctx.save_for_backward(t, texp, sdt, sw, params, ld)
        ctx.meta = (n_texp, n_sub, D, P, flags, nbytes)
        ctx.ws = ws
        box.append(ws)      # (the workspace is not an autograd output -- `vals` is a view of it: the caller's box keeps it)
        return vals

    @staticmethod
    def backward(ctx, gvals):
        t, texp, sdt, sw, params, ld = ctx.saved_tensors
        n_texp, n_sub, D, P, flags, nbytes = ctx.meta
        N = t.numel()
        if gvals is None:
            return (None,) * 9
        if not gvals.is_contiguous() or tuple(gvals.shape) != (D, P * N):
            raise ValueError("the cotangent of a sparse light curve's values must be a contiguous (n_draw, n_planet * n_cad) array")
        gparams, gld = torch.empty_like(params), torch.empty_like(ld)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            if ctx.jac is not None:
                _lib.check(lib.exo_transit_flux_jac_vjp_f64(_ptr(gvals), N, D, P, flags, _ptr(ctx.jac), _ptr(ctx.ws), nbytes,
                                                            _ptr(gparams), _ptr(gld), 0, _stream(t)), "exo_transit_flux_jac_vjp_f64")
            else:
                _lib.check(lib.exo_transit_flux_vjp_sparse_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                               _ptr(params), _ptr(ld), D, P, flags, _ptr(gvals), _ptr(gparams),
                                                               _ptr(gld), 0, _ptr(ctx.ws), nbytes, 1, _stream(t)),
                           "exo_transit_flux_vjp_sparse_f64")
        return None, None, None, None, gparams, gld, None, None, None


class _SparseToDense(torch.autograd.Function):
    """(D, N) dense summed flux of a sparse light curve, differentiable (the fallback for uses the sparse celerite entry does
    not cover): scatter forward, gather of the cotangent at the solved cadences backward -- both on the host-side indices of
    the runs, a handful of torch kernels (not a hot path)"""

    @staticmethod
    def forward(ctx, vals, sp):
        idx_val, idx_cad, idx_draw = sp._indices()
        out = torch.zeros(sp.n_draw, sp.n_cad, dtype=torch.float64, device=vals.device)
        out.index_put_((idx_draw, idx_cad), vals.reshape(-1)[idx_val], accumulate=True)
        ctx.sp = sp
        return out

    @staticmethod
    def backward(ctx, gout):
        sp = ctx.sp
        idx_val, idx_cad, idx_draw = sp._indices()
        g = torch.zeros(sp.n_draw, sp.n_planet * sp.n_cad, dtype=torch.float64, device=gout.device)
        g.view(-1)[idx_val] = gout[idx_draw, idx_cad]
        return g, None


class SparseLightCurve:
    """What ``get_light_curve(total=True, sparse=True)`` returns for a batch of draws: the summed light curve as the runs
    of cadences in which a planet can overlap the disk and the flux of exactly those cadences (every other cadence: 0),
    differentiable through ``values``.  It is the ``mean`` of a ``GaussianProcess``: the celerite kernels read the
    segments and write the mean's cotangent back at the values, and the (draws, cadences) array -- 97 % zeros -- never
    exists (C3: 3.8 -> ~3.1 ms per value + gradient).  ``dense()`` gives the ordinary (draws, cadences) tensor."""

    def __init__(self, values, ws, n_cad, n_draw, n_planet, flags):
        self.values, self._ws = values, ws
        self.n_cad, self.n_draw, self.n_planet, self.flags = n_cad, n_draw, n_planet, flags
        self.shape = (n_draw, n_cad)

    def layout(self):
        return _sparse_from_ws(self._ws, self.n_cad, self.n_draw, self.n_planet, self.flags)

    def model_struct(self):
        """exo_sparse_model for the celerite entries (a host struct of device pointers: keep ``self`` alive while it is used)"""
        m = _lib.SparseModel()
        import ctypes

        n_ev = 2 if self.flags & FLAG_SECONDARY else 1
        if self.n_planet * n_ev != 1:
            raise ValueError("a sparse GaussianProcess mean is one list of segments per draw; this light curve has "
                             f"{self.n_planet} planet(s) x {n_ev} event(s): use .merged() (or .dense())")

        nbytes = _lib.load().exo_transit_flux_workspace_bytes(self.n_cad, self.n_draw, self.n_planet)
        _lib.check(_lib.load().exo_transit_flux_sparse_model(_ptr(self._ws), nbytes, self.n_cad, self.n_draw, self.n_planet,
                                                             self.flags & FLAG_SECONDARY, ctypes.addressof(m)),
                   "exo_transit_flux_sparse_model")
        return m

    def _indices(self):
        """(position in values.view(-1), cadence, draw) of every solved cadence -- built with torch ops from the runs"""
        lay = self.layout()
        D, P, N = self.n_draw, self.n_planet, self.n_cad
        n_ev = lay.n_ev
        nrun = lay.nrun.long()                                   # (D, P, n_ev)
        K = int(nrun.max().item()) if nrun.numel() else 0
        dev = self.values.device
        if K == 0:
            z = torch.zeros(0, dtype=torch.long, device=dev)
            return z, z, z
        runs = lay.runs[..., :K, :].long()                       # (D, P, n_ev, K, 4)
        lo, hi = runs[..., 0], runs[..., 3]
        live = torch.arange(K, device=dev) < nrun.unsqueeze(-1)
        ln = torch.where(live, hi - lo, torch.zeros_like(lo))    # (D, P, n_ev, K)
        pre = lay.pre_all[..., :K].long()
        tot = torch.gather(lay.pre_all.long(), -1, nrun.unsqueeze(-1)).squeeze(-1)          # (D, P, n_ev)
        evbase = torch.cumsum(tot, -1) - tot                      # occultations behind the transits
        d_i = torch.arange(D, device=dev).view(D, 1, 1, 1)
        p_i = torch.arange(P, device=dev).view(1, P, 1, 1)
        vbase = (d_i * P + p_i) * N + evbase.unsqueeze(-1) + pre  # first value of every run
        flat_len = ln.reshape(-1)
        total = int(flat_len.sum().item())
        rid = torch.repeat_interleave(torch.arange(flat_len.numel(), device=dev), flat_len, output_size=total)
        start = torch.cumsum(flat_len, 0) - flat_len
        within = torch.arange(total, device=dev) - start[rid]
        idx_val = vbase.reshape(-1)[rid] + within
        idx_cad = lo.reshape(-1)[rid] + within
        idx_draw = d_i.expand_as(ln).reshape(-1)[rid]
        return idx_val, idx_cad, idx_draw

    def dense(self):
        """the (draws, cadences) summed flux, differentiable"""
        return _SparseToDense.apply(self.values, self)

    def merged(self):
        """this light curve with ONE list of segments per draw (what the celerite kernels take): itself when it already is
        (one planet, no occultations), otherwise the lists merged on the device -- union of the runs, values summed over the
        planets / events per cadence -- as a :class:`MergedSparseLightCurve`, differentiable through ``values``"""
        n_ev = 2 if self.flags & FLAG_SECONDARY else 1
        if self.n_planet * n_ev == 1:
            return self
        box = []
        mvals = _MergeSparse.apply(self.values, self, box)
        return MergedSparseLightCurve(mvals, box[0], self.n_cad, self.n_draw, self.n_planet, self.flags)

    def detach(self):
        return SparseLightCurve(self.values.detach(), self._ws, self.n_cad, self.n_draw, self.n_planet, self.flags)

    # Arithmetic: a sparse light curve in an expression -- ``offset + lc``, ``lc * depth`` -- becomes the ordinary
    # (draws, cadences) tensor (``dense()``: differentiable), so code written for the dense return type keeps working whatever
    # get_light_curve(sparse=True) could return for the model; only a GaussianProcess mean stays sparse.
    def __add__(self, other): return self.dense() + other
    def __radd__(self, other): return other + self.dense()
    def __sub__(self, other): return self.dense() - other
    def __rsub__(self, other): return other - self.dense()
    def __mul__(self, other): return self.dense() * other
    def __rmul__(self, other): return other * self.dense()
    def __truediv__(self, other): return self.dense() / other
    def __neg__(self): return -self.dense()


_POISON = [None]    # tests: fill the merge workspace and its cotangent buffer with this value first (nothing may depend on what they
                    # held before the call -- under hipGraph replay that is the previous step's: gp/celerite.py has the same hook)


def _scratch(n, device):
    x = torch.empty(n, dtype=torch.float64, device=device)
    if _POISON[0] is not None:
        x.fill_(_POISON[0])
    return x


class _MergeSparse(torch.autograd.Function):
    """values of a several-list sparse light curve (D, P * N: the sweep's layout) -> values of the merged model (D, N), of which
    row d's first off[d][nseg[d]] entries are defined (exo_sparse_model_merge_f64); backward: the cotangent of the merged values
    back at the lists' value positions (exo_sparse_model_merge_vjp_f64).  `box` receives the merge workspace."""

    @staticmethod
    def forward(ctx, vals, sp, box):
        lib = _lib.load()
        N, D, P = sp.n_cad, sp.n_draw, sp.n_planet
        nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
        mbytes = lib.exo_sparse_merge_workspace_bytes(N, D, P)
        mws = _scratch(max(mbytes // 8 + 1, 1), vals.device)
        with torch.cuda.device(vals.device):
            _lib.check(lib.exo_sparse_model_merge_f64(_ptr(sp._ws), nbytes, N, D, P, sp.flags & FLAG_SECONDARY, _ptr(mws), mbytes,
                                                      None, _stream(vals)), "exo_sparse_model_merge_f64")
        import ctypes

        lay = (ctypes.c_int64 * 5)()
        _lib.check(lib.exo_sparse_merge_layout(N, D, P, lay), "exo_sparse_merge_layout")
        ctx.sp, ctx.mws, ctx.sizes = sp, mws, (nbytes, mbytes)
        box.append(mws)
        return mws[lay[3] // 8: lay[3] // 8 + D * N].view(D, N)

    @staticmethod
    def backward(ctx, gm):
        sp = ctx.sp
        N, D, P = sp.n_cad, sp.n_draw, sp.n_planet
        nbytes, mbytes = ctx.sizes
        if not gm.is_contiguous() or tuple(gm.shape) != (D, N):
            raise ValueError("the cotangent of a merged sparse light curve's values must be a contiguous (n_draw, n_cad) array")
        # (only the positions the lists cover are defined, as in `vals` itself -- gp/celerite.py, _CeleriteLogLikeSparse.backward)
        gvals = _scratch(D * P * N, gm.device).view(D, P * N)
        with torch.cuda.device(gm.device):
            _lib.check(_lib.load().exo_sparse_model_merge_vjp_f64(_ptr(sp._ws), nbytes, N, D, P, sp.flags & FLAG_SECONDARY,
                                                                  _ptr(ctx.mws), mbytes, _ptr(gm), _ptr(gvals), _stream(gm)),
                       "exo_sparse_model_merge_vjp_f64")
        return gvals, None, None


class MergedSparseLightCurve(SparseLightCurve):
    """A sparse light curve of several lists per draw (planets, occultations) merged into one list of disjoint segments per
    draw, a cadence's value the sum over the lists: the ``mean`` of a ``GaussianProcess`` for multi-planet and
    secondary-eclipse models (round 6).  ``values`` is (draws, cadences) with the first ``off[nseg]`` entries of a row defined."""

    def merged(self):
        return self

    def _merge_layout(self):
        import ctypes

        lay = (ctypes.c_int64 * 5)()
        _lib.check(_lib.load().exo_sparse_merge_layout(self.n_cad, self.n_draw, self.n_planet, lay), "exo_sparse_merge_layout")
        return list(lay)

    def model_struct(self):
        import ctypes

        m = _lib.SparseModel()
        nbytes = _lib.load().exo_sparse_merge_workspace_bytes(self.n_cad, self.n_draw, self.n_planet)
        _lib.check(_lib.load().exo_sparse_model_merged(_ptr(self._ws), nbytes, self.n_cad, self.n_draw, self.n_planet,
                                                       ctypes.addressof(m)), "exo_sparse_model_merged")
        return m

    def segments(self):
        """(nseg (D,), seg (D, cap, 2), off (D, cap + 1)) int32 views of the merge workspace"""
        o_nseg, o_seg, o_off, _, cap = self._merge_layout()
        raw = self._ws.view(torch.int32)
        D = self.n_draw
        nseg = raw[o_nseg // 4: o_nseg // 4 + D]
        seg = raw[o_seg // 4: o_seg // 4 + D * cap * 2].view(D, cap, 2)
        off = raw[o_off // 4: o_off // 4 + D * (cap + 1)].view(D, cap + 1)
        return nseg, seg, off

    def layout(self):
        raise NotImplementedError("a merged light curve has segments(), not the sweep's per-list runs")

    def _indices(self):
        nseg, seg, off = self.segments()
        D, N = self.n_draw, self.n_cad
        dev = self.values.device
        K = int(nseg.max().item()) if nseg.numel() else 0
        if K == 0:
            z = torch.zeros(0, dtype=torch.long, device=dev)
            return z, z, z
        lo, hi = seg[:, :K, 0].long(), seg[:, :K, 1].long()
        live = torch.arange(K, device=dev) < nseg.long().unsqueeze(-1)
        ln = torch.where(live, hi - lo, torch.zeros_like(lo))
        flat_len = ln.reshape(-1)
        total = int(flat_len.sum().item())
        rid = torch.repeat_interleave(torch.arange(flat_len.numel(), device=dev), flat_len, output_size=total)
        start = torch.cumsum(flat_len, 0) - flat_len
        within = torch.arange(total, device=dev) - start[rid]
        d_i = torch.arange(D, device=dev).view(D, 1).expand(D, K).reshape(-1)[rid]
        idx_val = d_i * N + off[:, :K].long().reshape(-1)[rid] + within
        idx_cad = lo.reshape(-1)[rid] + within
        return idx_val, idx_cad, d_i

    def dense(self):
        return _MergedToDense.apply(self.values, self)

    def detach(self):
        return MergedSparseLightCurve(self.values.detach(), self._ws, self.n_cad, self.n_draw, self.n_planet, self.flags)


class _MergedToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vals, sp):
        idx_val, idx_cad, idx_draw = sp._indices()
        out = torch.zeros(sp.n_draw, sp.n_cad, dtype=torch.float64, device=vals.device)
        out[idx_draw, idx_cad] = vals.reshape(-1)[idx_val]
        ctx.sp = sp
        return out

    @staticmethod
    def backward(ctx, gout):
        sp = ctx.sp
        idx_val, idx_cad, idx_draw = sp._indices()
        g = torch.zeros(sp.n_draw, sp.n_cad, dtype=torch.float64, device=gout.device)
        g.view(-1)[idx_val] = gout[idx_draw, idx_cad]
        return g, None


def transit_flux_sparse_model(t, params, ld, texp=None, stencil_dt=None, stencil_w=None, flags=0):
    """the summed light curve of ``n_draw`` parameter sets as a :class:`SparseLightCurve` with one list of segments per draw
    (differentiable w.r.t. ``params`` and ``ld``): the sweep's own runs for one planet without occultations, the merged form
    (:class:`MergedSparseLightCurve`) otherwise; see :func:`sparse_mean_supported` for what qualifies"""
    box = []
    vals = _TransitFluxSparse.apply(t, texp, stencil_dt, stencil_w, params, ld, int(flags), box, torch.is_grad_enabled())
    return SparseLightCurve(vals, box[0], t.numel(), params.shape[0], params.shape[1], int(flags) | FLAG_SPARSE).merged()


class KeptDenseFlux:
    """A dense (D, N) flux array KEPT ACROSS STEPS (include/exoplanet_amd.h, exo_transit_sparse_scatter_f64).

    The dense sweep zero-fills every cadence of every draw on every call (1.2 GB at 1024 draws x 150 000 cadences) although a
    step of a sampler solves the same few per cent of them as the step before.  This object owns the flux array and the
    sweep's workspace: ``step`` zeroes the cadences the LAST step solved, runs the sparse sweep, and writes the cadences THIS
    step solved -- after every step ``flux`` holds what ``transit_flux`` / ``transit_flux_vjp`` would have returned, bit for
    bit, for the price of the sparse sweep.  The state is this object's, not the library's; ``flux`` must not be written by
    anyone else.  Total flux, run-enumeration sweeps (sorted times, a scalar or no exposure time), no timing tables."""

    def __init__(self, t, n_draw, n_planet, texp=None, stencil_dt=None, stencil_w=None, flags=0):
        if int(flags) & (FLAG_PER_PLANET | FLAG_CADENCE_MAJOR | FLAG_EXACT_SCAN | FLAG_SPARSE):
            raise ValueError("KeptDenseFlux: the summed flux in rows, run-enumeration sweeps")
        self.t = _dev(t, "t")
        self.texp, self.stencil = texp, (stencil_dt, stencil_w)
        self.flags = int(flags)
        self.D, self.P, self.N = int(n_draw), int(n_planet), self.t.numel()
        self._nbytes = _lib.load().exo_transit_flux_workspace_bytes(self.N, self.D, self.P)
        # zeroed: a workspace without runs (the first step has nothing to clear), a flux array without a transit
        self._ws = torch.zeros(max(self._nbytes // 8 + 1, 1), dtype=torch.float64, device=self.t.device)
        self.flux = torch.zeros(self.D, self.N, dtype=torch.float64, device=self.t.device)

    def _scatter(self, clear):
        with torch.cuda.device(self.t.device):
            _lib.check(_lib.load().exo_transit_sparse_scatter_f64(_ptr(self._ws), self._nbytes, self.N, self.D, self.P,
                                                                  self.flags & FLAG_SECONDARY, 1 if clear else 0, _ptr(self.flux),
                                                                  _stream(self.t)), "exo_transit_sparse_scatter_f64")

    @torch.no_grad()
    def step(self, params, ld, gflux=None):
        """-> flux (D, N) [, gparams, gld, dot with a cotangent ``gflux`` (D, N)]; ``flux`` is this object's array"""
        if params.shape[0] != self.D or params.shape[1] != self.P:
            raise ValueError("params: (n_draw, n_planet, NPAR) as this object was made for")
        self._scatter(True)
        out = transit_flux_sparse(self.t, params, ld, gflux, self.texp, self.stencil[0], self.stencil[1], self.flags, ws=self._ws)
        self._scatter(False)
        return self.flux if gflux is None else (self.flux,) + tuple(out[1:])


# ------------------------------------------------------------------------------
# timing tables of a TTVOrbit given O-C offsets
# ------------------------------------------------------------------------------
class _TtvTables(torch.autograd.Function):
    """(edges, shift) of exo_ttv_tables_f64 from period (D|1, P), t0 (D|1, P) and one (D|1, n_p) tensor of offsets per
    planet; differentiable in the offsets and the period (the edges carry no gradient, t0 none through the tables)"""

    @staticmethod
    def forward(ctx, period, t0, n_draw, *ttvs):
        import ctypes

        period, t0 = _dev(period.detach(), "period"), _dev(t0.detach(), "t0")
        ttvs = [_dev(x.detach(), "ttvs") for x in ttvs]
        P, D = len(ttvs), int(n_draw)
        counts = [int(x.shape[-1]) for x in ttvs]
        n_edge = max(counts) + 1
        edges = torch.empty(D, P, n_edge, dtype=torch.float64, device=period.device)
        shift = torch.empty(D, P, n_edge + 1, dtype=torch.float64, device=period.device)
        args = _TtvTables._args(period, t0, ttvs, counts)
        lib = _lib.load()
        with torch.cuda.device(period.device):
            _lib.check(lib.exo_ttv_tables_f64(*args, D, P, n_edge, _ptr(edges), _ptr(shift), _stream(period)), "exo_ttv_tables_f64")
        ctx.save_for_backward(period, t0, *ttvs)
        ctx.meta = (D, P, n_edge, counts)
        ctx.mark_non_differentiable(edges)
        return edges, shift

    @staticmethod
    def _args(period, t0, ttvs, counts):
        import ctypes

        def strides(x):      # (rows, P) -> element strides of (draw, planet); one row: broadcast over draws
            return (x.stride(0) if x.shape[0] > 1 else 0, x.stride(1) if x.shape[1] > 1 else 0)

        P = len(ttvs)
        tp = (ctypes.c_void_p * P)(*[x.data_ptr() for x in ttvs])
        td = (ctypes.c_int64 * P)(*[(x.stride(0) if x.shape[0] > 1 else 0) for x in ttvs])
        tn = (ctypes.c_int32 * P)(*counts)
        return (_ptr(period), *strides(period), _ptr(t0), *strides(t0), tp, td, tn)

    @staticmethod
    def backward(ctx, _gedges, gshift):
        import ctypes

        period, t0, *ttvs = ctx.saved_tensors
        D, P, n_edge, counts = ctx.meta
        if gshift is None:
            return (None,) * (3 + P)
        gshift = _dev(gshift, "gshift")
        need_p = ctx.needs_input_grad[0]
        gper = torch.empty(D, P, dtype=torch.float64, device=gshift.device) if need_p else None
        gt = [torch.empty(D, n, dtype=torch.float64, device=gshift.device) if ctx.needs_input_grad[3 + p] else None
              for p, n in enumerate(counts)]
        gp = (ctypes.c_void_p * P)(*[_ptr(x) for x in gt])
        lib = _lib.load()
        with torch.cuda.device(gshift.device):
            _lib.check(lib.exo_ttv_tables_vjp_f64(*_TtvTables._args(period, t0, ttvs, counts), D, P, n_edge, _ptr(gshift), gp,
                                                  _ptr(gper), _stream(gshift)), "exo_ttv_tables_vjp_f64")
        if gper is not None:
            gper = gper.sum_to_size(period.shape)
        gt = [None if g is None else g.sum_to_size(x.shape) for g, x in zip(gt, ttvs)]
        return (gper, None, None, *gt)


def ttv_tables(period, t0, ttvs, n_draw):
    """Timing tables ``(edges (D, P, E), shift (D, P, E + 1))`` of a TTVOrbit whose transits are all labelled and given as
    offsets from the linear ephemeris: ``period``, ``t0`` (1 | D, P), ``ttvs`` a list of (1 | D, n_p) tensors.  One
    launch each way (exo_ttv_tables_f64); differentiable in ``ttvs`` and ``period``."""
    return _TtvTables.apply(period, t0, int(n_draw), *ttvs)


# ------------------------------------------------------------------------------
# radial velocity
# ------------------------------------------------------------------------------
RV_NPAR = 6
OV_NPAR = 10          # include/exoplanet_amd.h EXO_OV_*
OV_VELOCITY = 1
OV_ACCELERATION = 2
RV_N, RV_TP, RV_ECC, RV_COSW, RV_SINW, RV_AMP = range(6)


class _RadialVelocity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, params):
        t = _dev(t, "t")
        params = _dev(params, "params")
        if t.dim() != 1:
            raise ValueError("t must be 1-D (n_cad,)")
        if params.dim() != 3 or params.shape[-1] != RV_NPAR:
            raise ValueError(f"params must be (n_draw, n_planet, {RV_NPAR})")
        D, P, _ = params.shape
        rv = torch.empty(D, t.numel(), P, dtype=torch.float64, device=t.device)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_radial_velocity_fwd_f64(_ptr(t), t.numel(), _ptr(params), D, P, _ptr(rv), _stream(t)),
                       "exo_radial_velocity_fwd_f64")
        ctx.save_for_backward(t, params)
        return rv

    @staticmethod
    def backward(ctx, grv):
        t, params = ctx.saved_tensors
        D, P, _ = params.shape
        grv = _dev(grv, "grv")
        gparams = torch.empty_like(params)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_radial_velocity_vjp_f64(_ptr(t), t.numel(), _ptr(params), D, P, _ptr(grv),
                                                       _ptr(gparams), _stream(t)), "exo_radial_velocity_vjp_f64")
        return None, gparams


class _OrbitVector(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, params, flags):
        t = _dev(t, "t")
        params = _dev(params, "params")
        if t.dim() != 1:
            raise ValueError("t must be 1-D (n_cad,)")
        if params.dim() != 3 or params.shape[-1] != OV_NPAR:
            raise ValueError(f"params must be (n_draw, n_planet, {OV_NPAR})")
        D, P, _ = params.shape
        out = torch.empty(D, t.numel(), P, 3, dtype=torch.float64, device=t.device)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_orbit_vector_fwd_f64(_ptr(t), t.numel(), _ptr(params), D, P, flags, _ptr(out), _stream(t)),
                       "exo_orbit_vector_fwd_f64")
        ctx.save_for_backward(t, params)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, gout):
        t, params = ctx.saved_tensors
        D, P, _ = params.shape
        gout = _dev(gout, "gout")
        gparams = torch.empty_like(params)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_orbit_vector_vjp_f64(_ptr(t), t.numel(), _ptr(params), D, P, ctx.flags, _ptr(gout),
                                                    _ptr(gparams), _stream(t)), "exo_orbit_vector_vjp_f64")
        return None, gparams, None


def orbit_vector(t, params, velocity=False, acceleration=False):
    """Position (or, ``velocity=True`` / ``acceleration=True``, velocity / acceleration) vectors in the observer frame for ``n_draw`` parameter sets, one
    fused launch each way: t (n_cad,), params (n_draw, n_planet, 10) with slots EXO_OV_* (n, t_periastron, ecc,
    cos / sin omega, cos / sin incl, amplitude, cos / sin Omega) -> (n_draw, n_cad, n_planet, 3) = (X, Y, Z).
    Differentiable with respect to ``params``.  (keplerian.py:380-409, :572-578, :283-322)"""
    return _OrbitVector.apply(t, params, OV_ACCELERATION if acceleration else (OV_VELOCITY if velocity else 0))


def radial_velocity(t, params):
    """Stellar reflex radial velocity, one column per planet: t (n_cad,), params
    (n_draw, n_planet, 6) = (n, t_periastron, ecc, cos omega, sin omega, amplitude)
    (include/exoplanet_amd.h EXO_RV_*) -> (n_draw, n_cad, n_planet); differentiable
    w.r.t. ``params``.  One launch each way (keplerian.py:633-677)."""
    return _RadialVelocity.apply(t.detach(), params)


# ------------------------------------------------------------------------------
# record packing: KeplerianOrbit.__init__ algebra + get_cl + windows as one kernel
# ------------------------------------------------------------------------------
class _PackRecords(torch.autograd.Function):
    @staticmethod
    def forward(ctx, orbit_in, ld_in, flags):
        orbit_in = _dev(orbit_in, "orbit_in")
        ld_in = _dev(ld_in, "ld_in")
        if orbit_in.dim() != 3 or orbit_in.shape[-1] != NIN:
            raise ValueError(f"orbit_in must be (n_draw, n_planet, {NIN})")
        D, P, _ = orbit_in.shape
        nset = 2 if flags & FLAG_SECONDARY else 1
        if ld_in.shape != (D, 2 * nset):
            raise ValueError(f"ld_in must be (n_draw, {2 * nset})")
        params = torch.empty(D, P, NPAR, dtype=torch.float64, device=orbit_in.device)
        ld = torch.empty(D, 3 * nset, dtype=torch.float64, device=orbit_in.device)
        lib = _lib.load()
        with torch.cuda.device(orbit_in.device):
            _lib.check(lib.exo_pack_records_f64(_ptr(orbit_in), _ptr(ld_in), D, P, flags, _ptr(params), _ptr(ld),
                                                _stream(orbit_in)), "exo_pack_records_f64")
        ctx.save_for_backward(orbit_in, ld_in)
        ctx.flags = flags
        return params, ld

    @staticmethod
    def backward(ctx, gparams, gld):
        orbit_in, ld_in = ctx.saved_tensors
        D, P, _ = orbit_in.shape
        gparams = _dev(gparams, "gparams")
        gld = _dev(gld, "gld")
        go = torch.empty_like(orbit_in)
        gl = torch.empty_like(ld_in)
        lib = _lib.load()
        with torch.cuda.device(orbit_in.device):
            _lib.check(lib.exo_pack_records_vjp_f64(_ptr(orbit_in), _ptr(ld_in), D, P, ctx.flags, _ptr(gparams),
                                                    _ptr(gld), _ptr(go), _ptr(gl), _stream(orbit_in)),
                       "exo_pack_records_vjp_f64")
        return go, gl, None


SHO_SIGMA, SHO_RHO, SHO_TAU = 1, 2, 4   # include/exoplanet_amd.h EXO_SHO_*


class _ShoCoefficients(torch.autograd.Function):
    @staticmethod
    def forward(ctx, amp, freq, damp, flags, eps):
        amp, freq, damp = (_dev(x, "SHO parameter").contiguous() for x in (amp, freq, damp))
        if not (amp.shape == freq.shape == damp.shape) or amp.dim() != 1:
            raise ValueError("amp, freq, damp must be 1-D tensors of one length")
        n = amp.numel()
        coef = torch.empty(n, 4, dtype=torch.float64, device=amp.device)
        kind = torch.empty(n, dtype=torch.int32, device=amp.device)
        lib = _lib.load()
        with torch.cuda.device(amp.device):
            _lib.check(lib.exo_sho_coefficients_f64(_ptr(amp), _ptr(freq), _ptr(damp), flags, eps, n, _ptr(coef), _ptr(kind),
                                                    _stream(amp)), "exo_sho_coefficients_f64")
        ctx.save_for_backward(amp, freq, damp)
        ctx.flags, ctx.eps = flags, eps
        ctx.mark_non_differentiable(kind)
        ctx.set_materialize_grads(False)      # (no zero-filled int32 "gradient" of `kind` per backward call: a launch each)
        return coef, kind

    @staticmethod
    def backward(ctx, gcoef, _gkind):
        amp, freq, damp = ctx.saved_tensors
        n = amp.numel()
        if gcoef is None:
            return None, None, None, None, None
        gcoef = _dev(gcoef, "gcoef").contiguous()
        ga, gf, gd = (torch.empty_like(amp) for _ in range(3))
        lib = _lib.load()
        with torch.cuda.device(amp.device):
            _lib.check(lib.exo_sho_coefficients_vjp_f64(_ptr(amp), _ptr(freq), _ptr(damp), ctx.flags, ctx.eps, n, _ptr(gcoef),
                                                        _ptr(ga), _ptr(gf), _ptr(gd), _stream(amp)),
                       "exo_sho_coefficients_vjp_f64")
        return ga, gf, gd, None, None


def sho_coefficients(amp, freq, damp, flags=0, eps=1e-5):
    """celerite2's SHOTerm in any of its parameterisations (``flags``: SHO_SIGMA | SHO_RHO | SHO_TAU say that ``amp``
    is sigma rather than S0, ``freq`` the undamped period rho rather than w0, ``damp`` tau rather than Q) -> the
    term's pair slot ``coef`` (n, 4) and ``kind`` (n,) int32 (1: two real terms, Q < 1/2), one fused launch each way."""
    return _ShoCoefficients.apply(amp, freq, damp, int(flags), float(eps))


SHO_MAX_TERMS = 8


class _ShoCoefficientsMulti(torch.autograd.Function):
    """several SHO terms in one launch each way (include/exoplanet_amd.h, exo_sho_coefficients_multi_f64): inputs
    (flags tuple, eps, amp_0, freq_0, damp_0, amp_1, ...) -> coef (n, T, 4), kind (n, T)"""

    @staticmethod
    def forward(ctx, flags, eps, *cols):
        import ctypes

        T = len(flags)
        cols = [_dev(x, "SHO parameter").contiguous() for x in cols]
        n = cols[0].numel()
        if len(cols) != 3 * T or not 1 <= T <= SHO_MAX_TERMS or any(c.dim() != 1 or c.numel() != n for c in cols):
            raise ValueError("three 1-D tensors of one length per term, at most %d terms" % SHO_MAX_TERMS)
        coef = torch.empty(n, T, 4, dtype=torch.float64, device=cols[0].device)
        kind = torch.empty(n, T, dtype=torch.int32, device=cols[0].device)
        vp = ctypes.c_void_p
        arr = lambda q: (vp * T)(*[cols[3 * k + q].data_ptr() for k in range(T)])  # noqa: E731
        fl = (ctypes.c_uint32 * T)(*[int(f) for f in flags])
        with torch.cuda.device(coef.device):
            _lib.check(_lib.load().exo_sho_coefficients_multi_f64(arr(0), arr(1), arr(2), fl, T, eps, n, _ptr(coef), _ptr(kind),
                                                                  _stream(coef)), "exo_sho_coefficients_multi_f64")
        ctx.save_for_backward(*cols)
        ctx.flags, ctx.eps = tuple(int(f) for f in flags), eps
        ctx.mark_non_differentiable(kind)
        ctx.set_materialize_grads(False)
        return coef, kind

    @staticmethod
    def backward(ctx, gcoef, _gkind):
        import ctypes

        cols = ctx.saved_tensors
        T = len(ctx.flags)
        if gcoef is None:
            return (None,) * (2 + 3 * T)
        n = cols[0].numel()
        gcoef = _dev(gcoef, "gcoef").contiguous()
        grads = [torch.empty_like(c) for c in cols]
        vp = ctypes.c_void_p
        arr = lambda xs, q: (vp * T)(*[xs[3 * k + q].data_ptr() for k in range(T)])  # noqa: E731
        fl = (ctypes.c_uint32 * T)(*ctx.flags)
        with torch.cuda.device(gcoef.device):
            _lib.check(_lib.load().exo_sho_coefficients_multi_vjp_f64(arr(cols, 0), arr(cols, 1), arr(cols, 2), fl, T, ctx.eps, n,
                                                                      _ptr(gcoef), arr(grads, 0), arr(grads, 1), arr(grads, 2),
                                                                      _stream(gcoef)), "exo_sho_coefficients_multi_vjp_f64")
        return (None, None) + tuple(grads)


def sho_coefficients_multi(terms, eps=1e-5):
    """``terms``: [(amp, freq, damp, flags), ...] of one length n each -> coef (n, T, 4), kind (n, T) int32: a sum of SHO terms
    in one launch each way (and no concatenation of the terms' slots)"""
    flat = [x for amp, freq, damp, _ in terms for x in (amp, freq, damp)]
    return _ShoCoefficientsMulti.apply(tuple(int(f) for *_, f in terms), float(eps), *flat)


def _cols_vjp(t, texp, stencil_dt, stencil_w, host, D, P, nset, flags, pack_flags, gflux, events=(None, None), fold=None):
    """exo_transit_flux_cols_vjp_f64: the constructor's columns (``host`` = the six host arrays of _cols_host) -> records, the
    value + VJP sweep and, ``fold = (gscale | None, gcols, gld_cols)`` (host arrays of output pointers), the packing VJP in
    the same call.  Returns (flux | SparseFlux, gparams, gld, dot)."""
    cp, ds, ps, df, lp, ls = host
    t = _dev(t, "t")
    N = t.numel()
    if t.dim() != 1:
        raise ValueError("t must be 1-D (n_cad,)")
    if texp is None:
        n_texp, n_sub, sdt, sw = 0, 1, None, None
    else:
        texp = _dev(texp, "texp").reshape(-1)
        sdt, sw = _dev(stencil_dt, "stencil_dt"), _dev(stencil_w, "stencil_w")
        n_texp, n_sub = texp.numel(), sdt.numel()
        if n_texp not in (1, N) or sw.numel() != n_sub or not 1 <= n_sub <= MAX_SUBEXP:
            raise ValueError("texp: a scalar or one entry per cadence; stencil: 1..%d points" % MAX_SUBEXP)
    if not 1 <= P <= MAX_PLANETS:
        raise ValueError(f"1 <= n_planet <= {MAX_PLANETS}")
    flags = int(flags) | _sorted_flag(t)
    shape = (D, N, P) if flags & FLAG_PER_PLANET else (D, N)
    if isinstance(gflux, torch.Tensor) and tuple(gflux.shape) != shape:
        raise ValueError(f"gflux must have shape {shape}")
    cm_sweep = n_texp <= 1 and not flags & (FLAG_EXACT_SCAN | FLAG_PER_PLANET)
    if is_cadence_major(gflux) and cm_sweep:
        flags |= FLAG_CADENCE_MAJOR
    elif flags & FLAG_CADENCE_MAJOR and cm_sweep:
        gflux = _dev(gflux, "gflux").t().contiguous().t()
    else:
        flags &= ~FLAG_CADENCE_MAJOR
        gflux = _dev(gflux, "gflux")
    dev = t.device
    lib = _lib.load()
    nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
    ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=dev)
    params = torch.empty(D, P, NPAR, dtype=torch.float64, device=dev)
    ld = torch.empty(D, 3 * nset, dtype=torch.float64, device=dev)
    gparams, gld = torch.empty_like(params), torch.empty_like(ld)
    dot = torch.empty(D, dtype=torch.float64, device=dev)
    flux = None
    if not flags & FLAG_SPARSE:
        flux = (torch.empty((N, D), dtype=torch.float64, device=dev).t() if flags & FLAG_CADENCE_MAJOR
                else torch.empty(shape, dtype=torch.float64, device=dev))
    gscale, gcp, glp = (None, None, None) if fold is None else fold
    with torch.cuda.device(dev):
        _lib.check(lib.exo_transit_flux_cols_vjp_f64(cp, ds, ps, df, lp, ls, int(pack_flags), _ptr(t), N, _ptr(texp), n_texp,
                                                     _ptr(sdt), _ptr(sw), n_sub, D, P, flags, _ptr(gflux), _ptr(flux),
                                                     _ptr(params), _ptr(ld), _ptr(gparams), _ptr(gld), _ptr(dot),
                                                     0 if fold is None else 1, _ptr(gscale), gcp, glp, _ptr(ws), nbytes,
                                                     _stream(t), events[0], events[1]), "exo_transit_flux_cols_vjp_f64")
    if flags & FLAG_SPARSE:
        flux = _sparse_from_ws(ws, N, D, P, flags)
    return flux, gparams, gld, dot


def _cols_host(ocols, lcols, D, P):
    """the host arrays exo_pack_records_cols_f64 / exo_transit_flux_cols_vjp_f64 take for a list of NIN orbit columns (None = the
    constructor's default) and the limb-darkening columns, + the expanded views (kept alive by the caller)"""
    import ctypes

    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    cp, ds, ps, df = (vp * NIN)(), (i64 * NIN)(), (i64 * NIN)(), (ctypes.c_double * NIN)(*_PACK_DEFAULTS)
    keep = []
    for k, c in enumerate(ocols):
        if c is None:
            continue
        c = _dev(c.detach(), "orbit parameter")
        v = c.reshape(1, 1) if c.dim() == 0 else (c.unsqueeze(0) if c.dim() == 1 else c)
        v = v.expand(D, P)
        keep.append(v)
        cp[k], ds[k], ps[k] = v.data_ptr(), v.stride(0), v.stride(1)
    lp, ls = (vp * 4)(), (i64 * 4)()
    for k, c in enumerate(lcols):
        c = _dev(c.detach(), "limb-darkening coefficient")
        v = (c.reshape(1) if c.dim() == 0 else c).expand(D)
        keep.append(v)
        lp[k], ls[k] = v.data_ptr(), v.stride(0)
    return (cp, ds, ps, df, lp, ls), keep


@torch.no_grad()
def orbit_flux_value_and_grad(t, gflux, orbit_cols, ld_cols, flags=0, pack_flags=0, texp=None, stencil_dt=None, stencil_w=None,
                              gscale=None, events=(None, None), wanted=None):
    """Value AND gradient of ``L[d] = sum_n gflux[d, n] flux[d, n]`` for orbits in the standard parameterisation given column
    by column (as :func:`orbit_flux_dot`), WITHOUT autograd: returns ``(flux, L, gcols, gld_cols)`` -- ``gcols[k]`` the
    gradient of ``sum_d gscale[d] L[d]`` (``gscale`` None: of ``sum_d L[d]``) with respect to orbit column k in that
    column's own shape (None where the column was None), ``gld_cols`` likewise for (u1, u2[, u1s, u2s]).  What a sampler's
    leapfrog step needs, in ONE call of the library (exo_transit_flux_cols_vjp_f64): the packing rides on the windows +
    enumeration launch, then the sweep, then the packing VJP -- three launches at >= 512 draws, no autograd graph.
    ``wanted``: one bool per orbit column -- False: no gradient for it (None in ``gcols``)."""
    import ctypes

    ocols, lcols = list(orbit_cols), list(ld_cols)
    if len(ocols) != NIN or len(lcols) not in (2, 4):
        raise ValueError(f"{NIN} orbit columns (None = default) and 2 or 4 limb-darkening columns")
    D = P = 1
    for c in ocols:
        if c is not None:
            if c.dim() > 2:
                raise ValueError("orbit parameters may carry at most one draw dimension here")
            P = max(P, c.shape[-1] if c.dim() >= 1 else 1)
            D = max(D, c.shape[0] if c.dim() == 2 else 1)
    for c in lcols:
        if c.dim() > 1:
            raise ValueError("limb-darkening coefficients may carry at most one draw dimension")
        D = max(D, c.shape[0] if c.dim() == 1 else 1)
    host, keep = _cols_host(ocols, lcols, D, P)
    dev = keep[0].device
    vp = ctypes.c_void_p
    gcp, glp = (vp * NIN)(), (vp * 4)()
    wanted = [True] * NIN if wanted is None else list(wanted)
    outs = [None if (c is None or not w) else torch.empty(D, P, dtype=torch.float64, device=dev) for c, w in zip(ocols, wanted)]
    louts = [torch.empty(D, dtype=torch.float64, device=dev) for _ in lcols]
    for k, o in enumerate(outs):
        if o is not None:
            gcp[k] = o.data_ptr()
    for k, o in enumerate(louts):
        glp[k] = o.data_ptr()
    if gscale is not None:
        gscale = _dev(gscale, "gscale")
        if tuple(gscale.shape) != (D,):
            raise ValueError("gscale: one factor per draw")
    flux, _, _, dot = _cols_vjp(t, texp, stencil_dt, stencil_w, host, D, P, len(lcols) // 2, flags, pack_flags, gflux, events,
                                fold=(gscale, gcp, glp))

    def back(g, c):       # dense (D, P) / (D,) cotangents back to the column's own shape (sums over what was broadcast)
        if g is None:
            return None
        shp = tuple(c.shape)
        return g.reshape(shp) if g.numel() == _numel(shp) else g.sum_to_size(_bshape(shp, g.dim())).reshape(shp)

    return flux, dot, [back(g, c) for g, c in zip(outs, ocols)], [back(g, c) for g, c in zip(louts, lcols)]


class _OrbitFluxDot(torch.autograd.Function):
    """Record packing (column form: every constructor argument its own tensor), the one-sweep value + VJP light
    curve, and -- backward -- the packing VJP with the cotangent of L folded in (``gscale``): the whole
    user-level step of a standard-parameterisation orbit is the sweep plus TWO small kernels, where the
    composition pack_records -> transit_flux_dot costs seven (two concatenations, the pack, two scalings of
    the saved cotangents, the pack VJP, and whatever produced the cotangent of L)."""

    @staticmethod
    def forward(ctx, t, gflux, texp, stencil_dt, stencil_w, flags, pack_flags, events, n_ld, *cols):
        import ctypes

        t = _dev(t, "t")
        ocols, lcols = list(cols[:NIN]), list(cols[NIN:NIN + n_ld])
        D = P = 1
        for c in ocols:
            if c is not None:
                if c.dim() > 2:
                    raise ValueError("orbit parameters may carry at most one draw dimension here")
                P = max(P, c.shape[-1] if c.dim() >= 1 else 1)
                D = max(D, c.shape[0] if c.dim() == 2 else 1)
        for c in lcols:
            if c.dim() > 1:
                raise ValueError("limb-darkening coefficients may carry at most one draw dimension")
            D = max(D, c.shape[0] if c.dim() == 1 else 1)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        cp, ds, ps, df = (vp * NIN)(), (i64 * NIN)(), (i64 * NIN)(), (ctypes.c_double * NIN)(*_PACK_DEFAULTS)
        keep = []
        for k, c in enumerate(ocols):
            if c is None:
                continue
            c = _dev(c.detach(), "orbit parameter")
            v = c.reshape(1, 1) if c.dim() == 0 else (c.unsqueeze(0) if c.dim() == 1 else c)
            v = v.expand(D, P)
            keep.append(v)
            cp[k], ds[k], ps[k] = v.data_ptr(), v.stride(0), v.stride(1)
        lp, ls = (vp * 4)(), (i64 * 4)()
        for k, c in enumerate(lcols):
            c = _dev(c.detach(), "limb-darkening coefficient")
            v = (c.reshape(1) if c.dim() == 0 else c).expand(D)
            keep.append(v)
            lp[k], ls[k] = v.data_ptr(), v.stride(0)
        nset = n_ld // 2
        # (packing, windows + enumeration in ONE launch, then the sweep: exo_transit_flux_cols_vjp_f64; the packing VJP stays in
        # backward() -- the cotangent of L is not known before)
        flux, gparams, gld, dot = _cols_vjp(t, texp, stencil_dt, stencil_w, (cp, ds, ps, df, lp, ls), D, P, nset, flags,
                                            pack_flags, gflux, events)
        ctx.save_for_backward(gparams, gld, *keep)
        ctx.meta = (D, P, pack_flags, n_ld, [c is not None for c in ocols], [None if c is None else tuple(c.shape) for c in cols])
        ctx.set_materialize_grads(False)
        if isinstance(flux, SparseFlux):
            shape = torch.tensor([D, P], dtype=torch.int64)      # (host tensor: the wrapper needs the batch the columns implied)
            ctx.mark_non_differentiable(flux._ws, shape)
            return flux._ws, dot, shape
        ctx.mark_non_differentiable(flux)
        return flux, dot

    @staticmethod
    def backward(ctx, _gflux_unused, gdot, *_shape_unused):
        import ctypes

        D, P, pack_flags, n_ld, present, shapes = ctx.meta
        nfix = 9
        if gdot is None:
            return (None,) * (nfix + len(shapes))
        gparams, gld, *keep = ctx.saved_tensors
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        cp, ds, ps, df = (vp * NIN)(), (i64 * NIN)(), (i64 * NIN)(), (ctypes.c_double * NIN)(*_PACK_DEFAULTS)
        it = iter(keep)
        for k in range(NIN):
            if present[k]:
                v = next(it)
                cp[k], ds[k], ps[k] = v.data_ptr(), v.stride(0), v.stride(1)
        lp, ls = (vp * 4)(), (i64 * 4)()
        for k in range(n_ld):
            v = next(it)
            lp[k], ls[k] = v.data_ptr(), v.stride(0)
        dev = gparams.device
        gcp, glp = (vp * NIN)(), (vp * 4)()
        outs = [None] * len(shapes)
        for k in range(NIN):
            if present[k] and ctx.needs_input_grad[nfix + k]:
                outs[k] = torch.empty(D, P, dtype=torch.float64, device=dev)
                gcp[k] = outs[k].data_ptr()
        for k in range(n_ld):
            if ctx.needs_input_grad[nfix + NIN + k]:
                outs[NIN + k] = torch.empty(D, dtype=torch.float64, device=dev)
                glp[k] = outs[NIN + k].data_ptr()
        gdot = _dev(gdot, "gdot")
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.exo_pack_records_cols_vjp_f64(cp, ds, ps, df, lp, ls, D, P, pack_flags, _ptr(gparams), _ptr(gld),
                                                         _ptr(gdot), gcp, glp, _stream(gparams)),
                       "exo_pack_records_cols_vjp_f64")
        # dense (D, P) / (D,) cotangents back to the shapes the caller passed (sums over what was broadcast)
        grads = [None if g is None else (g.reshape(shp) if g.numel() == _numel(shp) else g.sum_to_size(_bshape(shp, g.dim())).reshape(shp))
                 for g, shp in zip(outs, shapes)]
        return (None,) * nfix + tuple(grads)


_PACK_DEFAULTS = (float("nan"), 0.0, 0.0, 0.0, 0.0, float("nan"), 1.0, 1.0, 0.0, 0.0)   # EXO_IN_* order; period, r: required


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _bshape(shape, ndim):
    """`shape` left-padded with ones to `ndim` dimensions (what sum_to_size reduces to)"""
    return (1,) * (ndim - len(shape)) + tuple(shape)


def orbit_flux_dot(t, gflux, orbit_cols, ld_cols, flags=0, pack_flags=0, texp=None, stencil_dt=None, stencil_w=None,
                   events=(None, None)):
    """``transit_flux_dot(t, *pack_records(stack(orbit_cols), stack(ld_cols)), gflux)`` without the stacking, the
    intermediate autograd nodes and the separate scaling of the cotangents: ``orbit_cols`` are the EXO_IN_* inputs
    (period, t0, b, ecc, omega, r, m_star, r_star, m_planet, sbr), each a tensor of shape (), (P,), (D, 1) or (D, P)
    -- or None for the constructor default -- and ``ld_cols`` are (u1, u2[, u1s, u2s]) of shape () or (D,).
    Returns ``(flux, L)`` like :func:`transit_flux_dot`; L is differentiable w.r.t. every column."""
    out = _OrbitFluxDot.apply(t, gflux, texp, stencil_dt, stencil_w, int(flags), int(pack_flags), events, len(ld_cols),
                              *orbit_cols, *ld_cols)
    if int(flags) & FLAG_SPARSE:
        D, P = (int(x) for x in out[2])
        return _sparse_from_ws(out[0], t.numel(), D, P, int(flags)), out[1]
    return out


def _pack_cols_forward(cols, n_ld, n_draw, pack_flags):
    """exo_pack_records_cols_f64 on a list of NIN orbit columns (None = constructor default) + n_ld limb-darkening
    columns: (params (D, P, 20), ld (D, 3|6), the expanded views to keep for the reverse call, meta)"""
    import ctypes

    ocols, lcols = list(cols[:NIN]), list(cols[NIN:NIN + n_ld])
    ref = next(c for c in ocols + lcols if c is not None)
    D, P = int(n_draw), 1
    for c in ocols:
        if c is not None:
            if c.dim() > 2:
                raise ValueError("orbit parameters may carry at most one draw dimension here")
            P = max(P, c.shape[-1] if c.dim() >= 1 else 1)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    cp, ds, ps, df = (vp * NIN)(), (i64 * NIN)(), (i64 * NIN)(), (ctypes.c_double * NIN)(*_PACK_DEFAULTS)
    keep = []
    for k, c in enumerate(ocols):
        if c is None:
            continue
        c = _dev(c.detach(), "orbit parameter")
        v = c.reshape(1, 1) if c.dim() == 0 else (c.unsqueeze(0) if c.dim() == 1 else c)
        v = v.expand(D, P)
        keep.append(v)
        cp[k], ds[k], ps[k] = v.data_ptr(), v.stride(0), v.stride(1)
    lp, ls = (vp * 4)(), (i64 * 4)()
    for k, c in enumerate(lcols):
        c = _dev(c.detach(), "limb-darkening coefficient")
        v = (c.reshape(1) if c.dim() == 0 else c).expand(D)
        keep.append(v)
        lp[k], ls[k] = v.data_ptr(), v.stride(0)
    nset = n_ld // 2
    params = torch.empty(D, P, NPAR, dtype=torch.float64, device=ref.device)
    ld = torch.empty(D, 3 * nset, dtype=torch.float64, device=ref.device)
    lib = _lib.load()
    with torch.cuda.device(ref.device):
        _lib.check(lib.exo_pack_records_cols_f64(cp, ds, ps, df, lp, ls, D, P, pack_flags, _ptr(params), _ptr(ld),
                                                 _stream(params)), "exo_pack_records_cols_f64")
    meta = (D, P, pack_flags, n_ld, [c is not None for c in ocols], [None if c is None else tuple(c.shape) for c in cols])
    return params, ld, keep, meta


def _pack_cols_backward(keep, meta, gparams, gld, gscale, needs):
    """exo_pack_records_cols_vjp_f64: cotangents of (params, ld), optionally scaled per draw by ``gscale``, back to the
    columns' own shapes; ``needs[k]``: whether column k wants one"""
    import ctypes

    D, P, pack_flags, n_ld, present, shapes = meta
    dev = keep[0].device
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    cp, ds, ps, df = (vp * NIN)(), (i64 * NIN)(), (i64 * NIN)(), (ctypes.c_double * NIN)(*_PACK_DEFAULTS)
    it = iter(keep)
    for k in range(NIN):
        if present[k]:
            v = next(it)
            cp[k], ds[k], ps[k] = v.data_ptr(), v.stride(0), v.stride(1)
    lp, ls = (vp * 4)(), (i64 * 4)()
    for k in range(n_ld):
        v = next(it)
        lp[k], ls[k] = v.data_ptr(), v.stride(0)
    gcp, glp = (vp * NIN)(), (vp * 4)()
    outs = [None] * len(shapes)
    for k in range(NIN):
        if present[k] and needs[k]:
            outs[k] = torch.empty(D, P, dtype=torch.float64, device=dev)
            gcp[k] = outs[k].data_ptr()
    for k in range(n_ld):
        if needs[NIN + k]:
            outs[NIN + k] = torch.empty(D, dtype=torch.float64, device=dev)
            glp[k] = outs[NIN + k].data_ptr()
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.exo_pack_records_cols_vjp_f64(cp, ds, ps, df, lp, ls, D, P, pack_flags, _ptr(gparams), _ptr(gld),
                                                     _ptr(gscale), gcp, glp, _stream(gparams)), "exo_pack_records_cols_vjp_f64")
    # dense (D, P) / (D,) cotangents back to the shapes the caller passed (sums over what was broadcast)
    return [None if g is None else (g.reshape(shp) if g.numel() == _numel(shp) else g.sum_to_size(_bshape(shp, g.dim())).reshape(shp))
            for g, shp in zip(outs, shapes)]


class _PackCols(torch.autograd.Function):
    """exo_pack_records_cols_f64 on its own: every constructor argument its own tensor (shape (), (P,), (D, 1) or
    (D, P); None = the constructor default), limb-darkening coefficients () or (D,) -> records (D, P, 20), ld (D, 3|6).
    No stacking pass forward, one packing-VJP launch in the reverse pass (the composition stack -> pack_records costs
    a concatenation each way plus a slice and a reduction per broadcast column)."""

    @staticmethod
    def forward(ctx, pack_flags, n_ld, n_draw, *cols):
        params, ld, keep, ctx.meta = _pack_cols_forward(cols, n_ld, n_draw, pack_flags)
        ctx.save_for_backward(*keep)
        return params, ld

    @staticmethod
    def backward(ctx, gparams, gld):
        D, P, _, n_ld = ctx.meta[:4]
        keep = ctx.saved_tensors
        dev = keep[0].device
        gparams = torch.zeros(D, P, NPAR, dtype=torch.float64, device=dev) if gparams is None else _dev(gparams, "gparams")
        gld = torch.zeros(D, 3 * (n_ld // 2), dtype=torch.float64, device=dev) if gld is None else _dev(gld, "gld")
        return (None,) * 3 + tuple(_pack_cols_backward(keep, ctx.meta, gparams, gld, None, ctx.needs_input_grad[3:]))


class _OrbitLoglike(torch.autograd.Function):
    """The white-noise log-likelihood of a batch of standard-parameterisation orbits straight from the constructor
    arguments: column-form packing, the one-call misfit + gradient (exo_transit_chi2[_ttv]_vjp_f64), and -- backward --
    the packing VJP with the likelihood's cotangent folded in (gscale = -gll / 2).  Seven launches for value and
    gradient where pack_records_cols -> transit_chi2 -> torch algebra takes eleven."""

    @staticmethod
    def forward(ctx, t, texp, stencil_dt, stencil_w, obs, ivar, cterm, flags, pack_flags, n_ld, n_draw, ttv_edges, ttv_shift,
                *cols):
        params, ld, keep, meta = _pack_cols_forward(cols, n_ld, n_draw, pack_flags)
        t, texp, n_texp, sdt, sw, n_sub, params, ld, D, P = _transit_args(t, texp, stencil_dt, stencil_w, params, ld, flags)
        flags |= _sorted_flag(t)
        edges, shift, n_edge = _ttv_args(None if ttv_edges is None else (ttv_edges, ttv_shift), D, P)
        N = t.numel()
        lib = _lib.load()
        nbytes = lib.exo_transit_flux_workspace_bytes(N, D, P)
        ws = torch.empty(max(nbytes // 8 + 1, 1), dtype=torch.float64, device=t.device)
        chi2 = torch.empty(D, dtype=torch.float64, device=t.device)
        gparams, gld = torch.empty_like(params), torch.empty_like(ld)
        gshift = torch.empty_like(shift) if n_edge else None
        with torch.cuda.device(t.device):
            if n_edge:
                _lib.check(lib.exo_transit_chi2_ttv_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub,
                                                            _ptr(params), _ptr(ld), D, P, flags, _ptr(edges), _ptr(shift),
                                                            n_edge, _ptr(obs), _ptr(ivar), ivar.numel(), _ptr(chi2),
                                                            _ptr(gparams), _ptr(gld), _ptr(gshift), _ptr(ws), nbytes,
                                                            _stream(t)), "exo_transit_chi2_ttv_vjp_f64")
            else:
                _lib.check(lib.exo_transit_chi2_vjp_f64(_ptr(t), N, _ptr(texp), n_texp, _ptr(sdt), _ptr(sw), n_sub, _ptr(params),
                                                        _ptr(ld), D, P, flags, _ptr(obs), _ptr(ivar), ivar.numel(), _ptr(chi2),
                                                        _ptr(gparams), _ptr(gld), _ptr(ws), nbytes, _stream(t)),
                           "exo_transit_chi2_vjp_f64")
        ctx.meta = meta
        ctx.n_keep = len(keep)
        ctx.save_for_backward(gparams, gld, *keep, *([gshift] if n_edge else []))
        return torch.add(cterm, chi2, alpha=-0.5)

    @staticmethod
    def backward(ctx, gll):
        saved = ctx.saved_tensors
        gparams, gld, keep = saved[0], saved[1], saved[2:2 + ctx.n_keep]
        gscale = -0.5 * _dev(gll, "gll")
        nfix = 13
        grads = _pack_cols_backward(keep, ctx.meta, gparams, gld, gscale, ctx.needs_input_grad[nfix:])
        gshift = gscale[:, None, None] * saved[2 + ctx.n_keep] if len(saved) > 2 + ctx.n_keep else None
        return (None,) * 11 + (None, gshift) + tuple(grads)


def orbit_white_noise_loglike(t, y, yerr, orbit_cols, ld_cols, n_draw, mean=0.0, flags=0, pack_flags=0, texp=None,
                              stencil_dt=None, stencil_w=None, ttv=None):
    """:func:`white_noise_loglike` for orbits in the standard parameterisation given column by column (the EXO_IN_*
    inputs of :func:`pack_records_cols`): Gaussian log-likelihood (n_draw,), differentiable w.r.t. every column and the
    timing shifts, value and gradient in seven launches."""
    t = _dev(t, "t")
    y = _dev(y, "y")
    if tuple(y.shape) != (t.numel(),):
        raise ValueError("y must have shape (n_cad,)")
    edges, shift = (None, None) if ttv is None else ttv

    def run(obs, ivar, cterm):
        return _OrbitLoglike.apply(t, texp, stencil_dt, stencil_w, obs, ivar, cterm, int(flags), int(pack_flags),
                                   len(ld_cols), int(n_draw), edges, shift, *orbit_cols, *ld_cols)

    yd = per_draw_yerr(yerr, y.numel())
    if yd is not None:
        # (_OrbitLoglike returns cterm - chi2 / 2: with cterm = 0 the unit-weight misfit is -2 x that)
        zero = _const(0.0, t.device)
        return _scaled_loglike(lambda obs, one: -2.0 * run(obs, one, zero), y, _dev(yd, "yerr"), mean)
    _check_data_terms(y, yerr, mean)
    obs, ivar, _, _, cterm = _white_noise_terms(y, yerr, mean)
    if ivar.numel() not in (1, t.numel()):
        raise ValueError("yerr must be a number, one entry per cadence, or per draw: (n_draw, 1)")
    return run(obs, ivar, cterm)


def pack_records_cols(orbit_cols, ld_cols, n_draw, pack_flags=0):
    """:func:`pack_records` without the stacking: ``orbit_cols`` are the EXO_IN_* inputs (period, t0, b, ecc, omega, r,
    m_star, r_star, m_planet, sbr), each a tensor of shape (), (P,), (D, 1) or (D, P) -- or None for the constructor
    default -- and ``ld_cols`` (u1, u2[, u1s, u2s]) of shape () or (D,).  Returns records (D, P, 20) and ld (D, 3|6),
    differentiable w.r.t. every column (one launch each way)."""
    return _PackCols.apply(int(pack_flags), len(ld_cols), int(n_draw), *orbit_cols, *ld_cols)


def pack_records(orbit_in, ld_in, flags=0):
    """(period, t0, b, ecc, omega, r, m_star, r_star, m_planet, sbr) per (draw, planet) and
    (u1, u2[, u1s, u2s]) per draw -> kernel records (n_draw, n_planet, 20) and Green's-basis
    limb-darkening coefficients (n_draw, 3|6): KeplerianOrbit.__init__ + get_cl + windows in
    one kernel, differentiable.  Slot order: include/exoplanet_amd.h EXO_IN_*."""
    return _PackRecords.apply(orbit_in, ld_in, int(flags))
