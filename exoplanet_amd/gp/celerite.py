"""GaussianProcess on the batched celerite HIP kernels (value + gradient)."""
import os

import torch

from .. import _lib
from ..ops import SparseLightCurve, _dev, _ptr, _stream, is_cadence_major
from ..orbits.keplerian import as_tensor
from .terms import Term

__all__ = ["GaussianProcess", "celerite_loglike", "celerite_loglike_sparse"]

MAX_J = 16     # (every width on the time-parallel path since round 6 -- include/exoplanet_amd.h, EXO_GP_MAX_J)
PREPARE_ADJOINT = 0x40000000   # EXO_GP_PREPARE_ADJOINT: or-ed into n_chunks of both calls of a pair (include/exoplanet_amd.h)
# A forward call that autograd will follow with a reverse call may ask the library to run the adjoint scan beside the forward
# chunk kernel.  OFF unless EXO_GP_PREPARE_ADJOINT=1 (or gp.celerite._PREPARE[0] = True): measured at the bench shapes it is a
# wash to a loss -- the scan's short kernels are bound by memory latency, which a neighbour saturating HBM with checkpoint stores
# stretches by about what the overlap hides (C5 -2 %, J = 10 -3 %, sparse mean -3.5 %; C3 +2.5 %, a batch with draws on the
# robust route +11 %: DESIGN.md section 7).
_PREPARE = [os.environ.get("EXO_GP_PREPARE_ADJOINT", "0").strip() not in ("0", "")]


def default_chunks():
    """EXO_GP_CHUNKS (0 / unset: the library's plan; 1: sequential recurrences; > 1: that many chunks).
    Read here, by the caller -- the library itself reads no environment and keeps no state: the
    value travels to the forward call, and through the autograd context to the reverse call."""
    v = os.environ.get("EXO_GP_CHUNKS", "").strip()
    n = int(v) if v else 0
    if n < 0:
        raise ValueError("EXO_GP_CHUNKS must be >= 0")
    return n


_POISON = [None]    # tests: fill every buffer handed to the library with this value first (nothing may depend on what
                    # a workspace or an output held before the call -- under hipGraph replay that is the previous step's)


def _buffer(*shape, device):
    x = torch.empty(*shape, dtype=torch.float64, device=device)
    if _POISON[0] is not None:
        x.fill_(_POISON[0])
    return x


class _CeleriteLogLike(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, resid, diag, coef_real, coef_complex, obs, pair_kind, n_chunks):
        t = _dev(t, "t")
        cm = obs is not None and is_cadence_major(resid)
        if not cm:
            resid = _dev(resid, "resid")
        if obs is not None:
            obs = _dev(obs, "obs")
            if obs.shape != t.shape:
                raise ValueError("obs must have one entry per cadence")
        diag = _dev(diag, "diag")
        coef_real = _dev(coef_real, "coef_real")
        coef_complex = _dev(coef_complex, "coef_complex")
        D, N = resid.shape
        n_real, n_complex = coef_real.shape[1], coef_complex.shape[1]
        if t.shape != (N,) or diag.dim() != 2 or diag.shape[1] != N or diag.shape[0] not in (1, D):
            raise ValueError("shapes: t (N,), resid (D,N), diag (1|D, N)")
        if coef_real.shape != (D, n_real, 2) or coef_complex.shape != (D, n_complex, 4):
            raise ValueError("coef_real (D,Jr,2), coef_complex (D,Jc,4)")
        if pair_kind is not None:
            if not pair_kind.is_cuda or pair_kind.dtype != torch.int32 or tuple(pair_kind.shape) != (D, n_complex):
                raise ValueError("pair_kind must be an int32 device tensor of shape (D, Jc)")
            pair_kind = pair_kind.contiguous()
        J = n_real + 2 * n_complex
        if not 1 <= J <= MAX_J:
            raise ValueError(f"celerite state width J = {J} outside 1..{MAX_J}")
        if N < 1:
            raise ValueError("need at least one cadence")
        n_chunks = int(n_chunks)
        lib = _lib.load()
        need_grad = any(ctx.needs_input_grad)
        if need_grad and _PREPARE[0]:
            n_chunks |= PREPARE_ADJOINT     # (travels to the reverse call through ctx.dims)
        loglike = _buffer(D, device=t.device)
        # The state buffer is what the reverse pass re-reads, and it is also what lets the library run
        # the recurrences in parallel over time: a value-only call gets one too (scratch, freed on
        # return) unless the device cannot spare it, in which case the sequential kernels run.
        nstate = lib.exo_celerite_state_doubles(N, D, n_real, n_complex, n_chunks)
        try:
            state = _buffer(nstate, device=t.device)
        except torch.cuda.OutOfMemoryError:
            if need_grad:
                raise
            state, nstate = None, 0
        with torch.cuda.device(t.device):
            if obs is None:
                _lib.check(
                    lib.exo_celerite_loglike_fwd_f64(_ptr(t), _ptr(resid), _ptr(diag), diag.shape[0], N,
                                                     _ptr(coef_real), n_real, _ptr(coef_complex), n_complex,
                                                     _ptr(pair_kind), D, _ptr(loglike), _ptr(state), nstate, n_chunks,
                                                     _stream(t)),
                    "exo_celerite_loglike_fwd_f64",
                )
            else:
                fn = lib.exo_celerite_loglike_obs_fwd_cm_f64 if cm else lib.exo_celerite_loglike_obs_fwd_f64
                _lib.check(
                    fn(_ptr(t), _ptr(obs), _ptr(resid), _ptr(diag), diag.shape[0], N, _ptr(coef_real), n_real,
                       _ptr(coef_complex), n_complex, _ptr(pair_kind), D, _ptr(loglike), _ptr(state), nstate, n_chunks,
                       _stream(t)),
                    "exo_celerite_loglike_obs_fwd_cm_f64" if cm else "exo_celerite_loglike_obs_fwd_f64",
                )
        if need_grad:
            # the series itself is saved too: the reverse pass of the checkpointed path recomputes from it
            ctx.save_for_backward(t, resid, diag, coef_real, coef_complex, state, obs, pair_kind)
            ctx.dims = (D, N, n_real, n_complex, nstate, n_chunks)
            ctx.cm = cm
        return loglike

    @staticmethod
    def backward(ctx, gll):
        t, resid, diag, coef_real, coef_complex, state, obs, pair_kind = ctx.saved_tensors
        D, N, n_real, n_complex, nstate, n_chunks = ctx.dims
        gll = _dev(gll, "gloglike")
        lib = _lib.load()
        if ctx.cm:     # the cotangent of a cadence-major model in the model's layout
            gresid = _buffer(N, D, device=t.device).t()
        else:
            gresid = _buffer(D, N, device=t.device)
        want_diag = ctx.needs_input_grad[2]
        shared_diag = diag.shape[0] == 1
        gdiag = _buffer(D, N, device=t.device) if want_diag else None
        gcr = _buffer(*coef_real.shape, device=t.device)
        gcc = _buffer(*coef_complex.shape, device=t.device)
        with torch.cuda.device(t.device):
            tail = (_ptr(diag), diag.shape[0], N, _ptr(coef_real), n_real, _ptr(coef_complex), n_complex, _ptr(pair_kind),
                    D, _ptr(gll), _ptr(state), nstate, n_chunks, _ptr(gresid), _ptr(gdiag), None, _ptr(gcr), _ptr(gcc),
                    _stream(t))
            if obs is None:
                _lib.check(lib.exo_celerite_loglike_vjp_f64(_ptr(t), _ptr(resid), *tail), "exo_celerite_loglike_vjp_f64")
            elif ctx.cm:
                _lib.check(lib.exo_celerite_loglike_obs_vjp_cm_f64(_ptr(t), _ptr(obs), _ptr(resid), *tail),
                           "exo_celerite_loglike_obs_vjp_cm_f64")
            else:
                _lib.check(lib.exo_celerite_loglike_obs_vjp_f64(_ptr(t), _ptr(obs), _ptr(resid), *tail),
                           "exo_celerite_loglike_obs_vjp_f64")
        if want_diag and shared_diag:
            gdiag = gdiag.sum(0, keepdim=True)
        return None, gresid, gdiag, gcr, gcc, None, None, None


_SORT_DRAWS = [os.environ.get("EXO_SPARSE_SORT", "1") != "0"]     # (0: A/B -- the draws in the caller's order)


def _transit_order(sp):
    """int32 (D,): the draws of a sparse light curve in the order of the mean spacing of their segments -- the period, in
    cadences: draws with neighbouring periods keep their transits together all along the series, wherever the reference
    transit time lies -- with the start of the first segment breaking ties (draws with fewer than two segments: by that
    alone).  One launch of the library (exo_sparse_model_order); beyond its 4096 draws the same keys sorted by torch.  No host
    synchronisation either way (usable inside a captured step)."""
    import ctypes

    D = sp.n_draw
    lib = _lib.load()
    if D <= lib_max_order_draws():
        order = torch.empty(D, dtype=torch.int32, device=sp.values.device)
        model = sp.model_struct()
        with torch.cuda.device(order.device):
            _lib.check(lib.exo_sparse_model_order(ctypes.addressof(model), D, _ptr(order), _stream(order)), "exo_sparse_model_order")
        return order
    if hasattr(sp, "segments"):          # a merged light curve (ops.MergedSparseLightCurve)
        nseg, seg, _ = sp.segments()
        nrun, lo = nseg.long(), seg[:, :, 0]
    else:
        lay = sp.layout()
        nrun = lay.nrun.reshape(D).long()
        lo = lay.runs.reshape(D, lay.r_max, 4)[:, :, 0]
    first = lo[:, 0].double()
    last = lo.gather(1, (nrun - 1).clamp_min(0).unsqueeze(1)).squeeze(1).double()
    key = (last - first) / (nrun - 1).clamp_min(1).double() + 1e-9 * first
    return torch.argsort(key, stable=True).to(torch.int32)


def lib_max_order_draws():
    return 4096      # include/exoplanet_amd.h: EXO_SPARSE_ORDER_MAX_DRAWS


class _CeleriteLogLikeSparse(torch.autograd.Function):
    """log N(obs - model | 0, K + diag) per draw for a SPARSE per-draw model (ops.SparseLightCurve: segments of cadences +
    their values; exo_celerite_loglike_sparse_*_f64): `vals` is the differentiable input, its cotangent comes back at the
    positions of the values; neither the dense model nor its cotangent exists"""

    @staticmethod
    def forward(ctx, t, vals, diag, coef_real, coef_complex, obs, pair_kind, n_chunks, sp):
        import ctypes

        t, obs, diag = _dev(t, "t"), _dev(obs, "obs"), _dev(diag, "diag")
        coef_real, coef_complex = _dev(coef_real, "coef_real"), _dev(coef_complex, "coef_complex")
        D, N = sp.n_draw, sp.n_cad
        n_real, n_complex = coef_real.shape[1], coef_complex.shape[1]
        if t.shape != (N,) or obs.shape != (N,) or diag.dim() != 2 or diag.shape[1] != N or diag.shape[0] not in (1, D):
            raise ValueError("shapes: t (N,), obs (N,), diag (1|D, N)")
        if coef_real.shape != (D, n_real, 2) or coef_complex.shape != (D, n_complex, 4):
            raise ValueError("coef_real (D,Jr,2), coef_complex (D,Jc,4)")
        if vals.data_ptr() != sp.values.data_ptr() or not vals.is_contiguous():
            raise ValueError("vals must be the value array of the sparse light curve")
        if pair_kind is not None:
            if not pair_kind.is_cuda or pair_kind.dtype != torch.int32 or tuple(pair_kind.shape) != (D, n_complex):
                raise ValueError("pair_kind must be an int32 device tensor of shape (D, Jc)")
            pair_kind = pair_kind.contiguous()
        J = n_real + 2 * n_complex
        if not 1 <= J <= MAX_J:
            raise ValueError(f"celerite state width J = {J} outside 1..{MAX_J}")
        n_chunks = int(n_chunks)
        lib = _lib.load()
        if n_chunks == 0:     # the plan the sparse entries do best with (twice the dense plan's chunks for J <= 2: include/exoplanet_amd.h)
            n_chunks = int(lib.exo_celerite_default_chunks(N, D, n_real, n_complex, 1))
        need_grad = any(ctx.needs_input_grad)
        if need_grad and _PREPARE[0]:
            n_chunks |= PREPARE_ADJOINT     # (travels to the reverse call through ctx.dims)
        loglike = _buffer(D, device=t.device)
        nstate = lib.exo_celerite_state_doubles(N, D, n_real, n_complex, n_chunks)
        try:
            state = _buffer(nstate, device=t.device)
        except torch.cuda.OutOfMemoryError:
            if need_grad:
                raise
            state, nstate = None, 0
        # The kernels' lane is a draw, and a wave pays for a transit while ANY of its 64 draws is inside one: the kernels take the
        # draws in the order of their periods (draws with neighbouring periods keep their transits together all along the series)
        # -- exo_sparse_model.row_of_draw: every array stays in the caller's order, the library indexes through the table.
        perm = _transit_order(sp) if (D > 64 and _SORT_DRAWS[0]) else None
        model = sp.model_struct()
        model.row_of_draw = _ptr(perm)
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_celerite_loglike_sparse_fwd_f64(_ptr(t), _ptr(obs), ctypes.addressof(model), _ptr(diag),
                                                               diag.shape[0], N, _ptr(coef_real), n_real, _ptr(coef_complex),
                                                               n_complex, _ptr(pair_kind), D, _ptr(loglike), _ptr(state), nstate,
                                                               n_chunks, _stream(t)), "exo_celerite_loglike_sparse_fwd_f64")
        if need_grad:
            ctx.save_for_backward(t, vals, diag, coef_real, coef_complex, state, obs, pair_kind, perm)
            ctx.dims = (D, N, n_real, n_complex, nstate, n_chunks)
            ctx.sp = sp
        return loglike

    @staticmethod
    def backward(ctx, gll):
        import ctypes

        t, vals, diag, coef_real, coef_complex, state, obs, pair_kind, perm = ctx.saved_tensors
        D, N, n_real, n_complex, nstate, n_chunks = ctx.dims
        gll = _dev(gll, "gloglike")
        lib = _lib.load()
        # Only the positions a segment covers are DEFINED -- in `vals` (the rest of the sweep's value array is workspace the
        # forward never wrote either) and in this cotangent alike; the reverse sweep of the light curve walks the same runs and
        # reads nothing else.  A zero fill would be the 1.2 GB write per step the sparse route exists to avoid (C3), so the array
        # is not cleared; it goes through _buffer so that the poison test covers it, and
        # SparseLightCurve.values.grad is to be read through the runs (ops.SparseLightCurve._indices), not as a dense array.
        gvals = _buffer(*vals.shape, device=vals.device)
        want_diag = ctx.needs_input_grad[2]
        gdiag = _buffer(D, N, device=t.device) if want_diag else None
        gcr, gcc = _buffer(*coef_real.shape, device=t.device), _buffer(*coef_complex.shape, device=t.device)
        model = ctx.sp.model_struct()
        model.row_of_draw = _ptr(perm)
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_celerite_loglike_sparse_vjp_f64(_ptr(t), _ptr(obs), ctypes.addressof(model), _ptr(diag),
                                                               diag.shape[0], N, _ptr(coef_real), n_real, _ptr(coef_complex),
                                                               n_complex, _ptr(pair_kind), D, _ptr(gll), _ptr(state), nstate,
                                                               n_chunks, _ptr(gvals), _ptr(gdiag), None, _ptr(gcr), _ptr(gcc),
                                                               _stream(t)), "exo_celerite_loglike_sparse_vjp_f64")
        if want_diag and diag.shape[0] == 1:
            gdiag = gdiag.sum(0, keepdim=True)
        return None, gvals, gdiag, gcr, gcc, None, None, None, None


def celerite_loglike_sparse(t, model, diag, coef_real, coef_complex, obs, pair_kind=None, n_chunks=None):
    """:func:`celerite_loglike` of ``obs - model`` for a :class:`~exoplanet_amd.ops.SparseLightCurve` ``model``"""
    if n_chunks is None:
        n_chunks = default_chunks()
    return _CeleriteLogLikeSparse.apply(t, model.values, diag, coef_real, coef_complex, obs.detach(),
                                        None if pair_kind is None else pair_kind.detach(), n_chunks, model)


def celerite_loglike(t, resid, diag, coef_real, coef_complex, obs=None, pair_kind=None, n_chunks=None):
    """log N(resid | 0, K + diag) per draw.  t (N,), resid (D,N), diag (1|D,N),
    coef_real (D,Jr,2) = (a,c), coef_complex (D,Jc,4) = (a,b,c,d).  Differentiable
    w.r.t. resid, diag and the coefficients.  With ``obs`` (N,), ``resid`` is a per-draw
    MODEL and the likelihood is that of ``obs - resid``, formed inside the kernels (no
    residual array, no sign-flip pass in the backward); ``obs`` itself gets no gradient.
    ``pair_kind`` (D,Jc) int32: 1 where a row of ``coef_complex`` holds two real terms
    (a1, c1, a2, c2) instead of a complex one (include/exoplanet_amd.h).  ``n_chunks``: how the
    series is cut for the time-parallel recurrences (None: EXO_GP_CHUNKS, else the library's plan)."""
    if n_chunks is None:
        n_chunks = default_chunks()
    return _CeleriteLogLike.apply(t, resid, diag, coef_real, coef_complex, None if obs is None else obs.detach(),
                                  None if pair_kind is None else pair_kind.detach(), n_chunks)


_CONST_VAR = {}   # (n, device, dtype, yerr) -> the variance vector of a scalar error bar (a handful of series at most)


def _known_sorted(t):
    """ops.known_sorted: looked at once per (storage, version); inside a hipGraph capture an unseen tensor passes"""
    from ..ops import known_sorted

    return known_sorted(t, unknown=True)


class GaussianProcess:
    """``GaussianProcess(kernel, t=t, diag=..., yerr=..., mean=...)`` then
    ``log_likelihood(y)``; modelled on celerite2.GaussianProcess.

    Batched use: kernel hyper-parameters, ``mean`` and ``y`` may carry a leading
    draw dimension D; ``log_likelihood`` then returns (D,).
    """

    def __init__(self, kernel, t=None, *, mean=0.0, **kwargs):
        if not isinstance(kernel, Term):
            raise TypeError("kernel must be an exoplanet_amd.gp.terms.Term")
        self.kernel = kernel
        self.mean = mean
        self._t = None
        if t is not None:
            self.compute(t, **kwargs)

    def compute(self, t, *, yerr=None, diag=None, check_sorted=True, **_):
        t = as_tensor(t)
        if t.dim() != 1:
            raise ValueError("dimension mismatch: t must be 1-D")
        if check_sorted and t.numel() > 1 and not _known_sorted(t):
            raise ValueError("the input coordinates must be sorted")
        if yerr is None and diag is None:
            var = torch.zeros_like(t)
        elif yerr is not None:
            if diag is not None:
                raise ValueError("only one of 'diag' and 'yerr' can be provided")
            if isinstance(yerr, (int, float)):
                # one error bar for the whole series: the same constant vector for every object built on these times --
                # made once (a sampler builds a GaussianProcess per evaluation: three small kernels each time otherwise)
                # (a bool is an int to isinstance: as an error bar it is a mistake, caught by the float() below all the same)
                # SHARED and read-only: every object on this key holds the same tensor as `_diag` -- never written in
                # place by this package; a caller who wants to edit a GP's diagonal passes `diag=`.  Not made
                # during a hipGraph capture: that tensor would live in the graph's private pool, filled on replay only.
                key = (t.shape[0], str(t.device), str(t.dtype), float(yerr))
                var = _CONST_VAR.get(key)
                if var is None:
                    var = torch.full_like(t, float(yerr) ** 2)
                    if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
                        if len(_CONST_VAR) >= 8:
                            _CONST_VAR.clear()
                        _CONST_VAR[key] = var
            else:
                var = as_tensor(yerr, t) ** 2 + torch.zeros_like(t)
        else:
            var = as_tensor(diag, t) + torch.zeros_like(t)
        if var.shape[-1] != t.shape[0]:
            raise ValueError("dimension mismatch: diag / yerr must have one entry per cadence")
        self._t = t
        self._diag = var if var.dim() == 2 else var.reshape(1, -1)

    def _mean_at(self, tq, like):
        """the mean as a tensor: a callable evaluated at ``tq``, a sparse light curve as its dense array"""
        if isinstance(self.mean, SparseLightCurve):
            return self.mean.dense()
        return self.mean(tq) if callable(self.mean) else as_tensor(self.mean, like)

    def _coefficients(self):
        """(coef_real (D,Jr,2), pair slots (D,Jc,4), their kind (D,Jc) int32 or None, D, batched?)"""
        ar, cr, pairs, kind = self.kernel.pair_coefficients()
        batch = torch.broadcast_shapes(ar.shape[:-1], pairs.shape[:-2])
        if len(batch) > 1:
            raise ValueError("at most one draw dimension is supported")
        D = batch[0] if batch else 1
        real = torch.stack(torch.broadcast_tensors(ar, cr), dim=-1)
        real = real.expand((D,) + tuple(real.shape[-2:]))
        cplx = pairs.expand((D,) + tuple(pairs.shape[-2:]))
        if kind is not None:
            kind = kind.expand((D, kind.shape[-1]))
        return real, cplx, kind, D, bool(batch)

    def _prepare(self, y, fuse=False):
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")
        t = self._t
        y = as_tensor(y, t)
        sparse = None
        if isinstance(self.mean, SparseLightCurve):
            # one observed series against a sparse per-draw model: the kernels read the segments (log_likelihood); every other
            # use gets the dense array
            if (fuse and y.dim() == 1 and not y.requires_grad and y.is_cuda and y.shape[0] == t.shape[0]
                    and self.mean.n_cad == t.shape[0]):
                sparse = self.mean
                mean = None
            else:
                mean = self.mean.dense()
        else:
            mean = self.mean(t) if callable(self.mean) else as_tensor(self.mean, t)
        if sparse is not None:
            real, cplx, kind, D, batched = self._coefficients()
            if D not in (1, sparse.n_draw) or self._diag.shape[0] not in (1, sparse.n_draw):
                raise ValueError("dimension mismatch: the kernel's / diagonal's draws and the light curve's")
            D = sparse.n_draw
            real = real.expand(D, real.shape[1], 2).contiguous()
            cplx = cplx.expand(D, cplx.shape[1], 4).contiguous()
            if kind is not None:
                kind = kind.expand(D, kind.shape[1]).contiguous()
            return t, None, sparse, real, cplx, False, y, kind
        if isinstance(mean, torch.Tensor) and mean.dim() == 1 and mean.shape[0] != t.shape[0]:
            mean = mean.unsqueeze(-1)  # per-draw constant
        # one observed series against a per-draw mean model: the kernels form obs - model themselves
        # (no (D, N) residual array in the forward pass, no (D, N) negation in the backward one)
        obs = None
        if (fuse and y.dim() == 1 and not y.requires_grad and isinstance(mean, torch.Tensor) and mean.dim() == 2
                and mean.shape[-1] == t.shape[0] and y.shape[0] == t.shape[0] and y.is_cuda):
            obs, resid = y, mean
        else:
            resid = y - mean
        if resid.shape[-1] != t.shape[0]:
            raise ValueError("dimension mismatch")
        real, cplx, kind, D, batched = self._coefficients()
        squeeze = resid.dim() == 1 and not batched and self._diag.shape[0] == 1
        D = max(D, resid.shape[0] if resid.dim() == 2 else 1, self._diag.shape[0])
        resid = resid.expand(D, t.shape[0])
        if not (obs is not None and is_cadence_major(resid)):   # (a cadence-major model goes to the kernels as it is)
            resid = resid.contiguous()
        real = real.expand(D, real.shape[1], 2).contiguous()
        cplx = cplx.expand(D, cplx.shape[1], 4).contiguous()
        if kind is not None:
            kind = kind.expand(D, kind.shape[1]).contiguous()
        return t, mean, resid, real, cplx, squeeze, obs, kind

    def log_likelihood(self, y):
        t, _, resid, real, cplx, squeeze, obs, kind = self._prepare(y, fuse=True)
        if isinstance(resid, SparseLightCurve):
            return celerite_loglike_sparse(t.detach(), resid, self._diag.contiguous(), real, cplx, obs, pair_kind=kind)
        ll = celerite_loglike(t.detach(), resid, self._diag.contiguous(), real, cplx, obs=obs, pair_kind=kind)
        return ll[0] if squeeze else ll

    def apply_inverse(self, y):
        """alpha = (K + diag)^-1 (y - mean), per draw (detached).  It is minus the gradient of the
        log-likelihood with respect to y, i.e. one forward + one reverse pass of the recurrences."""
        t, _, resid, real, cplx, squeeze, _, kind = self._prepare(y)
        with torch.enable_grad():
            r = resid.detach().requires_grad_(True)
            ll = celerite_loglike(t.detach(), r, self._diag.detach().contiguous(), real.detach(), cplx.detach(),
                                  pair_kind=kind)
            (g,) = torch.autograd.grad(ll.sum(), r)
        alpha = -g
        return alpha[0] if squeeze else alpha

    def dot_tril(self, x):
        """``L x`` with ``K + diag = L L^T`` (celerite2's ``GaussianProcess.dot_tril``), per draw, detached:
        ``x`` (N,) or (D, N).  O(N) recurrence on the device."""
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")
        t = self._t
        real, cplx, kind, D, batched = self._coefficients()
        x = as_tensor(x, t).detach()
        squeeze = x.dim() == 1 and not batched and self._diag.shape[0] == 1
        D = max(D, x.shape[0] if x.dim() == 2 else 1, self._diag.shape[0])
        x = _dev(x.expand(D, t.shape[0]), "x")
        real = _dev(real.detach().expand(D, real.shape[1], 2), "coef_real")
        cplx = _dev(cplx.detach().expand(D, cplx.shape[1], 4), "coef_complex")
        kind = None if kind is None else kind.expand(D, kind.shape[1]).contiguous()
        diag = _dev(self._diag.detach(), "diag")
        z = torch.empty_like(x)
        lib = _lib.load()
        with torch.cuda.device(t.device):
            _lib.check(lib.exo_celerite_dot_tril_f64(_ptr(t), _ptr(diag), diag.shape[0], t.shape[0], _ptr(real), real.shape[1],
                                                     _ptr(cplx), cplx.shape[1], _ptr(kind), D, _ptr(x), _ptr(z), _stream(t)),
                       "exo_celerite_dot_tril_f64")
        return z[0] if squeeze else z

    def sample(self, size=None, include_mean=True, generator=None):
        """draws from the prior N(mean, K + diag) at the data times (celerite2's ``GaussianProcess.sample``):
        ``size`` draws per parameter draw -> (size, [D,] N); None -> ([D,] N)"""
        t = self._t
        if t is None:
            raise RuntimeError("you must call 'compute' first")
        _, _, _, D, batched = self._coefficients()
        D = max(D, self._diag.shape[0])
        n_s = 1 if size is None else int(size)
        out = []
        for _ in range(n_s):
            x = torch.randn(D, t.shape[0], dtype=torch.float64, device=t.device, generator=generator)
            z = self.dot_tril(x if (batched or D > 1) else x[0])
            if include_mean:
                m = self._mean_at(t, t)
                if isinstance(m, torch.Tensor) and m.dim() == 1 and m.shape[0] != t.shape[0]:
                    m = m.unsqueeze(-1)
                z = z + (m.detach() if isinstance(m, torch.Tensor) else m)
            out.append(z)
        return out[0] if size is None else torch.stack(out)

    def predict(self, y, t=None, *, include_mean=True):
        """Conditional mean of the process given ``y`` (celerite2's ``GaussianProcess.predict``
        without the variance), detached.  At the data times (``t=None``) it is
        ``y - diag * alpha``; at other times ``K(t, t_data) alpha`` by the O(N + M) forward / backward
        recurrences of exo_celerite_predict_f64 (``t`` sorted)."""
        tt, mean, resid, real, cplx, squeeze, _, kind = self._prepare(y)
        alpha = self.apply_inverse(y)
        alpha2 = alpha if alpha.dim() == 2 else alpha.unsqueeze(0)
        if t is None:
            mu = resid.detach() - self._diag.detach() * alpha2
            tq = tt
        else:
            tq = as_tensor(t, tt)
            if tq.dim() != 1:
                raise ValueError("dimension mismatch: t must be 1-D")
            if tq.numel() > 1 and bool((tq[1:] < tq[:-1]).any()):
                raise ValueError("the input coordinates must be sorted")
            tq = _dev(tq.detach(), "t")
            D = alpha2.shape[0]
            al = _dev(alpha2.detach(), "alpha")
            re, cx = _dev(real.detach(), "coef_real"), _dev(cplx.detach(), "coef_complex")
            mu = torch.empty(D, tq.shape[0], dtype=torch.float64, device=tt.device)
            lib = _lib.load()
            with torch.cuda.device(tt.device):
                _lib.check(lib.exo_celerite_predict_f64(_ptr(tt), tt.shape[0], _ptr(al), _ptr(re), re.shape[1], _ptr(cx),
                                                        cx.shape[1], _ptr(kind), D, _ptr(tq), tq.shape[0], _ptr(mu),
                                                        _stream(tt)), "exo_celerite_predict_f64")
        if include_mean:
            m = self._mean_at(tq, tt)
            if isinstance(m, torch.Tensor) and m.dim() == 1 and m.shape[0] != tq.shape[0]:
                m = m.unsqueeze(-1)
            if isinstance(m, torch.Tensor) and m.dim() >= 1 and m.shape[-1] == tt.shape[0] and t is not None:
                raise ValueError("a tabulated mean cannot be evaluated at new times: pass a callable mean")
            mu = mu + (m.detach() if isinstance(m, torch.Tensor) else m)
        return mu[0] if squeeze else mu
