"""Quadratically limb-darkened light curves on the fused HIP kernel.

Host-side mirror of ``exoplanet.light_curves.LimbDarkLightCurve``
(/root/reference/src/exoplanet/light_curves/limb_dark.py): same constructor,
same ``get_light_curve`` arguments / defaults / errors / output shape
``(n_cadence, n_planet)``.  For a :class:`~exoplanet_amd.orbits.KeplerianOrbit`
(the hot path) nothing O(N) happens in torch: the orbit packs per-(draw,
planet) records and one kernel goes from the time array to the flux array.
Any other orbit object (anything with ``get_relative_position``; with
``light_delay`` also a KeplerianOrbit) takes the composed path: its positions
feed ``ops.quad_solution_vector`` exactly as limb_dark.py:215-226 does.

Batched parameters (leading draw dimensions on the orbit / ``r`` / ``u``)
return ``(*draws, n_cadence, n_planet)``.
"""
import math
import warnings

import numpy as np
import torch

from .. import ops
from ..orbits.keplerian import KeplerianOrbit, _vec, as_tensor

__all__ = ["LimbDarkLightCurve"]


def get_cl(u1, u2):
    """(u1, u2) -> Green's-basis coefficients c, normalised so that an unocculted
    star has unit flux (limb_dark.py:11-18)."""
    u1 = as_tensor(u1)
    u2 = as_tensor(u2, u1)
    c0 = 1 - u1 - 1.5 * u2
    c1 = u1 + 2 * u2
    c2 = -0.25 * u2
    norm = math.pi * (c0 + c1 / 1.5)
    return torch.stack(torch.broadcast_tensors(c0, c1, c2), dim=-1) / norm.unsqueeze(-1)


def exposure_stencil(oversample=7, order=0):
    """sub-exposure offsets (in units of texp) and weights (limb_dark.py:181-197)"""
    oversample = int(oversample)
    oversample += 1 - oversample % 2
    stencil = np.ones(oversample)
    if order == 0:
        dt = np.linspace(-0.5, 0.5, 2 * oversample + 1)[1:-1:2]
    elif order == 1:
        dt = np.linspace(-0.5, 0.5, oversample)
        stencil[1:-1] = 2
    elif order == 2:
        dt = np.linspace(-0.5, 0.5, oversample)
        stencil[1:-1:2] = 4
        stencil[2:-1:2] = 2
    else:
        raise ValueError("order must be <= 2")
    return dt, stencil / np.sum(stencil)


_UPLOADS = {}


def _on_device(host, dev):
    """float64 device copy of a small host array (an exposure stencil), uploaded once per
    (content, device): a replayed hipGraph step cannot copy from the host, and need not"""
    host = np.ascontiguousarray(host, dtype=np.float64)
    key = (host.tobytes(), str(dev))
    t = _UPLOADS.get(key)
    if t is None:
        t = torch.as_tensor(host, dtype=torch.float64, device=dev)
        _UPLOADS[key] = t
    return t


def quad_limbdark_light_curve(c, b, r):
    """dot(s(b, r), c) - 1 (limb_dark.py:21-24)"""
    b = as_tensor(b)
    r = as_tensor(r, b)
    b, r = torch.broadcast_tensors(b, r)
    s = ops.quad_solution_vector(b.contiguous(), r.contiguous())
    return (s * c).sum(-1) - 1.0


def _batch_shape(ll, batch, batched):
    """the per-draw log-likelihoods in the orbit's batch shape; ONE system with several per-draw error bars (yerr of shape
    (n_draw, 1): a jitter chain per entry) keeps yerr's draws -- ADVICE r3: that case used to die in reshape(())"""
    n = 1
    for b in batch:
        n *= int(b)
    if ll.numel() == n:
        return ll.reshape(tuple(batch)) if batched else ll.reshape(())
    return ll.reshape(-1)


class LimbDarkLightCurve:
    """A quadratically limb darkened light curve.

    Args:
        u1, u2: the limb darkening coefficients (scalars, or one per draw).
            Passing a length-2 vector as ``u1`` alone is accepted with a
            DeprecationWarning, as in the reference (limb_dark.py:43-64).
    """

    def __init__(self, u1, u2=None, model=None):
        if u2 is None:
            warnings.warn("using a vector of limb darkening coefficients is deprecated; use u1 and u2 directly",
                          DeprecationWarning)
            u = as_tensor(u1)
            if u.dim() != 1 or u.shape[0] != 2:
                raise AssertionError("only quadratic limb darkening is supported; use `starry` for more flexibility")
            self.u1, self.u2 = u[0], u[1]
        else:
            self.u1 = as_tensor(u1)
            self.u2 = as_tensor(u2, self.u1)
        self._c = None

    @property
    def c(self):
        """Green's-basis coefficients (limb_dark.py:66: the reference builds them in the constructor, as a node of a lazy
        graph).  Here on first use: the fused paths never read them -- the packing kernel forms its own from (u1, u2) --
        and eleven small torch kernels per light-curve object were a quarter of a sampler's likelihood step."""
        if self._c is None:
            self._c = get_cl(self.u1, self.u2)
        return self._c

    def get_ror_from_approx_transit_depth(self, delta, b, jac=False):
        """radius ratio for an approximate depth in the small-planet limit (limb_dark.py:70-97)"""
        b = as_tensor(b)
        delta = as_tensor(delta, b)
        f0 = 1 - 2 * self.u1 / 6.0 - 2 * self.u2 / 12.0
        arg = 1 - torch.sqrt(1 - b ** 2)
        f = 1 - self.u1 * arg - self.u2 * arg ** 2
        factor = f0 / f
        ror = torch.sqrt(delta * factor)
        if not jac:
            return ror.reshape(b.shape)
        return ror.reshape(b.shape), (0.5 * factor / ror).reshape(b.shape)

    # ------------------------------------------------------------------
    def get_light_curve(self, orbit=None, r=None, t=None, texp=None, oversample=7, order=0,
                        use_in_transit=None, light_delay=False, total=False, cadence_major=False, sparse=False):
        """Relative flux ``(n_cadence, n_planet)``; arguments as in the reference
        (limb_dark.py:99-153).  ``use_in_transit`` defaults to ``not light_delay``.
        ``total=True`` (not in the reference): the sum over the planets, ``(n_cadence,)`` -- what the tutorials
        write as ``pt.sum(light_curves, axis=-1)`` -- formed inside the kernels; as a separate torch reduction it is a
        pass over the (draws, cadences) array of its own (0.5 ms of a 5.3 ms C3 step).
        ``cadence_major=True`` (with ``total``, a batch of draws): the (draws, cadences) result is the transposed view of
        a (cadences, draws) array -- same shape, same values, draws innermost in memory.  That is the layout the
        celerite kernels read a mean model in (and write its cotangent in) with contiguous accesses: pass the result
        as the ``mean`` of a ``GaussianProcess`` (C3: 4.3 -> 3.6 ms per value + gradient).  Ignored where the fused
        sweep does not offer it (per-cadence exposure times, occultations or light delay together with timing tables).
        ``sparse=True`` (with ``total``, a batch of draws): the result is an ``ops.SparseLightCurve`` -- the runs of cadences in
        which a planet can overlap the disk and the flux of those cadences, every other cadence being 0 -- for use as the
        ``mean`` of a ``GaussianProcess``, whose kernels read the segments directly: the (draws, cadences) array, 97 % zeros,
        and its cotangent are never written (C3: 3.8 -> 3.1 ms per value + gradient).  ``.dense()`` gives the ordinary tensor.
        Where the sparse form is not offered (several planets, occultations, timing tables, light delay, per-cadence exposure
        times) the cadence-major dense array is returned instead: a ``GaussianProcess`` takes either."""
        if orbit is None:
            raise ValueError("missing required argument 'orbit'")
        if r is None:
            raise ValueError("missing required argument 'r'")
        if t is None:
            raise ValueError("missing required argument 't'")
        use_in_transit = (not light_delay) if use_in_transit is None else use_in_transit
        if texp is not None:
            stencil = exposure_stencil(oversample, order)   # raises for order > 2 like the reference
        else:
            stencil = None
        if light_delay and use_in_transit:
            # keplerian.py:720-723: the reference's in_transit refuses light delay
            raise NotImplementedError("Light travel time delay not yet implemented for `in_transit`")
        keplerian = isinstance(orbit, KeplerianOrbit) and type(orbit)._warp_times is KeplerianOrbit._warp_times
        if keplerian and light_delay and self._fusable_delay(orbit, t, texp):
            # second Kepler solve in the same kernel (EXO_FLAG_LIGHT_DELAY)
            return self._fused(orbit, r, t, texp, stencil, use_in_transit, light_delay=True, total=total,
                               cadence_major=cadence_major or sparse)
        if isinstance(orbit, KeplerianOrbit) and not light_delay and (keplerian or hasattr(orbit, "kernel_ttv")):
            return self._fused(orbit, r, t, texp, stencil, use_in_transit, total=total, cadence_major=cadence_major,
                               sparse=sparse)
        lc = self._composed(orbit, r, t, texp, stencil, use_in_transit, light_delay)
        return lc.sum(-1) if total else lc

    @staticmethod
    def _fusable_delay(orbit, t, texp):
        """the light-delay kernels take a scalar (or no) exposure time and no sky rotation beyond what the
        flux needs; anything else stays on the composed path"""
        scalar_texp = texp is None or as_tensor(texp).numel() == 1
        return scalar_texp and as_tensor(t).dim() == 1

    # ---- hot path: one packing kernel + the fused light-curve kernels, nothing O(N) in torch
    def _fused(self, orbit, r, t, texp, stencil, use_in_transit, secondary=None, light_delay=False, total=False,
               cadence_major=False, sparse=False):
        t = as_tensor(t, r if isinstance(r, torch.Tensor) else self.u1)
        if t.dim() != 1:
            raise ValueError("t must be a vector of times")
        sec = None if secondary is None else ((secondary[0].u1, secondary[0].u2), secondary[1])
        rec, ld, batch, flags = orbit.kernel_inputs(r, (self.u1, self.u2), use_in_transit=use_in_transit,
                                                    secondary=sec, light_delay=light_delay)
        t = t.to(rec.device)
        kw = {}
        if hasattr(orbit, "kernel_ttv"):
            if secondary is not None:
                raise ValueError("a TTVOrbit has no flipped orbit (the reference's _flip cannot build one either)")
            # timing tables and records share one draw batch
            edges, shift = orbit.kernel_ttv()
            full = torch.broadcast_shapes(edges.shape[:-2], tuple(batch))
            P = rec.shape[1]
            rec = rec.reshape(tuple(batch) + rec.shape[1:]).expand(full + rec.shape[1:]).reshape(-1, P, rec.shape[2])
            ld = ld.reshape(tuple(batch) + ld.shape[1:]).expand(full + ld.shape[1:]).reshape(-1, ld.shape[1])
            kw["ttv"] = (edges.expand(full + edges.shape[-2:]).reshape(-1, P, edges.shape[-1]).contiguous(),
                         shift.expand(full + shift.shape[-2:]).reshape(-1, P, shift.shape[-1]).contiguous())
            rec, ld, batch = rec.contiguous(), ld.contiguous(), full
        if texp is not None:
            dt, w = stencil
            kw.update(texp=as_tensor(texp, t).reshape(-1).detach(), stencil_dt=_on_device(dt, t.device),
                      stencil_w=_on_device(w, t.device))
        if total:
            # cadence-major output: run-enumeration sweeps only (one exposure time at most; timing tables without
            # occultations / light delay), a one-dimensional batch of more than one draw
            n_texp = kw["texp"].numel() if "texp" in kw else 0
            if (sparse and len(tuple(batch)) == 1 and t.is_cuda
                    and ops.sparse_mean_supported(n_texp, "ttv" in kw, flags, rec.shape[1])):
                return ops.transit_flux_sparse_model(t.detach(), rec, ld, flags=flags, **kw)
            cadence_major = cadence_major or sparse
            cm = (cadence_major and len(tuple(batch)) == 1 and rec.shape[0] > 1 and n_texp <= 1
                  and not ("ttv" in kw and (secondary is not None or light_delay)))
            flux = ops.transit_flux(t.detach(), rec, ld, flags=flags | (ops.FLAG_CADENCE_MAJOR if cm else 0), **kw)
            return flux if cm else flux.reshape(tuple(batch) + (t.shape[0],))
        flux = ops.transit_flux(t.detach(), rec, ld, flags=flags | ops.FLAG_PER_PLANET, **kw)
        return flux.reshape(tuple(batch) + (t.shape[0], rec.shape[1]))

    def white_noise_log_likelihood(self, orbit=None, r=None, t=None, y=None, yerr=None, mean=0.0, texp=None, oversample=7,
                                   order=0, use_in_transit=False, light_delay=False):
        """Gaussian log-likelihood (one value per draw) of the observed series ``y`` with independent errors ``yerr``
        given ``mean + sum over planets of get_light_curve(...)`` -- what the reference's tutorials write as
        ``pm.Normal("obs", mu=mean + pt.sum(light_curves, axis=-1), sigma=yerr, observed=y)`` -- for a KeplerianOrbit
        or a TTVOrbit (gradients to its transit times / offsets included) with sorted times and one exposure time: value and gradient in ONE call on the sparse light curve
        (ops.transit_chi2), no (draws, cadences) array anywhere.  ``yerr``: a number, one value per cadence, or PER DRAW
        (a 0-d or (draws, 1) tensor, differentiable: a jitter term sampled per chain costs nothing extra).  Whatever
        the fused form cannot differentiate -- a per-cadence ``yerr``, a ``mean`` or a ``y`` that requires grad, a
        ``mean`` per draw -- takes the dense light curve (``get_light_curve(total=True)``) and torch: slower, never a
        partial gradient."""
        from ..orbits.keplerian import KeplerianOrbit

        if orbit is None or r is None or t is None or y is None or yerr is None:
            raise ValueError("orbit, r, t, y and yerr are required")
        needs = lambda x: isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled()  # noqa: E731
        n_cad = as_tensor(t).numel()
        per_draw = ops.per_draw_yerr(yerr, n_cad) is not None
        if needs(y) or needs(mean) or (needs(yerr) and not per_draw) or (isinstance(mean, torch.Tensor) and mean.numel() > 1):
            lc = self.get_light_curve(orbit=orbit, r=r, t=t, texp=texp, oversample=oversample, order=order,
                                      use_in_transit=use_in_transit, light_delay=light_delay, total=True)
            resid = as_tensor(y, lc).to(lc.device) - mean - lc
            w = as_tensor(yerr, lc).to(lc.device) ** -2
            lognorm = torch.log(w / (2.0 * math.pi))
            lognorm = lognorm.sum(-1) if (lognorm.dim() and lognorm.shape[-1] == n_cad) else lognorm.reshape(lognorm.shape[:-1] if lognorm.dim() else ()) * float(n_cad)
            return -0.5 * (w * resid * resid).sum(-1) + 0.5 * lognorm
        has_ttv = hasattr(orbit, "kernel_ttv")
        if not isinstance(orbit, KeplerianOrbit) or (type(orbit)._warp_times is not KeplerianOrbit._warp_times and not has_ttv):
            raise NotImplementedError("white_noise_log_likelihood needs a KeplerianOrbit or a TTVOrbit")
        if has_ttv and light_delay:
            raise NotImplementedError("white_noise_log_likelihood: no light delay together with timing variations")
        t = as_tensor(t, r if isinstance(r, torch.Tensor) else self.u1)
        fused = self._loglike_from_columns(orbit, r, t, y, yerr, mean, texp, oversample, order, use_in_transit, light_delay, has_ttv)
        if fused is not None:
            return fused
        rec, ld, batch, flags = orbit.kernel_inputs(r, (self.u1, self.u2), use_in_transit=use_in_transit,
                                                    light_delay=light_delay)
        t = t.to(rec.device)
        kw = {}
        if texp is not None:
            dt, w = exposure_stencil(oversample, order)
            kw.update(texp=as_tensor(texp, t).reshape(-1).detach(), stencil_dt=_on_device(dt, t.device),
                      stencil_w=_on_device(w, t.device))
        if has_ttv:
            # timing tables and records share one draw batch (as in get_light_curve)
            edges, shift = orbit.kernel_ttv()
            full = torch.broadcast_shapes(edges.shape[:-2], tuple(batch))
            P = rec.shape[1]
            rec = rec.reshape(tuple(batch) + rec.shape[1:]).expand(full + rec.shape[1:]).reshape(-1, P, rec.shape[2])
            ld = ld.reshape(tuple(batch) + ld.shape[1:]).expand(full + ld.shape[1:]).reshape(-1, ld.shape[1])
            kw["ttv"] = (edges.expand(full + edges.shape[-2:]).reshape(-1, P, edges.shape[-1]).contiguous(),
                         shift.expand(full + shift.shape[-2:]).reshape(-1, P, shift.shape[-1]).contiguous())
            rec, ld, batch = rec.contiguous(), ld.contiguous(), full
        ll = ops.white_noise_loglike(t.detach(), rec, ld, as_tensor(y, t).to(rec.device), yerr, mean=mean, flags=flags, **kw)
        return _batch_shape(ll, batch, bool(batch))

    def _loglike_from_columns(self, orbit, r, t, y, yerr, mean, texp, oversample, order, use_in_transit, light_delay, has_ttv):
        """white_noise_log_likelihood of the standard parameterisation with the constructor arguments handed to the
        kernels as they are (ops.orbit_white_noise_loglike: packing, misfit + gradient, packing VJP with the
        likelihood's cotangent folded in); None when that form does not apply"""
        if not getattr(orbit, "_standard", False) or not t.is_cuda or not isinstance(mean, (int, float)):
            return None
        A = orbit._args
        like = next((x for x in list(A.values()) + [r] if isinstance(x, torch.Tensor)), None)
        got = orbit._standard_cols(r, (self.u1, self.u2), None, like)
        if got is None:
            return None
        cols, us, D, batched = got
        kw = {}
        if has_ttv:
            edges, shift = orbit.kernel_ttv()
            if edges.dim() > 3:
                return None
            Dt = edges.shape[0] if edges.dim() == 3 else 1
            if Dt != 1 and D != 1 and Dt != D:
                return None
            D = max(D, Dt)
            batched = batched or edges.dim() == 3
            kw["ttv"] = (edges.expand((D,) + tuple(edges.shape[-2:])).contiguous(), shift.expand((D,) + tuple(shift.shape[-2:])).contiguous())
        if texp is not None:
            dt, w = exposure_stencil(oversample, order)
            kw.update(texp=as_tensor(texp, t).reshape(-1).detach(), stencil_dt=_on_device(dt, t.device),
                      stencil_w=_on_device(w, t.device))
        flags = (ops.FLAG_WINDOW if use_in_transit else 0) | (ops.FLAG_LIGHT_DELAY if light_delay else 0)
        pack_flags = (flags & ops.FLAG_WINDOW) | (ops.PACK_CIRCULAR if A["ecc"] is None else 0)
        yt = as_tensor(y, t).to(t.device)
        ll = ops.orbit_white_noise_loglike(t.detach(), yt, yerr, cols, us, D, mean=mean, flags=flags, pack_flags=pack_flags, **kw)
        return ll if batched else _batch_shape(ll, (), False)

    # ---- generic orbit objects: ops.quad_solution_vector on their positions
    def _composed(self, orbit, r, t, texp, stencil, use_in_transit, light_delay):
        t = as_tensor(t)
        r = _vec(r, t).reshape(-1)
        n_all = t.shape[0]
        if use_in_transit:
            inds = orbit.in_transit(t, r=r, texp=texp, light_delay=light_delay)
            t = t[inds]
        if texp is None:
            tgrid = t
        else:
            dt, w = stencil
            texp_t = as_tensor(texp, t)
            dt = torch.as_tensor(dt, dtype=torch.float64, device=t.device)
            if texp_t.dim() == 0:
                dt = texp_t * dt
            else:
                dt = (texp_t[inds] if use_in_transit else texp_t).unsqueeze(-1) * dt
            tgrid = t.unsqueeze(-1) + dt
        coords = orbit.get_relative_position(tgrid, light_delay=light_delay)
        shape = tuple(tgrid.shape) + (r.shape[0],)
        b = torch.sqrt(coords[0] ** 2 + coords[1] ** 2).reshape(shape)
        los = coords[2].reshape(shape)
        rs = orbit.r_star
        lc = self._compute_light_curve(b / rs, (r + torch.zeros(shape, dtype=torch.float64, device=t.device)) / rs,
                                       los / rs)
        if texp is not None:
            wt = torch.as_tensor(w, dtype=torch.float64, device=t.device)
            lc = (wt[None, :, None] * lc).sum(dim=1)
        if use_in_transit:
            out = torch.zeros((n_all, r.shape[0]), dtype=torch.float64, device=t.device)
            return out.index_put((inds,), lc)
        return lc

    def _compute_light_curve(self, b, r, los=None):
        """flux for separations ``b`` and radius ratios ``r`` in units of the stellar
        radius; zero where ``los <= 0`` (limb_dark.py:234-252)"""
        b = as_tensor(b)
        r = as_tensor(r, b)
        c = self.c.to(b.device)
        lc = quad_limbdark_light_curve(c, b, r)
        if los is None:
            return lc
        return torch.where(as_tensor(los, b) > 0, lc, torch.zeros_like(lc))
