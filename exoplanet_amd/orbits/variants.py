"""Orbit variants that reuse the Keplerian machinery.

``TTVOrbit``            transit-timing variations: every cadence is measured from its
                        nearest labelled transit (reference: src/exoplanet/orbits/ttv.py).
``SimpleTransitOrbit``  straight-line transit from observables, no Kepler solve
                        (reference: src/exoplanet/orbits/simple.py).

``TTVOrbit`` light curves run in the fused kernels (the per-planet transit lists are
ragged, so they live in one padded ``(..., n_planet, n_transit_max)`` table that the
kernels search per cadence; include/exoplanet_amd.h, ``exo_transit_flux_ttv_*``); its
position / velocity methods, and everything of ``SimpleTransitOrbit``, go through the
composed path (``ops.kepler`` / ``ops.quad_solution_vector``), unbatched.
"""
import numpy as np
import torch

from .keplerian import KeplerianOrbit, _vec, as_tensor

__all__ = ["TTVOrbit", "SimpleTransitOrbit", "compute_expected_transit_times"]


def compute_expected_transit_times(min_time, max_time, period, t0):
    """strictly periodic transit times inside [min_time, max_time]: one array per planet"""
    result = []
    for per, ref in zip(np.atleast_1d(period), np.atleast_1d(t0)):
        first = np.floor((min_time - ref) / per)
        last = np.ceil((max_time - ref) / per)
        cand = ref + per * np.arange(first, last, 1)
        result.append(cand[(cand >= min_time) & (cand <= max_time)])
    return result


class _TransitTable:
    """nearest-transit lookup for all planets (and all draws) at once.

    Row p holds planet p's complete transit list (observed or interpolated), shape
    ``batch + (n_p,)``; ``edges`` ``batch + (P, width + 1)`` are the midpoints between
    neighbours, closed by half a period on either side (ttv.py:158-166), padded with +inf;
    ``centres[..., p, k]`` is the transit a time in bin k belongs to (ttv.py:167-170)."""

    def __init__(self, all_times, ttv_period):
        P = len(all_times)
        width = max(int(x.shape[-1]) for x in all_times)
        batch = torch.broadcast_shapes(ttv_period.shape[:-1], *[x.shape[:-1] for x in all_times])
        dev = all_times[0].device
        self.edges = torch.full(batch + (P, width + 1), float("inf"), dtype=torch.float64, device=dev)
        centres = []
        for p, tts in enumerate(all_times):
            tts = tts.expand(batch + tts.shape[-1:])
            n = tts.shape[-1]
            half = 0.5 * ttv_period[..., p].expand(batch).unsqueeze(-1)
            row = torch.cat([tts[..., :1] - half, 0.5 * (tts[..., 1:] + tts[..., :-1]), tts[..., -1:] + half], dim=-1)
            self.edges[..., p, :n + 1] = row.detach()
            # bins: (-inf, e0] -> first transit, (e_k, e_k+1] -> transit k, beyond the last edge -> last transit
            centres.append(torch.cat([tts[..., :1], tts, tts[..., -1:].expand(batch + (width + 1 - n,))], dim=-1))
        self.centres = torch.stack(centres, dim=-2)   # batch + (P, width + 2), differentiable in the transit times

    def nearest(self, t_by_planet):
        """t_by_planet (P, ...) -> the matching transit time of each entry, same shape (unbatched tables)"""
        if self.edges.dim() != 2:
            raise ValueError("the composed TTVOrbit path is unbatched; batched timing tables go through "
                             "LimbDarkLightCurve.get_light_curve (fused kernels)")
        flat = t_by_planet.reshape(t_by_planet.shape[0], -1).detach().contiguous()
        idx = torch.searchsorted(self.edges, flat)
        return torch.gather(self.centres, 1, idx).reshape(t_by_planet.shape)


_INDEX_CACHE = {}


def _device_index(host, dev):
    """int64 device copy of a small host index array, uploaded once per (content, device): a
    host-to-device copy is not allowed while a hipGraph is capturing, the cached tensor is"""
    key = (host.tobytes(), str(dev))
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = torch.as_tensor(host, dtype=torch.int64, device=dev)
        _INDEX_CACHE[key] = t
    return t


class TTVOrbit(KeplerianOrbit):
    """KeplerianOrbit plus exactly one of ``ttvs`` (O-C offsets per labelled transit, per
    planet) or ``transit_times`` (observed times; the least-squares period and t0 follow
    from them), optionally ``transit_inds`` (zero-based transit numbers when transits are
    missing) and, with ``transit_times``, ``delta_log_period``.

    Beyond the reference: each ``ttvs[p]`` / ``transit_times[p]`` may carry leading draw
    dimensions ``(..., n_transit_p)`` (broadcast against the other parameters' draw
    dimensions); ``LimbDarkLightCurve.get_light_curve`` then evaluates all draws in the
    fused kernels.  The position / velocity methods stay unbatched like the reference.

    ``transit_inds`` are structure, not parameters: they are read on the host (lists / numpy
    arrays cost nothing; device tensors are copied back once, which a hipGraph capture does
    not allow)."""

    def __init__(self, *args, ttvs=None, transit_times=None, transit_inds=None, **kwargs):
        if ttvs is None and transit_times is None:
            raise ValueError("one of 'ttvs' or 'transit_times' must be defined")

        def rows(x):
            x = as_tensor(x)
            return x.reshape(-1) if x.dim() == 0 else x

        given = [rows(x) for x in (ttvs if ttvs is not None else transit_times)]
        dev = given[0].device
        if transit_inds is None:
            self._inds = [None] * len(given)         # 0, 1, 2, ...: nothing missing
            self._counts = [int(x.shape[-1]) for x in given]
            self.transit_inds = [torch.arange(n, device=dev) for n in self._counts]
        else:
            host = [(i.detach().cpu().numpy() if isinstance(i, torch.Tensor) else np.asarray(i)).astype(np.int64).reshape(-1)
                    for i in transit_inds]
            self._inds = host
            self._counts = [int(h.max()) + 1 for h in host]
            self.transit_inds = [_device_index(h, dev) for h in host]
        self._given = "ttvs" if ttvs is not None else "transit_times"
        if ttvs is not None:
            self.ttvs = given
        else:
            # straight-line fit time = intercept + slope * index per planet (ttv.py:99-123)
            self.transit_times = given
            fits = [self._line_fit(ix.to(torch.float64), tt) for ix, tt in zip(self.transit_inds, given)]
            slopes = torch.broadcast_tensors(*[f[0] for f in fits])
            self.ttv_period = torch.stack(slopes, dim=-1)
            kwargs["t0"] = torch.stack(torch.broadcast_tensors(*[f[1] for f in fits]), dim=-1)
            self.ttvs = [tt - (f[1].unsqueeze(-1) + ix.to(torch.float64) * f[0].unsqueeze(-1))
                         for f, ix, tt in zip(fits, self.transit_inds, given)]
            if "period" not in kwargs:
                dlp = kwargs.pop("delta_log_period", None)
                kwargs["period"] = self.ttv_period if dlp is None else torch.exp(torch.log(self.ttv_period) + as_tensor(dlp))
        super().__init__(*args, **kwargs)
        # (the base class's `_standard` stands: the fused kernels take the ordinary records plus the
        # timing tables, so the standard parameterisation still goes through the packing kernel)
        if ttvs is not None:
            self.ttv_period = self._ephemeris()[1]
        # transit_times (with `ttvs`), all_transit_times and the lookup table are built on first use (__getattr__): a
        # fused light-curve call of the common case -- offsets for every transit -- never needs them (kernel_ttv)
        self._tables_ready = False

    _TABLE_ATTRS = ("transit_times", "all_transit_times", "_table")

    def __getattr__(self, name):
        if name in TTVOrbit._TABLE_ATTRS and not self.__dict__.get("_tables_ready", True):
            self._build_tables()
            return self.__dict__[name]
        return super().__getattr__(name)

    def _build_tables(self):
        self._tables_ready = True
        dev = self.ttvs[0].device
        t0, period = self._ephemeris()
        if "transit_times" not in self.__dict__:
            self.transit_times = [t0[..., p:p + 1] + period[..., p:p + 1] * ix + dv
                                  for p, (ix, dv) in enumerate(zip(self.transit_inds, self.ttvs))]
        # fill unobserved transit numbers with the linear ephemeris (ttv.py:141-147)
        self.all_transit_times = []
        for p, host in enumerate(self._inds):
            obs = self.transit_times[p]
            if host is None:
                self.all_transit_times.append(obs)
                continue
            count = self._counts[p]
            grid = t0[..., p:p + 1] + period[..., p:p + 1] * torch.arange(count, device=dev)
            shape = torch.broadcast_shapes(grid.shape[:-1], obs.shape[:-1])
            grid = grid.expand(shape + (count,)).clone()
            grid[..., self.transit_inds[p]] = obs.expand(shape + obs.shape[-1:])
            self.all_transit_times.append(grid)
        self._table = _TransitTable(self.all_transit_times, self.ttv_period)

    def _ephemeris(self):
        """(t0, period), shape (..., P), without running the base class's attribute algebra when the
        constructor was handed both (the usual case; the fused path then never needs the algebra)"""
        A = self._args
        if A["period"] is not None and A["t_periastron"] is None:
            like = next((x for x in A.values() if isinstance(x, torch.Tensor)), None)
            period = _vec(A["period"], like)
            t0 = _vec(0.0 if A["t0"] is None else A["t0"], like)
            shape = torch.broadcast_shapes(period.shape, t0.shape)
            return t0.expand(shape), period.expand(shape)
        return self.t0, self.period

    @staticmethod
    def _line_fit(x, y):
        n = x.shape[0]
        sx, sxx, sy, sxy = x.sum(), (x * x).sum(), y.sum(-1), (x * y).sum(-1)
        det = n * sxx - sx * sx
        return (n * sxy - sx * sy) / det, (sxx * sy - sx * sxy) / det   # slope, intercept

    def _warp_times(self, t, _pad=True):
        """time since the nearest labelled transit, shape (..., P) (ttv.py:175-187)"""
        t = as_tensor(t, self.n)
        P = self._table.edges.shape[-2]
        per_planet = t.unsqueeze(0).expand((P,) + tuple(t.shape)) if _pad else t.movedim(-1, 0)
        return (per_planet - self._table.nearest(per_planet)).movedim(0, -1)

    def _fused_tables(self):
        """the common case -- offsets `ttvs` for every transit, at most one draw dimension, ROCm tensors -- in one launch
        (ops.ttv_tables: exo_ttv_tables_f64) instead of the torch construction; None when it does not apply"""
        from .. import ops

        if self._given != "ttvs" or any(h is not None for h in self._inds):
            return None
        t0, period = self._ephemeris()
        parts = list(self.ttvs) + [t0, period]
        if not all(isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float64 for x in parts):
            return None
        if any(x.dim() > 2 for x in parts):
            return None
        as2 = lambda x: x if x.dim() == 2 else x.unsqueeze(0)  # noqa: E731
        rows = [as2(x) for x in parts]
        D = max(x.shape[0] for x in rows)
        if any(x.shape[0] not in (1, D) for x in rows):
            return None
        P = len(self.ttvs)
        per, ref = rows[-1].expand(-1, P) if rows[-1].shape[1] == 1 else rows[-1], rows[-2].expand(-1, P) if rows[-2].shape[1] == 1 else rows[-2]
        edges, shift = ops.ttv_tables(per, ref, rows[:P], D)
        batched = any(x.dim() == 2 for x in parts)
        return (edges, shift) if batched else (edges[0], shift[0])

    def kernel_ttv(self):
        """the fused kernels' timing tables (include/exoplanet_amd.h): bin edges ``batch + (P, E)``
        (no gradient, like the reference's searchsorted) and per-bin shifts ``batch + (P, E + 1)``
        = transit time of the bin - t0, differentiable"""
        fused = self._fused_tables()
        if fused is not None:
            return fused
        centres = self._table.centres
        t0, _ = self._ephemeris()
        shape = torch.broadcast_shapes(centres.shape[:-1], t0.shape)
        shift = centres.expand(shape + centres.shape[-1:]) - t0.expand(shape).unsqueeze(-1)
        return self._table.edges.expand(shape + self._table.edges.shape[-1:]), shift


class SimpleTransitOrbit:
    """Planets crossing the stellar disk on straight lines at constant speed, set by the
    observables (period, duration, t0, b): x = speed * dt, y = b R_star, in front of the
    star only for |dt| < duration / 2."""

    def __init__(self, period, duration, t0=0.0, b=0.0, r_star=1.0, ror=0):
        self.period = _vec(period)
        like = self.period
        self.duration, self.t0, self.b = _vec(duration, like), _vec(t0, like), _vec(b, like)
        self.r_star = _vec(r_star, like)
        chord = self.r_star * torch.sqrt((1 + _vec(ror, like)) ** 2 - self.b ** 2)   # half the path across the disk
        self.speed = 2 * chord / self.duration
        self._y = self.b * self.r_star

    def _since_transit(self, t):
        """(N, P) time since the nearest transit centre, in [-period/2, period/2)"""
        t = as_tensor(t, self.period).unsqueeze(-1)
        half = 0.5 * self.period
        return torch.remainder(t - self.t0 + half, self.period) - half

    def get_relative_position(self, t, light_delay=False):
        if light_delay:
            raise NotImplementedError("Light travel time delay is not implemented for simple orbits")
        dt = self._since_transit(t)
        front = torch.where(dt.abs() < 0.5 * self.duration, 1.0, -1.0).to(torch.float64)
        return (self.speed * dt).squeeze(), (self._y + 0 * dt).squeeze(), front.squeeze()

    def get_planet_position(self, t, light_delay=False):
        return self.get_relative_position(t, light_delay=False)

    def get_star_position(self, t, light_delay=False):
        origin = torch.zeros_like(as_tensor(t, self.period))
        return origin, origin, origin

    def _no_velocity(self, *_, **__):
        raise NotImplementedError("a SimpleTransitOrbit has no velocity")

    get_planet_velocity = get_star_velocity = get_radial_velocity = _no_velocity

    def in_transit(self, t, r=None, texp=None, light_delay=False):
        if light_delay:
            raise NotImplementedError("Light travel time delay is not implemented for simple orbits")
        t = as_tensor(t, self.period)
        if r is None:
            reach = 0.5 * self.duration
        else:
            reach = torch.sqrt((_vec(r, self.period) + self.r_star) ** 2 - self._y ** 2) / self.speed
        if texp is not None:
            reach = reach + 0.5 * as_tensor(texp, self.period)
        hit = (self._since_transit(t).abs() < reach).any(dim=-1)
        return torch.nonzero(hit).reshape(-1)
