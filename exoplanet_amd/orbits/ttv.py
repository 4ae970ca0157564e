"""Keplerian orbit with transit-timing variations.

Mirror of ``exoplanet.orbits.TTVOrbit`` (/root/reference/src/exoplanet/orbits/ttv.py):
the time axis is warped so that each cadence is measured from its NEAREST
labelled transit, then the standard Keplerian machinery runs.  Only
``_warp_times`` differs from :class:`KeplerianOrbit`, so the light-curve classes
take their composed path for it (positions from ``ops.kepler``, flux from
``ops.quad_solution_vector``) -- the fused kernel assumes a single ``t0``.
Unbatched parameters only (per-planet transit lists are ragged).
"""
import numpy as np
import torch

from .keplerian import KeplerianOrbit, as_tensor

__all__ = ["TTVOrbit", "compute_expected_transit_times"]


def compute_expected_transit_times(min_time, max_time, period, t0):
    """expected (strictly periodic) transit times inside [min_time, max_time], one array per planet
    (ttv.py:10-33)"""
    out = []
    for p, t in zip(np.atleast_1d(period), np.atleast_1d(t0)):
        lo = np.floor((min_time - t) / p)
        hi = np.ceil((max_time - t) / p)
        times = t + p * np.arange(lo, hi, 1)
        out.append(times[(min_time <= times) & (times <= max_time)])
    return out


class TTVOrbit(KeplerianOrbit):
    """Args (beyond KeplerianOrbit's): exactly one of
        ttvs: per planet, the O-C offsets of each labelled transit (days);
        transit_times: per planet, the observed transit times (the least-squares
            period and t0 are derived from them);
    and optionally ``transit_inds`` (zero-based transit numbers when some transits
    are missing) and ``delta_log_period`` (with ``transit_times``)."""

    def __init__(self, *args, **kwargs):
        ttvs = kwargs.pop("ttvs", None)
        transit_times = kwargs.pop("transit_times", None)
        transit_inds = kwargs.pop("transit_inds", None)
        if ttvs is None and transit_times is None:
            raise ValueError("one of 'ttvs' or 'transit_times' must be defined")
        vec = lambda x: as_tensor(x).reshape(-1)  # noqa: E731
        if ttvs is not None:
            self.ttvs = [vec(x) for x in ttvs]
            if transit_inds is None:
                self.transit_inds = [torch.arange(x.shape[0], device=x.device) for x in self.ttvs]
            else:
                self.transit_inds = [torch.as_tensor(i, dtype=torch.int64, device=self.ttvs[0].device).reshape(-1)
                                     for i in transit_inds]
        else:
            # least-squares period and t0 from the labelled transit times (ttv.py:99-123)
            self.transit_times, self.ttvs, self.transit_inds = [], [], []
            period, t0 = [], []
            for i, times in enumerate(transit_times):
                times = vec(times)
                inds = (torch.arange(times.shape[0], device=times.device) if transit_inds is None
                        else torch.as_tensor(transit_inds[i], dtype=torch.int64, device=times.device).reshape(-1))
                self.transit_inds.append(inds)
                x = inds.to(torch.float64)
                N = times.shape[0]
                sumx, sumx2, sumy, sumxy = x.sum(), (x * x).sum(), times.sum(), (x * times).sum()
                denom = N * sumx2 - sumx ** 2
                slope = (N * sumxy - sumx * sumy) / denom
                intercept = (sumx2 * sumy - sumx * sumxy) / denom
                period.append(slope)
                t0.append(intercept)
                self.ttvs.append(times - (intercept + x * slope))
                self.transit_times.append(times)
            kwargs["t0"] = torch.stack(t0)
            self.ttv_period = torch.stack(period)
            if "period" not in kwargs:
                if "delta_log_period" in kwargs:
                    kwargs["period"] = torch.exp(torch.log(self.ttv_period) + as_tensor(kwargs.pop("delta_log_period")))
                else:
                    kwargs["period"] = self.ttv_period
        super().__init__(*args, **kwargs)
        self._standard = False  # the fused packing kernel assumes a single t0 per planet
        if ttvs is not None:
            self.ttv_period = self.period
            self.transit_times = [self.t0[i] + self.period[i] * self.transit_inds[i] + ttv
                                  for i, ttv in enumerate(self.ttvs)]
        # every transit, observed or not (ttv.py:141-147)
        self.all_transit_times = []
        for i, inds in enumerate(self.transit_inds):
            n_all = int(inds.max().item()) + 1
            expect = self.t0[i] + self.period[i] * torch.arange(n_all, device=inds.device)
            self.all_transit_times.append(expect.index_put((inds,), self.transit_times[i]))
        # histogram that maps a time to its nearest transit (ttv.py:149-163)
        self._bin_edges = [torch.cat([(tts[0] - 0.5 * self.ttv_period[i]).reshape(1), 0.5 * (tts[1:] + tts[:-1]),
                                      (tts[-1] + 0.5 * self.ttv_period[i]).reshape(1)])
                           for i, tts in enumerate(self.all_transit_times)]
        self._bin_values = [torch.cat([tts[:1], tts, tts[-1:]]) for tts in self.all_transit_times]

    def _get_model_dt(self, t):
        vals = []
        for edges, values in zip(self._bin_edges, self._bin_values):
            inds = torch.searchsorted(edges.detach().contiguous(), t.detach().contiguous())
            vals.append(values[inds])
        return torch.stack(vals, dim=-1)

    def _warp_times(self, t, _pad=True):
        """time since the nearest labelled transit (ttv.py:175-187)"""
        t = as_tensor(t, self.n)
        if _pad:
            return t.unsqueeze(-1) - self._get_model_dt(t)
        # already (..., P): warp each planet's column by its own transit list
        cols = []
        for i, (edges, values) in enumerate(zip(self._bin_edges, self._bin_values)):
            ti = t[..., i]
            inds = torch.searchsorted(edges.detach().contiguous(), ti.detach().contiguous())
            cols.append(ti - values[inds])
        return torch.stack(cols, dim=-1)
