"""Straight-line transit "orbit" parameterised by observables.

Mirror of ``exoplanet.orbits.SimpleTransitOrbit``
(/root/reference/src/exoplanet/orbits/simple.py): no Kepler solve at all; the
light-curve classes evaluate it through their composed path
(``ops.quad_solution_vector`` on these positions)."""
import torch

from .keplerian import as_tensor, _vec

__all__ = ["SimpleTransitOrbit"]


class SimpleTransitOrbit:
    def __init__(self, period, duration, t0=0.0, b=0.0, r_star=1.0, ror=0):
        self.period = _vec(period)
        self.t0 = _vec(t0, self.period)
        self.b = _vec(b, self.period)
        self.duration = _vec(duration, self.period)
        self.r_star = _vec(r_star, self.period)
        ror = _vec(ror, self.period)
        self._b_norm = self.b * self.r_star
        x2 = self.r_star ** 2 * ((1 + ror) ** 2 - self.b ** 2)
        self.speed = 2 * torch.sqrt(x2) / self.duration
        self._half_period = 0.5 * self.period
        self._ref_time = self.t0 - self._half_period

    def get_star_position(self, t, light_delay=False):
        z = torch.zeros_like(as_tensor(t, self.period))
        return z, z, z

    def get_planet_position(self, t, light_delay=False):
        return self.get_relative_position(t, light_delay=False)

    def _dt(self, t):
        t = as_tensor(t, self.period)
        return torch.remainder(t.unsqueeze(-1) - self._ref_time, self.period) - self._half_period

    def get_relative_position(self, t, light_delay=False):
        if light_delay:
            raise NotImplementedError("Light travel time delay is not implemented for simple orbits")
        dt = self._dt(t)
        x = (self.speed * dt).squeeze()
        y = (self._b_norm + torch.zeros_like(dt)).squeeze()
        m = dt.abs() < 0.5 * self.duration
        z = (m.to(torch.float64) * 2.0 - 1.0).squeeze()
        return x, y, z

    def get_planet_velocity(self, t):
        raise NotImplementedError("a SimpleTransitOrbit has no velocity")

    def get_star_velocity(self, t):
        raise NotImplementedError("a SimpleTransitOrbit has no velocity")

    def get_radial_velocity(self, t, output_units=None):
        raise NotImplementedError("a SimpleTransitOrbit has no velocity")

    def in_transit(self, t, r=None, texp=None, light_delay=False):
        if light_delay:
            raise NotImplementedError("Light travel time delay is not implemented for simple orbits")
        t = as_tensor(t, self.period)
        dt = self._dt(t)
        if r is None:
            tol = 0.5 * self.duration
        else:
            x = (_vec(r, self.period) + self.r_star) ** 2 - self._b_norm ** 2
            tol = torch.sqrt(x) / self.speed
        if texp is not None:
            tol = tol + 0.5 * as_tensor(texp, self.period)
        mask = (dt.abs() < tol).any(dim=-1)
        return torch.arange(t.numel(), device=t.device)[mask]
