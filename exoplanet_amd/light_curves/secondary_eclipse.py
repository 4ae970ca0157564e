"""Transit + occultation light curve.

Mirror of ``exoplanet.light_curves.SecondaryEclipseLightCurve``
(/root/reference/src/exoplanet/light_curves/secondary_eclipse.py:8-70).  The
reference evaluates two full light curves (``orbit`` and ``orbit._flip(r)``)
and blends them; for a KeplerianOrbit the two share their Kepler solve (the
flipped orbit is the same ellipse seen from the planet: omega - pi, or half a
period later when circular), so the fused kernel does both in one pass.
"""
import torch

from .limb_dark import LimbDarkLightCurve, exposure_stencil
from ..orbits.keplerian import KeplerianOrbit, _vec, as_tensor

__all__ = ["SecondaryEclipseLightCurve"]


class SecondaryEclipseLightCurve:
    """Args:
        u_primary, u_secondary: (u1, u2) of the star and of the companion.
        surface_brightness_ratio: companion / star surface brightness.
    """

    def __init__(self, u_primary, u_secondary, surface_brightness_ratio, model=None):
        self.primary = LimbDarkLightCurve(u_primary[0], u_primary[1])
        self.secondary = LimbDarkLightCurve(u_secondary[0], u_secondary[1])
        self.surface_brightness_ratio = as_tensor(surface_brightness_ratio)

    def get_light_curve(self, orbit=None, r=None, t=None, texp=None, oversample=7, order=0,
                        use_in_transit=None, light_delay=False, total=False, cadence_major=False, sparse=False):
        # (sparse: accepted for symmetry with LimbDarkLightCurve.get_light_curve -- transits and occultations are two lists per
        # draw, which the sparse model does not take yet: the cadence-major dense array is returned)
        if orbit is None:
            raise ValueError("missing required argument 'orbit'")
        if r is None:
            raise ValueError("missing required argument 'r'")
        if t is None:
            raise ValueError("missing required argument 't'")
        fused = (isinstance(orbit, KeplerianOrbit) and type(orbit)._warp_times is KeplerianOrbit._warp_times
                 and (not light_delay or self.primary._fusable_delay(orbit, t, texp)))
        if fused:
            use_in_transit = (not light_delay) if use_in_transit is None else use_in_transit
            if light_delay and use_in_transit:
                raise NotImplementedError("Light travel time delay not yet implemented for `in_transit`")
            stencil = exposure_stencil(oversample, order) if texp is not None else None
            return self.primary._fused(orbit, r, t, texp, stencil, use_in_transit,
                                       secondary=(self.secondary, self.surface_brightness_ratio),
                                       light_delay=light_delay, total=total, cadence_major=cadence_major, sparse=sparse)
        # composed path: exactly the reference's two-orbit blend (secondary_eclipse.py:45-70)
        r = _vec(r)
        orbit2 = orbit._flip(r)
        kw = dict(t=t, texp=texp, oversample=oversample, order=order, use_in_transit=use_in_transit,
                  light_delay=light_delay)
        lc1 = self.primary.get_light_curve(orbit=orbit, r=r, **kw)
        lc2 = self.secondary.get_light_curve(orbit=orbit2, r=orbit.r_star, **kw)
        k = r / orbit.r_star
        flux_ratio = self.surface_brightness_ratio.to(k.device) * k ** 2
        lc = (lc1 + flux_ratio * lc2) / (1 + flux_ratio)
        return lc.sum(-1) if total else lc
