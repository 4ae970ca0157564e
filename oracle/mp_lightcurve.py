"""End-to-end light curves in arbitrary precision (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

The whole chain of `LimbDarkLightCurve.get_light_curve` / `SecondaryEclipseLightCurve.get_light_curve`
evaluated with mpmath from the reference's own formulas -- written here directly from the reference
lines cited below, NOT through oracle/numpy_port.py, so that a fixture generated from this module pins
the numpy port, the C port and the device kernels to something none of them shares code with:

  orbit algebra     src/exoplanet/orbits/keplerian.py:146 (n), :205-214 (E0, M0, incl_factor),
                    :217-228 (cos i from b), :267-281 (t_periastron, sin i), :849-934 (Kepler's third law)
  positions         keplerian.py:283-334 (rotation, true anomaly), :380-409 (r), :517-542 (relative position)
  flipped orbit     keplerian.py:779-804
  exposure stencil  src/exoplanet/light_curves/limb_dark.py:178-226
  flux              limb_dark.py:11-24, :234-252;  blend: light_curves/secondary_eclipse.py:45-70
  Kepler's equation, solution vector: the DEFINITIONS of oracle/mp_reference.py (quadrature).
"""
import mpmath as mp

from . import mp_reference as R

G_GRAV = mp.mpf("2942.2062175044193")     # R_sun^3 / M_sun / day^2, orbits/constants.py:32


class Orbit:
    """standard transit parameterisation: period, t0, b, (ecc, omega), m_star, r_star, m_planet"""

    def __init__(self, period, t0=0.0, b=0.0, ecc=None, omega=None, m_star=1.0, r_star=1.0, m_planet=0.0,
                 t_periastron=None, incl=None):
        f = mp.mpf
        self.period, self.m_star, self.r_star, self.m_planet = f(period), f(m_star), f(r_star), f(m_planet)
        self.a = (G_GRAV * (self.m_star + self.m_planet) * self.period ** 2 / (4 * mp.pi ** 2)) ** (mp.mpf(1) / 3)   # :925-928
        self.n = 2 * mp.pi / self.period                                                                       # :146
        self.ecc = None if ecc is None else f(ecc)
        if self.ecc is None:
            self.M0 = mp.pi / 2                                                                                # :184
            self.cw, self.sw = mp.mpf(1), mp.mpf(0)
            incl_factor = mp.mpf(1)
        else:
            w = f(omega)
            self.omega = w
            self.cw, self.sw = mp.cos(w), mp.sin(w)
            E0 = 2 * mp.atan2(mp.sqrt(1 - self.ecc) * self.cw, mp.sqrt(1 + self.ecc) * (1 + self.sw))        # :205-209
            self.M0 = E0 - self.ecc * mp.sin(E0)                                                               # :210
            incl_factor = (1 + self.ecc * self.sw) / (1 - self.ecc ** 2)                                       # :212-214
        if incl is not None:
            self.cos_incl = mp.cos(f(incl))
            self.sin_incl = mp.sin(f(incl))
        else:
            self.cos_incl = incl_factor * self.r_star / self.a * f(b)                                          # :217-228
            self.sin_incl = mp.sin(mp.acos(self.cos_incl))                                                     # :281
        if t_periastron is not None:
            self.t_periastron = f(t_periastron)
            self.t0 = self.t_periastron + self.M0 / self.n
        else:
            self.t0 = f(t0)
            self.t_periastron = self.t0 - self.M0 / self.n                                                     # :277

    def relative_position(self, t):
        """(X, Y, Z) of the planet relative to the star (a -> -a, keplerian.py:540)"""
        M = (mp.mpf(t) - self.t_periastron) * self.n                                                          # :324-330
        if self.ecc is None:
            sinf, cosf = mp.sin(M), mp.cos(M)
            r = -self.a
        else:
            sinf, cosf = R.kepler(M, self.ecc)[:2]
            r = -self.a * (1 - self.ecc ** 2) / (1 + self.ecc * cosf)                                          # :403
        x, y = r * cosf, r * sinf
        x1 = self.cw * x - self.sw * y                                                                         # :303-308
        y1 = self.sw * x + self.cw * y
        return x1, self.cos_incl * y1, -self.sin_incl * y1                                                     # :311-314

    def flip(self, r_planet):
        """keplerian.py:779-804"""
        if self.ecc is None:
            o = Orbit(self.period, m_star=self.m_planet, m_planet=self.m_star, r_star=r_planet,
                      t_periastron=self.t_periastron + self.period / 2, incl=mp.acos(self.cos_incl))
        else:
            o = Orbit(self.period, ecc=self.ecc, omega=self.omega - mp.pi, m_star=self.m_planet, m_planet=self.m_star,
                      r_star=r_planet, t_periastron=self.t_periastron, incl=mp.acos(self.cos_incl))
        return o


def get_cl(u1, u2):
    """limb_dark.py:11-18"""
    u1, u2 = mp.mpf(u1), mp.mpf(u2)
    c = [1 - u1 - mp.mpf(3) / 2 * u2, u1 + 2 * u2, -u2 / 4]
    norm = mp.pi * (c[0] + c[1] / mp.mpf("1.5"))
    return [v / norm for v in c]


def stencil(oversample=7, order=0):
    """limb_dark.py:181-197 -> (offsets in units of texp, weights)"""
    oversample = int(oversample)
    oversample += 1 - oversample % 2
    w = [mp.mpf(1)] * oversample
    if order == 0:
        full = [mp.mpf(-1) / 2 + mp.mpf(k) / (2 * oversample) for k in range(2 * oversample + 1)]
        dt = full[1:-1:2]
    elif order == 1:
        dt = [mp.mpf(-1) / 2 + mp.mpf(k) / (oversample - 1) for k in range(oversample)]
        for k in range(1, oversample - 1):
            w[k] = mp.mpf(2)
    elif order == 2:
        dt = [mp.mpf(-1) / 2 + mp.mpf(k) / (oversample - 1) for k in range(oversample)]
        for k in range(1, oversample - 1, 2):
            w[k] = mp.mpf(4)
        for k in range(2, oversample - 1, 2):
            w[k] = mp.mpf(2)
    else:
        raise ValueError("order must be <= 2")
    tot = sum(w)
    return dt, [v / tot for v in w]


def flux_one(orbit, r, c, t):
    """limb_dark.py:215-252 for one planet, one time"""
    X, Y, Z = orbit.relative_position(t)
    if not (Z > 0):
        return mp.mpf(0)
    b = mp.sqrt(X * X + Y * Y) / orbit.r_star
    ror = mp.mpf(r) / orbit.r_star
    if b >= 1 + ror:
        return mp.mpf(0)
    s = R.quad_sv(b, ror)
    return s[0] * c[0] + s[1] * c[1] + s[2] * c[2] - 1


def light_curve(orbit, r, u, times, texp=None, oversample=7, order=0):
    c = get_cl(*u)
    if texp is None:
        return [flux_one(orbit, r, c, t) for t in times]
    dt, w = stencil(oversample, order)
    texp = mp.mpf(texp)
    return [sum(wk * flux_one(orbit, r, c, mp.mpf(t) + texp * dk) for dk, wk in zip(dt, w)) for t in times]


def secondary_light_curve(orbit, r, u_primary, u_secondary, sbr, times, texp=None, oversample=7, order=0):
    """secondary_eclipse.py:45-70"""
    lc1 = light_curve(orbit, r, u_primary, times, texp, oversample, order)
    o2 = orbit.flip(mp.mpf(r))
    lc2 = light_curve(o2, orbit.r_star, u_secondary, times, texp, oversample, order)
    k = mp.mpf(r) / orbit.r_star
    fr = mp.mpf(sbr) * k * k
    return [(a + fr * b) / (1 + fr) for a, b in zip(lc1, lc2)]
