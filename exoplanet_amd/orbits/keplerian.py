"""KeplerianOrbit on torch tensors, batched over posterior draws.

Host-side mirror of the reference's ``exoplanet.orbits.KeplerianOrbit``
(/root/reference/src/exoplanet/orbits/keplerian.py): same constructor
arguments, same derived attributes, same method names and error behaviour, so
a model written against the reference reads the same here.  What differs is
the execution model:

* every parameter may carry leading *draw* dimensions: shape ``(P,)`` as in the
  reference, or ``(D, P)`` for D posterior draws / chains evaluated at once;
* the O(P) scalar algebra below runs in torch (so autograd covers every
  parameterisation), while all O(N) work goes through the HIP ops in
  ``exoplanet_amd.ops`` -- the light-curve classes do not call
  ``get_relative_position`` for a KeplerianOrbit but hand ``kernel_records()``
  to the fused kernel.

Units: R_sun, M_sun, days; ``rho_star`` in g/cm^3 (keplerian.py:29-33).  The
astropy unit helpers of the reference are out of scope.
"""
import math
import warnings

import torch

from .. import ops
from .constants import G_grav, au_per_R_sun, c_light, gcc_per_sun, m_per_s_per_Rsun_per_day

__all__ = ["KeplerianOrbit", "get_true_anomaly", "get_aor_from_transit_duration"]

_TWO_PI = 2.0 * math.pi


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


_SCALARS = {}


def _scalar(value, dev):
    """0-dim constant, one per (value, device), never written to.  Made by a fill kernel rather
    than a host-to-device copy (legal while a hipGraph is capturing); cached so that a replayed
    step does not spend a launch per Python scalar -- unless the first request comes during a
    capture, whose allocations belong to the graph."""
    key = (value, dev)
    t = _SCALARS.get(key)
    if t is None:
        t = torch.full((), value, dtype=torch.float64, device=dev)
        if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _SCALARS[key] = t
    return t


def as_tensor(x, like=None):
    """float64 tensor (the reference casts everything to float64, utils.py:18-22)."""
    if isinstance(x, torch.Tensor):
        return x if x.dtype == torch.float64 else x.to(torch.float64)
    dev = like.device if isinstance(like, torch.Tensor) else _default_device()
    if isinstance(x, (int, float)):
        return _scalar(float(x), dev)
    return torch.as_tensor(x, dtype=torch.float64, device=dev)


def _vec(x, like=None):
    """at least 1-D: trailing axis = planets"""
    x = as_tensor(x, like)
    return x.reshape(1) if x.dim() == 0 else x


class KeplerianOrbit:
    """A system of bodies on Keplerian orbits around a common central body.

    Arguments and their allowed combinations follow the reference
    (keplerian.py:35-69): one of ``period`` / ``a`` (both only without
    ``m_star`` / ``rho_star``); one of ``incl`` / ``b`` / ``duration``;
    ``ecc`` needs ``omega`` (or ``sin_omega`` and ``cos_omega``); at most two of
    ``m_star, r_star, rho_star``; one of ``t0`` / ``t_periastron``.
    """

    # attributes computed lazily from the constructor arguments (first access runs the algebra once)
    _DERIVED = frozenset((
        "a", "period", "rho_star", "r_star", "m_star", "m_planet", "m_total", "n", "a_star", "a_planet", "K0",
        "Omega", "cos_Omega", "sin_Omega", "ecc", "omega", "cos_omega", "sin_omega", "M0", "dcosidb", "b",
        "cos_incl", "incl", "sin_incl", "duration", "t0", "t_periastron", "tref", "jacobians"))

    def __init__(self, period=None, a=None, t0=None, t_periastron=None, incl=None, b=None, duration=None,
                 ecc=None, omega=None, sin_omega=None, cos_omega=None, Omega=None, m_planet=0.0,
                 m_star=None, r_star=None, rho_star=None, ror=None, **kwargs):
        if kwargs:
            raise TypeError(f"unsupported arguments {sorted(kwargs)} (astropy units are not handled here)")
        self._args = dict(period=period, a=a, t0=t0, t_periastron=t_periastron, incl=incl, b=b, duration=duration,
                          ecc=ecc, omega=omega, sin_omega=sin_omega, cos_omega=cos_omega, Omega=Omega,
                          m_planet=m_planet, m_star=m_star, r_star=r_star, rho_star=rho_star, ror=ror)
        self._ready = False
        self._validate()
        # the standard transit parameterisation goes through the fused packing kernel and
        # never needs the attribute algebra; everything else materialises on first access
        A = self._args
        self._standard = (
            period is not None and a is None and t_periastron is None and incl is None and duration is None
            and rho_star is None and sin_omega is None and cos_omega is None
            and (ecc is None) == (omega is None) and (m_star is None) == (r_star is None))

    def _validate(self):
        """argument-combination errors of the reference constructor, without any tensor work
        (keplerian.py:116-120,189-203,222-232,267-268,850-902)"""
        A = self._args
        a, period, rho_star, r_star, m_star = A["a"], A["period"], A["rho_star"], A["r_star"], A["m_star"]
        duration_circular = A["ecc"] is None and A["duration"] is not None
        if duration_circular:
            if A["b"] is None:
                raise ValueError("'b' must be provided for a circular orbit with a 'duration'")
            if A["ror"] is None:
                warnings.warn("When using the 'duration' parameter in KeplerianOrbit, the 'ror' parameter "
                              "should also be provided.", UserWarning)
            a = True  # a is implied by the duration
            if r_star is None:
                r_star = 1.0
        if a is None and period is None:
            raise ValueError("values must be provided for at least one of a and period")
        implied = False
        if a is not None and period is not None:
            if rho_star is not None or m_star is not None:
                raise ValueError("if both a and period are given, you can't also define rho_star or m_star")
            implied = True
        if r_star is None and m_star is None:
            r_star = 1.0
            if rho_star is None:
                m_star = 1.0
        if (not implied) and sum(x is None for x in (rho_star, r_star, m_star)) != 1:
            raise ValueError("values must be provided for exactly two of rho_star, m_star, and r_star")
        if A["ecc"] is not None:
            if A["omega"] is not None:
                if A["sin_omega"] is not None and A["cos_omega"] is not None:
                    raise ValueError("either 'omega' or 'sin_omega' and 'cos_omega' can be provided")
            elif not (A["sin_omega"] is not None and A["cos_omega"] is not None):
                raise ValueError("both e and omega must be provided")
        dur = None if duration_circular else A["duration"]
        if A["b"] is not None and (A["incl"] is not None or dur is not None):
            raise ValueError("only one of 'incl', 'b', and 'duration' can be given")
        if A["incl"] is not None and dur is not None:
            raise ValueError("only one of 'incl', 'b', and 'duration' can be given")
        if A["t0"] is not None and A["t_periastron"] is not None:
            raise ValueError("you can't define both t0 and t_periastron")

    def __getattr__(self, name):
        # only reached when normal lookup fails: derived attributes before materialisation
        if name in KeplerianOrbit._DERIVED and not self.__dict__.get("_ready", True):
            self._materialize()
            return self.__dict__[name] if name in self.__dict__ else object.__getattribute__(self, name)
        raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")

    def _materialize(self):
        """the parameter algebra of the reference constructor (keplerian.py:110-281)"""
        A = self._args
        period, a, t0, t_periastron = A["period"], A["a"], A["t0"], A["t_periastron"]
        incl, b, duration, ecc, omega = A["incl"], A["b"], A["duration"], A["ecc"], A["omega"]
        sin_omega, cos_omega, Omega = A["sin_omega"], A["cos_omega"], A["Omega"]
        m_planet, m_star, r_star, rho_star, ror = A["m_planet"], A["m_star"], A["r_star"], A["rho_star"], A["ror"]
        self._ready = True
        self.jacobians = {}
        like = next((x for x in (period, a, t0, t_periastron, b, incl, ecc, omega, m_star, r_star, rho_star, m_planet)
                     if isinstance(x, torch.Tensor)), None)
        T = lambda x: None if x is None else _vec(x, like)  # noqa: E731

        # -- circular orbit parameterised by the transit duration (keplerian.py:112-131)
        daordtau = None
        if ecc is None and duration is not None:
            if r_star is None:
                r_star = 1.0
            aor, daordtau = get_aor_from_transit_duration(T(duration), T(period), T(b), ror=T(ror))
            a = T(r_star) * aor
            duration = None

        (self.a, self.period, self.rho_star, self.r_star, self.m_star, self.m_planet) = _consistent_inputs(
            T(a), T(period), T(rho_star), T(r_star), T(m_star), T(m_planet))
        self.m_total = self.m_star + self.m_planet
        self.n = _TWO_PI / self.period
        self.a_star = self.a * self.m_planet / self.m_total
        self.a_planet = -self.a * self.m_star / self.m_total

        if daordtau is not None:  # keplerian.py:151-170
            dadtau = self.r_star * daordtau
            self.jacobians["duration"] = {
                "a": dadtau,
                "a_star": dadtau * self.m_planet / self.m_total,
                "a_planet": -dadtau * self.m_star / self.m_total,
                "rho_star": 9 * math.pi * (self.a / self.r_star) ** 2 * daordtau * gcc_per_sun
                / (G_grav * self.period ** 2),
            }

        self.K0 = self.n * self.a / self.m_total
        self.Omega = T(Omega)
        if self.Omega is not None:
            self.cos_Omega, self.sin_Omega = torch.cos(self.Omega), torch.sin(self.Omega)

        if ecc is None:  # keplerian.py:182-185
            self.ecc = None
            self.M0 = 0.5 * math.pi + torch.zeros_like(self.n)
            incl_factor = 1.0
        else:
            self.ecc = T(ecc)
            if omega is not None:
                if sin_omega is not None and cos_omega is not None:
                    raise ValueError("either 'omega' or 'sin_omega' and 'cos_omega' can be provided")
                self.omega = T(omega)
                self.cos_omega, self.sin_omega = torch.cos(self.omega), torch.sin(self.omega)
            elif sin_omega is not None and cos_omega is not None:
                self.cos_omega, self.sin_omega = T(cos_omega), T(sin_omega)
                self.omega = torch.atan2(self.sin_omega, self.cos_omega)
            else:
                raise ValueError("both e and omega must be provided")
            # eccentric anomaly of mid-transit, f = pi/2 - omega (keplerian.py:205-210)
            E0 = 2 * torch.atan2(torch.sqrt(1 - self.ecc) * self.cos_omega,
                                 torch.sqrt(1 + self.ecc) * (1 + self.sin_omega))
            self.M0 = E0 - self.ecc * torch.sin(E0)
            ome2 = 1 - self.ecc ** 2
            self.K0 = self.K0 / torch.sqrt(ome2)
            incl_factor = (1 + self.ecc * self.sin_omega) / ome2

        # d cos(i) / d b  (keplerian.py:216-219)
        self.dcosidb = incl_factor * self.r_star / self.a
        self.jacobians["b"] = {"cos_incl": self.dcosidb}

        if b is not None:
            if incl is not None or duration is not None:
                raise ValueError("only one of 'incl', 'b', and 'duration' can be given")
            self.b = T(b) + torch.zeros_like(self.a)
            self.cos_incl = self.dcosidb * self.b
            self.incl = torch.acos(self.cos_incl)
        elif incl is not None:
            if duration is not None:
                raise ValueError("only one of 'incl', 'b', and 'duration' can be given")
            self.incl = T(incl) + torch.zeros_like(self.a)
            self.cos_incl = torch.cos(self.incl)
            self.b = self.cos_incl / self.dcosidb
        elif duration is not None:
            # eccentric orbit from its duration (keplerian.py:237-260)
            self.duration = T(duration)
            c = torch.sin(math.pi * self.duration * incl_factor / self.period)
            c2 = c * c
            aor = self.a_planet / self.r_star
            esinw = self.ecc * self.sin_omega
            self.b = torch.sqrt((aor ** 2 * c2 - 1)
                                / (c2 * esinw ** 2 + 2 * c2 * esinw + c2 - self.ecc ** 4 + 2 * self.ecc ** 2 - 1))
            self.b = self.b * (1 - self.ecc ** 2)
            self.cos_incl = self.dcosidb * self.b
            self.incl = torch.acos(self.cos_incl)
        else:
            zla = torch.zeros_like(self.a)
            self.incl = 0.5 * math.pi + zla
            self.cos_incl = zla
            self.b = zla

        if t0 is not None and t_periastron is not None:
            raise ValueError("you can't define both t0 and t_periastron")
        if t0 is None and t_periastron is None:
            t0 = torch.zeros_like(self.period)
        if t0 is None:
            self.t_periastron = T(t_periastron) + torch.zeros_like(self.period)
            self.t0 = self.t_periastron + self.M0 / self.n
        else:
            self.t0 = T(t0) + torch.zeros_like(self.period)
            self.t_periastron = self.t0 - self.M0 / self.n
        self.tref = self.t_periastron - self.t0
        self.sin_incl = torch.sin(self.incl)
        if "duration" not in self.__dict__:
            self.duration = None
        if self.ecc is None:
            self.omega = self.cos_omega = self.sin_omega = None
        if self.Omega is None:
            self.cos_Omega = self.sin_Omega = None

    # ------------------------------------------------------------------ shapes
    @property
    def shape(self):
        """broadcast shape of all per-planet parameters: (..., P)"""
        return torch.broadcast_shapes(self.a.shape, self.period.shape, self.t0.shape, self.cos_incl.shape,
                                      self.M0.shape, self.r_star.shape)

    def _ew(self):
        """(ecc, cos_omega, sin_omega) with the circular convention e=0, omega=0"""
        if self.ecc is None:
            z = torch.zeros_like(self.n)
            return z, z + 1.0, z
        return self.ecc, self.cos_omega, self.sin_omega

    # ------------------------------------------------------------------ anomaly / frames
    def _rotate_vector(self, x, y):
        """orbital plane -> observer frame (keplerian.py:283-322).  ``x, y`` have
        shape (..., N, P) and the per-planet constants broadcast as (..., 1, P)."""
        u = lambda v: v.unsqueeze(-2)  # noqa: E731
        if self.ecc is None:
            x1, y1 = x, y
        else:
            x1 = u(self.cos_omega) * x - u(self.sin_omega) * y
            y1 = u(self.sin_omega) * x + u(self.cos_omega) * y
        x2 = x1
        y2 = u(self.cos_incl) * y1
        Z = -u(self.sin_incl) * y1
        if self.Omega is None:
            return x2, y2, Z
        X = u(self.cos_Omega) * x2 - u(self.sin_Omega) * y2
        Y = u(self.sin_Omega) * x2 + u(self.cos_Omega) * y2
        return X, Y, Z

    def _warp_times(self, t, _pad=True):
        """time since the reference transit; overridden by TTV-type orbits (keplerian.py:324-327)"""
        t = as_tensor(t, self.n)
        if _pad:
            return t.unsqueeze(-1) - self.t0.unsqueeze(-2)
        return t - self.t0.unsqueeze(-2)

    def _get_true_anomaly(self, t, _pad=True):
        """(sin f, cos f), shape (..., N, P) (keplerian.py:329-334)"""
        M = (self._warp_times(t, _pad=_pad) - self.tref.unsqueeze(-2)) * self.n.unsqueeze(-2)
        if self.ecc is None:
            return torch.sin(M), torch.cos(M)
        e = self.ecc.unsqueeze(-2) + torch.zeros_like(M)
        return ops.kepler(M.contiguous(), e.contiguous())

    def _fused_vector(self, amp, t, velocity, acceleration=False):
        """(X, Y, Z) through ops.orbit_vector -- one launch each way instead of the solve, the radius, three
        rotations and their broadcasts as ~40 launch-bound torch kernels -- or None when the times are not a
        1-D device tensor / the orbit warps its times (TTV-type subclasses take the composed path)."""
        if type(self)._warp_times is not KeplerianOrbit._warp_times:
            return None
        if not (isinstance(t, torch.Tensor) and t.dim() == 1 and t.is_cuda and t.dtype == torch.float64):
            return None
        if t.requires_grad:      # the op has no cotangent for the times (keplerian_test.py:91-131 differentiates them)
            return None
        e, cw, sw = self._ew()
        one, zero = torch.ones_like(self.n), torch.zeros_like(self.n)
        cO, sO = (one, zero) if self.Omega is None else (self.cos_Omega, self.sin_Omega)
        cols = torch.broadcast_tensors(self.n, self.t_periastron, e, cw, sw, self.cos_incl, self.sin_incl, amp, cO, sO)
        shape = cols[0].shape
        params = torch.stack(cols, dim=-1).reshape(-1, shape[-1], ops.OV_NPAR).contiguous()
        out = ops.orbit_vector(t, params, velocity=velocity, acceleration=acceleration)
        out = out.reshape(tuple(shape[:-1]) + (t.shape[0], shape[-1], 3))
        return out[..., 0], out[..., 1], out[..., 2]

    def _get_position(self, a, t, parallax=None, light_delay=False, _pad=True):
        if light_delay:
            return self._get_retarded_position(a, t, parallax=None, _pad=_pad)
        if _pad:
            amp = a if parallax is None else a * parallax * au_per_R_sun
            fused = self._fused_vector(amp, t, velocity=False)
            if fused is not None:
                return fused
        sinf, cosf = self._get_true_anomaly(t, _pad=_pad)
        a = a.unsqueeze(-2)
        if self.ecc is None:
            r = a
        else:
            e = self.ecc.unsqueeze(-2)
            r = a * (1.0 - e ** 2) / (1 + e * cosf)
        if parallax is not None:
            r = r * parallax * au_per_R_sun
        return self._rotate_vector(r * cosf, r * sinf)

    def _get_retarded_position(self, a, t, parallax=None, z0=0.0, _pad=True):
        """position at the retarded time (light travel delay), keplerian.py:411-470"""
        sinf, cosf = self._get_true_anomaly(t, _pad=_pad)
        a_ = a.unsqueeze(-2)
        angvel = (_TWO_PI / self.period).unsqueeze(-2)
        si = self.sin_incl.unsqueeze(-2)
        if self.ecc is None:
            r = a_ + torch.zeros_like(cosf)
            vz = angvel * a_ * si * cosf
        else:
            e = self.ecc.unsqueeze(-2)
            cw, sw = self.cos_omega.unsqueeze(-2), self.sin_omega.unsqueeze(-2)
            r = a_ * (1.0 - e ** 2) / (1 + e * cosf)
            vamp = angvel * a_ / torch.sqrt(1 - e ** 2)
            vz = vamp * si * (e * cw + cw * cosf - sw * sinf)
        x, y, z = self._rotate_vector(r * cosf, r * sinf)
        az = -(angvel ** 2) * (a_ / r) ** 3 * z
        small = az.abs() < 1.0e-10
        az_safe = torch.where(small, torch.ones_like(az), az)
        disc = (1 + vz / c_light) ** 2 - 2 * az_safe * (z0 - z) / c_light ** 2
        delay = torch.where(small, (z0 - z) / (c_light + vz),
                            (c_light / az_safe) * ((1 + vz / c_light) - torch.sqrt(disc)))
        tt = as_tensor(t, self.n)
        new_t = (tt.unsqueeze(-1) if _pad else tt) - delay
        return self._get_position(a, new_t, parallax, _pad=False)

    @staticmethod
    def _squeeze(xs):
        return tuple(x.squeeze() for x in xs)

    def get_planet_position(self, t, parallax=None, light_delay=False):
        return self._squeeze(self._get_position(self.a_planet, t, parallax, light_delay=light_delay))

    def get_star_position(self, t, parallax=None, light_delay=False):
        return self._squeeze(self._get_position(self.a_star, t, parallax, light_delay=light_delay))

    def get_relative_position(self, t, parallax=None, light_delay=False):
        """planet - star in the observer frame, each (N, P) squeezed (keplerian.py:517-542)"""
        return self._squeeze(self._get_position(-self.a, t, parallax, light_delay=light_delay))

    def get_relative_angles(self, t, parallax=None, light_delay=False):
        X, Y, _ = self._get_position(-self.a, t, parallax, light_delay=light_delay)
        return torch.sqrt(X ** 2 + Y ** 2).squeeze(), torch.atan2(Y, X).squeeze()

    # ------------------------------------------------------------------ velocities ("next" row f-3)
    def _get_velocity(self, m, t):
        fused = self._fused_vector(self.K0 * m, t, velocity=True)
        if fused is not None:
            return fused
        sinf, cosf = self._get_true_anomaly(t)
        K = (self.K0 * m).unsqueeze(-2)
        if self.ecc is None:
            return self._rotate_vector(-K * sinf, K * cosf)
        return self._rotate_vector(-K * sinf, K * (cosf + self.ecc.unsqueeze(-2)))

    def get_planet_velocity(self, t):
        return self._squeeze(self._get_velocity(-self.m_star, t))

    def get_star_velocity(self, t):
        return self._squeeze(self._get_velocity(self.m_planet, t))

    def get_relative_velocity(self, t):
        return self._squeeze(self._get_velocity(-self.m_total, t))

    def get_radial_velocity(self, t, K=None, output_units=None):
        """stellar reflex RV, positive = redshift (keplerian.py:633-677).  Without
        ``K`` the result is in m/s (``output_units`` other than None is not supported:
        no astropy)."""
        if K is None and output_units is not None:
            raise NotImplementedError("unit conversion needs astropy; the default is m/s")
        t = as_tensor(t, next((x for x in self._args.values() if isinstance(x, torch.Tensor)), None))
        if type(self)._warp_times is KeplerianOrbit._warp_times and t.dim() == 1 and t.is_cuda:
            # one fused launch (and one for the reverse pass): amplitude x (cos w cos f - sin w sin f + e cos w)
            # with the caller's K, or -- the z-velocity of the star written out (keplerian.py:572-578 through
            # :283-322) -- conv sin(incl) K0 m_planet
            if K is not None and self._standard and not self._ready:
                # standard parameterisation: (n, t_periastron, e, cos w, sin w) straight from the record-
                # packing kernel -- no attribute algebra for a K-parameterised RV model
                rec, _, batch, _ = self.kernel_inputs(0.0, (0.0, 0.0))
                amp = _vec(K, rec).expand(tuple(batch) + (rec.shape[1],)).reshape(-1, rec.shape[1], 1)
                # record slots N, TP, ECC, COSW, SINW are the first five, in the RV record's order (a slice:
                # an index list would be uploaded from the host, which a hipGraph capture does not allow)
                params = torch.cat([rec[..., ops.P_N:ops.P_SINW + 1], amp], dim=-1)
                rv = ops.radial_velocity(t, params)
                return rv.reshape(tuple(batch) + (t.shape[0], rec.shape[1])).squeeze()
            if K is not None:
                amp = _vec(K, self.n)
            else:
                amp = m_per_s_per_Rsun_per_day * self.sin_incl * self.K0 * self.m_planet
            e, cw, sw = self._ew()
            cols = torch.broadcast_tensors(self.n, self.t_periastron, e, cw, sw, amp)
            shape = cols[0].shape
            params = torch.stack(cols, dim=-1).reshape(-1, shape[-1], ops.RV_NPAR).contiguous()
            rv = ops.radial_velocity(t, params)
            return rv.reshape(tuple(shape[:-1]) + (t.shape[0], shape[-1])).squeeze()
        if K is not None:
            sinf, cosf = self._get_true_anomaly(t)
            K = _vec(K, self.n).unsqueeze(-2)
            if self.ecc is None:
                return (K * cosf).squeeze()
            cw, sw, e = self.cos_omega.unsqueeze(-2), self.sin_omega.unsqueeze(-2), self.ecc.unsqueeze(-2)
            return (K * (cw * cosf - sw * sinf + e * cw)).squeeze()
        return -m_per_s_per_Rsun_per_day * self.get_star_velocity(t)[2]

    def _get_acceleration(self, a, m, t):
        fused = self._fused_vector((self.K0 * m) ** 2 / a, t, velocity=False, acceleration=True)
        if fused is not None:
            return fused
        sinf, cosf = self._get_true_anomaly(t)
        K = (self.K0 * m).unsqueeze(-2)
        a = a.unsqueeze(-2)
        if self.ecc is None:
            factor = -(K ** 2) / a
        else:
            e = self.ecc.unsqueeze(-2)
            factor = K ** 2 * (e * cosf + 1) ** 2 / (a * (e ** 2 - 1))
        return self._rotate_vector(factor * cosf, factor * sinf)

    def get_planet_acceleration(self, t):
        return self._squeeze(self._get_acceleration(self.a_planet, -self.m_star, t))

    def get_star_acceleration(self, t):
        return self._squeeze(self._get_acceleration(self.a_star, self.m_planet, t))

    def get_relative_acceleration(self, t):
        return self._squeeze(self._get_acceleration(-self.a, -self.m_total, t))

    # ------------------------------------------------------------------ transit windows
    def _transit_window(self, r, flip=False):
        """(t_start, t_end, flag): first / fourth contact relative to mid-transit,
        per planet, without exposure padding (keplerian.py:733-763).  No gradient."""
        with torch.no_grad():
            z = torch.zeros(self.shape, dtype=torch.float64, device=self.a.device)
            r = _vec(r, self.a).detach() + z
            R = self.r_star.detach() + z
            hp = 0.5 * self.period.detach() + z
            if self.ecc is None:
                k = r / R
                arg = (1 + k) ** 2 - self.b.detach() ** 2
                factor = R / (self.a.detach() * self.sin_incl.detach())
                hdur = hp * torch.asin(factor * torch.sqrt(arg)) / math.pi
                return -hdur, hdur, torch.zeros_like(z, dtype=torch.int32)
            P = self.period.detach() + z
            Ml, Mr, flag = ops.contact_points(
                (self.a.detach() + z).contiguous(), (self.ecc.detach() + z).contiguous(),
                (self.cos_omega.detach() + z).contiguous(), (self.sin_omega.detach() + z).contiguous(),
                (self.cos_incl.detach() + z).contiguous(), (self.sin_incl.detach() + z).contiguous(),
                (R + r).contiguous())
            M0, n = self.M0.detach(), self.n.detach()
            ts = torch.remainder((Ml - M0) / n + hp, P) - hp
            te = torch.remainder((Mr - M0) / n + hp, P) - hp
            ts = torch.where(ts > 0, ts - P, ts)
            te = torch.where(te < 0, te + P, te)
            return ts, te, flag

    def in_transit(self, t, r=0.0, texp=None, light_delay=False):
        """indices of the cadences between first and fourth contact of any planet
        (keplerian.py:708-777).  Unbatched orbits only (an index list is ragged
        across draws; the fused kernel applies the same window per draw instead)."""
        if light_delay:
            raise NotImplementedError("Light travel time delay not yet implemented for `in_transit`")
        if len(self.shape) != 1:
            raise ValueError("in_transit() returns a ragged index list: use it with unbatched parameters")
        t = as_tensor(t, self.n)
        ts, te, flag = self._transit_window(r)
        hp = 0.5 * self.period.detach()
        dt = torch.remainder(self._warp_times(t.detach()).detach() + hp, self.period.detach()) - hp
        if texp is not None:
            texp = as_tensor(texp, self.n).detach()
            h = 0.5 * (texp.unsqueeze(-1) if texp.dim() else texp)
            ts, te = ts - h, te + h
        mask = ((dt >= ts) & (dt <= te)).any(dim=-1)
        idx = torch.arange(t.shape[0], device=t.device)
        return idx[mask] if bool((flag == 0).all()) else idx

    def _flip(self, r_planet):
        """the orbit of the star around the planet (secondary eclipses), keplerian.py:779-804"""
        if self.ecc is None:
            return type(self)(period=self.period, t_periastron=self.t_periastron + 0.5 * self.period,
                              incl=self.incl, Omega=self.Omega, m_star=self.m_planet, m_planet=self.m_star,
                              r_star=r_planet)
        return type(self)(period=self.period, t_periastron=self.t_periastron, incl=self.incl, ecc=self.ecc,
                          omega=self.omega - math.pi, Omega=self.Omega, m_star=self.m_planet,
                          m_planet=self.m_star, r_star=r_planet)

    # ------------------------------------------------------------------ fused-kernel records
    def kernel_inputs(self, r, u, use_in_transit=False, secondary=None, light_delay=False):
        """Everything the fused light-curve kernels need, differentiable w.r.t. every
        orbit / limb-darkening parameter: ``(records (D,P,20), ld (D,3|6), batch_shape, flags)``.
        ``light_delay``: the kernels evaluate every sample at its retarded time (keplerian.py:411-470).

        ``u = (u1, u2)``; ``secondary = ((u1s, u2s), sbr)`` adds the occultation.  For the
        standard transit parameterisation (period, t0, b[, ecc, omega], r, [m_star, r_star],
        m_planet) this is ONE packing kernel (ops.pack_records); any other
        parameterisation runs the attribute algebra in torch."""
        flags = (ops.FLAG_WINDOW if use_in_transit else 0) | (ops.FLAG_SECONDARY if secondary is not None else 0)
        flags |= ops.FLAG_LIGHT_DELAY if light_delay else 0
        if self._standard:
            A = self._args
            like = next((x for x in list(A.values()) + [r] if isinstance(x, torch.Tensor)), None)
            V = lambda x, default: _vec(default if x is None else x, like)  # noqa: E731
            circular = A["ecc"] is None
            pack_flags = (flags & ~ops.FLAG_LIGHT_DELAY) | (ops.PACK_CIRCULAR if circular else 0)
            fast = self._kernel_inputs_cols(r, u, secondary, pack_flags, like)
            if fast is not None:
                return fast + (flags,)
            # the surface-brightness ratio is a per-light-curve scalar (like u): a 1-D value is per draw
            sbr = as_tensor(secondary[1] if secondary is not None else 0.0, like)
            sbr = sbr.unsqueeze(-1) if sbr.dim() >= 1 else sbr.reshape(1)
            cols = [V(A["period"], None), V(A["t0"], 0.0), V(A["b"], 0.0), V(A["ecc"], 0.0), V(A["omega"], 0.0),
                    V(r, None), V(A["m_star"], 1.0), V(A["r_star"], 1.0), V(A["m_planet"], 0.0), sbr]
            cols = torch.broadcast_tensors(*cols)
            shape = cols[0].shape
            orbit_in = torch.stack(cols, dim=-1).reshape(-1, shape[-1], ops.NIN)
            batch = tuple(shape[:-1])
            us = list(u) + (list(secondary[0]) if secondary is not None else [])
            us = [as_tensor(x, like) for x in us]
            ld_in = torch.stack(torch.broadcast_tensors(*us), dim=-1)
            if ld_in.dim() > 2:
                raise ValueError("limb-darkening coefficients may carry at most one draw dimension")
            ld_in = ld_in.expand(batch + (len(us),)).reshape(-1, len(us)) if batch else ld_in.reshape(1, len(us))
            if not batch and ld_in.shape[0] != 1:
                # draws only on the limb-darkening side: give the orbit the same batch
                orbit_in = orbit_in.expand(ld_in.shape[0], -1, -1)
                batch = (ld_in.shape[0],)
            rec, ld = ops.pack_records(orbit_in.contiguous(), ld_in.contiguous(),
                                       (flags & ~ops.FLAG_LIGHT_DELAY) | (ops.PACK_CIRCULAR if circular else 0))
            return rec, ld, batch, flags
        from ..light_curves.limb_dark import get_cl  # local import: light_curves imports this module

        rec, batch = self.kernel_records(r, use_in_transit=use_in_transit,
                                         secondary_sbr=None if secondary is None else secondary[1])
        c = get_cl(u[0], u[1])
        if secondary is not None:
            c = torch.cat(torch.broadcast_tensors(c, get_cl(secondary[0][0], secondary[0][1]).to(c.device)), dim=-1)
        D = rec.shape[0]
        if c.dim() == 2 and not batch:
            batch = (c.shape[0],)
            rec = rec.expand(c.shape[0], -1, -1)
            D = c.shape[0]
        ld = c.to(rec.device).expand(batch + (c.shape[-1],)).reshape(D, c.shape[-1])
        return rec.contiguous(), ld.contiguous(), batch, flags

    def _standard_cols(self, r, u, secondary, like):
        """the constructor arguments of the standard parameterisation as the packing kernel's columns -- ``(orbit
        columns, limb-darkening columns, draws, batched?)`` -- when they are ROCm float64 tensors with at most one draw
        dimension (then they go to the kernels as they are); None otherwise"""
        A = self._args
        opt = lambda x: None if x is None else _vec(x, like)  # noqa: E731
        sbr = None
        if secondary is not None:
            sbr = as_tensor(secondary[1], like)
            sbr = sbr.unsqueeze(-1) if sbr.dim() >= 1 else sbr.reshape(1)
        cols = [opt(A["period"]), opt(A["t0"]), opt(A["b"]), opt(A["ecc"]), opt(A["omega"]), opt(r), opt(A["m_star"]),
                opt(A["r_star"]), opt(A["m_planet"]), sbr]
        us = [as_tensor(x, like) for x in list(u) + (list(secondary[0]) if secondary is not None else [])]
        every = [c for c in cols if c is not None] + us
        if not all(x.is_cuda and x.dtype == torch.float64 for x in every):
            return None
        if any(c.dim() > 2 for c in cols if c is not None) or any(x.dim() > 1 for x in us):
            return None
        draws = {c.shape[0] for c in cols if c is not None and c.dim() == 2} | {x.shape[0] for x in us if x.dim() == 1}
        draws.discard(1)
        if len(draws) > 1:
            return None
        batched = any(c.dim() == 2 for c in cols if c is not None) or any(x.dim() == 1 for x in us)
        return cols, us, (draws.pop() if draws else 1), batched

    def _kernel_inputs_cols(self, r, u, secondary, pack_flags, like):
        """kernel_inputs of the standard parameterisation through ops.pack_records_cols (no stacking, one launch each
        way); None when that form does not apply"""
        got = self._standard_cols(r, u, secondary, like)
        if got is None:
            return None
        cols, us, D, batched = got
        rec, ld = ops.pack_records_cols(cols, us, D, pack_flags)
        return rec, ld, ((D,) if batched else ())

    def flux_dot(self, r, u, t, weights, use_in_transit=False, secondary=None, light_delay=False, texp=None,
                 stencil=None, sparse=False, events=(None, None)):
        """``(flux, L)`` with ``L[d] = sum_n weights[d, n] flux[d, n]``, L differentiable with respect to every
        orbit / limb-darkening parameter: value and gradient of a light-curve likelihood whose cotangent is known
        up front, in ONE sweep over the cadences (ops.transit_flux_dot).  For the standard parameterisation with
        at most one draw dimension the parameter tensors go to the kernels as they are (ops.orbit_flux_dot: no
        stacking pass, the cotangent of L folded into the packing VJP); otherwise kernel_inputs + transit_flux_dot.
        ``stencil = (dt, w)``: exposure-time integration (limb_dark.exposure_stencil)."""
        sdt, sw = (None, None) if stencil is None else stencil
        flags = (ops.FLAG_WINDOW if use_in_transit else 0) | (ops.FLAG_SECONDARY if secondary is not None else 0)
        flags |= (ops.FLAG_LIGHT_DELAY if light_delay else 0) | (ops.FLAG_SPARSE if sparse else 0)
        if self._standard:
            A = self._args
            like = next((x for x in list(A.values()) + [r] if isinstance(x, torch.Tensor)), None)
            opt = lambda x: None if x is None else _vec(x, like)  # noqa: E731
            sbr = None
            if secondary is not None:
                sbr = as_tensor(secondary[1], like)
                sbr = sbr.unsqueeze(-1) if sbr.dim() >= 1 else sbr.reshape(1)
            cols = [opt(A["period"]), opt(A["t0"]), opt(A["b"]), opt(A["ecc"]), opt(A["omega"]), opt(r), opt(A["m_star"]),
                    opt(A["r_star"]), opt(A["m_planet"]), sbr]
            us = [as_tensor(x, like) for x in list(u) + (list(secondary[0]) if secondary is not None else [])]
            if all(c is None or c.dim() <= 2 for c in cols) and all(x.dim() <= 1 for x in us):
                pack_flags = (flags & (ops.FLAG_WINDOW | ops.FLAG_SECONDARY)) | (ops.PACK_CIRCULAR if A["ecc"] is None else 0)
                return ops.orbit_flux_dot(t, weights, cols, us, flags=flags, pack_flags=pack_flags, texp=texp,
                                          stencil_dt=sdt, stencil_w=sw, events=events)
        rec, ld, _, fl = self.kernel_inputs(r, u, use_in_transit=use_in_transit, secondary=secondary, light_delay=light_delay)
        return ops.transit_flux_dot(t, rec, ld, weights, texp=texp, stencil_dt=sdt, stencil_w=sw,
                                    flags=fl | (ops.FLAG_SPARSE if sparse else 0), events=events)

    def flux_value_and_grad(self, r, u, t, weights, use_in_transit=False, secondary=None, light_delay=False, texp=None,
                            stencil=None, sparse=False, gscale=None, events=(None, None)):
        """``(flux, L, grads)`` -- :meth:`flux_dot` AND the gradient of ``sum_d gscale[d] L[d]`` (``gscale`` None: of
        ``sum_d L[d]``) in one call of the library, without autograd (ops.orbit_flux_value_and_grad: what a sampler's leapfrog
        step asks for).  ``grads``: a dict over the constructor arguments that were given as tensors (period, t0, b, ecc,
        omega, m_star, r_star, m_planet), ``r``, ``u1``, ``u2`` (``u1s``, ``u2s``, ``sbr`` with ``secondary``), each in its
        argument's own shape.  Standard parameterisation with at most one draw dimension (TypeError otherwise: use
        :meth:`flux_dot` and torch.autograd)."""
        sdt, sw = (None, None) if stencil is None else stencil
        flags = (ops.FLAG_WINDOW if use_in_transit else 0) | (ops.FLAG_SECONDARY if secondary is not None else 0)
        flags |= (ops.FLAG_LIGHT_DELAY if light_delay else 0) | (ops.FLAG_SPARSE if sparse else 0)
        if not self._standard:
            raise TypeError("flux_value_and_grad: the standard parameterisation (period, t0, b[, ecc, omega]) only")
        A = self._args
        like = next((x for x in list(A.values()) + [r] if isinstance(x, torch.Tensor)), None)
        opt = lambda x: None if x is None else _vec(x, like)  # noqa: E731
        sbr = None
        if secondary is not None:
            sbr = as_tensor(secondary[1], like)
            sbr = sbr.unsqueeze(-1) if sbr.dim() >= 1 else sbr.reshape(1)
        names = ["period", "t0", "b", "ecc", "omega", "r", "m_star", "r_star", "m_planet", "sbr"]
        given = [A["period"], A["t0"], A["b"], A["ecc"], A["omega"], r, A["m_star"], A["r_star"], A["m_planet"],
                 None if secondary is None else secondary[1]]
        cols = [opt(A["period"]), opt(A["t0"]), opt(A["b"]), opt(A["ecc"]), opt(A["omega"]), opt(r), opt(A["m_star"]),
                opt(A["r_star"]), opt(A["m_planet"]), sbr]
        us = [as_tensor(x, like) for x in list(u) + (list(secondary[0]) if secondary is not None else [])]
        if not (all(c is None or c.dim() <= 2 for c in cols) and all(x.dim() <= 1 for x in us)):
            raise TypeError("flux_value_and_grad: at most one draw dimension")
        pack_flags = (flags & (ops.FLAG_WINDOW | ops.FLAG_SECONDARY)) | (ops.PACK_CIRCULAR if A["ecc"] is None else 0)
        flux, L, gcols, gus = ops.orbit_flux_value_and_grad(t, weights, cols, us, flags=flags, pack_flags=pack_flags, texp=texp,
                                                            stencil_dt=sdt, stencil_w=sw, gscale=gscale, events=events,
                                                            wanted=[isinstance(x, torch.Tensor) for x in given])
        grads = {}
        for name, x, g in zip(names, given, gcols):
            if g is not None and isinstance(x, torch.Tensor):
                grads[name] = g.reshape(x.shape) if g.numel() == x.numel() else g.sum_to_size(x.shape)
        for name, x, g in zip(["u1", "u2", "u1s", "u2s"], list(u) + (list(secondary[0]) if secondary is not None else []), gus):
            if isinstance(x, torch.Tensor):
                grads[name] = g.reshape(x.shape) if g.numel() == x.numel() else g.sum_to_size(x.shape)
        return flux, L, grads

    def kernel_records(self, r, use_in_transit=False, secondary_sbr=None):
        """Pack the per-(draw, planet) parameter records of the fused transit
        kernel (layout: include/exoplanet_amd.h, EXO_P_*) from the orbit's attributes, in
        torch (any parameterisation).  Differentiable with respect to every orbit
        parameter; returns (D, P, 20) and the batch shape."""
        shape = self.shape
        z = torch.zeros(shape, dtype=torch.float64, device=self.a.device)
        r = _vec(r, self.a) + z
        e, cw, sw = self._ew()
        inf = torch.full_like(z, float("inf"))
        cols = [None] * ops.NPAR
        cols[ops.P_N] = self.n + z
        cols[ops.P_TP] = self.t_periastron + z
        cols[ops.P_ECC] = e + z
        cols[ops.P_COSW] = cw + z
        cols[ops.P_SINW] = sw + z
        cols[ops.P_COSI] = self.cos_incl + z
        # |cos i| > 1 (b beyond the orbit's largest impact parameter): sin(acos(.)) is NaN and the
        # reference's `switch(los > 0, lc, 0)` makes the flux 0 (limb_dark.py:252); sin i = 0 does the same
        # here -- as the packing kernel does for the standard parameterisation
        cols[ops.P_SINI] = torch.where((self.cos_incl + z).abs() > 1.0, z, self.sin_incl + z)
        cols[ops.P_AOR] = self.a / self.r_star + z
        cols[ops.P_ROR] = r / self.r_star
        cols[ops.P_T0] = (self.t0 + z).detach()
        cols[ops.P_PERIOD] = (self.period + z).detach()
        cols[ops.P_TS], cols[ops.P_TE] = -inf, inf
        cols[ops.P_FRATIO] = z
        cols[ops.P_TS2], cols[ops.P_TE2] = -inf, inf
        cols[ops.P_CLIGHT] = c_light / self.r_star + z          # read by light-delay sweeps only
        for k in range(ops.P_CLIGHT + 1, ops.NPAR):
            cols[k] = z                                         # reserved
        if secondary_sbr is not None:
            sbr = as_tensor(secondary_sbr, self.a)
            sbr = sbr.unsqueeze(-1) if sbr.dim() >= 1 else sbr
            cols[ops.P_FRATIO] = sbr * (r / self.r_star) ** 2 + z
        if use_in_transit:
            ts, te, flag = self._transit_window(r)
            bad = flag != 0
            cols[ops.P_TS] = torch.where(bad, -inf, ts)
            cols[ops.P_TE] = torch.where(bad, inf, te)
            if secondary_sbr is not None:
                # occultation window: the flipped orbit's own transit window, shifted to
                # its mid-occultation time and expressed in [0, P) after t0
                other = self._flip(r.detach())
                ts2, te2, flag2 = other._transit_window(self.r_star.detach() + z)
                P = (self.period + z).detach()
                shift = torch.remainder((other.t0 - self.t0).detach() + z, P)
                lo, hi = shift + ts2, shift + te2
                bad2 = (flag2 != 0) | (lo < 0) | (hi > P)
                cols[ops.P_TS2] = torch.where(bad2, -inf, lo)
                cols[ops.P_TE2] = torch.where(bad2, inf, hi)
        rec = torch.stack([c.expand(shape) for c in cols], dim=-1)
        batch = shape[:-1]
        return rec.reshape(-1, shape[-1], ops.NPAR), batch


def get_true_anomaly(M, e, **kwargs):
    """true anomaly from mean anomaly and eccentricity of the same shape (keplerian.py:807-819)"""
    sinf, cosf = ops.kepler(as_tensor(M), as_tensor(e))
    return torch.atan2(sinf, cosf)


def get_aor_from_transit_duration(duration, period, b, ror=None):
    """a / R_star implied by a circular orbit's duration, and d(a/R)/d(duration)
    (keplerian.py:822-846)"""
    if ror is None:
        ror = torch.zeros_like(b)
    b2 = b ** 2
    opk2 = (1 + ror) ** 2
    phi = math.pi * duration / period
    sinp, cosp = torch.sin(phi), torch.cos(phi)
    num = torch.sqrt(opk2 - b2 * cosp ** 2)
    aor = num / sinp
    grad = math.pi * cosp * (b2 - opk2) / (num * period * sinp ** 2)
    return aor, grad


def _consistent_inputs(a, period, rho_star, r_star, m_star, m_planet):
    """Complete {a, period, rho_star, r_star, m_star, m_planet} (keplerian.py:849-934)."""
    if a is None and period is None:
        raise ValueError("values must be provided for at least one of a and period")
    ref = a if a is not None else period
    if m_planet is None:
        m_planet = torch.zeros_like(ref)
    one = torch.ones(1, dtype=torch.float64, device=ref.device)

    implied_rho_star = False
    if a is not None and period is not None:
        if rho_star is not None or m_star is not None:
            raise ValueError("if both a and period are given, you can't also define rho_star or m_star")
        if r_star is None:
            r_star = one
        m_tot = 4 * math.pi * math.pi * a ** 3 / (G_grav * period ** 2)
        m_star = m_tot - m_planet
        rho_star = m_star / (4 * math.pi * r_star ** 3 / 3.0)
        implied_rho_star = True

    if r_star is None and m_star is None:
        r_star = one
        if rho_star is None:
            m_star = one
    if (not implied_rho_star) and sum(x is None for x in (rho_star, r_star, m_star)) != 1:
        raise ValueError("values must be provided for exactly two of rho_star, m_star, and r_star")

    if rho_star is not None and not implied_rho_star:
        rho_star = rho_star / gcc_per_sun
    if rho_star is None:
        rho_star = 3 * m_star / (4 * math.pi * r_star ** 3)
    elif r_star is None:
        r_star = (3 * m_star / (4 * math.pi * rho_star)) ** (1 / 3)
    elif m_star is None:
        m_star = 4 * math.pi * r_star ** 3 * rho_star / 3.0

    if a is None:
        a = (G_grav * (m_star + m_planet) * period ** 2 / (4 * math.pi ** 2)) ** (1.0 / 3)
    elif period is None:
        period = 2 * math.pi * a ** (3 / 2) / torch.sqrt(G_grav * (m_star + m_planet))
    return a, period, rho_star * gcc_per_sun, r_star, m_star, m_planet
