"""float64 numpy restatement of the hot path (TEST INFRASTRUCTURE ONLY).

Parity status: **parity unpinned** against exoplanet-core / celerite2 (absent
here, see oracle/__init__.py); pinned against oracle/mp_reference.py, which
evaluates the mathematical definitions at 34 digits.

Each function cites what it follows:
  * third-party ops  -> the published algorithm + the reference call site;
  * Python glue      -> the reference file:line it restates.
All citations are relative to /root/reference/.
"""
import numpy as np

# reference: src/exoplanet/orbits/constants.py:32-37 (literal fall-backs)
G_grav = 2942.2062175044193
gcc_per_sun = 5.905271918964842
au_per_R_sun = 0.00465046726096215
c_light = 37231.66360672704

TWO_PI_HI = 6.283185307179586
TWO_PI_LO = 2.4492935982947064e-16
# three-part split of 2 pi (30 + 30 + 53 bits): k * C1 and k * C2 are exact in
# float64 for |k| < 2^23, which stands in for the fused multiply-add numpy lacks
TWO_PI_C1 = 6.283185303211212
TWO_PI_C2 = 3.9683743166540886e-09
TWO_PI_C3 = 2.068073192717642e-18


# =============================================================================
# ops.kepler  (call sites: src/exoplanet/orbits/keplerian.py:333,818)
# =============================================================================
def _e_minus_sin(E):
    """E - sin(E) without cancellation for small E (series), direct otherwise."""
    E = np.asarray(E, dtype=np.float64)
    E2 = E * E
    # Taylor: E^3/6 (1 - E^2/20 (1 - E^2/42 (1 - E^2/72 (1 - E^2/110 (1 - E^2/156 (1-E^2/210))))))
    ser = (
        E * E2 / 6.0
        * (1 - E2 / 20 * (1 - E2 / 42 * (1 - E2 / 72 * (1 - E2 / 110 * (1 - E2 / 156 * (1 - E2 / 210 * (1 - E2 / 272)))))))
    )
    return np.where(np.abs(E) < 0.9, ser, E - np.sin(E))


def kepler_E(M, e):
    """Eccentric anomaly by Markley (1995, CeMDA 63, 101): cubic Pade starter
    + one fifth-order correction (fixed cost, no iteration).  M any real."""
    M = np.asarray(M, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64) + np.zeros_like(M)
    # Cody-Waite reduction of M to [-pi, pi] with exact partial products
    k = np.rint(M / TWO_PI_HI)
    Mr = ((M - k * TWO_PI_C1) - k * TWO_PI_C2) - k * TWO_PI_C3
    sgn = np.where(Mr < 0, -1.0, 1.0)
    Mr = np.abs(Mr)
    ome = 1.0 - e
    pi = np.pi
    alpha = (3 * pi * pi + 1.6 * pi * (pi - Mr) / (1 + e)) / (pi * pi - 6)
    d = 3 * ome + alpha * e
    q = 2 * alpha * d * ome - Mr * Mr
    r = 3 * alpha * d * (d - ome) * Mr + Mr ** 3
    w = (np.abs(r) + np.sqrt(q ** 3 + r * r)) ** (2.0 / 3.0)
    E = (2 * r * w / (w * w + w * q + q * q) + Mr) / d
    # fifth-order correction; residual via E(1-e) + e(E - sin E)
    sE, cE = np.sin(E), np.cos(E)
    f0 = ome * E + e * _e_minus_sin(E) - Mr
    f1 = 1 - e * cE
    f2 = e * sE
    f3 = 1 - f1
    f4 = -f2
    d3 = -f0 / (f1 - 0.5 * f0 * f2 / f1)
    d4 = -f0 / (f1 + 0.5 * d3 * f2 + d3 * d3 * f3 / 6)
    d5 = -f0 / (f1 + 0.5 * d4 * f2 + d4 * d4 * f3 / 6 + d4 ** 3 * f4 / 24)
    E = E + d5
    return sgn * E, k


def kepler(M, e):
    """(sinf, cosf) of the true anomaly.  e outside [0,1) -> NaN."""
    M = np.asarray(M, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64) + np.zeros_like(M)
    E, _ = kepler_E(M, e)
    # half-angle form: (cos f/2, sin f/2) ~ (sqrt(1-e) cos E/2, sqrt(1+e) sin E/2);
    # 1 - e cos E = X^2 + Y^2 has no cancellation as e -> 1, E -> 0
    with np.errstate(invalid="ignore"):
        X = np.sqrt(1 - e) * np.cos(0.5 * E)
        Y = np.sqrt(1 + e) * np.sin(0.5 * E)
    den = X * X + Y * Y
    cosf = (X * X - Y * Y) / den
    sinf = 2 * X * Y / den
    bad = ~((e >= 0) & (e < 1))
    return np.where(bad, np.nan, sinf), np.where(bad, np.nan, cosf)


def kepler_grad(sinf, cosf, e):
    """df/dM, df/de."""
    ome2 = 1 - e * e
    dfdM = (1 + e * cosf) ** 2 / ome2 ** 1.5
    dfde = (2 + e * cosf) * sinf / ome2
    return dfdM, dfde


# =============================================================================
# ops.quad_solution_vector  (call site: src/exoplanet/light_curves/limb_dark.py:24)
#
# Definition: s_n = int_{visible} g_n dA.  s0, s2: elementary (lens areas).
# s1: Green's theorem with the azimuthal field g(rho) = (1-(1-rho^2)^{3/2})/(3 rho)
# gives  s1 = 2pi/3 (1 - Theta(r-b)) + J,
#        J  = 1/3 int_arc (1-rho^2)^{3/2} dtheta
# reduced to complete elliptic integrals evaluated with Bulirsch's (1969) `cel`
# (the same building block Agol, Luger & Foreman-Mackey 2020 use).
# =============================================================================
def cel(kc, p, a, b, niter=9):
    """Bulirsch general complete elliptic integral, vectorised, fixed sweeps.
    cel = int_0^{pi/2} (a cos^2 + b sin^2)/(cos^2 + p sin^2)/sqrt(cos^2 + kc^2 sin^2)."""
    kc, p, a, b = np.broadcast_arrays(
        *[np.asarray(x, dtype=np.float64) for x in (kc, p, a, b)]
    )
    # floor: every caller's sin^2 coefficient vanishes with kc^2 (or the result is
    # multiplied by kc^2), so the floor costs O(1e-16 log) at most
    kc = np.maximum(np.abs(kc), 1e-8).copy()
    p = p.copy(); a = a.copy(); b = b.copy()
    e = kc.copy()
    em = np.ones_like(kc)
    pos = p > 0
    with np.errstate(all="ignore"):
        ps = np.sqrt(np.where(pos, p, 1.0))
        f = kc * kc
        q = 1.0 - f
        g = 1.0 - p
        f2 = f - p
        q2 = q * (b - a * p)
        pn = np.sqrt(np.where(pos, 1.0, f2 / g))
        an = (a - b) / g
        bn = -q2 / (g * g * pn) + an * pn
        b = np.where(pos, b / ps, bn)
        a = np.where(pos, a, an)
        p = np.where(pos, ps, pn)
        for _ in range(niter):
            f = a.copy()
            a = a + b / p
            g = e / p
            b = b + f * g
            b = b + b
            p = g + p
            g = em.copy()
            em = em + kc
            # every sweep is an exact (Landen) transformation of the integral,
            # so sweeping past convergence is harmless: no data-dependent exit
            kc = 2 * np.sqrt(e)
            e = kc * em
        return 0.5 * np.pi * (b + a * em) / (em * (em + p))


# (pi/2)-normalised moments  beta4[j] = (2/pi) int sin^{2j} cos^4 * (2j-1)!!/(2j)!!
def _c4_series_coeffs(n=40):
    out = []
    cj = 1.0  # (2j-1)!!/(2j)!!
    # (2/pi) int_0^{pi/2} sin^{2j} cos^4 = 3 (2j-1)!! / (2j+4)!! * ... compute by recurrence
    # I_j = int sin^{2j} cos^4 = I_{j-1} (2j-1)/(2j+4);  I_0 = 3 pi/16
    Ij = 3.0 / 16.0 * 2.0  # in units of pi/2
    for j in range(n):
        if j > 0:
            cj *= (2 * j - 1) / (2 * j)
            Ij *= (2 * j - 1) / (2 * j + 4)
        out.append(cj * Ij)
    return np.array(out)


_C4 = _c4_series_coeffs()


def _int_cos4(k2, kc2, E, K):
    """C4 = int_0^{pi/2} cos^4/sqrt(1-k2 sin^2): closed form, series for small k2."""
    with np.errstate(all="ignore"):
        closed = (2 * (2 * k2 - 1) * E + kc2 * (2 - 3 * k2) * K) / (3 * k2 * k2)
    ser = np.zeros_like(k2)
    for c in _C4[::-1]:
        ser = ser * k2 + c
    ser = ser * (0.5 * np.pi)
    return np.where(k2 < 0.3, ser, closed)


def quad_solution_vector(b, r):
    """-> s (...,3), dsdb (...,3), dsdr (...,3).  Takes |b| (the reference test
    feeds b in [-1.5,1.5], tests/light_curves_test.py:24-27); ds/db carries
    sign(b)."""
    b0 = np.asarray(b, dtype=np.float64)
    r0 = np.asarray(r, dtype=np.float64)
    b0, r0 = np.broadcast_arrays(b0, r0)
    sgn = np.where(b0 < 0, -1.0, 1.0)
    bb = np.abs(b0)
    shape = bb.shape
    bb = bb.ravel().copy()
    rr = r0.ravel().copy()
    n = bb.size
    s = np.empty((n, 3)); dsdb = np.zeros((n, 3)); dsdr = np.zeros((n, 3))
    s[:, 0] = np.pi; s[:, 1] = 2 * np.pi / 3; s[:, 2] = 0.0

    none = (rr <= 0) | (bb >= 1 + rr)
    full = (~none) & (rr >= 1 + bb)
    s[full] = 0.0
    act = ~(none | full)
    idx = np.nonzero(act)[0]
    if idx.size:
        b_ = bb[idx]; r_ = rr[idx]
        S, DB, DR = _sv_active(b_, r_)
        s[idx] = S; dsdb[idx] = DB; dsdr[idx] = DR
    nan = np.isnan(bb) | np.isnan(rr)
    s[nan] = np.nan; dsdb[nan] = np.nan; dsdr[nan] = np.nan
    dsdb *= sgn.ravel()[:, None]
    return s.reshape(shape + (3,)), dsdb.reshape(shape + (3,)), dsdr.reshape(shape + (3,))


def _x_minus_sin(x):
    """x - sin x, series below 0.9 (no cancellation), direct above."""
    return _e_minus_sin(x)


def _i4_num(k):
    """8 (k - sin k) - (2k - sin 2k)  (= 32 int_0^{k/2} sin^4): series for small k."""
    k2 = k * k
    c = [24.0 / 120, 120.0 / 5040, 504.0 / 362880, 2040.0 / 39916800,
         8184.0 / 6227020800, 32760.0 / 1307674368000, 131064.0 / 355687428096000,
         524280.0 / 121645100408832000]
    ser = np.zeros_like(k)
    for cj in c[::-1]:
        ser = cj - k2 * ser
    ser = ser * k ** 5
    return np.where(np.abs(k) < 0.4, ser, 8 * _x_minus_sin(k) - _x_minus_sin(2 * k))


def _sv_active(b, r):
    """Occultor overlaps the limb (partial) or sits inside the disk (inside)."""
    n = b.size
    S = np.empty((n, 3)); DB = np.empty((n, 3)); DR = np.empty((n, 3))
    r2 = r * r; b2 = b * b
    inside = b + r <= 1
    with np.errstate(all="ignore"):
        # 1-(b-r)^2 and (b+r)^2-1, factored and ordered so that the leading
        # subtraction is exact (Sterbenz) when the larger radius is near 1
        x = np.maximum(b, r); y = np.minimum(b, r)
        A = ((1 - x) + y) * (1 + (x - y))
        Bm = ((x - 1) + y) * ((x + y) + 1)
        kite = np.sqrt(np.maximum(0.0, A * Bm))
        kite = np.where(inside, 0.0, kite)      # = 2 b r sin k0 = 2 b sin k1
        # arc half-angles about the occultor centre (k0) and the star centre (k1);
        # atan2 keeps full relative accuracy for grazing (tiny) angles
        k0 = np.where(inside, np.pi, np.arctan2(kite, b2 + (r - 1) * (r + 1)))
        k1 = np.where(inside, 0.0, np.arctan2(kite, (1 - r) * (1 + r) + b2))
        sink0 = np.where(inside, 0.0, kite / (2 * b * r))
        # moments of sin^2, sin^4 over the half arc u in [0, k0/2]
        I2 = np.where(inside, np.pi / 4, _x_minus_sin(k0) / 4)
        I4 = np.where(inside, 3 * np.pi / 16, _i4_num(k0) / 32)
        u0 = 0.5 * k0
        rmb_ = r - b
        q = 1 - 2 * rmb_ * rmb_                 # (2 - 4 (b-r)^2)/2
        # ---- s0: pi - lens area, lens = two circular segments
        s0 = np.where(inside, np.pi * (1 - r2),
                      np.pi - 0.5 * (r2 * _x_minus_sin(2 * k0) + _x_minus_sin(2 * k1)))
        ds0dr = -2 * r * k0
        ds0db = 2 * r * sink0
        # ---- s2 = -int_lens (2 - 4 rho^2) dA via the field rho (1-rho^2) phi_hat
        #      (vanishes on the limb): only the occultor arc contributes
        s2 = -4 * r * (A * rmb_ * u0 + (2 * b * A - 4 * b * r * rmb_) * I2 - 8 * b2 * r * I4)
        ds2dr = -r * (4 * k0 * q - 64 * b * r * I2)
        ds2db = -4 * r * (-2 * q * u0 + (4 * q + 16 * b * r) * I2 - 32 * b * r * I4)

        # ---- s1
        sqA = np.sqrt(A)
        theta = np.where(r > b, 1.0, np.where(r == b, 0.5, 0.0))
        same = b == r
        rmb = np.where(same, 1.0, rmb_)
        # inside (complete integrals, modulus m = 4br/A)
        m = 4 * b * r / A
        kc2i = np.maximum(-Bm / A, 0.0)
        kci = np.sqrt(kc2i)
        Ei = cel(kci, 1.0, 1.0, kc2i)
        Ki = cel(kci, 1.0, 1.0, 1.0)
        t3 = (2 * (2 - m) * Ei - kc2i * Ki) / 3
        Ji = (2 * sqA / 3) * (A * t3 - (r2 - b2) * Ei)
        Pi = cel(kci, ((b + r) / rmb) ** 2, A, -Bm)
        Ji = Ji + np.where(same, 0.0, (2 * (r + b) / (3 * sqA * rmb)) * Pi)
        ds1dr_i = -4 * r * sqA * Ei
        ds1db_i = 4 * r * sqA * cel(kci, 1.0, 1.0, -kc2i) / 3
        # partial (modulus k2 = A/4br)
        k2 = np.minimum(A / (4 * b * r), 1.0)
        kc2p = np.maximum(1 - k2, 0.0)
        kcp = np.sqrt(kc2p)
        Ep = cel(kcp, 1.0, 1.0, kc2p)
        Kp = cel(kcp, 1.0, 1.0, 1.0)
        C2 = cel(kcp, 1.0, 1.0, 0.0)
        C4 = _int_cos4(k2, kc2p, Ep, Kp)
        kk = np.sqrt(k2)
        pref = 4 * sqA * kk
        Jp = (pref / 6) * (A * C4 - (r2 - b2) * C2)
        Pp = cel(kcp, 1.0 / (rmb * rmb), 1.0, 0.0)
        Jp = Jp + np.where(same, 0.0, ((r + b) / (6 * rmb)) * pref * Pp)
        ds1dr_p = -r * pref * C2
        ds1db_p = r * pref * (C2 * (1 - 2 * k2) + 2 * k2 * C4)

        J = np.where(inside, Ji, Jp)
        s1 = 2 * np.pi / 3 * (1 - theta) + J
        ds1dr = np.where(inside, ds1dr_i, ds1dr_p)
        ds1db = np.where(inside, ds1db_i, ds1db_p)

        # b == 0: s1 is elementary and every ds/db vanishes by symmetry
        z = b == 0
        s1 = np.where(z, 2 * np.pi / 3 * np.maximum(1 - r2, 0.0) ** 1.5, s1)
        ds1dr = np.where(z, -2 * np.pi * r * np.sqrt(np.maximum(1 - r2, 0.0)), ds1dr)
        ds1db = np.where(z, 0.0, ds1db)
        ds0db = np.where(z, 0.0, ds0db)
        ds2db = np.where(z, 0.0, ds2db)
    S[:, 0] = s0; S[:, 1] = s1; S[:, 2] = s2
    DB[:, 0] = ds0db; DB[:, 1] = ds1db; DB[:, 2] = ds2db
    DR[:, 0] = ds0dr; DR[:, 1] = ds1dr; DR[:, 2] = ds2dr
    return S, DB, DR


# =============================================================================
# ops.contact_points  (call site: src/exoplanet/orbits/keplerian.py:744-753)
#
# Mean anomalies of first / fourth contact: the two roots nearest the transit
# centre of  rho(f)^2 (1 - sin^2 i sin^2(omega+f)) = L^2,  L = R_star + r.
# Third-party algorithm unavailable; this is a bracketed bisection on the
# definition.  flag != 0  <=>  no bracket (caller then evaluates every cadence,
# keplerian.py:771-775).
# =============================================================================
def contact_points(a, e, cosw, sinw, cosi, sini, L):
    a, e, cosw, sinw, cosi, sini, L = [
        np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in (a, e, cosw, sinw, cosi, sini, L)
    ]
    a, e, cosw, sinw, cosi, sini, L = np.broadcast_arrays(a, e, cosw, sinw, cosi, sini, L)
    n = a.size
    Ml = np.zeros(n); Mr = np.zeros(n); flag = np.zeros(n, dtype=np.int32)
    for i in range(n):
        Ml[i], Mr[i], flag[i] = _contact_one(
            a.flat[i], e.flat[i], cosw.flat[i], sinw.flat[i], cosi.flat[i], sini.flat[i], L.flat[i]
        )
    return Ml.reshape(a.shape), Mr.reshape(a.shape), flag.reshape(a.shape)


def _contact_one(a, e, cosw, sinw, cosi, sini, L):
    p = a * (1 - e * e)

    def g(th):
        # th = omega + f - pi/2 ; cos f = sin(omega - th)
        st, ct = np.sin(th), np.cos(th)
        cosf = sinw * ct - cosw * st
        rho = p / (1 + e * cosf)
        return rho * rho * (st * st + cosi * cosi * ct * ct) - L * L

    if not (g(0.0) < 0):
        return 0.0, 0.0, 1
    out = []
    for sgn in (-1.0, 1.0):
        lo = 0.0
        hi = None
        for k in range(1, 33):
            th = sgn * k * (0.5 * np.pi / 32)
            if g(th) > 0:
                hi = th
                break
            lo = th
        if hi is None:
            return 0.0, 0.0, 1
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            if g(mid) > 0:
                hi = mid
            else:
                lo = mid
        th = 0.5 * (lo + hi)
        # f = th + pi/2 - omega ; half-angle to E, then M = E - e sin E
        w = np.arctan2(sinw, cosw)
        f = th + 0.5 * np.pi - w
        E = 2 * np.arctan2(np.sqrt(1 - e) * np.sin(0.5 * f), np.sqrt(1 + e) * np.cos(0.5 * f))
        out.append(E - e * np.sin(E))
    return out[0], out[1], 0


# =============================================================================
# Python glue, restated from the reference's own source
# =============================================================================
def get_aor_from_transit_duration(duration, period, b, ror=None):
    """reference: src/exoplanet/orbits/keplerian.py:822-846 (the value; the Jacobian is a derivative of it)"""
    ror = 0.0 if ror is None else ror
    phi = np.pi * duration / period
    return np.sqrt((1 + ror) ** 2 - b ** 2 * np.cos(phi) ** 2) / np.sin(phi)


def get_cl(u1, u2):
    """reference: src/exoplanet/light_curves/limb_dark.py:11-18"""
    c0 = 1 - u1 - 1.5 * u2
    c1 = u1 + 2 * u2
    c2 = -0.25 * u2
    norm = np.pi * (c0 + c1 / 1.5)
    return np.array([c0, c1, c2]) / norm


class KeplerianOrbit:
    """reference: src/exoplanet/orbits/keplerian.py:75-281 (constructor),
    :283-334 (rotation, anomaly), :380-409 (position), :708-777 (in_transit),
    :779-804 (_flip), :849-934 (_get_consistent_inputs).  Subset used by the
    hot path: no units, no Jacobian bookkeeping."""

    def __init__(self, period=None, a=None, t0=None, t_periastron=None, incl=None, b=None, duration=None,
                 ecc=None, omega=None, Omega=None, m_planet=0.0, m_star=None, r_star=None,
                 rho_star=None, ror=None):
        A = lambda x: None if x is None else np.atleast_1d(np.asarray(x, dtype=np.float64))
        if ecc is None and duration is not None:                      # :112-131 (circular orbit from its duration)
            if r_star is None:
                r_star = 1.0
            if b is None:
                raise ValueError("'b' must be provided for a circular orbit with a 'duration'")
            aor = get_aor_from_transit_duration(A(duration), A(period), A(b), ror=A(ror))
            a = A(r_star) * aor
            duration = None
        a, period, rho_star, r_star, m_star, m_planet = self._consistent(
            A(a), A(period), A(rho_star), A(r_star), A(m_star), A(m_planet))
        self.a, self.period, self.rho_star = a, period, rho_star
        self.r_star, self.m_star, self.m_planet = r_star, m_star, m_planet
        self.m_total = m_star + m_planet
        self.n = 2 * np.pi / period                                   # :146
        self.a_star = a * m_planet / self.m_total                     # :147
        self.a_planet = -a * m_star / self.m_total                    # :148
        self.K0 = self.n * a / self.m_total                           # :172 (divided by sqrt(1 - e^2) below, :213)
        self.Omega = A(Omega)
        if ecc is None:                                               # :182-185
            self.ecc = None
            self.M0 = 0.5 * np.pi + np.zeros_like(self.n)
            incl_factor = 1.0
        else:                                                         # :187-214
            self.ecc = A(ecc)
            if omega is None:
                raise ValueError("both e and omega must be provided")
            self.omega = A(omega)
            self.cos_omega = np.cos(self.omega)
            self.sin_omega = np.sin(self.omega)
            opsw = 1 + self.sin_omega
            E0 = 2 * np.arctan2(np.sqrt(1 - self.ecc) * self.cos_omega,
                                np.sqrt(1 + self.ecc) * opsw)
            self.M0 = E0 - self.ecc * np.sin(E0)
            ome2 = 1 - self.ecc ** 2
            self.K0 = self.K0 / np.sqrt(ome2)                         # :213
            incl_factor = (1 + self.ecc * self.sin_omega) / ome2
        self.dcosidb = incl_factor * self.r_star / self.a             # :217-219
        if b is not None:                                             # :221-228
            if incl is not None or duration is not None:
                raise ValueError("only one of 'incl', 'b', and 'duration' can be given")
            self.b = A(b) + np.zeros_like(self.a)
            self.cos_incl = self.dcosidb * self.b
            self.incl = np.arccos(self.cos_incl)
        elif incl is not None:                                        # :229-236
            self.incl = A(incl) + np.zeros_like(self.a)
            self.cos_incl = np.cos(self.incl)
            self.b = self.cos_incl / self.dcosidb
        elif duration is not None:                                    # :237-260 (eccentric orbit from its duration)
            self.duration = A(duration)
            c2 = np.sin(np.pi * self.duration * incl_factor / self.period) ** 2
            aor = self.a_planet / self.r_star
            esinw = self.ecc * self.sin_omega
            self.b = np.sqrt((aor ** 2 * c2 - 1) / (c2 * esinw ** 2 + 2 * c2 * esinw + c2 - self.ecc ** 4
                                                    + 2 * self.ecc ** 2 - 1))
            self.b = self.b * (1 - self.ecc ** 2)
            self.cos_incl = self.dcosidb * self.b
            self.incl = np.arccos(self.cos_incl)
        else:                                                         # :261-265
            zla = np.zeros_like(self.a)
            self.incl = 0.5 * np.pi + zla
            self.cos_incl = zla
            self.b = zla
        if t0 is not None and t_periastron is not None:               # :267-268
            raise ValueError("you can't define both t0 and t_periastron")
        if t0 is None and t_periastron is None:
            t0 = np.zeros_like(self.period)
        if t0 is None:                                                # :272-277
            self.t_periastron = A(t_periastron) + np.zeros_like(self.period)
            self.t0 = self.t_periastron + self.M0 / self.n
        else:
            self.t0 = A(t0) + np.zeros_like(self.period)
            self.t_periastron = self.t0 - self.M0 / self.n
        self.tref = self.t_periastron - self.t0                       # :279
        self.sin_incl = np.sin(self.incl)                             # :281

    @staticmethod
    def _consistent(a, period, rho_star, r_star, m_star, m_planet):
        """reference: keplerian.py:849-934"""
        if a is None and period is None:
            raise ValueError("values must be provided for at least one of a and period")
        implied = False
        if a is not None and period is not None:                      # :869-889
            if rho_star is not None or m_star is not None:
                raise ValueError("if both a and period are given, you can't also define rho_star or m_star")
            if r_star is None:
                r_star = np.array([1.0])
            m_tot = 4 * np.pi * np.pi * a ** 3 / (G_grav * period ** 2)
            m_star = m_tot - m_planet
            rho_star = m_star / (4 * np.pi * r_star ** 3 / 3.0)
            implied = True
        if r_star is None and m_star is None:                         # :892-895
            r_star = np.array([1.0])
            if rho_star is None:
                m_star = np.array([1.0])
        if (not implied) and sum(x is None for x in (rho_star, r_star, m_star)) != 1:
            raise ValueError("values must be provided for exactly two of rho_star, m_star, and r_star")
        if rho_star is not None and not implied:                      # :904-910
            rho_star = rho_star / gcc_per_sun
        if rho_star is None:                                          # :917-922
            rho_star = 3 * m_star / (4 * np.pi * r_star ** 3)
        elif r_star is None:
            r_star = (3 * m_star / (4 * np.pi * rho_star)) ** (1 / 3)
        elif m_star is None:
            m_star = 4 * np.pi * r_star ** 3 * rho_star / 3.0
        if a is None:                                                 # :925-932
            a = (G_grav * (m_star + m_planet) * period ** 2 / (4 * np.pi ** 2)) ** (1.0 / 3)
        elif period is None:
            period = 2 * np.pi * a ** 1.5 / np.sqrt(G_grav * (m_star + m_planet))
        return a, period, rho_star * gcc_per_sun, r_star, m_star, m_planet

    def _rotate_vector(self, x, y):
        """reference: keplerian.py:283-322"""
        if self.ecc is None:
            x1, y1 = x, y
        else:
            x1 = self.cos_omega * x - self.sin_omega * y
            y1 = self.sin_omega * x + self.cos_omega * y
        x2 = x1
        y2 = self.cos_incl * y1
        Z = -self.sin_incl * y1
        if self.Omega is None:
            return x2, y2, Z
        cO, sO = np.cos(self.Omega), np.sin(self.Omega)
        return cO * x2 - sO * y2, sO * x2 + cO * y2, Z

    def _warp_times(self, t, _pad=True):
        """reference: keplerian.py:324-327"""
        return (t[..., None] if _pad else t) - self.t0

    def _get_true_anomaly(self, t, _pad=True):
        """reference: keplerian.py:329-334"""
        M = (self._warp_times(t, _pad=_pad) - self.tref) * self.n
        if self.ecc is None:
            return np.sin(M), np.cos(M)
        return kepler(M, self.ecc + np.zeros_like(M))

    def _get_position(self, a, t, light_delay=False, _pad=True, parallax=None):
        """reference: keplerian.py:380-409 (and :411-470 for light_delay)"""
        t = np.asarray(t, dtype=np.float64)
        if light_delay:
            return self._get_retarded_position(a, t, _pad=_pad)
        sinf, cosf = self._get_true_anomaly(t, _pad=_pad)
        if self.ecc is None:
            r = a
        else:
            r = a * (1.0 - self.ecc ** 2) / (1 + self.ecc * cosf)
        if parallax is not None:                                      # :404-406
            r = r * parallax * au_per_R_sun
        return self._rotate_vector(r * cosf, r * sinf)

    def _get_retarded_position(self, a, t, z0=0.0, _pad=True):
        """reference: keplerian.py:411-470"""
        sinf, cosf = self._get_true_anomaly(t, _pad=_pad)
        angvel = 2 * np.pi / self.period
        if self.ecc is None:
            r = a
            vamp = angvel * a
            vz = vamp * self.sin_incl * cosf
        else:
            r = a * (1.0 - self.ecc ** 2) / (1 + self.ecc * cosf)
            vamp = angvel * a / np.sqrt(1 - self.ecc ** 2)
            cwf = self.cos_omega * cosf - self.sin_omega * sinf
            vz = vamp * self.sin_incl * (self.ecc * self.cos_omega + cwf)
        x, y, z = self._rotate_vector(r * cosf, r * sinf)
        az = -(angvel ** 2) * (a / r) ** 3 * z
        with np.errstate(all="ignore"):
            delay = np.where(
                np.abs(az) < 1.0e-10,
                (z0 - z) / (c_light + vz),
                (c_light / az) * ((1 + vz / c_light) - np.sqrt(
                    (1 + vz / c_light) * (1 + vz / c_light) - 2 * az * (z0 - z) / c_light ** 2)),
            )
        new_t = (t[..., None] if _pad else t) - delay
        return self._get_position(a, new_t, _pad=False)

    def get_relative_position(self, t, light_delay=False):
        """reference: keplerian.py:517-542"""
        return tuple(np.squeeze(x) for x in self._get_position(-self.a, t, light_delay=light_delay))

    def get_planet_position(self, t, parallax=None):
        """reference: keplerian.py:472-490"""
        return tuple(np.squeeze(x) for x in self._get_position(self.a_planet, t, parallax=parallax))

    def get_star_position(self, t, parallax=None):
        """reference: keplerian.py:492-515"""
        return tuple(np.squeeze(x) for x in self._get_position(self.a_star, t, parallax=parallax))

    def get_relative_angles(self, t, parallax=None):
        """reference: keplerian.py:544-570"""
        X, Y, Z = self._get_position(-self.a, t, parallax=parallax)
        return np.squeeze(np.sqrt(X ** 2 + Y ** 2)), np.squeeze(np.arctan2(Y, X))

    def _get_velocity(self, m, t):
        """reference: keplerian.py:572-578"""
        sinf, cosf = self._get_true_anomaly(np.asarray(t, dtype=np.float64))
        K = self.K0 * m
        if self.ecc is None:
            return self._rotate_vector(-K * sinf, K * cosf)
        return self._rotate_vector(-K * sinf, K * (cosf + self.ecc))

    def get_planet_velocity(self, t):
        """reference: keplerian.py:580-593"""
        return tuple(np.squeeze(x) for x in self._get_velocity(-self.m_star, t))

    def get_star_velocity(self, t):
        """reference: keplerian.py:595-612"""
        return tuple(np.squeeze(x) for x in self._get_velocity(self.m_planet, t))

    def get_relative_velocity(self, t):
        """reference: keplerian.py:614-631"""
        return tuple(np.squeeze(x) for x in self._get_velocity(-self.m_total, t))

    def _get_acceleration(self, a, m, t):
        """reference: keplerian.py:679-688"""
        sinf, cosf = self._get_true_anomaly(np.asarray(t, dtype=np.float64))
        K = self.K0 * m
        if self.ecc is None:
            factor = -(K ** 2) / a
        else:
            factor = K ** 2 * (self.ecc * cosf + 1) ** 2 / (a * (self.ecc ** 2 - 1))
        return self._rotate_vector(factor * cosf, factor * sinf)

    def get_planet_acceleration(self, t):
        """reference: keplerian.py:690-694"""
        return tuple(np.squeeze(x) for x in self._get_acceleration(self.a_planet, -self.m_star, t))

    def get_star_acceleration(self, t):
        """reference: keplerian.py:696-700"""
        return tuple(np.squeeze(x) for x in self._get_acceleration(self.a_star, self.m_planet, t))

    def get_relative_acceleration(self, t):
        """reference: keplerian.py:702-706"""
        return tuple(np.squeeze(x) for x in self._get_acceleration(-self.a, -self.m_total, t))

    def in_transit(self, t, r=0.0, texp=None):
        """reference: keplerian.py:708-777"""
        t = np.asarray(t, dtype=np.float64)
        z = np.zeros_like(self.a)
        r = np.asarray(r, dtype=np.float64) + z
        R = self.r_star + z
        hp = 0.5 * self.period
        dt = np.mod(self._warp_times(t) + hp, self.period) - hp
        if self.ecc is None:
            k = r / R
            arg = np.square(1 + k) - np.square(self.b)
            factor = R / (self.a * self.sin_incl)
            hdur = hp * np.arcsin(factor * np.sqrt(arg)) / np.pi
            t_start, t_end, flag = -hdur, hdur, z
        else:
            Ml, Mr, flag = contact_points(self.a, self.ecc, self.cos_omega, self.sin_omega,
                                          self.cos_incl + z, self.sin_incl + z, R + r)
            t_start = (Ml - self.M0) / self.n
            t_start = np.mod(t_start + hp, self.period) - hp
            t_end = (Mr - self.M0) / self.n
            t_end = np.mod(t_end + hp, self.period) - hp
            t_start = np.where(t_start > 0.0, t_start - self.period, t_start)
            t_end = np.where(t_end < 0.0, t_end + self.period, t_end)
        if texp is not None:
            t_start = t_start - 0.5 * texp
            t_end = t_end + 0.5 * texp
        mask = np.any((dt >= t_start) & (dt <= t_end), axis=-1)
        if np.all(flag == 0):
            return np.arange(t.shape[0])[mask]
        return np.arange(t.shape[0])

    def _flip(self, r_planet):
        """reference: keplerian.py:779-804"""
        if self.ecc is None:
            return KeplerianOrbit(period=self.period, t_periastron=self.t_periastron + 0.5 * self.period,
                                  incl=self.incl, Omega=self.Omega, m_star=self.m_planet,
                                  m_planet=self.m_star, r_star=r_planet)
        return KeplerianOrbit(period=self.period, t_periastron=self.t_periastron, incl=self.incl,
                              ecc=self.ecc, omega=self.omega - np.pi, Omega=self.Omega,
                              m_star=self.m_planet, m_planet=self.m_star, r_star=r_planet)


class TTVOrbit(KeplerianOrbit):
    """reference: src/exoplanet/orbits/ttv.py:71-187.  ``ttvs`` / ``transit_times`` /
    ``transit_inds``: one 1-D array per planet."""

    def __init__(self, *args, ttvs=None, transit_times=None, transit_inds=None, delta_log_period=None, **kwargs):
        if ttvs is None and transit_times is None:
            raise ValueError("one of 'ttvs' or 'transit_times' must be defined")
        if ttvs is not None:
            # ttv.py:79-89
            self.ttvs = [np.asarray(x, dtype=np.float64).reshape(-1) for x in ttvs]
            if transit_inds is None:
                self.transit_inds = [np.arange(x.size) for x in self.ttvs]
            else:
                self.transit_inds = [np.asarray(i, dtype=np.int64).reshape(-1) for i in transit_inds]
        else:
            # ttv.py:91-137: least-squares line through (index, time) per planet
            self.transit_times, self.ttvs, self.transit_inds = [], [], []
            period, t0 = [], []
            for i, times in enumerate(transit_times):
                times = np.asarray(times, dtype=np.float64).reshape(-1)
                inds = np.arange(times.size) if transit_inds is None else np.asarray(transit_inds[i], dtype=np.int64)
                self.transit_inds.append(inds)
                N = times.size
                sumx, sumx2, sumy, sumxy = np.sum(inds), np.sum(inds ** 2), np.sum(times), np.sum(inds * times)
                denom = N * sumx2 - sumx ** 2
                slope = (N * sumxy - sumx * sumy) / denom
                intercept = (sumx2 * sumy - sumx * sumxy) / denom
                period.append(slope)
                t0.append(intercept)
                self.ttvs.append(times - (intercept + inds * slope))
                self.transit_times.append(times)
            kwargs["t0"] = np.array(t0)
            self.ttv_period = np.array(period)
            if "period" not in kwargs:
                kwargs["period"] = (self.ttv_period if delta_log_period is None
                                    else np.exp(np.log(self.ttv_period) + delta_log_period))
        super().__init__(*args, **kwargs)
        t0 = np.atleast_1d(self.t0)
        per = np.atleast_1d(self.period)
        if ttvs is not None:
            # ttv.py:141-147
            self.ttv_period = per
            self.transit_times = [t0[i] + per[i] * self.transit_inds[i] + ttv for i, ttv in enumerate(self.ttvs)]
        # ttv.py:149-156: unobserved transit numbers follow the linear ephemeris
        self.all_transit_times = []
        for i, inds in enumerate(self.transit_inds):
            expect = t0[i] + per[i] * np.arange(inds.max() + 1)
            expect[inds] = self.transit_times[i]
            self.all_transit_times.append(expect)
        # ttv.py:158-170
        self._bin_edges = [np.concatenate(([tts[0] - 0.5 * self.ttv_period[i]], 0.5 * (tts[1:] + tts[:-1]),
                                           [tts[-1] + 0.5 * self.ttv_period[i]]))
                           for i, tts in enumerate(self.all_transit_times)]
        self._bin_values = [np.concatenate(([tts[0]], tts, [tts[-1]])) for tts in self.all_transit_times]

    def _get_model_dt(self, t):
        """reference: ttv.py:172-177"""
        return np.stack([self._bin_values[i][np.searchsorted(self._bin_edges[i], t)]
                         for i in range(len(self.ttvs))], axis=-1)

    def _warp_times(self, t, _pad=True):
        """reference: ttv.py:179-187"""
        if _pad:
            return t[..., None] - self._get_model_dt(t)
        return t - self._get_model_dt(t)

    def kernel_tables(self):
        """the fused entry points' view of the same histogram (include/exoplanet_amd.h,
        exo_transit_flux_ttv_*): edges (P, E) padded with +inf, shift (P, E + 1) = bin value - t0"""
        E = max(e.size for e in self._bin_edges)
        t0 = np.atleast_1d(self.t0)
        edges = np.full((len(self.ttvs), E), np.inf)
        shift = np.zeros((len(self.ttvs), E + 1))
        for i, (e, v) in enumerate(zip(self._bin_edges, self._bin_values)):
            edges[i, :e.size] = e
            shift[i, :v.size] = v - t0[i]
            shift[i, v.size:] = v[-1] - t0[i]
        return edges, shift


class LimbDarkLightCurve:
    """reference: src/exoplanet/light_curves/limb_dark.py:27-252"""

    def __init__(self, u1, u2):
        self.u1, self.u2 = float(u1), float(u2)
        self.c = get_cl(self.u1, self.u2)

    def _compute_light_curve(self, b, r, los=None):
        """reference: limb_dark.py:21-24, 234-252"""
        b = np.asarray(b, dtype=np.float64)
        r = np.asarray(r, dtype=np.float64) + np.zeros_like(b)
        s, _, _ = quad_solution_vector(b, r)
        lc = s @ self.c - 1.0
        if los is None:
            return lc
        return np.where(los > 0, lc, 0.0)

    def get_light_curve(self, orbit=None, r=None, t=None, texp=None, oversample=7, order=0,
                        use_in_transit=None, light_delay=False):
        """reference: limb_dark.py:99-232"""
        if orbit is None:
            raise ValueError("missing required argument 'orbit'")
        if r is None:
            raise ValueError("missing required argument 'r'")
        if t is None:
            raise ValueError("missing required argument 't'")
        use_in_transit = (not light_delay) if use_in_transit is None else use_in_transit
        r = np.atleast_1d(np.asarray(r, dtype=np.float64)).reshape(-1)
        t = np.asarray(t, dtype=np.float64)
        N = t.shape[0]
        if use_in_transit:
            model = np.zeros((N, r.size))
            inds = orbit.in_transit(t, r=r, texp=texp)
            t = t[inds]
        if texp is None:
            tgrid = t
        else:
            texp = np.asarray(texp, dtype=np.float64)
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
            stencil /= np.sum(stencil)
            if texp.ndim == 0:
                dt = texp * dt
            else:
                dt = (texp[inds] if use_in_transit else texp)[:, None] * dt
            tgrid = t[:, None] + dt
        coords = orbit.get_relative_position(tgrid, light_delay=light_delay)
        shape = tgrid.shape + (r.size,)
        b = np.sqrt(coords[0] ** 2 + coords[1] ** 2).reshape(shape)
        los = np.reshape(coords[2], shape)
        rs = orbit.r_star
        lc = self._compute_light_curve(b / rs, (r + np.zeros(shape)) / rs, los / rs)
        if texp is not None:
            lc = np.sum(stencil[None, :, None] * lc, axis=1)
        if use_in_transit:
            model[inds] = lc
            return model
        return lc


class SecondaryEclipseLightCurve:
    """reference: src/exoplanet/light_curves/secondary_eclipse.py:8-70"""

    def __init__(self, u_primary, u_secondary, surface_brightness_ratio):
        self.primary = LimbDarkLightCurve(u_primary[0], u_primary[1])
        self.secondary = LimbDarkLightCurve(u_secondary[0], u_secondary[1])
        self.surface_brightness_ratio = surface_brightness_ratio

    def get_light_curve(self, orbit=None, r=None, t=None, **kw):
        r = np.atleast_1d(np.asarray(r, dtype=np.float64))
        orbit2 = orbit._flip(r)
        lc1 = self.primary.get_light_curve(orbit=orbit, r=r, t=t, **kw)
        lc2 = self.secondary.get_light_curve(orbit=orbit2, r=orbit.r_star, t=t, **kw)
        k = r / orbit.r_star
        flux_ratio = self.surface_brightness_ratio * k ** 2
        return (lc1 + flux_ratio * lc2) / (1 + flux_ratio)


# =============================================================================
# Record-level restatement of the fused kernel's contract (include/exoplanet_amd.h)
# built from the pieces above; forward-mode Jacobian (the kernel is reverse-mode,
# so the two derivations are independent).
# =============================================================================
NPAR = 20
(P_N, P_TP, P_ECC, P_COSW, P_SINW, P_COSI, P_SINI, P_AOR, P_ROR, P_T0, P_PERIOD, P_TS, P_TE,
 P_FRATIO, P_TS2, P_TE2, P_CLIGHT) = range(17)   # slots 17..19 reserved (include/exoplanet_amd.h)
GRAD_SLOTS = (P_N, P_TP, P_ECC, P_COSW, P_SINW, P_COSI, P_AOR, P_ROR, P_FRATIO)


def exposure_stencil(oversample=7, order=0):
    """reference: limb_dark.py:181-197 -> (dt, weights) in units of texp"""
    oversample = int(oversample)
    oversample += 1 - oversample % 2
    w = np.ones(oversample)
    if order == 0:
        dt = np.linspace(-0.5, 0.5, 2 * oversample + 1)[1:-1:2]
    elif order == 1:
        dt = np.linspace(-0.5, 0.5, oversample)
        w[1:-1] = 2
    elif order == 2:
        dt = np.linspace(-0.5, 0.5, oversample)
        w[1:-1:2] = 4
        w[2:-1:2] = 2
    else:
        raise ValueError("order must be <= 2")
    return dt, w / np.sum(w)


def _window_mask(t, rec, htexp, secondary):
    hp = 0.5 * rec[P_PERIOD]
    dt = np.mod(t - rec[P_T0] + hp, rec[P_PERIOD]) - hp
    m = (dt >= rec[P_TS] - htexp) & (dt <= rec[P_TE] + htexp)
    if secondary:
        y = np.mod(t - rec[P_T0], rec[P_PERIOD])
        m |= (y >= rec[P_TS2] - htexp) & (y <= rec[P_TE2] + htexp)
    return m


def _sample(tt, rec, c, secondary, jac):
    """flux of one planet at times tt (any shape) -> F, and if jac: dict slot->dF, dF/dc (..,3|6)."""
    n, tp, e = rec[P_N], rec[P_TP], rec[P_ECC]
    cw, sw, ci, si, aor, ror, fr = (rec[P_COSW], rec[P_SINW], rec[P_COSI], rec[P_SINI], rec[P_AOR],
                                    rec[P_ROR], rec[P_FRATIO])
    M = (tt - tp) * n
    sinf, cosf = kepler(M, e + np.zeros_like(M))
    den = 1 + e * cosf
    rho = -aor * (1 - e * e) / den
    xo, yo = rho * cosf, rho * sinf
    x1 = cw * xo - sw * yo
    y1 = sw * xo + cw * yo
    Ys = ci * y1
    Z = -si * y1
    b = np.sqrt(x1 * x1 + Ys * Ys)
    front = Z > 0
    behind = secondary & (Z < 0)
    occ = behind
    bq = np.where(occ, b / ror, b)
    rq = np.where(occ, 1.0 / ror, ror) + np.zeros_like(b)
    s, dsdb, dsdr = quad_solution_vector(bq, rq)
    cc = np.where(occ[..., None], c[3:6] if secondary else c[:3], c[:3])
    Fq = np.sum(s * cc, axis=-1) - 1.0
    act = front | behind
    if secondary:
        wq = np.where(occ, fr / (1 + fr), 1 / (1 + fr))
    else:
        wq = 1.0
    F = np.where(act, Fq * wq, 0.0)
    if not jac:
        return F, None, None
    dFq_db = np.sum(dsdb * cc, axis=-1)
    dFq_dr = np.sum(dsdr * cc, axis=-1)
    dfdM, dfde = kepler_grad(sinf, cosf, e)
    with np.errstate(all="ignore"):
        ib = np.where(b > 0, 1.0 / b, 0.0)

    def chain(dxo, dyo, dcw=0.0, dsw=0.0, dci=0.0):
        dx1 = cw * dxo - sw * dyo + dcw * xo - dsw * yo
        dy1 = sw * dxo + cw * dyo + dsw * xo + dcw * yo
        dYs = ci * dy1 + dci * y1
        return (x1 * dx1 + Ys * dYs) * ib

    drho_df = rho * e * sinf / den
    dxo_df = drho_df * cosf - rho * sinf
    dyo_df = drho_df * sinf + rho * cosf
    db_df = chain(dxo_df, dyo_df)
    drho_de = rho * (-2 * e / (1 - e * e) - cosf / den)
    db = {
        P_N: db_df * dfdM * (tt - tp),
        P_TP: db_df * dfdM * (-n),
        P_ECC: db_df * dfde + chain(drho_de * cosf, drho_de * sinf),
        P_COSW: chain(0.0, 0.0, dcw=1.0),
        P_SINW: chain(0.0, 0.0, dsw=1.0),
        P_COSI: chain(0.0, 0.0, dci=1.0),
        P_AOR: b / aor,
    }
    scale_b = np.where(occ, 1.0 / ror, 1.0)  # d bq / d b
    out = {}
    for k, v in db.items():
        out[k] = np.where(act, wq * dFq_db * scale_b * v, 0.0)
    d_ror = np.where(occ, dFq_db * (-b / ror ** 2) + dFq_dr * (-1.0 / ror ** 2), dFq_dr)
    out[P_ROR] = np.where(act, wq * d_ror, 0.0)
    if secondary:
        out[P_FRATIO] = np.where(act, np.where(occ, 1.0, -1.0) * Fq / (1 + fr) ** 2, 0.0)
    else:
        out[P_FRATIO] = np.zeros_like(F)
    nld = 6 if secondary else 3
    dc = np.zeros(F.shape + (nld,))
    # Outside the overlap s = (pi, 2pi/3, 0) and s.c - 1 == 0 only because c is
    # normalised (pi c0 + 2pi/3 c1 = 1, limb_dark.py:17-18).  The fused op DEFINES
    # the flux as exactly 0 there, so those samples carry no c-cotangent; after
    # the chain through get_cl both conventions give identical d/du.
    overlap = act & (b < 1 + ror)
    sw_ = np.where(overlap, wq, 0.0)[..., None] * s
    if secondary:
        dc[..., :3] = np.where(occ[..., None], 0.0, sw_)
        dc[..., 3:] = np.where(occ[..., None], sw_, 0.0)
    else:
        dc[..., :3] = sw_
    return F, out, dc


def transit_flux(t, params, ld, texp=None, stencil_dt=None, stencil_w=None, per_planet=False, window=False,
                 secondary=False, jac=False, ttv=None):
    """flux [D,N] or [D,N,P]; with jac also J_params [D,N,(P),P,NPAR] is too big, so
    returns a callable-free form: (flux, dF_dparams [D,P,NPAR,N(,only own planet)], dF_dld [D,N(,P),nld])."""
    t = np.asarray(t, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    ld = np.asarray(ld, dtype=np.float64)
    D, P, _ = params.shape
    N = t.size
    if texp is None:
        sdt, sw_ = np.zeros(1), np.ones(1)
        tex = np.zeros(N)
    else:
        sdt, sw_ = np.asarray(stencil_dt, dtype=np.float64), np.asarray(stencil_w, dtype=np.float64)
        tex = np.asarray(texp, dtype=np.float64).reshape(-1) + np.zeros(N)
    tgrid = t[:, None] + tex[:, None] * sdt[None, :]           # [N,K]
    nld = 6 if secondary else 3
    flux = np.zeros((D, N, P))
    dpar = np.zeros((D, P, NPAR, N)) if jac else None           # d flux[d,:,p] / d params[d,p,slot]
    dld = np.zeros((D, N, P, nld)) if jac else None
    # timing variations (ttv.py:172-187): every time of the grid minus the shift of its bin;
    # dshift[d][p] = (bin of each grid time, d flux sample / d shift of that bin)
    dshift = [[None] * P for _ in range(D)] if (jac and ttv is not None) else None
    for d in range(D):
        for p in range(P):
            rec = params[d, p]
            tg, tc = tgrid, t
            if ttv is not None:
                edges, shift = np.asarray(ttv[0])[d, p], np.asarray(ttv[1])[d, p]
                bins = np.searchsorted(edges, tgrid)
                tg = tgrid - shift[bins]
                tc = t - shift[np.searchsorted(edges, t)]
            F, dF, dc = _sample(tg, rec, ld[d], secondary, jac)
            if window:
                m = _window_mask(tc, rec, 0.5 * tex, secondary)
            else:
                m = np.ones(N, dtype=bool)
            flux[d, :, p] = np.where(m, F @ sw_, 0.0)
            if jac:
                for k, v in dF.items():
                    dpar[d, p, k] = np.where(m, v @ sw_, 0.0)
                dld[d, :, p] = np.where(m[:, None], np.einsum("nkc,k->nc", dc, sw_), 0.0)
                if ttv is not None:
                    # the shift enters exactly like t_periastron: (t - shift - tp) n
                    dshift[d][p] = (bins, np.where(m[:, None], dF[P_TP] * sw_[None, :], 0.0))
    if not per_planet:
        fl = flux.sum(axis=2)
    else:
        fl = flux
    if jac and ttv is not None:
        return fl, dpar, dld, dshift
    return (fl, dpar, dld) if jac else fl


def transit_flux_vjp(t, params, ld, gflux, **kw):
    """Cotangents (gparams [D,P,NPAR], gld [D,nld]) for gflux shaped like the flux; with
    ``ttv=(edges, shift)`` also the cotangent of the shift table."""
    per_planet = kw.get("per_planet", False)
    out = transit_flux(t, params, ld, jac=True, **kw)
    fl, dpar, dld = out[:3]
    D, P, _ = np.asarray(params).shape
    g = np.asarray(gflux, dtype=np.float64)
    if not per_planet:
        g = np.repeat(g[:, :, None], P, axis=2)
    gparams = np.einsum("dpkn,dnp->dpk", dpar, g)
    gld = np.einsum("dnpc,dnp->dc", dld, g)
    if kw.get("ttv") is None:
        return fl, gparams, gld
    gshift = np.zeros_like(np.asarray(kw["ttv"][1], dtype=np.float64))
    for d in range(D):
        for p in range(P):
            bins, dF = out[3][d][p]
            np.add.at(gshift[d, p], bins, g[d, :, p][:, None] * dF)
    return fl, gparams, gld, gshift


# =============================================================================
# radial velocity: keplerian.py:633-677
# =============================================================================
RV_NPAR = 6
RV_N, RV_TP, RV_ECC, RV_COSW, RV_SINW, RV_AMP = range(6)


def radial_velocity(t, params, jac=False):
    """rv [D, N, P] = amp (cos w cos f - sin w sin f + e cos w) (keplerian.py:660-669; the mass-based
    form :671-676 is the same function of f with amp = conv sin(i) K0 m_planet, from :572-578 and
    :283-322).  params [D, P, 6] = (n, t_periastron, e, cos w, sin w, amp).  With jac: also
    d rv / d params [D, N, P, 6]."""
    t = np.asarray(t, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    n, tp, e, cw, sw, amp = (params[:, None, :, k] for k in range(RV_NPAR))
    M = (t[None, :, None] - tp) * n
    sinf, cosf = kepler(M, e + np.zeros_like(M))
    g = cw * cosf - sw * sinf + e * cw
    rv = amp * g
    if not jac:
        return rv
    dfdM, dfde = kepler_grad(sinf, cosf, e)
    dgdf = -(cw * sinf + sw * cosf)
    J = np.zeros(rv.shape + (RV_NPAR,))
    J[..., RV_N] = amp * dgdf * dfdM * (t[None, :, None] - tp)
    J[..., RV_TP] = -amp * dgdf * dfdM * n
    J[..., RV_ECC] = amp * (dgdf * dfde + cw)
    J[..., RV_COSW] = amp * (cosf + e)
    J[..., RV_SINW] = -amp * sinf
    J[..., RV_AMP] = g
    return rv, J


def radial_velocity_vjp(t, params, grv):
    """(rv, gparams [D, P, 6]) for a cotangent grv [D, N, P]"""
    rv, J = radial_velocity(t, params, jac=True)
    return rv, np.einsum("dnpk,dnp->dpk", J, np.asarray(grv, dtype=np.float64))


# =============================================================================
# celerite GP log-likelihood (celerite2 is absent and has NO call site in the
# reference tree: parity unpinned by construction; pinned here against the dense
# Cholesky likelihood).  Algorithm: Foreman-Mackey, Agol, Ambikasaran & Angus
# (2017) as restated with pre-conditioning in Foreman-Mackey (2018):
# SURVEY.md Appendix B.
# =============================================================================
def sho_coefficients(S0, w0, Q, eps=1e-5):
    """celerite coefficients (ar, cr, ac, bc, cc, dc) of a stochastically driven
    damped harmonic oscillator; Q < 1/2 -> two real terms, else one complex."""
    if Q < 0.5:
        f = np.sqrt(max(1.0 - 4.0 * Q * Q, eps))
        ar = 0.5 * S0 * w0 * Q * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f])
        cr = 0.5 * w0 / Q * np.array([1.0 - f, 1.0 + f])
        z = np.zeros(0)
        return ar, cr, z, z, z, z
    f = np.sqrt(max(4.0 * Q * Q - 1.0, eps))
    a = S0 * w0 * Q
    c = 0.5 * w0 / Q
    z = np.zeros(0)
    return z, z, np.array([a]), np.array([a / f]), np.array([c]), np.array([c * f])


def sho_from_sigma_rho(sigma, rho, Q):
    w0 = 2 * np.pi / rho
    S0 = sigma ** 2 / (w0 * Q)
    return S0, w0


def celerite_kernel(tau, ar, cr, ac, bc, cc, dc):
    tau = np.abs(tau)
    k = np.zeros_like(tau)
    for a, c in zip(ar, cr):
        k += a * np.exp(-c * tau)
    for a, b, c, d in zip(ac, bc, cc, dc):
        k += np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))
    return k


def gp_loglike_dense(t, y, diag, coeffs):
    """dense Cholesky Gaussian log-likelihood and its gradient w.r.t.
    (y, diag, ar, cr, ac, bc, cc, dc) via d loglike = 1/2 alpha^T dK alpha - 1/2 tr(K^-1 dK)."""
    ar, cr, ac, bc, cc, dc = [np.asarray(x, dtype=np.float64) for x in coeffs]
    t = np.asarray(t, dtype=np.float64)
    tau = np.abs(t[:, None] - t[None, :])
    K = celerite_kernel(tau, ar, cr, ac, bc, cc, dc) + np.diag(diag)
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
    ll = -0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * t.size * np.log(2 * np.pi)
    Kinv = np.linalg.inv(K)
    Wm = 0.5 * (np.outer(alpha, alpha) - Kinv)          # d ll / d K
    g = {"y": -alpha, "diag": np.diag(Wm).copy()}
    g["ar"] = np.array([np.sum(Wm * np.exp(-c * tau)) for c in cr])
    g["cr"] = np.array([np.sum(Wm * (-tau) * a * np.exp(-c * tau)) for a, c in zip(ar, cr)])
    ga, gb, gc, gd = [], [], [], []
    for a, b, c, d in zip(ac, bc, cc, dc):
        ex = np.exp(-c * tau); co = np.cos(d * tau); si = np.sin(d * tau)
        ga.append(np.sum(Wm * ex * co)); gb.append(np.sum(Wm * ex * si))
        gc.append(np.sum(Wm * (-tau) * ex * (a * co + b * si)))
        gd.append(np.sum(Wm * ex * tau * (-a * si + b * co)))
    g["ac"], g["bc"], g["cc"], g["dc"] = map(np.array, (ga, gb, gc, gd))
    return ll, g


def celerite_matrices(t, diag, coeffs):
    """(c, a, U, V) of SURVEY Appendix B from the term coefficients."""
    ar, cr, ac, bc, cc, dc = [np.asarray(x, dtype=np.float64) for x in coeffs]
    t = np.asarray(t, dtype=np.float64)
    N = t.size
    Jr, Jc = ar.size, ac.size
    J = Jr + 2 * Jc
    U = np.empty((N, J)); V = np.empty((N, J)); c = np.empty(J)
    U[:, :Jr] = ar; V[:, :Jr] = 1.0; c[:Jr] = cr
    for j in range(Jc):
        co, si = np.cos(dc[j] * t), np.sin(dc[j] * t)
        U[:, Jr + 2 * j] = ac[j] * co + bc[j] * si
        U[:, Jr + 2 * j + 1] = ac[j] * si - bc[j] * co
        V[:, Jr + 2 * j] = co
        V[:, Jr + 2 * j + 1] = si
        c[Jr + 2 * j] = c[Jr + 2 * j + 1] = cc[j]
    a = np.asarray(diag, dtype=np.float64) + np.sum(ar) + np.sum(ac)
    return c, a, U, V


# ------------------------------------------------------------------------------------------
# The same log-likelihood in parallel over time (restates exoplanet_amd/csrc/exo_celerite.hip,
# second half; docs/DESIGN_r1_r4.md 3.5).  The recurrences are a Kalman filter: with a symmetric Delta_n such
# that Delta_n U_n = V_n,  P_n = Delta_n - S_n  is the one-step prediction covariance and F_n its
# mean.  A run of cadences acts on the entering (F, P) as a filtering element (A, b, C, eta, J) of
# Sarkka & Garcia-Fernandez (2021, IEEE TAC 66, "Temporal parallelization of Bayesian smoothers"):
#     F' = A (I + P J)^-1 (F + P eta) + b ,     P' = A (I + P J)^-1 P A^T + C
# and its log-likelihood given the entering state is
#     const - 1/2 log det(I + P J) + 1/2 eta^T Y P eta + eta^T Y F - 1/2 F^T J Y F ,  Y = (I + P J)^-1.
# ------------------------------------------------------------------------------------------
def celerite_delta(coeffs, V_n):
    """Delta_n (J x J): 1/a for a real term; H Delta0 H for a complex pair, Delta0 = [[p, q], [q, r]],
    r = a/(a^2+b^2), q = -b/(a^2+b^2), p = (a^2+2b^2)/(a(a^2+b^2)), H = [[cs, sn], [sn, -cs]]."""
    ar, cr, ac, bc, cc, dc = [np.asarray(x, dtype=np.float64) for x in coeffs]
    Jr, Jc = ar.size, ac.size
    D = np.zeros((Jr + 2 * Jc, Jr + 2 * Jc))
    for j in range(Jr):
        D[j, j] = 1.0 / ar[j]
    for j in range(Jc):
        a, b = ac[j], bc[j]
        h = 1.0 / (a * a + b * b)
        D0 = np.array([[(a * a + 2 * b * b) * h / a, -b * h], [-b * h, a * h]])
        cs, sn = V_n[Jr + 2 * j], V_n[Jr + 2 * j + 1]
        H = np.array([[cs, sn], [sn, -cs]])
        D[Jr + 2 * j:Jr + 2 * j + 2, Jr + 2 * j:Jr + 2 * j + 2] = H @ D0 @ H
    return D


def celerite_chunk_element(t, y, diag, coeffs, n0, n1):
    """Filtering element (A, b, C, eta, Jm) of cadences [n0, n1): the filter run from (0, 0) beside
    the two sensitivity matrices A, Jm.  The last step of the series has no propagation."""
    c, a, U, V = celerite_matrices(t, diag, coeffs)
    J = U.shape[1]
    A = np.eye(J); b = np.zeros(J); C = np.zeros((J, J)); eta = np.zeros(J); Jm = np.zeros((J, J))
    for i in range(n0, n1):
        u, R = U[i], diag[i]
        r, cu = A.T @ u, C @ u
        s, zeta = R + u @ cu, y[i] - u @ b
        Jm += np.outer(r, r) / s
        eta += r * zeta / s
        if i + 1 < t.size:
            phi = np.exp(-c * (t[i + 1] - t[i]))
            Q = celerite_delta(coeffs, V[i + 1]) - np.outer(phi, phi) * celerite_delta(coeffs, V[i])
            k = cu / s
            A = phi[:, None] * (A - np.outer(k, r))
            b = phi * (b + k * zeta)
            C = np.outer(phi, phi) * (C - np.outer(k, cu)) + Q
    return A, b, C, eta, Jm


def celerite_apply_element(el, F, P):
    A, b, C, eta, Jm = el
    Y = np.linalg.inv(np.eye(F.size) + P @ Jm)
    Pn = A @ Y @ P @ A.T + C
    return A @ Y @ (F + P @ eta) + b, 0.5 * (Pn + Pn.T)


def celerite_chunk_loglike_closed_form(el, F, P, n_cad, const):
    """log p(y_chunk | entering state) from the element alone; `const` = the value at (F, P) = (0, 0)"""
    _, _, _, eta, Jm = el
    Y = np.linalg.inv(np.eye(F.size) + P @ Jm)
    return (const - 0.5 * np.linalg.slogdet(np.eye(F.size) + P @ Jm)[1] + 0.5 * eta @ Y @ P @ eta
            + eta @ Y @ F - 0.5 * F @ Jm @ Y @ F)


def celerite_run_chunk(t, y, diag, coeffs, n0, n1, F, S):
    """the ordinary recurrences over [n0, n1) from the entering (F, S); returns the chunk's
    sum (z^2/d + log d)"""
    c, a, U, V = celerite_matrices(t, diag, coeffs)
    acc = 0.0
    for n in range(n0, n1):
        if n > n0:
            Pp = np.exp(-c * (t[n] - t[n - 1]))
            S = np.outer(Pp, Pp) * (S + d * np.outer(W, W))
            F = Pp * (F + W * z)
        u = S @ U[n]
        d = a[n] - U[n] @ u
        W = (V[n] - u) / d
        z = y[n] - U[n] @ F
        acc += z * z / d + np.log(d)
    return acc


def celerite_loglike_chunked(t, y, diag, coeffs, n_chunks):
    """celerite_loglike by the time-parallel algorithm: elements per chunk, a scan over the chunks
    for the entering states, the ordinary recurrences per chunk from those states."""
    t = np.asarray(t, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    diag = np.asarray(diag, dtype=np.float64) + np.zeros_like(t)
    N = t.size
    L = -(-N // n_chunks)
    bounds = [(n0, min(N, n0 + L)) for n0 in range(0, N, L)]
    _, _, _, V = celerite_matrices(t, diag, coeffs)
    J = V.shape[1]
    F, P = np.zeros(J), celerite_delta(coeffs, V[0])          # S_0 = 0
    acc = 0.0
    for k, (n0, n1) in enumerate(bounds):
        acc += celerite_run_chunk(t, y, diag, coeffs, n0, n1, F, celerite_delta(coeffs, V[n0]) - P)
        if k + 1 < len(bounds):
            F, P = celerite_apply_element(celerite_chunk_element(t, y, diag, coeffs, n0, n1), F, P)
    return -0.5 * acc - 0.5 * N * np.log(2 * np.pi)


def celerite_chunk_adjoint_step(el, F, P, Fbar_next, Pbar_next, gL=1.0):
    """one step of the reverse scan over the chunks: the adjoint of the state (F, P) entering a chunk
    from the adjoint of the state entering the next one -- local term from the closed-form chunk
    likelihood plus the transposed element map (Abar = A Y, g = eta - Jm Y (F + P eta))."""
    A, b, C, eta, Jm = el
    Y = np.linalg.inv(np.eye(F.size) + P @ Jm)
    w = Y.T @ (eta - Jm @ F)
    g = eta - Jm @ Y @ (F + P @ eta)
    Ab = A @ Y
    x = Ab.T @ Fbar_next
    Fbar = gL * w + x
    Pbar = 0.5 * gL * (np.outer(w, w) - Jm @ Y) + Ab.T @ Pbar_next @ Ab + 0.5 * (np.outer(x, g) + np.outer(g, x))
    return Fbar, 0.5 * (Pbar + Pbar.T)


def celerite_loglike(t, y, diag, coeffs):
    """O(N J^2) semiseparable Cholesky + forward substitution -> log-likelihood."""
    c, a, U, V = celerite_matrices(t, diag, coeffs)
    t = np.asarray(t, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    N, J = U.shape
    S = np.zeros((J, J)); F = np.zeros(J)
    d = a[0]; W = V[0] / d; z = y[0]
    acc = z * z / d + np.log(d)
    for n in range(1, N):
        P = np.exp(-c * (t[n] - t[n - 1]))
        S = np.outer(P, P) * (S + d * np.outer(W, W))
        F = P * (F + W * z)
        u = S @ U[n]
        d = a[n] - U[n] @ u
        if not d > 0:
            return -np.inf
        W = (V[n] - u) / d
        z = y[n] - U[n] @ F
        acc += z * z / d + np.log(d)
    return -0.5 * acc - 0.5 * N * np.log(2 * np.pi)


# ---------------------------------------------------------------------------------------------
# Not a reference function: the conjunction-window bound of the HIP path (exoplanet_amd/csrc/exo_transit.hip,
# transit_window_kernel), restated so that its defining property -- the window holds every true anomaly at which
# the disks can overlap -- can be checked on the CPU against brute force (tests/test_window_bound.py).  The
# reference has no counterpart: its in_transit (keplerian.py:708-777) solves for the contacts themselves.
# ---------------------------------------------------------------------------------------------
def conjunction_window(e, omega, cosi, sini, aor, ror, event=0, rounds=3):
    """(f0, d_lo, d_hi): overlap of the disks (separation < 1 + ror, body on the near / far side for event 0 / 1)
    is only possible for true anomalies in [f0 - d_lo, f0 + d_hi]; None if the first bound does not exist"""
    lim = 1.0 + abs(ror)
    q = lim / (abs(aor) * (1.0 - e))
    if not (0.0 <= e < 1.0) or not q < 0.999:
        return None
    delta0 = np.arcsin(q)
    f0 = (-0.5 if sini < 0 else 0.5) * np.pi - omega + event * np.pi
    semi = abs(aor) * (1.0 - e * e)
    si2, ci2 = sini * sini, cosi * cosi
    out = []
    for sgn in (-1.0, 1.0):
        ub = delta0
        if si2 > 1e-12:
            def ang(dist):
                S = (lim * lim / (dist * dist) - ci2) / si2
                return 0.0 if S <= 0 else (np.arcsin(np.sqrt(S)) if S < 1 else delta0)
            dist_at = lambda f: semi / (1.0 + e * np.cos(f))  # noqa: E731
            d_c = dist_at(f0)
            for _ in range(rounds):
                f_end = f0 + sgn * ub
                lo, hi = min(f0, f_end), max(f0, f_end)
                apsis = np.floor(hi / np.pi) >= np.ceil(lo / np.pi)
                peri = np.floor(hi / (2 * np.pi)) >= np.ceil(lo / (2 * np.pi))
                d_end = dist_at(f_end)
                if not apsis and d_end >= d_c:
                    lb = min(ang(d_end), ub)
                    ub = min(ub, ang(dist_at(f0 + sgn * lb)))
                else:
                    ub = min(ub, ang(semi / (1.0 + e) if peri else min(d_end, d_c)))
        out.append(ub * (1.0 + 1e-6) + 1e-6)
    return f0, out[0], out[1]
