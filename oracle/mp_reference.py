"""Arbitrary-precision DEFINITIONS for the hot path (test infrastructure only).

Nothing here follows an implementation; every function evaluates the
mathematical definition the reference's third-party ops are documented to
compute, with mpmath at >= 30 digits.

  kepler(M, e)            root of E - e sin E = M  -> sin f, cos f and their
                          partials (call site: reference
                          src/exoplanet/orbits/keplerian.py:333)
  quad_sv(b, r)           s_n = int_{visible disk} g_n dA, g = (1, mu, 4mu^2-2)
                          (call site: src/exoplanet/light_curves/limb_dark.py:24)
  quad_sv_grad(b, r)      d s / d b, d s / d r as boundary (Leibniz) integrals
  gp_loglike_dense(...)   dense Cholesky log-likelihood of a celerite kernel
"""
import mpmath as mp

mp.mp.dps = 34


# ----------------------------------------------------------------------------
# Kepler
# ----------------------------------------------------------------------------
def kepler_E(M, e):
    """Eccentric anomaly: root of E - e sin E = M (any real M, 0 <= e < 1)."""
    M = mp.mpf(M)
    e = mp.mpf(e)
    two_pi = 2 * mp.pi
    # reduce to (-pi, pi]
    k = mp.floor((M + mp.pi) / two_pi)
    Mr = M - k * two_pi
    sgn = 1
    if Mr < 0:
        Mr = -Mr
        sgn = -1
    # bisection start then Newton in high precision: g(E)=E-e sinE-Mr on [0,pi]
    lo, hi = mp.mpf(0), mp.pi
    for _ in range(60):
        mid = (lo + hi) / 2
        if mid - e * mp.sin(mid) - Mr > 0:
            hi = mid
        else:
            lo = mid
    E = (lo + hi) / 2
    for _ in range(8):
        g = E - e * mp.sin(E) - Mr
        gp = 1 - e * mp.cos(E)
        if gp == 0:
            break
        E = E - g / gp
    return sgn * E + k * two_pi


def kepler(M, e):
    """-> (sinf, cosf, dsinf/dM, dcosf/dM, dsinf/de, dcosf/de).

    cosf = (cosE - e)/(1 - e cosE), sinf = sqrt(1-e^2) sinE/(1 - e cosE);
    df/dM = (1+e cosf)^2/(1-e^2)^{3/2}, df/de = (2+e cosf) sinf/(1-e^2)."""
    e = mp.mpf(e)
    E = kepler_E(M, e)
    cE, sE = mp.cos(E), mp.sin(E)
    den = 1 - e * cE
    cosf = (cE - e) / den
    sinf = mp.sqrt(1 - e * e) * sE / den
    ome2 = 1 - e * e
    dfdM = (1 + e * cosf) ** 2 / ome2 ** mp.mpf(1.5)
    dfde = (2 + e * cosf) * sinf / ome2
    return (sinf, cosf, cosf * dfdM, -sinf * dfdM, cosf * dfde, -sinf * dfde)


# ----------------------------------------------------------------------------
# Quadratic limb-darkening solution vector
# ----------------------------------------------------------------------------
def _width(rho, b, r):
    """Angular width (seen from the star centre) of the occultor at radius rho."""
    if rho + b <= r:
        return 2 * mp.pi
    if rho == 0:
        return 2 * mp.pi if b < r else mp.mpf(0)
    if abs(rho - b) >= r:
        return mp.mpf(0)
    c = (rho * rho + b * b - r * r) / (2 * rho * b)
    return 2 * mp.acos(c)


def quad_sv(b, r):
    """s = (s0, s1, s2): integrals over the visible part of the unit disk of
    (1, mu, 4 mu^2 - 2), mu = sqrt(1 - x^2 - y^2), for an occultor of radius r
    centred at distance |b|.  Polar quadrature about the star centre."""
    b = abs(mp.mpf(b))
    r = mp.mpf(r)
    full = (mp.pi, 2 * mp.pi / 3, mp.mpf(0))
    if r == 0 or b >= 1 + r:
        return full
    if r >= 1 + b:
        return (mp.mpf(0), mp.mpf(0), mp.mpf(0))
    pts = sorted({mp.mpf(0), abs(b - r), min(mp.mpf(1), b + r), mp.mpf(1)})
    pts = [p for p in pts if 0 <= p <= 1]
    out = []
    for n, g in enumerate(
        (
            lambda x: mp.mpf(1),
            lambda x: mp.sqrt(1 - x * x),
            lambda x: 2 - 4 * x * x,
        )
    ):
        occ = mp.quad(lambda x: x * g(x) * _width(x, b, r), pts)
        out.append(full[n] - occ)
    return tuple(out)


def _arc(b, r):
    """Half-angle (about the occultor centre, measured from the direction that
    points at the star centre) of the occultor boundary lying inside the star."""
    if b + r <= 1:
        return mp.pi
    c = (r * r + b * b - 1) / (2 * b * r)
    return mp.acos(c)


def quad_sv_grad(b, r):
    """(ds/db, ds/dr), each a 3-tuple, from the Leibniz boundary integrals

        ds_n/dr = -r int_arc g_n(rho(phi)) dphi,
        ds_n/db = -r int_arc g_n(rho(phi)) cos(phi) dphi,

    rho^2 = b^2 + r^2 + 2 b r cos(phi), over the part of the occultor boundary
    inside the star.  Sign of ds/db follows sign(b) (the op takes |b|)."""
    sgn = -1 if b < 0 else 1
    b = abs(mp.mpf(b))
    r = mp.mpf(r)
    zero = (mp.mpf(0),) * 3
    if r == 0 or b >= 1 + r or r >= 1 + b:
        return zero, zero
    al = _arc(b, r)
    gs = (
        lambda q: mp.mpf(1),
        lambda q: mp.sqrt(max(mp.mpf(0), 1 - q)),
        lambda q: 2 - 4 * q,
    )
    dsdb, dsdr = [], []
    for g in gs:
        f = lambda u: g(b * b + r * r - 2 * b * r * mp.cos(u))  # phi = pi + u
        dr = -r * 2 * mp.quad(f, [0, al / 2, al])
        db = -r * 2 * mp.quad(lambda u: -mp.cos(u) * f(u), [0, al / 2, al])
        dsdr.append(dr)
        dsdb.append(sgn * db)
    return tuple(dsdb), tuple(dsdr)


# ----------------------------------------------------------------------------
# celerite kernel, dense
# ----------------------------------------------------------------------------
def celerite_kernel(tau, ar, cr, ac, bc, cc, dc):
    tau = abs(tau)
    k = mp.mpf(0)
    for a, c in zip(ar, cr):
        k += a * mp.exp(-c * tau)
    for a, b, c, d in zip(ac, bc, cc, dc):
        k += mp.exp(-c * tau) * (a * mp.cos(d * tau) + b * mp.sin(d * tau))
    return k


def gp_loglike_dense(t, y, diag, ar, cr, ac, bc, cc, dc):
    n = len(t)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = celerite_kernel(t[i] - t[j], ar, cr, ac, bc, cc, dc)
        K[i, i] += diag[i]
    L = mp.cholesky(K)
    yv = mp.matrix(y)
    z = mp.lu_solve(L, yv)  # lower-triangular solve (generic LU on L)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    return -(z.T * z)[0] / 2 - logdet / 2 - n * mp.log(2 * mp.pi) / 2
