"""CPU: pin the oracle (numpy port, C port) and the host-compiled device math
against the golden vectors generated from the arbitrary-precision definitions
(oracle/make_golden.py -> tests/golden/*.npz)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import c_port as C
from oracle import numpy_port as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kepler_tol(M, e, cosf):
    """8 ulp scaled by the conditioning of (sinf, cosf) w.r.t. the rounding of M"""
    cond = 1 + (1 + e * cosf) ** 2 / (1 - e * e) ** 1.5 * np.abs(np.remainder(M + np.pi, 2 * np.pi) - np.pi)
    return 8 * 2.3e-16 * cond


@pytest.fixture(scope="module")
def harness():
    """the device math header compiled for the host (test harness, not a product path)"""
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "host_harness.so")
    srcs = [os.path.join(ROOT, "tests", "host_harness.cpp"), os.path.join(ROOT, "exoplanet_amd", "csrc", "exo_math.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, srcs[0]], check=True)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.mark.parametrize("impl", ["numpy", "c", "device_math_on_host"])
def test_kepler_golden(impl, harness):
    g = np.load(os.path.join(GOLD, "kepler.npz"))
    M, e = g["M"], g["ecc"]
    if impl == "numpy":
        s, c = P.kepler(M, e)
    elif impl == "c":
        s, c = C.kepler(M, e)
    else:
        s, c = np.empty_like(M), np.empty_like(M)
        harness.harness_kepler(_p(M), _p(e), _p(s), _p(c), ctypes.c_int64(M.size))
    tol = kepler_tol(M, e, g["cosf"])
    assert np.all(np.abs(s - g["sinf"]) <= tol)
    assert np.all(np.abs(c - g["cosf"]) <= tol)
    # closed-form partials
    m = e <= 0.999  # 1 - e^2 itself loses digits in float64 beyond that
    dfdM, dfde = P.kepler_grad(g["sinf"], g["cosf"], e)
    np.testing.assert_allclose((g["cosf"] * dfdM)[m], g["dsinf_dM"][m], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose((-g["sinf"] * dfde)[m], g["dcosf_de"][m], rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("impl", ["numpy", "c", "device_math_on_host"])
def test_quad_sv_golden(impl, harness):
    g = np.load(os.path.join(GOLD, "quad_sv.npz"))
    b, r = g["b"], g["r"]
    if impl == "numpy":
        s, db, dr = P.quad_solution_vector(b, r)
    elif impl == "c":
        s, db, dr = C.quad_solution_vector(b, r)
    else:
        s, db, dr = np.empty((b.size, 3)), np.empty((b.size, 3)), np.empty((b.size, 3))
        harness.harness_quad_sv(_p(b), _p(r), _p(s), _p(db), _p(dr), ctypes.c_int64(b.size))
    assert np.abs(s - g["s"]).max() < 5e-15
    # derivatives have square-root branch points at the contact loci; the points
    # placed 1e-8 from them amplify input rounding by ~1e4
    gap = np.minimum.reduce([np.abs(np.abs(b) - np.abs(1 - r)), np.abs(np.abs(b) - (1 + r)), np.abs(np.abs(b) - r) + 1e-3])
    tol = 5e-14 + 2e-15 / np.sqrt(np.maximum(gap, 1e-16))
    assert np.all(np.abs(db - g["dsdb"]).max(axis=1) <= tol)
    assert np.all(np.abs(dr - g["dsdr"]).max(axis=1) <= tol)


def test_singular_point_continuity():
    """reference tests/light_curves_test.py:220-254 on the oracle"""
    lc = P.LimbDarkLightCurve(0.2, 0.3)
    for b, r, be, re in [(0.1, 0.9, 1e-8, 0), (0.5, 0.5, 1e-8, 0), (0.0, 0.1, 1e-8, 0), (0.0, 1.0, 0, 1e-8),
                         (1.1, 0.1, 1e-8, 0)]:
        f = lc._compute_light_curve(np.array([b - be, b + be, b]), np.array([r - re, r + re, r]))
        assert np.allclose(np.mean(f[:2]), f[2])


def test_gp_golden():
    g = np.load(os.path.join(GOLD, "gp_sho.npz"))
    for tag in ("q03", "q07", "q3"):
        co = tuple(g[f"{tag}_{nm}"] for nm in ("ar", "cr", "ac", "bc", "cc", "dc"))
        want = float(g[f"{tag}_loglike"])
        got = P.celerite_loglike(g[f"{tag}_t"], g[f"{tag}_y"], g[f"{tag}_diag"], co)
        dense, _ = P.gp_loglike_dense(g[f"{tag}_t"], g[f"{tag}_y"], g[f"{tag}_diag"], co)
        assert abs(got - want) < 1e-12 * abs(want)
        assert abs(dense - want) < 1e-12 * abs(want)


def test_lightcurve_golden_numpy_and_c():
    from oracle.make_golden import LIGHTCURVE_CASES, case_time

    g = np.load(os.path.join(GOLD, "lightcurves.npz"))
    for name, case in LIGHTCURVE_CASES.items():
        okw = {k: (np.array(v, dtype=float) if isinstance(v, list) else v) for k, v in case["orbit"].items()}
        orbit = P.KeplerianOrbit(**okw)
        t = case_time(case["t"])
        r = np.array(case["r"])
        for texp in case["texp"]:
            want = g[f"{name}_texp{texp}"]
            assert want.min() < -1e-4
            # the reference's default compaction path == evaluating every cadence
            got = P.LimbDarkLightCurve(*case["u"]).get_light_curve(orbit=orbit, r=r, t=t, texp=texp)
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
            # C port through the record-level contract
            rec = _records(orbit, r)
            kw = {}
            if texp is not None:
                sdt, sw = P.exposure_stencil(7, 0)
                kw = dict(texp=texp, stencil_dt=sdt, stencil_w=sw)
            f, _, _ = C.transit(t, rec, P.get_cl(*case["u"])[None], per_planet=True, **kw)
            np.testing.assert_allclose(f[0], want, rtol=0, atol=5e-15)


def _records(orbit, r):
    n = orbit.a.size
    rec = np.zeros((1, n, P.NPAR))
    ecc = orbit.ecc if orbit.ecc is not None else np.zeros(n)
    rec[0, :, P.P_N] = orbit.n; rec[0, :, P.P_TP] = orbit.t_periastron; rec[0, :, P.P_ECC] = ecc
    rec[0, :, P.P_COSW] = orbit.cos_omega if orbit.ecc is not None else 1.0
    rec[0, :, P.P_SINW] = orbit.sin_omega if orbit.ecc is not None else 0.0
    rec[0, :, P.P_COSI] = orbit.cos_incl; rec[0, :, P.P_SINI] = orbit.sin_incl
    rec[0, :, P.P_AOR] = orbit.a / orbit.r_star; rec[0, :, P.P_ROR] = r / orbit.r_star
    rec[0, :, P.P_T0] = orbit.t0; rec[0, :, P.P_PERIOD] = orbit.period
    rec[0, :, [P.P_TS, P.P_TS2]] = -np.inf; rec[0, :, [P.P_TE, P.P_TE2]] = np.inf
    return rec


def test_c_port_vjp_matches_forward_mode_jacobian():
    """reverse-mode C port == forward-mode numpy Jacobian (independent derivations)"""
    rng = np.random.default_rng(0)
    t = np.linspace(-20, 20, 3000)
    orbit = P.KeplerianOrbit(m_star=1.45, r_star=1.5, t0=np.array([0.5, 17.4]), period=np.array([10.0, 5.3]),
                             ecc=np.array([0.1, 0.8]), omega=np.array([0.5, 1.3]), m_planet=np.array([0.3, 0.5]),
                             b=np.array([0.2, 0.5]))
    rec = _records(orbit, np.array([0.1, 0.05]))
    rec[0, :, P.P_FRATIO] = 0.3 * rec[0, :, P.P_ROR] ** 2
    c6 = np.concatenate([P.get_cl(0.2, 0.3), P.get_cl(0.4, 0.1)])[None]
    sdt, sw = P.exposure_stencil(7, 2)
    for kw in (dict(), dict(per_planet=True), dict(secondary=True)):
        cl = c6 if kw.get("secondary") else c6[:, :3]
        g = rng.normal(size=(1, t.size, 2) if kw.get("per_planet") else (1, t.size))
        f, gp, gl = C.transit(t, rec, cl, g, texp=0.1, stencil_dt=sdt, stencil_w=sw, **kw)
        F, GP, GL = P.transit_flux_vjp(t, rec, cl, g, texp=0.1, stencil_dt=sdt, stencil_w=sw, **kw)
        sl = list(P.GRAD_SLOTS)
        np.testing.assert_allclose(f, F, rtol=0, atol=1e-15)
        assert np.abs(gp[..., sl] - GP[..., sl]).max() <= 1e-12 * np.abs(GP[..., sl]).max()
        np.testing.assert_allclose(gl, GL, rtol=1e-12, atol=1e-12)


def test_reference_properties_on_oracle():
    """a few of the reference's self-consistency tests, restated on the oracle:
    in-transit == full (light_curves_test.py:75-102), b(t0) == impact parameter at e=0.8
    (keplerian_test.py:352-374), secondary blend (light_curves_test.py:285-311)."""
    t = np.linspace(-20, 20, 1000)
    orbit = P.KeplerianOrbit(m_star=1.45, r_star=1.5, t0=np.array([0.5, 17.4]), period=np.array([10.0, 5.3]),
                             ecc=np.array([0.1, 0.8]), omega=np.array([0.5, 1.3]), m_planet=np.array([0.3, 0.5]))
    lc = P.LimbDarkLightCurve(0.2, 0.3)
    r = np.array([0.1, 0.01])
    for texp in (None, 0.1):
        a = lc.get_light_curve(r=r, orbit=orbit, t=t, texp=texp)
        b = lc.get_light_curve(r=r, orbit=orbit, t=t, texp=texp, use_in_transit=False)
        assert np.allclose(a, b) and a.min() < 0
    orb = P.KeplerianOrbit(period=10.0, t0=0.3, b=0.4, ecc=0.8, omega=0.7, r_star=1.3, m_star=1.1)
    x, y, z = orb.get_relative_position(np.array([0.3]))
    assert np.allclose(np.sqrt(x ** 2 + y ** 2) / 1.3, 0.4) and z > 0
    u1, u2, s, ror = [0.3, 0.2], [0.4, 0.1], 0.3, 0.08
    t = np.linspace(-6.435, 10.4934, 5000)
    o1 = P.KeplerianOrbit(period=1.543, t0=-0.123)
    o2 = P.KeplerianOrbit(period=o1.period, t0=o1.t0 + 0.5 * o1.period, r_star=ror, m_star=1.0)
    y1 = P.LimbDarkLightCurve(*u1).get_light_curve(orbit=o1, r=ror, t=t)
    y2 = P.LimbDarkLightCurve(*u2).get_light_curve(orbit=o2, r=1.0, t=t)
    y = P.SecondaryEclipseLightCurve(u1, u2, s).get_light_curve(orbit=o1, r=ror, t=t)
    f = ror ** 2 * s
    assert np.allclose((y1 + f * y2) / (1 + f), y, atol=5e-6)


# ---------------------------------------------------------------------------------------------
# celerite in parallel over time: the numpy restatement of the algorithm the HIP kernels run
# (oracle/numpy_port.py, docs/DESIGN_r1_r4.md 3.5) against the sequential recurrence and the dense definition
# ---------------------------------------------------------------------------------------------
def _gp_case(rng, N=180):
    t = np.sort(rng.uniform(0, 30, N))
    y = 0.5 * rng.normal(size=N)
    diag = 0.1 + 0.05 * rng.uniform(size=N)
    co = (np.array([0.3]), np.array([0.2]), np.array([0.5, 0.2]), np.array([0.1, 0.05]),
          np.array([0.3, 0.1]), np.array([2.0, 0.7]))
    return t, y, diag, co


def test_celerite_time_parallel_restatement_matches_sequential_and_dense():
    rng = np.random.default_rng(31)
    t, y, diag, co = _gp_case(rng)
    want = P.celerite_loglike(t, y, diag, co)
    dense, _ = P.gp_loglike_dense(t, y, diag, co)
    assert abs(want - dense) < 1e-10 * abs(dense)
    for n_chunks in (2, 5, 17):
        got = P.celerite_loglike_chunked(t, y, diag, co, n_chunks)
        assert abs(got - want) < 1e-11 * abs(want), n_chunks
    # SHO term (the degenerate case |b d| = a c: Delta0 is the only admissible one)
    co2 = P.sho_coefficients(*P.sho_from_sigma_rho(0.8, 5.0, 0.7), 0.7)
    want2 = P.celerite_loglike(t, y, diag, co2)
    assert abs(P.celerite_loglike_chunked(t, y, diag, co2, 7) - want2) < 1e-11 * abs(want2)


def test_celerite_element_closed_form_likelihood_and_its_gradient():
    """a chunk's log-likelihood as a function of its entering state: closed form from the element ==
    running the recurrences from that state; its gradient w.r.t. (F, P) is what the reverse scan over
    the chunks uses:  d/dF = Y^T (eta - J F),  d/dP = 1/2 (w w^T - J Y)."""
    rng = np.random.default_rng(32)
    t, y, diag, co = _gp_case(rng, N=90)
    _, _, _, V = P.celerite_matrices(t, diag, co)
    n0, n1 = 30, 61
    el = P.celerite_chunk_element(t, y, diag, co, n0, n1)
    J = V.shape[1]
    Dl = P.celerite_delta(co, V[n0])
    assert np.allclose(Dl @ P.celerite_matrices(t, diag, co)[2][n0], V[n0], rtol=1e-13, atol=1e-14)   # Delta U = V

    def run(F, Pm):
        return -0.5 * P.celerite_run_chunk(t, y, diag, co, n0, n1, F, Dl - Pm)

    F0 = 0.1 * rng.normal(size=J)
    B = rng.normal(size=(J, J))
    P0 = 0.3 * Dl + 0.01 * B @ B.T
    const = run(np.zeros(J), np.zeros((J, J)))
    closed = P.celerite_chunk_loglike_closed_form(el, F0, P0, n1 - n0, const)
    assert abs(closed - run(F0, P0)) < 1e-10 * abs(closed)
    # gradient by central differences of the recurrences
    A, b, C, eta, Jm = el
    Y = np.linalg.inv(np.eye(J) + P0 @ Jm)
    w = Y.T @ (eta - Jm @ F0)
    gP = 0.5 * (np.outer(w, w) - Jm @ Y)
    h = 1e-6
    for j in range(J):
        e = np.zeros(J); e[j] = h
        assert abs((run(F0 + e, P0) - run(F0 - e, P0)) / (2 * h) - w[j]) < 1e-6 * (1 + abs(w[j]))
    for j in range(J):
        for l in range(j, J):
            E = np.zeros((J, J)); E[j, l] = E[l, j] = h
            fd = (run(F0, P0 + E) - run(F0, P0 - E)) / (2 * h)
            an = gP[j, l] + gP[l, j] if l != j else gP[j, j]
            assert abs(fd - an) < 1e-5 * (1 + abs(an)), (j, l)
    # the element maps the entering state to the state entering the next chunk
    Fn, Pn = P.celerite_apply_element(el, F0, P0)
    c, a, U, Vv = P.celerite_matrices(t, diag, co)
    S, F = Dl - P0, F0.copy()
    for n in range(n0, n1):
        if n > n0:
            Pp = np.exp(-c * (t[n] - t[n - 1])); S = np.outer(Pp, Pp) * (S + d * np.outer(W, W)); F = Pp * (F + W * z)
        u = S @ U[n]; d = a[n] - U[n] @ u; W = (Vv[n] - u) / d; z = y[n] - U[n] @ F
    Pp = np.exp(-c * (t[n1] - t[n1 - 1])); S = np.outer(Pp, Pp) * (S + d * np.outer(W, W)); F = Pp * (F + W * z)
    assert np.allclose(Fn, F, rtol=1e-10, atol=1e-12)
    assert np.allclose(P.celerite_delta(co, Vv[n1]) - Pn, S, rtol=1e-9, atol=1e-11)


def test_celerite_reverse_scan_over_chunks_by_finite_differences():
    """adjoint of the state entering a chunk, accumulated over all later chunks by the reverse scan,
    against central differences of the log-likelihood of the rest of the series"""
    rng = np.random.default_rng(33)
    t, y, diag, co = _gp_case(rng, N=120)
    _, _, _, V = P.celerite_matrices(t, diag, co)
    J = V.shape[1]
    bounds = [(30, 55), (55, 90), (90, 120)]
    els = [P.celerite_chunk_element(t, y, diag, co, a, b) for a, b in bounds]

    def rest(F, Pm):   # log-likelihood of cadences 30.. given the state entering cadence 30
        acc = 0.0
        for k, (a, b) in enumerate(bounds):
            acc += P.celerite_run_chunk(t, y, diag, co, a, b, F, P.celerite_delta(co, V[a]) - Pm)
            if k + 1 < len(bounds):
                F, Pm = P.celerite_apply_element(els[k], F, Pm)
        return -0.5 * acc

    F0 = 0.1 * rng.normal(size=J)
    B = rng.normal(size=(J, J))
    P0 = 0.3 * P.celerite_delta(co, V[30]) + 0.01 * B @ B.T
    states = [(F0, P0)]
    for k in range(len(bounds) - 1):
        states.append(P.celerite_apply_element(els[k], *states[-1]))
    Fb, Pb = np.zeros(J), np.zeros((J, J))
    for k in reversed(range(len(bounds))):
        Fb, Pb = P.celerite_chunk_adjoint_step(els[k], states[k][0], states[k][1], Fb, Pb)
    h = 1e-6
    for j in range(J):
        e = np.zeros(J); e[j] = h
        fd = (rest(F0 + e, P0) - rest(F0 - e, P0)) / (2 * h)
        assert abs(fd - Fb[j]) < 1e-6 * (1 + abs(Fb[j])), j
    for j in range(J):
        for l in range(j, J):
            E = np.zeros((J, J)); E[j, l] = E[l, j] = h
            fd = (rest(F0, P0 + E) - rest(F0, P0 - E)) / (2 * h)
            an = Pb[j, l] + Pb[l, j] if l != j else Pb[j, j]
            assert abs(fd - an) < 2e-5 * (1 + abs(an)), (j, l)


def test_radial_velocity_is_minus_the_stars_z_velocity():
    """/root/reference/tests/orbits/keplerian_test.py:134-154 on the oracle: the closed form used
    by the fused RV op, with amp = conv sin(i) K0 m_planet, equals -conv d z_star / d t (central
    differences of the oracle's own star position), and its Jacobian equals finite differences"""
    conv = 695700000.0 / 86400.0
    orbit = P.KeplerianOrbit(m_star=1.3, r_star=1.0, t0=np.array([0.5, 3.1]), period=np.array([100.0, 37.3]),
                             ecc=np.array([0.1, 0.45]), omega=np.array([0.5, -2.0]), incl=np.array([0.25 * np.pi, 1.3]),
                             m_planet=np.array([0.1, 0.02]))
    t = np.linspace(0, 100, 700)
    K0 = orbit.n * orbit.a / orbit.m_total / np.sqrt(1 - orbit.ecc ** 2)
    params = np.stack([orbit.n, orbit.t_periastron, orbit.ecc, orbit.cos_omega, orbit.sin_omega,
                       conv * orbit.sin_incl * K0 * orbit.m_planet], axis=-1)[None]
    rv = P.radial_velocity(t, params)[0]
    h = 1e-4
    z = lambda tt: np.stack(orbit._get_position(orbit.a_star, tt))[2]  # noqa: E731
    want = -conv * (z(t + h) - z(t - h)) / (2 * h)
    np.testing.assert_allclose(rv, want, rtol=2e-7, atol=1e-7 * np.abs(want).max())
    # Jacobian of the closed form
    _, J = P.radial_velocity(t, params, jac=True)
    for k in range(P.RV_NPAR):
        up, dn = params.copy(), params.copy()
        step = 1e-6 * max(1.0, abs(params[0, 0, k]))
        up[0, :, k] += step
        dn[0, :, k] -= step
        fd = (P.radial_velocity(t, up) - P.radial_velocity(t, dn)) / (2 * step)
        np.testing.assert_allclose(J[..., k], fd, rtol=1e-5, atol=1e-6 * np.abs(fd).max())


def test_oracle_velocities_are_time_derivatives_of_positions():
    """the property the reference's own test asserts (tests/orbits/keplerian_test.py:91-131, same system): star, planet
    and relative velocities equal d(position)/dt -- here by central differences of the numpy restatement; plus
    get_relative_angles = polar form of the sky-plane relative position (keplerian.py:544-570)"""
    t = np.linspace(0, 100, 1000)
    orbit = P.KeplerianOrbit(m_star=1.3, r_star=1.0, t0=0.5, period=100.0, ecc=0.1, omega=0.5, Omega=1.0, incl=0.25 * np.pi,
                             m_planet=0.1)
    h = 1e-4
    for pos, vel in (("get_star_position", "get_star_velocity"), ("get_planet_position", "get_planet_velocity"),
                     ("get_relative_position", "get_relative_velocity")):
        v = getattr(orbit, vel)(t)
        hi, lo = getattr(orbit, pos)(t + h), getattr(orbit, pos)(t - h)
        for k in range(3):
            fd = (hi[k] - lo[k]) / (2 * h)
            assert np.allclose(v[k], fd, rtol=1e-6, atol=1e-8 * np.abs(fd).max()), (pos, k)
    for vel, acc in (("get_star_velocity", "get_star_acceleration"), ("get_planet_velocity", "get_planet_acceleration"),
                     ("get_relative_velocity", "get_relative_acceleration")):   # keplerian_test.py:158-196
        a = getattr(orbit, acc)(t)
        hi, lo = getattr(orbit, vel)(t + h), getattr(orbit, vel)(t - h)
        for k in range(3):
            fd = (hi[k] - lo[k]) / (2 * h)
            assert np.allclose(a[k], fd, rtol=1e-6, atol=1e-8 * np.abs(fd).max()), (acc, k)
    X, Y, _ = orbit.get_relative_position(t)
    rho, theta = orbit.get_relative_angles(t, parallax=0.05)
    assert np.allclose(rho, np.hypot(X, Y) * 0.05 * P.au_per_R_sun, rtol=1e-14)
    assert np.allclose(theta, np.arctan2(Y, X), rtol=0, atol=1e-14)
    # barycentre: m_star x_star + m_planet x_planet = 0
    xs, xp = orbit.get_star_position(t), orbit.get_planet_position(t)
    for k in range(3):
        assert np.abs(1.3 * xs[k] + 0.1 * xp[k]).max() <= 1e-12 * np.abs(xp[k]).max()
