#!/usr/bin/env python
"""oracle/ref_glue_check.py -- run the reference's OWN Python glue, in place, against the oracle's restatement of it.

TEST INFRASTRUCTURE (build container only).  Nothing under exoplanet_amd/ imports this; the GPU box never runs it
(/root/reference does not exist there): what travels is the fixture it writes, tests/golden/glue_ref.npz.

What it does.  The reference's orbit / light-curve classes
    /root/reference/src/exoplanet/orbits/keplerian.py     (KeplerianOrbit: rows a1-a3, a5, a6, a11 `_flip`, a12 of SURVEY 8a)
    /root/reference/src/exoplanet/orbits/ttv.py           (TTVOrbit._warp_times: row f2)
    /root/reference/src/exoplanet/light_curves/limb_dark.py, secondary_eclipse.py   (rows a8-a11)
are imported FROM /root/reference, unmodified and uncopied, and EXECUTED on the reference-test systems of SURVEY 8c item (3).
They cannot be imported as they stand: `exoplanet.compat` needs pymc + pytensor + exoplanet_core, `orbits/constants.py` and
`units.py` need astropy -- none installable here.  So this script puts STAND-INS into sys.modules first:
  * `exoplanet.compat`: `tensor` = a numpy-backed module with the ~35 PyTensor functions the glue calls (eager evaluation:
    `pt.switch` = np.where, `set_subtensor` on a recorded index, ...), `ifelse`, `Assert`, and `ops` = the oracle's three Ops
    (oracle/numpy_port.py: kepler, quad_solution_vector, contact_points -- the arithmetic the absent exoplanet_core would do);
  * `astropy.units` / `astropy.constants`: unit objects that only support what the glue does with them when no unit-carrying
    input is passed; `astropy.constants.G.to` raises TypeError, which makes `orbits/constants.py` take ITS OWN literal
    fallback values (constants.py:32-37) -- the constants are the reference's, not ours;
  * `exoplanet.citations`: a no-op `add_citations_to_model` (the real one needs a PyMC model context).
Because of the stand-ins this is NOT the reference's arithmetic end to end, and it does not lift "parity unpinned"
(exoplanet-core and celerite2 stay absent).  What it does replace is "the restatement was read against the reference": every
line of the reference's glue on these systems is executed, and oracle/numpy_port.py must reproduce its outputs to 1e-14 on the
same Ops.  The outputs are stored in tests/golden/glue_ref.npz; tests/test_glue_ref.py holds numpy_port to them on CPU and
tests/test_gpu_glue_ref.py holds the HIP path to them.

Usage:  python oracle/ref_glue_check.py [--write]      (--write: regenerate tests/golden/glue_ref.npz)
"""
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------------------------------------------
# numpy-backed stand-in for pytensor.tensor (eager).  A TV is an ndarray that can carry attributes (units.py decorates
# tensors) and remembers, when it came out of an indexing expression, where it came from (pt.set_subtensor needs that).
# ------------------------------------------------------------------------------------------------------------------
class TV(np.ndarray):
    def __new__(cls, x):
        return np.asarray(x).view(cls)

    def __array_finalize__(self, obj):
        self._parent = None
        self._index = None

    def __getitem__(self, idx):
        out = np.ndarray.__getitem__(self, idx)
        if isinstance(out, np.ndarray):
            out = out.view(TV)
            out._parent, out._index = self, idx
        return out

    @property
    def broadcastable(self):
        return tuple(s == 1 for s in self.shape)

    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (tuple, list)):
            pattern = tuple(pattern[0])
        x = np.asarray(self)
        keep = [p for p in pattern if p != "x"]
        x = np.transpose(x, keep) if keep else x
        idx = tuple(None if p == "x" else slice(None) for p in pattern)
        return TV(x[idx])


def _tv(x):
    return x if isinstance(x, TV) else TV(np.asarray(x))


def make_tensor_module():
    pt = types.ModuleType("exoplanet_compat_tensor_standin")

    def as_tensor_variable(x, **kw):
        if isinstance(x, TV):
            return x
        return TV(np.asarray(x, dtype=np.float64) if not isinstance(x, np.ndarray) else x)

    pt.as_tensor_variable = as_tensor_variable
    pt.TensorVariable = TV
    for name in ("sqrt", "square", "sin", "cos", "arccos", "arcsin", "arctan2", "exp", "log", "floor", "abs", "mod"):
        fn = getattr(np, name)
        setattr(pt, name, (lambda f: lambda *a: _tv(f(*[np.asarray(x) for x in a])))(fn))
    pt.abs_ = pt.abs
    pt.eq = lambda a, b: _tv(np.equal(a, b))
    pt.gt = lambda a, b: _tv(np.greater(a, b))
    pt.ge = lambda a, b: _tv(np.greater_equal(a, b))
    pt.lt = lambda a, b: _tv(np.less(a, b))
    pt.and_ = lambda a, b: _tv(np.logical_and(a, b))
    pt.any = lambda x, axis=None: _tv(np.any(x, axis=axis))
    pt.all = lambda x, axis=None: _tv(np.all(x, axis=axis))
    pt.switch = lambda c, a, b: _tv(np.where(np.asarray(c), np.asarray(a), np.asarray(b)))
    pt.zeros_like = lambda x, dtype=None: _tv(np.zeros_like(np.asarray(x), dtype=dtype))
    pt.ones_like = lambda x, dtype=None: _tv(np.ones_like(np.asarray(x), dtype=dtype))
    pt.arange = lambda *a, **k: _tv(np.arange(*[int(np.asarray(x)) for x in a], **k))
    pt.cast = lambda x, dtype: _tv(np.asarray(x).astype(dtype))
    pt.concatenate = lambda xs, axis=0: _tv(np.concatenate([np.atleast_1d(np.asarray(x)) for x in xs], axis=axis))
    pt.stack = lambda xs, axis=0: _tv(np.stack([np.asarray(x) for x in xs], axis=axis))
    pt.sum = lambda x, axis=None, keepdims=False: _tv(np.sum(np.asarray(x), axis=axis, keepdims=keepdims))
    pt.dot = lambda a, b: _tv(np.dot(np.asarray(a), np.asarray(b)))
    pt.reshape = lambda x, shape, ndim=None: _tv(np.reshape(np.asarray(x), tuple(int(s) for s in shape)))

    def squeeze(x, axis=None):
        # PyTensor's squeeze drops the BROADCASTABLE (length-one) axes -- all of them when no axis is given
        return _tv(np.squeeze(np.asarray(x), axis=axis))

    pt.squeeze = squeeze

    def shape_padright(x, n_ones=1):
        x = np.asarray(x)
        return _tv(x.reshape(x.shape + (1,) * int(n_ones)))

    def shape_padleft(x, n_ones=1):
        x = np.asarray(x)
        return _tv(x.reshape((1,) * int(n_ones) + x.shape))

    pt.shape_padright, pt.shape_padleft = shape_padright, shape_padleft

    def set_subtensor(sub, y):
        if not isinstance(sub, TV) or sub._parent is None:
            raise TypeError("set_subtensor: the first argument must be an indexing expression x[idx]")
        out = np.array(sub._parent, copy=True)
        out[sub._index] = np.asarray(y)
        return _tv(out)

    pt.set_subtensor = set_subtensor
    extra = types.ModuleType("extra_ops")
    extra.searchsorted = lambda a, v, side="left", sorter=None: _tv(np.searchsorted(np.asarray(a), np.asarray(v), side=side))
    pt.extra_ops = extra
    return pt


def install_standins():
    """sys.modules entries that let the reference's glue import; returns the `exoplanet` namespace module"""
    from oracle import numpy_port as P

    if not os.path.isdir(os.path.join(REF_SRC, "exoplanet")):
        raise RuntimeError("the reference tree is not here (this script runs in the build container only)")

    pt = make_tensor_module()

    # exoplanet.compat
    compat = types.ModuleType("exoplanet.compat")
    compat.tensor = pt
    compat.USING_PYMC3 = False
    compat.pm = None

    def ifelse(cond, a, b):
        return a if bool(np.asarray(cond)) else b

    class Assert:
        def __init__(self, msg=""):
            self.msg = msg

        def __call__(self, value, *conds):
            for c in conds:
                if not bool(np.all(np.asarray(c))):
                    raise AssertionError(self.msg)
            return value

    ops = types.SimpleNamespace()

    def kepler(M, ecc):
        M, ecc = np.broadcast_arrays(np.asarray(M, dtype=np.float64), np.asarray(ecc, dtype=np.float64))
        s, c = P.kepler(M.ravel(), ecc.ravel())
        return _tv(s.reshape(M.shape)), _tv(c.reshape(M.shape))

    def quad_solution_vector(b, r):
        b, r = np.broadcast_arrays(np.asarray(b, dtype=np.float64), np.asarray(r, dtype=np.float64))
        s = P.quad_solution_vector(b.ravel(), r.ravel())[0]          # (s, ds/db, ds/dr): the glue contracts s with c
        return _tv(np.asarray(s).reshape(b.shape + (3,)))

    def contact_points(a, e, cosw, sinw, cosi, sini, L):
        args = np.broadcast_arrays(*[np.asarray(x, dtype=np.float64) for x in (a, e, cosw, sinw, cosi, sini, L)])
        shape = args[0].shape
        Ml, Mr, flag = P.contact_points(*[x.ravel() for x in args])
        return _tv(np.asarray(Ml).reshape(shape)), _tv(np.asarray(Mr).reshape(shape)), _tv(np.asarray(flag).reshape(shape))

    ops.kepler, ops.quad_solution_vector, ops.contact_points = kepler, quad_solution_vector, contact_points
    compat.ops, compat.ifelse, compat.Assert = ops, ifelse, Assert
    compat.function = compat.grad = compat.verify_grad = compat.change_flags = None

    # astropy: only what the glue touches without unit-carrying inputs; G.to raising TypeError selects the reference's own
    # literal constants (orbits/constants.py:26-37)
    class Unit:
        def __init__(self, name):
            self.name = name

        def _bin(self, other, op):
            return Unit(f"({self.name}{op}{getattr(other, 'name', other)})")

        def __mul__(self, o): return self._bin(o, "*")
        def __rmul__(self, o): return self._bin(o, "*")
        def __truediv__(self, o): return self._bin(o, "/")
        def __rtruediv__(self, o): return Unit(f"({o}/{self.name})")
        def __pow__(self, o): return self._bin(o, "**")

        def to(self, other, *a, **k):
            raise TypeError("astropy stand-in: no unit conversions (oracle/ref_glue_check.py)")

    astropy = types.ModuleType("astropy")
    units = types.ModuleType("astropy.units")
    for nm in ("R_sun", "M_sun", "day", "g", "cm", "au", "m", "s", "kg", "yr", "R_earth", "M_earth", "R_jup", "M_jup", "rad", "deg", "arcsec"):
        setattr(units, nm, Unit(nm))
    constants = types.ModuleType("astropy.constants")
    constants.G = Unit("G")
    constants.c = Unit("c")
    astropy.units, astropy.constants = units, constants

    citations = types.ModuleType("exoplanet.citations")
    citations.add_citations_to_model = lambda *a, **k: None
    citations.CITATIONS = {}

    # the package objects: namespaces whose __path__ points INTO the reference tree -- their modules are loaded from there,
    # their __init__.py (which pulls in PyMC distributions, estimators, ...) are not executed
    def namespace(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_SRC, rel)]
        m.__package__ = name
        return m

    pkg = namespace("exoplanet", "exoplanet")
    orbits = namespace("exoplanet.orbits", "exoplanet/orbits")
    lcs = namespace("exoplanet.light_curves", "exoplanet/light_curves")
    pkg.compat, pkg.citations, pkg.orbits, pkg.light_curves = compat, citations, orbits, lcs
    sys.modules.update({
        "exoplanet": pkg, "exoplanet.compat": compat, "exoplanet.citations": citations, "exoplanet.orbits": orbits,
        "exoplanet.light_curves": lcs, "astropy": astropy, "astropy.units": units, "astropy.constants": constants,
    })
    return pkg


def load_reference():
    """-> (module keplerian, module ttv, module limb_dark, module secondary_eclipse, module constants), executed from
    /root/reference in place"""
    import importlib

    install_standins()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)      # constants.py announces its fallback
        const = importlib.import_module("exoplanet.orbits.constants")
    kep = importlib.import_module("exoplanet.orbits.keplerian")
    ttv = importlib.import_module("exoplanet.orbits.ttv")
    ld = importlib.import_module("exoplanet.light_curves.limb_dark")
    sec = importlib.import_module("exoplanet.light_curves.secondary_eclipse")
    for m in (const, kep, ttv, ld, sec):
        assert os.path.abspath(m.__file__).startswith(REF_SRC), m.__file__      # the reference's files, nothing else
    return kep, ttv, ld, sec, const


# ------------------------------------------------------------------------------------------------------------------
# The systems (the reference's own test systems + the BASELINE configs at reduced N) and what is recorded
# ------------------------------------------------------------------------------------------------------------------
def systems():
    """name -> dict(orbit kwargs, r, u, t, texp, oversample, order, secondary (u_s, sbr) | None, ttv | None)"""
    S = {}
    t40 = np.linspace(-20, 20, 1000)
    # tests/light_curves_test.py:75-102 -- two planets, e = (0.1, 0.8)
    two = dict(period=np.array([10.0, 5.3]), t0=np.array([0.0, 3.2]), ecc=np.array([0.1, 0.8]), omega=np.array([0.5, 1.3]),
               b=np.array([0.2, 0.5]), m_star=1.3, r_star=1.1)
    S["two_planet"] = dict(orbit=two, r=np.array([0.1, 0.01]), u=(0.3, 0.2), t=t40, texp=None)
    S["two_planet_texp"] = dict(orbit=two, r=np.array([0.1, 0.01]), u=(0.3, 0.2), t=t40, texp=0.1, oversample=7, order=0)
    S["two_planet_texp_o1"] = dict(orbit=two, r=np.array([0.1, 0.01]), u=(0.3, 0.2), t=t40, texp=0.1, oversample=5, order=1)
    S["two_planet_texp_o2"] = dict(orbit=two, r=np.array([0.1, 0.01]), u=(0.3, 0.2), t=t40, texp=0.1, oversample=9, order=2)
    # tests/light_curves_test.py:148-164 -- P = 3.456, e = 0.6, omega = -1.5, texp = 0.02
    S["e06"] = dict(orbit=dict(period=np.array([3.456]), t0=np.array([0.45]), ecc=np.array([0.6]), omega=np.array([-1.5]),
                               b=np.array([0.4]), m_star=1.2, r_star=0.9),
                    r=np.array([0.03]), u=(0.3, 0.2), t=np.linspace(-5, 5, 2000), texp=0.02)
    # tests/light_curves_test.py:167-193 -- the small star (M dwarf)
    S["small_star"] = dict(orbit=dict(period=np.array([2.1]), t0=np.array([0.3]), b=np.array([0.1]), m_star=0.151, r_star=0.189),
                           r=np.array([0.189 * 0.1]), u=(0.2, 0.3), t=np.linspace(-1.5, 1.5, 1200), texp=None)
    # tests/light_curves_test.py:285-311 -- secondary eclipse, P = 1.543
    S["secondary"] = dict(orbit=dict(period=np.array([1.543]), t0=np.array([0.123]), ecc=np.array([0.1]), omega=np.array([0.4]),
                                     b=np.array([0.3]), m_star=1.0, r_star=1.0),
                          r=np.array([0.08]), u=(0.3, 0.2), t=np.linspace(-2.0, 2.0, 1500), texp=None,
                          secondary=((0.4, 0.1), 0.3))
    S["secondary_circular_texp"] = dict(orbit=dict(period=np.array([1.543]), t0=np.array([0.123]), b=np.array([0.3]),
                                                   m_star=1.0, r_star=1.0),
                                        r=np.array([0.08]), u=(0.3, 0.2), t=np.linspace(-2.0, 2.0, 1500), texp=0.05,
                                        oversample=7, order=0, secondary=((0.4, 0.1), 0.3))
    # BASELINE C1 / C2 / C4 / C5 at N = 2048
    cad = 2.0 / 1440.0
    S["c1"] = dict(orbit=dict(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3])), r=np.array([0.1]), u=(0.3, 0.2),
                   t=1.0 - 1024 * cad + np.arange(2048) * cad, texp=None)
    S["c2"] = dict(orbit=dict(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3]), ecc=np.array([0.3]), omega=np.array([1.1])),
                   r=np.array([0.1]), u=(0.3, 0.2), t=1.0 - 1024 * cad + np.arange(2048) * cad, texp=None)
    S["c4"] = dict(orbit=dict(period=np.array([3.5, 7.9, 13.1, 29.7]), t0=np.array([1.0, 2.3, 5.1, 11.7]), b=np.array([0.3, 0.1, 0.5, 0.2]),
                              ecc=np.array([0.05, 0.1, 0.2, 0.3]), omega=np.array([1.1, -0.4, 2.0, 0.3])),
                   r=np.array([0.1, 0.05, 0.07, 0.03]), u=(0.3, 0.2), t=np.arange(2048) * (12.0 / 2048), texp=None)
    lc = 29.4 / 1440.0
    S["c5"] = dict(orbit=dict(period=np.array([2.7]), t0=np.array([0.6]), b=np.array([0.3]), ecc=np.array([0.1]), omega=np.array([0.4])),
                   r=np.array([0.08]), u=(0.3, 0.2), t=np.arange(2048) * lc * 0.15, texp=lc, oversample=7, order=0,
                   secondary=((0.4, 0.1), 0.3))
    # light travel time (keplerian.py:411-470)
    S["light_delay"] = dict(orbit=dict(period=np.array([3.5]), t0=np.array([1.0]), b=np.array([0.3]), ecc=np.array([0.3]), omega=np.array([1.1]),
                                       m_star=1.1, r_star=0.95),
                            r=np.array([0.1]), u=(0.3, 0.2), t=1.0 - 600 * cad + np.arange(1200) * cad, texp=None, light_delay=True)
    # duration parameterisation, circular (keplerian.py:112-131) and eccentric (:237-260)
    S["duration_circ"] = dict(orbit=dict(period=np.array([4.2]), t0=np.array([0.7]), b=np.array([0.35]), duration=np.array([0.12]),
                                         ror=np.array([0.07]), r_star=1.0),
                              r=np.array([0.07]), u=(0.3, 0.2), t=np.linspace(0.0, 1.4, 900), texp=None)
    S["duration_ecc"] = dict(orbit=dict(period=np.array([4.2]), t0=np.array([0.7]), duration=np.array([0.08]), ror=np.array([0.07]),
                                        ecc=np.array([0.25]), omega=np.array([0.9]), r_star=1.0, m_star=1.0),
                             r=np.array([0.07]), u=(0.3, 0.2), t=np.linspace(0.0, 1.4, 900), texp=None)
    # a, incl, rho_star parameterisations (keplerian.py:849-934)
    S["a_incl"] = dict(orbit=dict(a=np.array([12.3]), t0=np.array([0.2]), incl=np.array([1.5]), ecc=np.array([0.2]), omega=np.array([-0.7]),
                                  m_star=0.9, r_star=1.05),
                       r=np.array([0.09]), u=(0.25, 0.3), t=np.linspace(-0.5, 0.9, 800), texp=None)
    S["rho_star"] = dict(orbit=dict(period=np.array([6.1]), t0=np.array([0.5]), b=np.array([0.6]), rho_star=1.7, r_star=0.8),
                         r=np.array([0.05]), u=(0.4, 0.2), t=np.linspace(0.0, 1.0, 800), texp=None)
    # t_periastron instead of t0, with Omega (keplerian.py:267-281,316-322)
    S["tperi_Omega"] = dict(orbit=dict(period=np.array([5.5]), t_periastron=np.array([0.4]), b=np.array([0.2]), ecc=np.array([0.4]),
                                       omega=np.array([2.2]), Omega=np.array([0.8]), m_planet=np.array([0.002]), m_star=1.0, r_star=1.0),
                            r=np.array([0.1]), u=(0.3, 0.2), t=np.linspace(-3.0, 3.0, 1500), texp=None)
    # timing variations (ttv.py:71-187): per-transit offsets, and transit times given directly
    rng = np.random.default_rng(42)
    S["ttv_offsets"] = dict(orbit=dict(period=np.array([3.1, 7.4]), t0=np.array([0.5, 1.2]), b=np.array([0.3, 0.4]),
                                       ecc=np.array([0.1, 0.2]), omega=np.array([0.3, -1.0])),
                            r=np.array([0.08, 0.05]), u=(0.3, 0.2), t=np.linspace(0.0, 30.0, 3000), texp=None,
                            ttv=dict(ttvs=[0.01 * rng.normal(size=10), 0.02 * rng.normal(size=5)]))
    tt0 = 0.5 + 3.1 * np.arange(10) + 0.01 * rng.normal(size=10)
    tt1 = 1.2 + 7.4 * np.arange(5) + 0.02 * rng.normal(size=5)
    S["ttv_times"] = dict(orbit=dict(b=np.array([0.3, 0.4]), ecc=np.array([0.1, 0.2]), omega=np.array([0.3, -1.0])),
                          r=np.array([0.08, 0.05]), u=(0.3, 0.2), t=np.linspace(0.0, 30.0, 3000), texp=0.02, oversample=3, order=0,
                          ttv=dict(transit_times=[tt0, tt1]))
    return S


FULL_VECTORS = ("two_planet", "tperi_Omega", "a_incl")
ORBIT_ATTRS = ("period", "a", "n", "t0", "t_periastron", "tref", "M0", "incl", "cos_incl", "sin_incl", "b", "ecc", "cos_omega",
               "sin_omega", "a_star", "a_planet", "K0", "m_star", "r_star", "rho_star", "m_planet", "m_total")


def evaluate(mods, name, spec):
    """run one system through a (KeplerianOrbit, TTVOrbit, LimbDarkLightCurve, SecondaryEclipseLightCurve) implementation;
    -> dict of arrays"""
    Kep, TTV, LD, Sec = mods
    out = {}
    kw = dict(spec["orbit"])
    if spec.get("ttv"):
        orbit = TTV(**kw, **spec["ttv"])
    else:
        orbit = Kep(**kw)
    for a in ORBIT_ATTRS:
        v = getattr(orbit, a, None)
        if v is not None:
            out["orbit_" + a] = np.atleast_1d(np.asarray(v, dtype=np.float64))
    t, r = spec["t"], spec["r"]
    ld = bool(spec.get("light_delay", False))
    pos = orbit.get_relative_position(t, light_delay=ld)
    texp = spec.get("texp")
    inds = np.asarray(orbit.in_transit(t, r=r, texp=texp), dtype=np.int64)
    # (kept at the in-transit cadences and every eighth one: the fixture stays small)
    keep = np.union1d(inds, np.arange(0, t.size, 8))
    out["relpos_idx"] = keep
    for i, c in enumerate("xyz"):
        out["relpos_" + c] = np.asarray(pos[i], dtype=np.float64)[keep]
    if name in FULL_VECTORS:          # (positions / velocities of every body: a few systems keep the fixture small)
        for nm in ("get_planet_position", "get_star_position", "get_planet_velocity", "get_star_velocity", "get_relative_velocity"):
            res = getattr(orbit, nm)(t)
            for i, c in enumerate("xyz"):
                out[f"{nm[4:]}_{c}"] = np.asarray(res[i], dtype=np.float64)[keep]
        # (get_radial_velocity without K is -conv * get_star_velocity(t)[2] with an astropy unit conversion, keplerian.py:672-677:
        # the star's velocity above is its whole content)
    out["in_transit"] = inds
    lckw = dict(orbit=orbit, r=r, t=t, texp=texp, oversample=spec.get("oversample", 7), order=spec.get("order", 0))
    if spec.get("secondary"):
        us, sbr = spec["secondary"]
        star = Sec(spec["u"], us, sbr)
        orb2 = orbit._flip(r)
        for a in ("t_periastron", "t0", "cos_omega", "sin_omega", "m_star", "r_star", "m_planet", "a", "cos_incl", "sin_incl", "incl", "b"):
            v = getattr(orb2, a, None)
            if v is not None:
                out["flip_" + a] = np.atleast_1d(np.asarray(v, dtype=np.float64))
    else:
        star = LD(*spec["u"])
        out["cl"] = np.asarray(star.c, dtype=np.float64)
    for uit in ((True, False) if not ld else (False,)):
        lc = star.get_light_curve(use_in_transit=uit, light_delay=ld, **lckw)
        out["lc_in_transit" if uit else "lc_full"] = np.asarray(lc, dtype=np.float64)
    return out


def extras(mods):
    """pieces of the glue the system loop does not reach, evaluated through an implementation's classes: the approximate-depth
    inversion and its Jacobian (limb_dark.py:68-97; reference test tests/light_curves_test.py:257-282), the duration -> a
    Jacobians of the circular `duration` parameterisation (keplerian.py:112-131,151-170), the Jacobian of `b` (:217-228)"""
    Kep, _, LD, _ = mods
    out = {}
    delta = np.array([1e-3, 4e-3, 9e-3, 1.6e-2, 2.5e-2])
    b = np.array([0.0, 0.2, 0.45, 0.7, 0.9])
    ror, jac = LD(0.3, 0.2).get_ror_from_approx_transit_depth(delta, b, jac=True)
    out["ror"], out["ror_jac"] = np.asarray(ror, dtype=np.float64), np.asarray(jac, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orbit = Kep(period=np.array([4.2, 9.1]), t0=np.array([0.7, 1.3]), b=np.array([0.35, 0.1]), duration=np.array([0.12, 0.2]),
                    ror=np.array([0.07, 0.03]), r_star=1.1, m_planet=np.array([1e-3, 3e-4]))
    for k in ("a", "a_star", "a_planet", "rho_star"):
        out["jac_duration_" + k] = np.atleast_1d(np.asarray(orbit.jacobians["duration"][k], dtype=np.float64))
    out["duration_a"] = np.atleast_1d(np.asarray(orbit.a, dtype=np.float64))
    orbit2 = Kep(period=np.array([3.3]), t0=np.array([0.2]), b=np.array([0.4]), ecc=np.array([0.3]), omega=np.array([0.8]))
    out["jac_b_cos_incl"] = np.atleast_1d(np.asarray(orbit2.jacobians["b"]["cos_incl"], dtype=np.float64))
    return out


def extras_simple(Simple, LD):
    """orbits/simple.py: SimpleTransitOrbit (period, duration, t0, b, r_star, ror) -- positions, the in-transit selection and the
    light curve LimbDarkLightCurve draws from it (with and without an exposure time)"""
    out = {}
    t = np.linspace(-1.0, 9.0, 1500)
    orbit = Simple(period=np.array([3.3, 4.7]), duration=np.array([0.12, 0.2]), t0=np.array([0.3, 1.1]), b=np.array([0.2, 0.5]),
                   r_star=1.1, ror=np.array([0.08, 0.05]))
    r = 1.1 * np.array([0.08, 0.05])
    pos = orbit.get_relative_position(t)
    for i, c in enumerate("xyz"):
        out["simple_pos_" + c] = np.asarray(pos[i], dtype=np.float64)[::5]
    out["simple_in_transit"] = np.asarray(orbit.in_transit(t, r=r), dtype=np.int64)
    out["simple_in_transit_texp"] = np.asarray(orbit.in_transit(t, r=r, texp=0.05), dtype=np.int64)
    star = LD(0.3, 0.2)
    out["simple_lc"] = np.asarray(star.get_light_curve(orbit=orbit, r=r, t=t), dtype=np.float64)
    out["simple_lc_texp"] = np.asarray(star.get_light_curve(orbit=orbit, r=r, t=t, texp=0.05, oversample=5, order=1), dtype=np.float64)
    assert out["simple_lc"].min() < -1e-3 and 0 < out["simple_in_transit"].size < t.size
    return out


def reference_simple():
    import importlib

    install_standins()
    m = importlib.import_module("exoplanet.orbits.simple")
    assert os.path.abspath(m.__file__).startswith(REF_SRC)
    return m.SimpleTransitOrbit


def reference_impl():
    kep, ttv, ld, sec, _ = load_reference()
    return kep.KeplerianOrbit, ttv.TTVOrbit, ld.LimbDarkLightCurve, sec.SecondaryEclipseLightCurve


def port_impl():
    from oracle import numpy_port as P

    return P.KeplerianOrbit, P.TTVOrbit, P.LimbDarkLightCurve, P.SecondaryEclipseLightCurve


def compare(ref, got, tol=1e-14):
    """max over keys of |ref - got| / max(1, |ref|); shapes and index arrays exactly"""
    worst, where = 0.0, None
    for k, a in ref.items():
        if k not in got:
            raise AssertionError(f"missing {k}")
        b = got[k]
        if a.dtype.kind in "iu":
            if a.shape != b.shape or not np.array_equal(a, b):
                raise AssertionError(f"{k}: index arrays differ")
            continue
        if np.squeeze(a).shape != np.squeeze(b).shape:
            raise AssertionError(f"{k}: shape {a.shape} vs {b.shape}")
        d = np.abs(np.squeeze(a) - np.squeeze(b)) / np.maximum(1.0, np.abs(np.squeeze(a)))
        m = float(np.max(d)) if d.size else 0.0
        if m > worst:
            worst, where = m, k
    return worst, where


def main(argv):
    write = "--write" in argv
    ref_mods, port_mods = reference_impl(), port_impl()
    import importlib

    const = importlib.import_module("exoplanet.orbits.constants")
    store = {"const_G_grav": np.array(const.G_grav), "const_gcc_per_sun": np.array(const.gcc_per_sun),
             "const_au_per_R_sun": np.array(const.au_per_R_sun), "const_c_light": np.array(const.c_light)}
    worst_all = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, spec in systems().items():
            ref = evaluate(ref_mods, name, spec)
            got = evaluate(port_mods, name, spec)
            w, where = compare(ref, got)
            worst_all = max(worst_all, w)
            depth = -min(float(np.min(ref[k])) for k in ref if k.startswith("lc_"))
            n_in = int(ref["in_transit"].size)
            assert depth > 1e-5 and 0 < n_in < spec["t"].size, (name, depth, n_in)        # no vacuous system
            print(f"{name:26s} {len(ref):3d} arrays, depth {depth:.2e}, {n_in:4d} of {spec['t'].size} cadences in transit:  "
                  f"max rel. difference numpy_port vs reference glue = {w:.2e}  ({where})")
            for k, v in ref.items():
                store[f"{name}__{k}"] = v
        for k, v in extras_simple(reference_simple(), ref_mods[2]).items():
            store[f"extras__{k}"] = v
        for k, v in extras(ref_mods).items():       # (reference side only: oracle/numpy_port.py has no counterpart -- the product is
            store[f"extras__{k}"] = v               #  held to these directly, tests/test_gpu_glue_ref.py)
    print(f"worst = {worst_all:.2e}")
    if worst_all > 1e-13:
        print("FAIL: oracle/numpy_port.py does not reproduce the reference's glue")
        return 1
    if write:
        path = os.path.join(ROOT, "tests", "golden", "glue_ref.npz")
        np.savez_compressed(path, **store)
        print("wrote", os.path.relpath(path, ROOT), f"({os.path.getsize(path) / 1024:.0f} KB, {len(store)} arrays)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
