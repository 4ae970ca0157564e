#!/usr/bin/env python
"""oracle/run_reference_tests.py -- run the reference's OWN test files, in place, on the oracle's Ops.

TEST INFRASTRUCTURE (build container only; nothing under exoplanet_amd/ imports this, the GPU box never runs it).

/root/reference/tests/{orbits/keplerian_test.py, orbits/ttv_test.py, orbits/simple_test.py, light_curves_test.py} are collected by
pytest FROM /root/reference, unmodified and uncopied, after oracle/ref_glue_check.py's stand-ins are in sys.modules: the reference's
glue modules come from /root/reference/src, `exoplanet.compat.tensor` is the eager numpy stand-in, `exoplanet.compat.ops` the
oracle's three Ops (oracle/numpy_port.py) -- so what the reference's tests ASSERT (in-transit selection == geometric test, flipped
orbit == swapped bodies, impact parameter at t0, continuity at the solution vector's singular points, secondary-eclipse blend, the
approximate-depth formula, TTV warps, SimpleTransitOrbit == KeplerianOrbit, ...) is asserted of the oracle's Ops behind the
reference's own glue.  Two more stand-ins for this: `compat.function([], outputs)` evaluates eagerly (a callable returning numpy
arrays), and tensors have `.eval()`.  Tests that need a SYMBOLIC graph (pt.dvector() inputs, `grad`, `verify_grad`), real astropy unit
conversions, or starry / batman are deselected by name below, with the reason; everything else must pass.  Like the glue check, this
does not lift "parity unpinned": exoplanet-core's arithmetic is absent.

Usage:  python oracle/run_reference_tests.py            (exit status 0 = all selected reference tests passed)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REF = "/root/reference"
FILES = ["tests/orbits/keplerian_test.py", "tests/orbits/ttv_test.py", "tests/orbits/simple_test.py", "tests/light_curves_test.py"]
# deselected by name: (test function, why) -- everything else in the four files must pass
DESELECT = [
    ("test_sky_coords", "needs batman"),
    ("test_small_star", "needs batman (both files)"),
    ("test_light_curve", "needs starry"),
    ("test_velocity", "symbolic graph: grad with respect to a pt.dvector() of times"),
    ("test_acceleration", "symbolic graph: grad with respect to a pt.dvector() of times"),
    ("test_light_delay", "symbolic graph (change_flags, grad); its seven *_shape_* siblings run"),
    ("test_get_aor_from_transit_duration", "symbolic graph: grad"),
    ("test_jacobians", "symbolic graph: grad"),
    ("test_light_curve_grad", "verify_grad (symbolic)"),
    ("test_vector_params", "symbolic inputs: pt.vector()"),
    ("test_singular_points", "symbolic inputs: function([b, r], ...)"),
    ("test_approx_transit_depth", "ends in a symbolic grad (its numeric half passes before that line)"),
    ("test_radial_velocity", "astropy unit conversion"),
    ("test_consistent_coords", "astropy constants"),
    ("test_get_consistent_inputs", "astropy unit conversion"),
    ("test_no_ttvs", "astropy unit conversion (get_radial_velocity)"),
]


def main(argv):
    import numpy as np
    import pytest

    from oracle import ref_glue_check as G

    G.install_standins()
    compat = sys.modules["exoplanet.compat"]

    def function(inputs, outputs, **kw):
        if inputs:
            raise NotImplementedError("symbolic inputs: the stand-in evaluates eagerly")
        as_np = lambda x: np.asarray(x)  # noqa: E731
        if isinstance(outputs, (list, tuple)):
            return lambda: [as_np(o) for o in outputs]
        return lambda: as_np(outputs)

    def no_graph(*a, **k):
        raise NotImplementedError("symbolic graph: not in the eager stand-in")

    compat.function = function
    compat.grad = no_graph
    compat.verify_grad = no_graph
    G.TV.eval = lambda self: np.asarray(self)

    class Shape(tuple):                  # `x.shape.eval()` in the reference's tests
        def eval(self):
            return np.array(self)

    G.TV.shape = property(lambda self: Shape(np.ndarray.shape.__get__(self)))
    pt = compat.tensor
    pt.dvector = pt.dscalar = pt.dmatrix = no_graph
    # `import exoplanet as xo` in the tests: the package namespace with what they use
    import importlib

    pkg = sys.modules["exoplanet"]
    pkg.orbits.KeplerianOrbit = importlib.import_module("exoplanet.orbits.keplerian").KeplerianOrbit
    pkg.orbits.TTVOrbit = importlib.import_module("exoplanet.orbits.ttv").TTVOrbit
    pkg.orbits.ttv = importlib.import_module("exoplanet.orbits.ttv")
    pkg.orbits.SimpleTransitOrbit = importlib.import_module("exoplanet.orbits.simple").SimpleTransitOrbit
    ld = importlib.import_module("exoplanet.light_curves.limb_dark")
    sec = importlib.import_module("exoplanet.light_curves.secondary_eclipse")
    pkg.light_curves.LimbDarkLightCurve = ld.LimbDarkLightCurve
    pkg.light_curves.SecondaryEclipseLightCurve = sec.SecondaryEclipseLightCurve
    pkg.LimbDarkLightCurve, pkg.SecondaryEclipseLightCurve = ld.LimbDarkLightCurve, sec.SecondaryEclipseLightCurve
    # (names are matched exactly: `test_light_delay` must not take `test_light_delay_shape_*` with it)
    names = {n for n, _ in DESELECT}

    class Select:
        def pytest_collection_modifyitems(self, config, items):
            keep, drop = [], []
            for it in items:
                (drop if it.name in names else keep).append(it)
            config.hook.pytest_deselected(items=drop)
            items[:] = keep

    args = ["-q", "-p", "no:cacheprovider", "--rootdir", "/tmp", "-c", "/dev/null", "-W", "ignore"]
    args += [os.path.join(REF, f) for f in FILES] + list(argv)
    return pytest.main(args, plugins=[Select()])


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
