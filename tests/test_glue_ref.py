"""CPU: oracle/numpy_port.py against tests/golden/glue_ref.npz -- the outputs of the reference's OWN Python glue
(orbits/keplerian.py, orbits/ttv.py, light_curves/limb_dark.py, secondary_eclipse.py), executed in place from /root/reference
by oracle/ref_glue_check.py (numpy stand-ins for PyTensor / astropy, the oracle's Ops where exoplanet_core would be; that
script's docstring says what this does and does not pin).  20 systems: the reference's test systems, the BASELINE configs at
N = 2048, every parameterisation of the constructor, light delay, timing variations.  Tolerance 1e-14 relative to
max(1, |value|); index arrays exactly."""
import os
import warnings

import numpy as np
import pytest

from oracle import ref_glue_check as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_ref.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _ref_of(gold, name):
    pre = name + "__"
    return {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}


def test_fixture_covers_every_system(gold):
    names = {k.split("__")[0] for k in gold.files if "__" in k} - {"extras"}
    assert names == set(G.systems()), names ^ set(G.systems())
    assert {"extras__ror", "extras__ror_jac", "extras__jac_duration_a", "extras__jac_b_cos_incl"} <= set(gold.files)
    # the constants are the reference's own literals (orbits/constants.py:32-37), reached through its fallback branch
    assert float(gold["const_G_grav"]) == 2942.2062175044193
    assert float(gold["const_c_light"]) == 37231.66360672704
    assert float(gold["const_gcc_per_sun"]) == 5.905271918964842


@pytest.mark.parametrize("name", sorted(G.systems()))
def test_numpy_port_reproduces_reference_glue(gold, name):
    spec = G.systems()[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = G.evaluate(G.port_impl(), name, spec)
    ref = _ref_of(gold, name)
    assert set(ref) == set(got), set(ref) ^ set(got)
    # no vacuous system: a transit is there and the in-transit selection is a proper subset
    assert min(float(ref[k].min()) for k in ref if k.startswith("lc_")) < -1e-5
    assert 0 < ref["in_transit"].size < spec["t"].size
    worst, where = G.compare(ref, got)
    assert worst <= 1e-14, (name, where, worst)


def test_constants_of_the_port_are_the_references(gold):
    from oracle import numpy_port as P

    assert P.G_grav == float(gold["const_G_grav"])
    assert P.c_light == float(gold["const_c_light"])
    assert P.gcc_per_sun == float(gold["const_gcc_per_sun"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/exoplanet"), reason="the reference tree is only in the build container")
def test_live_reference_glue_matches_fixture_and_port(gold):
    """where the reference tree exists: import its glue in place again and hold BOTH the committed fixture and numpy_port to it
    (the fixture cannot go stale against the reference unnoticed)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (a subprocess: the stand-ins go into sys.modules under the names `exoplanet`, `astropy`)
    code = ("import sys, warnings; sys.path.insert(0, %r); warnings.simplefilter('ignore');"
            "import numpy as np; from oracle import ref_glue_check as G; mods = G.reference_impl();"
            "gold = np.load(%r); worst = 0.0\n"
            "for name, spec in G.systems().items():\n"
            "    ref = G.evaluate(mods, name, spec)\n"
            "    fx = {k[len(name) + 2:]: gold[k] for k in gold.files if k.startswith(name + '__')}\n"
            "    assert set(fx) == set(ref), name\n"
            "    w, where = G.compare(ref, fx, 0.0); worst = max(worst, w)\n"
            "ex = dict(G.extras(mods)); ex.update(G.extras_simple(G.reference_simple(), mods[2]))\n"
            "for k, v in ex.items():\n"
            "    assert np.array_equal(v, gold['extras__' + k]), k\n"
            "print('WORST', worst)") % (root, GOLD)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.strip().split("WORST")[-1])
    assert worst == 0.0, worst        # the fixture IS the reference glue's output, bit for bit
