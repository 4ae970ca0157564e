"""oracle/ -- TEST INFRASTRUCTURE ONLY. Never imported by the product (exoplanet_amd/).

CPU restatement of the reference's per-leapfrog-step log-likelihood hot path
(SURVEY.md section 8): Kepler solve -> quadratic limb-darkened transit flux ->
celerite GP log-likelihood, value + gradient.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import, link or execute anything in this directory, and there only as
the checker / reported baseline -- never as the thing measured or shipped.

PARITY STATUS: **parity unpinned** against the reference implementation.
The arithmetic of this path lives in two third-party packages that are NOT in
/root/reference and not installed here: ``exoplanet-core`` (>=0.3.0; call
sites src/exoplanet/orbits/keplerian.py:333,744-753,818 and
src/exoplanet/light_curves/limb_dark.py:24) and ``celerite2`` (>=0.3.1, no
call site in the reference tree).  The reference's own tests hold no literal
golden vectors for this path and the reference cannot be
imported here (no pytensor / pymc / exoplanet_core).  There is also no native
reference source to build, so there is no ``oracle/_ref``.

The oracle is therefore pinned *mathematically*:
  * ``mp_reference.py``   -- arbitrary precision (mpmath) evaluation of the
    DEFINITIONS (root of Kepler's equation, disk integrals of the Green's
    basis, dense-Cholesky Gaussian likelihood).  Source of tests/golden/*.npz.
  * ``numpy_port.py``     -- float64 restatement of the published algorithms
    and of the reference's Python glue (each function cites the reference
    file:line it follows).  Checked against mp_reference in tests/.
  * ``c/``                -- plain C port of numpy_port (fast enough for the
    full-size configs; timed as bench.py's ``cpu_baseline`` kind "port").
"""
