"""celerite-style Gaussian processes on the batched HIP recurrence.

Interface modelled on celerite2 (the package the reference's users pair it
with: /root/reference/docs/index.rst:14,48-49; setup.py:36):
``terms.SHOTerm`` etc. and ``GaussianProcess(kernel, t=..., diag=...)`` with
``compute`` / ``log_likelihood``.  celerite2 itself is not available here and
the reference tree never calls it, so this part is **parity unpinned** against
celerite2 and checked against the dense-Cholesky definition instead.
"""
from . import terms  # noqa: F401
from .celerite import GaussianProcess, celerite_loglike  # noqa: F401

__all__ = ["terms", "GaussianProcess", "celerite_loglike"]
