"""exoplanet_amd -- MI355X-native per-leapfrog-step log-likelihood hot path of
exoplanet-dev/exoplanet (Kepler solve -> limb-darkened transit flux -> celerite
GP log-likelihood, value + gradient), as HIP kernels behind the reference's own
operator interface.  See DESIGN.md."""
from . import ops  # noqa: F401
from . import orbits, light_curves, gp, graph, sampling  # noqa: F401
from .light_curves import LimbDarkLightCurve, SecondaryEclipseLightCurve  # noqa: F401
from .orbits import KeplerianOrbit  # noqa: F401
from .graph import GraphedStep  # noqa: F401
from .sampling import HMC, NUTS  # noqa: F401

__version__ = "0.1.0"
