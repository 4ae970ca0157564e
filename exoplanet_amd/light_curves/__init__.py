from .limb_dark import LimbDarkLightCurve, get_cl, exposure_stencil  # noqa: F401
from .secondary_eclipse import SecondaryEclipseLightCurve  # noqa: F401

__all__ = ["LimbDarkLightCurve", "SecondaryEclipseLightCurve"]
