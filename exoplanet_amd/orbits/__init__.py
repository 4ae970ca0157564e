from .keplerian import KeplerianOrbit, get_true_anomaly, get_aor_from_transit_duration  # noqa: F401
from .variants import SimpleTransitOrbit, TTVOrbit, compute_expected_transit_times  # noqa: F401

__all__ = ["KeplerianOrbit", "TTVOrbit", "SimpleTransitOrbit", "get_true_anomaly", "get_aor_from_transit_duration",
           "compute_expected_transit_times"]
