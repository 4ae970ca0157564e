from .keplerian import KeplerianOrbit, get_true_anomaly, get_aor_from_transit_duration  # noqa: F401

__all__ = ["KeplerianOrbit", "get_true_anomaly", "get_aor_from_transit_duration"]
