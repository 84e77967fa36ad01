"""lattisense_amd — MI355X-native executor for LattiSense's RNS polynomial-arithmetic hot path."""
from . import params  # noqa: F401

__all__ = ["params"]
