"""MI355X-native calibration-target detection + pose hot path."""
from . import abi  # noqa: F401
