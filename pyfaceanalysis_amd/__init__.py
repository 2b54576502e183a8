"""MI355X-native HiGSFA inference path (drop-in for PyFaceAnalysis' ``flow.execute``)."""
from . import nodes  # noqa: F401

__version__ = "0.1.0"
