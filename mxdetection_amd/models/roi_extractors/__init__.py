"""models/roi_extractors (/root/reference/README.md:32)."""
from .fpn_roi_extractor import FPNRoIExtractor  # noqa: F401
