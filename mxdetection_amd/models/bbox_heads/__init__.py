"""models/bbox_heads (/root/reference/README.md:29)."""
from .convfc_bbox_head import BBoxHead  # noqa: F401
