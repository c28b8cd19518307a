"""models (/root/reference/README.md:26-33): backbones, necks, rpn_heads, roi_extractors, bbox_heads, mask_heads."""
from .faster_rcnn import FasterRCNN  # noqa: F401
from .retinanet import RetinaNet  # noqa: F401
