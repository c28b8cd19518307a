"""datasets (/root/reference/README.md:21-22): COCO-format roidb + the per-rank batch loader."""
from .coco import append_flipped, filter_roidb, load_coco_roidb  # noqa: F401
from .loader import DetectionLoader, epoch_order  # noqa: F401
from .synthetic import synthetic_roidb  # noqa: F401
from .voc import VOC_CLASSES, load_voc_roidb  # noqa: F401
