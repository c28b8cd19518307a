"""ops (/root/reference/README.md:24): the detection operators behind the HIP C-ABI."""
from .nms import nms_batched  # noqa: F401
from .proposal import PyramidProposal  # noqa: F401
from .roi_align import fpn_level_map, roi_align_forward, roi_align_backward, roi_align_backward_gather  # noqa: F401
