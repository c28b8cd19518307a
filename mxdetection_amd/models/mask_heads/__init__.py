"""models/mask_heads (/root/reference/README.md:30)."""
from .fcn_mask_head import FCNMaskHead  # noqa: F401
