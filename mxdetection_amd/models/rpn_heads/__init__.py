"""models/rpn_heads (/root/reference/README.md:28)."""
from .rpn_head import RPNHead  # noqa: F401
from .retina_head import RetinaHead  # noqa: F401
