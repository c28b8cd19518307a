"""models/necks (/root/reference/README.md:31)."""
from .fpn import FPN  # noqa: F401
