"""models/utils (/root/reference/README.md:33): parameter arenas and the conv / fc layer building block."""
from .layers import ConvLayer, ParamArena, Workspace  # noqa: F401
