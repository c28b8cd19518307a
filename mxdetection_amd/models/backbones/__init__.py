"""models/backbones (/root/reference/README.md:27)."""
from .resnet import ResNet  # noqa: F401
