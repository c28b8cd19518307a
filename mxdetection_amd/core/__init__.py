"""core/{anchor,bbox,loss} -- host-side mirror of /root/reference/README.md:15-19 over the HIP C-ABI."""
from . import anchor, bbox, loss  # noqa: F401
