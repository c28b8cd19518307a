"""core/{anchor,bbox,mask,loss} -- host-side mirror of /root/reference/README.md:15-19 over the HIP C-ABI."""
from . import anchor, bbox, evaluation, loss, mask  # noqa: F401
