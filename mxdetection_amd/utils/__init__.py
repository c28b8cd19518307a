"""mxdetection/utils (/root/reference/README.md:25): checkpoint I/O."""
from .params_io import fold_batchnorm, load_params, save_params  # noqa: F401
