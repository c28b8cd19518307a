"""mxdetection/utils (/root/reference/README.md:25): checkpoint I/O."""
from .params_io import fold_batchnorm, load_params, save_params  # noqa: F401
from .config import Config, default_config, load_config, update_config  # noqa: F401
from .lr_scheduler import WarmupMultiFactorScheduler, epoch_steps, scaled_lr  # noqa: F401
from .pretrained import load_pretrained_backbone, resnet_v1_names  # noqa: F401
