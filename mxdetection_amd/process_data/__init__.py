"""process_data (/root/reference/README.md:23): image / box / polygon transforms in front of the training path."""
from .transform import (BatchPreprocessor, flip_boxes, pad_shape, polygon_masks, resize_scale, resized_shape,  # noqa: F401
                        transform_boxes, transform_polygons)
