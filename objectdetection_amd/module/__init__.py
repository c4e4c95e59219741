"""Drop-in names of Faster_RCNN/module: same function names, argument order
and return tuples, on torch-ROCm tensors, computed by the HIP kernels."""
from .anchor import make_anchor  # noqa: F401
from .data import (patch_batch, inverse_trans, generate_coord, making_frcnn_input,  # noqa: F401
                   data_preprocess, augmentation_data, random_augmentation)
from .model import process_fmap, roi_pool, roi_pool_autograd  # noqa: F401
from .detect import detect  # noqa: F401
from .loss import Loss_bbr, rpn_losses, frcn_losses  # noqa: F401
