"""MI355X-native Faster R-CNN region-proposal + RoI hot path.

Layout
  csrc/      HIP kernels (gfx950) and the C ABI  -> lib/libfrcnn_hip.so
  _lib.py    ctypes binding of the C ABI (raises if the library is missing)
  ops.py     torch.Tensor front-end, one function per entry point
  module/    the reference's function names and argument order
             (Faster_RCNN/module/{anchor,data,model}.py)
  config.py  the sizes the reference hard-codes, as parameters
  shard.py   per-image sharding across ranks (one process per GPU)
"""
from .config import PathConfig, REFERENCE, SHIP  # noqa: F401

__version__ = "0.1.0"
