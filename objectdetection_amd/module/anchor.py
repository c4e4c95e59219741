"""make_anchor -- counterpart of Faster_RCNN/module/anchor.py:10-44."""
from __future__ import annotations

import torch

from .. import ops
from ..config import PathConfig, REFERENCE


def make_anchor(cfg: PathConfig = REFERENCE, device="cuda") -> torch.Tensor:
    """[H,W,A,4] float32 anchors, normalized [ymin,xmin,ymax,xmax], clipped.

    With no arguments this is the reference's 31x31x9 grid
    (module/anchor.py:10-44); `cfg` carries the other grids (e.g. the ship
    detector's 32x32x12, shipdetection/train.py:128-162).
    """
    return ops.anchors(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h, device)
