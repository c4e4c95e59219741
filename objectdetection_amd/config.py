"""Sizes and thresholds of the hot path.

The reference hard-codes every one of these as a literal; the defaults below
are those literals (file:line cited per field).
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
class PathConfig:
    # anchor grid                      Faster_RCNN/module/anchor.py:11,18-19,25
    H: int = 31
    W: int = 31
    scales: Tuple[float, ...] = (128, 256, 512)
    ratios: Tuple[float, ...] = (0.5, 1, 2)
    base_w: float = 500.0
    base_h: float = 500.0
    # GT padding                       Faster_RCNN/module/data.py:13-18
    G: int = 42
    # RPN mini-batch                   Faster_RCNN/module/data.py:106,110,115,125
    rpn_pos_thr: float = 0.3
    rpn_neg_thr: float = 0.3
    rpn_batch: int = 256
    rpn_max_pos: int = 128
    # proposal creator                 Faster_RCNN/module/model.py:45,51
    pre_k: int = 6000
    post_n: int = 1500
    nms_thr: float = 0.7
    # RoI crop + pool                  Faster_RCNN/module/model.py:54,30
    crop: Tuple[int, int] = (14, 14)
    pool: int = 2
    # head targets                     Faster_RCNN/module/data.py:228,233,242,246,273-274
    fg_thr: float = 0.5
    iou_floor: float = 0.1
    bg_label: int = 20
    num_classes: int = 21
    head_batch: int = 128
    head_max_pos: int = 32

    @property
    def A(self) -> int:
        return len(self.scales) * len(self.ratios)

    @property
    def N(self) -> int:
        return self.H * self.W * self.A

    def with_(self, **kw) -> "PathConfig":
        return replace(self, **kw)


REFERENCE = PathConfig()

# shipdetection/train.py:129,136-137,143 (32x32 grid, 12 anchors, still /500), :38 (GT pad 24)
SHIP = PathConfig(H=32, W=32, scales=(32, 64, 128, 256), G=24)
