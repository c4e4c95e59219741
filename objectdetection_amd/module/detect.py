"""Detection post-process -- counterpart of the per-epoch visual check in
Faster_RCNN/model_train.py:745-759 (class argmax -> generate_coord -> drop background ->
non_max_suppression_with_scores(30, 0.7) -> score >= 0.9), for a whole batch on the device."""
from __future__ import annotations

import torch

from .. import ops
from ..config import PathConfig, REFERENCE


def detect(proposed: torch.Tensor, pred_reg_valid: torch.Tensor, pred_obj_valid: torch.Tensor,
           cfg: PathConfig = REFERENCE, max_out: int = 30, iou_threshold: float = 0.7,
           score_threshold: float = 0.9):
    """proposed [B,R,4] (process_fmap's second output), pred_reg_valid [B,R,21*4] and
    pred_obj_valid [B,R,21] (the head's two outputs, module/model.py:36-38).
    Returns dict(boxes [B,max_out,4], scores, classes, src, count, decoded, cls_arg);
    rows beyond count[b] are zero / -1."""
    return ops.detect_postprocess(proposed, pred_obj_valid, pred_reg_valid, cfg.bg_label, max_out,
                                  iou_threshold, score_threshold)
