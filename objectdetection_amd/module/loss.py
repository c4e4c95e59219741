"""Counterpart of Faster_RCNN/module/loss.py and of the loss lines of the training loop
(Faster_RCNN/model_train.py:598-600, 626-635, 639-657), on the HIP loss-assembly kernels.

The reference builds each loss from a chain of gather_nd calls and three Keras loss objects;
here one call per mini-batch returns the three scalars and, when asked, the gradient of
`2*cls + reg` with respect to the network outputs in their dense layout (what the two
GradientTapes of the reference would back-propagate into the RPN / head outputs).
"""
from __future__ import annotations

import torch

from .. import ops


class Loss_bbr:
    """Smooth-L1 (Huber) with the reference's name and call form (module/loss.py:4-14):
    element-wise values, no reduction.  Plain tensor arithmetic: the fused kernels below are
    what the training path uses."""

    def __init__(self, threshold: float = 1.0):
        self.threshold = threshold

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
        error = y_true - y_pred
        small = error.abs() < self.threshold
        return torch.where(small, error * error / 2,
                           self.threshold * error.abs() - self.threshold ** 2 / 2)


def rpn_losses(pred_obj, pred_reg, batch_label, batch_pos, batch_reg_gt, want_grad=True):
    """model_train.py:626-635.  pred_obj [B,H,W,A] (sigmoid), pred_reg [B,H,W,4A]; batch_label
    [B,S(,1)], batch_pos [B,S,3], batch_reg_gt [B,S,4] from patch_batch.  Returns
    (rpn_objectness_loss, rpn_bounding_box_loss, rpn_train_sub_loss, d_pred_obj, d_pred_reg)."""
    B = pred_obj.shape[0]
    out = ops.rpn_loss(pred_obj, pred_reg, batch_pos, batch_label.reshape(B, -1), batch_reg_gt,
                       want_grad)
    d_reg = out["d_reg"].reshape(pred_reg.shape) if want_grad else None
    return out["loss"][0], out["loss"][1], out["loss"][2], out["d_obj"], d_reg


def frcn_losses(pred_obj, pred_reg, label, gt_mask, gt_coord, tindex, want_grad=True):
    """model_train.py:639-657.  pred_obj [B,R,K] (softmax), pred_reg [B,R,4K]; label [B,S,K],
    gt_mask [B,S], gt_coord [B,S,4], tindex [B,S] from making_frcnn_input.  Returns
    (frcn_objectness_loss, frcn_bounding_box_loss, frcn_train_sub_loss, d_pred_obj,
    d_pred_reg)."""
    out = ops.head_loss(pred_obj, pred_reg, tindex, label, gt_mask, gt_coord, want_grad)
    d_reg = out["d_reg"].reshape(pred_reg.shape) if want_grad else None
    return out["loss"][0], out["loss"][1], out["loss"][2], out["d_cls"], d_reg
