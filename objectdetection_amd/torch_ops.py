"""The C entry points as PyTorch operators (`torch.ops.frcnn.*`).

Registered with torch.library.custom_op: the dispatcher, fake tensors (shape propagation under
torch.compile / torch.export / FakeTensorMode) and autograd see the kernels as operators, while
the implementation stays the C ABI of libfrcnn_hip.so (through `ops.py`).  The fused RoI crop +
max-pool carries its adjoint (frcnn_roi_crop_pool_bwd) as the registered autograd formula; every
other operator produces indices, labels or targets and has no gradient, like the TensorFlow ops
it replaces (tf.math.top_k indices, combined_non_max_suppression, tf.where ...).

OP_ENTRY_POINTS maps each operator to the entry points of include/frcnn_hip.h it calls;
HELPER_ENTRY_POINTS lists the entry points that are not tensor computations (sizes, plans,
version, tuning) and the ones reached through another operator's implementation.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops

_lib_def = "frcnn"

OP_ENTRY_POINTS = {
    "anchors": ("frcnn_anchors",),
    "iou_assign": ("frcnn_iou_assign",),
    "rpn_sample": ("frcnn_rpn_sample",),
    "encode": ("frcnn_encode",),
    "decode_clip": ("frcnn_decode_clip",),
    "topk": ("frcnn_topk", "frcnn_topk_sort"),
    "nms": ("frcnn_nms",),
    "roi_crop_pool": ("frcnn_roi_crop_pool_fwd", "frcnn_roi_crop_pool_fwd_ws"),
    "roi_crop_pool_backward": ("frcnn_roi_crop_pool_bwd",),
    "nchw_to_nhwc": ("frcnn_nchw_to_nhwc",),
    "label_proposals": ("frcnn_label_proposals",),
    "sample_proposals": ("frcnn_sample_proposals",),
    "detect_postprocess": ("frcnn_detect_postprocess",),
    "rpn_loss": ("frcnn_rpn_loss",),
    "head_loss": ("frcnn_head_loss",),
    "pad_gt": ("frcnn_pad_gt",),
    "flip_images": ("frcnn_flip_images",),
    "fc1": ("frcnn_fc1_fwd",),
}
HELPER_ENTRY_POINTS = (
    "frcnn_abi_version", "frcnn_status_string", "frcnn_set_tuning", "frcnn_get_tuning",
    "frcnn_topk_sort_workspace_bytes", "frcnn_nms_workspace_bytes", "frcnn_roi_fwd_workspace_bytes",
    "frcnn_roi_fwd_plan", "frcnn_topk_sort_plan",
    # stream-ordered halves / allocation-free variants used by RegionProposalPath and serving
    # loops directly; as operators they would be the ones above
    "frcnn_roi_fwd_prepare", "frcnn_roi_crop_pool_fwd_prepared", "frcnn_nms_static",
    "frcnn_nms_fused", "frcnn_nms_fused_workspace_bytes",
)


def _op(name, mutates=()):
    return torch.library.custom_op(f"{_lib_def}::{name}", mutates_args=mutates, device_types="cuda")


def _e(ref: Tensor, shape, dtype=None) -> Tensor:
    return ref.new_empty(tuple(shape), dtype=dtype or ref.dtype)


# ---- a1 ---------------------------------------------------------------------------------------
@_op("anchors")
def anchors(like: Tensor, H: int, W: int, scales: Sequence[float], ratios: Sequence[float],
            base_w: float, base_h: float) -> Tensor:
    """[H,W,A,4] anchor grid on `like`'s device (module/anchor.py:10-44)."""
    return ops.anchors(H, W, list(scales), list(ratios), base_w, base_h, like.device)


@anchors.register_fake
def _(like, H, W, scales, ratios, base_w, base_h):
    return _e(like, (H, W, len(scales) * len(ratios), 4), torch.float32)


# ---- a2 / a3 ----------------------------------------------------------------------------------
@_op("iou_assign")
def iou_assign(anchors: Tensor, gt: Tensor, pos_thr: float, neg_thr: float) -> Tuple[Tensor, Tensor, Tensor]:
    """anchors [N,4], gt [B,G,4] -> max_iou [B,N], best_gt [B,N] i32, flags [B,N] u8."""
    mi, best, flags, _ = ops.iou_assign(anchors, gt, pos_thr, neg_thr)
    return mi, best, flags


@iou_assign.register_fake
def _(anchors, gt, pos_thr, neg_thr):
    B, N = gt.shape[0], anchors.numel() // 4
    return (_e(gt, (B, N)), _e(gt, (B, N), torch.int32), _e(gt, (B, N), torch.uint8))


@_op("rpn_sample")
def rpn_sample(anchor_box: Tensor, gt: Tensor, best_gt: Tensor, flags: Tensor, seed: int,
               image_base: int, batch: int, max_pos: int
               ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> label [B,S], anchor [B,S,4], gt [B,S,4], gt_list [B,S] i64, pos [B,S,3] i64,
    reg [B,S,4], counts [B,4] i32 (module/data.py:111-149; counter keys from `seed`)."""
    o = ops.rpn_sample(anchor_box, gt, best_gt, flags, None, seed, image_base, batch, max_pos)
    return o["label"], o["anchor"], o["gt"], o["gt_list"], o["pos"], o["reg"], o["counts"]


@rpn_sample.register_fake
def _(anchor_box, gt, best_gt, flags, seed, image_base, batch, max_pos):
    B, S = gt.shape[0], batch
    return (_e(gt, (B, S)), _e(gt, (B, S, 4)), _e(gt, (B, S, 4)), _e(gt, (B, S), torch.int64),
            _e(gt, (B, S, 3), torch.int64), _e(gt, (B, S, 4)), _e(gt, (B, 4), torch.int32))


# ---- a4 / a5 / a10 / a12 ----------------------------------------------------------------------
@_op("encode")
def encode(anchor: Tensor, gt: Tensor, clip_wh: bool) -> Tensor:
    return ops.encode(anchor, gt, clip_wh)


@encode.register_fake
def _(anchor, gt, clip_wh):
    return torch.empty_like(anchor)


@_op("decode_clip")
def decode_clip(anchors: Tensor, deltas: Tensor, clip_wh: bool) -> Tensor:
    """anchors [n,4] broadcast over deltas [B,n,4] -> boxes [B,n,4]."""
    return ops.decode_clip(anchors, deltas, clip_wh)


@decode_clip.register_fake
def _(anchors, deltas, clip_wh):
    n = anchors.numel() // 4
    return _e(deltas, (deltas.numel() // (4 * n), n, 4))


# ---- a6 ---------------------------------------------------------------------------------------
@_op("topk")
def topk(scores: Tensor, k: int, anchors: Tensor, deltas: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """scores [B,N] -> values [B,k], indices [B,k] i32, decoded boxes of the winners [B,k,4]."""
    return ops.topk(scores, k, anchors, deltas, "auto")


@topk.register_fake
def _(scores, k, anchors, deltas):
    B = scores.shape[0]
    return _e(scores, (B, k)), _e(scores, (B, k), torch.int32), _e(scores, (B, k, 4))


@_op("nms")
def nms(boxes: Tensor, iou_threshold: float, post_n: int) -> Tuple[Tensor, Tensor, Tensor]:
    """boxes [B,K,4] in score order -> proposed [B,post_n,4] (zero padded), keep_pos [B,post_n]
    i32 (-1 padded), count [B] i32."""
    return ops.nms(boxes, iou_threshold, post_n)


@nms.register_fake
def _(boxes, iou_threshold, post_n):
    B = boxes.shape[0]
    return (_e(boxes, (B, post_n, 4)), _e(boxes, (B, post_n), torch.int32),
            _e(boxes, (B,), torch.int32))


# ---- a7 + a8 + a13 ----------------------------------------------------------------------------
@_op("roi_crop_pool")
def roi_crop_pool(fmap: Tensor, rois: Tensor, crop_h: int, crop_w: int, pool: int) -> Tuple[Tensor, Tensor]:
    """fmap [B,H,W,C], rois [B,R,4] -> pooled [B,R,crop_h/pool,crop_w/pool,C] and the winner
    index (u8, same shape; all zeros for pool = 1).  Large channels-last calls run the
    band-resident kernel (the workspace is allocated here)."""
    B, H, W, C = fmap.shape
    R = rois.shape[1]
    ws = None
    if ops.roi_fwd_kernel_name(B, H, W, C, R, (crop_h, crop_w), pool, pool > 1).startswith("roi_fwd_band"):
        ws = torch.empty(ops.roi_fwd_workspace_bytes(B, R), dtype=torch.uint8, device=fmap.device)
    out, win = ops.roi_crop_pool_fwd(fmap, rois, (crop_h, crop_w), pool, want_winner=True,
                                     workspace=ws)
    return out, win


@roi_crop_pool.register_fake
def _(fmap, rois, crop_h, crop_w, pool):
    B, _, _, C = fmap.shape
    shape = (B, rois.shape[1], crop_h // pool, crop_w // pool, C)
    return _e(fmap, shape), _e(fmap, shape, torch.uint8)


@_op("roi_crop_pool_backward")
def roi_crop_pool_backward(grad_out: Tensor, winner: Tensor, rois: Tensor, B: int, H: int, W: int,
                           C: int, crop_h: int, crop_w: int, pool: int) -> Tensor:
    """Adjoint of roi_crop_pool with respect to the feature map: [B,H,W,C]."""
    return ops.roi_crop_pool_bwd(grad_out.contiguous(), winner if pool > 1 else None, rois,
                                 (B, H, W, C), (crop_h, crop_w), pool)


@roi_crop_pool_backward.register_fake
def _(grad_out, winner, rois, B, H, W, C, crop_h, crop_w, pool):
    return _e(grad_out, (B, H, W, C))


def _roi_setup(ctx, inputs, output):
    fmap, rois, crop_h, crop_w, pool = inputs
    ctx.save_for_backward(rois, output[1])
    ctx.meta = (tuple(fmap.shape), crop_h, crop_w, pool)
    ctx.mark_non_differentiable(output[1])


def _roi_backward(ctx, grad_out, grad_winner):
    rois, winner = ctx.saved_tensors
    (B, H, W, C), crop_h, crop_w, pool = ctx.meta
    g = torch.ops.frcnn.roi_crop_pool_backward(grad_out, winner, rois, B, H, W, C, crop_h, crop_w,
                                               pool)
    return g, None, None, None, None


roi_crop_pool.register_autograd(_roi_backward, setup_context=_roi_setup)


@_op("nchw_to_nhwc")
def nchw_to_nhwc(x: Tensor) -> Tensor:
    return ops.nchw_to_nhwc(x)


@nchw_to_nhwc.register_fake
def _(x):
    B, C, H, W = x.shape
    return _e(x, (B, H, W, C))


# ---- a9 - a11 ---------------------------------------------------------------------------------
@_op("label_proposals")
def label_proposals(proposed: Tensor, gt: Tensor, gt_label: Tensor, fg_thr: float, iou_floor: float,
                    bg_label: int, num_classes: int
                    ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> p_label [B,R] i64, gt_mask [B,R] i32, gt_onehot [B,R,K], gt_coord [B,R,4],
    best_gt [B,R] i32, max_iou [B,R]."""
    o = ops.label_proposals(proposed, gt, gt_label, fg_thr, iou_floor, bg_label, num_classes)
    return o["p_label"], o["gt_mask"], o["gt_onehot"], o["gt_coord"], o["best_gt"], o["max_iou"]


@label_proposals.register_fake
def _(proposed, gt, gt_label, fg_thr, iou_floor, bg_label, num_classes):
    B, R = proposed.shape[:2]
    return (_e(proposed, (B, R), torch.int64), _e(proposed, (B, R), torch.int32),
            _e(proposed, (B, R, num_classes)), _e(proposed, (B, R, 4)),
            _e(proposed, (B, R), torch.int32), _e(proposed, (B, R)))


@_op("sample_proposals")
def sample_proposals(p_label: Tensor, bg_label: int, seed: int, image_base: int, batch: int,
                     max_pos: int) -> Tuple[Tensor, Tensor]:
    """-> tindex [B,batch] i64, counts [B,4] i32 (counter keys from `seed`)."""
    return ops.sample_proposals(p_label, bg_label, None, seed, image_base, batch, max_pos)


@sample_proposals.register_fake
def _(p_label, bg_label, seed, image_base, batch, max_pos):
    B = p_label.shape[0]
    return _e(p_label, (B, batch), torch.int64), _e(p_label, (B, 4), torch.int32)


# ---- f1 - f3 ----------------------------------------------------------------------------------
@_op("detect_postprocess")
def detect_postprocess(proposed: Tensor, cls_prob: Tensor, reg: Tensor, bg_label: int, max_out: int,
                       iou_threshold: float, score_threshold: float
                       ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> boxes [B,max_out,4], scores [B,max_out], classes [B,max_out] i32, src [B,max_out] i32,
    count [B] i32 (model_train.py:745-759)."""
    o = ops.detect_postprocess(proposed, cls_prob, reg, bg_label, max_out, iou_threshold,
                               score_threshold)
    return o["boxes"], o["scores"], o["classes"], o["src"], o["count"]


@detect_postprocess.register_fake
def _(proposed, cls_prob, reg, bg_label, max_out, iou_threshold, score_threshold):
    B = proposed.shape[0]
    return (_e(proposed, (B, max_out, 4)), _e(proposed, (B, max_out)),
            _e(proposed, (B, max_out), torch.int32), _e(proposed, (B, max_out), torch.int32),
            _e(proposed, (B,), torch.int32))


@_op("rpn_loss")
def rpn_loss(pred_obj: Tensor, pred_reg: Tensor, pos: Tensor, label: Tensor, reg_gt: Tensor
             ) -> Tuple[Tensor, Tensor, Tensor]:
    """-> loss [3] = (cls, reg, 2 cls + reg), d loss[2] / d pred_obj, d loss[2] / d pred_reg
    (model_train.py:626-635)."""
    o = ops.rpn_loss(pred_obj, pred_reg, pos, label, reg_gt)
    return o["loss"], o["d_obj"], o["d_reg"]


@rpn_loss.register_fake
def _(pred_obj, pred_reg, pos, label, reg_gt):
    B, H, W, A = pred_obj.shape
    return _e(pred_obj, (3,)), torch.empty_like(pred_obj), _e(pred_obj, (B, H, W, A, 4))


@_op("head_loss")
def head_loss(cls_prob: Tensor, reg: Tensor, tindex: Tensor, label: Tensor, gt_mask: Tensor,
              gt_coord: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """-> loss [3], d loss[2] / d cls_prob [B,R,K], d loss[2] / d reg [B,R,K,4]
    (model_train.py:639-657)."""
    o = ops.head_loss(cls_prob, reg, tindex, label, gt_mask, gt_coord)
    return o["loss"], o["d_cls"], o["d_reg"]


@head_loss.register_fake
def _(cls_prob, reg, tindex, label, gt_mask, gt_coord):
    B, R, K = cls_prob.shape
    return _e(cls_prob, (3,)), torch.empty_like(cls_prob), _e(cls_prob, (B, R, K, 4))


@_op("pad_gt")
def pad_gt(boxes: Tensor, labels: Tensor, offsets: Tensor, G: int, use_flip: bool, seed: int,
           image_base: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Ragged boxes [total,4] / labels [total] / offsets [B+1] i32 -> gt [B,G,4], gt_label [B,G]
    i64, count [B] i32 (module/data.py:9-52)."""
    gt, gl, cnt = ops.pad_gt(boxes, labels, offsets, G, None, use_flip, seed, image_base)
    return gt, gl, cnt


@pad_gt.register_fake
def _(boxes, labels, offsets, G, use_flip, seed, image_base):
    B = offsets.numel() - 1
    return (_e(boxes, (B, G, 4)), _e(boxes, (B, G), torch.int64), _e(boxes, (B,), torch.int32))


@_op("flip_images")
def flip_images(images: Tensor, seed: int, image_base: int) -> Tensor:
    return ops.flip_images(images, None, seed, image_base)


@flip_images.register_fake
def _(images, seed, image_base):
    return torch.empty_like(images)


# ---- f4 ---------------------------------------------------------------------------------------
@_op("fc1")
def fc1(x: Tensor, weight: Tensor, bias: Tensor, relu: bool) -> Tensor:
    """x [M,K] (or the pooled tensor [B,R,ph,pw,C]), weight [K,N], bias [N] -> act(x W + b)
    on the float32 matrix cores (module/model.py:31-32).  Forward only."""
    return ops.fc1_fwd(x, weight, bias, relu)


@fc1.register_fake
def _(x, weight, bias, relu):
    lead = tuple(x.shape[:2]) if x.dim() == 5 else tuple(x.shape[:-1])
    return _e(x, lead + (weight.shape[1],))


OPS = tuple(OP_ENTRY_POINTS)
