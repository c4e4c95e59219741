"""patch_batch, inverse_trans, generate_coord, making_frcnn_input --
counterparts of Faster_RCNN/module/data.py:70-329.

tf.random.shuffle is unseeded in the reference (data.py:120,126,134,276-281);
here a shuffle orders a list by per-position uint32 keys, which come either
from the caller (`keys=`) or from Philox-4x32-10 counters (`seed=`).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .._lib import S_OK
from ..config import PathConfig, REFERENCE
from . import model as _model


class ShortBatchError(RuntimeError):
    """Fewer negatives than the mini-batch needs (the reference reads out of
    bounds there, module/data.py:125-137)."""


def _check_counts(counts: torch.Tensor, what: str) -> None:
    bad = (counts[:, 3] != S_OK).nonzero()
    if bad.numel():
        c = counts[bad[0, 0]].tolist()
        raise ShortBatchError(f"{what}: image {int(bad[0, 0])} has {c[1]} negatives, "
                              f"needs batch - {c[2]}")


def patch_batch(data, anchor_box: torch.Tensor, cfg: PathConfig = REFERENCE, keys=None,
                seed: int = 0, image_index: int = 0, check: bool = True):
    """RPN targets for one example.  Counterpart of module/data.py:70-151.

    data: dict with 'image', 'bbox' [G,4] f32, 'label' [G].  Returns the
    reference's 9-tuple (data.py:151): image, batch_label [S,1], batch_anchor
    [S,4], batch_gt [S,4], gt_list [S] i64, batch_pos [S,3] i64, batch_reg_gt
    [S,4], label, gt_box.  `check=True` synchronises once to raise
    ShortBatchError where the reference would fail.
    """
    image, gt_box, label = data["image"], data["bbox"], data["label"]
    out = patch_batch_many(gt_box.unsqueeze(0), anchor_box, cfg,
                           None if keys is None else tuple(k.unsqueeze(0) for k in keys),
                           seed, image_index, check)
    return (image, out["label"][0].unsqueeze(-1), out["anchor"][0], out["gt"][0],
            out["gt_list"][0], out["pos"][0], out["reg"][0], label, gt_box)


def patch_batch_many(gt_box: torch.Tensor, anchor_box: torch.Tensor, cfg: PathConfig = REFERENCE,
                     keys=None, seed: int = 0, image_base: int = 0, check: bool = False):
    """patch_batch for B images at once (gt_box [B,G,4]); one launch pair.
    Returns dict(label [B,S], anchor, gt, gt_list, pos, reg, counts, max_iou, best_gt, flags)."""
    a = anchor_box.contiguous()
    max_iou, best, flags, _ = ops.iou_assign(a.reshape(-1, 4), gt_box, cfg.rpn_pos_thr,
                                             cfg.rpn_neg_thr)
    out = ops.rpn_sample(a, gt_box, best, flags, keys, seed, image_base, cfg.rpn_batch,
                         cfg.rpn_max_pos)
    out.update(max_iou=max_iou, best_gt=best, flags=flags)
    if check:
        _check_counts(out["counts"], "patch_batch")
    return out


def inverse_trans(anchor_box: torch.Tensor, pred_reg: torch.Tensor) -> torch.Tensor:
    """RPN deltas -> boxes.  Counterpart of module/data.py:154-176.
    anchor_box [H,W,A,4], pred_reg [B,H,W,4A] -> [B,H,W,A,4]."""
    H, W, A = anchor_box.shape[:3]
    boxes = ops.decode_clip(anchor_box.reshape(-1, 4), pred_reg.reshape(-1, H * W * A, 4), False)
    return boxes.reshape(-1, H, W, A, 4)


def generate_coord(proposed: torch.Tensor, pred_reg: torch.Tensor) -> torch.Tensor:
    """Head deltas -> boxes.  Counterpart of module/data.py:179-198.  [B,R,4] each."""
    shape = proposed.shape
    n = proposed.numel() // 4
    boxes = ops.decode_clip(proposed.reshape(n, 4), pred_reg.reshape(1, n, 4), True)
    return boxes.reshape(shape)


def making_frcnn_input(gt_box: torch.Tensor, label: torch.Tensor, fmap: torch.Tensor,
                       pred_reg: torch.Tensor, pred_obj: torch.Tensor,
                       anchor_box: Optional[torch.Tensor] = None, cfg: PathConfig = REFERENCE,
                       keys=None, seed: int = 0, image_base: int = 0, fuse_pool: bool = False,
                       check: bool = True, cache: bool = True, inputs_ready=None):
    """Head inputs and targets.  Counterpart of module/data.py:200-329.

    The reference reads `anchor_box` from an undefined global (data.py:205);
    here it is the sixth argument.  Returns the reference's 6-tuple
    (data.py:329): crop_fmap [B,R,14,14,C], gt_label2 [B,S,21], gt_mask2 [B,S],
    gt_coord2 [B,S,4], proposed [B,R,4], tindex [B,S].  With fuse_pool=True
    crop_fmap is the 2x2-max-pooled [B,R,7,7,C] tensor (the head's first layer,
    module/model.py:30, fused into the crop).

    Sampling is per image for every batch size.  The reference samples per image only for
    B == 2 (data.py:266-298); its else branch (data.py:300-316), taken for any other B, draws
    one joint list over the whole [B,R] tensor, caps the positives at 32 in total and returns
    [1,128].  For B == 1 the two agree; the joint branch for B >= 3 is not reproduced.

    cache=True (default) with counter keys (keys=None): the call runs on a buffer set kept per
    (cfg, B, C, device), see process_fmap -- crop_fmap is overwritten by the next call of the same
    dimensions, proposed / tindex by the one after it; the three gathered targets are new
    tensors.  Injected keys (parity runs) and cache=False take the allocate-per-call route.
    """
    if anchor_box is None:
        raise ValueError("anchor_box is required (the reference reads an undefined global here)")
    B = fmap.shape[0]
    if cache and keys is None:
        path = _model._path_for(fmap, pred_obj, anchor_box, cfg, True, fuse_pool)
        path.run(fmap, pred_obj.contiguous(), pred_reg.contiguous(),
                 gt_box.reshape(B, -1, 4).contiguous(),
                 label.reshape(B, -1).to(torch.int64).contiguous(), seed=seed,
                 image_base=image_base, inputs_ready=inputs_ready)
        if check:
            _check_counts(path.head_counts, "making_frcnn_input")
        tindex = path.tindex
        gt_label2 = torch.gather(path.gt_onehot, 1,
                                 tindex.unsqueeze(-1).expand(-1, -1, cfg.num_classes))
        gt_coord2 = torch.gather(path.gt_coord, 1, tindex.unsqueeze(-1).expand(-1, -1, 4))
        gt_mask2 = torch.gather(path.gt_mask, 1, tindex)
        return path.pooled, gt_label2, gt_mask2, gt_coord2, path.proposed, tindex
    crop_fmap, proposed = _model.process_fmap(fmap, pred_obj, pred_reg, anchor_box, cfg,
                                              fuse_pool=fuse_pool, cache=False)
    lab = ops.label_proposals(proposed, gt_box.reshape(fmap.shape[0], -1, 4),
                              label.reshape(fmap.shape[0], -1).to(torch.int64), cfg.fg_thr,
                              cfg.iou_floor, cfg.bg_label, cfg.num_classes)
    tindex, counts = ops.sample_proposals(lab["p_label"], cfg.bg_label, keys, seed, image_base,
                                          cfg.head_batch, cfg.head_max_pos)
    if check:
        _check_counts(counts, "making_frcnn_input")
    gt_label2 = torch.gather(lab["gt_onehot"], 1,
                             tindex.unsqueeze(-1).expand(-1, -1, cfg.num_classes))
    gt_coord2 = torch.gather(lab["gt_coord"], 1, tindex.unsqueeze(-1).expand(-1, -1, 4))
    gt_mask2 = torch.gather(lab["gt_mask"], 1, tindex)
    return crop_fmap, gt_label2, gt_mask2, gt_coord2, proposed, tindex


# ---- input side (module/data.py:9-52), batched -------------------------------------------
def _ragged(bboxes, labels, device):
    """Lists of per-image [n_i,4] boxes / [n_i] labels -> (boxes, labels, offsets) on device."""
    counts = [int(torch.as_tensor(b).reshape(-1, 4).shape[0]) for b in bboxes]
    offsets = torch.zeros(len(counts) + 1, dtype=torch.int32)
    offsets[1:] = torch.tensor(counts, dtype=torch.int32).cumsum(0)
    cat = torch.cat([torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4) for b in bboxes])
    lab = None
    if labels is not None:
        lab = torch.cat([torch.as_tensor(l, dtype=torch.int64).reshape(-1) for l in labels])
        lab = lab.to(device)
    return cat.to(device), lab, offsets.to(device)


def data_preprocess(images: torch.Tensor, bboxes, labels=None, cfg: PathConfig = REFERENCE):
    """Batched counterpart of data_preprocess (module/data.py:9-21) without the resize:
    images [B,H,W,C] on the device, bboxes / labels lists of per-image tensors.  Returns
    {"image", "bbox" [B,G,4], "label" [B,G]} with zero padding to cfg.G."""
    boxes, lab, off = _ragged(bboxes, labels, images.device)
    gt, gl, _ = ops.pad_gt(boxes, lab, off, cfg.G)
    return {"image": images, "bbox": gt, "label": gl}


def augmentation_data(images: torch.Tensor, labels, bbox: torch.Tensor):
    """module/data.py:23-33 for a batch: every image and its padded-free boxes [n,4] flipped."""
    flip = torch.ones(images.shape[0], dtype=torch.uint8, device=images.device)
    out = torch.stack([bbox[:, 0], 1 - bbox[:, 3], bbox[:, 2], 1 - bbox[:, 1]], 1)
    return [ops.flip_images(images, flip), labels, out]


def random_augmentation(images: torch.Tensor, bboxes, labels=None, cfg: PathConfig = REFERENCE,
                        flip: Optional[torch.Tensor] = None, seed: int = 0, image_base: int = 0):
    """Batched counterpart of random_augmentation (module/data.py:36-52) without the resize:
    per image, flip image and boxes with probability 1/2 (`flip` [B] u8 makes the decision
    explicit, else Philox counters from `seed`), then pad boxes / labels to cfg.G."""
    boxes, lab, off = _ragged(bboxes, labels, images.device)
    gt, gl, _ = ops.pad_gt(boxes, lab, off, cfg.G, flip, True, seed, image_base)
    return {"image": ops.flip_images(images, flip, seed, image_base), "bbox": gt, "label": gl}
