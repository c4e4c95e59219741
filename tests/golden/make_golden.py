#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ from the CPU oracle.

The reference (TensorFlow 2.3, no tests, not installable here) cannot produce
vectors, so these are outputs of the oracle on seeded inputs, frozen so that
(a) later edits of the oracle are caught and (b) the GPU tests have fixed
expected values that do not depend on numpy's version of exp/log.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import frcnn_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, seed, H, W, scales, G, C, pre_k, post_n, B=2):
    A = len(scales) * 3
    a = O.make_anchor(H, W, scales, (0.5, 1, 2))
    d = O.synth_inputs(seed, B, H, W, A, C, G)
    pc = O.proposal_creator(d["score"], d["coord"], a, pre_k, post_n, 0.7)
    pooled, win = O.roi_crop_pool(d["fmap"], pc["proposed"][:, :24], (14, 14), 2)
    lab = O.label_proposals(pc["proposed"], d["gt"], d["label"])
    tindex = O.sample_proposals(lab["p_iou"], seed=seed)
    pb = [O.patch_batch(dict(image=None, bbox=d["gt"][b], label=d["label"][b]), a, seed=seed,
                        image_index=b) for b in range(B)]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        H=H, W=W, scales=np.asarray(scales, np.float64), G=G, C=C, pre_k=pre_k, post_n=post_n,
        seed=seed, score=d["score"], coord=d["coord"], fmap=d["fmap"], gt=d["gt"], label=d["label"],
        anchor_valid_count=int(O.anchor_valid(a).sum()),
        top_idx=pc["top_idx"], cand=pc["cand"], keep_pos=pc["keep_pos"], counts=pc["counts"],
        proposed=pc["proposed"], pooled24=pooled, winner24=win,
        p_label=lab["p_iou"], gt_mask=lab["gt_mask"], best=lab["best"], tindex=tindex,
        rpn_label=np.stack([p[1] for p in pb]), rpn_gt_list=np.stack([p[4] for p in pb]),
        rpn_pos=np.stack([p[5] for p in pb]), rpn_anchor=np.stack([p[2] for p in pb]))


def next_rows(name, seed):
    """Detection post-process, loss assembly and input side on one small seeded case."""
    rng = np.random.default_rng(seed)
    B, H, W, A, S, R, K, G = 2, 6, 7, 4, 24, 60, 21, 10
    proposed = np.sort(rng.random((B, R, 2, 2)).astype(np.float32), axis=2).transpose(0, 1, 3, 2)
    proposed = np.ascontiguousarray(proposed.reshape(B, R, 4)[..., [0, 2, 1, 3]])
    cls_prob = rng.random((B, R, K)).astype(np.float32) ** 6
    cls_prob[:, :, 3] += (rng.random((B, R)) < 0.4) * 2.0
    cls_prob = (cls_prob / cls_prob.sum(-1, keepdims=True)).astype(np.float32)
    reg = (rng.standard_normal((B, R, K * 4)) * 0.5).astype(np.float32)
    det = O.detect_postprocess(proposed, cls_prob, reg, bg_label=20, max_out=30,
                               iou_threshold=0.7, score_threshold=0.5)
    pred_obj = rng.random((B, H, W, A)).astype(np.float32)
    pred_reg = rng.standard_normal((B, H, W, A, 4)).astype(np.float32)
    pos = np.stack([np.stack(np.unravel_index(rng.permutation(H * W * A)[:S], (H, W, A)), -1)
                    for _ in range(B)]).astype(np.int64)
    label = (rng.random((B, S)) < 0.4).astype(np.float32)
    reg_gt = rng.standard_normal((B, S, 4)).astype(np.float32)
    rl, rd_obj, rd_reg = O.rpn_loss(pred_obj, pred_reg, pos, label, reg_gt)
    tindex = np.stack([rng.permutation(R)[:S] for _ in range(B)]).astype(np.int64)
    cls = rng.integers(0, K, (B, S))
    onehot = np.eye(K, dtype=np.float32)[cls]
    gt_mask = (cls != K - 1).astype(np.int32)
    gt_coord = (rng.standard_normal((B, S, 4)) * 60).astype(np.float32)
    hl, hd_cls, hd_reg = O.head_loss(cls_prob, reg.reshape(B, R, K, 4), tindex, onehot, gt_mask,
                                     gt_coord)
    n_obj = [4, 0]
    boxes = [np.sort(rng.random((n, 2, 2)).astype(np.float32), axis=1).transpose(0, 2, 1)
             .reshape(n, 4)[:, [0, 2, 1, 3]] for n in n_obj]
    labels = [rng.integers(0, 20, n).astype(np.int64) for n in n_obj]
    aug = [O.random_augmentation(None, labels[b], boxes[b], G, seed=seed, image_index=b)
           for b in range(B)]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=seed, G=G,
        proposed=proposed, cls_prob=cls_prob, reg=reg,
        det_boxes=det["boxes"], det_scores=det["scores"], det_classes=det["classes"],
        det_src=det["src"], det_count=det["count"],
        pred_obj=pred_obj, pred_reg=pred_reg, pos=pos, label=label, reg_gt=reg_gt,
        rpn_loss=np.float64(rl), rpn_d_obj=rd_obj.astype(np.float32),
        rpn_d_reg=rd_reg.astype(np.float32),
        tindex=tindex, onehot=onehot, gt_mask=gt_mask, gt_coord=gt_coord,
        head_loss=np.float64(hl), head_d_cls=hd_cls.astype(np.float32),
        head_d_reg=hd_reg.astype(np.float32),
        in_boxes=np.concatenate(boxes), in_labels=np.concatenate(labels),
        in_offsets=np.int32([0, n_obj[0], sum(n_obj)]),
        aug_boxes=np.stack([a[1] for a in aug]), aug_labels=np.stack([a[2] for a in aug]),
        flips=np.uint8([O.flip_decision(seed, b) for b in range(B)]))


if __name__ == "__main__":
    next_rows("next_rows_small", seed=23)
    case("ref_small", seed=21, H=12, W=12, scales=(128, 256, 512), G=10, C=8, pre_k=700, post_n=150)
    case("ship_small", seed=22, H=16, W=16, scales=(32, 64, 128, 256), G=24, C=8, pre_k=1500,
         post_n=200)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
