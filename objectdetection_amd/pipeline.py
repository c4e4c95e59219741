"""The whole hot path for a batch of images with every buffer allocated once.

RegionProposalPath.run() issues, on the current stream and without any host
synchronisation:

  1. anchors x GT IoU + flags            (patch_batch, module/data.py:79-110)
  2. RPN mini-batch sampling + encode    (patch_batch, module/data.py:111-149)
  3. score top-k + decode of the winners (process_fmap, module/model.py:45-48)
  4. NMS mask + scan -> proposals        (process_fmap, module/model.py:51)
  5. fused RoI crop + max-pool           (module/model.py:53-55 and :30)
  6. proposal <-> GT labels, targets     (making_frcnn_input, module/data.py:215-259)
  7. proposal sampling                   (making_frcnn_input, module/data.py:266-316)

Stages 1-2 and 6-7 are the training-only part (`train=True`).  They do not feed
stages 3-5, so they are issued on a second HIP stream: 1-2 run beside the top-k / NMS
chain, 6-7 beside the RoI kernel, and the streams are joined with events before run()
returns (still no host synchronisation).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from ._lib import lib, check
from .config import PathConfig, REFERENCE
from . import ops


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class RegionProposalPath:
    def __init__(self, cfg: PathConfig = REFERENCE, batch: int = 1, channels: int = 512,
                 device="cuda", train: bool = True, want_winner: bool = False,
                 fuse_pool: bool = True):
        self.cfg, self.B, self.C = cfg, batch, channels
        self.device = torch.device(device)
        self.train, self.fuse_pool = train, fuse_pool
        self.side = torch.cuda.Stream(self.device) if train else None
        self._ev = [torch.cuda.Event() for _ in range(3)] if train else None
        dev, B, N = self.device, batch, cfg.N
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.anchor = ops.anchors(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h, dev)
        self.pool = cfg.pool if fuse_pool else 1
        ph, pw = cfg.crop[0] // self.pool, cfg.crop[1] // self.pool
        K, R = cfg.pre_k, cfg.post_n
        self.top_scores, self.top_idx = e((B, K), f32), e((B, K), i32)
        self.cand = e((B, K, 4), f32)
        self.nms_ws = e((ops.nms_workspace_bytes(B, K, R),), u8)
        self.proposed, self.keep_pos = e((B, R, 4), f32), e((B, R), i32)
        self.keep_count = e((B,), i32)
        self.pooled = e((B, R, ph, pw, channels), f32)
        self.winner = e((B, R, ph, pw, channels), u8) if want_winner else None
        if train:
            S, S2 = cfg.rpn_batch, cfg.head_batch
            self.max_iou, self.best_gt, self.flags = e((B, N), f32), e((B, N), i32), e((B, N), u8)
            self.rpn = dict(label=e((B, S), f32), anchor=e((B, S, 4), f32), gt=e((B, S, 4), f32),
                            gt_list=e((B, S), i64), pos=e((B, S, 3), i64), reg=e((B, S, 4), f32),
                            counts=e((B, 4), i32))
            self.p_label, self.gt_mask = e((B, R), i64), e((B, R), i32)
            self.gt_onehot = e((B, R, cfg.num_classes), f32)
            self.gt_coord = e((B, R, 4), f32)
            self.p_best, self.p_iou = e((B, R), i32), e((B, R), f32)
            self.tindex, self.head_counts = e((B, S2), i64), e((B, 4), i32)

    def _validate(self, fmap, score, coord, gt, label):
        cfg, B = self.cfg, self.B

        def chk(t, name, shape, dtype, contiguous=True):
            if not (isinstance(t, torch.Tensor) and t.device == self.device):
                raise RuntimeError(f"{name}: must be a tensor on {self.device}")
            if t.dtype != dtype or tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: expected {dtype} {tuple(shape)}, got {t.dtype} "
                                 f"{tuple(t.shape)}")
            if contiguous and not t.is_contiguous():
                raise ValueError(f"{name}: must be contiguous")

        chk(fmap, "fmap", (B, cfg.H, cfg.W, self.C), torch.float32, contiguous=False)
        chk(score, "score", (B, cfg.H, cfg.W, cfg.A), torch.float32)
        chk(coord, "coord", (B, cfg.H, cfg.W, 4 * cfg.A), torch.float32)
        if self.train:
            if gt is None or label is None:
                raise ValueError("gt and label are required when train=True")
            chk(gt, "gt", (B, gt.shape[1] if gt.dim() == 3 else -1, 4), torch.float32)
            chk(label, "label", (B, gt.shape[1]), torch.int64)

    def run(self, fmap: torch.Tensor, score: torch.Tensor, coord: torch.Tensor,
            gt: Optional[torch.Tensor] = None, label: Optional[torch.Tensor] = None,
            seed: int = 0, image_base: int = 0, roi_events=None):
        """fmap [B,H,W,C] channels-last, score [B,H,W,A], coord [B,H,W,4A],
        gt [B,G,4], label [B,G] i64 (both only when train).  Returns self."""
        cfg, B, N = self.cfg, self.B, self.cfg.N
        self._validate(fmap, score, coord, gt, label)
        main = torch.cuda.current_stream(self.device)
        st = main.cuda_stream
        seed &= 0xFFFFFFFFFFFFFFFF
        if self.train:
            G = gt.shape[1]
            ss = self.side.cuda_stream
            self._ev[0].record(main)
            self.side.wait_event(self._ev[0])          # inputs were produced on the main stream
            check("frcnn_iou_assign", lib.frcnn_iou_assign(
                _p(self.anchor), N, _p(gt), B, G, cfg.rpn_pos_thr, cfg.rpn_neg_thr,
                _p(self.max_iou), _p(self.best_gt), _p(self.flags), None, ss))
            r = self.rpn
            check("frcnn_rpn_sample", lib.frcnn_rpn_sample(
                _p(self.anchor), cfg.H, cfg.W, cfg.A, _p(gt), B, G, _p(self.best_gt),
                _p(self.flags), None, None, None, seed, image_base, cfg.rpn_batch, cfg.rpn_max_pos,
                _p(r["label"]), _p(r["anchor"]), _p(r["gt"]), _p(r["gt_list"]), _p(r["pos"]),
                _p(r["reg"]), _p(r["counts"]), ss))
        check("frcnn_topk", lib.frcnn_topk(
            _p(score), B, N, cfg.pre_k, _p(self.anchor), _p(coord), _p(self.top_scores),
            _p(self.top_idx), _p(self.cand), st))
        check("frcnn_nms", lib.frcnn_nms(
            _p(self.cand), B, cfg.pre_k, cfg.nms_thr, cfg.post_n, _p(self.proposed),
            _p(self.keep_pos), _p(self.keep_count), _p(self.nms_ws), self.nms_ws.numel(), st))
        if self.train:
            self._ev[1].record(main)
            self.side.wait_event(self._ev[1])          # proposals ready
            ss = self.side.cuda_stream
            check("frcnn_label_proposals", lib.frcnn_label_proposals(
                _p(self.proposed), _p(gt), _p(label), B, cfg.post_n, gt.shape[1], cfg.fg_thr,
                cfg.iou_floor, cfg.bg_label, cfg.num_classes, _p(self.p_label), _p(self.gt_mask),
                _p(self.gt_onehot), _p(self.gt_coord), _p(self.p_best), _p(self.p_iou), ss))
            check("frcnn_sample_proposals", lib.frcnn_sample_proposals(
                _p(self.p_label), B, cfg.post_n, cfg.bg_label, None, None, None, seed, image_base,
                cfg.head_batch, cfg.head_max_pos, _p(self.tindex), _p(self.head_counts), ss))
            self._ev[2].record(self.side)
        sb, sh, sw, sc = fmap.stride()
        if roi_events is not None:
            roi_events[0].record(main)
        check("frcnn_roi_crop_pool_fwd", lib.frcnn_roi_crop_pool_fwd(
            _p(fmap), B, cfg.H, cfg.W, self.C, sb, sh, sw, sc, _p(self.proposed), cfg.post_n,
            cfg.crop[0], cfg.crop[1], self.pool, _p(self.pooled), _p(self.winner), st))
        if roi_events is not None:
            roi_events[1].record(main)
        if self.train:
            main.wait_event(self._ev[2])               # join: everything is ordered on `main` again
        return self
