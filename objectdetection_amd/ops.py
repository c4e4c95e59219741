"""torch.Tensor front-end of the C ABI: one function per entry point.

PyTorch supplies device memory and the current HIP stream only; every
computation happens in libfrcnn_hip.so.  Tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import lib, check

__all__ = [
    "anchors", "iou_assign", "rpn_sample", "encode", "decode_clip", "topk", "nms",
    "nms_workspace_bytes", "roi_crop_pool_fwd", "roi_crop_pool_bwd", "label_proposals",
    "sample_proposals", "detect_postprocess", "fc1_fwd",
]


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor, name: str, dtype: torch.dtype) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must be on a ROCm device (the HIP kernels are the only path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _cont(t: torch.Tensor, name: str, dtype: torch.dtype) -> torch.Tensor:
    return _dev(t, name, dtype).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def anchors(H: int, W: int, scales: Sequence[float], ratios: Sequence[float], base_w: float = 500.0,
            base_h: Optional[float] = None, device="cuda") -> torch.Tensor:
    """[H,W,A,4] float32 anchor grid (frcnn_anchors)."""
    base_h = base_w if base_h is None else base_h
    A = len(scales) * len(ratios)
    out = torch.empty((H, W, A, 4), dtype=torch.float32, device=device)
    sc = (C.c_double * len(scales))(*[float(s) for s in scales])
    rt = (C.c_double * len(ratios))(*[float(r) for r in ratios])
    with torch.cuda.device(out.device):
        check("frcnn_anchors", lib.frcnn_anchors(_ptr(out), H, W, sc, len(scales), rt, len(ratios),
                                                 float(base_w), float(base_h), _stream(out)))
    return out


def iou_assign(anchor_flat: torch.Tensor, gt: torch.Tensor, pos_thr: float = 0.3,
               neg_thr: float = 0.3, want_iou: bool = False):
    """anchors [N,4], gt [B,G,4] -> max_iou [B,N], best_gt [B,N] i32, flags [B,N] u8, iou|None."""
    a = _cont(anchor_flat, "anchors", torch.float32).reshape(-1, 4)
    g = _cont(gt, "gt", torch.float32)
    if g.dim() != 3 or g.shape[-1] != 4:
        raise ValueError("gt must be [B,G,4]")
    B, G = g.shape[0], g.shape[1]
    N = a.shape[0]
    max_iou = torch.empty((B, N), dtype=torch.float32, device=a.device)
    best = torch.empty((B, N), dtype=torch.int32, device=a.device)
    flags = torch.empty((B, N), dtype=torch.uint8, device=a.device)
    full = torch.empty((B, N, G), dtype=torch.float32, device=a.device) if want_iou else None
    with torch.cuda.device(a.device):
        check("frcnn_iou_assign", lib.frcnn_iou_assign(_ptr(a), N, _ptr(g), B, G, pos_thr, neg_thr,
                                                       _ptr(max_iou), _ptr(best), _ptr(flags),
                                                       _ptr(full), _stream(a)))
    return max_iou, best, flags, full


def _keys(keys, B, n_list, batch, device):
    if keys is None:
        return None, None, None
    def as_i32(t, name, cols):
        if isinstance(t, torch.Tensor) and t.dtype == torch.uint32:
            t = t.view(torch.int32)
        return _cont(t, name, torch.int32).reshape(B, cols)

    kp, kn, ka = keys
    return (as_i32(kp, "keys_pos", n_list), as_i32(kn, "keys_neg", n_list),
            as_i32(ka, "keys_all", batch))


def rpn_sample(anchor_box: torch.Tensor, gt: torch.Tensor, best_gt: torch.Tensor,
               flags: torch.Tensor, keys=None, seed: int = 0, image_base: int = 0,
               batch: int = 256, max_pos: int = 128):
    """RPN mini-batch (frcnn_rpn_sample).  anchor_box [H,W,A,4]; gt [B,G,4].

    keys = (keys_pos [B,N], keys_neg [B,N], keys_all [B,batch]) as int32 tensors
    holding uint32 bit patterns, or None for counter keys from `seed`.
    Returns dict(label, anchor, gt, gt_list, pos, reg, counts)."""
    a = _cont(anchor_box, "anchor_box", torch.float32)
    H, W, A = a.shape[0], a.shape[1], a.shape[2]
    N = H * W * A
    g = _cont(gt, "gt", torch.float32)
    B, G = g.shape[0], g.shape[1]
    bg = _cont(best_gt, "best_gt", torch.int32).reshape(B, N)
    fl = _cont(flags, "flags", torch.uint8).reshape(B, N)
    kp, kn, ka = _keys(keys, B, N, batch, a.device)
    dev = a.device
    out = dict(
        label=torch.empty((B, batch), dtype=torch.float32, device=dev),
        anchor=torch.empty((B, batch, 4), dtype=torch.float32, device=dev),
        gt=torch.empty((B, batch, 4), dtype=torch.float32, device=dev),
        gt_list=torch.empty((B, batch), dtype=torch.int64, device=dev),
        pos=torch.empty((B, batch, 3), dtype=torch.int64, device=dev),
        reg=torch.empty((B, batch, 4), dtype=torch.float32, device=dev),
        counts=torch.empty((B, 4), dtype=torch.int32, device=dev),
    )
    with torch.cuda.device(dev):
        check("frcnn_rpn_sample", lib.frcnn_rpn_sample(
            _ptr(a), H, W, A, _ptr(g), B, G, _ptr(bg), _ptr(fl), _ptr(kp), _ptr(kn), _ptr(ka),
            seed & 0xFFFFFFFFFFFFFFFF, image_base, batch, max_pos, _ptr(out["label"]),
            _ptr(out["anchor"]), _ptr(out["gt"]), _ptr(out["gt_list"]), _ptr(out["pos"]),
            _ptr(out["reg"]), _ptr(out["counts"]), _stream(a)))
    return out


def encode(anchor: torch.Tensor, gt: torch.Tensor, clip_wh: bool) -> torch.Tensor:
    a = _cont(anchor, "anchor", torch.float32)
    g = _cont(gt, "gt", torch.float32)
    if a.shape != g.shape or a.shape[-1] != 4:
        raise ValueError("anchor and gt must have the same [...,4] shape")
    out = torch.empty_like(a)
    n = a.numel() // 4
    if n:
        with torch.cuda.device(a.device):
            check("frcnn_encode", lib.frcnn_encode(_ptr(a), _ptr(g), n, int(clip_wh), _ptr(out),
                                                   _stream(a)))
    return out


def decode_clip(anchor_flat: torch.Tensor, deltas: torch.Tensor, clip_wh: bool) -> torch.Tensor:
    """anchors [n,4] broadcast over deltas [B,n,4] -> boxes [B,n,4]."""
    a = _cont(anchor_flat, "anchors", torch.float32).reshape(-1, 4)
    d = _cont(deltas, "deltas", torch.float32)
    n = a.shape[0]
    d = d.reshape(-1, n, 4)
    out = torch.empty_like(d)
    with torch.cuda.device(a.device):
        check("frcnn_decode_clip", lib.frcnn_decode_clip(_ptr(a), _ptr(d), d.shape[0], n,
                                                         int(clip_wh), _ptr(out), _stream(a)))
    return out


def topk_sort_workspace_bytes(B: int, N: int) -> int:
    return int(lib.frcnn_topk_sort_workspace_bytes(B, N))


def topk_sort_route(B: int, N: int, workspace_bytes: Optional[int] = None) -> str:
    """Which form frcnn_topk_sort takes for a call of this size (frcnn_topk_sort_plan): "sorted runs"
    (two launches, a few images) or "radix passes"."""
    if workspace_bytes is None:
        workspace_bytes = topk_sort_workspace_bytes(B, N)
    return "sorted runs" if lib.frcnn_topk_sort_plan(B, N, workspace_bytes) else "radix passes"


def topk(scores: torch.Tensor, k: int, anchor_flat: Optional[torch.Tensor] = None,
         deltas: Optional[torch.Tensor] = None, method: str = "auto",
         workspace: Optional[torch.Tensor] = None):
    """scores [B,N] -> (values [B,k], idx [B,k] i32, cand_boxes [B,k,4] | None).
    method: "lds" (frcnn_topk, k <= 16384), "sort" (frcnn_topk_sort, any k) or "auto" (= "sort":
    it is the faster one at every size measured -- one image 36 us against 250, 64 images 0.12 ms
    against 0.27 -- also when its workspace has to come from torch's caching allocator)."""
    s = _cont(scores, "scores", torch.float32)
    if s.dim() != 2:
        raise ValueError("scores must be [B,N]")
    B, N = s.shape
    vals = torch.empty((B, k), dtype=torch.float32, device=s.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=s.device)
    a = d = cand = None
    if anchor_flat is not None:
        a = _cont(anchor_flat, "anchors", torch.float32).reshape(-1, 4)
        d = _cont(deltas, "deltas", torch.float32).reshape(B, -1, 4)
        if a.shape[0] != N or d.shape[1] != N:
            raise ValueError("anchors / deltas do not match scores")
        cand = torch.empty((B, k, 4), dtype=torch.float32, device=s.device)
    if method == "auto":
        method = "sort"
    with torch.cuda.device(s.device):
        if method == "lds":
            check("frcnn_topk", lib.frcnn_topk(_ptr(s), B, N, k, _ptr(a), _ptr(d), _ptr(vals),
                                               _ptr(idx), _ptr(cand), _stream(s)))
        elif method == "sort":
            need = topk_sort_workspace_bytes(B, N)
            if workspace is None:
                workspace = torch.empty(need, dtype=torch.uint8, device=s.device)
            check("frcnn_topk_sort", lib.frcnn_topk_sort(
                _ptr(s), B, N, k, _ptr(a), _ptr(d), _ptr(vals), _ptr(idx), _ptr(cand),
                _ptr(workspace), workspace.numel() * workspace.element_size(), _stream(s)))
        else:
            raise ValueError("method must be 'auto', 'lds' or 'sort'")
    return vals, idx, cand


def nms_workspace_bytes(B: int, K: int, post_n: int) -> int:
    return int(lib.frcnn_nms_workspace_bytes(B, K, post_n))


def nms(boxes: torch.Tensor, iou_threshold: float, post_n: int,
        workspace: Optional[torch.Tensor] = None):
    """boxes [B,K,4] score-sorted -> (proposed [B,post_n,4], keep_pos [B,post_n] i32, count [B])."""
    b = _cont(boxes, "boxes", torch.float32)
    if b.dim() != 3 or b.shape[-1] != 4:
        raise ValueError("boxes must be [B,K,4]")
    B, K = b.shape[0], b.shape[1]
    need = nms_workspace_bytes(B, K, post_n)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=b.device)
    proposed = torch.empty((B, post_n, 4), dtype=torch.float32, device=b.device)
    keep = torch.empty((B, post_n), dtype=torch.int32, device=b.device)
    count = torch.empty((B,), dtype=torch.int32, device=b.device)
    with torch.cuda.device(b.device):
        check("frcnn_nms", lib.frcnn_nms(_ptr(b), B, K, iou_threshold, post_n, _ptr(proposed),
                                         _ptr(keep), _ptr(count), _ptr(workspace),
                                         workspace.numel() * workspace.element_size(), _stream(b)))
    return proposed, keep, count


NMS_FUSED_MAX_B = 8


def nms_fused_workspace_bytes(B: int, K: int, post_n: int) -> int:
    return int(lib.frcnn_nms_fused_workspace_bytes(B, K, post_n))


def nms_fused(boxes: torch.Tensor, iou_threshold: float, post_n: int,
              workspace: Optional[torch.Tensor] = None):
    """frcnn_nms_fused: the NMS of up to 8 images in one launch (same results as nms())."""
    b = _cont(boxes, "boxes", torch.float32)
    if b.dim() != 3 or b.shape[-1] != 4:
        raise ValueError("boxes must be [B,K,4]")
    B, K = b.shape[0], b.shape[1]
    if workspace is None:
        workspace = torch.empty(nms_fused_workspace_bytes(B, K, post_n), dtype=torch.uint8,
                                device=b.device)
    proposed = torch.empty((B, post_n, 4), dtype=torch.float32, device=b.device)
    keep = torch.empty((B, post_n), dtype=torch.int32, device=b.device)
    count = torch.empty((B,), dtype=torch.int32, device=b.device)
    with torch.cuda.device(b.device):
        check("frcnn_nms_fused", lib.frcnn_nms_fused(
            _ptr(b), B, K, iou_threshold, post_n, _ptr(proposed), _ptr(keep), _ptr(count),
            _ptr(workspace), workspace.numel() * workspace.element_size(), _stream(b)))
    return proposed, keep, count


def roi_fwd_workspace_bytes(B: int, R: int, planar_shape: Optional[Tuple[int, int, int]] = None) -> int:
    """Workspace of frcnn_roi_crop_pool_fwd_ws; planar_shape = (H, W, C) adds the room for the
    channels-last copy of a [B,C,H,W] feature map."""
    n = int(lib.frcnn_roi_fwd_workspace_bytes(B, R))
    if planar_shape is not None:
        H, W, Cc = planar_shape
        n += B * H * W * Cc * 4
    return n


def roi_fwd_kernel_name(B: int, H: int, W: int, Cc: int, R: int, crop: Sequence[int], pool: int,
                        want_winner: bool = False) -> str:
    """Name of the kernel frcnn_roi_crop_pool_fwd_ws launches for a channels-last, aligned map
    and a full workspace (frcnn_roi_fwd_plan)."""
    band = lib.frcnn_roi_fwd_plan(B, H, W, Cc, R, int(crop[0]), int(crop[1]), pool, int(want_winner))
    w = "true" if want_winner else "false"
    return f"roi_fwd_band_kernel<{pool},{w}>" if band else f"roi_fwd_stage_kernel<{pool},{w}>"


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] contiguous -> new [B,H,W,C] contiguous tensor (frcnn_nchw_to_nhwc)."""
    s = _cont(x, "x", torch.float32)
    B, Cc, H, W = s.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=s.device)
    with torch.cuda.device(s.device):
        check("frcnn_nchw_to_nhwc", lib.frcnn_nchw_to_nhwc(_ptr(s), B, Cc, H, W, _ptr(out), _stream(s)))
    return out


def roi_crop_pool_fwd(fmap: torch.Tensor, rois: torch.Tensor, crop: Tuple[int, int] = (14, 14),
                      pool: int = 2, want_winner: bool = False, out: Optional[torch.Tensor] = None,
                      workspace: Optional[torch.Tensor] = None,
                      winner: Optional[torch.Tensor] = None, prepared: bool = False):
    """fmap [B,H,W,C] (any strides; channels-last is the fast layout), rois [B,R,4]
    -> pooled [B,R,crop_h/pool,crop_w/pool,C] (+ winner u8).  With `workspace` (a uint8 tensor
    of roi_fwd_workspace_bytes(B,R) bytes) the call goes through frcnn_roi_crop_pool_fwd_ws
    (band-resident kernel for large calls, the same kernels otherwise); prepared=True after
    roi_fwd_prepare() returned True for the same rois and workspace."""
    f = _dev(fmap, "fmap", torch.float32)
    if f.dim() != 4:
        raise ValueError("fmap must be [B,H,W,C]")
    r = _cont(rois, "rois", torch.float32)
    B, H, W, Cc = f.shape
    if r.dim() != 3 or r.shape[0] != B or r.shape[2] != 4:
        raise ValueError("rois must be [B,R,4]")
    R = r.shape[1]
    ph, pw = crop[0] // pool, crop[1] // pool
    if out is None:
        out = torch.empty((B, R, ph, pw, Cc), dtype=torch.float32, device=f.device)
    else:
        _dev(out, "out", torch.float32)
        if tuple(out.shape) != (B, R, ph, pw, Cc) or not out.is_contiguous():
            raise ValueError("out has the wrong shape or is not contiguous")
    win = winner
    if want_winner and win is None:
        win = torch.empty((B, R, ph, pw, Cc), dtype=torch.uint8, device=f.device)
    sb, sh, sw, sc = f.stride()
    with torch.cuda.device(f.device):
        if workspace is not None:
            _dev(workspace, "workspace", torch.uint8)
            fn = lib.frcnn_roi_crop_pool_fwd_prepared if prepared else lib.frcnn_roi_crop_pool_fwd_ws
            check("frcnn_roi_crop_pool_fwd_ws", fn(
                _ptr(f), B, H, W, Cc, sb, sh, sw, sc, _ptr(r), R, crop[0], crop[1], pool,
                _ptr(out), _ptr(win), _ptr(workspace), workspace.numel(), _stream(f)))
        else:
            check("frcnn_roi_crop_pool_fwd", lib.frcnn_roi_crop_pool_fwd(
                _ptr(f), B, H, W, Cc, sb, sh, sw, sc, _ptr(r), R, crop[0], crop[1], pool, _ptr(out),
                _ptr(win), _stream(f)))
    return out, win


def roi_fwd_prepare(rois: torch.Tensor, fmap_shape: Tuple[int, int, int, int],
                    crop: Tuple[int, int], pool: int, want_winner: bool,
                    workspace: torch.Tensor) -> bool:
    """frcnn_roi_fwd_prepare on the current stream: True if the workspace now holds the sample
    geometry of `rois` (pass prepared=True to roi_crop_pool_fwd), False if there was nothing to
    prepare for a call of these dimensions."""
    r = _cont(rois, "rois", torch.float32)
    _dev(workspace, "workspace", torch.uint8)
    B, H, W, Cc = fmap_shape
    with torch.cuda.device(r.device):
        rc = lib.frcnn_roi_fwd_prepare(_ptr(r), B, H, W, Cc, r.shape[1], crop[0], crop[1], pool,
                                       int(want_winner), _ptr(workspace), workspace.numel(),
                                       _stream(r))
    if rc < 0:
        check("frcnn_roi_fwd_prepare", rc)
    return rc == 1


def roi_crop_pool_bwd(dY: torch.Tensor, winner: Optional[torch.Tensor], rois: torch.Tensor,
                      fmap_shape: Tuple[int, int, int, int], crop: Tuple[int, int] = (14, 14),
                      pool: int = 2, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dY [B,R,ph,pw,C], winner u8 of the same shape (pool 2), rois [B,R,4] -> the gradient of
    the feature map [B,H,W,C] (written in full; `out`: a contiguous float32 tensor to write into)."""
    g = _cont(dY, "dY", torch.float32)
    r = _cont(rois, "rois", torch.float32)
    w = _cont(winner, "winner", torch.uint8) if winner is not None else None
    B, H, W, Cc = fmap_shape
    R = r.shape[1]
    if out is None:
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
    elif (out.dtype != torch.float32 or tuple(out.shape) != (B, H, W, Cc) or not out.is_contiguous()
          or out.device != g.device):
        raise ValueError("out must be a contiguous float32 [B,H,W,C] tensor on dY's device")
    with torch.cuda.device(g.device):
        check("frcnn_roi_crop_pool_bwd", lib.frcnn_roi_crop_pool_bwd(
            _ptr(g), _ptr(w), _ptr(r), B, R, H, W, Cc, crop[0], crop[1], pool, _ptr(out),
            _stream(g)))
    return out


def label_proposals(proposed: torch.Tensor, gt: torch.Tensor, gt_label: torch.Tensor,
                    fg_thr: float = 0.5, iou_floor: float = 0.1, bg_label: int = 20,
                    num_classes: int = 21, want_onehot: bool = True):
    p = _cont(proposed, "proposed", torch.float32)
    g = _cont(gt, "gt", torch.float32)
    gl = _cont(gt_label, "label", torch.int64)
    B, R = p.shape[0], p.shape[1]
    G = g.shape[1]
    dev = p.device
    out = dict(
        p_label=torch.empty((B, R), dtype=torch.int64, device=dev),
        gt_mask=torch.empty((B, R), dtype=torch.int32, device=dev),
        gt_onehot=torch.empty((B, R, num_classes), dtype=torch.float32, device=dev)
        if want_onehot else None,
        gt_coord=torch.empty((B, R, 4), dtype=torch.float32, device=dev),
        best_gt=torch.empty((B, R), dtype=torch.int32, device=dev),
        max_iou=torch.empty((B, R), dtype=torch.float32, device=dev),
    )
    with torch.cuda.device(dev):
        check("frcnn_label_proposals", lib.frcnn_label_proposals(
            _ptr(p), _ptr(g), _ptr(gl), B, R, G, fg_thr, iou_floor, bg_label, num_classes,
            _ptr(out["p_label"]), _ptr(out["gt_mask"]), _ptr(out["gt_onehot"]),
            _ptr(out["gt_coord"]), _ptr(out["best_gt"]), _ptr(out["max_iou"]), _stream(p)))
    return out


def sample_proposals(p_label: torch.Tensor, bg_label: int = 20, keys=None, seed: int = 0,
                     image_base: int = 0, batch: int = 128, max_pos: int = 32):
    """p_label [B,R] i64 -> (tindex [B,batch] i64, counts [B,4] i32)."""
    pl = _cont(p_label, "p_label", torch.int64)
    B, R = pl.shape
    kp, kn, ka = _keys(keys, B, R, batch, pl.device)
    tindex = torch.empty((B, batch), dtype=torch.int64, device=pl.device)
    counts = torch.empty((B, 4), dtype=torch.int32, device=pl.device)
    with torch.cuda.device(pl.device):
        check("frcnn_sample_proposals", lib.frcnn_sample_proposals(
            _ptr(pl), B, R, bg_label, _ptr(kp), _ptr(kn), _ptr(ka), seed & 0xFFFFFFFFFFFFFFFF,
            image_base, batch, max_pos, _ptr(tindex), _ptr(counts), _stream(pl)))
    return tindex, counts


def detect_postprocess(proposed: torch.Tensor, cls_prob: torch.Tensor, reg: torch.Tensor,
                       bg_label: int = 20, max_out: int = 30, iou_threshold: float = 0.7,
                       score_threshold: float = 0.9):
    """proposed [B,R,4], cls_prob [B,R,NC], reg [B,R,NC*4] -> dict(boxes [B,max_out,4],
    scores, classes, src, count, decoded [B,R,4], cls_arg [B,R])."""
    p = _cont(proposed, "proposed", torch.float32)
    c = _cont(cls_prob, "cls_prob", torch.float32)
    B, R, NC = c.shape
    g = _cont(reg, "reg", torch.float32).reshape(B, R, NC, 4)
    if tuple(p.shape) != (B, R, 4):
        raise ValueError("proposed must be [B,R,4]")
    dev = p.device
    out = dict(
        decoded=torch.empty((B, R, 4), dtype=torch.float32, device=dev),
        cls_arg=torch.empty((B, R), dtype=torch.int32, device=dev),
        boxes=torch.empty((B, max_out, 4), dtype=torch.float32, device=dev),
        scores=torch.empty((B, max_out), dtype=torch.float32, device=dev),
        classes=torch.empty((B, max_out), dtype=torch.int32, device=dev),
        src=torch.empty((B, max_out), dtype=torch.int32, device=dev),
        count=torch.empty((B,), dtype=torch.int32, device=dev),
    )
    with torch.cuda.device(dev):
        check("frcnn_detect_postprocess", lib.frcnn_detect_postprocess(
            _ptr(p), _ptr(c), _ptr(g), B, R, NC, bg_label, max_out, iou_threshold, score_threshold,
            _ptr(out["decoded"]), _ptr(out["cls_arg"]), _ptr(out["boxes"]), _ptr(out["scores"]),
            _ptr(out["classes"]), _ptr(out["src"]), _ptr(out["count"]), _stream(p)))
    return out


def rpn_loss(pred_obj: torch.Tensor, pred_reg: torch.Tensor, pos: torch.Tensor,
             label: torch.Tensor, reg_gt: torch.Tensor, want_grad: bool = True):
    """pred_obj [B,H,W,A], pred_reg [B,H,W,A,4] (or 4A packed), pos [B,S,3] i64, label [B,S],
    reg_gt [B,S,4] -> dict(loss [3] = (cls, reg, 2*cls+reg), partial [B,4], d_obj, d_reg)."""
    po = _cont(pred_obj, "pred_obj", torch.float32)
    B, H, W, A = po.shape
    pr = _cont(pred_reg, "pred_reg", torch.float32).reshape(B, H, W, A, 4)
    ps = _cont(pos, "pos", torch.int64)
    S = ps.shape[1]
    lb = _cont(label, "label", torch.float32).reshape(B, S)
    rg = _cont(reg_gt, "reg_gt", torch.float32).reshape(B, S, 4)
    dev = po.device
    out = dict(loss=torch.empty(3, dtype=torch.float32, device=dev),
               partial=torch.empty((B, 4), dtype=torch.float32, device=dev),
               d_obj=torch.empty_like(po) if want_grad else None,
               d_reg=torch.empty_like(pr) if want_grad else None)
    with torch.cuda.device(dev):
        check("frcnn_rpn_loss", lib.frcnn_rpn_loss(
            _ptr(po), _ptr(pr), _ptr(ps), _ptr(lb), _ptr(rg), B, H, W, A, S, _ptr(out["loss"]),
            _ptr(out["partial"]), _ptr(out["d_obj"]), _ptr(out["d_reg"]), _stream(po)))
    return out


def head_loss(cls_prob: torch.Tensor, reg: torch.Tensor, tindex: torch.Tensor,
              label: torch.Tensor, gt_mask: torch.Tensor, gt_coord: torch.Tensor,
              want_grad: bool = True):
    """cls_prob [B,R,K], reg [B,R,K*4], tindex [B,S] i64, label [B,S,K], gt_mask [B,S],
    gt_coord [B,S,4] -> dict(loss [3], partial [B,4], d_cls [B,R,K], d_reg [B,R,K,4])."""
    cp = _cont(cls_prob, "cls_prob", torch.float32)
    B, R, K = cp.shape
    rg = _cont(reg, "reg", torch.float32).reshape(B, R, K, 4)
    ti = _cont(tindex, "tindex", torch.int64)
    S = ti.shape[1]
    lb = _cont(label, "label", torch.float32).reshape(B, S, K)
    gm = _cont(gt_mask.to(torch.int32), "gt_mask", torch.int32).reshape(B, S)
    gc = _cont(gt_coord, "gt_coord", torch.float32).reshape(B, S, 4)
    dev = cp.device
    out = dict(loss=torch.empty(3, dtype=torch.float32, device=dev),
               partial=torch.empty((B, 4), dtype=torch.float32, device=dev),
               d_cls=torch.empty_like(cp) if want_grad else None,
               d_reg=torch.empty_like(rg) if want_grad else None)
    with torch.cuda.device(dev):
        check("frcnn_head_loss", lib.frcnn_head_loss(
            _ptr(cp), _ptr(rg), _ptr(ti), _ptr(lb), _ptr(gm), _ptr(gc), B, R, S, K,
            _ptr(out["loss"]), _ptr(out["partial"]), _ptr(out["d_cls"]), _ptr(out["d_reg"]),
            _stream(cp)))
    return out


def pad_gt(boxes: torch.Tensor, labels: Optional[torch.Tensor], offsets: torch.Tensor, G: int,
           flip: Optional[torch.Tensor] = None, use_flip: bool = False, seed: int = 0,
           image_base: int = 0):
    """Ragged boxes [total,4] / labels [total] with offsets [B+1] -> (gt [B,G,4], gt_label
    [B,G] i64 | None, count [B] i32); optional horizontal flip of the boxes."""
    off = _cont(offsets, "offsets", torch.int32)
    B = off.numel() - 1
    dev = off.device
    bx = _cont(boxes, "boxes", torch.float32).reshape(-1, 4)
    if bx.shape[0] == 0:
        bx = torch.zeros((1, 4), dtype=torch.float32, device=dev)
    lb = None if labels is None else _cont(labels, "labels", torch.int64)
    if lb is not None and lb.numel() == 0:
        lb = torch.zeros(1, dtype=torch.int64, device=dev)
    fl = None if flip is None else _cont(flip, "flip", torch.uint8)
    gt = torch.empty((B, G, 4), dtype=torch.float32, device=dev)
    gl = None if lb is None else torch.empty((B, G), dtype=torch.int64, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check("frcnn_pad_gt", lib.frcnn_pad_gt(
            _ptr(bx), _ptr(lb), _ptr(off), _ptr(fl), int(bool(use_flip)),
            seed & 0xFFFFFFFFFFFFFFFF, image_base, B, G, _ptr(gt), _ptr(gl), _ptr(cnt),
            _stream(off)))
    return gt, gl, cnt


def flip_images(images: torch.Tensor, flip: Optional[torch.Tensor] = None, seed: int = 0,
                image_base: int = 0) -> torch.Tensor:
    """images [B,H,W,C] f32 -> copy with the selected images flipped left-right."""
    im = _cont(images, "images", torch.float32)
    B, H, W, Cc = im.shape
    fl = None if flip is None else _cont(flip, "flip", torch.uint8)
    out = torch.empty_like(im)
    with torch.cuda.device(im.device):
        check("frcnn_flip_images", lib.frcnn_flip_images(
            _ptr(im), _ptr(fl), seed & 0xFFFFFFFFFFFFFFFF, image_base, B, H, W, Cc, _ptr(out),
            _stream(im)))
    return out


def fc1_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
            relu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The head's first dense layer (frcnn_fc1_fwd; module/model.py:31-32).

    x [..., K] float32 (the pooled tensor [B,R,ph,pw,C] is taken as [B*R, ph*pw*C], a view),
    weight [K,N] (Keras kernel layout), bias [N] or None -> [..., N] = act(x @ weight + bias)."""
    w = _cont(weight, "weight", torch.float32)
    if w.dim() != 2:
        raise ValueError("weight must be [K,N]")
    K, N = w.shape
    xx = _cont(x, "x", torch.float32)
    if xx.dim() == 5:                                  # pooled [B,R,ph,pw,C]
        lead = tuple(xx.shape[:2])
        xx = xx.reshape(lead[0] * lead[1], -1)
    else:
        lead = tuple(xx.shape[:-1])
        xx = xx.reshape(-1, xx.shape[-1])
    if xx.shape[1] != K:
        raise ValueError(f"x has {xx.shape[1]} features per row, weight expects {K}")
    b = None if bias is None else _cont(bias, "bias", torch.float32).reshape(N)
    M = xx.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=xx.device)
    elif out.dtype != torch.float32 or out.numel() != M * N or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of M*N elements")
    with torch.cuda.device(xx.device):
        check("frcnn_fc1_fwd", lib.frcnn_fc1_fwd(_ptr(xx), _ptr(w), _ptr(b), M, K, N, int(bool(relu)),
                                                 _ptr(out), _stream(xx)))
    return out.view(*lead, N)
