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

With `depth=2` consecutive batches are software-pipelined: stages 3-4 (latency-bound, a few
dozen small launches) go to a proposal stream and write one of two buffer sets, so the
proposals of batch i+1 are computed while the RoI kernel of batch i (bandwidth-bound, fills
the chip) is still running on the caller's stream.  Results of a run() are valid on the
caller's stream as before; the set they live in is reused two calls later.  The proposal
stream has normal priority on purpose: a high-priority stream in this wait pattern stopped
making progress on MI355X (ROCm 7.2) once earlier streams of the process shared the hardware
queues.
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


def _lib_tuning(key, default):
    from ._lib import get_tuning
    v = get_tuning(key)
    return default if v is None else int(v)


class RegionProposalPath:
    def __init__(self, cfg: PathConfig = REFERENCE, batch: int = 1, channels: int = 512,
                 device="cuda", train: bool = True, want_winner: bool = False,
                 fuse_pool: bool = True, depth: int = 1, topk: str = "sort",
                 alloc_pooled: bool = True, fmap_planar: bool = False, rpn_targets: bool = True,
                 nms: str = "auto"):
        if depth not in (1, 2):
            raise ValueError("depth must be 1 or 2")
        if topk not in ("sort", "lds"):
            raise ValueError("topk must be 'sort' (radix sort, small workgroups) or 'lds'")
        self.topk = topk
        if nms not in ("auto", "chunked", "fused"):
            raise ValueError("nms must be 'auto', 'chunked' or 'fused'")
        # one image at a time (the reference's inference call, model_train.py:741-743) is a chain
        # of dependent launches in the chunked NMS: frcnn_nms_fused runs it as one launch.  From a
        # few images on the chunked form wins (its mask tiles of all images run side by side).
        if nms == "auto":
            limit = _lib_tuning("nms_fused_max_b", 1)
            nms = "fused" if depth == 1 and batch <= min(limit, ops.NMS_FUSED_MAX_B) else "chunked"
        if nms == "fused" and batch > ops.NMS_FUSED_MAX_B:
            raise ValueError(f"the one-launch NMS takes at most {ops.NMS_FUSED_MAX_B} images")
        self.nms = nms
        self.cfg, self.B, self.C = cfg, batch, channels
        self.device = torch.device(device)
        self.train, self.fuse_pool, self.depth = train, fuse_pool, depth
        # rpn_targets=False: train without stages 1-2 (making_frcnn_input's share of the targets;
        # the reference computes the RPN mini-batch elsewhere, in patch_batch)
        self.rpn_targets = bool(train and rpn_targets)
        self.side = torch.cuda.Stream(self.device) if train else None
        self.prop_stream = torch.cuda.Stream(self.device) if depth == 2 else None
        # the RoI geometry (frcnn_roi_fwd_prepare) is computed on the proposal stream right behind
        # the NMS, not in front of the RoI kernel (a stream of its own for it measured the same)
        self._calls = 0
        dev, B, N = self.device, batch, cfg.N
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.anchor = ops.anchors(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h, dev)
        self.pool = cfg.pool if fuse_pool else 1
        ph, pw = cfg.crop[0] // self.pool, cfg.crop[1] // self.pool
        K, R = cfg.pre_k, cfg.post_n
        # alloc_pooled=False: every run() is called with skip_roi=True (the caller owns the
        # pooled tensor), so the path holds no output buffer for the crop + pool stage
        # workspace of frcnn_roi_crop_pool_fwd_ws; fmap_planar: the caller's feature maps are
        # stored [B,C,H,W] (passed as the permuted [B,H,W,C] view) and are transposed into this
        # workspace before the channels-last RoI kernel runs
        # One workspace per buffer set: the sample geometry of a batch's RoIs is written on the
        # proposal stream right after the NMS (frcnn_roi_fwd_prepare), while the RoI kernel of the
        # batch before may still be reading its own.
        self.fmap_planar = fmap_planar
        roi_ws_bytes = ops.roi_fwd_workspace_bytes(B, R, (cfg.H, cfg.W, channels) if fmap_planar else None)
        self.pooled = e((B, R, ph, pw, channels), f32) if alloc_pooled else None
        self.winner = e((B, R, ph, pw, channels), u8) if want_winner and alloc_pooled else None
        self._sets = []
        for _ in range(depth):
            d = dict(top_scores=e((B, K), f32), top_idx=e((B, K), i32), cand=e((B, K, 4), f32),
                     nms_ws=e((ops.nms_fused_workspace_bytes(B, K, R) if self.nms == "fused"
                               else ops.nms_workspace_bytes(B, K, R),), u8),
                     topk_ws=e((ops.topk_sort_workspace_bytes(B, N) if topk == "sort" else 1,), u8),
                     proposed=e((B, R, 4), f32), keep_pos=e((B, R), i32), keep_count=e((B,), i32),
                     roi_ws=e((roi_ws_bytes,), u8),
                     ev=[torch.cuda.Event() for _ in range(5)])
            if self.rpn_targets:
                S = cfg.rpn_batch
                d.update(max_iou=e((B, N), f32), best_gt=e((B, N), i32), flags=e((B, N), u8),
                         rpn=dict(label=e((B, S), f32), anchor=e((B, S, 4), f32),
                                  gt=e((B, S, 4), f32), gt_list=e((B, S), i64),
                                  pos=e((B, S, 3), i64), reg=e((B, S, 4), f32),
                                  counts=e((B, 4), i32)))
            if train:
                S2 = cfg.head_batch
                d.update(p_label=e((B, R), i64), gt_mask=e((B, R), i32),
                         gt_onehot=e((B, R, cfg.num_classes), f32), gt_coord=e((B, R, 4), f32),
                         p_best=e((B, R), i32), p_iou=e((B, R), f32),
                         tindex=e((B, S2), i64), head_counts=e((B, 4), i32))
            self._sets.append(d)
        self._bind(self._sets[0])

    def _bind(self, d):
        """Expose one buffer set under the attribute names callers read results from."""
        for k, v in d.items():
            if k != "ev":
                setattr(self, k, v)

    def capture(self, fmap, score, coord, gt=None, label=None, seed: int = 0,
                image_base: int = 0):
        """Records one run() over these tensors into a HIP graph and returns it.

        `graph.replay()` then re-issues the whole path (some 45 kernel launches) with one
        host call: for a single image the path is launch-bound, and the replay is what a
        serving loop would call.  The tensors are read in place at every replay (copy new
        inputs into them); results land in the same buffers as after run().  Scalars are
        launch arguments, so a replay repeats `seed`: meant for inference, or for training
        targets that may repeat their sampling.  Needs depth == 1."""
        if self.depth != 1:
            raise ValueError("capture() records one batch at a time: build the path with depth=1")
        warm = torch.cuda.Stream(self.device)
        warm.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(warm):                      # first launches load the code objects
            self.run(fmap, score, coord, gt, label, seed=seed, image_base=image_base)
        torch.cuda.current_stream(self.device).wait_stream(warm)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.run(fmap, score, coord, gt, label, seed=seed, image_base=image_base)
        return graph

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
            seed: int = 0, image_base: int = 0, roi_events=None, inputs_ready=None,
            nms_events=None, skip_roi: bool = False):
        """fmap [B,H,W,C] channels-last, score [B,H,W,A], coord [B,H,W,4A],
        gt [B,G,4], label [B,G] i64 (both only when train).  Returns self.

        inputs_ready: an event after which the inputs may be read.  Without it the inputs are
        taken to be ready where the caller's stream stands at the call, which with depth=2
        orders this batch's proposals behind everything already queued there (no overlap).
        skip_roi: leave the crop + pool stage to the caller (a training step runs it as a
        differentiable op on the proposals this call produced)."""
        cfg, B, N = self.cfg, self.B, self.cfg.N
        self._validate(fmap, score, coord, gt, label)
        if self.pooled is None and not skip_roi:
            raise ValueError("this path was built with alloc_pooled=False: call run(skip_roi=True)")
        d = self._sets[self._calls % self.depth]
        self._calls += 1
        self._bind(d)
        ev = d["ev"]
        main = torch.cuda.current_stream(self.device)
        prop = self.prop_stream if self.depth == 2 else main
        st, ps = main.cuda_stream, prop.cuda_stream
        seed &= 0xFFFFFFFFFFFFFFFF
        # inputs were produced on the caller's stream; this set's previous user (two calls
        # ago) finished reading it before the point ev[4] marks
        if inputs_ready is not None:
            ready = inputs_ready
        else:
            ready = ev[0]
            ready.record(main)
        if self.depth == 2:
            prop.wait_event(ready)
        elif inputs_ready is not None:
            main.wait_event(ready)                     # inputs may come from another stream
        if self.topk == "sort":
            check("frcnn_topk_sort", lib.frcnn_topk_sort(
                _p(score), B, N, cfg.pre_k, _p(self.anchor), _p(coord), _p(d["top_scores"]),
                _p(d["top_idx"]), _p(d["cand"]), _p(d["topk_ws"]), d["topk_ws"].numel(), ps))
        else:
            check("frcnn_topk", lib.frcnn_topk(
                _p(score), B, N, cfg.pre_k, _p(self.anchor), _p(coord), _p(d["top_scores"]),
                _p(d["top_idx"]), _p(d["cand"]), ps))
        # The top-k writes buffers that only this stream reads, so it may run while the set's
        # previous user (two calls ago) is still busy; the NMS rewrites `proposed`, which that
        # user's RoI and target kernels read until the point ev[4] marks.  (With small images
        # the proposal chain is the longer one: it has to start as early as it can.)
        if self.depth == 2 and self._calls > self.depth:
            prop.wait_event(ev[4])
        if nms_events is not None:
            nms_events[0].record(prop)
        nms_fn = lib.frcnn_nms_fused if self.nms == "fused" else lib.frcnn_nms
        check("frcnn_nms", nms_fn(
            _p(d["cand"]), B, cfg.pre_k, cfg.nms_thr, cfg.post_n, _p(d["proposed"]),
            _p(d["keep_pos"]), _p(d["keep_count"]), _p(d["nms_ws"]), d["nms_ws"].numel(), ps))
        if nms_events is not None:
            nms_events[1].record(prop)
        ev[1].record(prop)                             # proposals ready
        # (the host queues the proposal chain first: with small images it is the longer one)
        if self.train:
            self.side.wait_event(ready)
            if self.depth == 2 and self._calls > self.depth:
                self.side.wait_event(ev[4])
        if self.rpn_targets:
            G = gt.shape[1]
            ss = self.side.cuda_stream
            check("frcnn_iou_assign", lib.frcnn_iou_assign(
                _p(self.anchor), N, _p(gt), B, G, cfg.rpn_pos_thr, cfg.rpn_neg_thr,
                _p(d["max_iou"]), _p(d["best_gt"]), _p(d["flags"]), None, ss))
            r = d["rpn"]
            check("frcnn_rpn_sample", lib.frcnn_rpn_sample(
                _p(self.anchor), cfg.H, cfg.W, cfg.A, _p(gt), B, G, _p(d["best_gt"]),
                _p(d["flags"]), None, None, None, seed, image_base, cfg.rpn_batch, cfg.rpn_max_pos,
                _p(r["label"]), _p(r["anchor"]), _p(r["gt"]), _p(r["gt_list"]), _p(r["pos"]),
                _p(r["reg"]), _p(r["counts"]), ss))
        prepared = 0
        if not skip_roi:
            prep = prop
            prepared = lib.frcnn_roi_fwd_prepare(
                _p(d["proposed"]), B, cfg.H, cfg.W, self.C, cfg.post_n, cfg.crop[0], cfg.crop[1],
                self.pool, int(self.winner is not None), _p(d["roi_ws"]), d["roi_ws"].numel(),
                prep.cuda_stream)
            if prepared < 0:
                check("frcnn_roi_fwd_prepare", prepared)
            if self.depth == 2:
                ev[3].record(prep)
        if self.depth == 2:
            # one wait on `main` (every event packet between two RoI kernels is time in which
            # the stream does nothing): the geometry is written behind the NMS on the same stream
            main.wait_event(ev[1] if skip_roi else ev[3])
        if self.train:
            self.side.wait_event(ev[1])
            ss = self.side.cuda_stream
            check("frcnn_label_proposals", lib.frcnn_label_proposals(
                _p(d["proposed"]), _p(gt), _p(label), B, cfg.post_n, gt.shape[1], cfg.fg_thr,
                cfg.iou_floor, cfg.bg_label, cfg.num_classes, _p(d["p_label"]), _p(d["gt_mask"]),
                _p(d["gt_onehot"]), _p(d["gt_coord"]), _p(d["p_best"]), _p(d["p_iou"]), ss))
            check("frcnn_sample_proposals", lib.frcnn_sample_proposals(
                _p(d["p_label"]), B, cfg.post_n, cfg.bg_label, None, None, None, seed, image_base,
                cfg.head_batch, cfg.head_max_pos, _p(d["tindex"]), _p(d["head_counts"]), ss))
            ev[2].record(self.side)
        sb, sh, sw, sc = fmap.stride()
        if roi_events is not None:
            roi_events[0].record(main)
        if not skip_roi:
            fwd = lib.frcnn_roi_crop_pool_fwd_prepared if prepared == 1 else lib.frcnn_roi_crop_pool_fwd_ws
            check("frcnn_roi_crop_pool_fwd_ws", fwd(
                _p(fmap), B, cfg.H, cfg.W, self.C, sb, sh, sw, sc, _p(d["proposed"]), cfg.post_n,
                cfg.crop[0], cfg.crop[1], self.pool, _p(self.pooled), _p(self.winner),
                _p(d["roi_ws"]), d["roi_ws"].numel(), st))
        if roi_events is not None:
            roi_events[1].record(main)
        if self.train:
            main.wait_event(ev[2])                     # join: everything is ordered on `main` again
        ev[4].record(main)                             # this set may be rewritten after here
        return self


# ---- buffer sets for the reference-named functions ---------------------------------------------
# process_fmap / making_frcnn_input (module/model.py, module/data.py) are plain functions in the
# reference; here they run on a RegionProposalPath that is built on first use for a
# (configuration, batch, channels, device, mode) and kept, so that a caller of the reference's
# names gets the workspace-backed kernels (band-resident RoI forward, radix-sort top-k) and no
# per-call allocation.  A few paths are kept (least recently used goes first): a path at the
# benchmark size holds a 12.8 GB pooled tensor.
_PATH_CACHE: "dict" = {}
_PATH_CACHE_MAX = 4


def cached_path(cfg: PathConfig, batch: int, channels: int, device, *, train: bool,
                fuse_pool: bool, want_winner: bool = False, fmap_planar: bool = False,
                depth: int = 2) -> RegionProposalPath:
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    key = (cfg, batch, channels, device, train, fuse_pool, want_winner, fmap_planar, depth)
    path = _PATH_CACHE.pop(key, None)
    if path is None:
        while len(_PATH_CACHE) >= _PATH_CACHE_MAX:
            _PATH_CACHE.pop(next(iter(_PATH_CACHE)))
        path = RegionProposalPath(cfg, batch, channels, device, train=train,
                                  want_winner=want_winner, fuse_pool=fuse_pool, depth=depth,
                                  fmap_planar=fmap_planar, rpn_targets=False)
    _PATH_CACHE[key] = path                      # most recently used last
    return path


def clear_path_cache() -> None:
    """Drops the buffer sets process_fmap / making_frcnn_input keep between calls."""
    _PATH_CACHE.clear()
