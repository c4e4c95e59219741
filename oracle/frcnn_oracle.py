"""CPU oracle for the Faster R-CNN region-proposal + RoI hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module; it is the checker used by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py.

PARITY UNPINNED: the reference (ttjh1234/ObjectDetection) is TensorFlow 2.3
code with no tests, no fixtures and no golden vectors, and TensorFlow is not
installable here.  Every function below is a strict float32, op-for-op numpy
restatement of the reference source it cites, with the TensorFlow op
behaviour (top_k, combined_non_max_suppression, crop_and_resize, MaxPool2D,
argmax, where) restated from TF 2.3's documented / CPU-kernel semantics.
The known-answer checks in tests/test_oracle_*.py pin what can be pinned
without TensorFlow (anchor counts derived from the reference formula,
hand-computed IoU / NMS / crop cases, torch-CPU cross-checks).

Conventions (all from the reference):
  * boxes are normalized [ymin, xmin, ymax, xmax], float32
  * anchor linear index idx = (y*W + x)*A + a   (module/data.py:117)
  * feature maps are NHWC
  * python float literals act as float32 constants (TF converts them to the
    tensor dtype); numpy >= 2 does the same for python scalars.
All sizes the reference hard-codes (31, 9, 42, 6000, 1500, 14, 256, 128 ...)
are parameters here whose defaults equal the reference literals.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
I64 = np.int64

# --------------------------------------------------------------------------
# a1  anchors                                   Faster_RCNN/module/anchor.py:10-44
# --------------------------------------------------------------------------

def make_anchor(H=31, W=31, scales=(128, 256, 512), ratios=(0.5, 1, 2),
                base_w=500.0, base_h=None):
    """Anchor grid [H,W,A,4] float32, f64 math then one cast.

    Follows module/anchor.py:10-44 (and shipdetection/train.py:128-162 for the
    32x32x12 variant): centre=(i+0.5)/dim, w=sqrt((s/base)^2/r), h=w*r, box =
    centre -/+ half extents clipped to [0,1]; anchor order a = scale-major,
    ratio-minor (anchor.py:23-27).  The reference divides both axes by 500
    (square input).  `base_h` generalises to non-square inputs: h is
    multiplied by base_w/base_h, which is exactly 1.0 (a bit-exact no-op) in
    the reference case.
    """
    if base_h is None:
        base_h = base_w
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    X, Y = np.meshgrid(x, y)                      # [H,W]
    cx = (X + 0.5) / np.float64(W)
    cy = (Y + 0.5) / np.float64(H)
    aspect = np.float64(base_w) / np.float64(base_h)
    out = []
    for s in scales:
        for r in ratios:
            s64, r64 = np.float64(s), np.float64(r)
            q = s64 / np.float64(base_w)
            w = np.sqrt((q * q) / r64)
            h = (w * r64) * aspect
            hw, hh = w / 2.0, h / 2.0
            xmin = np.clip(cx - hw, 0.0, 1.0)
            xmax = np.clip(cx + hw, 0.0, 1.0)
            ymin = np.clip(cy - hh, 0.0, 1.0)
            ymax = np.clip(cy + hh, 0.0, 1.0)
            out.append(np.stack([ymin, xmin, ymax, xmax], axis=2))   # [H,W,4]
    anchor = np.stack(out, axis=2)                # [H,W,A,4]
    return anchor.astype(F32)


def anchor_valid(anchor_flat):
    """All four coords strictly inside (0,1).  module/data.py:79-80."""
    a = np.asarray(anchor_flat, F32).reshape(-1, 4)
    return (((a > 0) & (a < 1)).sum(axis=1) == 4)


# --------------------------------------------------------------------------
# a2  anchor x GT IoU                              Faster_RCNN/module/data.py:82-96
# --------------------------------------------------------------------------

def iou_anchor_gt(anchor_flat, gt_box):
    """[N,4] x [G,4] -> [N,G] float32, op order of module/data.py:82-96."""
    a = np.asarray(anchor_flat, F32).reshape(-1, 1, 4)
    g = np.asarray(gt_box, F32).reshape(-1, 4)
    g_size = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])                # [G]
    a_size = (a[:, :, 2] - a[:, :, 0]) * (a[:, :, 3] - a[:, :, 1])    # [N,1]
    xmin = np.maximum(a[:, :, 1], g[:, 1])
    ymin = np.maximum(a[:, :, 0], g[:, 0])
    xmax = np.minimum(a[:, :, 3], g[:, 3])
    ymax = np.minimum(a[:, :, 2], g[:, 2])
    inter = (xmax - xmin) * (ymax - ymin)
    inter2 = np.where((xmax > xmin) & (ymax > ymin), inter, F32(0))
    inter3 = np.where(inter2 > 0, inter2, F32(0))
    union = g_size[None, :] + a_size - inter3
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter3 / union
    return iou.astype(F32)


# --------------------------------------------------------------------------
# counter-based random keys (build-defined; the reference shuffles unseeded)
# --------------------------------------------------------------------------

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
STREAM_RPN_POS, STREAM_RPN_NEG, STREAM_RPN_ALL = 0, 1, 2
STREAM_HEAD_POS, STREAM_HEAD_NEG, STREAM_HEAD_ALL = 3, 4, 5


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al. 2011).  All args broadcastable uint32."""
    c0 = np.asarray(c0, np.uint64); c1 = np.asarray(c1, np.uint64)
    c2 = np.asarray(c2, np.uint64); c3 = np.asarray(c3, np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def random_keys(n, stream, image, seed):
    """uint32 key of list position i: philox(ctr=(i,stream,image,0), key=seed)[0]."""
    i = np.arange(n, dtype=np.uint64)
    r0, _, _, _ = philox4x32_10(i, np.uint64(stream), np.uint64(image), np.uint64(0),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return r0


def shuffle_order(keys, m):
    """Permutation that `shuffle` applies to a list of length m.

    Build-defined replacement for tf.random.shuffle (module/data.py:120,126,134):
    element i of the list carries keys[i]; the shuffled list is the list
    ordered by (key, i) ascending.  With keys[perm[r]] = r this realises any
    explicit permutation `perm`.
    """
    k = np.asarray(keys)[:m].astype(np.uint64)
    idx = np.arange(m, dtype=np.uint64)
    return np.argsort((k << np.uint64(32)) | idx, kind="stable").astype(I64)


def keys_from_perm(perm, capacity=None):
    """Keys that make shuffle_order reproduce an explicit permutation."""
    perm = np.asarray(perm, I64)
    n = perm.shape[0] if capacity is None else capacity
    keys = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    keys[perm] = np.arange(perm.shape[0], dtype=np.uint32)
    return keys


# --------------------------------------------------------------------------
# a3/a4  RPN target assignment + encode          Faster_RCNN/module/data.py:70-151
# --------------------------------------------------------------------------

class ShortBatchError(ValueError):
    """Fewer negatives than `batch - nop2` (module/data.py:125-137 reads OOB)."""


def rpn_assign(anchor_box, gt_box, pos_thr=0.3, neg_thr=0.3):
    """Stage results of module/data.py:77-111 before any shuffling.

    Returns dict with
      valid      bool[N]
      best       int64[N]   argmax_g iou (first max)
      max_iou    f32[N]
      pos_anchor int64[P]   anchor ids of positives, ascending (data.py:106-107)
      pos_gt     int64[P]
      neg_pos    int64[Nn]  positions within the valid list (data.py:110) --
                 the reference then decodes them as anchor ids (data.py:118)
    """
    H, W, A = anchor_box.shape[:3]
    N = H * W * A
    a3 = np.asarray(anchor_box, F32).reshape(N, 4)
    valid = anchor_valid(a3)
    valid_ind = np.nonzero(valid)[0].astype(I64)
    iou = iou_anchor_gt(a3, gt_box)
    best = iou.argmax(axis=1).astype(I64)
    max_iou = iou[np.arange(N), best]
    pos = best[valid_ind]
    pos2 = iou[valid_ind, pos]
    survive = np.nonzero(pos2 > F32(pos_thr))[0]
    iou2 = iou[valid_ind]
    neg_pos = np.nonzero(iou2.max(axis=1) < F32(neg_thr))[0].astype(I64) \
        if valid_ind.size else np.zeros(0, I64)
    return dict(valid=valid, best=best, max_iou=max_iou.astype(F32), iou=iou,
                pos_anchor=valid_ind[survive], pos_gt=pos[survive], neg_pos=neg_pos)


def encode_targets(anchor, gt, clip_wh=False):
    """Corner-based box targets.  module/data.py:143-149 (clip_wh=False) and
    :252-259 (clip_wh=True: w_a,h_a clipped to [1e-2,1] first)."""
    anchor = np.asarray(anchor, F32); gt = np.asarray(gt, F32)
    w_a = anchor[..., 3] - anchor[..., 1]
    h_a = anchor[..., 2] - anchor[..., 0]
    if clip_wh:
        w_a = np.clip(w_a, F32(1e-2), F32(1))
        h_a = np.clip(h_a, F32(1e-2), F32(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = (gt[..., 1] - anchor[..., 1]) / w_a / F32(0.1)
        ty = (gt[..., 0] - anchor[..., 0]) / h_a / F32(0.1)
        tw = np.log((gt[..., 3] - gt[..., 1]) / w_a) / F32(0.2)
        th = np.log((gt[..., 2] - gt[..., 0]) / h_a) / F32(0.2)
    return np.stack([tx, ty, tw, th], axis=-1).astype(F32)


def patch_batch(data, anchor_box, keys=None, seed=0, image_index=0,
                batch=256, max_pos=128, pos_thr=0.3, neg_thr=0.3):
    """module/data.py:70-151 with explicit shuffle keys.

    keys = (kp, kn, kt): uint32 keys for the positive list, the negative list
    and range(batch); None -> counter keys from (seed, image_index).
    Returns the reference's 9-tuple (data.py:151).
    """
    image, gt_box, label = data["image"], np.asarray(data["bbox"], F32), data["label"]
    anchor_box = np.asarray(anchor_box, F32)
    H, W, A = anchor_box.shape[:3]
    G = gt_box.shape[0]
    st = rpn_assign(anchor_box, gt_box, pos_thr, neg_thr)
    P, Nn = st["pos_anchor"].shape[0], st["neg_pos"].shape[0]
    if keys is None:
        kp = random_keys(P, STREAM_RPN_POS, image_index, seed)
        kn = random_keys(Nn, STREAM_RPN_NEG, image_index, seed)
        kt = random_keys(batch, STREAM_RPN_ALL, image_index, seed)
    else:
        kp, kn, kt = keys
    nop2 = min(P, max_pos)
    non = batch - nop2
    if Nn < non:
        raise ShortBatchError(f"{Nn} negatives < {non} needed")

    def split(idx):   # data.py:117-118
        return (idx // A) // W, (idx // A) % W, idx % A

    py, px, pa = split(st["pos_anchor"])
    positive_ind = np.stack([py, px, pa, st["pos_gt"]], axis=1).reshape(-1, 4)
    ny, nx, na = split(st["neg_pos"])                       # quirk: list position as anchor id
    negative_ind = np.stack([ny, nx, na, np.full(Nn, G - 1, I64)], axis=1).reshape(-1, 4)

    pindex = positive_ind[shuffle_order(kp, P)][:nop2]
    nindex = negative_ind[shuffle_order(kn, Nn)][:non]
    pdata = anchor_box[pindex[:, 0], pindex[:, 1], pindex[:, 2]].reshape(-1, 4)
    ndata = anchor_box[nindex[:, 0], nindex[:, 1], nindex[:, 2]].reshape(-1, 4)

    label_list = np.concatenate([np.ones((nop2, 1), F32), np.zeros((non, 1), F32)], axis=0)
    anchor_list = np.concatenate([pdata, ndata], axis=0)
    gt_list = np.concatenate([pindex[:, 3], nindex[:, 3]], axis=0)
    batch_pos = np.concatenate([pindex[:, :3], nindex[:, :3]], axis=0)
    tindex = shuffle_order(kt, batch)

    batch_label = label_list[tindex]
    batch_anchor = anchor_list[tindex]
    gt_list = gt_list[tindex]
    batch_gt = gt_box[gt_list]
    batch_pos = batch_pos[tindex]
    batch_reg_gt = encode_targets(batch_anchor, batch_gt, clip_wh=False)
    return (image, batch_label, batch_anchor, batch_gt, gt_list.astype(I64),
            batch_pos.astype(I64), batch_reg_gt, label, gt_box)


# --------------------------------------------------------------------------
# a5/a12  decode + clip                    Faster_RCNN/module/data.py:154-198
# --------------------------------------------------------------------------

def _decode(anc, t, clip_wh):
    w_a = anc[..., 3] - anc[..., 1]
    h_a = anc[..., 2] - anc[..., 0]
    if clip_wh:
        w_a = np.clip(w_a, F32(1e-2), F32(1))
        h_a = np.clip(h_a, F32(1e-2), F32(1))
    x = t[..., 0] * w_a * F32(0.1) + anc[..., 1]
    y = t[..., 1] * h_a * F32(0.1) + anc[..., 0]
    with np.errstate(over="ignore"):
        w = np.exp(t[..., 2] * F32(0.2)) * w_a
        h = np.exp(t[..., 3] * F32(0.2)) * h_a
    x_max = x + w
    y_max = y + h
    y = np.clip(y, F32(0), F32(1))
    x = np.clip(x, F32(0), F32(1))
    y_max = np.clip(y_max, F32(0), F32(1))
    x_max = np.clip(x_max, F32(0), F32(1))
    return np.stack([y, x, y_max, x_max], axis=-1).astype(F32)


def inverse_trans(anchor_box, pred_reg):
    """module/data.py:154-176: [H,W,A,4] x [B,H,W,4A] -> [B,H,W,A,4]."""
    anchor_box = np.asarray(anchor_box, F32)
    H, W, A = anchor_box.shape[:3]
    t = np.asarray(pred_reg, F32).reshape(-1, H, W, A, 4)
    return _decode(anchor_box[None], t, clip_wh=False)


def generate_coord(proposed, pred_reg):
    """module/data.py:179-198: [B,R,4] x [B,R,4] -> [B,R,4]."""
    return _decode(np.asarray(proposed, F32), np.asarray(pred_reg, F32), clip_wh=True)


# --------------------------------------------------------------------------
# a6  proposal creator: top-k, gather, greedy NMS     Faster_RCNN/module/model.py:44-52
# --------------------------------------------------------------------------

def top_k(scores, k):
    """tf.math.top_k(sorted=True): descending values, equal values -> lower
    index first; int32 indices.  Call site module/model.py:45."""
    s = np.asarray(scores, F32)
    idx = np.argsort(-s, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(s, idx, axis=-1), idx.astype(np.int32)


def nms_iou(box, others):
    """IoU of one box against [S,4] boxes, TF NMS kernel formula:
    corners re-ordered by min/max; area<=0 on either side -> 0;
    inter = max(dy,0)*max(dx,0); iou = inter / (area_i + area_j - inter)."""
    b = np.asarray(box, F32); o = np.asarray(others, F32).reshape(-1, 4)
    ymin_i = np.minimum(b[0], b[2]); xmin_i = np.minimum(b[1], b[3])
    ymax_i = np.maximum(b[0], b[2]); xmax_i = np.maximum(b[1], b[3])
    ymin_j = np.minimum(o[:, 0], o[:, 2]); xmin_j = np.minimum(o[:, 1], o[:, 3])
    ymax_j = np.maximum(o[:, 0], o[:, 2]); xmax_j = np.maximum(o[:, 1], o[:, 3])
    area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i)
    area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j)
    iy0 = np.maximum(ymin_i, ymin_j); ix0 = np.maximum(xmin_i, xmin_j)
    iy1 = np.minimum(ymax_i, ymax_j); ix1 = np.minimum(xmax_i, xmax_j)
    inter = np.maximum(iy1 - iy0, F32(0)) * np.maximum(ix1 - ix0, F32(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (area_i + area_j - inter)
    iou = np.where((area_i <= 0) | (area_j <= 0), F32(0), iou)
    return iou.astype(F32)


def nms_keep(boxes_sorted, iou_threshold=0.7, max_out=1500):
    """Greedy NMS over boxes already in descending-score order (ties resolved
    by position: earlier first).  Returns kept positions, ascending.
    Follows tf.image.combined_non_max_suppression's per-class loop (call site
    module/model.py:51): pop best, reject if IoU > threshold (strict) with any
    selected box, stop at max_out."""
    boxes = np.asarray(boxes_sorted, F32).reshape(-1, 4)
    thr = F32(iou_threshold)
    sel = np.zeros((max_out, 4), F32)
    keep = []
    for i in range(boxes.shape[0]):
        if len(keep) >= max_out:
            break
        n = len(keep)
        if n and bool((nms_iou(boxes[i], sel[:n]) > thr).any()):
            continue
        sel[n] = boxes[i]
        keep.append(i)
    return np.asarray(keep, I64)


def combined_nms(boxes, scores=None, post_n=1500, iou_threshold=0.7):
    """[B,K,4] score-sorted boxes -> proposed [B,post_n,4] zero padded, clipped
    to [0,1]; plus keep positions [B,post_n] (-1 padded) and counts [B]."""
    boxes = np.asarray(boxes, F32)
    B = boxes.shape[0]
    proposed = np.zeros((B, post_n, 4), F32)
    keep_pos = np.full((B, post_n), -1, np.int32)
    counts = np.zeros(B, np.int32)
    for b in range(B):
        k = nms_keep(boxes[b], iou_threshold, post_n)
        counts[b] = k.shape[0]
        keep_pos[b, :k.shape[0]] = k
        proposed[b, :k.shape[0]] = np.clip(boxes[b][k], F32(0), F32(1))
    return proposed, keep_pos, counts


def proposal_creator(score, coord, anchor_box, pre_k=6000, post_n=1500, iou_threshold=0.7):
    """module/model.py:44-52.  score [B,H,W,A], coord [B,H,W,4A]."""
    anchor_box = np.asarray(anchor_box, F32)
    H, W, A = anchor_box.shape[:3]
    B = np.asarray(score).shape[0]
    pred_obj = np.asarray(score, F32).reshape(B, -1)
    a, b = top_k(pred_obj, pre_k)
    pred_value = inverse_trans(anchor_box, coord).reshape(B, -1, 4)
    cand = np.take_along_axis(pred_value, b.astype(I64)[..., None], axis=1)
    proposed, keep_pos, counts = combined_nms(cand, a, post_n, iou_threshold)
    return dict(top_scores=a, top_idx=b, cand=cand, proposed=proposed,
                keep_pos=keep_pos, counts=counts)


# --------------------------------------------------------------------------
# a7/a8  RoI crop (bilinear) + 2x2 max-pool   module/model.py:53-55, module/model.py:30
# --------------------------------------------------------------------------

def _crop_geometry(boxes, H, W, ch, cw):
    boxes = np.asarray(boxes, F32).reshape(-1, 4)
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    if ch > 1:
        hs = ((y2 - y1) * F32(H - 1)) / F32(ch - 1)
        in_y = (y1 * F32(H - 1))[:, None] + np.arange(ch, dtype=F32)[None, :] * hs[:, None]
    else:
        in_y = (0.5 * (y1 + y2).astype(np.float64) * np.float64(H - 1)).astype(F32)[:, None]
    if cw > 1:
        ws = ((x2 - x1) * F32(W - 1)) / F32(cw - 1)
        in_x = (x1 * F32(W - 1))[:, None] + np.arange(cw, dtype=F32)[None, :] * ws[:, None]
    else:
        in_x = (0.5 * (x1 + x2).astype(np.float64) * np.float64(W - 1)).astype(F32)[:, None]
    in_y = in_y.astype(F32); in_x = in_x.astype(F32)
    y_ok = (in_y >= 0) & (in_y <= F32(H - 1))      # NaN -> extrapolated (build-defined)
    x_ok = (in_x >= 0) & (in_x <= F32(W - 1))
    top = np.floor(in_y); bot = np.ceil(in_y)
    left = np.floor(in_x); right = np.ceil(in_x)
    ly = (in_y - top).astype(F32); lx = (in_x - left).astype(F32)

    def ix(v, hi):
        return np.clip(np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0), 0, hi).astype(I64)
    return dict(y_ok=y_ok, x_ok=x_ok, top=ix(top, H - 1), bot=ix(bot, H - 1),
                left=ix(left, W - 1), right=ix(right, W - 1), ly=ly, lx=lx)


def crop_and_resize(fmap, boxes, box_idx, crop=(14, 14), chunk=64):
    """tf.image.crop_and_resize, bilinear, extrapolation 0 (CPU kernel op order):
    top = TL + (TR-TL)*lx; bottom = BL + (BR-BL)*lx; out = top + (bottom-top)*ly.
    fmap [B,H,W,C]; boxes [M,4]; box_idx [M] -> [M,ch,cw,C].  module/model.py:54."""
    fmap = np.asarray(fmap, F32)
    _, H, W, C = fmap.shape
    ch, cw = crop
    boxes = np.asarray(boxes, F32).reshape(-1, 4)
    box_idx = np.asarray(box_idx, I64)
    M = boxes.shape[0]
    out = np.zeros((M, ch, cw, C), F32)
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        g = _crop_geometry(boxes[s:e], H, W, ch, cw)
        bi = box_idx[s:e][:, None, None]
        T = g["top"][:, :, None]; Bo = g["bot"][:, :, None]
        L = g["left"][:, None, :]; R = g["right"][:, None, :]
        TL = fmap[bi, T, L]; TR = fmap[bi, T, R]
        BL = fmap[bi, Bo, L]; BR = fmap[bi, Bo, R]
        lx = g["lx"][:, None, :, None]; ly = g["ly"][:, :, None, None]
        topv = TL + (TR - TL) * lx
        botv = BL + (BR - BL) * lx
        val = topv + (botv - topv) * ly
        ok = (g["y_ok"][:, :, None] & g["x_ok"][:, None, :])[..., None]
        out[s:e] = np.where(ok, val, F32(0))
    return out


def max_pool(crops, pool=2):
    """MaxPool2D((pool,pool)), stride pool, 'valid' (module/model.py:30) over
    [...,h,w,C].  Also returns the winner index within each window: first
    maximum in row-major window order, strict '>' (build-defined; TF returns
    no index)."""
    crops = np.asarray(crops, F32)
    *lead, h, w, C = crops.shape
    ph, pw = h // pool, w // pool
    v = crops[..., :ph * pool, :pw * pool, :].reshape(*lead, ph, pool, pw, pool, C)
    v = np.moveaxis(v, -4, -3)                       # [..., ph, pw, pool, pool, C]
    v = v.reshape(*lead, ph, pw, pool * pool, C)
    best = v[..., 0, :].copy()
    arg = np.zeros(best.shape, np.uint8)
    for k in range(1, pool * pool):
        m = v[..., k, :] > best
        best = np.where(m, v[..., k, :], best)
        arg = np.where(m, np.uint8(k), arg)
    return best, arg


def roi_crop_pool(fmap, proposed, crop=(14, 14), pool=2, chunk=64):
    """crop_and_resize of every proposal followed by max_pool, as the
    reference does in two steps (module/model.py:53-55 then :30).
    fmap [B,H,W,C], proposed [B,R,4] -> pooled [B,R,ph,pw,C], winner uint8."""
    fmap = np.asarray(fmap, F32)
    B = fmap.shape[0]
    proposed = np.asarray(proposed, F32)
    R = proposed.shape[1]
    C = fmap.shape[3]
    ph, pw = crop[0] // pool, crop[1] // pool
    out = np.zeros((B, R, ph, pw, C), F32)
    arg = np.zeros((B, R, ph, pw, C), np.uint8)
    for b in range(B):
        for s in range(0, R, chunk):
            e = min(R, s + chunk)
            crops = crop_and_resize(fmap, proposed[b, s:e], np.full(e - s, b, I64), crop, chunk)
            if pool > 1:
                out[b, s:e], arg[b, s:e] = max_pool(crops, pool)
            else:
                out[b, s:e] = crops
    return out, arg


def process_fmap(fmap, score, coord, anchor_box, pre_k=6000, post_n=1500,
                 iou_threshold=0.7, crop=(14, 14)):
    """module/model.py:43-57 -> (crop_fmap [B,post_n,ch,cw,C], proposed)."""
    pc = proposal_creator(score, coord, anchor_box, pre_k, post_n, iou_threshold)
    proposed = pc["proposed"]
    B = proposed.shape[0]
    box_idx = np.repeat(np.arange(B), post_n)
    crops = crop_and_resize(fmap, proposed.reshape(-1, 4), box_idx, crop)
    return crops.reshape(B, post_n, crop[0], crop[1], -1), proposed


def roi_crop_pool_bwd(dY, winner, fmap_shape, proposed, crop=(14, 14), pool=2):
    """Adjoint of roi_crop_pool w.r.t. fmap (absent in the reference: the crop
    is under stop_gradient, module/model.py:71).  Build-defined:
      * max-pool routes dY to the recorded winner sample only;
      * a sample outside the image (extrapolated) passes no gradient;
      * bilinear adjoint: dtop=(1-ly)*g, dbot=ly*g; TL+=(1-lx)*dtop,
        TR+=lx*dtop, BL+=(1-lx)*dbot, BR+=lx*dbot;
      * each feature cell sums its contributions in float32 in the fixed order
        RoI ascending, sample (sy,sx) row-major, corner order TL,TR,BL,BR.
    """
    B, H, W, C = fmap_shape
    dY = np.asarray(dY, F32); winner = np.asarray(winner)
    proposed = np.asarray(proposed, F32)
    R = proposed.shape[1]
    ch, cw = crop
    ph, pw = ch // pool, cw // pool
    dF = np.zeros((B, H, W, C), F32)
    one = F32(1)
    for b in range(B):
        g = _crop_geometry(proposed[b], H, W, ch, cw)
        for r in range(R):
            for sy in range(ph * pool):
                if not g["y_ok"][r, sy]:
                    continue
                for sx in range(pw * pool):
                    if not g["x_ok"][r, sx]:
                        continue
                    p, q = sy // pool, sx // pool
                    k = (sy % pool) * pool + (sx % pool)
                    cm = (winner[b, r, p, q] == k) if pool > 1 else np.ones(C, bool)
                    if not cm.any():
                        continue
                    ly = g["ly"][r, sy]; lx = g["lx"][r, sx]
                    t, bo = g["top"][r, sy], g["bot"][r, sy]
                    l, ri = g["left"][r, sx], g["right"][r, sx]
                    gv = np.where(cm, dY[b, r, p, q], F32(0))
                    dtop = (one - ly) * gv
                    dbot = ly * gv
                    dF[b, t, l] = np.where(cm, dF[b, t, l] + (one - lx) * dtop, dF[b, t, l])
                    dF[b, t, ri] = np.where(cm, dF[b, t, ri] + lx * dtop, dF[b, t, ri])
                    dF[b, bo, l] = np.where(cm, dF[b, bo, l] + (one - lx) * dbot, dF[b, bo, l])
                    dF[b, bo, ri] = np.where(cm, dF[b, bo, ri] + lx * dbot, dF[b, bo, ri])
    return dF


# --------------------------------------------------------------------------
# a9/a10/a11  proposal <-> GT labelling and sampling  module/data.py:215-329
# --------------------------------------------------------------------------

def label_proposals(proposed, gt_box, label, fg_thr=0.5, iou_floor=0.1,
                    bg_label=20, num_classes=21, no_overlap=1e-5):
    """module/data.py:215-259.  proposed [B,R,4], gt_box [B,G,4], label [B,G]."""
    proposed = np.asarray(proposed, F32)
    gt_box = np.asarray(gt_box, F32)
    label = np.asarray(label, I64)
    B, R, _ = proposed.shape
    G = gt_box.shape[1]
    gt = gt_box.reshape(B, 1, G, 4)
    gt_size = (gt[..., 2] - gt[..., 0]) * (gt[..., 3] - gt[..., 1])
    ac = proposed[:, :, None, :]
    ac_size = (ac[..., 2] - ac[..., 0]) * (ac[..., 3] - ac[..., 1])
    xmin = np.maximum(ac[..., 1], gt[..., 1]); ymin = np.maximum(ac[..., 0], gt[..., 0])
    xmax = np.minimum(ac[..., 3], gt[..., 3]); ymax = np.minimum(ac[..., 2], gt[..., 2])
    inter = (xmax - xmin) * (ymax - ymin)
    inter2 = np.where((xmax > xmin) & (ymax > ymin), inter, F32(0))
    inter3 = np.where(inter2 > 0, inter2, F32(no_overlap))
    union = gt_size + ac_size - inter3
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter3 / union).astype(F32)
    iou4 = np.where(iou >= F32(iou_floor), iou, F32(0))
    iou4 = np.clip(iou4, F32(0), F32(1))
    best = iou4.argmax(axis=2)                                   # [B,R]
    plabel = np.take_along_axis(label, best, axis=1)
    iou5 = np.take_along_axis(iou4, best[..., None], axis=2)[..., 0]
    p_iou = np.where(iou5 < F32(fg_thr), I64(bg_label), plabel)
    p_iou = np.where(iou5 >= F32(fg_thr), plabel, p_iou)
    gt_mask = np.where(p_iou != bg_label, 1, 0).astype(np.int32)
    gt_label = np.zeros((B, R, num_classes), F32)
    inr = (p_iou >= 0) & (p_iou < num_classes)
    bi, ri = np.nonzero(inr)
    gt_label[bi, ri, p_iou[bi, ri]] = 1
    p_coord = np.take_along_axis(gt_box, best[..., None], axis=1)
    gt_coord = encode_targets(proposed, p_coord, clip_wh=True)
    return dict(iou=iou, best=best.astype(I64), max_iou=iou5.astype(F32), p_iou=p_iou.astype(I64),
                gt_mask=gt_mask, gt_label=gt_label, p_coord=p_coord, gt_coord=gt_coord)


def sample_proposals(p_iou, keys=None, seed=0, image_base=0, batch=128, max_pos=32, bg_label=20):
    """module/data.py:266-316 generalised to any B: per image take
    nop=min(#pos,max_pos) shuffled positives, batch-nop shuffled negatives,
    shuffle the concatenation.  keys[b] = (kp, kn, kt).  Returns tindex [B,batch]."""
    p_iou = np.asarray(p_iou)
    B = p_iou.shape[0]
    tindex = np.zeros((B, batch), I64)
    for b in range(B):
        pos = np.nonzero(p_iou[b] != bg_label)[0]
        neg = np.nonzero(p_iou[b] == bg_label)[0]
        if keys is None:
            kp = random_keys(pos.size, STREAM_HEAD_POS, image_base + b, seed)
            kn = random_keys(neg.size, STREAM_HEAD_NEG, image_base + b, seed)
            kt = random_keys(batch, STREAM_HEAD_ALL, image_base + b, seed)
        else:
            kp, kn, kt = keys[b]
        nop = min(pos.size, max_pos)
        non = batch - nop
        if neg.size < non:
            raise ShortBatchError(f"image {b}: {neg.size} negatives < {non} needed")
        pindex = pos[shuffle_order(kp, pos.size)][:nop]
        nindex = neg[shuffle_order(kn, neg.size)][:non]
        t = np.concatenate([pindex, nindex])
        tindex[b] = t[shuffle_order(kt, batch)]
    return tindex


def making_frcnn_input(gt_box, label, fmap, pred_reg, pred_obj, anchor_box, keys=None, seed=0,
                       image_base=0, pre_k=6000, post_n=1500, iou_threshold=0.7, crop=(14, 14),
                       batch=128, max_pos=32, fg_thr=0.5, iou_floor=0.1, bg_label=20,
                       num_classes=21):
    """module/data.py:200-329 (anchor_box made an explicit argument: the
    reference reads an undefined global at :205)."""
    crop_fmap, proposed = process_fmap(fmap, pred_obj, pred_reg, anchor_box, pre_k, post_n,
                                       iou_threshold, crop)
    lab = label_proposals(proposed, gt_box, label, fg_thr, iou_floor, bg_label, num_classes)
    tindex = sample_proposals(lab["p_iou"], keys, seed, image_base, batch, max_pos, bg_label)
    gt_label2 = np.take_along_axis(lab["gt_label"], tindex[..., None], axis=1)
    gt_coord2 = np.take_along_axis(lab["gt_coord"], tindex[..., None], axis=1)
    gt_mask2 = np.take_along_axis(lab["gt_mask"], tindex, axis=1)
    return crop_fmap, gt_label2, gt_mask2, gt_coord2, proposed, tindex


# --------------------------------------------------------------------------
# synthetic inputs shared by tests and bench (seeded, PCG64)
# --------------------------------------------------------------------------

def synth_inputs(seed, B, H, W, A, C, G, n_gt=None, gt_side=(0.05, 0.6), num_classes=20,
                 with_fmap=True):
    """scores sigmoid(N(0,2)); deltas N(0,0.5); fmap relu(N(0,1)); GT uniform
    centres, log-uniform sides, zero-padded to G rows."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.normal(0.0, 2.0, size=(B, H, W, A)).astype(F32)
    score = (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    coord = rng.normal(0.0, 0.5, size=(B, H, W, 4 * A)).astype(F32)
    fmap = None
    if with_fmap:
        fmap = np.maximum(rng.standard_normal(size=(B, H, W, C), dtype=F32), F32(0))
    gt = np.zeros((B, G, 4), F32)
    label = np.zeros((B, G), I64)
    for b in range(B):
        n = int(rng.integers(1, G + 1)) if n_gt is None else min(n_gt, G)
        cy = rng.uniform(0.05, 0.95, n); cx = rng.uniform(0.05, 0.95, n)
        lo, hi = np.log(gt_side[0]), np.log(gt_side[1])
        hh = np.exp(rng.uniform(lo, hi, n)); ww = np.exp(rng.uniform(lo, hi, n))
        box = np.stack([cy - hh / 2, cx - ww / 2, cy + hh / 2, cx + ww / 2], axis=1)
        gt[b, :n] = np.clip(box, 0.0, 1.0).astype(F32)
        label[b, :n] = rng.integers(0, num_classes, n)
    return dict(score=score, coord=coord, fmap=fmap, gt=gt, label=label)


# --------------------------------------------------------------------------
# f1  detection post-process            Faster_RCNN/model_train.py:745-759
# --------------------------------------------------------------------------

def nms_with_scores(boxes, scores, max_out, iou_threshold=0.7):
    """tf.image.non_max_suppression_with_scores, hard NMS (soft_nms_sigma=0),
    score_threshold=-inf: candidates by descending score (equal scores: lower index
    first), greedy IoU > threshold, at most max_out.  Returns (indices, scores)."""
    boxes = np.asarray(boxes, F32).reshape(-1, 4)
    scores = np.asarray(scores, F32).reshape(-1)
    order = np.argsort(-scores, kind="stable")
    keep = nms_keep(boxes[order], iou_threshold, max_out)
    sel = order[keep]
    return sel.astype(np.int32), scores[sel]


def detect_postprocess(proposed, cls_prob, reg, bg_label=20, max_out=30, iou_threshold=0.7,
                       score_threshold=0.9):
    """model_train.py:745-759 per image: class argmax -> that class's offsets ->
    generate_coord -> drop background -> NMS with scores (max 30, IoU 0.7) -> keep
    score >= 0.9.  proposed [B,R,4], cls_prob [B,R,NC], reg [B,R,NC*4].
    Returns boxes [B,max_out,4], scores [B,max_out], classes [B,max_out] (-1 pad),
    src [B,max_out] (proposal index, -1 pad), count [B]."""
    proposed = np.asarray(proposed, F32)
    cls_prob = np.asarray(cls_prob, F32)
    B, R, NC = cls_prob.shape
    reg = np.asarray(reg, F32).reshape(B, R, NC, 4)
    argindex = cls_prob.argmax(axis=2)
    offset = np.take_along_axis(reg, argindex[..., None, None], axis=2)[:, :, 0, :]
    result = generate_coord(proposed, offset)
    pscore = np.take_along_axis(cls_prob, argindex[..., None], axis=2)[..., 0]
    boxes = np.zeros((B, max_out, 4), F32)
    scores = np.zeros((B, max_out), F32)
    classes = np.full((B, max_out), -1, np.int32)
    src = np.full((B, max_out), -1, np.int32)
    count = np.zeros(B, np.int32)
    for b in range(B):
        fg = np.nonzero(argindex[b] != bg_label)[0]
        sel, ssc = nms_with_scores(result[b][fg], pscore[b][fg], max_out, iou_threshold)
        ok = ssc >= F32(score_threshold)
        sel, ssc = sel[ok], ssc[ok]
        n = sel.shape[0]
        count[b] = n
        boxes[b, :n] = result[b][fg][sel]
        scores[b, :n] = ssc
        classes[b, :n] = argindex[b][fg][sel]
        src[b, :n] = fg[sel]
    return dict(boxes=boxes, scores=scores, classes=classes, src=src, count=count, decoded=result,
                argindex=argindex.astype(np.int32))


# --------------------------------------------------------------------------
# next row f2: loss assembly (model_train.py:598-600, 626-635, 639-657)
# --------------------------------------------------------------------------
KERAS_EPS = F32(1e-7)


def loss_bbr(y_true, y_pred):
    """module/loss.py:9-14 (threshold 1): element-wise smooth-L1 and d/d(y_pred)."""
    e = (np.asarray(y_true, F32) - np.asarray(y_pred, F32)).astype(F32)
    ae = np.abs(e)
    small = ae < F32(1)
    val = np.where(small, e * e / F32(2), F32(1) * ae - F32(0.5)).astype(F32)
    grad = np.where(small, -e, np.where(e > 0, F32(-1), F32(1))).astype(F32)
    return val, grad


def keras_bce(y, p):
    """tf.keras.backend.binary_crossentropy on probabilities (eager path): clip to
    [eps, 1-eps], -(y log(p+eps) + (1-y) log(1-p+eps)); value and d/dp (clip_by_value passes
    the gradient on [eps, 1-eps])."""
    y = np.asarray(y, F32)
    p = np.asarray(p, F32)
    pc = np.clip(p, KERAS_EPS, F32(1) - KERAS_EPS).astype(F32)
    a = (pc + KERAS_EPS).astype(F32)
    b = (F32(1) - pc + KERAS_EPS).astype(F32)
    v = (y * np.log(a)).astype(F32)
    v = (v + (F32(1) - y) * np.log(b)).astype(F32)
    inside = (p >= KERAS_EPS) & (p <= F32(1) - KERAS_EPS)
    g = np.where(inside, -(y / a - (F32(1) - y) / b), F32(0)).astype(F32)
    return (-v).astype(F32), g


def keras_cce(y, p):
    """tf.keras.backend.categorical_crossentropy on probabilities: q = p / sum(p), clip,
    -sum(y log q); value [...,] and d/dp [..., K]."""
    y = np.asarray(y, F32)
    p = np.asarray(p, F32)
    sp = p.sum(-1, keepdims=True, dtype=np.float64).astype(F32)
    q = (p / sp).astype(F32)
    qc = np.clip(q, KERAS_EPS, F32(1) - KERAS_EPS).astype(F32)
    val = -(y * np.log(qc)).sum(-1, dtype=np.float64)
    inside = (q >= KERAS_EPS) & (q <= F32(1) - KERAS_EPS)
    g = np.where(inside, -y / qc, F32(0)).astype(np.float64)        # dL/dq
    gp = (g * p).sum(-1, keepdims=True)
    dp = g / sp - gp / (sp.astype(np.float64) ** 2)
    return val, dp


def rpn_loss(pred_obj, pred_reg, pos, label, reg_gt):
    """model_train.py:626-635.  pred_obj [B,H,W,A], pred_reg [B,H,W,A,4], pos [B,S,3],
    label [B,S], reg_gt [B,S,4] -> (cls, reg, total), d_obj, d_reg (float64 sums)."""
    B, H, W, A = pred_obj.shape
    S = pos.shape[1]
    bi = np.repeat(np.arange(B), S)
    y, x, a = pos.reshape(-1, 3).T
    p = pred_obj[bi, y, x, a]
    lab = np.asarray(label, F32).reshape(-1)
    v, g = keras_bce(lab, p)
    cls = v.sum(dtype=np.float64) / (B * S)
    is_pos = lab.astype(np.int64) == 1
    hv, hg = loss_bbr(reg_gt.reshape(-1, 4)[is_pos], pred_reg[bi, y, x, a][is_pos])
    P = int(is_pos.sum())
    reg = hv.sum(dtype=np.float64) / (4 * P) if P else 0.0
    # the mini-batch can hold one (y,x,a) twice (a negative's position is a rank in the
    # valid-anchor list, module/data.py:108-118): the gradient of a gather accumulates
    d_obj = np.zeros(pred_obj.shape, np.float64)
    np.add.at(d_obj, (bi, y, x, a), 2.0 * g / (B * S))
    d_reg = np.zeros(pred_reg.shape, np.float64)
    if P:
        np.add.at(d_reg, (bi[is_pos], y[is_pos], x[is_pos], a[is_pos]), hg / (4.0 * P))
    return (cls, reg, 2.0 * cls + reg), d_obj, d_reg


def head_loss(cls_prob, reg, tindex, label, gt_mask, gt_coord):
    """model_train.py:639-657.  cls_prob [B,R,K], reg [B,R,K,4], tindex [B,S], label
    [B,S,K], gt_mask [B,S], gt_coord [B,S,4] -> (cls, reg, total), d_cls, d_reg."""
    B, R, K = cls_prob.shape
    S = tindex.shape[1]
    bi = np.repeat(np.arange(B), S)
    r = np.asarray(tindex).reshape(-1)
    lab = np.asarray(label, F32).reshape(-1, K)
    v, dp = keras_cce(lab, cls_prob[bi, r])
    cls = v.sum() / (B * S)
    m = np.asarray(gt_mask).reshape(-1) != 0
    c = lab.argmax(-1)
    gt = np.clip(np.asarray(gt_coord, F32).reshape(-1, 4), F32(-100), F32(100))
    hv, hg = loss_bbr(gt[m], reg[bi, r, c][m])
    P = int(m.sum())
    lreg = hv.sum(dtype=np.float64) / (4 * P) if P else 0.0
    d_cls = np.zeros(cls_prob.shape, np.float64)
    np.add.at(d_cls, (bi, r), 2.0 * dp / (B * S))                   # repeated rows accumulate
    d_reg = np.zeros(reg.shape, np.float64)
    if P:
        np.add.at(d_reg, (bi[m], r[m], c[m]), hg / (4.0 * P))
    return (cls, lreg, 2.0 * cls + lreg), d_cls, d_reg


# --------------------------------------------------------------------------
# next row f3: input side (module/data.py:9-52, shipdetection/train.py:28-45)
# --------------------------------------------------------------------------
def flip_decision(seed, image_index):
    """Build-defined replacement for `tf.random.uniform([]) > 0.5` (module/data.py:44-45):
    Philox counter key (position 0, stream 6, image) above 2^31."""
    return bool(random_keys(1, 6, image_index, seed)[0] > np.uint32(0x80000000))


def augmentation_data(img, label, bbox):
    """module/data.py:23-33: left-right flip of the image and of the boxes."""
    image = None if img is None else np.asarray(img)[..., ::-1, :].copy()
    b = np.asarray(bbox, F32)
    xmin = (F32(1) - b[:, 3]).astype(F32)
    xmax = (F32(1) - b[:, 1]).astype(F32)
    return image, label, np.stack([b[:, 0], xmin, b[:, 2], xmax], 1).astype(F32)


def pad_gt(bbox, label, G=42):
    """module/data.py:13-18: zero padding of one image's boxes / labels to G rows (more than G
    objects: the reference's negative pad raises; here the first G are kept)."""
    b = np.asarray(bbox, F32).reshape(-1, 4)[:G]
    out = np.zeros((G, 4), F32)
    out[:len(b)] = b
    lab = np.zeros(G, I64)
    if label is not None:
        l = np.asarray(label, I64)[:G]
        lab[:len(l)] = l
    return out, lab


def random_augmentation(img, label, bbox, G=42, flip=None, seed=0, image_index=0):
    """module/data.py:36-52 without the resize: optional flip, then padding."""
    if flip is None:
        flip = flip_decision(seed, image_index)
    if flip:
        img, label, bbox = augmentation_data(img, label, bbox)
    b, l = pad_gt(bbox, label, G)
    return img, b, l
