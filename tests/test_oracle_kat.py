"""Known-answer and cross-checks that pin the CPU oracle (no GPU needed).

The reference has no tests or golden vectors, so these are the pins that exist:
values derived from the reference's own formulas, hand-computed cases, the
published Philox-4x32-10 test vectors, and independent torch-CPU ops.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import frcnn_oracle as O

F32 = np.float32


# ---------------------------------------------------------------- anchors (a1)
def test_anchor_counts_reference_grid():
    a = O.make_anchor()
    assert a.shape == (31, 31, 9, 4) and a.dtype == np.float32
    v = O.anchor_valid(a)
    assert a.reshape(-1, 4).shape[0] == 8649
    assert int(v.sum()) == 2046
    assert v.reshape(31, 31, 9).sum((0, 1)).tolist() == [475, 529, 475, 171, 225, 171, 0, 0, 0]
    assert a.min() >= 0 and a.max() <= 1


def test_anchor_counts_ship_grid():
    a = O.make_anchor(32, 32, (32, 64, 128, 256), (0.5, 1, 2))
    assert a.shape == (32, 32, 12, 4)
    assert int(O.anchor_valid(a).sum()) == 7236


def test_anchor_formula_one_cell():
    a = O.make_anchor()
    # cell (y=15,x=15), scale 128 ratio 1: centre 15.5/31 = 0.5, half = 0.128
    np.testing.assert_allclose(a[15, 15, 1], [0.5 - 0.128, 0.5 - 0.128, 0.5 + 0.128, 0.5 + 0.128],
                               rtol=0, atol=1e-7)
    # ratio 0.5: w = 0.256*sqrt(2), h = w/2
    w = np.sqrt(0.256 ** 2 / 0.5)
    np.testing.assert_allclose(a[15, 15, 0], [0.5 - w / 4, 0.5 - w / 2, 0.5 + w / 4, 0.5 + w / 2],
                               rtol=0, atol=1e-7)


def test_anchor_nonsquare_reduces_to_reference():
    np.testing.assert_array_equal(O.make_anchor(31, 31, base_w=500.0, base_h=500.0), O.make_anchor())


# ------------------------------------------------------------------- IoU (a2)
def test_iou_hand_cases():
    a = np.array([[0.2, 0.2, 0.6, 0.6]], F32)
    g = np.array([[0.2, 0.2, 0.6, 0.6],      # identical -> 1
                  [0.7, 0.7, 0.9, 0.9],      # disjoint -> 0
                  [0.6, 0.2, 0.8, 0.6],      # touching edge -> 0
                  [0.2, 0.4, 0.6, 0.8],      # half overlap -> 1/3
                  [0.0, 0.0, 0.0, 0.0]], F32)  # zero pad row -> 0
    iou = O.iou_anchor_gt(a, g)[0]
    assert iou[0] == 1.0 and iou[1] == 0.0 and iou[2] == 0.0 and iou[4] == 0.0
    np.testing.assert_allclose(iou[3], 1.0 / 3.0, rtol=1e-6)


def test_rpn_assign_quirks():
    a = O.make_anchor()
    gt = np.zeros((42, 4), F32)
    gt[0] = [0.3, 0.3, 0.7, 0.7]
    st = O.rpn_assign(a, gt)
    # positives are valid anchors with best IoU > 0.3; negatives are positions in the valid list
    assert st["pos_anchor"].size > 0
    assert st["valid"][st["pos_anchor"]].all()
    assert (st["max_iou"][st["pos_anchor"]] > F32(0.3)).all()
    assert st["neg_pos"].max() < int(st["valid"].sum())
    out = O.patch_batch(dict(image=None, bbox=gt, label=np.zeros(42, np.int64)), a, seed=3)
    neg = out[1][:, 0] == 0
    assert (out[4][neg] == 41).all()                      # negatives carry GT index G-1
    assert (out[4][~neg] == 0).all()


def test_patch_batch_identity_permutation():
    a = O.make_anchor()
    d = O.synth_inputs(1, 1, 31, 31, 9, 4, 42, with_fmap=False)
    N = 8649
    ident = (np.arange(N, dtype=np.uint32), np.arange(N, dtype=np.uint32),
             np.arange(256, dtype=np.uint32))
    st = O.rpn_assign(a, d["gt"][0])
    out = O.patch_batch(dict(image=None, bbox=d["gt"][0], label=d["label"][0]), a, keys=ident)
    nop = min(st["pos_anchor"].size, 128)
    ids = (out[5][:, 0] * 31 + out[5][:, 1]) * 9 + out[5][:, 2]
    np.testing.assert_array_equal(ids[:nop], st["pos_anchor"][:nop])
    np.testing.assert_array_equal(ids[nop:], st["neg_pos"][:256 - nop])   # quirk: list positions
    # an explicit permutation is realised by keys_from_perm
    rng = np.random.default_rng(0)
    perm = rng.permutation(256)
    np.testing.assert_array_equal(O.shuffle_order(O.keys_from_perm(perm), 256), perm)


def test_short_batch_raises():
    a = O.make_anchor()
    gt = np.zeros((42, 4), F32)
    gt[:] = [0.0, 0.0, 1.0, 1.0]          # every valid anchor overlaps something a bit
    gt[0] = [0.01, 0.01, 0.99, 0.99]
    with pytest.raises(O.ShortBatchError):
        O.patch_batch(dict(image=None, bbox=gt, label=np.zeros(42, np.int64)), a, neg_thr=0.0)


# ------------------------------------------------------------ Philox-4x32-10
def test_philox_published_vectors():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = O.philox4x32_10(*[np.uint64(c) for c in ctr], key[0], key[1])
        assert tuple(int(np.asarray(g).reshape(-1)[0]) for g in got) == want


# --------------------------------------------------------- decode / encode (a4,a5)
def test_decode_zero_deltas_returns_clipped_anchor():
    a = O.make_anchor()
    out = O.inverse_trans(a, np.zeros((1, 31, 31, 36), F32))[0]
    np.testing.assert_allclose(out, a, rtol=0, atol=6e-8)
    assert out.min() >= 0 and out.max() <= 1


def test_encode_decode_roundtrip():
    rng = np.random.default_rng(0)
    anc = np.sort(rng.uniform(0.05, 0.95, (100, 2, 2)).astype(F32), axis=1).reshape(100, 4)
    anc = anc[:, [0, 2, 1, 3]]
    anc = np.stack([np.minimum(anc[:, 0], anc[:, 2]), np.minimum(anc[:, 1], anc[:, 3]),
                    np.maximum(anc[:, 0], anc[:, 2]) + 0.02, np.maximum(anc[:, 1], anc[:, 3]) + 0.02],
                   1).clip(0, 1).astype(F32)
    gt = (anc + rng.uniform(-0.01, 0.01, anc.shape)).clip(0.001, 0.999).astype(F32)
    gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 0.01)
    t = O.encode_targets(anc, gt, clip_wh=True)
    back = O.generate_coord(anc[None], t[None])[0]
    np.testing.assert_allclose(back, gt.clip(0, 1), atol=2e-6)


# -------------------------------------------------------------- top-k (a6)
def test_topk_ties_lower_index_first_and_matches_torch_on_distinct():
    s = np.array([[0.5, 1.0, 0.25, 1.0, 0.5, 1.0]], F32)
    v, i = O.top_k(s, 5)
    assert i.tolist() == [[1, 3, 5, 0, 4]]
    rng = np.random.default_rng(0)
    x = rng.permutation(5000).astype(F32)[None]
    v, i = O.top_k(x, 777)
    tv, ti = torch.topk(torch.from_numpy(x), 777, dim=1)
    np.testing.assert_array_equal(i, ti.numpy())
    np.testing.assert_array_equal(v, tv.numpy())


# ----------------------------------------------------------------- NMS (a6)
def _naive_nms(boxes, thr, max_out):
    keep = []
    for i in range(len(boxes)):
        if len(keep) >= max_out:
            break
        ok = True
        for j in reversed(keep):                   # most recent first, like TF
            if float(O.nms_iou(boxes[i], boxes[j][None])[0]) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_nms_toy_and_zero_area_rule():
    b = np.array([[0.0, 0.0, 0.5, 0.5],
                  [0.01, 0.01, 0.5, 0.5],     # IoU 0.96 with 0 -> suppressed
                  [0.5, 0.5, 1.0, 1.0],       # disjoint -> kept
                  [0.2, 0.2, 0.2, 0.9],       # zero area: never suppressed, never suppresses
                  [0.2, 0.2, 0.2, 0.9],
                  [0.0, 0.0, 0.5, 0.36]], F32)  # IoU 0.72 with 0 -> suppressed (strict >0.7)
    assert O.nms_keep(b, 0.7, 10).tolist() == [0, 2, 3, 4]
    # exactly at the threshold is NOT suppressed (strict comparison)
    c = np.array([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 0.5]], F32)   # IoU exactly 0.5
    assert O.nms_keep(c, 0.5, 10).tolist() == [0, 1]
    assert O.nms_keep(b, 0.7, 2).tolist() == [0, 2]                   # max_out stops early


def test_nms_matches_naive_greedy_and_pads():
    d = O.synth_inputs(5, 1, 16, 16, 9, 4, 8, with_fmap=False)
    a = O.make_anchor(16, 16)
    pc = O.proposal_creator(d["score"], d["coord"], a, pre_k=600, post_n=100)
    cand = pc["cand"][0]
    assert _naive_nms(cand, np.float32(0.7), 100) == pc["keep_pos"][0][:pc["counts"][0]].tolist()
    n = pc["counts"][0]
    assert (pc["proposed"][0, n:] == 0).all() and (pc["keep_pos"][0, n:] == -1).all()


def test_nms_equal_scores_earlier_position_first():
    # two identical boxes with equal scores: the earlier candidate wins (build-defined rule)
    b = np.array([[0.1, 0.1, 0.6, 0.6], [0.1, 0.1, 0.6, 0.6], [0.1, 0.1, 0.6, 0.6]], F32)
    assert O.nms_keep(b, 0.7, 10).tolist() == [0]


# ------------------------------------------------------ crop + pool (a7, a8)
def test_crop_identity_and_pool():
    rng = np.random.default_rng(0)
    f = rng.standard_normal((1, 14, 14, 3)).astype(F32)
    crop = O.crop_and_resize(f, np.array([[0, 0, 1, 1]], F32), np.array([0]), (14, 14))
    np.testing.assert_array_equal(crop[0], f[0])
    pooled, arg = O.max_pool(crop, 2)
    tp, ti = F.max_pool2d(torch.from_numpy(f).permute(0, 3, 1, 2), 2, return_indices=True)
    np.testing.assert_array_equal(pooled[0], tp[0].permute(1, 2, 0).numpy())
    # winner index agrees with torch's flat index where the maximum is unique
    ty, tx = (ti[0] // 14).permute(1, 2, 0).numpy(), (ti[0] % 14).permute(1, 2, 0).numpy()
    np.testing.assert_array_equal(arg[0], (ty % 2) * 2 + (tx % 2))


def test_crop_matches_grid_sample_in_range():
    rng = np.random.default_rng(1)
    H, W, C = 9, 13, 5
    f = rng.standard_normal((1, H, W, C)).astype(F32)
    y1, x1 = rng.uniform(0, 0.5, 20), rng.uniform(0, 0.5, 20)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.1, 0.5, 20), x1 + rng.uniform(0.1, 0.5, 20)],
                     1).astype(F32)
    got = O.crop_and_resize(f, boxes, np.zeros(20, np.int64), (14, 14))
    g = O._crop_geometry(boxes, H, W, 14, 14)
    iy = (boxes[:, 0] * (H - 1))[:, None] + np.arange(14)[None] * ((boxes[:, 2] - boxes[:, 0]) * (H - 1) / 13)[:, None]
    ix = (boxes[:, 1] * (W - 1))[:, None] + np.arange(14)[None] * ((boxes[:, 3] - boxes[:, 1]) * (W - 1) / 13)[:, None]
    gy = 2 * iy / (H - 1) - 1
    gx = 2 * ix / (W - 1) - 1
    grid = np.stack(np.broadcast_arrays(gx[:, None, :], gy[:, :, None]), -1).astype(F32)
    ft = torch.from_numpy(f).permute(0, 3, 1, 2).expand(20, -1, -1, -1)
    ref = F.grid_sample(ft, torch.from_numpy(grid), mode="bilinear", align_corners=True)
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), atol=2e-5)
    assert g["y_ok"].all() and g["x_ok"].all()


def test_crop_extrapolation_rows_are_zero():
    f = np.ones((1, 5, 5, 2), F32)
    crop = O.crop_and_resize(f, np.array([[-0.5, 0.0, 0.5, 1.0]], F32), np.array([0]), (4, 4))
    assert (crop[0, 0] == 0).all() and (crop[0, 1] == 0).all()      # in_y = -2, -2/3
    assert (crop[0, 3] == 1).all()
    z = O.crop_and_resize(f * 3, np.zeros((1, 4), F32), np.array([0]), (14, 14))
    assert (z == 3).all()                                           # padded proposal samples cell (0,0)


def test_roi_backward_is_adjoint():
    rng = np.random.default_rng(2)
    H, W, C, R = 6, 7, 3, 5
    f = rng.standard_normal((1, H, W, C)).astype(F32)
    y1, x1 = rng.uniform(0, 0.5, R), rng.uniform(0, 0.5, R)
    prop = np.stack([y1, x1, y1 + rng.uniform(0.1, 0.5, R), x1 + rng.uniform(0.1, 0.5, R)],
                    1).astype(F32)[None]
    out, win = O.roi_crop_pool(f, prop, (4, 4), 2)
    dY = rng.standard_normal(out.shape).astype(F32)
    dF = O.roi_crop_pool_bwd(dY, win, f.shape, prop, (4, 4), 2)
    # linear in f for a fixed winner pattern: <dY, out(f)> == <dF, f> up to rounding, when the
    # winners of out(f) are used
    np.testing.assert_allclose((dY * out).sum(), (dF * f).sum(), rtol=1e-4)


# --------------------------------------------- proposal labels / sampling (a9-a11)
def test_label_proposals_quirks():
    gt = np.zeros((1, 42, 4), F32)
    gt[0, 0] = [0.1, 0.1, 0.5, 0.5]
    gt[0, 1] = [0.5, 0.5, 0.9, 0.9]
    label = np.zeros((1, 42), np.int64)
    label[0, :2] = [7, 3]
    prop = np.array([[[0.1, 0.1, 0.5, 0.5],        # IoU 1 with gt0 -> class 7
                      [0.52, 0.5, 0.9, 0.9],       # IoU .95 with gt1 -> class 3
                      [0.0, 0.0, 0.0, 0.0],        # zero pad proposal
                      [0.1, 0.1, 0.3, 0.5]]], F32)  # IoU .5 with gt0 -> fg (>= 0.5)
    lab = O.label_proposals(prop, gt, label)
    assert lab["p_iou"][0].tolist()[:2] == [7, 3]
    assert lab["p_iou"][0, 3] == 7
    assert lab["gt_mask"][0].tolist()[:2] == [1, 1]
    # zero proposal vs zero pad rows: 1e-5/(0+0-1e-5) = -1 -> floored to 0 -> background
    assert lab["p_iou"][0, 2] == 20
    assert lab["gt_label"].shape == (1, 4, 21) and lab["gt_label"][0, 0, 7] == 1


def test_sample_proposals_counts():
    p = np.full((2, 300), 20, np.int64)
    p[0, :50] = 1
    p[1, :5] = 2
    t = O.sample_proposals(p, seed=1)
    assert t.shape == (2, 128)
    assert (p[0][t[0]] != 20).sum() == 32 and (p[1][t[1]] != 20).sum() == 5
    assert len(set(t[0].tolist())) == 128
    t2 = O.sample_proposals(p, seed=1)
    np.testing.assert_array_equal(t, t2)
    assert not np.array_equal(t, O.sample_proposals(p, seed=2))


# ------------------------------------------------ detection post-process (f1)
def test_detect_postprocess_hand_case():
    prop = np.array([[[0.1, 0.1, 0.5, 0.5],       # class 3, score .95 -> kept
                      [0.1, 0.1, 0.5, 0.52],      # class 3, score .93, IoU .95 with #0 -> suppressed
                      [0.6, 0.6, 0.9, 0.9],       # class 7, score .91 -> kept
                      [0.0, 0.6, 0.3, 0.9],       # background -> dropped
                      [0.3, 0.0, 0.6, 0.3]]], F32)  # class 1, score .5 -> kept by NMS, cut by score
    cls = np.full((1, 5, 21), 0.0, F32)
    cls[0, 0, 3], cls[0, 1, 3], cls[0, 2, 7], cls[0, 3, 20], cls[0, 4, 1] = .95, .93, .91, .99, .5
    reg = np.zeros((1, 5, 84), F32)               # zero offsets: decoded box = proposal
    d = O.detect_postprocess(prop, cls, reg)
    assert d["count"].tolist() == [2]
    assert d["src"][0, :2].tolist() == [0, 2] and d["classes"][0, :2].tolist() == [3, 7]
    np.testing.assert_allclose(d["boxes"][0, :2], prop[0, [0, 2]], atol=1e-7)
    assert (d["src"][0, 2:] == -1).all()
    # equal scores: lower index first
    sel, _ = O.nms_with_scores(np.array([[0, 0, .1, .1], [.5, .5, .6, .6]], F32),
                               np.array([.7, .7], F32), 5)
    assert sel.tolist() == [0, 1]
