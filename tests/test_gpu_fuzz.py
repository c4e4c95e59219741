"""Randomised differential tests: each entry point against the CPU oracle on drawn shapes
and inputs.  hypothesis runs derandomised (every run draws the same cases) with a small
example budget; a failing case is shrunk and printed with its arguments."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, example, given, settings, strategies as st

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu

FUZZ = settings(derandomize=True, deadline=None, max_examples=60, database=None,
                suppress_health_check=list(HealthCheck))
DEV = torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def draw_boxes(rng, n, mode):
    """[n,4] float32 [y1,x1,y2,x2] of one of several families."""
    if mode == "uniform":
        c = rng.uniform(0, 1, (n, 2)); h = rng.uniform(0.01, 0.7, (n, 2))
        b = np.concatenate([c - h / 2, c + h / 2], 1)
        return np.clip(b, 0, 1).astype(np.float32)
    if mode == "clustered":
        k = max(1, n // 40)
        c = rng.uniform(0.2, 0.8, (k, 2))[rng.integers(0, k, n)] + rng.normal(0, 0.01, (n, 2))
        h = 0.2 + rng.normal(0, 0.01, (n, 2))
        return np.concatenate([c - h / 2, c + h / 2], 1).astype(np.float32)
    if mode == "outside":                                   # partly or wholly outside [0,1]
        c = rng.uniform(-0.3, 1.3, (n, 2)); h = rng.uniform(0.0, 0.8, (n, 2))
        return np.concatenate([c - h / 2, c + h / 2], 1).astype(np.float32)
    if mode == "degenerate":                                # zero area, repeated, flipped
        b = draw_boxes(rng, n, "uniform")
        j = rng.integers(0, 4, n)
        b[j == 0, 2] = b[j == 0, 0]                         # zero height
        b[j == 1] = b[0]                                    # copies of box 0
        f = j == 2
        b[f] = b[f][:, [2, 1, 0, 3]]                        # y flipped
        return b
    if mode == "grid":                                      # corners exactly on cell centres
        q = rng.integers(0, 9, (n, 4)).astype(np.float32) / np.float32(8)
        return np.stack([np.minimum(q[:, 0], q[:, 2]), np.minimum(q[:, 1], q[:, 3]),
                         np.maximum(q[:, 0], q[:, 2]), np.maximum(q[:, 1], q[:, 3])], 1)
    raise ValueError(mode)


BOX_MODES = st.sampled_from(["uniform", "clustered", "outside", "degenerate", "grid"])


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), H=st.integers(1, 12), W=st.integers(1, 12),
       ns=st.integers(1, 4), nr=st.integers(1, 3), G=st.integers(1, 70),
       thr=st.sampled_from([0.1, 0.3, 0.5]), mode=BOX_MODES)
def test_fuzz_anchor_assignment(seed, H, W, ns, nr, G, thr, mode):
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    scales = tuple(float(s) for s in rng.choice([16, 32, 64, 128, 256, 512], ns, replace=False))
    ratios = tuple(float(r) for r in rng.choice([0.5, 1.0, 2.0], nr, replace=False))
    base_w, base_h = float(rng.choice([500, 1000, 768])), float(rng.choice([500, 600, 768]))
    a = O.make_anchor(H, W, scales, ratios, base_w, base_h)
    da = ops.anchors(H, W, scales, ratios, base_w, base_h, DEV)
    np.testing.assert_array_equal(da.cpu().numpy(), a)
    n_real = int(rng.integers(0, G + 1))
    gt = np.zeros((2, G, 4), np.float32)
    gt[0, :n_real] = draw_boxes(rng, n_real, mode) if n_real else 0
    gt[1] = draw_boxes(rng, G, "uniform")
    mi, best, flags, full = ops.iou_assign(da.reshape(-1, 4), T(gt), pos_thr=thr, neg_thr=thr,
                                           want_iou=True)
    for b in range(2):
        ref = O.rpn_assign(a, gt[b], thr, thr)
        np.testing.assert_array_equal(full[b].cpu().numpy(), ref["iou"])
        np.testing.assert_array_equal(best[b].cpu().numpy(), ref["best"])
        np.testing.assert_array_equal(mi[b].cpu().numpy(), ref["max_iou"])
        f = flags[b].cpu().numpy()
        np.testing.assert_array_equal((f & 1) != 0, ref["valid"])
        np.testing.assert_array_equal(np.nonzero(f & 2)[0], ref["pos_anchor"])
        valid_ind = np.nonzero(ref["valid"])[0]
        np.testing.assert_array_equal(np.nonzero(f & 4)[0], valid_ind[ref["neg_pos"]])


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 3), N=st.integers(1, 40000),
       kfrac=st.floats(0.0, 1.0), vals=st.sampled_from(["normal", "few", "zeros", "equal", "big"]),
       method=st.sampled_from(["lds", "sort"]))
@example(seed=1, B=2, N=39999, kfrac=0.5, vals="few", method="sort")
@example(seed=2, B=3, N=20000, kfrac=1.0, vals="zeros", method="lds")
@example(seed=3, B=1, N=40000, kfrac=1.0, vals="equal", method="sort")
@example(seed=12, B=11, N=2500, kfrac=0.3, vals="few", method="sort")     # 8 images XCD-ordered + 3 in order
@example(seed=13, B=16, N=1025, kfrac=1.0, vals="normal", method="sort")
def test_fuzz_topk(seed, B, N, kfrac, vals, method):
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    if vals == "normal":
        s = rng.standard_normal((B, N))
    elif vals == "few":
        s = rng.choice([0.25, 0.5, 1.0, -1.0], (B, N))
    elif vals == "zeros":
        s = rng.choice([0.0, -0.0, 1e-30, -1e-30], (B, N))
    elif vals == "equal":
        s = np.full((B, N), 0.75)
    else:
        s = rng.standard_normal((B, N)) * 1e30
    s = s.astype(np.float32)
    kmax = min(N, 16384) if method == "lds" else N
    k = max(1, min(kmax, int(round(kfrac * kmax))))
    v, i, _ = ops.topk(T(s), k, method=method)
    rv, ri = O.top_k(s, k)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy().view(np.uint32), rv.view(np.uint32))


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 2), K=st.integers(1, 3000),
       post_n=st.integers(1, 400), thr=st.sampled_from([0.3, 0.5, 0.7, 0.9]), mode=BOX_MODES)
@example(seed=4, B=2, K=3000, post_n=400, thr=0.7, mode="clustered")
@example(seed=5, B=1, K=2500, post_n=64, thr=0.3, mode="degenerate")
@example(seed=6, B=2, K=1025, post_n=400, thr=0.9, mode="grid")
def test_fuzz_nms(seed, B, K, post_n, thr, mode):
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    boxes = np.stack([np.clip(draw_boxes(rng, K, mode), 0, 1) for _ in range(B)])
    prop, keep, cnt = ops.nms(T(boxes), thr, post_n)
    rp, rk, rc = O.combined_nms(boxes, None, post_n, thr)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    np.testing.assert_array_equal(prop.cpu().numpy(), rp)


@pytest.mark.parametrize("gk", [1, 2, 3, 5])
@pytest.mark.parametrize("mode", ["uniform", "clustered", "degenerate"])
def test_nms_kept_tiles_shared_with_a_stride(gk, mode, tuning):
    """With many images a mask workgroup takes several kept tiles of a column block in turn
    (stride gk); forced here on a small batch, with enough kept boxes for several tiles and
    candidates beyond the second chunk."""
    from objectdetection_amd import ops
    tuning("nms_gk", gk)
    rng = np.random.default_rng(100 + gk)
    B, K, post_n, thr = 3, 2600, 700, 0.7
    boxes = np.stack([np.clip(draw_boxes(rng, K, mode), 0, 1) for _ in range(B)])
    prop, keep, cnt = ops.nms(T(boxes), thr, post_n)
    rp, rk, rc = O.combined_nms(boxes, None, post_n, thr)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    np.testing.assert_array_equal(prop.cpu().numpy(), rp)


CROPS = st.sampled_from([((14, 14), 2), ((14, 14), 1), ((7, 7), 1), ((6, 10), 2), ((2, 2), 2),
                         ((1, 1), 1), ((4, 4), 2), ((16, 16), 2), ((3, 5), 1)])


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 2), H=st.integers(1, 20),
       W=st.integers(1, 20), C=st.sampled_from([4, 8, 12, 64, 68, 128, 192, 256, 6, 1]),
       R=st.integers(1, 10), cp=CROPS, mode=BOX_MODES, strided=st.booleans())
@example(seed=7, B=2, H=20, W=20, C=192, R=10, cp=((14, 14), 2), mode="outside", strided=True)
@example(seed=8, B=1, H=17, W=19, C=256, R=10, cp=((16, 16), 2), mode="grid", strided=False)
@example(seed=9, B=2, H=20, W=3, C=68, R=9, cp=((6, 10), 2), mode="degenerate", strided=True)
def test_fuzz_roi_forward_backward(seed, B, H, W, C, R, cp, mode, strided):
    from objectdetection_amd import ops
    crop, pool = cp
    rng = np.random.default_rng(seed)
    fm = rng.standard_normal((B, H, W, C)).astype(np.float32)
    prop = np.stack([draw_boxes(rng, R, mode) for _ in range(B)])
    f = T(fm)
    if strided:                                            # a channel window of a wider map
        wide = torch.zeros((B, H, W, C + 8), dtype=torch.float32, device=DEV)
        wide[..., 4:4 + C] = f
        f = wide[..., 4:4 + C]
    out, win = ops.roi_crop_pool_fwd(f, T(prop), crop, pool, want_winner=pool > 1)
    ro, ra = O.roi_crop_pool(fm, prop, crop, pool)
    np.testing.assert_array_equal(out.cpu().numpy(), ro)
    if pool > 1:
        np.testing.assert_array_equal(win.cpu().numpy(), ra)
    if C % 4 == 0:
        dY = rng.standard_normal(ro.shape).astype(np.float32)
        got = ops.roi_crop_pool_bwd(T(dY), win, T(prop), fm.shape, crop, pool)
        ref = O.roi_crop_pool_bwd(dY, ra, fm.shape, prop, crop, pool)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 2), H=st.integers(1, 90),
       W=st.integers(1, 40), C=st.sampled_from([32, 64, 96, 160]), R=st.integers(1, 40), cp=CROPS,
       mode=BOX_MODES, prepared=st.booleans())
@example(seed=21, B=2, H=90, W=40, C=64, R=40, cp=((14, 14), 2), mode="outside", prepared=True)
@example(seed=22, B=1, H=1, W=1, C=32, R=7, cp=((14, 14), 2), mode="uniform", prepared=False)
@example(seed=23, B=1, H=80, W=1, C=96, R=33, cp=((16, 16), 2), mode="grid", prepared=True)
@example(seed=24, B=2, H=2, W=37, C=160, R=5, cp=((1, 1), 1), mode="degenerate", prepared=False)
def test_fuzz_roi_forward_band_kernel(seed, B, H, W, C, R, cp, mode, prepared):
    """The band-resident forward kernel (forced on at these sizes) on maps of one to many bands,
    in one call and as prepare + run; the pool = 1 winner case falls back to the per-RoI kernel."""
    import os
    from objectdetection_amd import ops
    crop, pool = cp
    rng = np.random.default_rng(seed)
    fm = rng.standard_normal((B, H, W, C)).astype(np.float32)
    prop = np.stack([draw_boxes(rng, R, mode) for _ in range(B)])
    ws = torch.empty(ops.roi_fwd_workspace_bytes(B, R), dtype=torch.uint8, device=DEV)
    from objectdetection_amd import _lib
    old = _lib.get_tuning("roi_band_min_wg")
    _lib.set_tuning("roi_band_min_wg", 1)
    try:
        ready = prepared and ops.roi_fwd_prepare(T(prop), fm.shape, crop, pool, pool > 1, ws)
        out, win = ops.roi_crop_pool_fwd(T(fm), T(prop), crop, pool, want_winner=pool > 1,
                                         workspace=ws, prepared=ready)
    finally:
        _lib.set_tuning("roi_band_min_wg", old)
    ro, ra = O.roi_crop_pool(fm, prop, crop, pool)
    np.testing.assert_array_equal(out.cpu().numpy(), ro)
    if pool > 1:
        np.testing.assert_array_equal(win.cpu().numpy(), ra)


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 3), R=st.integers(1, 600),
       G=st.integers(1, 60), batch=st.sampled_from([8, 64, 128]), mode=BOX_MODES)
@example(seed=10, B=3, R=600, G=60, batch=128, mode="clustered")
@example(seed=11, B=2, R=599, G=1, batch=64, mode="uniform")
def test_fuzz_proposal_targets(seed, B, R, G, batch, mode):
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    prop = np.stack([np.clip(draw_boxes(rng, R, mode), 0, 1) for _ in range(B)])
    gt = np.zeros((B, G, 4), np.float32)
    label = np.zeros((B, G), np.int64)
    for b in range(B):
        n = int(rng.integers(0, G + 1))
        src = prop[b, rng.integers(0, R, n)] if n else np.zeros((0, 4), np.float32)
        gt[b, :n] = np.clip(src + rng.normal(0, 0.02, (n, 4)).astype(np.float32), 0, 1)
        label[b, :n] = rng.integers(0, 20, n)
    lab = O.label_proposals(prop, gt, label)
    dl = ops.label_proposals(T(prop), T(gt), T(label))
    np.testing.assert_array_equal(dl["p_label"].cpu().numpy(), lab["p_iou"])
    np.testing.assert_array_equal(dl["gt_mask"].cpu().numpy(), lab["gt_mask"])
    np.testing.assert_array_equal(dl["best_gt"].cpu().numpy(), lab["best"])
    np.testing.assert_array_equal(dl["max_iou"].cpu().numpy(), lab["max_iou"])
    np.testing.assert_array_equal(dl["gt_onehot"].cpu().numpy(), lab["gt_label"])
    max_pos = batch // 4
    tindex, counts = ops.sample_proposals(dl["p_label"], seed=seed, image_base=3, batch=batch,
                                          max_pos=max_pos)
    counts = counts.cpu().numpy()
    for b in range(B):
        npos = int((lab["p_iou"][b] != 20).sum())
        short = (R - npos) < batch - min(npos, max_pos)
        assert (counts[b, 3] != 0) == short
        if not short:
            ref = O.sample_proposals(lab["p_iou"][b:b + 1], seed=seed, image_base=3 + b,
                                     batch=batch, max_pos=max_pos)
            np.testing.assert_array_equal(tindex[b].cpu().numpy(), ref[0])


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), H=st.integers(2, 14), W=st.integers(2, 14),
       G=st.integers(1, 30), batch=st.sampled_from([16, 64, 256]))
def test_fuzz_rpn_sampling(seed, H, W, G, batch):
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    scales, ratios = (64.0, 128.0, 256.0), (0.5, 1.0, 2.0)
    a = O.make_anchor(H, W, scales, ratios, 500.0, 500.0)
    da = ops.anchors(H, W, scales, ratios, 500.0, 500.0, DEV)
    gt = np.zeros((1, G, 4), np.float32)
    n = int(rng.integers(1, G + 1))
    gt[0, :n] = draw_boxes(rng, n, "uniform")
    label = np.zeros((G,), np.int64)
    _, best, flags, _ = ops.iou_assign(da.reshape(-1, 4), T(gt))
    out = ops.rpn_sample(da, T(gt), best, flags, seed=seed, image_base=5, batch=batch,
                         max_pos=batch // 2)
    status = int(out["counts"][0, 3])
    try:
        ref = O.patch_batch(dict(image=None, bbox=gt[0], label=label), a, seed=seed, image_index=5,
                            batch=batch, max_pos=batch // 2)
    except O.ShortBatchError:
        assert status != 0
        return
    assert status == 0
    np.testing.assert_array_equal(out["label"][0].cpu().numpy(), ref[1][:, 0])
    np.testing.assert_array_equal(out["anchor"][0].cpu().numpy(), ref[2])
    np.testing.assert_array_equal(out["gt"][0].cpu().numpy(), ref[3])
    np.testing.assert_array_equal(out["gt_list"][0].cpu().numpy(), ref[4])
    np.testing.assert_array_equal(out["pos"][0].cpu().numpy(), ref[5])


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 3), n=st.integers(1, 5000),
       clip=st.booleans(), scale=st.sampled_from([0.1, 1.0, 5.0]))
def test_fuzz_decode_encode(seed, B, n, clip, scale):
    """Decoded boxes within 1e-5 relative of the oracle (device expf/logf differ by ULPs)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    anc = draw_boxes(rng, n, "uniform")
    t = (rng.standard_normal((B, n, 4)) * scale).astype(np.float32)
    got = ops.decode_clip(T(anc), T(t), clip).cpu().numpy()
    ref = O._decode(anc[None], t, clip_wh=clip)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
    g = draw_boxes(rng, n, "uniform")
    enc = ops.encode(T(anc), T(g), clip).cpu().numpy()
    renc = O.encode_targets(anc, g, clip_wh=clip)
    fin = np.isfinite(renc)
    np.testing.assert_array_equal(np.isfinite(enc), fin)
    np.testing.assert_allclose(enc[fin], renc[fin], rtol=1e-5, atol=1e-5)


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 3), R=st.integers(1, 700),
       NC=st.sampled_from([2, 5, 21]), max_out=st.sampled_from([1, 10, 30, 100]),
       sthr=st.sampled_from([0.0, 0.5, 0.9]), mode=BOX_MODES)
@example(seed=12, B=3, R=700, NC=21, max_out=30, sthr=0.5, mode="clustered")
def test_fuzz_detect_postprocess(seed, B, R, NC, max_out, sthr, mode):
    """Class argmax, decode, background drop, final NMS and score cut: keep lists and classes
    bit-exact on the device-decoded boxes, decoded boxes within 1e-5."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    prop = np.stack([np.clip(draw_boxes(rng, R, mode), 0, 1) for _ in range(B)])
    z = rng.normal(0, 3.0, (B, R, NC))
    p = (np.exp(z) / np.exp(z).sum(-1, keepdims=True)).astype(np.float32)
    if R > 4:
        p[0, 1:4] = p[0, 1]                                  # equal scores -> lower index first
    reg = rng.normal(0, 0.5, (B, R, NC * 4)).astype(np.float32)
    bg = NC - 1
    out = ops.detect_postprocess(T(prop), T(p), T(reg), bg_label=bg, max_out=max_out,
                                 score_threshold=sthr)
    ref = O.detect_postprocess(prop, p, reg, bg_label=bg, max_out=max_out, score_threshold=sthr)
    np.testing.assert_array_equal(out["cls_arg"].cpu().numpy(), ref["argindex"])
    dec = out["decoded"].cpu().numpy()
    np.testing.assert_allclose(dec, ref["decoded"], rtol=1e-5, atol=1e-7)
    score = np.take_along_axis(p, ref["argindex"].astype(np.int64)[..., None], 2)[..., 0]
    for b in range(B):
        fg = np.nonzero(ref["argindex"][b] != bg)[0]
        n = 0
        if fg.size:
            sel, ssc = O.nms_with_scores(dec[b][fg], score[b][fg], max_out, 0.7)
            ok = ssc >= np.float32(sthr)
            n = int(ok.sum())
            np.testing.assert_array_equal(out["src"][b].cpu().numpy()[:n], fg[sel[ok]])
            np.testing.assert_array_equal(out["scores"][b].cpu().numpy()[:n], ssc[ok])
        assert int(out["count"][b]) == n
        assert (out["src"][b].cpu().numpy()[n:] == -1).all()


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 4), H=st.integers(1, 9),
       W=st.integers(1, 9), A=st.integers(1, 12), S=st.sampled_from([1, 16, 256]),
       R=st.integers(1, 300), K=st.sampled_from([2, 7, 21]), pfrac=st.floats(0.0, 1.0))
def test_fuzz_losses(seed, B, H, W, A, S, R, K, pfrac):
    """RPN and head losses with their gradients, 1e-5 relative (f32 tree sums vs float64)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    pred_obj = rng.random((B, H, W, A), dtype=np.float32)
    pred_reg = (rng.normal(size=(B, H, W, A, 4)) * 2).astype(np.float32)
    S = min(S, H * W * A, R)                                  # samples are distinct anchors / proposals
    flat = np.stack([rng.permutation(H * W * A)[:S] for _ in range(B)])
    pos = np.stack(np.unravel_index(flat, (H, W, A)), -1).astype(np.int64)
    label = (rng.random((B, S)) < pfrac).astype(np.float32)
    reg_gt = (rng.normal(size=(B, S, 4)) * 2).astype(np.float32)
    (cls, reg, tot), d_obj, d_reg = O.rpn_loss(pred_obj, pred_reg, pos, label, reg_gt)
    out = ops.rpn_loss(T(pred_obj), T(pred_reg), T(pos), T(label), T(reg_gt))
    np.testing.assert_allclose(out["loss"].cpu().numpy(), [cls, reg, tot], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["d_obj"].cpu().numpy(), d_obj, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(out["d_reg"].cpu().numpy(), d_reg, rtol=2e-5, atol=1e-9)
    cls_prob = rng.random((B, R, K), dtype=np.float32) + np.float32(1e-3)
    cls_prob /= cls_prob.sum(-1, keepdims=True)
    hreg = (rng.normal(size=(B, R, K, 4)) * 2).astype(np.float32)
    tindex = np.stack([rng.permutation(R)[:S] for _ in range(B)]).astype(np.int64)
    c = rng.integers(0, K, (B, S))
    onehot = np.eye(K, dtype=np.float32)[c]
    gt_mask = (rng.random((B, S)) < pfrac).astype(np.int32)
    gt_coord = (rng.normal(size=(B, S, 4)) * 3).astype(np.float32)
    (hc, hr, ht), d_cls, d_hreg = O.head_loss(cls_prob, hreg, tindex, onehot, gt_mask, gt_coord)
    ho = ops.head_loss(T(cls_prob), T(hreg), T(tindex), T(onehot), T(gt_mask), T(gt_coord))
    np.testing.assert_allclose(ho["loss"].cpu().numpy(), [hc, hr, ht], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ho["d_cls"].cpu().numpy(), d_cls, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(ho["d_reg"].cpu().numpy(), d_hreg, rtol=2e-5, atol=1e-9)


@FUZZ
@given(seed=st.integers(0, 2**31 - 1), B=st.integers(1, 6), G=st.integers(1, 50),
       H=st.integers(1, 9), W=st.integers(1, 40), C=st.sampled_from([1, 3, 4]))
def test_fuzz_input_side(seed, B, G, H, W, C):
    """Ragged ground truth -> padded tensors with the flip of the boxes, and the image flip,
    bit-exact, with explicit flags and with the counter-based decision."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, G + 1, B)
    boxes = [draw_boxes(rng, int(n), "uniform") if n else np.zeros((0, 4), np.float32)
             for n in counts]
    labels = [rng.integers(0, 20, int(n)).astype(np.int64) for n in counts]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    cat = np.concatenate(boxes) if off[-1] else np.zeros((0, 4), np.float32)
    lab = np.concatenate(labels) if off[-1] else np.zeros((0,), np.int64)
    img = rng.random((B, H, W, C), dtype=np.float32)
    for flags in (rng.integers(0, 2, B).astype(np.uint8), None):
        fl = None if flags is None else T(flags)
        gt, gl, cnt = ops.pad_gt(T(cat), T(lab), T(off), G, flip=fl, use_flip=True, seed=seed,
                                 image_base=2)
        fimg = ops.flip_images(T(img), fl, seed=seed, image_base=2).cpu().numpy()
        np.testing.assert_array_equal(cnt.cpu().numpy(), counts)
        for b in range(B):
            f = bool(flags[b]) if flags is not None else O.flip_decision(seed, 2 + b)
            bb = boxes[b]
            if f and bb.shape[0]:
                _, _, bb = O.augmentation_data(img[b], labels[b], bb)
            rb, rl = O.pad_gt(bb, labels[b], G)
            np.testing.assert_array_equal(gt[b].cpu().numpy(), rb)
            np.testing.assert_array_equal(gl[b].cpu().numpy(), rl)
            np.testing.assert_array_equal(fimg[b], img[b][:, ::-1] if f else img[b])
