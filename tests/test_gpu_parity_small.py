"""HIP path vs oracle on seeded inputs at the reference's own sizes
(31x31x9 anchors, G=42, 6000 -> 1500 proposals, 14 -> 7 RoI)."""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu


def T(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


@pytest.fixture(scope="module")
def ref_inputs():
    d = O.synth_inputs(seed=0, B=2, H=31, W=31, A=9, C=64, G=42)
    d["anchor"] = O.make_anchor()
    return d


def test_anchors_bit_exact(dev):
    from objectdetection_amd.module import make_anchor
    from objectdetection_amd.config import SHIP
    a = make_anchor(device=dev).cpu().numpy()
    assert a.shape == (31, 31, 9, 4)
    np.testing.assert_array_equal(a, O.make_anchor())
    s = make_anchor(SHIP, device=dev).cpu().numpy()
    np.testing.assert_array_equal(s, O.make_anchor(32, 32, (32, 64, 128, 256), (0.5, 1, 2)))


def test_iou_assign_bit_exact(dev, ref_inputs):
    from objectdetection_amd import ops
    a = ref_inputs["anchor"]
    gt = ref_inputs["gt"]
    mi, best, flags, full = ops.iou_assign(T(a, dev).reshape(-1, 4), T(gt, dev), want_iou=True)
    for b in range(gt.shape[0]):
        st = O.rpn_assign(a, gt[b])
        np.testing.assert_array_equal(full[b].cpu().numpy(), st["iou"])
        np.testing.assert_array_equal(best[b].cpu().numpy(), st["best"])
        np.testing.assert_array_equal(mi[b].cpu().numpy(), st["max_iou"])
        f = flags[b].cpu().numpy()
        np.testing.assert_array_equal((f & 1) != 0, st["valid"])
        np.testing.assert_array_equal(np.nonzero(f & 2)[0], st["pos_anchor"])


@pytest.mark.parametrize("mode", ["keys", "philox"])
def test_patch_batch_bit_exact(dev, ref_inputs, mode):
    from objectdetection_amd.module import patch_batch
    a = ref_inputs["anchor"]
    N = a.shape[0] * a.shape[1] * a.shape[2]
    rng = np.random.default_rng(5)
    for b in range(2):
        gt = ref_inputs["gt"][b]
        lab = ref_inputs["label"][b]
        if mode == "keys":
            kp = rng.permutation(N).astype(np.uint32)
            kn = rng.permutation(N).astype(np.uint32)
            kt = rng.permutation(256).astype(np.uint32)
            okeys = (kp, kn, kt)
            dkeys = tuple(T(k.view(np.int32), dev) for k in okeys)
        else:
            okeys = dkeys = None
        ref = O.patch_batch(dict(image=None, bbox=gt, label=lab), a, keys=okeys, seed=1234,
                            image_index=b)
        out = patch_batch(dict(image=None, bbox=T(gt, dev), label=T(lab, dev)), T(a, dev),
                          keys=dkeys, seed=1234, image_index=b)
        names = ["image", "batch_label", "batch_anchor", "batch_gt", "gt_list", "batch_pos",
                 "batch_reg_gt"]
        for i in (1, 2, 3, 4, 5):
            np.testing.assert_array_equal(out[i].cpu().numpy(), ref[i], err_msg=names[i])
        # encode uses log: positives within 1e-5 relative; negatives are -inf/nan by design
        pos = ref[1][:, 0] == 1
        np.testing.assert_allclose(out[6].cpu().numpy()[pos], ref[6][pos], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(out[6].cpu().numpy()[:, :2], ref[6][:, :2])


def test_decode_close(dev, ref_inputs):
    from objectdetection_amd.module import inverse_trans, generate_coord
    a = ref_inputs["anchor"]
    got = inverse_trans(T(a, dev), T(ref_inputs["coord"], dev)).cpu().numpy()
    ref = O.inverse_trans(a, ref_inputs["coord"])
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
    prop = ref.reshape(2, -1, 4)[:, :1500]
    t = ref_inputs["coord"].reshape(2, -1, 4)[:, :1500]
    got2 = generate_coord(T(prop, dev), T(t, dev)).cpu().numpy()
    np.testing.assert_allclose(got2, O.generate_coord(prop, t), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("method", ["lds", "sort"])
def test_topk_bit_exact_with_ties(dev, ref_inputs, method):
    """Both top-k entry points: one workgroup per image in LDS, and the radix sort through
    global memory."""
    from objectdetection_amd import ops
    s = ref_inputs["score"].reshape(2, -1).copy()
    s[0, 100:400] = 1.0            # saturated sigmoid ties
    s[1, ::7] = s[1, 3]
    s[1, 5] = -0.0
    s[1, 6] = 0.0
    for k in (1, 257, 6000, s.shape[1]):
        v, i, _ = ops.topk(T(s, dev), k, method=method)
        rv, ri = O.top_k(s, k)
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_topk_sort_large_and_negative(dev):
    """Radix-sort top-k beyond the LDS kernel's K limit, scores of both signs, N not a
    multiple of the block size, fused decode equal to the LDS kernel's."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(4)
    B, N = 3, 63001
    s = rng.standard_normal((B, N)).astype(np.float32)
    s[0, :5000] = np.float32(0.25)
    s[2] = np.abs(s[2])
    for k in (50000, N):
        v, i, _ = ops.topk(T(s, dev), k, method="sort")
        rv, ri = O.top_k(s, k)
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(v.cpu().numpy(), rv)
    anc = np.sort(rng.random((N, 2, 2)).astype(np.float32), axis=1).transpose(0, 2, 1).reshape(N, 4)
    anc = np.ascontiguousarray(anc[:, [0, 2, 1, 3]])
    dl = (rng.standard_normal((B, N, 4)) * 0.3).astype(np.float32)
    a = ops.topk(T(s, dev), 12000, T(anc, dev), T(dl, dev), method="lds")
    b = ops.topk(T(s, dev), 12000, T(anc, dev), T(dl, dev), method="sort")
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,N,K", [(1, 20646, 12000), (2, 1000, 1000), (3, 63001, 50000), (5, 4097, 17),
                                   (1, 64 * 1024, 64 * 1024), (4, 1025, 1025), (1, 1, 1), (3, 1024, 1024),
                                   (2, 1023, 5), (1, 2049, 2049), (7, 9, 9)])
def test_topk_sort_routes_agree(dev, B, N, K):
    """frcnn_topk_sort picks between the radix passes and the two-launch sorted-runs form by the
    size of the call: both against the oracle, and against each other with the fused decode, on
    scores with long runs of ties, both zeros and both signs."""
    from objectdetection_amd import _lib, ops
    rng = np.random.default_rng(B * 1000 + N)
    s = rng.standard_normal((B, N)).astype(np.float32)
    s[0, : N // 3] = np.float32(0.5)
    if N >= 4:
        s[-1, ::5] = s[-1, 1]
        s[-1, 2], s[-1, 3] = np.float32(-0.0), np.float32(0.0)
    anc = np.sort(rng.random((N, 2, 2)).astype(np.float32), axis=1).transpose(0, 2, 1).reshape(N, 4)
    anc = np.ascontiguousarray(anc[:, [0, 2, 1, 3]])
    dl = (rng.standard_normal((B, N, 4)) * 0.3).astype(np.float32)
    rv, ri = O.top_k(s, K)
    got = {}
    try:
        for route, runs in (("radix", 0), ("runs", 1 << 30)):
            _lib.set_tuning("topk_merge_max_runs", runs)
            # the route that is asked for is the one that runs (up to 64 blocks per image)
            assert ops.topk_sort_route(B, N) == ("radix passes" if route == "radix" else "sorted runs")
            got[route] = ops.topk(T(s, dev), K, T(anc, dev), T(dl, dev), method="sort")
            np.testing.assert_array_equal(got[route][1].cpu().numpy(), ri, err_msg=route)
            np.testing.assert_array_equal(got[route][0].cpu().numpy(), rv, err_msg=route)
    finally:
        _lib.set_tuning("topk_merge_max_runs", None)
    for x, y in zip(got["radix"], got["runs"]):
        assert torch.equal(x, y)


def test_proposals_bit_exact_given_boxes(dev, ref_inputs):
    """NMS keep list on identical box inputs (device-decoded boxes fed to the oracle)."""
    from objectdetection_amd.module.model import propose
    a = ref_inputs["anchor"]
    pr = propose(T(ref_inputs["score"], dev), T(ref_inputs["coord"], dev), T(a, dev))
    cand = pr["cand"].cpu().numpy()
    rv, ri = O.top_k(ref_inputs["score"].reshape(2, -1), 6000)
    np.testing.assert_array_equal(pr["top_idx"].cpu().numpy(), ri)
    ref_cand = np.take_along_axis(O.inverse_trans(a, ref_inputs["coord"]).reshape(2, -1, 4),
                                  ri.astype(np.int64)[..., None], axis=1)
    np.testing.assert_allclose(cand, ref_cand, rtol=1e-5, atol=1e-7)
    proposed, keep, counts = O.combined_nms(cand, None, 1500, 0.7)
    np.testing.assert_array_equal(pr["counts"].cpu().numpy(), counts)
    np.testing.assert_array_equal(pr["keep_pos"].cpu().numpy(), keep)
    np.testing.assert_array_equal(pr["proposed"].cpu().numpy(), proposed)


@pytest.mark.parametrize("C", [64, 32, 192, 6])
def test_roi_crop_pool_exact(dev, ref_inputs, C):
    """C=64/192: LDS-staged kernel (1 and 3 channel slices); C=32: direct
    channels-last kernel; C=6: any-stride kernel."""
    from objectdetection_amd import ops
    a = ref_inputs["anchor"]
    pc = O.proposal_creator(ref_inputs["score"], ref_inputs["coord"], a)
    prop = pc["proposed"][:, :300].copy()
    prop[0, 0] = [0, 0, 1, 1]
    prop[0, 1] = [0, 0, 0, 0]
    prop[0, 2] = [0.25, 0.25, 0.25, 0.75]
    prop[0, 3] = [0.5, 0.5, 1.0, 1.0]          # edge samples land exactly on H-1 / W-1
    prop[0, 4] = [0.9, 0.1, 0.2, 0.8]          # flipped box (y2 < y1)
    prop[0, 5] = [0.0, 0.0, 1.0 + 1e-6, 1.0]   # last row falls outside -> zeros
    fm = np.tile(ref_inputs["fmap"], (1, 1, 1, 3))[..., :C]
    fm = np.ascontiguousarray(fm * np.linspace(0.5, 1.5, C, dtype=np.float32))
    out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (14, 14), 2, want_winner=True)
    ro, ra = O.roi_crop_pool(fm, prop, (14, 14), 2)
    np.testing.assert_array_equal(out.cpu().numpy(), ro)
    np.testing.assert_array_equal(win.cpu().numpy(), ra)
    crops, _ = ops.roi_crop_pool_fwd(T(fm, dev), T(prop[:, :50], dev), (14, 14), 1)
    rc = O.crop_and_resize(fm, prop[:, :50].reshape(-1, 4), np.repeat(np.arange(2), 50))
    np.testing.assert_array_equal(crops.cpu().numpy().reshape(rc.shape), rc)
    # NCHW-contiguous storage viewed as NHWC goes through the any-stride kernel
    f_nchw = T(fm, dev).permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    out2, _ = ops.roi_crop_pool_fwd(f_nchw, T(prop[:, :50], dev), (14, 14), 2)
    np.testing.assert_array_equal(out2.cpu().numpy(), ro[:, :50])


def test_plain_crop_with_winner_is_zero(dev):
    """pool = 1: every window has one sample, so the winner bytes are 0 on every kernel path
    (C = 64: staged walk; 32: channels-last; 6: any-stride)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(12)
    for C in (64, 32, 6):
        fm = rng.standard_normal((2, 9, 11, C)).astype(np.float32)
        y1, x1 = rng.uniform(0, 0.6, (2, 7)), rng.uniform(0, 0.6, (2, 7))
        prop = np.stack([y1, x1, y1 + 0.3, x1 + 0.35], -1).astype(np.float32)
        out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (6, 6), 1, want_winner=True)
        win.fill_(7)
        del win
        out, win2 = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (6, 6), 1, want_winner=True)
        # a fresh tensor may reuse the block just filled with 7: zeros prove the call wrote it
        ro, ra = O.roi_crop_pool(fm, prop, (6, 6), 1)
        np.testing.assert_array_equal(out.cpu().numpy(), ro)
        assert not win2.cpu().numpy().any()
        assert not ra.any()


def test_roi_backward_exact(dev):
    from objectdetection_amd import ops
    d = O.synth_inputs(seed=3, B=1, H=9, W=11, A=1, C=8, G=4)
    rng = np.random.default_rng(0)
    R = 12
    y1 = rng.uniform(0, 0.6, R); x1 = rng.uniform(0, 0.6, R)
    prop = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.4, R), x1 + rng.uniform(0.05, 0.4, R)],
                    1).astype(np.float32)[None]
    prop[0, 0] = [0, 0, 1, 1]
    prop[0, 1] = [0, 0, 0, 0]
    fm = d["fmap"]
    out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (14, 14), 2, want_winner=True)
    dY = rng.standard_normal(out.shape).astype(np.float32)
    got = ops.roi_crop_pool_bwd(T(dY, dev), win, T(prop, dev), fm.shape, (14, 14), 2)
    ref = O.roi_crop_pool_bwd(dY, win.cpu().numpy(), fm.shape, prop, (14, 14), 2)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("C,pool,crop,R", [(512, 2, (14, 14), 30), (320, 2, (14, 14), 20),
                                              (64, 1, (6, 6), 25), (2048, 2, (8, 12), 6), (2304, 2, (4, 4), 3),
                                              (12, 2, (14, 14), 10)])
def test_roi_backward_exact_shapes(dev, C, pool, crop, R):
    """wave-per-cell kernel with 1/2/8 channel quads per lane, ragged C, plain crop."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(C + R)
    H, W = 10, 13
    fm = np.maximum(rng.standard_normal((2, H, W, C)).astype(np.float32), 0)
    y1 = rng.uniform(-0.05, 0.7, (2, R)); x1 = rng.uniform(-0.05, 0.7, (2, R))
    prop = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.6, (2, R)), x1 + rng.uniform(0.02, 0.6, (2, R))],
                    -1).astype(np.float32)
    prop[0, 0] = [0, 0, 1, 1]
    prop[0, 1] = [0, 0, 0, 0]
    prop[1, 0] = [0.8, 0.1, 0.1, 0.9]      # flipped
    prop[1, 1] = [0.2, 0.2, 0.2, 0.7]      # zero height: all sample rows coincide
    out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), crop, pool, want_winner=pool > 1)
    ro, ra = O.roi_crop_pool(fm, prop, crop, pool)
    np.testing.assert_array_equal(out.cpu().numpy(), ro)
    dY = rng.standard_normal(ro.shape).astype(np.float32)
    got = ops.roi_crop_pool_bwd(T(dY, dev), win, T(prop, dev), fm.shape, crop, pool)
    ref = O.roi_crop_pool_bwd(dY, ra, fm.shape, prop, crop, pool)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("chunk,cx", [("64", "2"), ("128", "1"), ("64", "1")])
def test_roi_backward_in_ranges_of_rois(dev, chunk, cx, tuning):
    """From 32 images the backward is launched per range of RoIs (later ranges continue the
    earlier sums), and a wave may own one cell instead of two: both forced here on a small
    batch; the sums must not depend on either."""
    from objectdetection_amd import ops
    tuning("bwd_chunk", int(chunk))
    tuning("bwd_cx", int(cx))
    tuning("bwd_band", 0)
    rng = np.random.default_rng(int(chunk) + int(cx))
    B, H, W, C, R = 2, 11, 15, 192, 150
    fm = np.maximum(rng.standard_normal((B, H, W, C)).astype(np.float32), 0)
    y1 = rng.uniform(-0.05, 0.7, (B, R)); x1 = rng.uniform(-0.05, 0.7, (B, R))
    prop = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.6, (B, R)), x1 + rng.uniform(0.02, 0.6, (B, R))],
                    -1).astype(np.float32)
    out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (14, 14), 2, want_winner=True)
    ro, ra = O.roi_crop_pool(fm, prop, (14, 14), 2)
    dY = rng.standard_normal(ro.shape).astype(np.float32)
    got = ops.roi_crop_pool_bwd(T(dY, dev), win, T(prop, dev), fm.shape, (14, 14), 2)
    ref = O.roi_crop_pool_bwd(dY, ra, fm.shape, prop, (14, 14), 2)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [44, 34, 46, 26])
def test_roi_backward_patch_shapes(dev, shape, tuning):
    """The patch-owner backward picks its patch shape by launch size; every shape is forced here
    (ragged map edges in both directions, boxes reaching beyond the image, two channel slices,
    one of them partly beyond C): the sums must not depend on it."""
    from objectdetection_amd import ops
    tuning("bwd_band", 1)
    tuning("bwd_cx", int(shape))
    rng = np.random.default_rng(shape)
    B, H, W, C, R = 2, 13, 17, 320, 130
    fm = np.maximum(rng.standard_normal((B, H, W, C)).astype(np.float32), 0)
    y1 = rng.uniform(-0.05, 0.7, (B, R)); x1 = rng.uniform(-0.05, 0.7, (B, R))
    prop = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.6, (B, R)), x1 + rng.uniform(0.02, 0.6, (B, R))],
                    -1).astype(np.float32)
    prop[0, 3] = [0.0, 0.0, 1.0, 1.0]                  # samples exactly on feature rows / columns
    prop[0, 4] = [0.25, 0.25, 0.25, 0.25]              # a point: every sample in one place
    prop[1, 5] = [0.0, 0.0, 0.0, 0.0]                  # the zero padding of the proposal list
    for crop in ((14, 14), (16, 12)):
        out, win = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), crop, 2, want_winner=True)
        ro, ra = O.roi_crop_pool(fm, prop, crop, 2)
        dY = rng.standard_normal(ro.shape).astype(np.float32)
        got = ops.roi_crop_pool_bwd(T(dY, dev), win, T(prop, dev), fm.shape, crop, 2)
        ref = O.roi_crop_pool_bwd(dY, ra, fm.shape, prop, crop, 2)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", ["keys", "philox"])
def test_making_frcnn_input(dev, ref_inputs, mode):
    from objectdetection_amd import ops
    from objectdetection_amd.module import making_frcnn_input
    a = ref_inputs["anchor"]
    d = ref_inputs
    rng = np.random.default_rng(9)
    if mode == "keys":
        ok = [(rng.permutation(1500).astype(np.uint32), rng.permutation(1500).astype(np.uint32),
               rng.permutation(128).astype(np.uint32)) for _ in range(2)]
        dk = tuple(T(np.stack([k[j] for k in ok]).view(np.int32), dev) for j in range(3))
    else:
        ok = dk = None
    crop, gl, gm, gc, proposed, tindex = making_frcnn_input(
        T(d["gt"], dev), T(d["label"], dev), T(d["fmap"], dev), T(d["coord"], dev),
        T(d["score"], dev), T(a, dev), keys=dk, seed=77)
    # oracle labelling on the device's proposals (identical inputs)
    prop = proposed.cpu().numpy()
    lab = O.label_proposals(prop, d["gt"], d["label"])
    dl = ops.label_proposals(proposed, T(d["gt"], dev), T(d["label"], dev))
    np.testing.assert_array_equal(dl["p_label"].cpu().numpy(), lab["p_iou"])
    np.testing.assert_array_equal(dl["gt_mask"].cpu().numpy(), lab["gt_mask"])
    np.testing.assert_array_equal(dl["best_gt"].cpu().numpy(), lab["best"])
    np.testing.assert_array_equal(dl["max_iou"].cpu().numpy(), lab["max_iou"])
    np.testing.assert_array_equal(dl["gt_onehot"].cpu().numpy(), lab["gt_label"])
    ti = O.sample_proposals(lab["p_iou"], keys=ok, seed=77)
    np.testing.assert_array_equal(tindex.cpu().numpy(), ti)
    np.testing.assert_array_equal(gm.cpu().numpy(), np.take_along_axis(lab["gt_mask"], ti, 1))
    np.testing.assert_array_equal(gl.cpu().numpy(),
                                  np.take_along_axis(lab["gt_label"], ti[..., None], 1))
    ref_gc = np.take_along_axis(lab["gt_coord"], ti[..., None], 1)
    fin = np.isfinite(ref_gc)
    np.testing.assert_array_equal(np.isfinite(gc.cpu().numpy()), fin)
    np.testing.assert_allclose(gc.cpu().numpy()[fin], ref_gc[fin], rtol=1e-5, atol=1e-6)
    rc = O.crop_and_resize(d["fmap"], prop[:, :40].reshape(-1, 4), np.repeat(np.arange(2), 40))
    np.testing.assert_array_equal(crop[:, :40].cpu().numpy().reshape(rc.shape), rc)


def test_roi_pool_autograd(dev):
    from objectdetection_amd import ops
    from objectdetection_amd.module import roi_pool_autograd
    rng = np.random.default_rng(4)
    fm = torch.from_numpy(rng.standard_normal((1, 9, 11, 64)).astype(np.float32)).to(dev)
    fm.requires_grad_(True)
    y1, x1 = rng.uniform(0, 0.6, 15), rng.uniform(0, 0.6, 15)
    prop = T(np.stack([y1, x1, y1 + 0.3, x1 + 0.35], 1).astype(np.float32)[None], dev)
    out = roi_pool_autograd(fm, prop)
    w = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).to(dev)
    (out * w).sum().backward()
    _, win = ops.roi_crop_pool_fwd(fm.detach(), prop, (14, 14), 2, want_winner=True)
    ref = ops.roi_crop_pool_bwd(w, win, prop, tuple(fm.shape), (14, 14), 2)
    assert torch.equal(fm.grad, ref)
    # adjoint identity: <w, out(f)> == <grad, f> for the recorded winners
    lhs, rhs = (out.detach() * w).sum().item(), (fm.grad * fm.detach()).sum().item()
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs))
