"""Parity at the full sizes of the benchmark configurations.

Config 1: 600x1000 image, 37x62x9 anchors, G=42, 12k -> 2k NMS, C=512  (direct oracle comparison)
Config 2: ship detector, 768x768 -> 48x48x12, batch 16, <=50 small GT   (direct, 2 of 16 images + properties)
Config 3: 1333x800 -> 50x83x15, G=200, 100,000 NMS inputs               (direct oracle comparison + properties)
"""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def nms_properties(boxes, keep, count, thr, post_n):
    """Size-independent checks on a keep list (numpy, vectorised)."""
    n = int(count)
    k = keep[:n].astype(np.int64)
    assert (np.diff(k) > 0).all(), "keep positions ascending"
    kb = boxes[k]
    # kept boxes do not suppress each other (first 400 of them)
    sub = kb[: min(n, 400)]
    for j in range(1, sub.shape[0]):
        assert not (O.nms_iou(sub[j], sub[:j]) > np.float32(thr)).any()
    # every skipped candidate before the last kept one is suppressed by an earlier kept box
    if n:
        last = int(k[-1])
        skipped = np.setdiff1d(np.arange(last), k)[:300]
        for s in skipped:
            earlier = kb[k < s]
            assert (O.nms_iou(boxes[s], earlier) > np.float32(thr)).any()


def test_config1_full_size_direct(dev):
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    cfg = PathConfig(H=37, W=62, base_w=1000.0, base_h=600.0, G=42, pre_k=12000, post_n=2000)
    B, C = 2, 512
    d = O.synth_inputs(seed=101, B=B, H=cfg.H, W=cfg.W, A=cfg.A, C=C, G=cfg.G)
    path = RegionProposalPath(cfg, B, C, dev, train=True, want_winner=True)
    path.run(T(d["fmap"], dev), T(d["score"], dev), T(d["coord"], dev), T(d["gt"], dev),
             T(d["label"], dev), seed=9)
    a = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)
    assert a.shape[0] * a.shape[1] * a.shape[2] == 20646
    np.testing.assert_array_equal(path.anchor.cpu().numpy(), a)
    _, ri = O.top_k(d["score"].reshape(B, -1), cfg.pre_k)
    np.testing.assert_array_equal(path.top_idx.cpu().numpy(), ri)
    cand = path.cand.cpu().numpy()
    ref_cand = np.take_along_axis(O.inverse_trans(a, d["coord"]).reshape(B, -1, 4),
                                  ri.astype(np.int64)[..., None], 1)
    np.testing.assert_allclose(cand, ref_cand, rtol=1e-5, atol=1e-7)
    proposed, keep, counts = O.combined_nms(cand, None, cfg.post_n, cfg.nms_thr)
    np.testing.assert_array_equal(path.keep_pos.cpu().numpy(), keep)
    np.testing.assert_array_equal(path.proposed.cpu().numpy(), proposed)
    ro, ra = O.roi_crop_pool(d["fmap"], proposed, cfg.crop, cfg.pool)
    np.testing.assert_array_equal(path.pooled.cpu().numpy(), ro)
    np.testing.assert_array_equal(path.winner.cpu().numpy(), ra)
    lab = O.label_proposals(proposed, d["gt"], d["label"])
    np.testing.assert_array_equal(path.p_label.cpu().numpy(), lab["p_iou"])
    np.testing.assert_array_equal(path.tindex.cpu().numpy(), O.sample_proposals(lab["p_iou"], seed=9))
    for b in range(B):
        ref = O.patch_batch(dict(image=None, bbox=d["gt"][b], label=d["label"][b]), a, seed=9,
                            image_index=b)
        np.testing.assert_array_equal(path.rpn["pos"][b].cpu().numpy(), ref[5])
        np.testing.assert_array_equal(path.rpn["gt_list"][b].cpu().numpy(), ref[4])


def test_config2_ship_batch16(dev):
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    cfg = PathConfig(H=48, W=48, scales=(32, 64, 128, 256), base_w=500.0, base_h=500.0, G=50,
                     pre_k=6000, post_n=1500)
    B, C = 16, 128
    d = O.synth_inputs(seed=202, B=B, H=cfg.H, W=cfg.W, A=cfg.A, C=C, G=cfg.G,
                       gt_side=(0.01, 0.06))
    path = RegionProposalPath(cfg, B, C, dev, train=True)
    path.run(T(d["fmap"], dev), T(d["score"], dev), T(d["coord"], dev), T(d["gt"], dev),
             T(d["label"], dev), seed=4)
    a = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)
    assert a.reshape(-1, 4).shape[0] == 27648
    _, ri = O.top_k(d["score"].reshape(B, -1), cfg.pre_k)
    np.testing.assert_array_equal(path.top_idx.cpu().numpy(), ri)
    cand = path.cand.cpu().numpy()
    keep = path.keep_pos.cpu().numpy()
    cnt = path.keep_count.cpu().numpy()
    prop = path.proposed.cpu().numpy()
    for b in (0, 15):                                         # direct on two images
        rp, rk, rc = O.combined_nms(cand[b:b + 1], None, cfg.post_n, cfg.nms_thr)
        np.testing.assert_array_equal(keep[b], rk[0])
        np.testing.assert_array_equal(prop[b], rp[0])
        ro, _ = O.roi_crop_pool(d["fmap"][b:b + 1], rp, cfg.crop, cfg.pool)
        np.testing.assert_array_equal(path.pooled[b].cpu().numpy(), ro[0])
        ref = O.patch_batch(dict(image=None, bbox=d["gt"][b], label=d["label"][b]), a, seed=4,
                            image_index=b)
        np.testing.assert_array_equal(path.rpn["pos"][b].cpu().numpy(), ref[5])
    for b in range(B):                                        # properties on all 16
        nms_properties(cand[b], keep[b], cnt[b], cfg.nms_thr, cfg.post_n)
        assert (prop[b, cnt[b]:] == 0).all()
    lab = O.label_proposals(prop, d["gt"], d["label"])
    np.testing.assert_array_equal(path.p_label.cpu().numpy(), lab["p_iou"])
    np.testing.assert_array_equal(path.tindex.cpu().numpy(), O.sample_proposals(lab["p_iou"], seed=4))


def test_config3_dense_100k_nms(dev):
    from objectdetection_amd import ops
    rng = np.random.default_rng(303)
    K, post_n = 100_000, 2000
    cy, cx = rng.uniform(0, 1, K), rng.uniform(0, 1, K)
    h, w = np.exp(rng.uniform(np.log(0.01), np.log(0.3), K)), np.exp(rng.uniform(np.log(0.01), np.log(0.3), K))
    boxes = np.clip(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1), 0, 1)
    boxes = boxes.astype(np.float32)[None]
    prop, keep, cnt = ops.nms(T(boxes, dev), 0.7, post_n)
    rp, rk, rc = O.combined_nms(boxes, None, post_n, 0.7)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    np.testing.assert_array_equal(prop.cpu().numpy(), rp)
    prop, keep, cnt = ops.nms_fused(T(boxes, dev), 0.7, post_n)        # the one-launch route
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    np.testing.assert_array_equal(prop.cpu().numpy(), rp)
    # a tight cluster that forces the scan through all 1,563 column blocks
    c = 0.5 + rng.normal(0, 0.003, (K, 2))
    half = 0.2 + rng.normal(0, 0.002, (K, 2))
    dense = np.concatenate([c - half, c + half], 1).astype(np.float32)[None]
    prop, keep, cnt = ops.nms(T(dense, dev), 0.7, post_n)
    rp, rk, rc = O.combined_nms(dense, None, post_n, 0.7)
    assert rc[0] < post_n
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    _, keep_f, cnt_f = ops.nms_fused(T(dense, dev), 0.7, post_n)       # all 98 chunks, one launch
    np.testing.assert_array_equal(keep_f.cpu().numpy(), rk)
    assert int(cnt_f[0]) == rc[0]
    nms_properties(dense[0], keep[0].cpu().numpy(), int(cnt[0]), 0.7, post_n)
    # idempotence: NMS of the kept boxes keeps every one of them
    kept = rp[:, : rc[0]]
    _, keep2, cnt2 = ops.nms(T(kept, dev), 0.7, post_n)
    assert int(cnt2[0]) == rc[0]


def test_config3_anchor_assignment_g200(dev):
    from objectdetection_amd import ops
    H, W = 50, 83
    scales, ratios = (64, 128, 256, 384, 512), (0.5, 1, 2)
    a = O.make_anchor(H, W, scales, ratios, 1333.0, 800.0)
    assert a.reshape(-1, 4).shape[0] == 62250
    d = O.synth_inputs(seed=7, B=1, H=H, W=W, A=15, C=4, G=200, n_gt=200, with_fmap=False)
    da = ops.anchors(H, W, scales, ratios, 1333.0, 800.0, dev)
    np.testing.assert_array_equal(da.cpu().numpy(), a)
    mi, best, flags, _ = ops.iou_assign(da.reshape(-1, 4), T(d["gt"], dev))
    st = O.rpn_assign(a, d["gt"][0])
    np.testing.assert_array_equal(best[0].cpu().numpy(), st["best"])
    np.testing.assert_array_equal(mi[0].cpu().numpy(), st["max_iou"])
    out = ops.rpn_sample(da, T(d["gt"], dev), best, flags, seed=3)
    ref = O.patch_batch(dict(image=None, bbox=d["gt"][0], label=d["label"][0]), a, seed=3)
    np.testing.assert_array_equal(out["pos"][0].cpu().numpy(), ref[5])
    np.testing.assert_array_equal(out["gt_list"][0].cpu().numpy(), ref[4])


def test_config3_whole_path_50x83x15_g200(dev):
    """Config 3's map at its stated size inside the path: 1333x800 -> 50x83x15 = 62,250 anchors,
    G = 200, top-12,000 of 62,250 (the radix-sort top-k at that N), NMS -> 2,000, RoI crop + pool
    on the 50x83x512 map (four bands of rows), proposal labels and head sampling at G = 200."""
    from oracle import c_oracle
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    cfg = PathConfig(H=50, W=83, scales=(64, 128, 256, 384, 512), ratios=(0.5, 1, 2),
                     base_w=1333.0, base_h=800.0, G=200, pre_k=12000, post_n=2000)
    assert cfg.N == 62250
    B, C = 2, 512
    lib = c_oracle.load()
    d = O.synth_inputs(seed=33, B=B, H=cfg.H, W=cfg.W, A=cfg.A, C=C, G=cfg.G, n_gt=200)
    for topk in ("sort", "lds"):
        path = RegionProposalPath(cfg, B, C, dev, train=True, want_winner=True, topk=topk)
        path.run(T(d["fmap"], dev), T(d["score"], dev), T(d["coord"], dev), T(d["gt"], dev),
                 T(d["label"], dev), seed=6)
        _, ri = O.top_k(d["score"].reshape(B, -1), cfg.pre_k)
        np.testing.assert_array_equal(path.top_idx.cpu().numpy(), ri)
        cand = path.cand.cpu().numpy()
        prop = path.proposed.cpu().numpy()
        for b in range(B):
            rp, rk, rc = c_oracle.nms(lib, cand[b], cfg.nms_thr, cfg.post_n)
            np.testing.assert_array_equal(path.keep_pos[b].cpu().numpy(), rk)
            np.testing.assert_array_equal(prop[b], rp)
        lab = O.label_proposals(prop, d["gt"], d["label"])
        np.testing.assert_array_equal(path.p_label.cpu().numpy(), lab["p_iou"])
        np.testing.assert_array_equal(path.p_best.cpu().numpy(), lab["best"])
        np.testing.assert_array_equal(path.p_iou.cpu().numpy(), lab["max_iou"])
        np.testing.assert_array_equal(path.gt_mask.cpu().numpy(), lab["gt_mask"])
        assert (lab["p_iou"] != cfg.bg_label).sum() > 0
        np.testing.assert_array_equal(path.tindex.cpu().numpy(), O.sample_proposals(lab["p_iou"], seed=6))
        ro, ra = c_oracle.roi_crop_pool(lib, d["fmap"][1], prop[1], cfg.crop, cfg.pool)
        np.testing.assert_array_equal(path.pooled[1].cpu().numpy(), ro)
        np.testing.assert_array_equal(path.winner[1].cpu().numpy(), ra)
        del path
    # the band-resident kernel on the same map (it is chosen from 768 workgroups on: B = 12 here)
    from objectdetection_amd import ops
    Bb = 12
    assert ops.roi_fwd_kernel_name(Bb, cfg.H, cfg.W, C, cfg.post_n, cfg.crop, cfg.pool).startswith(
        "roi_fwd_band")
    fm = torch.relu(torch.randn((Bb, cfg.H, cfg.W, C), device=dev,
                                generator=torch.Generator(device=dev).manual_seed(3)))
    rois = T(prop[:1], dev).expand(Bb, -1, -1).contiguous()
    ws = torch.empty(ops.roi_fwd_workspace_bytes(Bb, cfg.post_n), dtype=torch.uint8, device=dev)
    out, _ = ops.roi_crop_pool_fwd(fm, rois, cfg.crop, cfg.pool, workspace=ws)
    for b in (0, 11):
        ro, _ = c_oracle.roi_crop_pool(lib, fm[b].cpu().numpy(), prop[0], cfg.crop, cfg.pool,
                                       want_winner=False)
        np.testing.assert_array_equal(out[b].cpu().numpy(), ro)


def test_config2_ship_batch16_c512_band_kernel(dev):
    """Config 2 at the bench's own size: 16 ship images, 48x48x12 anchors, C = 512 -- the
    band-resident RoI kernel on the 48x48x512 map.  Pooled tensor of two images against the C
    oracle, keep lists of the same two against the sequential NMS."""
    from oracle import c_oracle
    from objectdetection_amd import ops
    from objectdetection_amd.pipeline import RegionProposalPath
    import bench
    cfg = bench.workload_cfg("ship768")
    B, C = 16, 512
    assert ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, cfg.post_n, cfg.crop, cfg.pool) == \
        "roi_fwd_band_kernel<2,false>"
    lib = c_oracle.load()
    f, s, c, g, l = bench.make_inputs(cfg, B, C, dev, seed=2202, gt_side=(0.01, 0.06))
    path = RegionProposalPath(cfg, B, C, dev, train=True, depth=2)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))
    path.run(f, s, c, g, l, seed=1, inputs_ready=ready)
    torch.cuda.synchronize()
    cand, prop = path.cand.cpu().numpy(), path.proposed.cpu().numpy()
    for b in (2, 13):
        rp, rk, rc = c_oracle.nms(lib, cand[b], cfg.nms_thr, cfg.post_n)
        np.testing.assert_array_equal(path.keep_pos[b].cpu().numpy(), rk)
        np.testing.assert_array_equal(prop[b], rp)
        ro, _ = c_oracle.roi_crop_pool(lib, f[b].cpu().numpy(), prop[b], cfg.crop, cfg.pool,
                                       want_winner=False)
        np.testing.assert_array_equal(path.pooled[b].cpu().numpy(), ro)
    lab = O.label_proposals(prop, g.cpu().numpy(), l.cpu().numpy())
    np.testing.assert_array_equal(path.p_label.cpu().numpy(), lab["p_iou"])
    np.testing.assert_array_equal(path.tindex.cpu().numpy(), O.sample_proposals(lab["p_iou"], seed=1))


def _cfg1():
    from objectdetection_amd.config import PathConfig
    return PathConfig(H=37, W=62, base_w=1000.0, base_h=600.0, G=42, pre_k=12000, post_n=2000)


def test_config4_batch64_two_batches_in_flight(dev):
    """Config 4 at its stated size: 600x1000, 64 images per step, C = 512, two batches in
    flight.  Three different batches go through back to back without a host synchronisation;
    every image's proposals, keep list and sampled indices must equal those of a plain
    one-batch-at-a-time path, and the pooled tensor of 4 images per batch must equal the C
    oracle's crop + pool on the device's proposals (bit-exact)."""
    from oracle import c_oracle
    from objectdetection_amd.pipeline import RegionProposalPath
    import bench
    cfg = _cfg1()
    B, C = 64, 512
    lib = c_oracle.load()
    batches = [bench.make_inputs(cfg, B, C, dev, seed=4000 + i) for i in range(3)]
    keys = ("proposed", "keep_pos", "keep_count", "tindex", "p_label")
    check_imgs = torch.tensor([0, 21, 42, 63], device=dev)
    piped = RegionProposalPath(cfg, B, C, dev, train=True, depth=2)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))
    got = []
    for i, (f, s, c, g, l) in enumerate(batches):             # no synchronisation in between
        piped.run(f, s, c, g, l, seed=i, image_base=0, inputs_ready=ready)
        got.append({k: getattr(piped, k).clone() for k in keys}
                   | {"pooled": piped.pooled.index_select(0, check_imgs),
                      "rpn_pos": piped.rpn["pos"].clone()})
    torch.cuda.synchronize()
    del piped
    plain = RegionProposalPath(cfg, B, C, dev, train=True, depth=1)
    for i, (f, s, c, g, l) in enumerate(batches):
        plain.run(f, s, c, g, l, seed=i, image_base=0)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[i][k], getattr(plain, k)), (i, k)
        assert torch.equal(got[i]["rpn_pos"], plain.rpn["pos"]), i
        assert torch.equal(got[i]["pooled"], plain.pooled.index_select(0, check_imgs)), i
        assert int(plain.keep_count.min()) > 0
        prop = plain.proposed.cpu().numpy()
        for j, b in enumerate(check_imgs.tolist()):
            ro, _ = c_oracle.roi_crop_pool(lib, f[b].cpu().numpy(), prop[b], cfg.crop, cfg.pool,
                                           want_winner=False)
            np.testing.assert_array_equal(got[i]["pooled"][j].cpu().numpy(), ro)
        # keep lists against the sequential greedy NMS on the device's candidates: every image
        # of the first batch, two of the others
        cand = plain.cand.cpu().numpy()
        for b in (range(B) if i == 0 else (5, 60)):
            rp, rk, rc = c_oracle.nms(lib, cand[b], cfg.nms_thr, cfg.post_n)
            np.testing.assert_array_equal(got[i]["keep_pos"][b].cpu().numpy(), rk)
            np.testing.assert_array_equal(got[i]["proposed"][b].cpu().numpy(), rp)


def test_config4_clustered_boxes_nms(dev):
    """The bench's second NMS input at its full size: 64 images of 12,000 candidates in 40 tight
    clusters (some 42 kept, every chunk walked) -- the case in which most of the mask kernel's
    workgroups find nothing to do.  Keep lists of every image against the sequential NMS."""
    from oracle import c_oracle
    from objectdetection_amd import ops
    import bench
    cfg = _cfg1()
    lib = c_oracle.load()
    cl = bench.clustered_boxes(64, cfg.pre_k, dev, seed=77)
    prop, keep, cnt = ops.nms(cl, cfg.nms_thr, cfg.post_n)
    boxes = cl.cpu().numpy()
    for b in range(64):
        rp, rk, rc = c_oracle.nms(lib, boxes[b], cfg.nms_thr, cfg.post_n)
        assert int(cnt[b]) == rc
        np.testing.assert_array_equal(keep[b].cpu().numpy(), rk)
        np.testing.assert_array_equal(prop[b].cpu().numpy(), rp)


def test_config5_train_batch8_roi_backward_full_size(dev):
    """Config 5's path at its stated size: 8 images per GPU, 37x62x512 map, 2,000 RoIs:
    forward with the winner index, then the RoI backward.  One image's whole gradient map
    against the C oracle (bit-exact: the summation order is part of the definition), the
    adjoint identity <dY, pool(f)> = <dF, f> on all 8."""
    from oracle import c_oracle
    from objectdetection_amd import ops
    from objectdetection_amd.pipeline import RegionProposalPath
    import bench
    cfg = _cfg1()
    B, C = 8, 512
    lib = c_oracle.load()
    f, s, c, g, l = bench.make_inputs(cfg, B, C, dev, seed=5005)
    path = RegionProposalPath(cfg, B, C, dev, train=True, want_winner=True)
    path.run(f, s, c, g, l, seed=2)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    dY = torch.randn(path.pooled.shape, generator=gen, device=dev)
    dF = ops.roi_crop_pool_bwd(dY, path.winner, path.proposed, tuple(f.shape), cfg.crop, cfg.pool)
    torch.cuda.synchronize()
    prop = path.proposed.cpu().numpy()
    for b in (3,):
        ro, ra = c_oracle.roi_crop_pool(lib, f[b].cpu().numpy(), prop[b], cfg.crop, cfg.pool)
        np.testing.assert_array_equal(path.pooled[b].cpu().numpy(), ro)
        np.testing.assert_array_equal(path.winner[b].cpu().numpy(), ra)
        ref = c_oracle.roi_crop_pool_bwd(lib, dY[b].cpu().numpy(), ra, prop[b], (cfg.H, cfg.W, C),
                                         cfg.crop, cfg.pool)
        np.testing.assert_array_equal(dF[b].cpu().numpy(), ref)
    for b in range(B):
        lhs = (path.pooled[b].double() * dY[b].double()).sum().item()
        rhs = (dF[b].double() * f[b].double()).sum().item()
        assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs), abs(rhs)), (b, lhs, rhs)
