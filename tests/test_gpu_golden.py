"""HIP path vs the committed golden fixtures (through the C ABI)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.mark.parametrize("name", ["ref_small", "ship_small"])
def test_hip_path_matches_golden(dev, name):
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    from objectdetection_amd import ops
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], int(g["C"])
    path = RegionProposalPath(cfg, B, C, dev, train=True, want_winner=True)
    path.run(T(g["fmap"], dev), T(g["score"], dev), T(g["coord"], dev), T(g["gt"], dev),
             T(g["label"], dev), seed=int(g["seed"]))
    np.testing.assert_array_equal(path.top_idx.cpu().numpy(), g["top_idx"])
    np.testing.assert_allclose(path.cand.cpu().numpy(), g["cand"], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(path.rpn["gt_list"].cpu().numpy(), g["rpn_gt_list"])
    np.testing.assert_array_equal(path.rpn["pos"].cpu().numpy(), g["rpn_pos"])
    np.testing.assert_array_equal(path.rpn["label"].cpu().numpy()[..., None], g["rpn_label"])
    np.testing.assert_array_equal(path.rpn["anchor"].cpu().numpy(), g["rpn_anchor"])
    # the stages after decode are checked on the frozen boxes (identical inputs)
    proposed, keep, counts = ops.nms(T(g["cand"], dev), 0.7, cfg.post_n)
    np.testing.assert_array_equal(keep.cpu().numpy(), g["keep_pos"])
    np.testing.assert_array_equal(counts.cpu().numpy(), g["counts"])
    np.testing.assert_array_equal(proposed.cpu().numpy(), g["proposed"])
    pooled, win = ops.roi_crop_pool_fwd(T(g["fmap"], dev), T(g["proposed"][:, :24], dev), (14, 14),
                                        2, want_winner=True)
    np.testing.assert_array_equal(pooled.cpu().numpy(), g["pooled24"])
    np.testing.assert_array_equal(win.cpu().numpy(), g["winner24"])
    lab = ops.label_proposals(T(g["proposed"], dev), T(g["gt"], dev), T(g["label"], dev))
    np.testing.assert_array_equal(lab["p_label"].cpu().numpy(), g["p_label"])
    np.testing.assert_array_equal(lab["best_gt"].cpu().numpy(), g["best"])
    tindex, _ = ops.sample_proposals(lab["p_label"], seed=int(g["seed"]))
    np.testing.assert_array_equal(tindex.cpu().numpy(), g["tindex"])


def test_next_rows_match_golden(dev):
    """Detection post-process, loss assembly and input side against the frozen fixture."""
    from objectdetection_amd import ops
    g = np.load(os.path.join(GOLD, "next_rows_small.npz"), allow_pickle=False)
    det = ops.detect_postprocess(T(g["proposed"], dev), T(g["cls_prob"], dev), T(g["reg"], dev),
                                 bg_label=20, max_out=30, iou_threshold=0.7, score_threshold=0.5)
    np.testing.assert_array_equal(det["count"].cpu().numpy(), g["det_count"])
    np.testing.assert_array_equal(det["src"].cpu().numpy(), g["det_src"])
    np.testing.assert_array_equal(det["classes"].cpu().numpy(), g["det_classes"])
    np.testing.assert_allclose(det["boxes"].cpu().numpy(), g["det_boxes"], rtol=1e-5, atol=1e-7)
    r = ops.rpn_loss(T(g["pred_obj"], dev), T(g["pred_reg"], dev), T(g["pos"], dev),
                     T(g["label"], dev), T(g["reg_gt"], dev))
    np.testing.assert_allclose(r["loss"].cpu().numpy(), g["rpn_loss"], rtol=1e-5)
    np.testing.assert_allclose(r["d_obj"].cpu().numpy(), g["rpn_d_obj"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(r["d_reg"].cpu().numpy(), g["rpn_d_reg"], rtol=1e-5, atol=1e-9)
    h = ops.head_loss(T(g["cls_prob"], dev), T(g["reg"], dev), T(g["tindex"], dev),
                      T(g["onehot"], dev), T(g["gt_mask"], dev), T(g["gt_coord"], dev))
    np.testing.assert_allclose(h["loss"].cpu().numpy(), g["head_loss"], rtol=1e-5)
    np.testing.assert_allclose(h["d_cls"].cpu().numpy(), g["head_d_cls"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(h["d_reg"].cpu().numpy(), g["head_d_reg"], rtol=1e-5, atol=1e-9)
    gt, gl, _ = ops.pad_gt(T(g["in_boxes"], dev), T(g["in_labels"], dev), T(g["in_offsets"], dev),
                           int(g["G"]), None, True, seed=int(g["seed"]))
    np.testing.assert_array_equal(gt.cpu().numpy(), g["aug_boxes"])
    np.testing.assert_array_equal(gl.cpu().numpy(), g["aug_labels"])
