"""The oracle must reproduce the frozen fixtures (integer results exactly; box
values to 1e-6, which only guards numpy's exp implementation)."""
import os

import numpy as np
import pytest

from oracle import frcnn_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["ref_small", "ship_small"]


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    g = load(name)
    H, W, G, C = int(g["H"]), int(g["W"]), int(g["G"]), int(g["C"])
    scales = tuple(g["scales"].tolist())
    a = O.make_anchor(H, W, scales, (0.5, 1, 2))
    assert int(O.anchor_valid(a).sum()) == int(g["anchor_valid_count"])
    pc = O.proposal_creator(g["score"], g["coord"], a, int(g["pre_k"]), int(g["post_n"]), 0.7)
    np.testing.assert_array_equal(pc["top_idx"], g["top_idx"])
    np.testing.assert_allclose(pc["cand"], g["cand"], rtol=1e-6, atol=1e-7)
    # keep list on the frozen candidate boxes is exact
    proposed, keep, counts = O.combined_nms(g["cand"], None, int(g["post_n"]), 0.7)
    np.testing.assert_array_equal(keep, g["keep_pos"])
    np.testing.assert_array_equal(counts, g["counts"])
    np.testing.assert_array_equal(proposed, g["proposed"])
    pooled, win = O.roi_crop_pool(g["fmap"], g["proposed"][:, :24], (14, 14), 2)
    np.testing.assert_array_equal(pooled, g["pooled24"])
    np.testing.assert_array_equal(win, g["winner24"])
    lab = O.label_proposals(g["proposed"], g["gt"], g["label"])
    np.testing.assert_array_equal(lab["p_iou"], g["p_label"])
    np.testing.assert_array_equal(lab["best"], g["best"])
    np.testing.assert_array_equal(O.sample_proposals(lab["p_iou"], seed=int(g["seed"])), g["tindex"])
    for b in range(g["gt"].shape[0]):
        pb = O.patch_batch(dict(image=None, bbox=g["gt"][b], label=g["label"][b]), a,
                           seed=int(g["seed"]), image_index=b)
        np.testing.assert_array_equal(pb[4], g["rpn_gt_list"][b])
        np.testing.assert_array_equal(pb[5], g["rpn_pos"][b])
        np.testing.assert_array_equal(pb[1], g["rpn_label"][b])


def test_oracle_reproduces_next_rows_golden():
    g = load("next_rows_small")
    det = O.detect_postprocess(g["proposed"], g["cls_prob"], g["reg"], bg_label=20, max_out=30,
                               iou_threshold=0.7, score_threshold=0.5)
    np.testing.assert_array_equal(det["src"], g["det_src"])
    np.testing.assert_array_equal(det["classes"], g["det_classes"])
    np.testing.assert_array_equal(det["count"], g["det_count"])
    np.testing.assert_allclose(det["boxes"], g["det_boxes"], rtol=1e-6, atol=1e-7)
    rl, d_obj, d_reg = O.rpn_loss(g["pred_obj"], g["pred_reg"], g["pos"], g["label"], g["reg_gt"])
    np.testing.assert_allclose(rl, g["rpn_loss"], rtol=1e-6)
    np.testing.assert_allclose(d_obj, g["rpn_d_obj"], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(d_reg, g["rpn_d_reg"], rtol=1e-5, atol=1e-10)
    B, R, K = g["cls_prob"].shape
    hl, d_cls, d_reg2 = O.head_loss(g["cls_prob"], g["reg"].reshape(B, R, K, 4), g["tindex"],
                                    g["onehot"], g["gt_mask"], g["gt_coord"])
    np.testing.assert_allclose(hl, g["head_loss"], rtol=1e-6)
    np.testing.assert_allclose(d_cls, g["head_d_cls"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(d_reg2, g["head_d_reg"], rtol=1e-5, atol=1e-10)
    off = g["in_offsets"]
    for b in range(len(off) - 1):
        _, bb, ll = O.random_augmentation(None, g["in_labels"][off[b]:off[b + 1]],
                                          g["in_boxes"][off[b]:off[b + 1]], int(g["G"]),
                                          seed=int(g["seed"]), image_index=b)
        np.testing.assert_array_equal(bb, g["aug_boxes"][b])
        np.testing.assert_array_equal(ll, g["aug_labels"][b])
        assert O.flip_decision(int(g["seed"]), b) == bool(g["flips"][b])
