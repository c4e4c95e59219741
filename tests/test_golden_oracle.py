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
