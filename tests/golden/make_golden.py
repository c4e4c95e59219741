#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ from the CPU oracle.

The reference (TensorFlow 2.3, no tests, not installable here) cannot produce
vectors, so these are outputs of the oracle on seeded inputs, frozen so that
(a) later edits of the oracle are caught and (b) the GPU tests have fixed
expected values that do not depend on numpy's version of exp/log.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import frcnn_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, seed, H, W, scales, G, C, pre_k, post_n, B=2):
    A = len(scales) * 3
    a = O.make_anchor(H, W, scales, (0.5, 1, 2))
    d = O.synth_inputs(seed, B, H, W, A, C, G)
    pc = O.proposal_creator(d["score"], d["coord"], a, pre_k, post_n, 0.7)
    pooled, win = O.roi_crop_pool(d["fmap"], pc["proposed"][:, :24], (14, 14), 2)
    lab = O.label_proposals(pc["proposed"], d["gt"], d["label"])
    tindex = O.sample_proposals(lab["p_iou"], seed=seed)
    pb = [O.patch_batch(dict(image=None, bbox=d["gt"][b], label=d["label"][b]), a, seed=seed,
                        image_index=b) for b in range(B)]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        H=H, W=W, scales=np.asarray(scales, np.float64), G=G, C=C, pre_k=pre_k, post_n=post_n,
        seed=seed, score=d["score"], coord=d["coord"], fmap=d["fmap"], gt=d["gt"], label=d["label"],
        anchor_valid_count=int(O.anchor_valid(a).sum()),
        top_idx=pc["top_idx"], cand=pc["cand"], keep_pos=pc["keep_pos"], counts=pc["counts"],
        proposed=pc["proposed"], pooled24=pooled, winner24=win,
        p_label=lab["p_iou"], gt_mask=lab["gt_mask"], best=lab["best"], tindex=tindex,
        rpn_label=np.stack([p[1] for p in pb]), rpn_gt_list=np.stack([p[4] for p in pb]),
        rpn_pos=np.stack([p[5] for p in pb]), rpn_anchor=np.stack([p[2] for p in pb]))


if __name__ == "__main__":
    case("ref_small", seed=21, H=12, W=12, scales=(128, 256, 512), G=10, C=8, pre_k=700, post_n=150)
    case("ship_small", seed=22, H=16, W=16, scales=(32, 64, 128, 256), G=24, C=8, pre_k=1500,
         post_n=200)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
