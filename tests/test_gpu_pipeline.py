"""Two-deep software pipelining of consecutive batches gives the results of the plain path,
batch by batch, with no host synchronisation between the calls."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("top_idx", "cand", "proposed", "keep_pos", "keep_count", "tindex", "gt_coord", "p_label")


def test_depth2_equals_depth1(dev):
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    g = np.load(os.path.join(GOLD, "ref_small.npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], int(g["C"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    gt, label = torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["label"]).to(dev)
    batches = []
    for i in range(5):
        fmap = torch.rand((B, cfg.H, cfg.W, C), generator=gen, device=dev)
        score = torch.rand((B, cfg.H, cfg.W, cfg.A), generator=gen, device=dev)
        coord = torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=gen, device=dev) * 0.5
        batches.append((fmap, score, coord))
    plain = RegionProposalPath(cfg, B, C, dev, train=True)
    want = []
    for i, (f, s, c) in enumerate(batches):
        plain.run(f, s, c, gt, label, seed=i)
        torch.cuda.synchronize()
        want.append({k: getattr(plain, k).clone() for k in KEYS} | {"pooled": plain.pooled.clone(),
                                                                   "rpn_pos": plain.rpn["pos"].clone()})
    piped = RegionProposalPath(cfg, B, C, dev, train=True, depth=2)
    got = []
    for i, (f, s, c) in enumerate(batches):                  # no synchronisation in between
        piped.run(f, s, c, gt, label, seed=i)
        got.append({k: getattr(piped, k).clone() for k in KEYS} | {"pooled": piped.pooled.clone(),
                                                                  "rpn_pos": piped.rpn["pos"].clone()})
    torch.cuda.synchronize()
    for w, h in zip(want, got):
        for k in w:
            assert torch.equal(w[k], h[k]), k
