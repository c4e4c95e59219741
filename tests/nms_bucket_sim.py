#!/usr/bin/env python3
"""What a size-keyed NMS would save on the bench's boxes (CPU; it uses the oracle, hence its place under tests/: a tool, not product
code): kept boxes sorted by a size key per 1,024-candidate run, a kept tile of 64 rows tested only
against the columns whose key lies within one bucket (width log2(1/0.7): IoU > 0.7 needs height,
width and area each within that factor) of the tile's key range.  Prints the share of the kept
tiles' column tests that remain, for a 1-D area key and for 2-D (height, width) keys, and the
share of the in-chunk triangles, which do not profit."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import frcnn_oracle as O
cfg = bench.workload_cfg()
fmap, score, coord, gt, label = bench.make_inputs(cfg, 1, 8, torch.device('cpu'), 1000)
anc = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)
s = score.numpy().reshape(1, -1); d = coord.numpy().reshape(1, -1, 4)
rv, ri = O.top_k(s, cfg.pre_k)
dec = O._decode(anc.reshape(-1, 4), d[0], clip_wh=False)
cand = dec[ri[0]]                     # [K,4] in score order
keep = O.nms_keep(cand, 0.7, cfg.post_n)
keep = np.asarray(keep)
print("kept", len(keep), "last kept index", keep[-1])
h = np.maximum(cand[:, 2] - cand[:, 0], 1e-9); w = np.maximum(cand[:, 3] - cand[:, 1], 1e-9)
W = np.log2(1 / 0.7)
kh = np.floor(np.log2(h) / W).astype(int); kw = np.floor(np.log2(w) / W).astype(int)
ka = np.floor(np.log2(h * w) / W).astype(int)
nchunk = keep[-1] // 1024 + 1
dense = 0; sk_area = 0; sk_hw = 0; sk_hw_lane = 0
for c in range(1, nchunk):
    cols = np.arange(c * 1024, min((c + 1) * 1024, len(cand)))
    kept_before = keep[keep < c * 1024]
    # runs = kept boxes per earlier chunk, each sorted by key
    for mode in ("area", "hw"):
        order = []
        for r in range(c):
            run = kept_before[(kept_before >= r * 1024) & (kept_before < (r + 1) * 1024)]
            if mode == "area":
                run = run[np.argsort(ka[run], kind="stable")]
            else:
                run = run[np.lexsort((kw[run], kh[run]))]
            order.append(run)
        order = np.concatenate(order)
        ntile = (len(order) + 63) // 64
        for t in range(ntile):
            rows = order[t * 64:(t + 1) * 64]
            if mode == "area":
                lo, hi = ka[rows].min() - 1, ka[rows].max() + 1
                rel = (ka[cols] >= lo) & (ka[cols] <= hi)
                sk_area += rel.sum()
                dense += len(cols)
            else:
                rel = ((kh[cols] >= kh[rows].min() - 1) & (kh[cols] <= kh[rows].max() + 1) &
                       (kw[cols] >= kw[rows].min() - 1) & (kw[cols] <= kw[rows].max() + 1))
                sk_hw += rel.sum()
                # exact per-lane adjacency (vector prefilter): any row adjacent in both
                adj = (np.abs(kh[cols][:, None] - kh[rows][None, :]) <= 1) & (np.abs(kw[cols][:, None] - kw[rows][None, :]) <= 1)
                sk_hw_lane += adj.any(1).sum()
tri = sum(136 * 64 for c in range(nchunk))
print("chunks", nchunk, "kept-tile column tests dense", dense, "tile-equivalents", dense / 64)
print("  1-D area key, tile interval:", sk_area, f"{sk_area / dense:.3f}")
print("  2-D (h,w) keys, tile box   :", sk_hw, f"{sk_hw / dense:.3f}")
print("  2-D per-lane adjacency     :", sk_hw_lane, f"{sk_hw_lane / dense:.3f}")
print("in-chunk triangle column tests", tri, "share of all", tri / (tri + dense))
