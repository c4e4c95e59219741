"""First contact with frcnn_nms_fused: small cases against the chunked NMS, then timing at B = 1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from objectdetection_amd import ops
dev = torch.device("cuda:0")
def rb(rng, B, K, spread):
    cy, cx = rng.uniform(0, 1, (B, K)), rng.uniform(0, 1, (B, K))
    h, w = rng.uniform(0.02, spread, (B, K)), rng.uniform(0.02, spread, (B, K))
    return torch.from_numpy(np.clip(np.stack([cy-h/2, cx-w/2, cy+h/2, cx+w/2], -1), 0, 1).astype(np.float32)).to(dev)
rng = np.random.default_rng(0)
for (B, K, post_n, spread) in [(1, 64, 64, 0.3), (1, 1000, 300, 0.2), (2, 3000, 3000, 0.6), (3, 2500, 50, 0.1), (1, 12000, 2000, 0.4), (4, 6000, 1500, 0.3)]:
    b = rb(rng, B, K, spread)
    p0, k0, c0 = ops.nms(b, 0.7, post_n)
    p1, k1, c1 = ops.nms_fused(b, 0.7, post_n)
    torch.cuda.synchronize()
    print(B, K, post_n, "count", c0.tolist(), c1.tolist(), "equal", torch.equal(k0, k1), torch.equal(p0, p1), flush=True)
cfg = bench.workload_cfg("600x1000")
from objectdetection_amd.pipeline import RegionProposalPath
for B in (1, 2, 4):
    f, s, c, g, l = bench.make_inputs(cfg, B, 64, dev, seed=5)
    path = RegionProposalPath(cfg, B, 64, dev, train=False)
    path.run(f, s, c); torch.cuda.synchronize()
    ws = torch.empty(ops.nms_fused_workspace_bytes(B, cfg.pre_k, cfg.post_n), dtype=torch.uint8, device=dev)
    p1, k1, c1 = ops.nms_fused(path.cand, cfg.nms_thr, cfg.post_n, ws)
    print("B", B, "path boxes equal", torch.equal(k1, path.keep_pos), c1.tolist())
    t_f = bench.timed_alone(lambda: ops.nms_fused(path.cand, cfg.nms_thr, cfg.post_n, ws), dev, reps=30, skip=10)
    t_c = bench.timed_alone(lambda: ops.nms(path.cand, cfg.nms_thr, cfg.post_n, path.nms_ws), dev, reps=30, skip=10)
    cl = bench.clustered_boxes(B, cfg.pre_k, dev, seed=77)
    t_fc = bench.timed_alone(lambda: ops.nms_fused(cl, cfg.nms_thr, cfg.post_n, ws), dev, reps=30, skip=10)
    t_cc = bench.timed_alone(lambda: ops.nms(cl, cfg.nms_thr, cfg.post_n, path.nms_ws), dev, reps=30, skip=10)
    print(f"B {B}: fused {t_f*1e3:.1f} us, chunked {t_c*1e3:.1f} us; clustered fused {t_fc*1e3:.1f} chunked {t_cc*1e3:.1f}", flush=True)
