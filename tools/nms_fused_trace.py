"""Time stamps of frcnn_nms_fused (FRCNN_HIP_LIB = an FRCNN_NF_TRACE build), B = 1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from objectdetection_amd import ops
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg("600x1000")
B = 1
f, s, c, g, l = bench.make_inputs(cfg, B, 64, dev, seed=5)
path = RegionProposalPath(cfg, B, 64, dev, train=False)
path.run(f, s, c); torch.cuda.synchronize()
nb = ops.nms_fused_workspace_bytes(B, cfg.pre_k, cfg.post_n)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
for _ in range(5):
    ops.nms_fused(path.cand, cfg.nms_thr, cfg.post_n, ws)
torch.cuda.synchronize()
tr = ws[nb - 16384:].view(torch.int64).cpu().numpy()
t = tr >> 8; x = tr & 255
t0 = t[0]
print("resolver: chunk | wait_start wait_end loaded relaxed(rounds) published(count&255)   [us from start]")
for c_ in range(12):
    r = t[c_*8:c_*8+5]
    if r[0] == 0: break
    print(c_, " ".join(f"{(v - t0)/100:8.2f}" for v in r), "rounds", x[c_*8+3], "cnt", x[c_*8+4])
print("worker 0: stage | wait_start wait_end jobs_done signaled")
for s_ in range(12):
    r = t[1024 + s_*4: 1024 + s_*4 + 4]
    if r[0] == 0: break
    print(s_, " ".join(f"{(v - t0)/100:8.2f}" for v in r), "njobs&255", x[1024+s_*4+2])
print("worker 0 wave 0 (triangle tile): stage | fenced probe_loaded tile_done")
for s_ in range(7):
    r = t[1536 + s_*4: 1536 + s_*4 + 3]
    if r[0] == 0: break
    print(s_, " ".join(f"{(v - t0)/100:8.2f}" for v in r))
dr, dc = tr[2042] - tr[2040], tr[2043] - tr[2041]
print(f"resolver: {dr/100:.1f} us real, {dc} shader-clock ticks -> {dc/(dr/100)/1e3:.3f} GHz")
