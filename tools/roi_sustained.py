import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd import ops
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
B = 64
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
path = RegionProposalPath(cfg, B, 512, dev, train=False)
path.run(fmap, score, coord)
torch.cuda.synchronize()
f = lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool, out=path.pooled, workspace=path.roi_ws)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (1, 5, 20, 60, 120):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{n} launches back to back: {e0.elapsed_time(e1)/n:.3f} ms each", flush=True)
