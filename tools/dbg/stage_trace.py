import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
B = int(os.environ.get("NMS_B", "64"))
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
path = RegionProposalPath(cfg, B, 512, dev, train=True, depth=1)
for i in range(6):
    path.run(fmap, score, coord, gt, label, seed=i, skip_roi=True)
    torch.cuda.synchronize()
