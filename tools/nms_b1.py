import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd import ops
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
B = int(os.environ.get("NMS_B", "1"))
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 64, dev, 1000)
path = RegionProposalPath(cfg, B, 64, dev, train=False)
path.run(fmap, score, coord)
torch.cuda.synchronize()
for _ in range(20):
    ops.nms(path.cand, cfg.nms_thr, cfg.post_n, workspace=path.nms_ws)
    torch.cuda.synchronize()
