import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
for B in (1, 2, 4, 8):
    fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
    for topk in ("sort", "lds"):
        path = RegionProposalPath(cfg, B, 512, dev, train=False, topk=topk)
        for _ in range(3): path.run(fmap, score, coord)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(); path.run(fmap, score, coord); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        g = path.capture(fmap, score, coord)
        tg = []
        for _ in range(20):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            tg.append(e0.elapsed_time(e1))
        print(f"B={B} topk={topk}: run {min(ts):.3f} ms, graph {min(tg):.3f} ms", flush=True)
