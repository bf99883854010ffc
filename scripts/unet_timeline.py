"""One network forward repeated under a chosen executor, for a rocprofv3 kernel trace (scripts/rocpd_gaps.py reads it).
usage: unet_timeline.py D executor(c|python) graph(0|1) [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixie_amd.synthetic import feature_grid
from pixie_amd.unet import SegmentationUNet
from pixie_amd.unet_plan import synthetic_state_dict

D, executor, graph = int(sys.argv[1]), sys.argv[2], sys.argv[3] == "1"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
net = SegmentationUNet(64, 32, 64, 3, (1, 1, 2, 4), (8,), D, 8)
net.load_numpy_state(synthetic_state_dict(net.cfg, 0))
net = net.to(dev).eval()
net.executor, net.use_graph = executor, graph
feat = torch.from_numpy(feature_grid(D, 64, seed=1)).to(dev)
for _ in range(4):
    net(feat)
torch.cuda.synchronize()
for _ in range(reps):
    net(feat)
torch.cuda.synchronize()
