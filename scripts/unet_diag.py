"""Diagnosis: one RegressionUNet, eager vs graph-replayed, f16x3 vs exact fp32, at D^3 x C: pairwise rel-L2."""
import sys, os, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.unet import RegressionUNet, SegmentationUNet  # noqa: E402
from pixie_amd.unet_plan import synthetic_state_dict  # noqa: E402
D = int(sys.argv[1]); C = int(sys.argv[2]); which = sys.argv[3] if len(sys.argv) > 3 else "cont"
dev = torch.device("cuda:0")
kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
net = RegressionUNet(out_channels=3, **kw) if which == "cont" else SegmentationUNet(num_classes=8, **kw)
net.load_numpy_state(synthetic_state_dict(net.cfg, 1000 if which == "cont" else 0))
net = net.to(dev).eval()
feat = torch.randn((1, C, D, D, D), generator=torch.Generator(device=dev).manual_seed(7), device=dev).half().float()
res = {}
for prec in ("f16x3", "f32"):
    net.conv_precision = prec
    for graph in (False, True):
        net.use_graph = graph
        for rep in range(3):
            y = net(feat)
        torch.cuda.synchronize()
        res[(prec, "graph" if graph else "eager")] = y[0].clone()
    net.executor = "python"; net.use_graph = False
    res[(prec, "python")] = net(feat)[0].clone()
    net.executor = "c"
for a, b in itertools.combinations(res, 2):
    e = float((res[a].double() - res[b].double()).norm() / res[b].double().norm())
    print(f"D={D} C={C} {which}: {a} vs {b}: {e:.3e}", flush=True)
