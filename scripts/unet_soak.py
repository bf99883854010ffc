"""Soak test for the two-network forward at D^3 x C in a long-lived process (recycled allocator blocks, several
captures): every call's outputs must be finite and equal, bit for bit, to the first eager result of their precision."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.unet import RegressionUNet, SegmentationUNet, predict_material_field  # noqa: E402
from pixie_amd.unet_plan import synthetic_state_dict  # noqa: E402

D = int(sys.argv[1]); C = int(sys.argv[2]); rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
sd_s = synthetic_state_dict(SegmentationUNet(num_classes=8, **kw).cfg, 0)
sd_c = synthetic_state_dict(RegressionUNet(out_channels=3, **kw).cfg, 1000)
feat = torch.randn((1, C, D, D, D), generator=torch.Generator(device=dev).manual_seed(7), device=dev).half().float()
ref = {}
bad = 0
for rnd in range(rounds):
    # dirty the allocator's cache: what a long test session leaves behind
    junk = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(8 if D >= 256 else 2)]
    del junk
    seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
    seg.load_numpy_state(sd_s); cont.load_numpy_state(sd_c)
    seg, cont = seg.to(dev).eval(), cont.to(dev).eval()
    for prec in ("f16x3", "f32"):
        seg.conv_precision = cont.conv_precision = prec
        for call in range(4):
            _, _, lg, cp = predict_material_field(seg, cont, feat)
            torch.cuda.synchronize()
            fin = (bool(torch.isfinite(lg).all()), bool(torch.isfinite(cp).all()))
            if prec not in ref:
                ref[prec] = (lg.clone(), cp.clone())
            same = (bool(torch.equal(lg, ref[prec][0])), bool(torch.equal(cp, ref[prec][1])))
            ok = all(fin) and all(same)
            bad += not ok
            print(f"round {rnd} {prec} call {call}: finite {fin} equal-to-first {same} reserved {torch.cuda.memory_reserved() / 2 ** 30:.0f} GiB{'' if ok else '   <-- MISMATCH'}", flush=True)
            del lg, cp
    del seg, cont
    gc.collect()
    if rnd % 2 == 1:
        torch.cuda.empty_cache()
print("soak:", "FAILED" if bad else "ok", bad)
