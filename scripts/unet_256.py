"""BASELINE configs[3] shape on one GPU: both networks on a 256^3 x 128 feature grid (217 TFLOP per scene).
No CPU oracle finishes at this size, so the checks are size-independent: (1) one full-resolution conv launch agrees
between the two independent kernels (f16x3 and exact-fp32 MFMA); (2) the whole forward is finite, the one-hot channels
sum to 1 per voxel, and the f16x3 and fp32 executions of the same network agree."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.unet import HipOps, RegressionUNet, SegmentationUNet, predict_material_field  # noqa: E402
from pixie_amd.unet_plan import synthetic_state_dict  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
ops = HipOps(dev)

# (1) one 64 -> 64 3^3 launch at D^3, both kernels
x = torch.randn((64, D, D, D), generator=g, device=dev)
w = torch.randn((64, 64, 3, 3, 3), generator=g, device=dev) / (64 * 27) ** 0.5
b = torch.randn(64, generator=g, device=dev)
slot = torch.zeros(1, dtype=torch.int32, device=dev)
ops.channel_stats(x, slot)
y16 = ops.conv([x], None, b, 64, 3, w16=ops.pack_conv16(w), in_amax=[slot])
y32 = ops.conv([x], ops.pack_conv(w), b, 64, 3)
torch.cuda.synchronize()
err = float((y16 - y32).norm() / y32.norm())
print(f"conv 64->64 3^3 at {D}^3: f16x3 vs exact-fp32 kernel rel-L2 {err:.2e}")
assert err < 1e-5
del x, y16, y32
torch.cuda.empty_cache()

# (2) the two networks
kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
seg.load_numpy_state(synthetic_state_dict(seg.cfg, 0))
cont.load_numpy_state(synthetic_state_dict(cont.cfg, 1000))
seg, cont = seg.to(dev).eval(), cont.to(dev).eval()
feat = torch.randn((1, C, D, D, D), generator=g, device=dev).half().float()
zz = torch.arange(D, device=dev, dtype=torch.float32) - (D - 1) / 2
occ = (zz[:, None, None] ** 2 + zz[None, :, None] ** 2 + zz[None, None, :] ** 2) < (0.35 * D) ** 2
feat *= occ
res = {}
for prec in ("f16x3", "f32"):
    seg.conv_precision = cont.conv_precision = prec
    combined, seg_pred, _, cont_pred = predict_material_field(seg, cont, feat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    combined, seg_pred, seg_logits, cont_pred = predict_material_field(seg, cont, feat)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(combined).all())
    assert bool((combined[0, 3:].sum(0) == 1).all())
    res[prec] = (seg_logits.clone(), cont_pred.clone(), seg_pred.clone())
    print(f"{prec}: {D}^3 x {C} two-network forward {dt * 1e3:.1f} ms = {D ** 3 / dt / 1e6:.2f} M voxels/s; peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
e_seg = float((res["f16x3"][0] - res["f32"][0]).norm() / res["f32"][0].norm())
e_cont = float((res["f16x3"][1] - res["f32"][1]).norm() / res["f32"][1].norm())
agree = float((res["f16x3"][2] == res["f32"][2]).float().mean())
print(f"f16x3 vs fp32 execution: logits rel-L2 {e_seg:.2e}, regression rel-L2 {e_cont:.2e}, argmax agreement {agree:.6f}")
assert e_seg < 1e-4 and e_cont < 1e-4
print("ok")
