"""Times pixie_attention_forward at the U-Net's shape (C=256, T=16^3) and checks it against a float64 softmax."""
import sys, time, torch
sys.path.insert(0, ".")
from pixie_amd.unet import HipOps

C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ops = HipOps(torch.device("cuda:0"))
g = torch.Generator().manual_seed(1)
qkv = torch.randn((3 * C, T), generator=g)
qkv[:C] *= 2.0
d = qkv.cuda()
out = ops.attention(d, C, T)
torch.cuda.synchronize()
q, k, v = torch.split(qkv.double(), C, dim=0)
s = C ** -0.25
ref = v @ torch.softmax((q * s).t() @ (k * s), dim=-1).t()
err = ((out.cpu().double() - ref).norm() / ref.norm()).item()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ops.attention(d, C, T)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"attention C={C} T={T}: {ms:.3f} ms/launch, {4.0 * T * T * C / ms / 1e9:.1f} TFLOP/s, rel-L2 {err:.2e}")
