"""Phase timeline of the f16x3 conv kernel (conv_dbg bit 2): per workgroup, 100 MHz timestamps around staging / MFMA / epilogue."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.unet import ACT_LEAKY, HipOps  # noqa: E402

dev = torch.device("cuda:0")
ops = HipOps(dev)
g = torch.Generator().manual_seed(0)
cin, cout, D = int(os.environ.get("CONV_CIN", "64")), int(os.environ.get("CONV_COUT", "64")), 128
K = int(os.environ.get("CONV_K", "3"))
x = torch.randn((cin, D, D, D), generator=g).to(dev)
w = (torch.randn((cout, cin, K, K, K), generator=g) / (cin * K ** 3) ** 0.5).to(dev)
b = torch.randn(cout, generator=g).to(dev)
pro = (torch.ones(cin, device=dev), torch.zeros(cin, device=dev))
aff = (torch.ones((D, D, D), device=dev), torch.zeros((D, D, D), device=dev))
kw = dict(pro=pro, affine=aff, act=ACT_LEAKY, in_bound=64.0, w16=ops.pack_conv16(w))
extra = int(os.environ.get("PIXIE_CONV_DBG", "0"))
for _ in range(2):
    ops.conv([x], None, b, cout, K, **kw)
ops.lib.pixie_set_option(b"conv_dbg", 4 | extra)
ops.conv([x], None, b, cout, K, **kw)
torch.cuda.synchronize()
ops.lib.pixie_set_option(b"conv_dbg", 0)
n_wg, words = 4096, 16
buf = np.zeros(n_wg * words, dtype=np.uint64)
rd = ops.lib._lib["_ZN5pixie15conv_trace_readEPyi"] if hasattr(ops.lib, "_lib") else None
if rd is None:
    import pixie_amd._lib as L
    rd = C.CDLL(L.LIB_PATH)["_ZN5pixie15conv_trace_readEPyi"]
rd.argtypes = [C.c_void_p, C.c_int]
assert rd(buf.ctypes.data, buf.size) == 0
t = buf.reshape(n_wg, words).astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :15] - t0) / 100.0
nst = t[:, 13]
print("stamps per WG:", np.unique(nst))
start, end = us[:, 0], us[:, 14]
print(f"kernel span {end.max():.1f} us; WG lifetime mean {np.mean(end - start):.1f} us (min {np.min(end - start):.1f}, max {np.max(end - start):.1f})")
nch = (int(nst[0]) - 1) // 2
for c in range(nch):
    prev = us[:, 2 * c] if c > 0 else us[:, 0]
    st = us[:, 1 + 2 * c] - prev
    mf = us[:, 2 + 2 * c] - us[:, 1 + 2 * c]
    print(f"chunk {c}: stage {st.mean():6.2f} us (p10 {np.percentile(st, 10):6.2f}, p90 {np.percentile(st, 90):6.2f})   mfma {mf.mean():6.2f} us (p10 {np.percentile(mf, 10):6.2f}, p90 {np.percentile(mf, 90):6.2f})")
ep = us[:, 14] - us[:, 2 * nch]
print(f"epilogue {ep.mean():6.2f} us (p10 {np.percentile(ep, 10):6.2f}, p90 {np.percentile(ep, 90):6.2f})")
hw = buf.reshape(n_wg, words)[:, 15]
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(int)
slot = (hw & np.uint64(0xF)).astype(int)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
key = xcc * 10000 + se * 100 + sh * 50 + cu
print("distinct CUs seen:", len(np.unique(key)), " wave slots used:", np.unique(slot))
k0 = key[0]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(start[idx])]
print(f"timeline of CU key {k0} ({len(idx)} workgroups): S = stage end, M = mfma end")
for i in idx[:10]:
    row = " ".join(f"{v:7.1f}" for v in us[i, :2 * nch + 1]) + f" | end {us[i, 14]:7.1f}"
    print(f"  wg {i:4d} slot {slot[i]}: {row}")

# ---- interference: duration of each phase vs what the co-resident workgroup (other wave slot of the same CU) was doing
import collections
byslot = collections.defaultdict(list)
for i in range(n_wg):
    byslot[(key[i], slot[i])].append(i)
def phases(i):
    out = []
    prev = us[i, 0]
    for c in range(nch):
        out.append(("S", prev, us[i, 1 + 2 * c])); out.append(("M", us[i, 1 + 2 * c], us[i, 2 + 2 * c])); prev = us[i, 2 + 2 * c]
    out.append(("E", prev, us[i, 14]))
    return out
stats = collections.defaultdict(list)
for (k, sl), lst in byslot.items():
    partner = sorted((phases(j) for j in byslot.get((k, 1 - sl), [])), key=lambda p: p[0][1])
    pint = [ph for pl in partner for ph in pl]
    for i in lst:
        for name, a, b in phases(i):
            if b <= a:
                continue
            ov = collections.Counter()
            for pn, pa_, pb_ in pint:
                o = min(b, pb_) - max(a, pa_)
                if o > 0:
                    ov[pn] += o
            tot = b - a
            dom = max(("S", "M", "E", "idle"), key=lambda n: ov[n] if n != "idle" else tot - sum(ov.values()))
            frac = (ov[dom] if dom != "idle" else tot - sum(ov.values())) / tot
            if frac > 0.7:
                stats[(name, dom)].append(tot)
print("phase duration by what the co-resident workgroup was doing for > 70 % of it:")
for (name, dom), v in sorted(stats.items()):
    print(f"  {name} while partner {dom:4s}: n={len(v):5d}  mean {np.mean(v):6.2f} us  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}")
