"""Phase timeline of the fused MPM block kernel (pixie_mpm_set_scalar "trace"): per work item, 100 MHz timestamps."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pixie_amd._lib as L  # noqa: E402
from pixie_amd.mpm_solver import MPM_Simulator_WARP  # noqa: E402
from pixie_amd.synthetic import apply_scene, mpm_ball_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sc = mpm_ball_scene(n, seed=0, n_grid=ng)
s = MPM_Simulator_WARP(10, diag=True)
s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]),
                               n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
apply_scene(s, sc)
s.run(sc["dt"], 100)
torch.cuda.synchronize()
s._set_scalar("trace", 1)
s.run(sc["dt"], 3)     # the buffer keeps the last substep
torch.cuda.synchronize()
s._set_scalar("trace", 0)
items = int(s._get_scalar("n_work_items"))
rd = C.CDLL(L.DIAG_LIB_PATH)["_ZN5pixie14mpm_trace_readEPyi"]
rd.argtypes = [C.c_void_p, C.c_int]
m = min(items, 32768)
buf = np.zeros(m * 8, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
t = buf.reshape(m, 8).astype(np.int64)
us = (t[:, :6] - t[:, 0].min()) / 100.0
names = ["tile staged (+ particle loads issued)", "G2P + stress + x/F update", "bound reduction", "scatter (LDS atomics)", "publish tile"]
print(f"n={n} ng={ng}: {items} work items; kernel span {us[:, 5].max():.1f} us; work-item lifetime mean {np.mean(us[:, 5] - us[:, 0]):.2f} us "
      f"(p10 {np.percentile(us[:, 5] - us[:, 0], 10):.2f}, p90 {np.percentile(us[:, 5] - us[:, 0], 90):.2f})")
for i, nm in enumerate(names):
    d = us[:, i + 1] - us[:, i]
    print(f"  {nm:42s} {d.mean():6.2f} us  (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})")
start = np.sort(us[:, 0])
print("  start times: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % tuple(np.percentile(start, [10, 50, 90, 100])))

# ---- placement: workgroups per CU and lifetime by co-residency
hw = buf.reshape(m, 8)[:, 6]
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(int); xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
key = xcc * 10000 + se * 100 + sh * 50 + cu
life = us[:, 5] - us[:, 0]
first = us[:, 0] < 2.0           # first dispatch round
uk, cnt = np.unique(key[first], return_counts=True)
print(f"first round: {first.sum()} workgroups on {len(uk)} CUs; workgroups per CU: " + ", ".join(f"{c}: {int((cnt == c).sum())} CUs" for c in np.unique(cnt)))
per = dict(zip(uk, cnt))
co = np.array([per.get(k, 0) for k in key])
for c in np.unique(co[first]):
    sel = first & (co == c)
    print(f"  {c} workgroup(s) on the CU: lifetime mean {life[sel].mean():6.2f} us, max {life[sel].max():6.2f} (n={sel.sum()})")
print(f"  particles per item: mean {n / items:.0f}")
