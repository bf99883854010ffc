"""Timeline of the MPM grid kernel (diag library, set_scalar "trace" 4): per active block four 100 MHz stamps -- start, neighbour row arrived,
tiles summed (masks + tile loads), node values stored.  Usage: python scripts/mpm_grid_trace.py N NGRID [scenario]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pixie_amd._lib as L  # noqa: E402
from pixie_amd.mpm_solver import MPM_Simulator_WARP  # noqa: E402
from pixie_amd.synthetic import PLASTIC_CONFIGS, apply_scene, mpm_ball_scene, mpm_plastic_scene, start_plastic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 120
scenario = sys.argv[3] if len(sys.argv) > 3 else "tree"
if scenario in PLASTIC_CONFIGS:
    sc = mpm_plastic_scene(scenario, n, seed=0)
else:
    sc = mpm_ball_scene(n, seed=0, n_grid=ng)
s = MPM_Simulator_WARP(10, diag=True)
s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]), n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
if scenario in PLASTIC_CONFIGS:
    start_plastic(s, sc, lambda f, a: s.set_field(f, a.reshape(n, -1)))
else:
    apply_scene(s, sc)
s.run(sc["dt"], 300)
torch.cuda.synchronize()
s._set_scalar("trace", 4)
s.run(sc["dt"], 3)     # the buffer keeps the last substep's grid launch
torch.cuda.synchronize()
s._set_scalar("trace", 0)
blocks = int(s._get_scalar("n_active_blocks"))
rd = C.CDLL(L.DIAG_LIB_PATH)["_ZN5pixie14mpm_trace_readEPyi"]
rd.argtypes = [C.c_void_p, C.c_int]
m = min(blocks, 32768)
buf = np.zeros(m * 8, dtype=np.uint64)
assert rd(buf.ctypes.data, buf.size) == 0
t = buf.reshape(m, 8).astype(np.int64)
us = (t[:, :4] - t[:, 0].min()) / 100.0
life = us[:, 3] - us[:, 0]
print(f"{scenario} n={n} ng={sc['n_grid']}: {blocks} active blocks (one wave each); kernel span {us[:, 3].max():.2f} us; wave lifetime mean {life.mean():.2f} us "
      f"(p10 {np.percentile(life, 10):.2f}, p50 {np.percentile(life, 50):.2f}, p90 {np.percentile(life, 90):.2f})")
for i, nm in enumerate(["neighbour row (1 load)", "gather: masks, then tiles", "flag, finish_node, store"]):
    d = us[:, i + 1] - us[:, i]
    print(f"  {nm:28s} {d.mean():6.2f} us  (p10 {np.percentile(d, 10):5.2f}, p50 {np.percentile(d, 50):5.2f}, p90 {np.percentile(d, 90):5.2f})")
st = us[:, 0]
print("  wave start times: p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % tuple(np.percentile(st, [10, 50, 90, 100])))
print("  wave end times:   p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % tuple(np.percentile(us[:, 3], [10, 50, 90, 100])))
hw = buf.reshape(m, 8)[:, 6]
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(int); xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
key = xcc * 10000 + se * 100 + sh * 50 + cu
uk, cnt = np.unique(key, return_counts=True)
print(f"  {len(uk)} CUs used; waves per CU over the launch: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
# concurrency: how many waves are alive at once
ev = np.concatenate([np.stack([us[:, 0], np.ones(m)], 1), np.stack([us[:, 3], -np.ones(m)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
print(f"  waves alive at once: max {int(alive.max())}, time-average {float((alive[:-1] * np.diff(ev[:, 0])).sum() / max(ev[-1, 0] - ev[0, 0], 1e-9)):.0f}")
