#!/usr/bin/env python
"""sha256 of the particle state (x, v, C, F_trial) after a rollout: two builds of the library that claim bit-identical arithmetic
must print the same line.  Usage: python scripts/mpm_state_hash.py N NGRID SUBSTEPS [scatter_bits]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.mpm_solver import MPM_Simulator_WARP  # noqa: E402
from pixie_amd.synthetic import apply_scene, mpm_ball_scene  # noqa: E402

n, ng, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc = mpm_ball_scene(n, seed=0, n_grid=ng)
s = MPM_Simulator_WARP(10)
s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]), n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
apply_scene(s, sc)
if len(sys.argv) > 4:
    s._set_scalar("scatter_bits", int(sys.argv[4]))
if os.environ.get("PIXIE_MPM_XCD"):
    s._set_scalar("xcd_order", int(os.environ["PIXIE_MPM_XCD"]))
g = torch.Generator().manual_seed(1)
s.import_particle_v_from_torch(0.6 * torch.randn((n, 3), generator=g))
s.run(sc["dt"], steps)
h = hashlib.sha256()
for f in ("x", "v", "C", "F_trial"):
    h.update(s.get_field(f).cpu().numpy().tobytes())
print(f"xcd_order={os.environ.get('PIXIE_MPM_XCD', 'dflt')} n={n} ng={ng} substeps={steps} bits={sys.argv[4] if len(sys.argv) > 4 else 'dflt'} rebins={int(s._get_scalar('n_rebins'))} sha256={h.hexdigest()[:32]}")
