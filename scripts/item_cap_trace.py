#!/usr/bin/env python
"""Which work-item capacity is in force over a run of the sand scene (item_cap "auto"), sampled every 20 substeps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.mpm_solver import MPM_Simulator_WARP
from pixie_amd.synthetic import mpm_plastic_scene, start_plastic
n = 1_000_000
sc = mpm_plastic_scene(os.environ.get("PIXIE_MPM_SCENARIO", "sand"), n, seed=0)
s = MPM_Simulator_WARP(10)
s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]), n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
start_plastic(s, sc, lambda f, a: s.set_field(f, a.reshape(n, -1)))
seq = []
for k in range(30):
    s.run(sc["dt"], 20)
    seq.append((int(s._get_scalar("item_cap")), int(s._get_scalar("n_work_items")), int(s._get_scalar("n_rebins"))))
print(seq)
