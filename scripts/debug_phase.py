import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.mpm_oracle import OracleMPM
from pixie_amd.synthetic import apply_scene, mpm_ball_scene
from pixie_amd.mpm_solver import MPM_Simulator_WARP
sc = mpm_ball_scene(20000, seed=1)
n = 20000
rng = np.random.default_rng(0)
v0 = (0.5 * rng.normal(size=(n, 3))).astype(np.float32)
h = MPM_Simulator_WARP(10, diag=True)
h.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]), n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
apply_scene(h, sc)
o = OracleMPM(n, sc["n_grid"], sc["grid_lim"], "f32"); o.load_initial_data(sc["x"], sc["vol"], sc["cov"]); apply_scene(o, sc)
h.set_field("v", v0); o.field("v")[:] = v0
dt = sc["dt"]
o.phase("zero_grid"); o.phase("pre_p2g", dt); o.phase("compute_stress", dt); o.phase("p2g", dt)
h.phase(0, dt)
gm = h.get_field("grid_m").cpu().numpy(); gvi = h.get_field("grid_v_in").cpu().numpy()
print("grid_m err", np.abs(gm - o.field("grid_m")).max(), "grid_v_in err", np.abs(gvi - o.field("grid_v_in")).max())
o.phase("grid_update", dt); o.phase("grid_damping"); o.phase("apply_bcs", dt)
h.phase(1, dt)
gvo = h.get_field("grid_v_out").cpu().numpy(); ref = o.field("grid_v_out")
d = np.abs(gvo - ref).max(axis=3)
print("grid_v_out max err", d.max(), "nonzero ref", (np.abs(ref).max(axis=3) > 0).sum(), "nonzero hip", (np.abs(gvo).max(axis=3) > 0).sum())
bad = np.argwhere(d > 1e-4)
print("bad nodes", len(bad), bad[:10].tolist())
for b in bad[:5]:
    print(b, "hip", gvo[tuple(b)], "ref", ref[tuple(b)], "m", gm[tuple(b)], "mv", gvi[tuple(b)])
print("slow", h._get_scalar("slow_path_particles"), "items", h._get_scalar("n_work_items"))
