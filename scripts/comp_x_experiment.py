#!/usr/bin/env python
"""Round-5 experiment: does carrying the rounding error of x += dt v (set_scalar "compensated_x") move the
north-star scene -- BASELINE configs[2], 100 k particles, 1000 substeps -- from the float32 oracle's drift towards the float64
oracle, and what does it cost?  Compares both modes with the committed float64 trajectory (tests/golden/mpm_config3.npz) on the scales
of tests/test_mpm_hip.py::test_rollout_parity_config3, and times 100 k / 1 M particles in both modes."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pixie_amd.mpm_solver import MPM_Simulator_WARP  # noqa: E402
from pixie_amd.synthetic import apply_scene, mpm_ball_scene  # noqa: E402


def solver(sc, comp):
    s = MPM_Simulator_WARP(10)
    s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]), n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
    apply_scene(s, sc)
    s._set_scalar("compensated_x", comp)
    return s


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


g = np.load(os.path.join(REPO, "tests", "golden", "mpm_config3.npz"))
n, stride = int(g["n"]), int(g["stride"])
sc = mpm_ball_scene(n, seed=int(g["seed"]))
x0 = sc["x"].astype(np.float64)
inv_dx = sc["n_grid"] / sc["grid_lim"]
for comp in (0, 1):
    h = solver(sc, comp)
    done = 0
    for cp in [int(c) for c in g["checkpoints"]]:
        h.run(sc["dt"], cp - done); done = cp
        x, v, C, F = (h.get_field(f).cpu().numpy().astype(np.float64) for f in ("x", "v", "C", "F_trial"))
        m = n // stride + (1 if n % stride else 0)
        ns = np.sqrt(m)
        v_rms = float(np.linalg.norm(g[f"v_{cp}"]) / ns)
        c_scale = max(v_rms * inv_dx, float(np.linalg.norm(g[f"C_{cp}"]) / ns))
        d_x, d_disp, d_v, d_C, d_F = (float(t) for t in g[f"drift_{cp}"])
        dv = d_v * float(g[f"norms_{cp}"][1]) / np.sqrt(n) / v_rms
        dC = d_C * float(g[f"norms_{cp}"][2]) / np.sqrt(n) / c_scale
        e_disp = rel((x - x0)[::stride], g[f"x_{cp}"] - x0[::stride])
        e_v = float(np.linalg.norm(v[::stride] - g[f"v_{cp}"]) / ns) / v_rms
        e_C = float(np.linalg.norm(C.reshape(-1, 3, 3)[::stride] - g[f"C_{cp}"]) / ns) / c_scale
        e_F = rel(F.reshape(-1, 3, 3)[::stride], g[f"F_trial_{cp}"])
        print(f"compensated_x={comp} substep {cp}: displacement {e_disp:.2e} (f32 oracle {d_disp:.2e}), v {e_v:.2e} ({dv:.2e}), C {e_C:.2e} ({dC:.2e}), "
              f"F_trial {e_F:.2e} ({d_F:.2e}), x {rel(x[::stride], g[f'x_{cp}']):.2e}", flush=True)
for npart, ng, steps in ((100_000, 50, 3000), (1_000_000, 120, 600)):
    scb = mpm_ball_scene(npart, seed=0, n_grid=ng)
    for rep in range(2):
        for comp in (0, 1):
            s = solver(scb, comp)
            s.run(scb["dt"], 100)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.run(scb["dt"], steps)
            torch.cuda.synchronize()
            print(f"timing n={npart} compensated_x={comp} (repetition {rep}): {1e6 * (time.perf_counter() - t0) / steps:.2f} us/substep", flush=True)
            del s
