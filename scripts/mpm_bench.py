#!/usr/bin/env python
"""MPM-only timing on the GPU box: us/substep, fused-kernel / grid-kernel durations (HIP events on the launch
stream), work items, slow-path particles.  Usage: python scripts/mpm_bench.py N NGRID [SUBSTEPS] [RESORT]
PIXIE_MPM_SCENARIO = tree (default) | ball | sand | snow | metal | mixed (the last four: pixie_amd.synthetic.PLASTIC_CONFIGS; NGRID 0 = the config's)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.mpm_solver import MPM_Simulator_WARP  # noqa: E402
from pixie_amd.synthetic import PLASTIC_CONFIGS, apply_scene, mpm_ball_scene, mpm_plastic_scene, start_plastic  # noqa: E402


def main():
    n = int(sys.argv[1]); ng = int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    resort = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    scenario = os.environ.get("PIXIE_MPM_SCENARIO", "tree")
    plastic = scenario in PLASTIC_CONFIGS   # sand / snow / metal / mixed: the reference's own config (NGRID 0 = its n_grid), perturbed start
    if plastic:
        sc = mpm_plastic_scene(scenario, n, seed=0)
        if ng > 0:
            sc["n_grid"] = ng
        ng = sc["n_grid"]
    else:
        sc = mpm_ball_scene(n, seed=0, n_grid=ng, scenario=scenario, dt=float(os.environ.get("PIXIE_MPM_DT", "1e-4")))
    # PIXIE_MPM_DIAG=1 (implied by PIXIE_MPM_TRACE): the handle lives in libpixie_hip_diag.so -- same sources and kernels + the per-launch
    # event timing and the trace; without it the product library is timed and the per-kernel columns read 0
    diag = bool(os.environ.get("PIXIE_MPM_TRACE")) or os.environ.get("PIXIE_MPM_DIAG", "0") != "0"
    s = MPM_Simulator_WARP(10, diag=diag)
    s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]),
                                   n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
    if plastic:
        start_plastic(s, sc, lambda f, a: s.set_field(f, a.reshape(n, -1)))
    else:
        apply_scene(s, sc)
    if resort > 0:
        s._set_scalar("resort_interval", resort)
    if os.environ.get("PIXIE_MPM_TRACE"):
        s._set_scalar("trace", int(os.environ["PIXIE_MPM_TRACE"], 0))
    if os.environ.get("PIXIE_MPM_OCC"):
        s._set_scalar("occupancy", int(os.environ["PIXIE_MPM_OCC"]))
    if os.environ.get("PIXIE_MPM_ITEM_CAP"):
        s._set_scalar("item_cap", int(os.environ["PIXIE_MPM_ITEM_CAP"]))
    for env, key in (("PIXIE_MPM_BITS", "scatter_bits"), ("PIXIE_MPM_WIDE", "wide"), ("PIXIE_MPM_SPARSE", "sparse_tiles"), ("PIXIE_MPM_GRID_RB", "grid_rb"), ("PIXIE_MPM_XCD", "xcd_order")):
        if os.environ.get(env):
            s._set_scalar(key, int(os.environ[env]))
    if os.environ.get("PIXIE_MPM_V0"):     # a scene in motion: random particle velocities of this rms per component (strains of a few %)
        g = torch.Generator().manual_seed(1)
        s.import_particle_v_from_torch(float(os.environ["PIXIE_MPM_V0"]) * torch.randn((n, 3), generator=g))
    s.run(sc["dt"], int(os.environ.get("PIXIE_MPM_WARM", "64")))
    torch.cuda.synchronize()
    rebins0 = int(s._get_scalar("n_rebins"))
    t0 = time.perf_counter()
    s.run(sc["dt"], steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rebins_timed = int(s._get_scalar("n_rebins")) - rebins0
    p_ms = g_ms = 0.0
    if diag:
        s.set_profile(True)
        s.run(sc["dt"], 128)
        torch.cuda.synchronize()
        p_ms, g_ms, nl = s.kernel_times()
        s.set_profile(False)
    alg = 212.0 * n + 44.0 * ng ** 3
    print(f"n={n} ng={ng} resort={resort} {scenario} lib={'diag' if diag else 'product'} occ={os.environ.get('PIXIE_MPM_OCC', '5')} dbg={os.environ.get('PIXIE_MPM_TRACE', '0')} cap={os.environ.get('PIXIE_MPM_ITEM_CAP', 'auto')}->{int(s._get_scalar('item_cap'))} "
          f"bits={os.environ.get('PIXIE_MPM_BITS', 'dflt')} v0={os.environ.get('PIXIE_MPM_V0', '0')} wide={os.environ.get('PIXIE_MPM_WIDE', 'auto')}: {1e6 * dt / steps:.2f} us/substep  {n * steps / dt:.3e} particle-steps/s  "
          f"alg {alg * steps / dt / 1e9:.1f} GB/s ({alg * steps / dt / 8e12 * 100:.2f}% of 8TB/s) | fused kernel {1e3 * p_ms:.2f} us "
          f"({212.0 * n / (max(p_ms, 1e-9) * 1e-3) / 1e9:.1f} GB/s) grid kernel {1e3 * g_ms:.2f} us | items {int(s._get_scalar('n_work_items'))} "
          f"active blocks {int(s._get_scalar('n_active_blocks'))} rebins {int(s._get_scalar('n_rebins'))} (timed region: {rebins_timed}) slow {int(s._get_scalar('slow_path_particles'))} oob {s.out_of_bounds} "
          f"finite {bool(torch.isfinite(s.get_field('x')).all())}", flush=True)


if __name__ == "__main__":
    main()
