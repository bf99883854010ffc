#!/usr/bin/env python
"""Writes tests/golden/mpm_ref_golden.npz: MPM rollouts computed by the REFERENCE'S OWN SOLVER CODE.

third_party/PhysGaussian/mpm_solver_warp/{mpm_solver_warp,mpm_utils,warp_utils,engine_utils}.py are imported UNMODIFIED
(sys.path points into /root/reference) on top of tests/golden/wp_shim -- a numpy interpreter of the Warp 0.10.1 API
subset those files use (Warp itself is not installed and has no ROCm back end).  `MPM_Simulator_WARP` then runs as
written: its set-up methods, its BC / modifier closures, `p2g2p` in the reference's launch order, every @wp.kernel and
@wp.func of mpm_utils.py executed once per thread.  The interpreter reproduces Warp's typing (float32 literals, float32
struct members and launch scalars, column constructor of mat33, truncating wp.int) and evaluates kernel expressions in
float64 on that float32 problem data -- the rule oracle/mpm_oracle.c's float64 build follows -- so the fixture pins the
oracle to ~1e-12.  A second pass in float32 arithmetic records how far the reference's own code drifts from its float64
self in single precision (`drift/...`), the yardstick for float32 implementations.

The one built-in that is a stand-in is `wp.svd3` (Warp's native svd.h is not in the reference tree): LAPACK canonicalised
to Warp's convention (U, V proper rotations, sign of det on the last singular value).  The scene `inverted` starts a third
of its particles with det F < 0 and is stored under both conventions (`inverted` / `inverted_lapack`), so the tests can
show what depends on the convention and that the oracle follows Warp's.

Run in the build container (needs /root/reference):   python tests/golden/make_mpm_ref_golden.py      (~2 min)

`--long` writes tests/golden/mpm_ref_long_golden.npz instead: two ROLLOUTS of 150 substeps by the same reference code (the tree
scenario of PhysGaussian/config/objaverse/custom_tree_config.json on a ball with per-particle E / nu / density, and a sand
column falling onto a sticky floor with the flags of custom_sand_config.json), checkpoints at 50 / 100 / 150 substeps, 20^3
grid, 600 / 500 particles -- how the restatements track the reference's own code over a rollout rather than a few substeps;
a metal column likewise; and `knife_edge_floor`, three substeps of a scene whose collider plane coincides with a node plane, stored
in BOTH evaluations because they differ (see long_scenes).  (~35 min: every kernel thread is interpreted.)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/third_party/PhysGaussian/mpm_solver_warp"
sys.path[:0] = [os.path.join(HERE, "wp_shim"), REF, ROOT]

import warp as wp  # noqa: E402  (the interpreter)
import mpm_solver_warp as ref  # noqa: E402  (the reference's module, unmodified)

from tests._mpm_ref_driver import ReferenceAdapter, STATE_FIELDS, run  # noqa: E402

assert ref.__file__.startswith("/root/reference/"), ref.__file__
assert wp.__file__.startswith(HERE), wp.__file__

N_GRID, GRID_LIM, DT = 16, 2.0, 1e-4
CHECKPOINTS = (1, 3, 6)


def particles(seed, n, lo, hi, spread=0.15, v=0.8, c=3.0, inverted=0.0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    vol = np.full(n, (hi - lo) ** 3 / n, np.float32)
    A = rng.normal(size=(n, 3, 3)) * 3e-3
    S = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    Ft = np.eye(3) + spread * rng.normal(size=(n, 3, 3))
    if inverted:
        k = int(n * inverted)
        Ft[:k] = Ft[:k] @ np.diag([1.0, 1.0, -1.0])           # reflected: det F < 0
        Ft[: k // 3, :, 1] *= 0.02                            # some of them nearly rank-deficient as well
    return dict(x0=x, vol=vol, cov=cov, v0=(v * rng.normal(size=(n, 3))).astype(np.float32),
                C0=(c * rng.normal(size=(n, 3, 3))).astype(np.float32), Ft0=Ft.astype(np.float32)), rng


def base(**kw):
    p = dict(n_grid=N_GRID, grid_lim=GRID_LIM)
    p.update(kw)
    return p


def scenes():
    out = {}

    # 1. elastic, APIC, gravity, grid damping, bounding box (low walls), impulse window, per-particle E / nu / density
    arr, rng = particles(1, 200, 0.35, 1.25)
    arr.update(E=(10 ** rng.uniform(5.0, 6.3, 200)).astype(np.float32), nu=rng.uniform(0.2, 0.4, 200).astype(np.float32),
               density=rng.uniform(200, 2000, 200).astype(np.float32))
    out["jelly_apic"] = (dict(params=base(material="jelly", E=2e5, nu=0.3, density=1000.0, g=[0.0, 0.0, -9.8], grid_v_damping_scale=0.999),
                              calls=[["add_bounding_box", {}],
                                     ["add_impulse_on_particles", dict(force=[-0.48, 0.1, 0.2], dt=DT, point=[0.9, 0.9, 0.9], size=[0.3, 0.4, 0.5],
                                                                       num_dt=2, start_time=0.0)]]), arr)

    # 2. RPIC blend, bounding box (high walls), a moving cuboid that ends inside the run (reset branch), sticky floor
    arr, rng = particles(2, 200, 0.8, 1.7)
    out["jelly_rpic"] = (dict(params=base(material="jelly", E=4e5, nu=0.35, density=800.0, g=[0.0, 0.0, -9.8], rpic_damping=0.4),
                              calls=[["add_bounding_box", {}],
                                     ["add_surface_collider", dict(point=[1.0, 1.0, 0.95], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0,
                                                                   start_time=0.0, end_time=999.0)],
                                     ["set_velocity_on_cuboid", dict(point=[1.2, 1.2, 1.3], size=[0.6, 0.6, 0.12], velocity=[0.5, 0.0, 0.2],
                                                                     start_time=0.0, end_time=2.5e-4, reset=1)]]), arr)

    # 3. PIC (rpic_damping < -0.001), slip / cut / frictional surfaces (the reference zeroes the velocity for all of them,
    #    :821-840), one of them only active from the second substep on, a cuboid without reset
    arr, rng = particles(3, 200, 0.5, 1.5)
    out["jelly_pic"] = (dict(params=base(material="jelly", E=1e5, nu=0.3, density=500.0, g=[0.0, 0.0, 0.0], rpic_damping=-1.0),
                             calls=[["add_surface_collider", dict(point=[1.0, 1.0, 0.7], normal=[0.3, 0.2, 1.0], surface="slip", friction=0.0,
                                                                  start_time=0.0, end_time=999.0)],
                                    ["add_surface_collider", dict(point=[0.6, 1.0, 1.0], normal=[1.0, 0.0, 0.0], surface="cut", friction=0.0,
                                                                  start_time=1.5e-4, end_time=999.0)],
                                    ["add_surface_collider", dict(point=[1.0, 1.42, 1.0], normal=[0.0, -1.0, 0.0], surface="wall", friction=0.5,
                                                                  start_time=0.0, end_time=4.5e-4)],
                                    ["set_velocity_on_cuboid", dict(point=[1.3, 0.7, 1.3], size=[0.2, 0.2, 0.2], velocity=[0.0, 0.3, 0.0],
                                                                    start_time=0.5e-4, end_time=3.5e-4, reset=0)]]), arr)

    # 4. every material id in one scene through additional_material_params boxes (ids 1, 2, 3, 5, 6; the rest stays 0),
    #    hardening, xi, softening, plastic viscosity, friction angle; velocity-pin modifier with a time window
    arr, rng = particles(4, 240, 0.5, 1.5, spread=0.2)
    arr["yield_stress"] = (10 ** rng.uniform(1.0, 3.7, 240)).astype(np.float32)
    boxes = [dict(point=[0.75, 0.75, 0.75], size=[0.25, 0.25, 0.25], E=8e5, nu=0.3, density=2700.0, material=1),
             dict(point=[1.25, 0.75, 0.75], size=[0.25, 0.25, 0.25], E=5e5, nu=0.25, density=2000.0, material=2),
             dict(point=[0.75, 1.25, 0.75], size=[0.25, 0.25, 0.25], E=3e5, nu=0.3, density=1200.0, material=3),
             dict(point=[1.25, 1.25, 0.75], size=[0.25, 0.25, 0.25], E=2e5, nu=0.2, density=400.0, material="snow"),
             dict(point=[1.0, 1.0, 1.3], size=[0.5, 0.2, 0.2], E=1e5, nu=0.3, density=1000.0, material="stationary")]
    out["mixed_materials"] = (dict(params=base(material="jelly", E=2e5, nu=0.3, density=1000.0, g=[0.0, 0.0, -9.8], yield_stress=2e3,
                                               hardening=1, xi=0.3, softening=300.0, plastic_viscosity=5.0, friction_angle=35.0,
                                               additional_material_params=boxes),
                                   calls=[["add_bounding_box", {}],
                                          ["enforce_particle_velocity_translation", dict(point=[1.0, 1.0, 0.6], size=[0.3, 0.3, 0.1], velocity=[0.0, 0.2, 0.0],
                                                                                          start_time=0.5e-4, end_time=4.5e-4)]],
                                   bulk=True), arr)

    # 5-7. the three plastic materials by NAME with the flags of PhysGaussian/config/objaverse/custom_{sand,snow,metal}_config.json
    arr, rng = particles(5, 200, 0.5, 1.5, spread=0.12)
    arr["Ft0"][:60] = (np.eye(3) * 1.08 + 0.02 * rng.normal(size=(60, 3, 3))).astype(np.float32)     # expansion: tr > 0
    arr["Ft0"][60:120] = (np.eye(3) * 0.93 + 0.02 * rng.normal(size=(60, 3, 3))).astype(np.float32)  # compression
    out["sand"] = (dict(params=base(material="sand", E=5e5, nu=0.3, density=2000.0, g=[0.0, 0.0, -9.8], friction_angle=30.0),
                        calls=[["add_bounding_box", {}],
                               ["add_surface_collider", dict(point=[1.0, 1.0, 0.6], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0,
                                                             start_time=0.0, end_time=1e3)]]), arr)
    arr, rng = particles(6, 200, 0.5, 1.5, spread=0.12)
    arr["yield_stress"] = (10 ** rng.uniform(0.5, 3.5, 200)).astype(np.float32)
    out["snow"] = (dict(params=base(material="snow", E=2e5, nu=0.2, density=400.0, g=[0.0, 0.0, -9.8], yield_stress=1e3, softening=500.0,
                                    grid_v_damping_scale=0.9995),
                        calls=[["add_bounding_box", {}]]), arr)
    arr, rng = particles(7, 200, 0.5, 1.5, spread=0.12)
    out["metal"] = (dict(params=base(material="metal", E=2e6, nu=0.3, density=2700.0, g=[0.0, 0.0, -9.8], yield_stress=8e4, hardening=1, xi=0.2),
                         calls=[["add_bounding_box", {}]]), arr)

    # 8. material 6 with a bulk modulus (finalize_mu_lam_bulk): the weakly compressible branch
    arr, rng = particles(8, 200, 0.5, 1.5, spread=0.05)
    out["water"] = (dict(params=base(material="stationary", E=1e5, nu=0.3, density=1000.0, g=[0.0, 0.0, -9.8]),
                         calls=[["add_bounding_box", {}]], bulk=True), arr)

    # 9. cylinder rotation modifier + release_particles_sequentially (50 velocity pins with staggered end times)
    arr, rng = particles(9, 200, 0.55, 1.45)
    out["rotation_release"] = (dict(params=base(material="jelly", E=2e5, nu=0.3, density=1000.0, g=[0.0, 0.0, 0.0]),
                                    calls=[["enforce_particle_velocity_rotation", dict(point=[1.0, 1.0, 1.0], normal=[0.0, 0.0, 1.0],
                                                                                       half_height_and_radius=[0.3, 0.35], rotation_scale=2.0,
                                                                                       translation_scale=0.1, start_time=0.0, end_time=3.5e-4)],
                                           ["release_particles_sequentially", dict(normal=[0, 0, 1], start_position=0.7, end_position=1.3,
                                                                                   num_layers=50, start_time=0.0, end_time=0.0173)]]), arr)

    # 10. inverted elements (stored under both SVD conventions) in the materials whose reference code survives them: the
    #     Drucker-Prager stress takes log(sigma_3) (mpm_utils.py:75-84) and the water law pow(J, -1.1) (:24) -- both NaN for
    #     det F < 0 in the reference itself, and one NaN stress poisons the grid -- so ids 2 and 6 are left out here
    arr, rng = particles(10, 240, 0.5, 1.5, spread=0.15, inverted=1 / 3)
    arr["material"] = np.tile(np.array([0, 1, 3, 5], np.int32), 60)
    arr["yield_stress"] = (10 ** rng.uniform(2.0, 4.0, 240)).astype(np.float32)
    out["inverted"] = (dict(params=base(material="jelly", E=3e5, nu=0.3, density=1000.0, g=[0.0, 0.0, 0.0], yield_stress=1e3, friction_angle=30.0,
                                        plastic_viscosity=2.0),
                            calls=[], bulk=True), arr)
    for sc, _ in out.values():
        sc.update(n_grid=N_GRID, grid_lim=GRID_LIM, dt=DT, checkpoints=list(CHECKPOINTS))
    return out


def long_scenes():
    out = {}
    n_grid = 20
    # the tree scenario (BASELINE configs[2]; gs_simulation.py with custom_tree_config.json): zero gravity, grid damping 0.9999,
    # one impulse on a box of particles for one substep, a sticky ground slab, bounding box; per-particle material fields as
    # apply_material_field_to_simulation leaves them
    rng = np.random.default_rng(21)
    n = 600
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    x = (1.0 + 0.42 * d * rng.uniform(0, 1, (n, 1)) ** (1 / 3)).astype(np.float32)
    vol = np.full(n, 4 / 3 * np.pi * 0.42 ** 3 / n, np.float32)
    arr = dict(x0=x, vol=vol, cov=np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32), (n, 1)),
               E=(10 ** rng.uniform(5.0, 6.3, n)).astype(np.float32), nu=rng.uniform(0.2, 0.4, n).astype(np.float32),
               density=rng.uniform(200, 2000, n).astype(np.float32),
               v0=(0.6 * rng.normal(size=(n, 3))).astype(np.float32))          # (the config's impulse alone moves the ball by 1e-4 of a cell)
    out["tree_rollout"] = (dict(params=dict(n_grid=n_grid, grid_lim=GRID_LIM, material="jelly", E=2e5, nu=0.3, density=1000.0, g=[0.0, 0.0, 0.0],
                                            grid_v_damping_scale=0.9999, rpic_damping=0.0),
                                calls=[["add_bounding_box", {}],
                                       ["set_velocity_on_cuboid", dict(point=[1.0, 1.0, 0.62], size=[1.0, 1.0, 0.06], velocity=[0.0, 0.0, 0.0],
                                                                       start_time=0.0, end_time=1e3, reset=0)],
                                       ["add_impulse_on_particles", dict(force=[-0.48, 0.0, 0.0], dt=DT, point=[1.0, 1.0, 1.25], size=[0.5, 0.5, 0.2],
                                                                         num_dt=1, start_time=0.0)]]), arr)
    # a column falling onto a sticky floor under gravity: sand with the flags of custom_sand_config.json (friction_angle 30, sticky
    # surface collider, bounding box), and metal (von Mises with hardening).  The floor plane lies BETWEEN two node planes (6.3 dx).
    def column(seed, z0):
        rng = np.random.default_rng(seed)
        n = 500
        x = np.stack([rng.uniform(0.8, 1.2, n), rng.uniform(0.8, 1.2, n), rng.uniform(z0, 1.3, n)], 1).astype(np.float32)
        return dict(x0=x, vol=np.full(n, 0.4 * 0.4 * (1.3 - z0) / n, np.float32), cov=np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32), (n, 1)),
                    v0=(np.array([0.0, 0.0, -1.5]) + 0.3 * rng.normal(size=(n, 3))).astype(np.float32))

    def floor(z):
        return [["add_bounding_box", {}],
                ["add_surface_collider", dict(point=[1.0, 1.0, z], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0, start_time=0.0, end_time=1e3)]]
    out["sand_rollout"] = (dict(params=dict(n_grid=n_grid, grid_lim=GRID_LIM, material="sand", E=5e5, nu=0.3, density=2000.0, g=[0.0, 0.0, -9.8],
                                            friction_angle=30.0), calls=floor(0.63)), column(22, 0.66))
    out["metal_rollout"] = (dict(params=dict(n_grid=n_grid, grid_lim=GRID_LIM, material="metal", E=2e6, nu=0.3, density=2700.0, g=[0.0, 0.0, -9.8],
                                             yield_stress=2e3, hardening=1, xi=0.2), calls=floor(0.63)), column(23, 0.66))
    # The knife edge: a collider plane that COINCIDES with a node plane (plane z = 0.6, dx = 0.1f).  Whether those nodes are "below"
    # it (mpm_solver_warp.py:821-840: dot(x_node - point, normal) < 0) is decided by the last bit of float(k) * dx - point:
    # evaluated in float32 with every operation rounded, 6 * 0.1f rounds to exactly 0.6f and the node plane is NOT in the collider;
    # evaluated exactly on the same float32 data (float64 here; a fused multiply-add in a float32 build, which a compiler is free
    # to contract `float(k) * dx - p` into) 6 * 0.1f - 0.6f = -1.5e-8 and it IS.  The two evaluations of the reference's
    # own source differ by 16 % in v after ONE substep; which one a float32 build gives is the compiler's contraction choice, not
    # the source's.  This scene stores both (`k*_f32/...`) so the tests can say which side an implementation is on.
    # (custom_sand_config.json's own floor, z = 0.48 with dx = 0.01f, is NOT such a case: 48 * 0.01f equals 0.48f exactly.)
    out["knife_edge_floor"] = (dict(params=dict(n_grid=n_grid, grid_lim=GRID_LIM, material="sand", E=5e5, nu=0.3, density=2000.0, g=[0.0, 0.0, -9.8],
                                                friction_angle=30.0), calls=floor(0.6), store_f32=True, checkpoints=[1, 3]), column(24, 0.62))
    for sc, _ in out.values():
        sc.update(n_grid=n_grid, grid_lim=GRID_LIM, dt=DT, checkpoints=sc.get("checkpoints", [50, 100, 150]))
    return out


def rollout(scene, arrays, precision, svd="warp"):
    wp.set_precision(precision)
    wp.set_svd_convention(svd)
    ad = ReferenceAdapter(ref, scene, arrays)
    snaps = {}
    del wp.LAUNCH_LOG[:]
    run(ad, scene, arrays, lambda cp, st: snaps.__setitem__(cp, st))
    extra = {k: ad.read(k) for k in ("grid_m", "grid_v_in", "grid_v_out", "mass", "material")}
    extra["cov_out"], extra["R_out"] = ad.exports()
    extra["time"] = np.float64(ad.time)
    extra["alpha"] = np.float64(ad.s.mpm_model.alpha)
    for k, prm in enumerate(ad.s.particle_velocity_modifier_params[:1]):     # host-side axis set-up of the rotation modifier
        if prm.horizontal_axis_1 is not None:
            extra["mod0_axes"] = np.stack([np.array(prm.normal.a, np.float64), np.array(prm.horizontal_axis_1.a, np.float64),
                                           np.array(prm.horizontal_axis_2.a, np.float64)])
    names = [n for n, _ in wp.LAUNCH_LOG]
    first = names.index("zero_grid")
    second = names.index("zero_grid", first + 1) if names.count("zero_grid") > 1 else len(names)
    return snaps, extra, names[first:second]


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    t0 = time.time()
    store, meta = {}, {}
    long_run = "--long" in sys.argv
    for name, (scene, arrays) in (long_scenes() if long_run else scenes()).items():
        variants = [(name, "warp")] + ([(name + "_lapack", "lapack")] if name == "inverted" else [])
        for vname, svd in variants:
            s64, e64, order = rollout(scene, arrays, "f64", svd)
            s32, e32, _ = rollout(scene, arrays, "f32", svd)
            meta[vname] = dict(scene, svd=svd, substep_launch_order=order)
            for k, v in arrays.items():
                store[f"{vname}/in/{k}"] = v
            for cp in scene["checkpoints"]:
                for f in STATE_FIELDS:
                    store[f"{vname}/k{cp}/{f}"] = s64[cp][f]
                    if scene.get("store_f32"):
                        store[f"{vname}/k{cp}_f32/{f}"] = s32[cp][f]
                store[f"{vname}/drift/k{cp}"] = np.array([rel(s32[cp][f], s64[cp][f]) for f in STATE_FIELDS])
                store[f"{vname}/drift_dx/k{cp}"] = np.float64(rel(s32[cp]["x"] - arrays["x0"], s64[cp]["x"] - arrays["x0"]))
            for k, v in e64.items():
                store[f"{vname}/{k}"] = v
            worst = max(store[f"{vname}/drift/k{scene['checkpoints'][-1]}"])
            print(f"{vname:18s} launches/substep {len(order):3d}  f32-vs-f64 worst field drift {worst:.2e}   [{time.time() - t0:.0f} s]", flush=True)
    store["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "mpm_ref_long_golden.npz" if long_run else "mpm_ref_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
