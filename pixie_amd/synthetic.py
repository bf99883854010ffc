"""Seeded synthetic inputs for the hot path.

No datasets or checkpoints exist offline, and the reference's spatial LayerNorm
affine parameters are grid-size-shaped (diffusion_network.py:674,679,870), so the
BASELINE configurations run on synthetic grids, weights and particle scenes
generated here.  Everything is numpy (CPU, deterministic across machines).
"""
from __future__ import annotations

import math

import numpy as np


# --------------------------------------------------------------------------- MPM
def mpm_ball_scene(n_particles: int = 100_000, seed: int = 0, n_grid: int = 50, grid_lim: float = 2.0,
                   dt: float = 1e-4, scenario: str = "tree"):
    """BASELINE config 3/5 particle scene.

    Particles uniform in the ball of radius 0.5 centred (1,1,1) -- the frame the
    reference driver leaves particles in after transform2origin + shift2center111
    (PhysGaussian/utils/transformation_utils.py:6-16,103-105).  Material 0 ("jelly",
    fixed-corotated) with per-particle E = 10^U(5,6.3), nu = U(0.2,0.4), rho = U(200,2000).

    scenario "tree"  mirrors PhysGaussian/config/objaverse/custom_tree_config.json:
        g = 0, grid_v_damping_scale 0.9999, rpic 0, one particle_impulse (-0.48,0,0) for 1 dt,
        plus the fix_to_ground slab (material_field.py:485-550: delta_z 0.05, buffer_xy 0.5).
    scenario "ball"  mirrors custom_sport_balls_config.json: g = (0,0,-9.8), bounding_box.
    """
    rng = np.random.default_rng(seed)
    # rejection-free uniform ball sampling
    d = rng.normal(size=(n_particles, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.5 * rng.random(n_particles) ** (1.0 / 3.0)
    x = (1.0 + d * r[:, None]).astype(np.float32)
    vol = np.full(n_particles, (4.0 / 3.0) * math.pi * 0.5 ** 3 / n_particles, np.float32)
    E = (10.0 ** rng.uniform(5.0, 6.3, n_particles)).astype(np.float32)
    nu = rng.uniform(0.2, 0.4, n_particles).astype(np.float32)
    rho = rng.uniform(200.0, 2000.0, n_particles).astype(np.float32)
    # isotropic initial covariances (6-float upper triangles xx,xy,xz,yy,yz,zz)
    s2 = rng.uniform(1e-5, 4e-5, n_particles).astype(np.float32)
    cov = np.zeros((n_particles, 6), np.float32)
    cov[:, 0] = s2; cov[:, 3] = s2; cov[:, 5] = s2

    scene = dict(x=x, vol=vol, cov=cov, E=E, nu=nu, density=rho, material=np.zeros(n_particles, np.int32),
                 n_grid=n_grid, grid_lim=grid_lim, dt=dt)
    if scenario == "tree":
        scene["params"] = dict(material="jelly", g=[0.0, 0.0, 0.0], grid_v_damping_scale=0.9999, rpic_damping=0.0,
                               E=2e6, nu=0.4, density=200.0)
        scene["bcs"] = [dict(type="particle_impulse", force=[-0.48, 0.0, 0.0], num_dt=1, start_time=0.0)]
        scene["fix_ground"] = dict(delta_z=0.05, buffer_xy=0.5)
    elif scenario == "ball":
        scene["params"] = dict(material="jelly", g=[0.0, 0.0, -9.8], E=4e4, nu=0.4, density=200.0)
        scene["bcs"] = [dict(type="bounding_box")]
        scene["fix_ground"] = None
    elif scenario == "sand":
        # PhysGaussian/config/objaverse/custom_sand_config.json: Drucker-Prager sand dropped on a sticky floor inside the
        # bounding box (n_grid 200, substep 2e-5 in the reference; the caller chooses n_grid / dt)
        scene["params"] = dict(material="sand", g=[0.0, 0.0, -9.8], E=5e7, nu=0.3, density=2000.0, friction_angle=30.0)
        scene["bcs"] = [dict(type="bounding_box"),
                        dict(type="surface_collider", point=[1.0, 1.0, 0.48], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0,
                             start_time=0.0, end_time=1e3)]
        scene["fix_ground"] = None
        scene["E"] = np.full(n_particles, 5e7, np.float32); scene["nu"] = np.full(n_particles, 0.3, np.float32)
        scene["density"] = np.full(n_particles, 2000.0, np.float32)
        scene["material"] = np.full(n_particles, 2, np.int32)
    else:
        raise ValueError(scenario)
    return scene


# The reference's own plastic configurations (PhysGaussian/config/objaverse/custom_{sand,snow,metal}_config.json: material
# parameters, n_grid, substep_dt, gravity, damping, boundary conditions as shipped).
_BBOX = dict(type="bounding_box")
PLASTIC_CONFIGS = {
    "sand": dict(n_grid=200, dt=2e-5, params=dict(material="sand", E=5e7, nu=0.3, density=2000.0, g=[0.0, 0.0, -9.8], friction_angle=30.0),
                 bcs=[_BBOX, dict(type="surface_collider", point=[1.0, 1.0, 0.48], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0,
                                  start_time=0.0, end_time=1e3)]),
    "snow": dict(n_grid=120, dt=1e-5, params=dict(material="snow", E=1e5, yield_stress=5e2, nu=0.2, softening=0.5, grid_v_damping_scale=0.9999,
                                                  density=2700.0, g=[0.0, 0.0, -9.8]), bcs=[_BBOX]),
    "metal": dict(n_grid=120, dt=1e-5, params=dict(material="metal", E=1e8, yield_stress=1e7, nu=0.3, hardening=1, xi=0.1,
                                                   grid_v_damping_scale=0.9999, density=2700.0, g=[0.0, 0.0, -9.8]), bcs=[_BBOX]),
    # what material_mode=neural produces: ONE set of solver scalars (the scene's config) and a per-particle material id from
    # the predicted field (PhysGaussian/material_field.py:343-363).  Ids 0 / 1 / 2 / 5 drawn per particle -- every wave of the
    # fused kernel then holds all four constitutive branches (the worst case for divergence; a real field is piecewise constant).
    "mixed": dict(n_grid=120, dt=1e-5, params=dict(material="jelly", E=2e6, yield_stress=2e4, nu=0.3, hardening=1, xi=0.1, softening=0.1,
                                                   friction_angle=30.0, grid_v_damping_scale=0.9999, density=1500.0, g=[0.0, 0.0, -9.8]), bcs=[_BBOX]),
}


def mpm_plastic_scene(name: str, n_particles: int = 100_000, seed: int = 0):
    """The ball of BASELINE config 3 / 5 under one of PLASTIC_CONFIGS.  A ball released at rest stays rigid (F = I, no stress)
    until it reaches a wall, thousands of substeps away, so the initial state is perturbed to put the return mappings to work from
    the first substep: F_trial = I + 0.02 N(0,1) per entry (0.15 for metal, whose yield strain sigma_y / 2 mu is 0.13) and
    v = (0.3, -0.2, -1.0) + 0.2 N(0,1) m/s -- scene["F0"], scene["v0"], applied by start_plastic().  (n = 100 000, seed 0 is
    the scene of tests/golden/mpm_plastic_*.npz.)"""
    cfg = PLASTIC_CONFIGS[name]
    sc = mpm_ball_scene(n_particles, seed=seed, n_grid=cfg["n_grid"], dt=cfg["dt"], scenario="ball")
    sc["params"] = dict(cfg["params"]); sc["bcs"] = list(cfg["bcs"]); sc["fix_ground"] = None
    rng = np.random.default_rng(100 + len(name))
    amp = 0.15 if name == "metal" else 0.02
    sc["F0"] = (np.eye(3) + amp * rng.normal(size=(n_particles, 3, 3))).astype(np.float32)
    sc["v0"] = (np.array([0.3, -0.2, -1.0]) + 0.2 * rng.normal(size=(n_particles, 3))).astype(np.float32)
    p = cfg["params"]
    sc["E"] = np.full(n_particles, p["E"], np.float32); sc["nu"] = np.full(n_particles, p["nu"], np.float32)
    sc["density"] = np.full(n_particles, p["density"], np.float32)
    if name == "mixed":
        sc["material"] = np.array([0, 1, 2, 5], np.int32)[rng.integers(0, 4, n_particles)]
        sc["per_particle"] = True
    else:
        sc["per_particle"] = False
    return sc


def start_plastic(solver, sc, set_field):
    """apply_scene + the perturbed initial state; set_field(name, array) writes a particle field of `solver`."""
    apply_scene(solver, sc, per_particle=sc.get("per_particle", False))
    set_field("F_trial", sc["F0"]); set_field("v", sc["v0"])


def ground_slab(positions: np.ndarray, delta_z: float = 0.02, buffer_xy: float = 0.5):
    """Cuboid of fix_to_ground (PhysGaussian/material_field.py:485-550), min_z_percentile=1."""
    min_xy = positions[:, :2].min(axis=0)
    max_xy = positions[:, :2].max(axis=0)
    size_xy = max_xy - min_xy
    min_z = positions[:, 2].min()
    center = [float((min_xy[0] + max_xy[0]) / 2), float((min_xy[1] + max_xy[1]) / 2), float(min_z + delta_z / 2)]
    half = [float(size_xy[0] / 2 + buffer_xy), float(size_xy[1] / 2 + buffer_xy), float(delta_z / 2)]
    return dict(point=center, size=half, velocity=[0.0, 0.0, 0.0], start_time=0.0, end_time=1e6, reset=1)


def apply_scene(solver, scene, per_particle: bool = True):
    """Drive any solver exposing the reference's MPM_Simulator_WARP method names
    (the HIP solver or the oracle) through the set-up order of gs_simulation.py:483-531."""
    solver.set_parameters_dict(scene["params"])
    for bc in scene["bcs"]:
        if bc["type"] == "particle_impulse":
            solver.add_impulse_on_particles(force=bc["force"], dt=scene["dt"], point=bc.get("point", [1, 1, 1]),
                                            size=bc.get("size", [1, 1, 1]), num_dt=bc.get("num_dt", 1),
                                            start_time=bc.get("start_time", 0.0))
        elif bc["type"] == "bounding_box":
            solver.add_bounding_box()
        elif bc["type"] == "cuboid":
            solver.set_velocity_on_cuboid(point=bc["point"], size=bc["size"], velocity=bc["velocity"],
                                          start_time=bc.get("start_time", 0.0), end_time=bc.get("end_time", 1e3),
                                          reset=bc.get("reset", 0))
        elif bc["type"] == "surface_collider":
            solver.add_surface_collider(point=bc["point"], normal=bc["normal"], surface=bc["surface"],
                                        friction=bc["friction"], start_time=bc["start_time"], end_time=bc["end_time"])
        elif bc["type"] == "enforce_particle_translation":
            solver.enforce_particle_velocity_translation(point=bc["point"], size=bc["size"], velocity=bc["velocity"],
                                                         start_time=bc["start_time"], end_time=bc["end_time"])
        else:
            raise TypeError("Undefined BC type")
    if scene.get("fix_ground"):
        slab = ground_slab(scene["x"], **scene["fix_ground"])
        solver.set_velocity_on_cuboid(**slab)
    if per_particle:
        solver.set_per_particle(E=scene["E"], nu=scene["nu"], density=scene["density"], material=scene["material"])
    solver.finalize_mu_lam()


# --------------------------------------------------------------------------- the two halves together (BASELINE configs[2])
# Normalisation ranges of the SYNTHETIC scene, in the format of normalization_stats/normalization_ranges.yaml (log10 for density and
# E): the material ranges SURVEY 8d prescribes for the MPM scene (E = 10^[5, 6.3] Pa, nu in [0.2, 0.4], rho = 10^[2.3, 3.3]), so that
# whatever the randomly initialised networks predict is un-scaled into a CFL-safe material (the un-scaling clamps to [-1, 1]); the
# shipped ranges reach E = 10^10.9 Pa, for which dt = 1e-4 is unstable ("smoke test only")
PIPELINE_RANGES = {"density_min": 2.3, "density_max": 3.3, "E_min": 5.0, "E_max": 6.3, "nu_min": 0.2, "nu_max": 0.4}


def pipeline_scene(D: int = 128, C: int = 64, n_particles: int = 100_000, seed: int = 0, n_grid: int = 50):
    """BASELINE configs[2] as ONE scene: the 128^3 x 64 feature grid of configs[1] with its occupancy ball (radius 0.35 D), and the
    100 k-particle ball of the MPM configuration inside the occupied region.  Field frame: [-1, 1]^3 over the grid; simulation
    frame: the ball of radius 0.5 at (1, 1, 1) in [0, 2]^3 -- field = (sim - 1) * 1.3 puts the particle ball (radius 0.65 in field
    units) inside the occupied voxels (radius 0.7).  Materials come from the networks (un-scaled with PIPELINE_RANGES); the solver
    scalars are the jelly defaults of the tree scenario without its impulse / slab (the predicted ids include every material)."""
    sc = mpm_ball_scene(n_particles, seed=seed, n_grid=n_grid, scenario="ball")
    g = (np.arange(D, dtype=np.float32) - (D - 1) / 2.0) / (D / 2.0)
    rr = np.sqrt(g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2)
    sc["mask"] = (rr < 0.7).astype(np.float32)
    sc["feat"] = feature_grid(D, C, seed=100 + seed)
    sc["field_scale"] = 1.3
    sc["min_bounds"], sc["max_bounds"] = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
    sc["params"] = dict(material="jelly", g=[0.0, 0.0, -9.8], E=2e6, nu=0.4, density=200.0, yield_stress=2e4, hardening=1, xi=0.1,
                        softening=0.1, friction_angle=30.0, grid_v_damping_scale=0.9999)
    sc["ranges"] = dict(PIPELINE_RANGES)
    return sc


# --------------------------------------------------------------------------- U-Net
def feature_grid(D: int, C: int = 64, seed: int = 0, occupancy: bool = False) -> np.ndarray:
    """(1,C,D,D,D) float32 feature grid; `occupancy` masks to a ball of radius 0.35 D and
    rounds through fp16 as the reference stores features (pixie/voxel/voxelize.py:86,111)."""
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((1, C, D, D, D), dtype=np.float32)
    if occupancy:
        g = np.arange(D, dtype=np.float32) - (D - 1) / 2.0
        rr = g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2
        feat *= (rr <= (0.35 * D) ** 2)[None, None]
        feat = feat.astype(np.float16).astype(np.float32)
    return feat
