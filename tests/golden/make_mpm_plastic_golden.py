"""Generates tests/golden/mpm_plastic_{sand,snow,metal,mixed}.npz: the C oracle (float64 and float32 builds) on the reference's own
plastic-material configurations -- third_party/PhysGaussian/config/objaverse/custom_{sand,snow,metal}_config.json: material
parameters, n_grid, substep_dt, gravity, damping and boundary conditions as shipped -- with 100 000 particles for 200 substeps; `mixed` is the mixed-material scene bench.py times (pixie_amd.synthetic.PLASTIC_CONFIGS:
material ids 0 / 1 / 2 / 5 drawn per particle, one set of solver scalars).

    python tests/golden/make_mpm_plastic_golden.py        (~3 min: six oracle runs in six threads)

Scene: the 100 000-particle ball of BASELINE config 3 (radius 0.5 at (1,1,1), grid_lim 2).  A ball released at rest stays
rigid (F = I, no stress) until it reaches a wall, thousands of substeps away, so the initial state is perturbed to put the
return mappings to work from the first substep: F_trial = I + 0.02 N(0,1) per entry (0.15 for metal, whose yield strain is
0.13) and v = (0.3, -0.2, -1.0) + 0.2 N(0,1) m/s
(seeded).  Checkpoints 50 and 200; contents as tests/golden/make_mpm_golden.py (every 16th particle's x, v, F, yield stress;
whole-population norms; the float32 oracle's drift).
"""
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.mpm_oracle import OracleMPM  # noqa: E402
from pixie_amd.synthetic import apply_scene, mpm_ball_scene  # noqa: E402

from pixie_amd.synthetic import PLASTIC_CONFIGS, mpm_plastic_scene, start_plastic  # noqa: E402

N, SEED, STRIDE = 100_000, 0, 16
CHECKPOINTS = (50, 200)
CONFIGS = {k: PLASTIC_CONFIGS[k] for k in ("sand", "snow", "metal", "mixed")}   # `python make_mpm_plastic_golden.py mixed` regenerates one


def plastic_scene(name):
    return mpm_plastic_scene(name, N, SEED)


def start(solver, sc, set_field):
    start_plastic(solver, sc, set_field)


def main():
    results = {}
    t0 = time.time()

    def work(name, prec):
        sc = plastic_scene(name)
        o = OracleMPM(N, sc["n_grid"], sc["grid_lim"], prec)
        o.load_initial_data(sc["x"], sc["vol"], sc["cov"])
        start(o, sc, lambda f, a: o.field(f).__setitem__(slice(None), a))
        done, snaps = 0, {}
        for cp in CHECKPOINTS:
            o.run(sc["dt"], cp - done); done = cp
            snaps[cp] = {f: np.array(o.field(f), dtype=np.float64) for f in ("x", "v", "F", "yield_stress")}
            print(f"{name} {prec}: substep {cp} at {time.time() - t0:.0f} s", flush=True)
        # how much of the population is yielding at the end: one more return mapping (compute_stress_from_F_trial) and the
        # share of particles whose F it moved away from F_trial
        o.phase("compute_stress", sc["dt"])
        moved = np.linalg.norm((o.field("F") - o.field("F_trial")).reshape(N, 9), axis=1) > 1e-6
        ys_end = snaps[CHECKPOINTS[-1]]["yield_stress"]
        ever = float((ys_end != np.float64(np.float32(sc["params"].get("yield_stress", 0.0)))).mean())   # hardening / softening moved it
        results[(name, prec)] = (snaps, o.out_of_bounds, max(float(moved.mean()), ever))

    names = [a for a in sys.argv[1:] if a in CONFIGS] or list(CONFIGS)
    threads = [threading.Thread(target=work, args=(n, p)) for n in names for p in ("f64", "f32")]
    [t.start() for t in threads]; [t.join() for t in threads]
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    for name in names:
        s64, oob64, yielding = results[(name, "f64")]; s32 = results[(name, "f32")][0]
        res = dict(n=N, seed=SEED, stride=STRIDE, checkpoints=np.array(CHECKPOINTS), oob=oob64, yielded_fraction=yielding)
        x0 = plastic_scene(name)["x"].astype(np.float64)
        for cp in CHECKPOINTS:
            for f in ("x", "v", "F", "yield_stress"):
                res[f"{f}_{cp}"] = s64[cp][f][::STRIDE]
            res[f"norms_{cp}"] = np.array([np.linalg.norm(s64[cp]["x"] - x0), np.linalg.norm(s64[cp]["v"]), np.linalg.norm(s64[cp]["F"] - np.eye(3))])
            res[f"drift_{cp}"] = np.array([rel(s32[cp]["x"], s64[cp]["x"]), rel(s32[cp]["x"] - x0, s64[cp]["x"] - x0), rel(s32[cp]["v"], s64[cp]["v"]),
                                           rel(s32[cp]["F"], s64[cp]["F"]), rel(s32[cp]["yield_stress"], s64[cp]["yield_stress"])])
            print(name, cp, "drift (x, disp, v, F, ys)", res[f"drift_{cp}"], flush=True)
        np.savez_compressed(os.path.join(HERE, f"mpm_plastic_{name}.npz"), **res)
        print(name, "share of particles that yielded (still yielding at the end, or whose yield stress was moved):", yielding)
    print("done", f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
