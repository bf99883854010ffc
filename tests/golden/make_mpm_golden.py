"""Generates tests/golden/mpm_config3.npz: the C oracle (oracle/mpm_oracle.c, float64 and float32 builds) on the
north-star MPM configuration -- BASELINE.json configs[2]: 100 000 particles, n_grid 50, grid_lim 2, dt 1e-4, the "tree"
scenario (impulse + ground slab), exactly the scene bench.py times (mpm_ball_scene(100_000, seed=0)) -- for
1 000 substeps.

    python tests/golden/make_mpm_golden.py            (~6 min: the two builds run in two threads)

The scalar oracle needs ~0.25 s per substep at this size, i.e. 2 x 4 minutes per run: too much to spend on the GPU box
at every test run, so its trajectory is committed as a fixture instead and tests/test_mpm_hip.py compares the HIP
solver with it.  tests/test_mpm_oracle.py re-runs the first checkpoint live on the CPU, which ties the fixture to the
oracle source in the tree.

Per checkpoint (substeps 20, 100, 500, 1000) the fixture holds, from the float64 run,
  x, v, F_trial, C   of every STRIDE-th particle (caller order), float64
  norms              L2 norms over ALL particles of x - x0, v, C, F_trial - I   (whole-population parity)
  momentum, com      sum(m v), sum(m x) / sum(m) over all particles
and, from the float32 run of the same oracle, `drift_*` / `drift_agg_*`: its distance from the float64 run over all
particles (x, the displacement, v, C, F_trial; the norm ratios, total momentum and centre of mass) -- the rounding-error floor that any float32 implementation (the reference's Warp kernels
included) is entitled to, which the GPU test uses to scale its tolerances.

Round 6 ("say what the config-3 drift is"): the float32 run's own trajectory is stored too -- `x32_`, `v32_`, `C32_`,
`F_trial32_` of the same every-16th particles -- so that the GPU test can measure product-vs-float32-oracle directly instead of inferring it
from two similar distances to the float64 run; and `order_{cp}` = the distance (displacement, v, C, F_trial; all particles) between that
float32 run and a SECOND float32 run of the same algorithm that differs only in the order of the P2G sums (the OpenMP build scatters tile by
tile, 8 colours, instead of particle by particle): how far two legitimate float32 evaluations of the reference's algorithm are from each other.
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

N, SEED, STRIDE = 100_000, 0, 16
CHECKPOINTS = (20, 100, 500, 1000)


def make(scene, precision):
    o = OracleMPM(N, scene["n_grid"], scene["grid_lim"], precision)
    o.load_initial_data(scene["x"], scene["vol"], scene["cov"])
    apply_scene(o, scene)
    return o


def main(out_name="mpm_config3.npz", checkpoints=CHECKPOINTS):
    sc = mpm_ball_scene(N, seed=SEED)
    runs = {p: make(sc, p) for p in ("f64", "f32", "f32_omp")}
    snaps = {p: {} for p in runs}

    def work(p):
        o, done = runs[p], 0
        for cp in checkpoints:
            o.run(sc["dt"], cp - done)
            done = cp
            snaps[p][cp] = {f: np.array(o.field(f), dtype=np.float64) for f in ("x", "v", "F_trial", "C")}
            print(f"{p}: substep {cp} at {time.time() - t0:.0f} s", flush=True)

    t0 = time.time()
    threads = [threading.Thread(target=work, args=(p,)) for p in runs]
    [t.start() for t in threads]
    [t.join() for t in threads]

    mass = np.array(runs["f64"].field("mass"), dtype=np.float64)
    x0 = sc["x"].astype(np.float64)
    eye = np.eye(3)
    res = dict(n=N, seed=SEED, stride=STRIDE, checkpoints=np.array(checkpoints), n_grid=sc["n_grid"], dt=sc["dt"],
               oob=np.array([runs["f64"].out_of_bounds, runs["f32"].out_of_bounds]))

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    for cp in checkpoints:
        s64, s32, s32b = snaps["f64"][cp], snaps["f32"][cp], snaps["f32_omp"][cp]
        for f in ("x", "v", "F_trial", "C"):
            res[f"{f}_{cp}"] = s64[f][::STRIDE]
            res[f"{f}32_{cp}"] = s32[f][::STRIDE].astype(np.float32)
        res[f"order_{cp}"] = np.array([rel(s32b["x"] - x0, s32["x"] - x0), rel(s32b["v"], s32["v"]), rel(s32b["C"], s32["C"]),
                                       rel(s32b["F_trial"], s32["F_trial"])])
        res[f"norms_{cp}"] = np.array([np.linalg.norm(s64["x"] - x0), np.linalg.norm(s64["v"]), np.linalg.norm(s64["C"]),
                                       np.linalg.norm(s64["F_trial"] - eye)])
        res[f"momentum_{cp}"] = (mass[:, None] * s64["v"]).sum(0)
        res[f"com_{cp}"] = (mass[:, None] * s64["x"]).sum(0) / mass.sum()
        res[f"drift_{cp}"] = np.array([rel(s32["x"], s64["x"]), rel(s32["x"] - x0, s64["x"] - x0), rel(s32["v"], s64["v"]),
                                       rel(s32["C"], s64["C"]), rel(s32["F_trial"], s64["F_trial"])])
        # the same float32 run's whole-population aggregates: |norm ratio - 1| (displacement, v, C, F - I), momentum error
        # relative to sum(m) * rms|v|, centre-of-mass error (absolute)
        n32 = np.array([np.linalg.norm(s32["x"] - x0), np.linalg.norm(s32["v"]), np.linalg.norm(s32["C"]), np.linalg.norm(s32["F_trial"] - eye)])
        v_rms = np.linalg.norm(s64["v"]) / np.sqrt(N)
        res[f"drift_agg_{cp}"] = np.concatenate([np.abs(n32 / res[f"norms_{cp}"] - 1.0),
                                                 [np.linalg.norm((mass[:, None] * s32["v"]).sum(0) - res[f"momentum_{cp}"]) / (mass.sum() * v_rms),
                                                  np.abs((mass[:, None] * s32["x"]).sum(0) / mass.sum() - res[f"com_{cp}"]).max()]])
        print(cp, "norms", res[f"norms_{cp}"], "drift (x, disp, v, C, F)", res[f"drift_{cp}"], "aggregates", res[f"drift_agg_{cp}"],
              "two float32 summation orders apart (disp, v, C, F)", res[f"order_{cp}"], flush=True)
    np.savez_compressed(os.path.join(HERE, out_name), **res)
    print("wrote", out_name, f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
