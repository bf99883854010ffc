"""Where does the float32 drift of the north-star rollout (config 3: 100 000 particles, n_grid 50, tree scenario) come from?  

Four builds of the SAME C oracle on the same scene, CPU only:
  f64                        the float64 trajectory
  f32                        every operation in float32 (the reference's precision)
  f32 + positions in double  x = float(double accumulation of dt * v); everything else float32       (ORACLE_EXPERIMENT=1)
  f32 + F in double          F_trial = float(double accumulation of (I + dt grad v) F); rest float32   (ORACLE_EXPERIMENT=2)
  f32 + both                                                                                           (ORACLE_EXPERIMENT=3)
and prints each float32 variant's distance from the float64 run (displacement, v, C, F_trial) at the checkpoints.

    python tests/golden/attribute_config3_drift.py [substeps=500] [particles=100000]      (~4 min on 8 cores)
"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle.mpm_oracle as om  # noqa: E402
from pixie_amd.synthetic import apply_scene, mpm_ball_scene  # noqa: E402


def build_variants():
    src = os.path.join(ROOT, "oracle", "mpm_oracle.c")
    out = {}
    for tag, exp in (("f32_x", 1), ("f32_F", 2), ("f32_xF", 3)):
        so = os.path.join(ROOT, "oracle", "build", f"libmpm_oracle_{tag}.so")
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DREAL=float", f"-DORACLE_EXPERIMENT={exp}",
                               src, "-o", so, "-lm"])
        out[tag] = so
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    om.build()
    build_variants()
    sc = mpm_ball_scene(n, seed=0)
    cps = [c for c in (20, 100, 500, 1000) if c <= steps]
    tags = ["f64", "f32", "f32_x", "f32_F", "f32_xF"]
    snaps = {t: {} for t in tags}

    def work(tag):
        o = om.OracleMPM(n, sc["n_grid"], sc["grid_lim"], tag)    # (OracleMPM loads build/libmpm_oracle_<tag>.so; dtype float32 unless the tag starts with f64)
        o.load_initial_data(sc["x"], sc["vol"], sc["cov"])
        apply_scene(o, sc)
        done = 0
        for cp in cps:
            o.run(sc["dt"], cp - done)
            done = cp
            snaps[tag][cp] = {f: np.array(o.field(f), dtype=np.float64) for f in ("x", "v", "C", "F_trial")}
            print(f"{tag}: substep {cp} at {time.time() - t0:.0f} s", flush=True)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(t,)) for t in tags]
    [t.start() for t in th]
    [t.join() for t in th]
    x0 = sc["x"].astype(np.float64)

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    print(f"\nconfig 3, {n} particles: distance of each float32 variant from the float64 run (rel-L2 over all particles)")
    print(f"{'substep':>8} {'variant':>28} {'displacement':>13} {'v':>10} {'C':>10} {'F_trial':>10}")
    names = {"f32": "all float32", "f32_x": "positions in double", "f32_F": "F_trial in double", "f32_xF": "positions + F in double"}
    for cp in cps:
        r = snaps["f64"][cp]
        for t in tags[1:]:
            s = snaps[t][cp]
            print(f"{cp:8d} {names[t]:>28} {rel(s['x'] - x0, r['x'] - x0):13.3e} {rel(s['v'], r['v']):10.3e} {rel(s['C'], r['C']):10.3e} {rel(s['F_trial'], r['F_trial']):10.3e}")


if __name__ == "__main__":
    main()
