#!/usr/bin/env python
"""Two 1 M-particle scenes on two HIP streams: how the host drives them.  (a) one host thread per scene (bench.py's leg),
(b) ONE host thread alternating `run(dt, k)` between the two solvers (k substeps per call, all launches asynchronous).
Each mode is repeated; per repetition: microseconds per scene-substep."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from pixie_amd.synthetic import mpm_ball_scene  # noqa: E402

n, ng, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
scenes = [mpm_ball_scene(n, seed=10 + i, n_grid=ng) for i in range(2)]
solvers = [bench._mpm_solver(sc) for sc in scenes]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
dt = scenes[0]["dt"]


def threads(nsub):
    def work(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            solvers[i].run(dt, nsub)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()


def alternate(nsub, k):
    done = 0
    while done < nsub:
        c = min(k, nsub - done)
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                solvers[i].run(dt, c)
        done += c
    torch.cuda.synchronize()


def single(nsub):
    with torch.cuda.stream(streams[0]):
        solvers[0].run(dt, nsub)
    torch.cuda.synchronize()


def timed(fn, *a):
    t0 = time.perf_counter()
    fn(*a)
    return 1e6 * (time.perf_counter() - t0) / steps


threads(50)
print(f"{n} particles, n_grid {ng}, {steps} substeps per repetition; us per scene-substep (two scenes: wall / substeps / 1 -> per pair / 2)")
print("one scene alone        :", " ".join(f"{timed(single, steps):7.2f}" for _ in range(reps)), flush=True)
print("two threads            :", " ".join(f"{timed(threads, steps) / 2:7.2f}" for _ in range(reps)), flush=True)
for k in (1, 4, 16, 64):
    print(f"one thread, k = {k:<3d}    :", " ".join(f"{timed(alternate, steps, k) / 2:7.2f}" for _ in range(reps)), flush=True)
print("two threads (again)    :", " ".join(f"{timed(threads, steps) / 2:7.2f}" for _ in range(reps)), flush=True)
print("rebins:", [int(s._get_scalar("n_rebins")) for s in solvers])
