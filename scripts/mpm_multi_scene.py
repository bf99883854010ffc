#!/usr/bin/env python
"""N independent MPM scenes on N HIP streams of one GPU (bench.py's multi-scene leg on its own)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

n, ng, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scenes = [int(v) for v in sys.argv[4].split(",")]
dev = torch.device("cuda", 0)
for k in scenes:
    r = bench.bench_mpm_multi_scene(types.SimpleNamespace(), dev, n, ng, steps, k)
    print(f"{k} scenes x {n} particles (n_grid {ng}): {r['value']:.4e} particle-steps/s, {r['us_per_substep_per_scene']:.2f} us per substep of the batch, finite {r['finite']}", flush=True)
