"""Generates tests/golden/unet_full64.npz and unet_full128.npz: the REFERENCE's own modules on the
north-star configuration (BASELINE.json configs[1]: 128^3 x 64 grid, both networks), fp32 on PyTorch CPU --
exactly the input (feature_grid seed 100) and weights (seeds 0 / 1000) bench.py times.

Run in the build container (needs /root/reference; ~1 min at 64^3, ~10 min and ~25 GB at 128^3):
    python tests/golden/make_unet_golden_large.py [64] [128] [--f64]

A full 128^3 output pair is 88 MiB, so the fixture keeps, per head:
  sub      every STRIDE-th voxel per axis starting at OFFSET (26^3 samples at 128^3): pointwise parity
  block    one dense 12^3 block in the interior and one touching the (0,0,0) corner: zero-padding / halo parity
  l2       per-channel L2 norm of the FULL output (float64): catches errors anywhere in the volume
  sum      per-channel sum of the full output (float64)
  hist     (seg only) histogram of argmax over the full volume: class-decision parity at scale
  f64_*    the same quantities in float64 (optional, --f64), from oracle/unet_oracle.py -- the restatement that is pinned
           bit-for-bit to the reference modules in fp32; the reference itself hard-codes float32 in inner_dtype and
           GroupNorm32.  reference-f32 vs f64 is the reference's own rounding error: the noise floor any fp32
           implementation is entitled to
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_unet_golden import HEADS, reference_model  # noqa: E402

from oracle import unet_oracle  # noqa: E402
from pixie_amd.synthetic import feature_grid  # noqa: E402
from pixie_amd.unet_plan import UNetConfig, synthetic_state_dict  # noqa: E402

STRIDE, OFFSET, BLOCK = 5, 2, 12
INPUT_SEED = 100   # bench.py: scene i uses seed 100 + i


def digest(out: np.ndarray, is_seg: bool, prefix: str = ""):
    """out: (1, C, D, D, D)"""
    o = out[0]
    D = o.shape[-1]
    mid = D // 2 - BLOCK // 2
    d = {
        prefix + "sub": np.ascontiguousarray(o[:, OFFSET::STRIDE, OFFSET::STRIDE, OFFSET::STRIDE]).astype(np.float32),
        prefix + "block_mid": np.ascontiguousarray(o[:, mid:mid + BLOCK, mid:mid + BLOCK, mid:mid + BLOCK]).astype(np.float32),
        prefix + "block_corner": np.ascontiguousarray(o[:, :BLOCK, :BLOCK, :BLOCK]).astype(np.float32),
        prefix + "l2": np.sqrt((o.astype(np.float64) ** 2).reshape(o.shape[0], -1).sum(1)),
        prefix + "sum": o.astype(np.float64).reshape(o.shape[0], -1).sum(1),
    }
    if is_seg:
        d[prefix + "hist"] = np.bincount(o.argmax(0).ravel(), minlength=o.shape[0]).astype(np.int64)
    return d


def run(D: int, with_f64: bool):
    kw = dict(feature_channels=64, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4),
              attention_resolutions=(), grid_size=D)
    feat = feature_grid(D, 64, seed=INPUT_SEED)
    res = dict(stride=STRIDE, offset=OFFSET, block=BLOCK, input_seed=INPUT_SEED, grid=D)
    for head, oc, off in HEADS:
        cfg = UNetConfig(out_channels=oc, **kw)
        sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, off).items()}
        model = reference_model(cfg)
        model.load_state_dict(sd, strict=True)
        t0 = time.time()
        with torch.no_grad():
            out = model(torch.from_numpy(feat)).numpy()
        print(f"{D}^3 {head} f32: {time.time() - t0:.1f} s, mean|out| {np.abs(out).mean():.4f}", flush=True)
        for k, v in digest(out, head == "seg").items():
            res[f"{head}_{k}"] = v
        if with_f64:
            t0 = time.time()
            with torch.no_grad():
                out64 = unet_oracle.unet_forward(synthetic_state_dict(cfg, off), cfg, feat, dtype=torch.float64).numpy()
            print(f"{D}^3 {head} f64: {time.time() - t0:.1f} s; reference f32 vs f64 rel-L2 "
                  f"{np.linalg.norm(out - out64) / np.linalg.norm(out64):.3e}", flush=True)
            for k, v in digest(out64, head == "seg", "f64_").items():
                res[f"{head}_{k}"] = v
            res[f"{head}_ref_f32_vs_f64_rel_l2"] = float(np.linalg.norm(out - out64) / np.linalg.norm(out64))
            if head == "seg":
                res["seg_ref_f32_vs_f64_argmax_agreement"] = float((out[0].argmax(0) == out64[0].argmax(0)).mean())
            del out64
        del model, out
    np.savez_compressed(os.path.join(HERE, f"unet_full{D}.npz"), **res)
    print("wrote", f"unet_full{D}.npz", flush=True)


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 128]
    for D in sizes:
        run(D, "--f64" in sys.argv)
