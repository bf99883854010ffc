#!/usr/bin/env python
"""Per-layer-shape timing of pixie_conv3d_forward on the GPU box, both precisions (HIP events on the launch stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixie_amd.unet import ACT_LEAKY, HipOps  # noqa: E402

SHAPES = [  # (cin parts, cout, D, ksize, upsample, prologue)
    ((64,), 64, 128, 3, False, True),
    ((64, 64), 64, 128, 3, False, True),
    ((128,), 128, 128, 3, False, True),
    ((64,), 128, 128, 1, False, False),
    ((128,), 32, 128, 1, False, True),
    ((64, 64), 64, 128, 1, False, False),
    ((64,), 8, 128, 3, False, True),
    ((64,), 64, 64, 3, True, False),
    ((64,), 64, 64, 3, False, True),
    ((128,), 128, 32, 3, False, True),
    ((256,), 256, 16, 3, False, True),
    ((256, 256), 256, 16, 3, False, True),
]


def main():
    dev = torch.device("cuda:0")
    ops = HipOps(dev)
    g = torch.Generator().manual_seed(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    nshapes = int(os.environ.get("PIXIE_CONV_NSHAPES", len(SHAPES)))
    for cins, cout, D, k, ups, prologue in SHAPES[:nshapes]:
        cin = sum(cins)
        parts = [torch.randn((c, D, D, D), generator=g).to(dev) for c in cins]
        w = (torch.randn((cout, cin, k, k, k), generator=g) / (cin * k ** 3) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        OD = 2 * D if ups else D
        pro = (torch.ones(cin, device=dev), torch.zeros(cin, device=dev)) if prologue else None
        affine = (torch.ones((D, D, D), device=dev), torch.zeros((D, D, D), device=dev)) if prologue else None
        flop = 2.0 * k ** 3 * cin * cout * OD ** 3
        res = {}
        for prec in ("f32", "f16x3"):
            kw = dict(upsample=ups, pro=pro, affine=affine, act=ACT_LEAKY if prologue else 0)
            if prec == "f16x3":
                if prologue:
                    kw["in_bound"] = 64.0
                else:
                    slots = torch.zeros(len(parts), dtype=torch.int32, device=dev)
                    for i, t in enumerate(parts):
                        ops.channel_stats(t, slots[i:i + 1])
                    kw["in_amax"] = [slots[i:i + 1] for i in range(len(parts))]
                args = (parts, None, b, cout, k)
                kw["w16"] = ops.pack_conv16(w)
            else:
                args = (parts, ops.pack_conv(w), b, cout, k)
            out = ops.conv(*args, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                out = ops.conv(*args, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[prec] = (e0.elapsed_time(e1) / reps, out)
        err = float((res["f16x3"][1] - res["f32"][1]).norm() / res["f32"][1].norm())
        print(f"cin={cins} cout={cout} D={D} k={k} ups={int(ups)}: f32 {res['f32'][0]:8.3f} ms ({flop / res['f32'][0] / 1e9:7.1f} TF)  "
              f"f16x3 {res['f16x3'][0]:8.3f} ms ({flop / res['f16x3'][0] / 1e9:7.1f} TF)  speedup {res['f32'][0] / res['f16x3'][0]:.2f}x  "
              f"rel diff {err:.2e}", flush=True)
        del parts, out, res


if __name__ == "__main__":
    main()
