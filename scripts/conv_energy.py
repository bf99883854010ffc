#!/usr/bin/env python
"""Energy per launch of the dominant convolution against the bare MFMA loop -- makes the "runs at the board's power limit"
reading of DESIGN 3.1 falsifiable: a power-limited kernel's duration is (energy per launch) / (power limit),
so what matters is joules per output, not schedule.

Samples the board sensors (sysfs hwmon: socket power, shader clock) at ~100 Hz from a thread while
  (a) conv3d_f16x3_c64_fullres_kernel (64 -> 64, 3^3, 128^3, LayerNorm prologue) runs back to back for ~4 s,
  (b) scripts/microbench/mfma_lds.exe sustain 4: the same tap loop on random fp16 operands with no staging / prologue / epilogue,
  (c) the exact-fp32 variant of the same layer,
and prints power, clock, ms per launch, joules per launch and picojoules per algorithmic / issued MFMA flop."""
import os
import subprocess
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import gpu_telemetry  # noqa: E402
from pixie_amd.unet import ACT_LEAKY, HipOps  # noqa: E402


class Sampler:
    def __init__(self):
        self.rows, self._stop = [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            r = gpu_telemetry(0)
            if r:
                self.rows.append((time.perf_counter(), r.get("power_w"), r.get("sclk_mhz")))
            time.sleep(0.01)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()

    def mean(self, t0, t1):
        sel = [(p, c) for t, p, c in self.rows if t0 + 0.5 <= t <= t1 and p is not None]     # skip the ramp
        if not sel:
            return None, None, 0
        return sum(p for p, _ in sel) / len(sel), sum(c for _, c in sel if c) / max(1, sum(1 for _, c in sel if c)), len(sel)


def conv_leg(ops, prec, seconds=4.0):
    dev = ops.device
    g = torch.Generator().manual_seed(0)
    D, cin, cout = 128, 64, 64
    x = torch.randn((cin, D, D, D), generator=g).to(dev)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) / (cin * 27) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    kw = dict(upsample=False, pro=(torch.ones(cin, device=dev), torch.zeros(cin, device=dev)),
              affine=(torch.ones((D, D, D), device=dev), torch.zeros((D, D, D), device=dev)), act=ACT_LEAKY)
    if prec == "f16x3":
        kw["in_bound"] = 64.0
        kw["w16"] = ops.pack_conv16(w)
        args = ([x], None, b, cout, 3)
    else:
        args = ([x], ops.pack_conv(w), b, cout, 3)
    ops.conv(*args, **kw)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            ops.conv(*args, **kw)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return t0, time.perf_counter(), e0.elapsed_time(e1) / n, 2.0 * 27 * cin * cout * D ** 3


def main():
    ops = HipOps(torch.device("cuda:0"))
    idle = gpu_telemetry(0)
    print(f"idle: {idle}")
    out = []
    with Sampler() as s:
        for prec, issue in (("f16x3", 3), ("f32", 1)):
            t0, t1, ms, flop = conv_leg(ops, prec)
            p, c, k = s.mean(t0, t1)
            out.append((f"conv 64->64 3^3 128^3 [{prec}]", ms, flop, issue, p, c, k))
            time.sleep(1.0)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(REPO, "scripts", "microbench", "mfma_lds.exe"), "sustain", "4"], capture_output=True, text=True, timeout=120)
        t1 = time.perf_counter()
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SUSTAIN")]
        if line:
            tok = line[0].split()
            p, c, k = s.mean(t0 + 0.5, t1)
            out.append(("bare tap loop, random fp16 operands (mfma_lds.exe sustain)", float(tok[4]), float(tok[6]), 1, p, c, k))
        else:
            print("mfma_lds.exe sustain failed:", r.stdout[-300:], r.stderr[-300:])
    for name, ms, flop, issue, p, c, k in out:
        if p is None:
            print(f"{name}: {ms:.4f} ms per launch; no power sensor readable")
            continue
        joule = p * ms * 1e-3
        print(f"{name}: {ms:.4f} ms per launch at {p:.0f} W / {c:.0f} MHz ({k} samples) = {joule:.3f} J per launch; "
              f"{flop / ms / 1e9:.1f} TFLOP/s algorithmic, {joule / flop * 1e12:.3f} pJ per algorithmic flop, "
              f"{joule / (flop * issue) * 1e12:.3f} pJ per flop issued on the matrix pipe (x{issue})")


if __name__ == "__main__":
    main()
