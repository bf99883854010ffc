#!/usr/bin/env python
"""Benchmark of the material_mode=neural hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one scene through the U-Net stage of the hot path: SegmentationUNet + RegressionUNet
forward on a device-resident 128^3 x 64 feature grid (BASELINE.json configs[1]) + argmax/one-hot
combine, and (N > 1) the all-gather of the compact fields.  `value` = voxels/s over all ranks
(weak scaling: one scene per rank per step).  The MPM half of the metric (BASELINE configs[2]:
100k particles, n_grid 50) is timed in the same run and reported under "mpm" with its own roofline.
Synthetic data and seeded random-init weights (no datasets/checkpoints exist offline).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from pixie_amd import distributed as pd  # noqa: E402
from pixie_amd.synthetic import PLASTIC_CONFIGS, apply_scene, feature_grid, mpm_ball_scene, mpm_plastic_scene, start_plastic  # noqa: E402
from pixie_amd.unet_plan import UNetConfig, conv_flops, synthetic_state_dict  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def sustained_f16_mfma_random_tflops():
    """What this kernel's tap loop sustains, bare, on random fp16 mantissas (B from LDS + A from global, 2 workgroups per
    CU): parsed from the committed raw output of scripts/microbench/mfma_lds.exe under profiles/ (the newest file wins)."""
    import glob
    import re
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*mfma_lds_microbench*.txt"))):
        for line in open(path):
            m = re.match(r"RANDOM data: B from LDS \+ A from global, 2 WG/CU\s+[\d.]+ ms\s+([\d.]+) TFLOP/s", line)
            if m:
                best = (float(m.group(1)), os.path.relpath(path, REPO))
    return best or (None, None)

PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_f16 dense peak (no sparsity)
PEAK_HBM_GBPS = 8000.0         # HBM3E spec (6.3 TB/s achievable per the same guide)


def gpu_telemetry(index=0):
    """Shader clock (MHz), socket power (W) and temperature of THIS process's GPU, read from sysfs (no subprocess: the
    readings bracket the timed region).  The conv kernels run power-limited, so a bench line without these cannot be
    compared across boxes.  The box may expose several cards in sysfs while the process sees one: the card is found by the
    PCI address torch reports for the device; failing that, the card drawing the most power.  Returns {} where nothing is readable."""
    import glob

    def read_card(hw):
        out = {}
        for key, fname, scale in (("sclk_mhz", "freq1_input", 1e-6), ("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6),
                                  ("temp_c", "temp1_input", 1e-3)):
            if key not in out:
                try:
                    with open(os.path.join(hw, fname)) as f:
                        out[key] = round(float(f.read().strip()) * scale, 1)
                except (OSError, ValueError):
                    pass
        return out
    try:
        want = None
        try:
            p = torch.cuda.get_device_properties(index)
            want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:
            pass
        best, best_power = {}, -1.0
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            addr = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(hw))))
            r = read_card(hw)
            if not r:
                continue
            r["pci"] = addr
            if want and addr == want:
                r["matched_by"] = "pci address"
                return r
            if r.get("power_w", 0.0) > best_power:
                best, best_power = r, r.get("power_w", 0.0)
        if best:
            best["matched_by"] = "highest power draw"
        return best
    except Exception:
        return {}


def cpu_quota():
    """CPUs this container may actually use: the cgroup CPU quota (cpu.max / cfs_quota_us) when one is set, else None.  A box can
    show 256 logical CPUs to os.cpu_count() and still schedule the container on a handful: more threads than that run slower."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=128, help="U-Net grid size D (BASELINE config: 128)")
    ap.add_argument("--feature-channels", type=int, default=64)
    ap.add_argument("--particles", type=int, default=100_000)
    ap.add_argument("--n-grid", type=int, default=50)
    ap.add_argument("--mpm-substeps", type=int, default=1000)
    ap.add_argument("--mpm-large-substeps", type=int, default=2000, help="substeps of the 1M-particle leg (BASELINE configs[4]: 2k)")
    ap.add_argument("--full", action="store_true",
                    help="every leg.  Default: headline U-Net (+ its strict-fp32 twin), MPM 100 k (+ floor, 3-scene batch), MPM 1 M (+ two scenes), the "
                         "reference's sand configuration at 1 M, the configs[2] pipeline, the CPU baselines.  --full adds: the other scatter mode at both "
                         "sizes, the 1 M scene in motion, snow / metal / mixed at 1 M, six scenes per GPU, the 64^3 x 768 shipped shape, U-Net 256^3 x 128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-small", action="store_true", help="time the U-Net CPU baseline at 64^3 instead of the headline grid (saves ~1 min)")
    ap.add_argument("--no-mpm", action="store_true")
    ap.add_argument("--no-shipped-shape", action="store_true", help="skip the 64^3 x 768 fp16-grid sub-record")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the exact-fp32-MFMA sub-record of the same U-Net step")
    ap.add_argument("--no-mpm-large", action="store_true", help="skip the 1M-particle / n_grid 120 MPM leg")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the end-to-end BASELINE configs[2] scene (U-Net -> field transfer -> MPM rollout)")
    ap.add_argument("--no-mpm-plastic", action="store_true", help="skip the sand / snow / metal / mixed-material 1M-particle legs")
    ap.add_argument("--mpm-plastic-substeps", type=int, default=400)
    ap.add_argument("--no-unet-256", action="store_true", help="skip the 256^3 x 128 U-Net sub-record (BASELINE configs[4]'s per-GPU grid)")
    ap.add_argument("--dual-stream-diagnostic", action="store_true",
                    help="also time one scene's launches with the two networks on two HIP streams (roofline.avg_launch_ms_dual_stream)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="torch.distributed backend for --gpus N > 1 (default nccl = RCCL over xGMI; gloo only for --launcher-selftest)")
    ap.add_argument("--dry-run", action="store_true",
                    help="the whole N-rank bench path -- self-launch, rendezvous, per-rank set-up, the timing protocol of every leg, the field "
                         "all-gather, the JSON assembly -- on CPU tensors with gloo and no kernels (what an 8-GPU run does around its kernels)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU-only check of the multi-process path (self-launch, rendezvous, barrier, max-over-ranks timing, "
                         "field all-gather) on a tiny synthetic field; runs no kernel and reports no throughput")
    ap.add_argument("--conv-precision", choices=["f16x3", "f32"], default=None,
                    help="f16x3 (default): fp32 operands split into fp16 hi+lo, 3 f16 MFMAs per product, fp32 accumulate; "
                         "f32: exact-fp32 MFMA everywhere")
    return ap.parse_args()


def barrier_sync(world):
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(step, n_steps, world, device):
    """The timing protocol of every leg: barrier + device sync, n_steps steps, barrier + device sync, MAX over ranks."""
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    barrier_sync(world)
    return max_over_ranks(time.perf_counter() - t0, world, device)


def time_allgather(cont_pred, seg_pred, world, device, reps=5):
    """The collective alone (ms per all-gather of one step's fields, max over ranks): lets a 1 -> 8 GPU curve be split into
    compute and exchange.  None on one rank."""
    if world == 1:
        return None
    pd.all_gather_fields(cont_pred, seg_pred)
    return 1e3 * timed_steps(lambda: pd.all_gather_fields(cont_pred, seg_pred), reps, world, device) / reps


def pin_rank_resources(rank, world):
    """N ranks on one host: give each its share of the cores (torch.distributed.run sets OMP_NUM_THREADS=1 when it is unset,
    which would make the CPU-side set-up of every rank single-threaded) and keep the ranks off each other's cores."""
    cores = os.cpu_count() or 1
    if world <= 1:
        q = cpu_quota()
        if q and q < torch.get_num_threads():
            torch.set_num_threads(max(1, int(round(q))))     # the container's CPU quota, not the host's core count
        return torch.get_num_threads()      # what the CPU legs actually run on
    share = max(1, cores // world)
    try:
        avail = sorted(os.sched_getaffinity(0))
        mine = avail[rank * len(avail) // world:(rank + 1) * len(avail) // world]
        if mine:
            os.sched_setaffinity(0, mine)
            share = len(mine)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(share)
    return share


def max_over_ranks(seconds: float, world: int, device) -> float:
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


class ConvProfiler:
    """HIP-event timing of every conv launch on the stream it is launched on (torch's current stream),
    recorded live inside the timed region; aggregated per layer shape afterwards."""

    def __init__(self):
        self.records = []
        self.variants = []   # (ksize*100 + MB*10 + NB, split-K slices) of each launch: rocprofv3's grouping key

    def wrap(self, ops):
        inner = ops.conv
        prof = self

        def conv(parts, packed_w, bias, cout, ksize, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(parts, packed_w, bias, cout, ksize, **kw)
            e1.record()
            cin = sum(int(p.shape[0]) for p in parts)
            o = out[0] if isinstance(out, tuple) else out   # (output, epilogue channel sums) on the fused-statistics path
            folded = sum(int(t.shape[0]) for t in kw["skip"]["parts"]) if kw.get("skip") else 0   # input channels of a folded 1x1x1 skip conv
            prof.records.append(((cin, cout, ksize, kw.get("stride", 1), bool(kw.get("upsample", False)), tuple(o.shape[1:]), folded), e0, e1))
            prof.variants.append(ops.last_variant)
            return out

        ops.record_variant = True
        ops.conv = conv

    def by_variant(self):
        """mean launch duration per kernel instantiation -- what `rocprofv3 --kernel-trace --stats` reports per kernel name
        (the event pair also brackets the small split-K reduce / statistics launches that follow some convs)"""
        torch.cuda.synchronize()
        agg = {}
        for (key, e0, e1), var in zip(self.records, self.variants):
            a = agg.setdefault(var, [0.0, 0])
            a[0] += e0.elapsed_time(e1); a[1] += 1
        def name(v, sl):
            if v == 9324:
                return "conv3d_f16x3_c64_fullres_kernel"
            if not v:
                return "conv3d_exact_kernel (exact fp32: the tiled body on v_mfma_f32_32x32x2_f32)"
            return f"conv3d_f16x3_kernel<{v // 100},{(v // 10) % 10},{v % 10}>" + (f" x{sl} slices" if sl > 1 else "")
        return {name(v, sl):
                {"launches": n, "avg_ms": round(t / n, 4)} for (v, sl), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(key, [0.0, 0])
            a[0] += ms; a[1] += 1
        return agg


def bench_unet(args, rank, world, device, precision_override=None, steps=None, warmup=None, device_input=False):
    from pixie_amd.unet import RegressionUNet, SegmentationUNet, predict_material_field
    D, C = args.grid, args.feature_channels
    n_steps = args.steps if steps is None else steps
    n_warm = args.warmup if warmup is None else warmup
    kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4),
              attention_resolutions=(), grid_size=D)
    seg = SegmentationUNet(num_classes=8, **kw)
    cont = RegressionUNet(out_channels=3, **kw)
    seg.load_numpy_state(synthetic_state_dict(seg.cfg, 0))
    cont.load_numpy_state(synthetic_state_dict(cont.cfg, 1000))
    seg, cont = seg.to(device).eval(), cont.to(device).eval()
    if precision_override or args.conv_precision:
        seg.conv_precision = cont.conv_precision = precision_override or args.conv_precision
    precision = seg.conv_precision
    if device_input:   # large grids: seeded on the device (numpy needs ~30 s for the 2.1e9 normals of a 256^3 x 128 grid)
        feat = torch.randn((1, C, D, D, D), generator=torch.Generator(device=device).manual_seed(100 + rank), device=device)
    else:
        feat = torch.from_numpy(feature_grid(D, C, seed=100 + rank)).to(device)  # scene i uses seed 100+i

    def step(dual_stream=None):
        combined, seg_pred, _, cont_pred = predict_material_field(seg, cont, feat, dual_stream=dual_stream)
        if world > 1:
            pd.all_gather_fields(cont_pred, seg_pred)
        return combined

    for _ in range(3):   # set-up, not warm-up steps: pack the weights, take the normalisation bounds, record the HIP graphs, and
        step()           # replay them twice (the first replays of a freshly instantiated graph run ~8 % slower: 90 vs 84 ms)
    for _ in range(n_warm):
        step()
    executor = f"{seg.executor}{'+hip_graph' if seg.use_graph else ''}"
    tele0 = gpu_telemetry(device.index or 0)
    tele_mid = {}

    def step_and_sample(_n=[0]):
        step()
        _n[0] += 1
        if _n[0] == n_steps:
            tele_mid.update(gpu_telemetry(device.index or 0))    # while the last steps are still running on the device
    dt = timed_steps(step_and_sample, n_steps, world, device)
    ag_ms = None
    if world > 1:
        _, sp_, _, cp_ = predict_material_field(seg, cont, feat)
        ag_ms = time_allgather(cp_, sp_, world, device)
    # where a step's time goes: events around each network's forward (one graph replay + the copy of its output) and the
    # combine, three more scenes right after the timed region
    parts = {"seg_forward_ms": 0.0, "cont_forward_ms": 0.0, "combine_and_stack_ms": 0.0}
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(3):
        evs[0].record()
        a = seg(feat)
        evs[1].record()
        b = cont(feat)
        evs[2].record()
        seg._runner.ops.combine(a[0].contiguous(), b[0].contiguous())
        evs[3].record()
        torch.cuda.synchronize()
        parts["seg_forward_ms"] += evs[0].elapsed_time(evs[1]) / 3
        parts["cont_forward_ms"] += evs[1].elapsed_time(evs[2]) / 3
        parts["combine_and_stack_ms"] += evs[2].elapsed_time(evs[3]) / 3
    parts = {k: round(v, 3) for k, v in parts.items()}
    # the one kernel of the field exchange (13 B/voxel wire buffer), timed on the device whatever the world size: with N ranks the
    # all-gather follows it; with one rank nothing is packed in the step itself
    try:
        _, sp_, _, cp_ = predict_material_field(seg, cont, feat)
        pd.pack_fields(cp_, sp_)
        evs[0].record()
        for _ in range(10):
            pd.pack_fields(cp_, sp_)
        evs[1].record()
        torch.cuda.synchronize()
        parts["pack_fields_us"] = round(1e3 * evs[0].elapsed_time(evs[1]) / 10, 2)
    except Exception as exc:      # (never loses the run)
        parts["pack_fields_us"] = None
        print(f"bench.py: pack_fields timing failed: {exc}", file=sys.stderr)
    # what a caller that hands over HOST buffers pays on top (the reference's DataLoader does: inference_combined.py:247-256 with
    # pin_memory=True, then `.to(rank)`): one upload of the fp32 feature grid.  Never part of `value` (inputs are resident when the
    # timed region starts); reported so that the PCIe-inclusive rate can be derived.
    try:
        host = torch.empty(feat.shape, dtype=feat.dtype, pin_memory=True)
        dst = torch.empty_like(feat)
        dst.copy_(host, non_blocking=True)
        evs[0].record()
        for _ in range(3):
            dst.copy_(host, non_blocking=True)
        evs[1].record()
        torch.cuda.synchronize()
        parts["h2d_feature_grid_ms_pinned"] = round(evs[0].elapsed_time(evs[1]) / 3, 2)
        del host, dst
    except Exception as exc:
        parts["h2d_feature_grid_ms_pinned"] = None
        print(f"bench.py: H2D timing failed: {exc}", file=sys.stderr)
    parts["telemetry_before"] = tele0
    parts["telemetry_during"] = tele_mid
    flops_scene = conv_flops(seg.cfg) + conv_flops(cont.cfg)
    # dominant kernel: the full-resolution 64->64 3x3x3 conv (82 % of FLOPs are at full resolution).  The timed region above is
    # the product default: each network is ONE pixie_unet_forward call replayed as a captured HIP graph, the two networks back
    # to back on one stream -- no per-launch events can be placed inside a replay.  The launch durations the roofline is computed
    # from come from scenes run right after it through the Python plan walk (the same kernels on the same stream, one foreign
    # call per launch, HIP events around each): two scenes on ONE stream (`avg_launch_ms`, the roofline's duration; rocprofv3's
    # per-kernel mean over the whole run, graph replays included, is the cross-check).  --dual-stream-diagnostic adds one scene
    # with the networks on two streams (`avg_launch_ms_dual_stream`: an event pair then also brackets the other network's kernels).
    for net in (seg, cont):
        net.executor, net.use_graph = "python", False
    step(dual_stream=False)                      # the plan walk packs its own copy of the weights
    prof = ConvProfiler()
    prof.wrap(seg._runner.ops)   # one HipOps instance per network
    if cont._runner.ops is not seg._runner.ops:
        prof.wrap(cont._runner.ops)
    agg_timed, kernel_avg_timed = {}, None
    if args.dual_stream_diagnostic:   # off by default: its overlapped launches would pollute rocprofv3's per-kernel means of this command
        step(dual_stream=True)
        torch.cuda.synchronize()
        agg_timed = prof.summary()
        kernel_avg_timed = prof.by_variant()
        prof.records.clear(); prof.variants.clear()
    for _ in range(2):
        step(dual_stream=False)
    agg = prof.summary()
    dom_key = (64, 64, 3, 1, False, (D, D, D), 0)   # (c_in, c_out, ksize, stride, upsample, output dims, folded skip channels)
    roof = None
    if dom_key in agg:
        ms = agg[dom_key][0] / agg[dom_key][1]
        fl = 2.0 * 27 * 64 * 64 * D ** 3
        ach = fl / (ms * 1e-3) / 1e12
        if precision == "f16x3":
            # achieved = ALGORITHMIC (fp32-equivalent) FLOP/s; the kernel issues 3 f16 MFMAs per algorithmic product,
            # so the matrix pipe is doing 3x that.  peak = dense f16 MFMA peak; frac = achieved/peak (conservative).
            roof = {"bound": "mfma", "kernel": "conv3d_f16x3_c64_fullres_kernel (= conv3d_f16x3_kernel<3,2,4>; 64->64 3^3 conv, %d^3)" % D, "achieved": round(ach, 2),
                    "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4),
                    # the PMC reading of THIS layer shape (profiles/pmc_traffic.json key conv_64_64_<D>); None where no pass was taken
                    "traffic": (load_traffic().get(f"conv_64_64_{D}") or {}).get("hbm_bytes_per_launch"),
                    "traffic_source": "profiles/pmc_traffic.json" if load_traffic().get(f"conv_64_64_{D}") else None,
                    "avg_launch_ms": round(ms, 4), "launches": agg[dom_key][1], "flop_per_launch": fl,
                    "avg_launch_ms_dual_stream": (round(agg_timed[dom_key][0] / agg_timed[dom_key][1], 4) if dom_key in agg_timed else None),
                    "launch_timing": "HIP events on the launch stream around every launch of the Python plan walk; single-stream pass of 2 scenes right after the timed region (the timed region replays HIP graphs; see bench.py)",
                    "mfma_issue_ratio": 3, "mfma_hw_tflops": round(3 * ach, 1), "mfma_hw_frac": round(3 * ach / PEAK_F16_MFMA_TFLOPS, 4),
                    "vs_exact_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 3),
                    # scripts/microbench/mfma_lds.hip: this kernel's tap loop, bare, on random fp16 operands (the nominal peak
                    # is only approached on constant operands; the matrix cores are power-limited once the multipliers toggle)
                    "mfma_sustained_random_operands_tflops": sustained_f16_mfma_random_tflops()[0],
                    "mfma_sustained_source": sustained_f16_mfma_random_tflops()[1],
                    "mfma_hw_frac_of_sustained": (round(3 * ach / sustained_f16_mfma_random_tflops()[0], 4)
                                                  if sustained_f16_mfma_random_tflops()[0] else None)}
        else:
            roof = {"bound": "mfma", "kernel": "conv3d_exact_kernel<3,2,4> (64->64 3^3 conv, %d^3; v_mfma_f32_32x32x2_f32)" % D, "achieved": round(ach, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "avg_launch_ms": round(ms, 4), "launches": agg[dom_key][1], "flop_per_launch": fl}
    conv_ms = sum(v[0] for v in agg.values()) / 2.0
    return dict(seconds=dt, steps=n_steps, voxels=world * n_steps * D ** 3, flops_scene=flops_scene, roofline=roof, precision=precision, executor=executor,
                conv_ms_per_step=conv_ms, layer_ms={str(k): round(v[0] / 2.0, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]},
                kernel_avg=prof.by_variant(), kernel_avg_timed=kernel_avg_timed, step_parts=parts, allgather_ms=ag_ms)


def bench_shipped_shape(args, device):
    """The reference's SHIPPED input shape (config/training/default.yaml:5,29: 64^3 grid, 768 CLIP channels, stored as a
    (D, H, W, C) float16 .npy): one scene through (a) the reference's own route -- fp16 DHWC -> fp32 NCDHW loader kernel, then
    each network's first projector conv reads the 805 MB float32 tensor -- (b) the fused route -- pixie_projector_conv0 reads
    the 403 MB grid once for both networks -- and (c) route (a) with each network replayed as one captured HIP graph (what
    is left when the ~800 Python -> ctypes launches per scene are out of the way).  Device-resident input, HIP-event timing."""
    from pixie_amd.unet import RegressionUNet, SegmentationUNet, predict_material_field, predict_material_field_from_voxel_grid
    from pixie_amd.voxel_grid import load_voxel_grid
    D, C = 64, 768
    kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
    seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
    seg.load_numpy_state(synthetic_state_dict(seg.cfg, 0)); cont.load_numpy_state(synthetic_state_dict(cont.cfg, 1000))
    seg, cont = seg.to(device).eval(), cont.to(device).eval()
    gen = torch.Generator(device=device).manual_seed(7)
    grid = torch.randn((D, D, D, C), generator=gen, device=device).to(torch.float16)

    def timed(fn, reps=5):
        for _ in range(4):                       # the first call runs eagerly, the second captures the graphs (pixie_amd/unet.py);
            fn()                                 # a leg whose input address differs between calls captures once more
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, 1e3 * (time.perf_counter() - t0) / reps

    res = {"workload": f"{D}^3 x {C} float16 (D,H,W,C) voxel grid -> SegmentationUNet+RegressionUNet forward + combine, 1 scene"}
    feat32 = load_voxel_grid(grid, device)
    legs = {"loader_then_networks": lambda: predict_material_field(seg, cont, load_voxel_grid(grid, device)),
            "fused_first_projector_conv": lambda: predict_material_field_from_voxel_grid(seg, cont, grid),
            "networks_only_fp32_input_resident": lambda: predict_material_field(seg, cont, feat32)}
    for graph in (True, False):          # product default: each network one pixie_unet_forward call replayed as a HIP graph
        seg.use_graph = cont.use_graph = graph
        for name, fn in legs.items():
            ms, wall = timed(fn)
            res[name + ("" if graph else "_eager")] = {"ms_per_scene": ms, "wall_ms_per_scene": wall, "voxels_per_s": D ** 3 / (ms * 1e-3)}
    seg.use_graph = cont.use_graph = True
    res["flops_per_scene"] = conv_flops(seg.cfg) + conv_flops(cont.cfg)
    return res


def _source_sha16(rel_path):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(REPO, rel_path), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _fresh(entries):
    """Keep only the profile entries taken from the kernel source that is in the tree NOW: every entry of profiles/pmc_traffic.json and
    profiles/mpm_counters.json carries `source` (the .hip file of its kernel) and `source_sha16` (its sha256 when the pass ran).  An
    entry that predates the last change to that file is dropped -- the line then prints `traffic: null` instead of a stale reading."""
    out = {}
    for k, v in (entries or {}).items():
        if isinstance(v, dict) and v.get("source") and v.get("source_sha16") == _source_sha16(v["source"]) and \
                (v.get("mpm_math_sha16") is None or v["mpm_math_sha16"] == _source_sha16("pixie_amd/csrc/mpm_math.h")):
            out[k] = v
    return out


def load_traffic():
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE in
    separate passes, corrected with calibration kernels of the same access widths, scripts/sessions/gpu_pmc.sh).  PMC counters
    cannot be collected from inside this process, so `traffic` is the profile's reading for the same kernel + shape -- if it was
    taken from the current source of that kernel (_fresh), else None."""
    try:
        return _fresh(json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))))
    except Exception:
        return {}


PLASTIC_LEGS = ("sand", "snow", "metal", "mixed")


def load_counters():
    """SQ counters and rocprofv3 kernel durations of the MPM block kernel per scene from the committed profile passes
    (profiles/mpm_counters.json, written by scripts/mpm_counters.py from separate `rocprofv3 --pmc` / `--kernel-trace --stats`
    runs of scripts/mpm_bench.py): counters cannot be collected from inside this process."""
    try:
        return _fresh(json.load(open(os.path.join(REPO, "profiles", "mpm_counters.json"))))
    except Exception:
        return {}


def _mpm_solver(sc, scatter_bits=None, wide=None, diag=False):
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    # THIS rank's GPU (main() made it current): the reference's signature defaults to "cuda:0", which under N ranks would put every
    # rank's solver on GPU 0 (found in round 6 by reading, not by running: no multi-GPU node has been available)
    dev = f"cuda:{torch.cuda.current_device()}"
    s = MPM_Simulator_WARP(10, device=dev, diag=diag)
    s.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]),
                                   n_grid=sc["n_grid"], grid_lim=sc["grid_lim"], device=dev)
    assert s.device.index == torch.cuda.current_device()
    if "F0" in sc:    # a plastic scene: the reference's config + a perturbed start, so that the return mappings work from substep 1
        start_plastic(s, sc, lambda f, a: s.set_field(f, a.reshape(a.shape[0], -1)))
    else:
        apply_scene(s, sc)
    if scatter_bits:
        s._set_scalar("scatter_bits", scatter_bits)
    if wide is not None:
        s._set_scalar("wide", wide)
    return s


def bench_mpm(args, rank, world, device, particles, n_grid, substeps, tag, scatter_bits=None, loop_api=False, v0_rms=None, scenario=None):
    """One scene per GPU: `substeps` substeps through run() (the fused step loop), timed with barriers; a short separate pass
    with HIP events around every launch for the kernel roofline; optionally the reference driver's own loop
    (gs_simulation.py:633-634: one p2g2p() call per substep, then an export), which the shim defers into the same run().
    `scenario`: None = the jelly ball of BASELINE configs[2]/[4]; "sand" / "snow" / "metal" / "mixed" = the reference's own plastic
    configuration of that name (pixie_amd.synthetic.PLASTIC_CONFIGS: its n_grid and substep; `n_grid` is ignored)."""
    if scenario:
        sc = mpm_plastic_scene(scenario, particles, seed=rank)
        n_grid = sc["n_grid"]
    else:
        sc = mpm_ball_scene(particles, seed=rank, n_grid=n_grid)
    s = _mpm_solver(sc, scatter_bits)
    if v0_rms:   # a scene in motion (strains of a few per cent): the block kernel's polar iteration then takes 2-3 steps per particle, not 1
        s.import_particle_v_from_torch(v0_rms * torch.randn((particles, 3), generator=torch.Generator().manual_seed(1 + rank)))
    s.run(sc["dt"], 300 if v0_rms else 100)  # warm-up: through the cautious first re-binning intervals (4, 16, 64 substeps) -- what scripts/mpm_bench.py does
    barrier_sync(world)
    t0 = time.perf_counter()
    s.run(sc["dt"], substeps)
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device)
    loop = None
    if loop_api:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(substeps):
            s.p2g2p(i, sc["dt"])
        t_host = time.perf_counter() - t1
        x = s.export_particle_x_to_torch()      # the per-frame observation: flushes the queued substeps
        torch.cuda.synchronize()
        dl = time.perf_counter() - t1
        loop = {"us_per_substep": 1e6 * dl / substeps, "value": particles * substeps / dl, "unit": "particle-steps/s",
                "host_us_per_p2g2p_call": 1e6 * t_host / substeps, "vs_run": dl / dt,
                "what": f"for step in range({substeps}): solver.p2g2p(step, dt); then export_particle_x_to_torch() -- the calls are queued "
                        "and run as ONE fused run() at the export (pixie_amd/mpm_solver.py)"}
    # separate short pass with per-launch HIP events on the launch stream for the roofline of the fused block kernel: the event
    # bracketing is a diagnostic of the PIXIE_DIAG build (libpixie_hip_diag.so: the same sources and kernels + the diagnostic
    # entry points), so this pass runs the same scene on a second solver from that library; the timed region above is the product's
    sd = _mpm_solver(sc, scatter_bits, diag=True)
    if v0_rms:
        sd.import_particle_v_from_torch(v0_rms * torch.randn((particles, 3), generator=torch.Generator().manual_seed(1 + rank)))
    sd.run(sc["dt"], 300 if v0_rms else 50)
    sd.set_profile(True)
    sd.run(sc["dt"], 200)
    torch.cuda.synchronize()
    p_ms, g_ms, n_launch = sd.kernel_times()
    sd.set_profile(False)
    del sd
    alg_bytes = 212.0 * particles + 44.0 * n_grid ** 3  # SURVEY.md section 8d (fused minimum, DENSE grid term)
    active_blocks = int(s._get_scalar("n_active_blocks"))
    touched_bytes = 212.0 * particles + 44.0 * 64 * active_blocks   # same, with the cells of the ACTIVE 4^3 blocks only
    part_bytes = 212.0 * particles
    ach = part_bytes / (p_ms * 1e-3) / 1e9 if p_ms > 0 else 0.0
    tr = (load_traffic().get(f"mpm_{tag}_block") or {})
    bits = int(s._get_scalar("scatter_bits"))
    roof = {"bound": "hbm", "kernel": f"mpm_block_kernel<true,true,..> (fused G2P + stress + P2G, one launch per substep; {bits}-bit fixed-point scatter)",
            "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBPS, 4),
            "traffic": tr.get("hbm_bytes_per_launch"), "traffic_source": "profiles/pmc_traffic.json" if tr else None,
            "avg_launch_ms": round(p_ms, 5), "grid_kernel_ms": round(g_ms, 5), "launches": int(n_launch),
            "bytes_per_launch": part_bytes, "substep_algorithmic_bytes": alg_bytes}
    oob = s.out_of_bounds
    finite = bool(torch.isfinite(s.get_field("x")).all())
    ps = world * particles * substeps / dt
    out = {"value": ps, "unit": "particle-steps/s", "substeps": substeps, "us_per_substep": 1e6 * dt / substeps,
           "config": {"workload": (f"{particles} particles, n_grid {n_grid}, grid_lim 2, dt {sc['dt']:g}, " +
                                   (f"the reference's custom_{scenario}_config.json (ball, perturbed F and v)" if scenario and scenario != "mixed" else
                                    "material ids 0/1/2/5 drawn per particle (ball, perturbed F and v)" if scenario else "jelly ball, tree scenario (impulse + ground slab)")
                                   + ", 1 scene per GPU"), "scatter_bits": bits, "n_grid": int(n_grid)},
           "algorithmic_GBps": alg_bytes * substeps / dt / 1e9,
           # SURVEY 8d lets a sparse-grid implementation count touched cells "but must report which it used": both are reported;
           # `frac_touched_cells` is the honest one for this implementation (its kernels never move the inactive cells' bytes)
           "frac_of_hbm_roofline_per_gpu": alg_bytes * substeps / dt / 1e9 / PEAK_HBM_GBPS,
           "frac_dense_grid": alg_bytes * substeps / dt / 1e9 / PEAK_HBM_GBPS,
           "frac_touched_cells": touched_bytes * substeps / dt / 1e9 / PEAK_HBM_GBPS,
           "active_blocks": active_blocks, "touched_cells": 64 * active_blocks, "substep_bytes_dense": alg_bytes, "substep_bytes_touched": touched_bytes,
           "roofline": roof, "finite": finite, "out_of_bounds": oob,
           "rebins": int(s._get_scalar("n_rebins")), "slow_path_particle_substeps": int(s._get_scalar("slow_path_particles"))}
    if loop is not None:
        out["p2g2p_loop"] = loop
    return out


def bench_mpm_floor(device, n_grid, substeps=2000):
    """What a substep costs when there is (almost) nothing to do: the same two launches per substep on a 2 000-particle ball in the
    same grid -- launch boundaries, one work item's life, the grid kernel's dependent round trips.  The 100 k-particle scene of
    BASELINE configs[2] (26.7 MB per substep = 3.3 us at 8 TB/s) sits a few us above this floor, which is why its HBM-roofline
    fraction is a statement about launch latency, not about the kernels (profiles/r4g_*)."""
    sc = mpm_ball_scene(2000, seed=3, n_grid=n_grid)
    s = _mpm_solver(sc)
    s.run(sc["dt"], 300)
    reps = []
    for _ in range(3):       # a repetition is ~30 ms: one host hiccup shows (one run of r5 read 21.9 us where every other read 13.7-14.1) -> the minimum
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(sc["dt"], substeps)
        torch.cuda.synchronize()
        reps.append(1e6 * (time.perf_counter() - t0) / substeps)
    return min(reps)


def bench_mpm_multi_scene(args, device, particles, n_grid, substeps, n_scenes):
    """BASELINE configs[3] runs a batch of scenes.  One 100 k-particle scene fills 41 % of the chip's workgroup slots and its
    launches are latency-bound, so several independent scenes on their own HIP streams (one host thread each: the launch
    loop is a foreign call that releases the GIL) share the GPU.  Reports aggregate particle-steps/s for this GPU."""
    import threading
    scenes = [mpm_ball_scene(particles, seed=10 + i, n_grid=n_grid) for i in range(n_scenes)]
    solvers = [_mpm_solver(sc) for sc in scenes]    # (forcing the five-waves-per-SIMD block kernel changes nothing here: profiles/r4k_mpm_multi_scene_variant.txt)
    from pixie_amd.mpm_solver import run_batch

    def run_all(n):     # the product's batch entry: one HIP stream and one host thread per scene (pixie_amd/mpm_solver.py: run_batch)
        run_batch(solvers, scenes[0]["dt"], n)
        torch.cuda.synchronize()

    run_all(50)
    reps = []          # a repetition is 25-70 ms of device time: one host hiccup (thread start, a re-binning's sync) shows; median of 3
    for _ in range(3):
        t0 = time.perf_counter()
        run_all(substeps)
        reps.append(time.perf_counter() - t0)
    dt = sorted(reps)[1]
    finite = all(bool(torch.isfinite(s.get_field("x")).all()) for s in solvers)
    return {"value": n_scenes * particles * substeps / dt, "unit": "particle-steps/s", "scenes": n_scenes, "substeps": substeps,
            "us_per_substep_per_scene": 1e6 * dt / substeps, "finite": finite, "repetitions_us_per_substep": [round(1e6 * r / substeps, 2) for r in reps],
            "timing": "median of 3 repetitions",
            "config": {"workload": f"{n_scenes} independent scenes of {particles} particles (n_grid {n_grid}) on {n_scenes} HIP streams of one GPU"}}


def bench_pipeline(args, device, substeps=1000):
    """BASELINE configs[2] as ONE timed scene, device-resident end to end: seeded 128^3 x 64 feature grid -> both networks
    (+ combine) -> un-scaling + K = 10 field transfer onto 100 k particles -> solver set-up -> `substeps` substeps
    (pixie_amd/pipeline.py; the reference: pixie/utils.py:736-779 + gs_simulation.py:483-531,633-634, three programs and two file
    hand-overs).  One warm scene, then the mean of two; parts from HIP events on the stream, total = wall clock with a device
    synchronisation at both ends.  The materials are whatever the randomly initialised networks predict (a mixed-material ball),
    un-scaled into the CFL-safe ranges of pixie_amd.synthetic.PIPELINE_RANGES."""
    from pixie_amd.pipeline import neural_scene_rollout
    from pixie_amd.synthetic import pipeline_scene
    from pixie_amd.unet import RegressionUNet, SegmentationUNet
    D, C = args.grid, args.feature_channels
    kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
    seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
    seg.load_numpy_state(synthetic_state_dict(seg.cfg, 0)); cont.load_numpy_state(synthetic_state_dict(cont.cfg, 1000))
    seg, cont = seg.to(device).eval(), cont.to(device).eval()
    sc = pipeline_scene(D, C, args.particles, seed=0, n_grid=args.n_grid)
    feat, mask = torch.from_numpy(sc["feat"]).to(device), torch.from_numpy(sc["mask"]).to(device)
    x0, vol = torch.from_numpy(sc["x"]).to(device), torch.from_numpy(sc["vol"]).to(device)

    def one(timings=None):
        return neural_scene_rollout(seg, cont, feat, mask, x0, vol, n_grid=sc["n_grid"], grid_lim=sc["grid_lim"], dt=sc["dt"], n_substeps=substeps,
                                    params=sc["params"], min_bounds=sc["min_bounds"], max_bounds=sc["max_bounds"],
                                    to_field_frame=lambda x: (x - 1.0) * sc["field_scale"], configure=lambda s: s.add_bounding_box(),
                                    ranges=sc["ranges"], timings=timings)
    for _ in range(3):      # set-up passes: pack the weights, capture and settle the networks' graphs (as in bench_unet)
        one()
    torch.cuda.synchronize()
    parts, walls = {}, []
    for _ in range(2):
        tm = {}
        t0 = time.perf_counter()
        solver, _, _ = one(tm)
        walls.append(1e3 * (time.perf_counter() - t0))
        for k_, v_ in tm.items():
            parts[k_] = parts.get(k_, 0.0) + v_ / 2
    # the same scenes as a software-pipelined batch: the rollout of scene i on a side stream under the networks of scene i + 1
    from pixie_amd.pipeline import neural_scene_batch
    nb = 4
    batch_kw = dict(n_grid=sc["n_grid"], grid_lim=sc["grid_lim"], dt=sc["dt"], n_substeps=substeps, params=sc["params"], min_bounds=sc["min_bounds"],
                    max_bounds=sc["max_bounds"], to_field_frame=lambda x: (x - 1.0) * sc["field_scale"], configure=lambda s: s.add_bounding_box(),
                    ranges=sc["ranges"])
    neural_scene_batch(seg, cont, [(feat, mask, x0, vol)] * 2, **batch_kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = neural_scene_batch(seg, cont, [(feat, mask, x0, vol)] * nb, **batch_kw)
    torch.cuda.synchronize()
    batch_ms = 1e3 * (time.perf_counter() - t0) / nb
    batch_finite = all(bool(torch.isfinite(s_.get_field("x")).all()) for s_ in done)
    del done
    mats = torch.bincount(solver.get_field("material").to(torch.int64), minlength=8).tolist()
    return {"pipeline_ms_per_scene": sum(walls) / 2, "parts_ms": {k_: round(v_, 3) for k_, v_ in parts.items()}, "substeps": substeps,
            "pipelined_batch_ms_per_scene": batch_ms, "pipelined_batch_scenes": nb, "pipelined_batch_finite": batch_finite,
            "finite": bool(torch.isfinite(solver.get_field("x")).all()), "out_of_bounds": solver.out_of_bounds, "particles_per_material_id": mats,
            "workload": f"{D}^3 x {C} feature grid -> SegmentationUNet + RegressionUNet -> field transfer onto {args.particles} particles -> "
                        f"{substeps} substeps (n_grid {args.n_grid}, dt {sc['dt']:g}); device-resident, 1 scene"}


def bench_field_transfer(args, device):
    """predicted 128^3 field -> per-particle properties (K = 10) for the MPM leg's particle count."""
    from pixie_amd.material_field import field_to_particles
    D, n = args.grid, args.particles
    gen = torch.Generator().manual_seed(0)
    pred = torch.zeros((11, D, D, D))
    pred[:3] = torch.randn((3, D, D, D), generator=gen) * 0.5
    pred[3:] = torch.nn.functional.one_hot(torch.randint(0, 8, (D, D, D), generator=gen), 8).permute(3, 0, 1, 2).float()
    g = (torch.arange(D) - (D - 1) / 2) / (D / 2)
    mask = ((g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2).sqrt() < 0.7).float()
    d = torch.randn((n, 3), generator=gen); d = d / d.norm(dim=1, keepdim=True)
    pos = d * (0.6 * torch.rand(n, generator=gen) ** (1 / 3))[:, None]
    pred_d, mask_d, pos_d = pred.to(device), mask.to(device), pos.to(device)
    field_to_particles(pred_d, mask_d, [-1, -1, -1], [1, 1, 1], pos_d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = field_to_particles(pred_d, mask_d, [-1, -1, -1], [1, 1, 1], pos_d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res = {"ms": ms, "particles_per_s": n / (ms * 1e-3), "too_far": int(out["n_too_far"]),
           "workload": f"{D}^3 field (18 % occupied), {n} particles, K=10 nearest voxels, mean/mode"}
    if not args.no_cpu_baseline:
        from oracle import field_oracle
        ns = min(n, 20000)
        t0 = time.perf_counter()
        field_oracle.field_to_particles(pred.numpy(), mask.numpy(), [-1, -1, -1], [1, 1, 1], pos[:ns].numpy())
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": ns / dt, "unit": "particles/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{ns} particles, same field: numpy + sklearn KNN + per-particle Python loop (the reference's method)"}
    return res


def cpu_baselines(args):
    """CPU baselines on this box's host cores, bounded samples (reported next to the GPU numbers, not a target):
    U-Net: oracle/unet_oracle.py on PyTorch's CPU kernels -- the reference's own modules do not exist on the GPU box
      (/root/reference is absent there); the oracle is pinned bit-for-bit to them (tests/golden).  Timed at the headline
      grid (one 128^3 pair: ~1 min, 25 GB of host memory); --cpu-baseline-small times 64^3 instead.
    MPM: oracle/mpm_oracle.c -- the restatement pinned to the reference's own kernels -- built with OpenMP on all host cores
      (`cores` = os.cpu_count()) at the bench's 100 k-particle scene and at the 1 M scene, the scalar build on one core beside it."""
    from oracle import unet_oracle
    from oracle.mpm_oracle import OracleMPM
    out = {}
    Dc = args.grid if not args.cpu_baseline_small else min(args.grid, 64)
    feat = feature_grid(Dc, args.feature_channels, seed=100)
    nets = []
    for oc, ws in ((8, 0), (3, 1000)):
        cfg = UNetConfig(feature_channels=args.feature_channels, grid_size=Dc, out_channels=oc)
        nets.append((cfg, synthetic_state_dict(cfg, ws)))
    # torch defaults to one thread per physical core of the HOST; a container with a CPU quota below that is oversubscribed 8x
    default_threads = torch.get_num_threads()
    if cpu_quota() and cpu_quota() < default_threads:
        torch.set_num_threads(max(1, int(round(cpu_quota()))))
    unet_threads = torch.get_num_threads()
    t0 = time.perf_counter()
    for cfg, sd in nets:
        unet_oracle.unet_forward(sd, cfg, feat)
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    out["unet"] = {"value": Dc ** 3 / dt, "unit": "voxels/s", "cores": unet_threads, "kind": "port",
                   "sample": f"{Dc}^3x{args.feature_channels} grid (" + ("the headline size" if Dc == args.grid else f"NOT the {args.grid}^3 headline size: 1/{(args.grid // Dc) ** 3} of its voxels, same work per voxel")
                             + f"), SegmentationUNet+RegressionUNet forward once, oracle/unet_oracle.py on PyTorch CPU "
                             f"({dt:.1f} s; the reference's modules are not present on this box, the oracle is pinned to them)",
                   "cpu_quota": cpu_quota(), "logical_cpus": os.cpu_count()}

    def timed(make, n, n_grid, steps, what, cores):
        sc = mpm_ball_scene(n, seed=0, n_grid=n_grid)
        o = make(n, sc)
        o.load_initial_data(sc["x"], sc["vol"], sc["cov"])
        apply_scene(o, sc)
        o.run(sc["dt"], 1)
        t0 = time.perf_counter()
        o.run(sc["dt"], steps)
        dt = time.perf_counter() - t0
        return {"value": n * steps / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
                "sample": f"{n} particles, n_grid {n_grid}, {steps} substeps ({dt:.1f} s), {what}"}

    n = min(args.particles, 100_000)
    # The OpenMP build of the C oracle (pinned to the reference's kernels, tests/test_mpm_ref_golden.py; OpenMP over particles / grid
    # nodes, atomic-free P2G: particles sorted into 4^3-cell tiles, 8 colours of tiles one after the other) runs in CHILD processes:
    # libgomp reads OMP_NUM_THREADS once, when it is first loaded, and this process has long since loaded it.  The thread count is the
    # one that is FASTEST on this host among {the cgroup CPU quota; all, 1/4, 1/16 of torch's threads}, probed with 10 substeps each --
    # a container with a CPU quota below its visible CPU count runs slower on more threads (the boxes of rounds 3-5 show 256 logical
    # CPUs and schedule 16: 1.4-1.8x one core on 128-256 threads, 5-6x on 16).
    import subprocess
    child = ("import sys, time; sys.path.insert(0, %r); from oracle.mpm_oracle import OracleMPM; from pixie_amd.synthetic import apply_scene, mpm_ball_scene;"
             "n, ng, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]);"
             "sc = mpm_ball_scene(n, seed=0, n_grid=ng); o = OracleMPM(n, sc['n_grid'], sc['grid_lim'], 'f32_omp'); o.load_initial_data(sc['x'], sc['vol'], sc['cov']);"
             "apply_scene(o, sc); o.run(sc['dt'], 1); t0 = time.perf_counter(); o.run(sc['dt'], steps); print(time.perf_counter() - t0)") % REPO

    def omp_seconds(threads, n_, ng_, steps_):
        r = subprocess.run([sys.executable, "-c", child, str(n_), str(ng_), str(steps_)], env={**os.environ, "OMP_NUM_THREADS": str(threads)},
                           capture_output=True, text=True, timeout=600)
        return float(r.stdout.strip().splitlines()[-1])

    tried, best = {}, None
    cand = {max(1, torch.get_num_threads() // k) for k in (1, 4, 16)}
    if cpu_quota():
        cand.add(max(1, int(round(cpu_quota()))))     # what the container is actually scheduled on (r5c: 16 of 256 visible CPUs)
    for th in sorted(cand, reverse=True):
        try:
            tried[th] = omp_seconds(th, n, args.n_grid, 10)
            if best is None or tried[th] < tried[best]:
                best = th
        except Exception:
            pass
    cores = best or torch.get_num_threads()

    def timed_omp(n_, ng_, steps_):
        dt = omp_seconds(cores, n_, ng_, steps_)
        return {"value": n_ * steps_ / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
                "sample": f"{n_} particles, n_grid {ng_}, {steps_} substeps ({dt:.1f} s), oracle/mpm_oracle.c float32, OpenMP on {cores} host threads "
                          "(atomic-free coloured P2G; child process)"}
    out["mpm"] = timed_omp(n, args.n_grid, 400)
    out["mpm"]["thread_probe_s_per_10_substeps"] = {str(k): round(v, 3) for k, v in tried.items()}
    out["mpm"]["cpu_quota"] = cpu_quota()
    out["mpm"]["single_core"] = timed(lambda n_, sc: OracleMPM(n_, sc["n_grid"], sc["grid_lim"], "f32"), n, args.n_grid, 20,
                                      "oracle/mpm_oracle.c float32, scalar C", 1)
    if not (args.no_mpm or args.no_mpm_large):
        out["mpm_1m"] = timed_omp(1_000_000, 120, 60)
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves, as the reference's
    inference program does with mp.spawn (WG/trainer/inference_combined.py:335-353).  Re-executes this file under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1, a free port) with the same arguments; the
    ranks' stdout passes through, so exactly one JSON line (rank 0's) is printed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


def launcher_selftest(args, rank, world):
    """The N > 1 control path without a device: rendezvous, barrier, max-over-ranks timing, the field all-gather in its
    wire format, scene -> rank mapping.  Prints the same single JSON line shape (no throughput claim)."""
    import torch.distributed as dist
    D = 8
    mine = pd.shard_scenes(world, rank, world)
    cont = torch.stack([torch.full((3, D, D, D), float(s)) for s in mine])
    seg = torch.stack([torch.full((D, D, D), s % 8, dtype=torch.int32) for s in mine])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g_cont, g_seg = pd.all_gather_fields(cont, seg)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    order = pd.unshard_order(world, world)
    ok = all(float(g_cont[order[i], 0, 0, 0, 0]) == float(i) and int(g_seg[order[i], 0, 0, 0]) == i % 8 for i in range(world))
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest (no kernels)", "value": None, "unit": None, "n_gpus": world, "world": world,
                          "collective_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                          "backend": dist.get_backend() if dist.is_initialized() else None, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "gather_ok": bool(ok)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def bench_unet_dry(args, rank, world, device, **_):
    """--dry-run stand-in for bench_unet: the same control flow (per-rank set-up, set-up passes, warm-up, the timing protocol,
    the field all-gather in its wire format, max over ranks) on CPU tensors of a small grid; no kernel runs."""
    D = min(args.grid, 16)
    cfgs = [UNetConfig(feature_channels=args.feature_channels, grid_size=D, out_channels=oc) for oc in (8, 3)]
    feat = torch.from_numpy(feature_grid(D, args.feature_channels, seed=100 + rank))
    gen = torch.Generator().manual_seed(rank)
    cont_pred = torch.randn((1, 3, D, D, D), generator=gen)
    seg_pred = torch.randint(0, 8, (1, D, D, D), generator=gen, dtype=torch.int32)

    def step():
        time.sleep(1e-3)     # (the device work of a scene)
        if world > 1:
            pd.all_gather_fields(cont_pred, seg_pred)
    for _ in range(3 + args.warmup):
        step()
    dt = timed_steps(step, args.steps, world, device)
    return dict(seconds=dt, steps=args.steps, voxels=world * args.steps * D ** 3, flops_scene=sum(conv_flops(c) for c in cfgs), roofline=None,
                precision="f16x3", executor="dry-run", conv_ms_per_step=0.0, layer_ms={}, kernel_avg={}, kernel_avg_timed=None,
                step_parts={"note": f"dry run on a {D}^3 grid of CPU tensors", "feature_grid_sum": float(feat.sum())},
                allgather_ms=time_allgather(cont_pred, seg_pred, world, device))


def bench_mpm_dry(args, rank, world, device, particles, n_grid, substeps, tag, **_):
    """--dry-run stand-in for bench_mpm: per-rank scene generation (the CPU-side cost every rank pays concurrently) and the
    timing protocol; no kernel runs."""
    sc = mpm_ball_scene(min(particles, 20000), seed=rank, n_grid=n_grid)
    dt = timed_steps(lambda: time.sleep(1e-5), 10, world, device)
    return {"value": world * sc["x"].shape[0] * 10 / dt, "unit": "particle-steps/s", "substeps": 10, "us_per_substep": 1e5 * dt,
            "config": {"workload": f"dry run: {sc['x'].shape[0]} particles generated per rank, no kernels", "scatter_bits": 64},
            "algorithmic_GBps": None, "frac_of_hbm_roofline_per_gpu": None, "frac_dense_grid": None, "frac_touched_cells": None,
            "active_blocks": None, "roofline": None, "finite": True, "out_of_bounds": 0,
            "rebins": 0, "slow_path_particle_substeps": 0}


from bench_report import assemble_detail, compact_line  # noqa: E402,F401  (the record / the one-line rendering of it)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.launcher_selftest:
        rank, world, local = pd.init_process_group(args.backend or "gloo")
        raise SystemExit(launcher_selftest(args, rank, world))
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (pixie_amd has no CPU path); --dry-run exercises the N-rank control path on CPU tensors")
    rank, world, local = pd.init_process_group((args.backend or "gloo") if dry else args.backend)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    threads = pin_rank_resources(rank, world)
    if dry:
        device = torch.device("cpu")
        U, M = bench_unet_dry, bench_mpm_dry
    else:
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        U, M = bench_unet, bench_mpm

    # stdout carries exactly ONE line (the JSON): the solver shim's reference-style progress prints are switched off (the driver's
    # record keeps one tail window for both streams), anything else a library prints goes to stderr
    import pixie_amd.mpm_solver as _shim
    _shim.VERBOSE = False
    with contextlib.redirect_stdout(sys.stderr):
        u = U(args, rank, world, device)
        # the same step on the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32, no operand splitting): the precision ruling
        # of VERDICT r1 asks for this line beside the headline; a few steps suffice (bounded run time)
        u32 = None
        if not args.no_exact_f32 and u["precision"] == "f16x3":
            u32 = U(args, rank, world, device, precision_override="f32", steps=min(args.steps, 3), warmup=1)
        m = None if args.no_mpm else M(args, rank, world, device, args.particles, args.n_grid, args.mpm_substeps, "100k", loop_api=True)
        # BASELINE configs[4]'s per-GPU MPM workload (1M particles, n_grid 120, 2000 substeps): where the HBM roofline fraction is meaningful
        m_large = None if (args.no_mpm or args.no_mpm_large) else M(args, rank, world, device, 1_000_000, 120, args.mpm_large_substeps, "1m")
        other_bits = {64: 32, 32: 64}
        m_alt = m_large_alt = m_multi = ft = shipped = u256 = cpu = pipe = None
        if rank == 0 and world == 1 and not args.no_mpm and not dry:
            # the other scatter mode beside the default (exact 64-bit <-> packed 32-bit pairs), and the multi-scene leg
            if args.full:
                m_alt = bench_mpm(args, rank, world, device, args.particles, args.n_grid, args.mpm_substeps, "100k",
                                  scatter_bits=other_bits[m["config"]["scatter_bits"]])
            if m_large is not None and args.full:
                m_large_alt = bench_mpm(args, rank, world, device, 1_000_000, 120, min(args.mpm_large_substeps, 500), "1m",
                                        scatter_bits=other_bits[m_large["config"]["scatter_bits"]])
            if m_large is not None and args.full:
                # the same scene IN MOTION (random particle velocities, 0.6 m/s rms per component): the fused kernel's work depends on the
                # data -- a nearly rigid particle leaves the polar iteration after one step -- so the headline scene (one impulse,
                # quasi-static) is the kernel's best case; this is the other one
                mv = bench_mpm(args, rank, world, device, 1_000_000, 120, min(args.mpm_large_substeps, 500), "1m", v0_rms=0.6)
                m_large["in_motion"] = {k: mv[k] for k in ("value", "substeps", "us_per_substep", "frac_dense_grid", "frac_touched_cells", "active_blocks", "finite", "rebins")}
                m_large["in_motion"]["block_kernel_us"] = round(1e3 * mv["roofline"]["avg_launch_ms"], 2)
                m_large["in_motion"]["what"] = "same 1 M scene with random initial particle velocities (0.6 m/s rms per component): strains of a few per cent"
            if m_large is not None:
                # two 1 M scenes on two streams: the latency-bound grid kernel of one scene runs under the VALU-bound block kernel
                # of the other (a single scene cannot overlap them: each needs the other's output)
                two = bench_mpm_multi_scene(args, device, 1_000_000, 120, min(args.mpm_large_substeps, 600), 2)
                per_scene_us = two["us_per_substep_per_scene"] / 2.0
                m_large["two_scenes"] = {"value": two["value"], "unit": "particle-steps/s", "scenes": 2, "us_per_scene_substep": per_scene_us, "finite": two["finite"],
                                         "frac_dense_grid": m_large["substep_bytes_dense"] / (per_scene_us * 1e-6) / 1e9 / PEAK_HBM_GBPS,
                                         "frac_touched_cells": m_large["substep_bytes_touched"] / (per_scene_us * 1e-6) / 1e9 / PEAK_HBM_GBPS,
                                         "config": two["config"], "timing": two["timing"],
                                         "repetitions_us_per_scene_substep": [round(r / 2.0, 2) for r in two["repetitions_us_per_substep"]]}
            if m_large is not None and not args.no_mpm_plastic:
                # SURVEY 8f-4 "plastic materials at scale": the reference's own sand / snow / metal configurations and a mixed-material
                # scene at the 1 M size (the constitutive branch is the only thing that differs from the jelly leg above)
                m_plastic = {}
                for name in (PLASTIC_LEGS if args.full else PLASTIC_LEGS[:1]):     # default: the reference's own sand configuration
                    mp = bench_mpm(args, rank, world, device, 1_000_000, 0, args.mpm_plastic_substeps, "1m_" + name, scenario=name)
                    m_plastic[name] = {k: mp[k] for k in ("value", "substeps", "us_per_substep", "frac_dense_grid", "frac_touched_cells", "active_blocks",
                                                          "finite", "out_of_bounds", "rebins", "config")}
                    m_plastic[name]["block_kernel_us"] = round(1e3 * mp["roofline"]["avg_launch_ms"], 2)
                    m_plastic[name]["grid_kernel_us"] = round(1e3 * mp["roofline"]["grid_kernel_ms"], 2)
                    m_plastic[name]["vs_jelly_1m_substep"] = mp["us_per_substep"] / m_large["us_per_substep"]
                    m_plastic[name]["counters"] = load_counters().get("1m_" + name)
                m_large["plastic"] = m_plastic
                m_large["counters"] = load_counters().get("1m_jelly")
            m_multi = bench_mpm_multi_scene(args, device, args.particles, args.n_grid, args.mpm_substeps, 3)
            if args.full:
                six = bench_mpm_multi_scene(args, device, args.particles, args.n_grid, args.mpm_substeps, 6)
                m_multi["six_scenes"] = {k: six[k] for k in ("value", "unit", "scenes", "us_per_substep_per_scene", "finite")}
            # the batch configuration (BASELINE configs[3]: several scenes per job) as roofline fractions: the bytes of one scene's
            # substep x the scenes, over the time the GPU needs for one substep of all of them
            for mm in ([m_multi, m_multi["six_scenes"]] if args.full else [m_multi]):
                t = mm["us_per_substep_per_scene"] * 1e-6         # (wall time of one substep of ALL scenes of the leg)
                mm["frac_dense_grid"] = mm["scenes"] * m["substep_bytes_dense"] / t / 1e9 / PEAK_HBM_GBPS
                mm["frac_touched_cells"] = mm["scenes"] * m["substep_bytes_touched"] / t / 1e9 / PEAK_HBM_GBPS
            m["floor_us"] = bench_mpm_floor(device, args.n_grid)
            m["frac_of_floor"] = m["floor_us"] / m["us_per_substep"]
        if not dry:
            ft = bench_field_transfer(args, device) if (rank == 0 and not args.no_mpm) else None
            pipe = bench_pipeline(args, device, args.mpm_substeps) if (rank == 0 and world == 1 and not args.no_mpm and not args.no_pipeline) else None
            shipped = bench_shipped_shape(args, device) if (rank == 0 and world == 1 and args.full and not args.no_shipped_shape) else None
            # BASELINE configs[4]'s per-GPU U-Net workload: 256^3 x 128 (217 TFLOP per scene, ~60 GiB of workspace)
            if rank == 0 and world == 1 and args.full and not args.no_unet_256 and u["precision"] == "f16x3" and args.grid == 128:
                torch.cuda.empty_cache()
                big = argparse.Namespace(**{**vars(args), "grid": 256, "feature_channels": 128, "dual_stream_diagnostic": False})
                u256 = bench_unet(big, rank, world, device, steps=2, warmup=0, device_input=True)
                torch.cuda.empty_cache()
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                cpu = cpu_baselines(args)

    if rank == 0:
        detail = assemble_detail(args, world, u, u32, m, m_large, m_alt, m_large_alt, m_multi, ft, shipped, u256, cpu,
                                 threads_per_rank=threads, dry=dry)
        if pipe is not None:
            detail["pipeline_configs2"] = pipe
        path = None
        try:       # gpurun_out/ travels back from the GPU box; the session scripts copy the file into profiles/
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            path = os.path.join("gpurun_out", "bench_detail.json")
            with open(os.path.join(REPO, path), "w") as f:
                json.dump(detail, f, indent=1)
        except OSError:
            path = None
        line = compact_line(detail, path)
        text = json.dumps(line)
        # the driver's record keeps ~6 KB of the line: drop optional keys (they stay in the detail file) rather than lose the run
        for key in ("telemetry", "exact_f32", "mpm_cpu_baseline", "mpm_1m_cpu_baseline", "mpm_1m_kernel", "mpm_kernel", "pipeline_parts_ms",
                    "mpm_1m_in_motion_us_per_substep", "mpm_1m_in_motion_frac_touched", "p2g2p_loop_vs_run", "mpm_exact_scatter_us_per_substep"):
            if len(text) < 6144:
                break
            if line.pop(key, None) is not None:
                print(f"bench.py: line was {len(text)} bytes, dropped '{key}' (kept in {path})", file=sys.stderr)
                text = json.dumps(line)
        print(text)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
