"""How bench.py's legs become its output: `assemble_detail` builds the full record of a run (written to gpurun_out/bench_detail.json, copied
to profiles/bench_detail_<tag>.json by the session scripts) and `compact_line` renders the ONE JSON line of the driver contract from it
(<= 6 KB: the driver's record keeps a tail window).  Pure dictionary work: no kernels, no timing.  Split out of bench.py in round 6."""
import torch


def assemble_detail(args, world, u, u32=None, m=None, m_large=None, m_alt=None, m_large_alt=None, m_multi=None, ft=None, shipped=None,
                  u256=None, cpu=None, threads_per_rank=None, dry=False):
    """The full record of a run, from the legs' records (shared by the real run and --dry-run).  Rank 0 writes it to
    `gpurun_out/bench_detail.json` (copied to profiles/bench_detail_<tag>.json by the session scripts); the ONE JSON line on
    stdout is `compact_line()` of it."""
    vps = u["voxels"] / u["seconds"]
    ms_step = 1e3 * u["seconds"] / u["steps"]
    line = {
        "metric": "voxels/s (128^3 U-Net fwd) + MPM particle-steps/s",
        "value": vps, "unit": "voxels/s", "n_gpus": world, "world": world,
        "collective_ranks": torch.distributed.get_world_size() if world > 1 else 1,
        "backend": (torch.distributed.get_backend() + (" (RCCL)" if torch.distributed.get_backend() == "nccl" else "")) if world > 1 else None,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if u["precision"] == "f32" else "f32 (operands split fp16 hi+lo, 3 f16 MFMAs/product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"{args.grid}^3x{args.feature_channels} feature grid -> SegmentationUNet+RegressionUNet forward "
                               f"(+argmax/one-hot combine" + (", + all-gather of fields" if world > 1 else "") + "), 1 scene per GPU per step",
                   "grid": args.grid, "feature_channels": args.feature_channels, "parallelism": f"scene-parallel x{world}",
                   "weights": "seeded random init of the reference architecture",
                   "executor": u["executor"] + " (one pixie_unet_forward call per network; graph replays of >= 128^3 grids run back to back on one stream)"},
        "unet_tflops": u["flops_scene"] * world * u["steps"] / u["seconds"] / 1e12,
        "unet_conv_ms_per_step": u["conv_ms_per_step"],
        "step_decomposition": u["step_parts"],
        # N > 1: the collective timed alone, so that the driver's 1 -> 8 curve splits into compute and exchange
        "allgather_ms": u.get("allgather_ms"),
        "compute_ms_per_step": (ms_step - u["allgather_ms"]) if u.get("allgather_ms") is not None else ms_step,
        "host_threads_per_rank": threads_per_rank,
        "roofline": u["roofline"],
    }
    if dry:
        line["dry_run"] = True
        line["metric"] = "DRY RUN (CPU tensors, no kernels): " + line["metric"]
    if u32 is not None:
        line["exact_f32"] = {"dtype": "f32 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32)", "value": u32["voxels"] / u32["seconds"], "unit": "voxels/s",
                             "steps": u32["steps"], "ms_per_step": 1e3 * u32["seconds"] / u32["steps"],
                             "unet_tflops": u32["flops_scene"] * world * u32["steps"] / u32["seconds"] / 1e12, "roofline": u32["roofline"]}
    if m is not None:
        line["mpm"] = m
        if m_alt is not None:
            line["mpm"]["other_scatter_mode"] = {k: m_alt[k] for k in ("value", "us_per_substep", "config", "frac_of_hbm_roofline_per_gpu", "roofline", "finite")}
        if m_multi is not None:
            line["mpm"]["multi_scene"] = m_multi
    if m_large is not None:
        line["mpm_1m"] = m_large
        if m_large_alt is not None:
            line["mpm_1m"]["other_scatter_mode"] = {k: m_large_alt[k] for k in ("value", "substeps", "us_per_substep", "config", "frac_of_hbm_roofline_per_gpu", "roofline", "finite")}
    if ft is not None:
        line["field_to_particles"] = ft
    if shipped is not None:
        line["shipped_shape_64x768"] = shipped
    if u256 is not None:
        line["unet_256x128"] = {"workload": "256^3 x 128 feature grid -> SegmentationUNet+RegressionUNet forward (+combine), 1 scene per step (BASELINE configs[4] per-GPU grid)",
                                "value": u256["voxels"] / u256["seconds"], "unit": "voxels/s", "steps": u256["steps"],
                                "ms_per_step": 1e3 * u256["seconds"] / u256["steps"],
                                "unet_tflops": u256["flops_scene"] * u256["steps"] / u256["seconds"] / 1e12, "roofline": u256["roofline"],
                                "step_decomposition": u256["step_parts"], "peak_device_memory_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if cpu is not None:
        line["cpu_baseline"] = cpu["unet"]
        if "mpm" in line:
            line["mpm"]["cpu_baseline"] = cpu["mpm"]
        if "mpm_1m" in line and "mpm_1m" in cpu:
            line["mpm_1m"]["cpu_baseline"] = cpu["mpm_1m"]
    line["layer_ms_top"] = u["layer_ms"]
    # per kernel NAME, all shapes pooled: comparable with the avg column of profiles/*_kernel_stats.csv (rocprofv3 --stats)
    line["conv_kernel_avg_ms"] = u["kernel_avg"]                           # single-stream pass
    if u["kernel_avg_timed"] is not None:
        line["conv_kernel_avg_ms_dual_stream"] = u["kernel_avg_timed"]    # two streams: launches of the two networks overlap
    return line


def _pick(d, *keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def _r(x, digits=4):
    return None if x is None else (float(f"{x:.{digits}g}") if isinstance(x, float) else x)


def compact_line(d, detail_path=None):
    """The ONE JSON line on stdout: the driver's contract keys, `roofline`, `cpu_baseline` and the headline number of every
    leg as a top-level scalar -- kept under 6 KB so the driver's record holds all of it (the 9.4 KB line of
    round 3 lost the MPM numbers).  Per-layer / per-variant tables live in the detail file."""
    line = _pick(d, "metric", "value", "unit", "n_gpus", "world", "collective_ranks", "backend", "steps", "warmup", "ms_per_step",
                 "higher_is_better", "scaling", "dtype", "data", "dry_run", "allgather_ms", "compute_ms_per_step", "host_threads_per_rank")
    line["vs_baseline"] = None
    line["backend"] = d.get("backend")
    line["allgather_ms"] = d.get("allgather_ms")
    line["config"] = _pick(d["config"], "workload", "grid", "feature_channels", "parallelism")
    line["config"]["executor"] = d["config"]["executor"].split(" (")[0]
    line["unet_tflops"] = _r(d.get("unet_tflops"))
    rf = d.get("roofline")
    line["roofline"] = None if rf is None else {**_pick(rf, "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches",
                                                      "mfma_hw_frac", "mfma_hw_frac_of_sustained"),
                                                "kernel": rf["kernel"].split(" (")[0]}
    if (d.get("step_decomposition") or {}).get("h2d_feature_grid_ms_pinned") is not None:
        line["h2d_feature_grid_ms_pinned"] = d["step_decomposition"]["h2d_feature_grid_ms_pinned"]     # not in `value`: inputs are resident
    if (d.get("step_decomposition") or {}).get("pack_fields_us") is not None:
        line["pack_fields_us"] = d["step_decomposition"]["pack_fields_us"]
    tele = (d.get("step_decomposition") or {}).get("telemetry_during") or {}
    if tele:
        line["telemetry"] = _pick(tele, "sclk_mhz", "power_w", "temp_c")
    ex = d.get("exact_f32")
    if ex:
        line["exact_f32"] = {"voxels_per_s": _r(ex["value"]), "ms_per_step": _r(ex["ms_per_step"]),
                             "frac_of_f32_mfma_peak": (ex.get("roofline") or {}).get("frac")}
        line["exact_f32_voxels_per_s"] = _r(ex["value"])
    u256 = d.get("unet_256x128")
    if u256:
        line["unet_256x128_voxels_per_s"], line["unet_256x128_ms_per_step"] = _r(u256["value"]), _r(u256["ms_per_step"])
    for key, tag in (("mpm", "mpm"), ("mpm_1m", "mpm_1m")):
        m = d.get(key)
        if not m:
            continue
        line[key] = True          # the leg ran; its numbers are the `<leg>_*` scalars below
        line[f"{tag}_particle_steps_per_s"] = _r(m["value"])
        line[f"{tag}_us_per_substep"] = _r(m["us_per_substep"])
        line[f"{tag}_substeps"] = m["substeps"]
        line[f"{tag}_frac_dense"] = _r(m.get("frac_dense_grid"))
        line[f"{tag}_frac_touched"] = _r(m.get("frac_touched_cells"))
        line[f"{tag}_active_blocks"] = m.get("active_blocks")
        mr = m.get("roofline") or {}
        if mr:
            line[f"{tag}_kernel"] = {"name": "mpm_block_kernel", "achieved_GBps": mr.get("achieved"), "frac": mr.get("frac"),
                                     "avg_launch_us": _r(1e3 * mr["avg_launch_ms"]), "grid_kernel_us": _r(1e3 * mr["grid_kernel_ms"]),
                                     "traffic": mr.get("traffic")}
        if m.get("cpu_baseline"):
            cb = m["cpu_baseline"]
            line[f"{tag}_cpu_baseline"] = {**_pick(cb, "unit", "cores", "kind"), "value": _r(cb["value"]), "sample": cb["sample"][:110],
                                           "single_core_value": _r((cb.get("single_core") or {}).get("value"))}
    two = (d.get("mpm_1m") or {}).get("two_scenes")
    if two:
        line["mpm_1m_2_scenes_particle_steps_per_s"] = _r(two["value"])
        line["mpm_1m_2_scenes_us_per_scene_substep"] = _r(two["us_per_scene_substep"])
        line["mpm_1m_2_scenes_frac_touched"] = _r(two["frac_touched_cells"])
        line["mpm_1m_2_scenes_frac_dense"] = _r(two["frac_dense_grid"])
    for name, mp in ((d.get("mpm_1m") or {}).get("plastic") or {}).items():
        # SURVEY 8f-4 "plastic materials at scale": top-level scalars per leg (the reference's sand / snow / metal configs, a mixed scene)
        ctr = mp.get("counters") or {}
        pre = f"mpm_1m_{name}"
        line[pre + "_us_per_substep"] = _r(mp["us_per_substep"])
        # a scene that fills a few per cent of a 200^3 grid: the dense-grid figure counts 8 M mostly empty cells and is not quoted (VERDICT r5 #4c)
        line[pre + "_frac_dense"] = _r(mp["frac_dense_grid"], 3) if (mp.get("config") or {}).get("n_grid", 0) <= 120 else None
        line[pre + "_frac_touched"] = _r(mp["frac_touched_cells"], 3)
        line[pre + "_vs_jelly"] = _r(mp["vs_jelly_1m_substep"], 3)
        line[pre + "_block_us_rocprofv3"], line[pre + "_valu_per_wave"] = ctr.get("block_kernel_us"), ctr.get("valu_per_wave")
    jc = (d.get("mpm_1m") or {}).get("counters") or {}
    if jc:
        line["mpm_1m_valu_per_wave"], line["mpm_1m_block_kernel_us_rocprofv3"] = jc.get("valu_per_wave"), jc.get("block_kernel_us")
    mv = (d.get("mpm_1m") or {}).get("in_motion")
    if mv:
        line["mpm_1m_in_motion_us_per_substep"] = _r(mv["us_per_substep"])
        line["mpm_1m_in_motion_frac_touched"] = _r(mv["frac_touched_cells"])
    m = d.get("mpm") or {}
    line["mpm_frac"] = line.get("mpm_frac_dense")
    if m.get("floor_us"):
        # single 100 k scene: launch-latency bound -- the substep against the empty-scene substep of the same two launches
        line["mpm_floor_us"], line["mpm_frac_of_floor"] = _r(m["floor_us"]), _r(m["frac_of_floor"])
    if m.get("p2g2p_loop"):
        line["p2g2p_loop_vs_run"] = _r(m["p2g2p_loop"]["vs_run"])
        line["p2g2p_loop_us_per_substep"] = _r(m["p2g2p_loop"]["us_per_substep"])
    if m.get("multi_scene"):
        ms = m["multi_scene"]
        # the MPM headline of the BATCH configuration (BASELINE configs[3]): >= 3 scenes per GPU, where the launch latency of one
        # scene is hidden behind the others
        line["mpm_batch"] = {"scenes_per_gpu": ms["scenes"], "particle_steps_per_s": _r(ms["value"]), "frac_dense": _r(ms.get("frac_dense_grid")),
                             "frac_touched": _r(ms.get("frac_touched_cells"))}
        line["mpm_3_scenes_particle_steps_per_s"] = _r(ms["value"])
        if ms.get("six_scenes"):
            line["mpm_6_scenes_particle_steps_per_s"] = _r(ms["six_scenes"]["value"])
            line["mpm_6_scenes_frac_dense"] = _r(ms["six_scenes"].get("frac_dense_grid"))
    if m.get("other_scatter_mode"):
        line["mpm_exact_scatter_us_per_substep"] = _r(m["other_scatter_mode"]["us_per_substep"])
    sh = d.get("shipped_shape_64x768") or {}
    if sh.get("fused_first_projector_conv"):
        line["shipped_64x768_ms_per_scene"] = _r(sh["fused_first_projector_conv"]["ms_per_scene"])
    if d.get("pipeline_configs2"):
        pp = d["pipeline_configs2"]
        line["pipeline_ms_per_scene"] = _r(pp["pipeline_ms_per_scene"])
        line["pipeline_batch_ms_per_scene"] = _r(pp.get("pipelined_batch_ms_per_scene"))     # rollout of scene i under the networks of scene i + 1
        line["pipeline_parts_ms"] = {k.replace("_ms", ""): _r(v) for k, v in pp["parts_ms"].items() if k != "total_ms"}
    if d.get("field_to_particles"):
        line["field_to_particles_ms"] = _r(d["field_to_particles"]["ms"])
    cb = d.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**_pick(cb, "unit", "cores", "kind", "cpu_quota", "logical_cpus"), "value": _r(cb["value"]), "sample": cb["sample"][:160]}
    line["detail_file"] = detail_path
    return line
