"""world_size-2 gloo tests (CPU) of the multi-GPU path: scene sharding == DistributedSampler, and the
field all-gather round-trips the compact wire format."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data.distributed import DistributedSampler

from pixie_amd import distributed as pd


@pytest.mark.parametrize("n,world", [(8, 2), (7, 2), (5, 4), (1, 2), (16, 8), (3, 8)])
def test_shard_equals_distributed_sampler(n, world):
    ds = list(range(n))
    for r in range(world):
        want = list(DistributedSampler(ds, num_replicas=world, rank=r, shuffle=False))
        assert pd.shard_scenes(n, r, world) == want
    order = pd.unshard_order(n, world)
    gathered = sum((pd.shard_scenes(n, r, world) for r in range(world)), [])
    assert [gathered[p] for p in order] == list(range(n))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_scenes, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    pd.init_process_group("gloo")
    mine = pd.shard_scenes(n_scenes, rank, world)
    D = 6
    cont = torch.stack([torch.full((3, D, D, D), float(s)) + torch.arange(3.0)[:, None, None, None] for s in mine])
    seg = torch.stack([torch.full((D, D, D), s % 8, dtype=torch.int32) for s in mine])
    g_cont, g_seg = pd.all_gather_fields(cont, seg)
    order = pd.unshard_order(n_scenes, world)
    ok = g_cont.shape == (world * len(mine), 3, D, D, D) and g_seg.dtype == torch.uint8
    for i in range(n_scenes):
        ok = ok and float(g_cont[order[i], 0, 0, 0, 0]) == float(i) and float(g_cont[order[i], 2, 1, 1, 1]) == i + 2.0
        ok = ok and int(g_seg[order[i], 0, 0, 0]) == i % 8
    comb = pd.combined_from_wire(g_cont, g_seg)
    ok = ok and comb.shape == (world * len(mine), 11, D, D, D) and bool(torch.all(comb[:, 3:].sum(1) == 1))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scenes", [4, 3])
def test_all_gather_fields_gloo_world2(n_scenes):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_scenes, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_single_process_passthrough():
    cont = torch.randn(2, 3, 4, 4, 4); seg = torch.randint(0, 8, (2, 4, 4, 4))
    c, s = pd.all_gather_fields(cont, seg)
    assert torch.equal(c, cont) and s.dtype == torch.uint8 and torch.equal(s.long(), seg)


def test_bench_self_launch_gloo_world2():
    """`python bench.py --gpus 2` with no launcher must spawn its own ranks (the first scaling run must not die
    on a launcher error).  --launcher-selftest exercises exactly that path on CPU: self-launch under torch.distributed.run,
    gloo rendezvous on 127.0.0.1, barrier, max-over-ranks timing and the field all-gather; one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--launcher-selftest", "--backend", "gloo"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world"] == 2 and rec["collective_ranks"] == 2 and rec["backend"] == "gloo" and rec["gather_ok"]


def test_bench_dry_run_gloo_world2():
    """`python bench.py --gpus 2 --dry-run`: everything an N-GPU bench run does AROUND its kernels -- self-launch, rendezvous,
    per-rank core pinning and synthetic set-up, the timing protocol of the U-Net and both MPM legs, the all-gather timed on its
    own, and the assembly of the full JSON line by the same code as the real run -- on CPU tensors (the first
    8-GPU run must not die in code the CPU suite never executed)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["dry_run"] and rec["n_gpus"] == 2 and rec["collective_ranks"] == 2 and rec["backend"] == "gloo"
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "allgather_ms", "compute_ms_per_step", "host_threads_per_rank", "mpm", "mpm_1m", "exact_f32"):
        assert key in rec, key
    assert rec["allgather_ms"] > 0 and rec["compute_ms_per_step"] < rec["ms_per_step"]
    assert rec["scaling"] == "weak" and rec["config"]["parallelism"] == "scene-parallel x2"
    assert rec["host_threads_per_rank"] >= 1


@pytest.mark.gpu
def test_rccl_process_group_and_graph_capture_coexist(hip_device):
    """A live RCCL process group (communicator, watchdog thread) while the U-Net forward is captured into HIP graphs and
    replayed between collectives -- the order of operations of `bench.py --gpus N` on each rank.  World size 1 (one GPU on the
    test box): what is exercised is RCCL and graph capture sharing a process, not the wire."""
    import socket

    import torch.distributed as dist
    from pixie_amd.synthetic import feature_grid
    from pixie_amd.unet import RegressionUNet, SegmentationUNet, predict_material_field
    from pixie_amd.unet_plan import synthetic_state_dict
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        t = torch.ones(8, device=hip_device)
        dist.all_reduce(t)                                        # creates the communicator and starts the watchdog
        kw = dict(feature_channels=64, cond_dim=32, model_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions=(), grid_size=16)
        seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
        seg.load_numpy_state(synthetic_state_dict(seg.cfg, 3)); cont.load_numpy_state(synthetic_state_dict(cont.cfg, 4))
        seg, cont = seg.to(hip_device).eval(), cont.to(hip_device).eval()
        feat = torch.from_numpy(feature_grid(16, 64, seed=5)).to(hip_device)
        seg.use_graph = cont.use_graph = False
        want = predict_material_field(seg, cont, feat)[0]
        seg.use_graph = cont.use_graph = True
        for _ in range(3):                                        # capture (first pass), then replays, a collective after each
            combined, seg_pred, _, cont_pred = predict_material_field(seg, cont, feat)
            gathered = torch.empty_like(cont_pred)
            dist.all_gather_into_tensor(gathered, cont_pred.contiguous())
            dist.barrier()
            assert torch.equal(combined, want) and torch.equal(gathered, cont_pred)
        assert seg.use_graph and cont.use_graph                   # the capture did not fall back to eager launches
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_wire_path_executes_on_one_rank(hip_device):
    """no box of any round had two GPUs, so the device branch of the exchange -- pixie_pack_fields -> uint8
    dist.all_gather_into_tensor under backend "nccl" (= RCCL) -> 16-byte-aligned unpack -- had never executed, not even with one rank
    (all_gather_fields short-circuits at world 1).  `force_wire` sends a one-rank group through it, at the bench's 128^3 (and an odd
    size, whose float block ends off a 16-byte boundary): the gathered fields must equal the inputs bit for bit.  Not a scaling
    number; the proof that the call, dtype and alignment path run on RCCL."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        for n, shape in ((1, (128, 128, 128)), (2, (5, 7, 9))):
            g = torch.Generator().manual_seed(n)
            cont = torch.randn((n, 3) + shape, generator=g).to(hip_device)
            seg = torch.randint(0, 8, (n,) + shape, generator=g, dtype=torch.int32).to(hip_device)
            g_cont, g_seg = pd.all_gather_fields(cont, seg, force_wire=True)
            torch.cuda.synchronize()
            assert g_cont.data_ptr() != cont.data_ptr() and g_cont.is_cuda and g_seg.dtype == torch.uint8     # went through the wire buffer
            assert torch.equal(g_cont, cont) and torch.equal(g_seg.to(torch.int32), seg)
    finally:
        dist.destroy_process_group()


def test_wire_format_round_trip_single_process():
    """pack -> (what all_gather_into_tensor does: concatenate the ranks' buffers) -> unpack, odd voxel counts included
    (the float32 block of every rank must stay 4-byte aligned inside the gathered buffer: 16-byte padding)."""
    for spatial in ((4, 4, 4), (3, 5, 7), (1, 1, 1)):
        for n in (1, 2):
            ranks = []
            for r in range(3):
                g = torch.Generator().manual_seed(10 * r + n)
                ranks.append((torch.randn((n, 3) + spatial, generator=g), torch.randint(0, 8, (n,) + spatial, generator=g)))
            bufs = [pd.pack_fields(c, s) for c, s in ranks]
            assert all(b.numel() == pd.wire_bytes(n, int(np.prod(spatial))) and b.numel() % 16 == 0 for b in bufs)
            cont, seg = pd.unpack_fields(torch.cat(bufs), 3, n, spatial)
            assert torch.equal(cont, torch.cat([c for c, _ in ranks])) and torch.equal(seg.long(), torch.cat([s for _, s in ranks]))


def _rccl_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    pd.init_process_group("nccl")
    dev = torch.device("cuda", rank)
    D = 32
    cont = torch.full((1, 3, D, D, D), float(rank + 1), device=dev) + torch.arange(3.0, device=dev)[None, :, None, None, None]
    seg = torch.full((1, D, D, D), (rank + 3) % 8, dtype=torch.int32, device=dev)
    g_cont, g_seg = pd.all_gather_fields(cont, seg)
    torch.cuda.synchronize(dev)
    ok = g_cont.shape == (world, 3, D, D, D) and g_seg.dtype == torch.uint8
    for r in range(world):
        ok = ok and float(g_cont[r, 2, 1, 1, 1]) == r + 3.0 and int(g_seg[r, 5, 5, 5]) == (r + 3) % 8
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_all_gather_fields_rccl_two_ranks():
    """The packed all-gather over RCCL between two GPUs.  The test boxes of rounds 1-4 had ONE GPU, so this skips there; the
    first multi-GPU box exercises the wire automatically."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box has {torch.cuda.device_count()}")
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rccl_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


@pytest.mark.gpu
@pytest.mark.parametrize("n,shape", [(1, (16, 16, 16)), (2, (8, 12, 20)), (3, (5, 7, 9)), (1, (1, 1, 3))])
def test_pack_kernel_writes_the_wire_format_of_the_host_path(hip_device, n, shape):
    """pixie_pack_fields (one launch, device tensors) against the torch packing the gloo tests use (host tensors): same bytes,
    incl. voxel counts that are not a multiple of 4 and the zero pad; and the round trip through unpack_fields."""
    import torch
    from pixie_amd import distributed as pd
    g = torch.Generator().manual_seed(n + shape[2])
    cont = torch.randn((n, 3) + shape, generator=g)
    seg = torch.randint(0, 8, (n,) + shape, generator=g, dtype=torch.int32)
    want = pd.pack_fields(cont, seg)
    got = pd.pack_fields(cont.to(hip_device), seg.to(hip_device))
    assert got.is_cuda and got.dtype == torch.uint8 and got.numel() == pd.wire_bytes(n, seg[0].numel()) == want.numel()
    assert torch.equal(got.cpu(), want)
    c2, s2 = pd.unpack_fields(got, 1, n, shape)
    assert torch.equal(c2.cpu(), cont) and torch.equal(s2.cpu().to(torch.int32), seg)
    if n > 1:     # a contiguous view at an offset that is not a multiple of 16 bytes (odd voxel count): copied before the 16-byte moves
        dc, ds = cont.to(hip_device), seg.to(hip_device)
        assert torch.equal(pd.pack_fields(dc[1:], ds[1:]).cpu(), pd.pack_fields(cont[1:], seg[1:]))


def test_bench_line_of_a_full_run_fits_the_drivers_record():
    """The compact line bench.py prints (the driver keeps ~6 KB of it) rendered from the committed full record of the closing
    session: every contract key is there, `roofline` and `cpu_baseline` carry their fields, the plastic legs are top-level scalars,
    and the whole line stays under 6 KB."""
    import importlib.util
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(repo, "profiles", "bench_detail_r6last_full.json")   # (the --full record: every leg)
    if not os.path.exists(path):
        pytest.skip("no committed bench detail")
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    line = bench.compact_line(json.load(open(path)), "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 6144, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["vs_baseline"] is None and "workload" in line["config"]
    for leg in ("sand", "snow", "metal", "mixed"):
        for f in ("us_per_substep", "frac_dense", "frac_touched", "valu_per_wave", "block_us_rocprofv3"):
            if leg == "sand" and f == "frac_dense":      # n_grid 200: the dense-grid figure counts 8 M mostly empty cells and is not quoted
                assert line["mpm_1m_sand_frac_dense"] is None
                continue
            assert isinstance(line[f"mpm_1m_{leg}_{f}"], (int, float)), (leg, f)
    for k in ("mpm_floor_us", "mpm_frac_of_floor", "pipeline_ms_per_scene", "mpm_1m_frac_touched", "mpm_1m_frac_dense"):
        assert isinstance(line[k], (int, float)), k
