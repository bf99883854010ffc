"""Scene-parallel execution across the GPUs of one node.

The reference shards scenes with DistributedSampler(shuffle=False), one process per GPU, and every
rank writes its own files -- no tensor collectives (trainer/inference_combined.py:247-256, :335-353).
Scenes are independent, so the data path here has no collective either; the only exchange is ONE
all-gather of the predicted per-voxel fields after the U-Net stage, in a compact wire format
(3 x fp32 continuous channels + 1 x uint8 class id per voxel = 13 B/voxel instead of the 44 B/voxel
of the reference's (11, D, D, D) float file), issued through torch.distributed (backend "nccl" is RCCL
on ROCm; "gloo" for the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend: str | None = None) -> Tuple[int, int, int]:
    """One process per GPU.  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Indices DistributedSampler(dataset, shuffle=False, drop_last=False) gives `rank`
    (torch/utils/data/distributed.py semantics: pad by wrapping so every rank gets ceil(n/world))."""
    if n_scenes == 0:
        return []
    per = (n_scenes + world - 1) // world
    total = per * world
    idx = list(range(n_scenes))
    while len(idx) < total:
        idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def wire_bytes(n_local: int, voxels: int) -> int:
    """Bytes one rank contributes: n_local scenes x (3 fp32 + 1 uint8) per voxel = 13 B/voxel, padded to 16 B so that every
    rank's float32 block starts aligned inside the gathered buffer."""
    return (13 * n_local * voxels + 15) // 16 * 16


def pack_fields(cont_pred: torch.Tensor, seg_pred: torch.Tensor) -> torch.Tensor:
    """(n, 3, D, H, W) fp32 + (n, D, H, W) integer class ids -> ONE uint8 wire buffer: [n * 3 * V floats | n * V bytes | pad].
    Device tensors: one launch of pixie_pack_fields (csrc/unet_ops.hip; three torch copy kernels until round 4).  Host tensors --
    the gloo tests and `bench.py --dry-run`, which exercise the N-rank control path without a device -- are packed with torch."""
    n, vox = cont_pred.shape[0], seg_pred[0].numel() if seg_pred.shape[0] else 0
    buf = torch.empty(wire_bytes(n, vox), dtype=torch.uint8, device=cont_pred.device)
    if cont_pred.is_cuda and n * vox > 0:
        import ctypes as C
        from . import _lib
        cont = cont_pred.detach().to(torch.float32).contiguous()
        seg = seg_pred.to(torch.int32).contiguous()
        if cont.data_ptr() % 16:       # the kernel moves 16-byte words: a contiguous VIEW at an odd offset is copied first
            cont = cont.clone()
        if seg.data_ptr() % 16:
            seg = seg.clone()
        _lib.check(_lib.load().pixie_pack_fields(C.c_void_p(cont.data_ptr()), C.c_void_p(seg.data_ptr()), n, vox, C.c_void_p(buf.data_ptr()),
                                                 buf.numel(), _lib.current_stream_ptr()), "pixie_pack_fields")
        return buf
    nf = 12 * n * vox
    buf[:nf].view(torch.float32).copy_(cont_pred.reshape(-1))          # dtype conversion (if any) rides in the copy
    buf[nf:nf + n * vox].copy_(seg_pred.reshape(-1))
    if buf.numel() > nf + n * vox:
        buf[nf + n * vox:].zero_()
    return buf


def unpack_fields(buf: torch.Tensor, world: int, n_local: int, spatial: Tuple[int, ...]) -> Tuple[torch.Tensor, torch.Tensor]:
    """The gathered wire buffer (world x wire_bytes) -> (world * n_local, 3, D, H, W) fp32, (world * n_local, D, H, W) uint8, rank-major."""
    vox = 1
    for d in spatial:
        vox *= int(d)
    nf = 12 * n_local * vox
    rows = buf.view(world, -1)
    cont = rows[:, :nf].view(torch.float32).reshape((world * n_local, 3) + tuple(spatial))
    seg = rows[:, nf:nf + n_local * vox].reshape((world * n_local,) + tuple(spatial))
    return cont, seg


def all_gather_fields(cont_pred: torch.Tensor, seg_pred: torch.Tensor, force_wire: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the local scenes' fields with ONE collective on the packed 13 B/voxel buffer (two
    collectives -- fp32, then uint8 -- until round 3).  Returns tensors ordered rank-major: (world * n_local, 3, D, H, W) fp32
    and (world * n_local, D, H, W) uint8.  With world == 1 (or no process group) nothing is packed or copied -- unless
    `force_wire`, which sends a one-rank group through pack kernel -> all_gather_into_tensor (uint8) -> unpack all the same (the
    one-GPU test boxes' proof that the RCCL call, its dtype and the 16-byte alignment path are live)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_wire):
        return cont_pred.contiguous().to(torch.float32), seg_pred.contiguous().to(torch.uint8)
    world = dist.get_world_size()
    mine = pack_fields(cont_pred, seg_pred)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine)        # concatenation along dim 0, rank-major
    return unpack_fields(out, world, cont_pred.shape[0], tuple(seg_pred.shape[1:]))


def unshard_order(n_scenes: int, world: int) -> List[int]:
    """Position in the gathered (rank-major) tensors of scene i, i = 0..n_scenes-1."""
    per = (n_scenes + world - 1) // world
    pos = {}
    for r in range(world):
        for j, s in enumerate(shard_scenes(n_scenes, r, world)):
            pos.setdefault(s, r * per + j)
    return [pos[i] for i in range(n_scenes)]


def combined_from_wire(cont: torch.Tensor, seg: torch.Tensor, num_classes: int = 8) -> torch.Tensor:
    """Wire format -> the reference's (n, 3 + num_classes, D, H, W) float32 prediction tensor
    (trainer/inference_combined.py:186-195)."""
    onehot = torch.nn.functional.one_hot(seg.long(), num_classes).permute(0, 4, 1, 2, 3).to(torch.float32)
    return torch.cat([cont, onehot], dim=1)
