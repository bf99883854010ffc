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


def pack_fields(cont_pred: torch.Tensor, seg_pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(n, 3, D, H, W) fp32, (n, D, H, W) int -> wire tensors (fp32, uint8), contiguous."""
    return cont_pred.contiguous().to(torch.float32), seg_pred.contiguous().to(torch.uint8)


def all_gather_fields(cont_pred: torch.Tensor, seg_pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the local scenes' fields; returns tensors ordered rank-major:
    (world * n_local, 3, D, H, W) fp32 and (world * n_local, D, H, W) uint8.
    With world == 1 (or no process group) it returns the inputs unchanged."""
    cont, seg = pack_fields(cont_pred, seg_pred)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return cont, seg
    world = dist.get_world_size()
    cont_out = torch.empty((world * cont.shape[0],) + tuple(cont.shape[1:]), dtype=cont.dtype, device=cont.device)
    seg_out = torch.empty((world * seg.shape[0],) + tuple(seg.shape[1:]), dtype=seg.dtype, device=seg.device)
    dist.all_gather_into_tensor(cont_out, cont)  # concatenation along dim 0, rank-major
    dist.all_gather_into_tensor(seg_out, seg)
    return cont_out, seg_out


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
