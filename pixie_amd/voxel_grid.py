"""Voxel-grid input format of the hot path on the device.

The reference stores the voxelised feature grid as `clip_features_features.npy`, shape (D, H, W, C) float16
(pixie/voxel/voxelize.py:86,111,144-185) and its dataset item turns it into the network input with
`.astype(np.float32)` + `permute(3, 0, 1, 2)` on the CPU (WG/data_utils/my_data.py:160-224).  `load_voxel_grid` does that
cast + transpose in one HIP kernel on the uploaded float16 array (half the PCIe bytes of the float32 grid).
There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def load_voxel_grid(feat_dhwc, device="cuda:0") -> torch.Tensor:
    """(D, H, W, C) float16 array / tensor (or a path to the .npy) -> (1, C, D, H, W) float32 tensor on `device`,
    bit-identical to torch.from_numpy(feat.astype(np.float32)).permute(3, 0, 1, 2)[None]."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.PixieHipError("load_voxel_grid runs on a HIP device only (no CPU fallback)")
    if isinstance(feat_dhwc, str):
        feat_dhwc = np.load(feat_dhwc)
    if isinstance(feat_dhwc, np.ndarray):
        feat_dhwc = torch.from_numpy(np.ascontiguousarray(feat_dhwc))
    if feat_dhwc.dim() == 3:   # occupancy features: (D,D,D) -> (D,D,D,1), my_data.py:166-167
        feat_dhwc = feat_dhwc[..., None]
    if feat_dhwc.dtype != torch.float16:
        # the reference accepts any dtype through .astype(float32); float16 is what voxelize.py writes
        return feat_dhwc.to(dev, torch.float32).permute(3, 0, 1, 2).contiguous()[None]
    src = feat_dhwc.to(dev).contiguous()
    d, h, w, c = (int(s) for s in src.shape)
    out = torch.empty((1, c, d, h, w), dtype=torch.float32, device=dev)
    check(_lib.load().pixie_voxel_grid_to_ncdhw(C.c_void_p(src.data_ptr()), d, h, w, c, C.c_void_p(out.data_ptr()),
                                                _lib.current_stream_ptr()), "pixie_voxel_grid_to_ncdhw")
    return out
