"""The file contracts between the two halves of the pipeline, with the reference's function names and signatures.

Mirrors pixie/voxel/map_pred_to_coords.py (`unscale_prediction` :41-75, `transform_nerf_to_world` :77-120, `get_mat_id`
:122-126, `map_pred_to_ply` :128-283) and `save_predictions` of third_party/Wavelet-Generation/trainer/
inference_combined.py:173-217, so that pixie/utils.py:736,763-779 can run the swapped-import programs end to end:

    inference_combined.py  --sample_{id}_{pred,gt,mask,info}.npy-->  map_pred_to_ply  --PLY-->  gs_simulation.py

Arithmetic (un-scaling, argmax / confidence, the masked voxel list, the one-hot combine) runs in libpixie_hip.so
(pixie_unscale_prediction, pixie_field_points, pixie_combine_predictions); this file moves arrays between files and the
device.  There is no CPU path: without a HIP device these functions raise.  The device-resident route that skips the files
altogether is pixie_amd.material_field.apply_material_field_to_solver.
"""
from __future__ import annotations

import ctypes as C
import json
import logging
import os
from pathlib import Path
from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import FieldDesc, check
from .material_field import NORMALIZATION_RANGES
from .ply_io import read_ply, write_ply

# the vertex layout map_pred_to_ply writes (map_pred_to_coords.py:224-233)
PLY_VERTEX_DTYPE = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1"),
                    ("part_label", "i4"), ("density", "f4"), ("E", "f4"), ("nu", "f4"), ("material_id", "i4"), ("conf", "f4")]
_RANGE_KEYS = ("density_min", "density_max", "E_min", "E_max", "nu_min", "nu_max")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.PixieHipError("pixie_amd.field_mapping needs a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _ranges(cfg) -> Dict[str, float]:
    """cfg.training.{density,E,nu}_{min,max} as load_normalization_ranges leaves them (pixie/training_utils.py); a plain
    dict with those keys, or None for the shipped normalization_stats/normalization_ranges.yaml values."""
    if cfg is None:
        return dict(NORMALIZATION_RANGES)
    src = cfg
    if not isinstance(cfg, dict) or "training" in cfg:
        src = cfg["training"] if isinstance(cfg, dict) else getattr(cfg, "training")
    get = (lambda k: src[k]) if isinstance(src, dict) else (lambda k: getattr(src, k))
    return {k: float(get(k)) for k in _RANGE_KEYS}


def unscale_prediction(pred_tensor, cfg=None):
    """map_pred_to_coords.py:41-75.  (3 + n_classes, D, H, W) network-unit prediction -> same layout with density, E
    (10^log) and nu in physical units; class channels untouched.  numpy in -> numpy out (as the reference); a device
    tensor in -> a device tensor out (no host round trip)."""
    dev = _device()
    r = _ranges(cfg)
    is_np = isinstance(pred_tensor, np.ndarray)
    t = (torch.from_numpy(np.ascontiguousarray(pred_tensor, dtype=np.float32)) if is_np else pred_tensor.detach().float()).to(dev).contiguous()
    out = torch.empty_like(t)
    spatial = t[0].numel()
    check(_lib.load().pixie_unscale_prediction(C.c_void_p(t.data_ptr()), int(t.shape[0]), spatial, r["density_min"], r["density_max"],
                                               r["E_min"], r["E_max"], r["nu_min"], r["nu_max"], C.c_void_p(out.data_ptr()),
                                               _lib.current_stream_ptr()), "pixie_unscale_prediction")
    return out.cpu().numpy() if is_np else out


def get_mat_id(arr):
    """map_pred_to_coords.py:122-126: a single class channel is the class index itself, otherwise argmax over channels."""
    if arr.shape[0] == 1:
        return arr[0]
    return arr.argmax(0) if isinstance(arr, np.ndarray) else torch.argmax(arr, dim=0)


def voxel_points(pred, mask, min_bounds: Sequence[float], max_bounds: Sequence[float], cfg=None) -> Dict[str, torch.Tensor]:
    """The masked voxel point list of map_pred_to_ply (:192-252) as device tensors, in C order of the grid:
    xyz (n,3), density, E, nu, conf (n,) float32, material_id (n,) int32.  `pred` is the NETWORK-unit (3 + n_classes, D, H, W)
    prediction (the un-scaling happens in the kernel)."""
    dev = _device()
    lib = _lib.load()
    r = _ranges(cfg)
    t = (torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)) if isinstance(pred, np.ndarray) else pred.detach().float()).to(dev).contiguous()
    m = torch.from_numpy(np.ascontiguousarray(mask)) if isinstance(mask, np.ndarray) else mask
    mask8 = (m.to(dev) > 0).to(torch.uint8).contiguous()
    D, H, W = (int(s) for s in t.shape[1:])
    if tuple(mask8.shape) != (D, H, W):
        raise ValueError(f"Mask shape {tuple(mask8.shape)} does not match grid shape {(D, H, W)}")
    axes = [torch.from_numpy(np.linspace(float(min_bounds[a]), float(max_bounds[a]), n).astype(np.float32)).to(dev)
            for a, n in enumerate((D, H, W))]   # np.linspace in float64, stored as 'f4' (:192-196, :236-238)
    f = FieldDesc()
    f.d_pred, f.d_mask = t.data_ptr(), mask8.data_ptr()
    f.d_axis_x, f.d_axis_y, f.d_axis_z = (a.data_ptr() for a in axes)
    f.n_classes, f.d, f.h, f.w = int(t.shape[0]) - 3, D, H, W
    for k in _RANGE_KEYS:
        setattr(f, k, r[k])
    scratch = torch.empty(max(int(lib.pixie_field_points_scratch_bytes(C.byref(f))), 8), dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    st = _lib.current_stream_ptr()
    p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)
    check(lib.pixie_field_points(C.byref(f), 0, None, None, None, None, None, None, p(count), p(scratch), st), "pixie_field_points(count)")
    n = int(count.item())   # the file needs the length: one host read-back per scene
    out = {"xyz": torch.empty((n, 3), dtype=torch.float32, device=dev), "material_id": torch.empty(n, dtype=torch.int32, device=dev)}
    for k in ("density", "E", "nu", "conf"):
        out[k] = torch.empty(n, dtype=torch.float32, device=dev)
    if n:
        check(lib.pixie_field_points(C.byref(f), n, p(out["xyz"]), p(out["density"]), p(out["E"]), p(out["nu"]), p(out["material_id"]),
                                     p(out["conf"]), p(count), p(scratch), st), "pixie_field_points")
    return out


def map_pred_to_ply(pred_path: str, mask_path: str, grid_feature_path: str, output_path: str, obj_id: str,
                    world_output_path: str = None, dataparser_path: str = None, cfg=None):
    """map_pred_to_coords.py:128-283: sample_*_pred.npy (+ mask, + the voxel grid's bounds) -> the material-field PLY that
    gs_simulation.py's load_point_cloud reads; optionally also in the world frame."""
    metadata = np.load(grid_feature_path)
    min_bounds, max_bounds, grid_shape = metadata["min_bounds"], metadata["max_bounds"], metadata["grid_shape"]
    logging.info(f"Grid shape: {grid_shape}")
    logging.info(f"Bounds: min={min_bounds}, max={max_bounds}")
    scaled_pred = np.load(pred_path)
    mask = np.load(mask_path)
    if not np.array_equal(scaled_pred.shape[1:4], grid_shape):
        raise ValueError(f"Prediction spatial dimensions {scaled_pred.shape[1:4]} do not match grid shape {grid_shape}")
    if not np.array_equal(mask.shape, grid_shape):
        raise ValueError(f"Mask shape {mask.shape} does not match grid shape {grid_shape}")
    pts = voxel_points(scaled_pred, mask, min_bounds, max_bounds, cfg)
    n = pts["xyz"].shape[0]
    vertex_data = np.zeros(n, dtype=PLY_VERTEX_DTYPE)
    xyz = pts["xyz"].cpu().numpy()
    vertex_data["x"], vertex_data["y"], vertex_data["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    for c in ("red", "green", "blue", "alpha"):
        vertex_data[c] = 255
    mat = pts["material_id"].cpu().numpy()
    vertex_data["part_label"] = mat
    vertex_data["material_id"] = mat
    for k in ("density", "E", "nu", "conf"):
        vertex_data[k] = pts[k].cpu().numpy()
    write_ply(output_path, vertex_data, text=False)
    logging.info(f"Saved PLY file to {output_path} from {pred_path}")
    if world_output_path is not None:
        if dataparser_path is None:
            dataparser_path = Path(grid_feature_path).parent / "dataparser_transforms.json"
            if not dataparser_path.exists():
                raise FileNotFoundError(f"Could not find dataparser_transforms.json at {dataparser_path}. "
                                        "Please provide the path using --dataparser_path argument.")
        transform_nerf_to_world(output_path, dataparser_path, world_output_path)


def transform_nerf_to_world(ply_path: str, dataparser_path: str, world_output_path: str):
    """map_pred_to_coords.py:77-120: take the PLY's vertices from nerfstudio's training frame back to the world frame.
    The dataparser maps world -> training as  q = s * (A p + t)  (its 3x4 `transform` is [A | t], `scale` is s), so the way
    back is  p = A^-1 (q / s - t): a 3x3 inverse and one affine map over the (n, 3) vertex array, on the host."""
    verts, _ = read_ply(str(ply_path))
    with open(dataparser_path, "r") as fh:
        parser = json.load(fh)
    affine = np.asarray(parser["transform"], dtype=np.float64).reshape(3, 4)
    lin_inv = np.linalg.inv(affine[:, :3])
    q = np.stack([np.asarray(verts[ax], dtype=np.float64) for ax in ("x", "y", "z")], axis=1)
    p = (q / float(parser["scale"]) - affine[:, 3]) @ lin_inv.T
    moved = verts.copy()
    for col, ax in enumerate(("x", "y", "z")):
        moved[ax] = p[:, col].astype(np.float32)
    write_ply(str(world_output_path), moved, text=False)
    logging.info(f"Saved WORLD-frame PLY to {world_output_path}")


def load_material_points(ply_path: str, device=None) -> Dict[str, object]:
    """The material-field part of load_point_cloud (PhysGaussian/gs_simulation.py:108-204): positions on the device, the
    per-point properties as numpy arrays under the keys perform_knn_smoothing / extract_material_properties expect."""
    vertex, _ = read_ply(ply_path)
    names = vertex.dtype.names
    dev = device or _device()
    pos = torch.from_numpy(np.column_stack((vertex["x"], vertex["y"], vertex["z"])).astype(np.float32)).to(dev)
    conf = vertex["conf"].astype(np.float32) if "conf" in names else np.ones(len(vertex), dtype=np.float32)
    part = vertex["part_label"] if "part_label" in names else vertex["material_id"]
    return {"pos": pos, "part_labels": part, "density": vertex["density"], "E": vertex["E"], "nu": vertex["nu"],
            "material_id": vertex["material_id"], "conf": conf}


def save_predictions(config, output_dir, batch_idx, obj_id, info_batch, seg_pred, cont_pred, gt_tensor, feat_tensor, mask, D):
    """inference_combined.py:173-217: writes <output_dir>/<obj_id>/sample_{id}_{pred,gt,mask,info}.npy; `pred` is the
    (3 + num_material_classes, D, D, D) float32 tensor [cont(3) ; one-hot(seg_pred)] assembled on the device
    (pixie_combine_class_ids)."""
    sample_id = info_batch["sample_id"][batch_idx]
    if isinstance(sample_id, torch.Tensor):
        sample_id = str(sample_id.item() if sample_id.numel() == 1 else sample_id[0].item())
    else:
        sample_id = str(sample_id)
    obj_out_dir = os.path.join(output_dir, obj_id)
    os.makedirs(obj_out_dir, exist_ok=True)
    get = (lambda o, k: o[k]) if isinstance(config, dict) else getattr
    n_classes = int(get(get(config, "training"), "num_material_classes"))
    dev = cont_pred.device
    if dev.type != "cuda":
        raise _lib.PixieHipError("save_predictions assembles the prediction on a HIP device (no CPU fallback)")
    seg = seg_pred.to(dev).to(torch.int32).contiguous()
    cont = cont_pred.detach().to(torch.float32).contiguous()
    combined = torch.empty((3 + n_classes, D, D, D), dtype=torch.float32, device=dev)
    check(_lib.load().pixie_combine_class_ids(C.c_void_p(seg.data_ptr()), n_classes, C.c_void_p(cont.data_ptr()), D * D * D,
                                              C.c_void_p(combined.data_ptr()), _lib.current_stream_ptr()), "pixie_combine_class_ids")
    np.save(os.path.join(obj_out_dir, f"sample_{sample_id}_pred.npy"), combined.cpu().numpy())
    np.save(os.path.join(obj_out_dir, f"sample_{sample_id}_gt.npy"), gt_tensor.cpu().numpy())
    np.save(os.path.join(obj_out_dir, f"sample_{sample_id}_mask.npy"), mask.cpu().numpy())
    info_to_save = {"obj_id": obj_id, "sample_id": sample_id, "data_path": info_batch["data_path"][batch_idx],
                    "feature_path": info_batch["feature_path"][batch_idx], "mask_path": info_batch["mask_path"][batch_idx]}
    np.save(os.path.join(obj_out_dir, f"sample_{sample_id}_info.npy"), info_to_save)
