"""Predicted material field -> MPM particles on the device.

Drop-in for the middle of the reference pipeline: `unscale_prediction` + the masked voxel point list of
pixie/voxel/map_pred_to_coords.py:41-75,192-252 and `perform_knn_smoothing` /
`_apply_material_properties_to_solver` of third_party/PhysGaussian/material_field.py:228-363, without the PLY file,
sklearn and the Python loops over every particle: one HIP launch (csrc/field_transfer.hip) produces the per-particle
density / E / nu / material id / part label / confidence, and `set_per_particle` uploads them to the solver.
There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import FieldDesc, check

# normalization_stats/normalization_ranges.yaml (the file load_normalization_ranges reads, pixie/training_utils.py)
NORMALIZATION_RANGES = {
    "density_min": 1.7031893730163574, "density_max": 3.871432304382324,
    "E_min": 3.0183002948760986, "E_max": 10.881680488586426,
    "nu_min": 0.21027633547782898, "nu_max": 0.4492689371109009,
}
STATIONARY_ID = 6      # get_material_name("stationary")
DEFAULT_PART_LABEL = 0


def field_to_particles(pred: torch.Tensor, mask: torch.Tensor, min_bounds: Sequence[float], max_bounds: Sequence[float],
                       particle_pos: torch.Tensor, k: int = 10, nn_distance_threshold: float = 0.1, weighted: bool = False,
                       ranges: Optional[Dict[str, float]] = None) -> Dict[str, torch.Tensor]:
    """pred: (3 + n_classes, D, H, W) float32 network output (e.g. predict_material_field(...)[0][i]); mask: (D, H, W)
    occupancy; particle_pos: (n, 3) float32 in the field's coordinate frame (the caller undoes the simulation's
    shift / scale / rotation exactly like transform_to_original_coordinates, material_field.py:81-87).
    Returns device tensors: density, E, nu, conf, nearest_dist (float32), material_id, part_labels (int32),
    n_too_far (0-d int64, device)."""
    if pred.device.type != "cuda":
        raise _lib.PixieHipError("field_to_particles runs on a HIP device only (no CPU fallback)")
    lib = _lib.load()
    dev = pred.device
    ranges = ranges or NORMALIZATION_RANGES
    pred = pred.detach().to(torch.float32).contiguous()
    ncls = pred.shape[0] - 3
    D, H, W = (int(s) for s in pred.shape[1:])
    mask8 = (mask.to(dev) > 0).to(torch.uint8).contiguous()
    axes, spacing = [], []
    for a, n in enumerate((D, H, W)):
        lin = np.linspace(float(min_bounds[a]), float(max_bounds[a]), n)   # float64, as map_pred_to_ply builds it
        axes.append(torch.from_numpy(lin.astype(np.float32)).to(dev))      # the PLY stores 'f4'
        spacing.append(abs(float(lin[1] - lin[0])) if n > 1 else float("inf"))
    pos = particle_pos.detach().to(dev, torch.float32).contiguous()
    n = pos.shape[0]
    out = {key: torch.empty(n, dtype=torch.float32, device=dev) for key in ("density", "E", "nu", "conf", "nearest_dist")}
    out["material_id"] = torch.empty(n, dtype=torch.int32, device=dev)
    out["part_labels"] = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(8, dtype=torch.float64, device=dev)
    f = FieldDesc()
    f.d_pred, f.d_mask = pred.data_ptr(), mask8.data_ptr()
    f.d_axis_x, f.d_axis_y, f.d_axis_z = (t.data_ptr() for t in axes)
    f.n_classes, f.d, f.h, f.w = ncls, D, H, W
    f.min_spacing = min(spacing)
    for key in ("density_min", "density_max", "E_min", "E_max", "nu_min", "nu_max"):
        setattr(f, key, float(ranges[key]))
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.pixie_field_to_particles(C.byref(f), p(pos), n, int(k), float(nn_distance_threshold), int(bool(weighted)),
                                       STATIONARY_ID, DEFAULT_PART_LABEL, p(out["density"]), p(out["E"]), p(out["nu"]),
                                       p(out["material_id"]), p(out["part_labels"]), p(out["conf"]), p(out["nearest_dist"]),
                                       p(scratch), _lib.current_stream_ptr()), "pixie_field_to_particles")
    out["n_too_far"] = scratch[5:6].view(torch.int64)[0]
    return out


def apply_material_field_to_solver(mpm_solver, pred: torch.Tensor, mask: torch.Tensor, min_bounds, max_bounds,
                                   particle_pos_field_frame: torch.Tensor, k_smoothing_neighbors: int = 10,
                                   nn_distance_threshold: float = 0.1, weighted_assignment: bool = False):
    """The device-resident equivalent of apply_material_field_to_simulation's property hand-off
    (material_field.py:303-363, minus the ground / stationary-cluster boundary conditions, which stay with the
    caller): K-NN transfer, the reference's 10 % too-far assertion (:275), then one array upload per property instead of
    one apply_additional_params launch per particle.  Returns the per-particle confidence."""
    res = field_to_particles(pred, mask, min_bounds, max_bounds, particle_pos_field_frame, k_smoothing_neighbors,
                             nn_distance_threshold, weighted_assignment)
    n = particle_pos_field_frame.shape[0]
    n_far = int(res["n_too_far"])
    assert n_far <= 0.1 * n, (f"[CRITICAL] More than 10% of particles are too far from nearest neighbor. "
                              f"Distance threshold: {nn_distance_threshold}.")
    mpm_solver.set_per_particle(E=res["E"], nu=res["nu"], density=res["density"], material=res["material_id"])
    mpm_solver.finalize_mu_lam()
    return res["conf"]
