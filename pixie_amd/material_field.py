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
    out["n_occupied_voxels"] = scratch[4]      # (0-d float64, device) how many voxels the mask keeps: fewer than k => every particle got the defaults
    return out


def apply_material_field_to_solver(mpm_solver, pred: torch.Tensor, mask: torch.Tensor, min_bounds, max_bounds,
                                   particle_pos_field_frame: torch.Tensor, k_smoothing_neighbors: int = 10,
                                   nn_distance_threshold: float = 0.1, weighted_assignment: bool = False,
                                   ranges: Optional[Dict[str, float]] = None):
    """The device-resident equivalent of apply_material_field_to_simulation's property hand-off
    (material_field.py:303-363, minus the ground / stationary-cluster boundary conditions, which stay with the
    caller): K-NN transfer, the reference's 10 % too-far assertion (:275), then one array upload per property instead of
    one apply_additional_params launch per particle.  Returns the per-particle confidence."""
    res = field_to_particles(pred, mask, min_bounds, max_bounds, particle_pos_field_frame, k_smoothing_neighbors,
                             nn_distance_threshold, weighted_assignment, ranges=ranges)   # ranges: the dataset's normalization_ranges.yaml (default: the shipped one)
    n = particle_pos_field_frame.shape[0]
    n_vox = int(res["n_occupied_voxels"])
    if n_vox < int(k_smoothing_neighbors):     # what sklearn's NearestNeighbors.kneighbors raises inside perform_knn_smoothing (material_field.py:228-300)
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {int(k_smoothing_neighbors)}, n_samples_fit = {n_vox}, "
                         f"n_samples = {n}")
    n_far = int(res["n_too_far"])
    assert n_far <= 0.1 * n, (f"[CRITICAL] More than 10% of particles are too far from nearest neighbor. "
                              f"Distance threshold: {nn_distance_threshold}.")
    mpm_solver.set_per_particle(E=res["E"], nu=res["nu"], density=res["density"], material=res["material_id"])
    mpm_solver.finalize_mu_lam()
    return res["conf"]


# ----------------------------------------------------------------------------------------------------------------------
# Boundary conditions derived from the transferred field (third_party/PhysGaussian/material_field.py:364-550).
def dbscan_labels(points, eps: float, min_samples: int) -> torch.Tensor:
    """Labels as sklearn.cluster.DBSCAN(eps, min_samples).fit_predict assigns them (the reference's call,
    material_field.py:405-406), computed on the device (csrc/field_transfer.hip: pixie_dbscan_roots): a point is a core
    point when its closed eps-ball holds >= min_samples points; clusters are the connected components of the core points
    and are numbered in the order of their lowest-index core point; a border point joins the earliest-numbered cluster that
    has a core point within eps; everything else is noise (-1).  `points`: (n, 3) tensor or array; returns int64 labels on
    the device.  The points are binned on a lattice of cell size >= eps here (one sort), the kernels search 27 cells."""
    if not torch.cuda.is_available():
        raise _lib.PixieHipError("dbscan_labels runs on a HIP device only (no CPU fallback)")
    dev = points.device if isinstance(points, torch.Tensor) and points.is_cuda else torch.device("cuda", torch.cuda.current_device())
    pts = torch.as_tensor(points).detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    n = pts.shape[0]
    labels = torch.full((n,), -1, dtype=torch.int64, device=dev)
    if n == 0:
        return labels
    lo = pts.amin(dim=0)
    extent = float((pts.amax(dim=0) - lo).max())
    # cell >= eps with a margin ABOVE the float32 error of the cell index ((p - lo) * inv reaches ~256, i.e. an absolute
    # rounding error of ~1.5e-5 cells: two points exactly eps apart must never end up two cells apart -- ADVICE r3)
    cell = max(float(eps) * (1.0 + 1e-3), extent / 256.0)          # <= 257^3 cells whatever the extent
    inv = float(np.float32(1.0 / cell))                             # (the kernels recompute the cells with this float32 arithmetic)
    cells = torch.floor((pts - lo) * inv).to(torch.int64)
    dims = (cells.amax(dim=0) + 1).tolist()
    nx, ny, nz = (int(d) for d in dims)
    cid = (cells[:, 0] * ny + cells[:, 1]) * nz + cells[:, 2]
    order = torch.argsort(cid, stable=True)
    cell_start = torch.searchsorted(cid[order].contiguous(), torch.arange(nx * ny * nz + 1, device=dev)).to(torch.int32).contiguous()
    pos_sorted = pts[order].contiguous()
    orig = order.to(torch.int32).contiguous()
    core, parent, root = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    p = lambda t: C.c_void_p(t.data_ptr())
    lo64 = (C.c_double * 3)(*[float(v) for v in lo.tolist()])
    check(_lib.load().pixie_dbscan_roots(p(pos_sorted), p(orig), p(cell_start), n, nx, ny, nz, lo64, float(1.0 / inv), float(eps),
                                         int(min_samples), p(core), p(parent), p(root), _lib.current_stream_ptr()), "pixie_dbscan_roots")
    found = root >= 0
    if bool(found.any()):
        roots = torch.unique(root[found])                       # ascending: DBSCAN's cluster numbering
        labels[found] = torch.searchsorted(roots, root[found].contiguous())
    return labels


def handle_stationary_clusters(mpm_solver, positions, material_ids, eps: float = 0.03, min_samples: int = 10,
                               start_time: float = 0.0, end_time: float = 1e6, buffer: float = 0.0,
                               only_handle_largest_cluster: bool = True, debug_output_dir=None, debug: bool = False):
    """material_field.py:364-479: one velocity-zero cuboid (reset=1) around every density cluster of the particles whose
    material is "stationary" -- or only around the largest one.  Returns the list of BC records the reference returns.
    The clustering and the clusters' bounding boxes are computed where the particles live; six numbers per cluster reach
    the host.  (`debug_output_dir` / `debug` are accepted for signature compatibility; the PLY dumps are not reproduced.)"""
    if not torch.cuda.is_available():
        raise _lib.PixieHipError("handle_stationary_clusters runs on a HIP device only (no CPU fallback)")
    dev = positions.device if isinstance(positions, torch.Tensor) and positions.is_cuda else torch.device("cuda", torch.cuda.current_device())
    pos = torch.as_tensor(positions).detach().to(dev).reshape(-1, 3)
    mat = torch.as_tensor(material_ids).detach().to(dev).reshape(-1)
    stationary = pos[mat == STATIONARY_ID]
    if stationary.shape[0] == 0:
        return []
    labels = dbscan_labels(stationary, eps, min_samples)
    n_clusters = int(labels.max()) + 1
    if n_clusters <= 0:
        return []
    counts = torch.bincount(labels[labels >= 0], minlength=n_clusters)
    ids = list(range(n_clusters))
    if only_handle_largest_cluster and n_clusters > 1:
        ids = [int(torch.argmax(counts))]                                        # first maximum, as the reference
    sizes = counts.tolist()
    records = []
    for cid in ids:
        cluster = stationary[labels == cid]
        lo, hi = cluster.amin(dim=0).cpu().numpy(), cluster.amax(dim=0).cpu().numpy()     # (the particles' own dtype, as the reference)
        center = (0.5 * (lo + hi)).tolist()
        half = (0.5 * (hi - lo) + buffer).tolist()
        mpm_solver.set_velocity_on_cuboid(point=center, size=half, velocity=[0.0, 0.0, 0.0], start_time=start_time,
                                          end_time=end_time, reset=1)
        records.append({"type": "stationary_cluster", "cluster_id": int(cid), "point": center, "size": half,
                        "velocity": [0.0, 0.0, 0.0], "start_time": start_time, "end_time": end_time, "reset": 1,
                        "cluster_size": int(sizes[cid])})
    return records


def fix_to_ground(mpm_solver, positions, delta_z: float = 0.02, buffer_xy: float = 0.5, min_z_percentile: float = 1,
                  start_time: float = 0.0, end_time: float = 1e6):
    """material_field.py:485-550: a thin velocity-zero slab (reset=1) under the point cloud, z taken as up.  `positions`
    may be a device tensor: the six extrema are reduced where the data lives."""
    if isinstance(positions, torch.Tensor):
        lo, hi = positions.amin(dim=0), positions.amax(dim=0)
        min_xy, max_xy = lo[:2].double().cpu().numpy(), hi[:2].double().cpu().numpy()
        min_z = float(torch.quantile(positions[:, 2].double(), min_z_percentile / 100.0)) if min_z_percentile > 1 else float(lo[2])
    else:
        pos = np.asarray(positions)
        min_xy, max_xy = pos[:, :2].min(axis=0), pos[:, :2].max(axis=0)
        min_z = float(np.percentile(pos[:, 2], min_z_percentile)) if min_z_percentile > 1 else float(pos[:, 2].min())
    size_xy = max_xy - min_xy
    center = [float(min_xy[0] + max_xy[0]) / 2, float(min_xy[1] + max_xy[1]) / 2, min_z + delta_z / 2]
    half = [float(size_xy[0]) / 2 + buffer_xy, float(size_xy[1]) / 2 + buffer_xy, delta_z / 2]
    mpm_solver.set_velocity_on_cuboid(point=center, size=half, velocity=[0.0, 0.0, 0.0], start_time=start_time,
                                      end_time=end_time, reset=1)
    return [{"type": "ground", "point": center, "size": half, "velocity": [0.0, 0.0, 0.0], "start_time": start_time,
             "end_time": end_time, "reset": 1}]
