"""CPU restatement of the field -> particle transfer between the two halves of the hot path.
TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows, function by function (paths relative to /root/reference):
  unscale_prediction     pixie/voxel/map_pred_to_coords.py:41-75   (clip to [-1,1], 10**log ranges)
  voxel_point_cloud      pixie/voxel/map_pred_to_coords.py:192-252 (np.linspace coordinates, mask > 0, argmax id, max conf;
                          the PLY stores x,y,z,density,E,nu,conf as 'f4' and the ids as 'i4')
  knn_assign             third_party/PhysGaussian/material_field.py:228-300 (sklearn NearestNeighbors, K = 10, too-far test
                          on the nearest distance only) with MaterialProperties.get_defaults (:38-50) and
                          .assign_from_neighbors (:52-78: np.mean / Counter.most_common, or inverse-distance weights)
Pinned: tests/golden/field_transfer.npz is produced by tests/golden/make_field_golden.py, which executes the
reference's own unscale_prediction and MaterialProperties source (extracted from the files above with `ast`, because
their modules import hydra / warp / plyfile, which are not installed) -- tests/test_field_oracle.py checks this
restatement against it bit for bit.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

NORMALIZATION_RANGES = {  # normalization_stats/normalization_ranges.yaml
    "density_min": 1.7031893730163574, "density_max": 3.871432304382324,
    "E_min": 3.0183002948760986, "E_max": 10.881680488586426,
    "nu_min": 0.21027633547782898, "nu_max": 0.4492689371109009,
}
STATIONARY_ID = 6          # get_material_name("stationary"), mpm_solver_warp.py:10-39
DEFAULT_PART_LABEL = 0     # DEFAULT_VALUES['part_label'], material_field.py:17-23


def unscale_prediction(pred: np.ndarray, ranges=NORMALIZATION_RANGES) -> np.ndarray:
    """map_pred_to_coords.py:41-75"""
    cont = np.clip(pred[:3], -1.0, 1.0)
    out = pred.copy().astype(np.float32)
    dens_log = (cont[0] + 1.0) * (ranges["density_max"] - ranges["density_min"]) / 2.0 + ranges["density_min"]
    out[0] = 10 ** dens_log
    e_log = (cont[1] + 1.0) * (ranges["E_max"] - ranges["E_min"]) / 2.0 + ranges["E_min"]
    out[1] = 10 ** e_log
    out[2] = (cont[2] + 1.0) * (ranges["nu_max"] - ranges["nu_min"]) / 2.0 + ranges["nu_min"]
    return out


def voxel_point_cloud(pred_unscaled: np.ndarray, mask: np.ndarray, min_bounds, max_bounds):
    """map_pred_to_coords.py:192-252 -> dict of the PLY vertex columns (float32 / int32)."""
    cont, seg = pred_unscaled[:3], pred_unscaled[3:]
    material_id = np.argmax(seg, axis=0)
    D, H, W = mask.shape
    x = np.linspace(min_bounds[0], max_bounds[0], D)
    y = np.linspace(min_bounds[1], max_bounds[1], H)
    z = np.linspace(min_bounds[2], max_bounds[2], W)
    gx, gy, gz = np.meshgrid(x, y, z, indexing="ij")
    valid = mask > 0
    pos = np.stack([gx[valid], gy[valid], gz[valid]], axis=-1).astype(np.float32)
    conf = np.max(seg, axis=0)[valid].astype(np.float32) if seg.shape[0] > 1 else np.ones(int(valid.sum()), np.float32)
    mid = material_id[valid].astype(np.int32)
    return dict(pos=pos, density=cont[0][valid].astype(np.float32), E=cont[1][valid].astype(np.float32),
                nu=cont[2][valid].astype(np.float32), material_id=mid, part_labels=mid.copy(), conf=conf)


def knn_assign(cloud: dict, particle_pos: np.ndarray, k: int = 10, nn_distance_threshold: float = 0.1, weighted: bool = False):
    """material_field.py:228-300: returns dict(part_labels, density, E, nu, material_id, conf, nearest_dist, too_far)."""
    from sklearn.neighbors import NearestNeighbors
    n = particle_pos.shape[0]
    nn_model = NearestNeighbors(n_neighbors=k, algorithm="auto").fit(cloud["pos"])
    dist, idx = nn_model.kneighbors(particle_pos)
    too_far = dist[:, 0] > nn_distance_threshold
    props = {key: cloud[key] for key in ("part_labels", "density", "E", "nu", "material_id", "conf")}
    out = {}
    for key, values in props.items():   # get_defaults, :38-50
        if key == "material_id":
            default = STATIONARY_ID
        elif key == "part_labels":
            default = DEFAULT_PART_LABEL
        else:
            default = np.mean(values)
        out[key] = np.full(n, default, dtype=values.dtype)
    for i in np.where(~too_far)[0]:     # assign_from_neighbors, :52-78
        ni, d = idx[i], dist[i]
        w = 1.0 / (d + 1e-8)
        w = w / np.sum(w)
        for key, values in props.items():
            nv = values[ni]
            if key in ("material_id", "part_labels"):
                if weighted:
                    uniq, inv = np.unique(nv, return_inverse=True)
                    out[key][i] = uniq[np.argmax(np.bincount(inv, weights=w))]
                else:
                    out[key][i] = Counter(nv).most_common(1)[0][0]
            else:
                out[key][i] = np.dot(w, nv) if weighted else np.mean(nv)
    out["nearest_dist"] = dist[:, 0]
    out["too_far"] = too_far
    return out


def field_to_particles(pred: np.ndarray, mask: np.ndarray, min_bounds, max_bounds, particle_pos: np.ndarray, k: int = 10,
                       nn_distance_threshold: float = 0.1, weighted: bool = False, ranges=NORMALIZATION_RANGES):
    """The whole transfer: network output (11,D,H,W) + occupancy mask -> per-particle material properties."""
    cloud = voxel_point_cloud(unscale_prediction(pred, ranges), mask, min_bounds, max_bounds)
    return knn_assign(cloud, particle_pos, k, nn_distance_threshold, weighted)


def dbscan_labels(points: np.ndarray, eps: float, min_samples: int) -> np.ndarray:
    """The labels sklearn.cluster.DBSCAN(eps, min_samples).fit_predict assigns (PhysGaussian/material_field.py:405-406),
    restated without sklearn as a radius graph (tests/test_bc_construction.py pins it to sklearn itself): a point is a core
    point when its closed eps-ball holds >= min_samples points; clusters are the connected components of the core points,
    numbered in the order of their lowest-index core point; a border point joins the earliest-numbered cluster that has a
    core point within eps; everything else is noise (-1)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree

    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = len(pts)
    labels = np.full(n, -1, dtype=np.int64)
    if n == 0:
        return labels
    pairs = cKDTree(pts).query_pairs(float(eps), output_type="ndarray")           # i < j, |pi - pj| <= eps
    degree = np.bincount(pairs.ravel(), minlength=n) + 1                          # the point itself counts
    core = degree >= min_samples
    if not core.any():
        return labels
    both = core[pairs[:, 0]] & core[pairs[:, 1]]
    graph = coo_matrix((np.ones(int(both.sum()), dtype=np.int8), (pairs[both, 0], pairs[both, 1])), shape=(n, n))
    _, comp = connected_components(graph, directed=False)
    core_idx = np.flatnonzero(core)
    first_core = np.full(comp.max() + 1, n, dtype=np.int64)
    np.minimum.at(first_core, comp[core_idx], core_idx)                           # lowest core index of each component
    order = np.argsort(first_core, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    labels[core_idx] = rank[comp[core_idx]]
    # border points: earliest cluster among their core neighbours
    edge = np.concatenate([pairs[core[pairs[:, 1]] & ~core[pairs[:, 0]]],
                           pairs[core[pairs[:, 0]] & ~core[pairs[:, 1]]][:, ::-1]])   # (border, core)
    if len(edge):
        best = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
        np.minimum.at(best, edge[:, 0], labels[edge[:, 1]])
        hit = best != np.iinfo(np.int64).max
        labels[hit] = best[hit]
    return labels
