#!/usr/bin/env python
"""Writes tests/golden/field_transfer.npz: the field -> particle transfer computed by the REFERENCE's own code.

pixie/voxel/map_pred_to_coords.py and third_party/PhysGaussian/material_field.py cannot be imported here (hydra, warp,
plyfile are not installed), so the source of `unscale_prediction`, `get_mat_id` and `MaterialProperties` is cut out of
those files with `ast` and executed unmodified in a namespace that provides numpy, Counter, a no-op logging and a
get_material_name with the reference's table.  The point-cloud construction of map_pred_to_ply (:192-252) and the K-NN
loop of perform_knn_smoothing (:256-292) are driven exactly as those functions drive them (they are interleaved with
file I/O and tqdm and cannot be cut out as units).  Run in the build container (needs /root/reference); the .npz is committed.
"""
import ast
import os
import sys
import types
from collections import Counter

import numpy as np
from sklearn.neighbors import NearestNeighbors

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def cut(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    out = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out.append(ast.get_source_segment(src, node))
    assert len(out) == len(names), (path, names)
    return "\n\n".join(out)


def main():
    ns = {"np": np, "Counter": Counter, "logging": types.SimpleNamespace(info=lambda *a, **k: None),
          "get_material_name": lambda name: {"stationary": 6}[name],
          "DEFAULT_VALUES": {"density": 1000.0, "E": 5000.0, "nu": 0.3, "part_label": 0, "material_id": "stationary"},
          "DictConfig": object}
    exec(cut(f"{REF}/pixie/voxel/map_pred_to_coords.py", ["unscale_prediction", "get_mat_id"]), ns)
    exec(cut(f"{REF}/third_party/PhysGaussian/material_field.py", ["MaterialProperties"]), ns)
    import json
    ranges = json.load(open(f"{REF}/normalization_stats/normalization_ranges.yaml"))
    cfg = types.SimpleNamespace(training=types.SimpleNamespace(**ranges))

    rng = np.random.default_rng(7)
    D = 24
    pred = np.zeros((11, D, D, D), np.float32)
    pred[:3] = rng.normal(0, 0.7, size=(3, D, D, D)).astype(np.float32)         # some values beyond [-1, 1]: clipped
    logits = rng.normal(size=(8, D, D, D))
    zz, yy, xx = np.meshgrid(*(np.arange(D),) * 3, indexing="ij")
    logits[2] += 1.5 * (zz < D // 2)                                             # spatially coherent classes
    logits[5] += 1.5 * (zz >= D // 2)
    amax = logits.argmax(0)
    pred[3:] = (np.arange(8)[:, None, None, None] == amax[None]).astype(np.float32)   # one-hot, as save_predictions writes
    g = (np.arange(D) - (D - 1) / 2) / (D / 2)
    rr = np.sqrt(g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2)
    mask = ((rr < 0.8) & (rng.random((D, D, D)) < 0.85)).astype(np.float32)      # ball with holes
    min_bounds = np.array([-1.0, -0.9, -1.1]); max_bounds = np.array([1.0, 1.1, 0.9])

    # map_pred_to_ply :187-252, with the PLY column dtypes
    un = ns["unscale_prediction"](pred, cfg)
    cont, seg = un[:3, :], un[3:, :]
    material_id = ns["get_mat_id"](seg)
    x = np.linspace(min_bounds[0], max_bounds[0], D); y = np.linspace(min_bounds[1], max_bounds[1], D); z = np.linspace(min_bounds[2], max_bounds[2], D)
    gx, gy, gz = np.meshgrid(x, y, z, indexing="ij")
    coords = np.stack([gx, gy, gz], axis=-1)
    valid = mask > 0
    pos = coords[valid].astype("f4")
    cloud = dict(pos=pos, density=cont[0][valid].astype("f4"), E=cont[1][valid].astype("f4"), nu=cont[2][valid].astype("f4"),
                 material_id=material_id[valid].astype("i4"), part_labels=material_id[valid].astype("i4"),
                 conf=np.max(seg, axis=0)[valid].astype("f4"))

    n = 3000
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ppos = (d * (0.95 * rng.random(n) ** (1 / 3))[:, None] * 0.9 + np.array([0.0, 0.1, -0.1])).astype(np.float32)  # some outside the ball
    out = {}
    for weighted in (False, True):
        # perform_knn_smoothing :256-292
        props = ns["MaterialProperties"](cloud["part_labels"], cloud["density"], cloud["E"], cloud["nu"], cloud["material_id"], cloud["conf"])
        nn_model = NearestNeighbors(n_neighbors=10, algorithm="auto").fit(cloud["pos"])
        distances_all_k, k_indices = nn_model.kneighbors(ppos)
        too_far_mask = distances_all_k[:, 0] > 0.1
        mapped = props.get_defaults(n)
        for i in np.where(~too_far_mask)[0]:
            for prop_name, value in props.assign_from_neighbors(i, k_indices[i], distances_all_k[i], weighted).items():
                mapped[prop_name][i] = value
        tag = "w_" if weighted else "u_"
        for key, val in mapped.items():
            out[tag + key] = val
        out[tag + "nearest_dist"] = distances_all_k[:, 0]
        out[tag + "too_far"] = too_far_mask
    np.savez_compressed(os.path.join(HERE, "field_transfer.npz"), pred=pred, mask=mask, min_bounds=min_bounds, max_bounds=max_bounds,
                        particle_pos=ppos, unscaled=un, **{"cloud_" + k: v for k, v in cloud.items()}, **out)
    print("wrote field_transfer.npz:", {k: v.shape for k, v in out.items() if k.startswith("u_")}, "too far:", int(out["u_too_far"].sum()))


if __name__ == "__main__":
    main()
