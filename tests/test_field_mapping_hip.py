"""GPU test of the file contract between the two programs (SURVEY 8b "Field mapping" / "U-Net program"):
pixie_amd.field_mapping.{save_predictions, unscale_prediction, map_pred_to_ply, transform_nerf_to_world,
load_material_points} driven through the reference's own flow

    seg_pred, cont_pred --save_predictions--> sample_0_{pred,gt,mask,info}.npy --map_pred_to_ply--> PLY (+ world PLY)

on the 16^3 scene of tests/golden/make_mapping_golden.py, whose golden arrays were produced by the REFERENCE's own
save_predictions / map_pred_to_ply source (cut out with ast, executed unmodified).  Integers and the one-hot tensor must
be identical; float columns agree to float32 roundoff (powf on the device vs numpy's float32 power)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_mapping_golden import D, scene  # noqa: E402  (scene() only; nothing of the reference is imported)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "field_mapping.npz")


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_save_predictions_then_map_pred_to_ply(hip_device, tmp_path):
    from pixie_amd import field_mapping as fm
    from pixie_amd.ply_io import read_ply
    g = np.load(GOLDEN)
    sc = scene()
    G = int(g["pred_shape"][1])
    pad = lambda a: np.pad(a, [(0, 0)] * (a.ndim - 3) + [(0, G - D)] * 3)
    cont, seg, mask, gt = (torch.from_numpy(pad(sc[k])).to(hip_device) for k in ("cont", "seg", "mask", "gt"))
    cfg = types.SimpleNamespace(training=types.SimpleNamespace(num_material_classes=8, **fm.NORMALIZATION_RANGES))
    info = {"sample_id": [torch.tensor(0)], "data_path": ["d/p"], "feature_path": ["f/p"], "mask_path": ["m/p"]}
    out_dir = str(tmp_path)
    fm.save_predictions(cfg, out_dir, 0, "obj", info, seg, cont, gt, None, mask, G)
    odir = os.path.join(out_dir, "obj")
    assert sorted(os.listdir(odir)) == ["sample_0_gt.npy", "sample_0_info.npy", "sample_0_mask.npy", "sample_0_pred.npy"]
    pred = np.load(os.path.join(odir, "sample_0_pred.npy"))
    assert pred.dtype == np.float32 and tuple(pred.shape) == tuple(g["pred_shape"])
    assert np.array_equal(pred[:, :D, :D, :D], g["pred_crop"])                       # cont copied, one-hot exact
    assert float(pred[3, D:, D:, D:].min()) == g["pred_pad_onehot_class0"][0] and float(pred[4:, D:, D:, D:].max()) == g["pred_pad_onehot_class0"][1]
    assert np.load(os.path.join(odir, "sample_0_mask.npy")).sum() == g["saved_mask_sum"]
    assert np.array_equal(np.load(os.path.join(odir, "sample_0_gt.npy"))[:, :D, :D, :D], g["saved_gt_crop"])
    saved = np.load(os.path.join(odir, "sample_0_info.npy"), allow_pickle=True).item()
    assert sorted(saved.keys()) == list(g["info_keys"]) and saved["sample_id"] == str(g["info_sample_id"])

    # unscale_prediction: numpy in -> numpy out; device tensor in -> device tensor out
    un = fm.unscale_prediction(pred, cfg)
    assert isinstance(un, np.ndarray) and un.dtype == np.float32 and un.shape == pred.shape
    assert rel_l2(un[:, :D, :D, :D], g["unscaled_crop"]) < 2e-6 and np.array_equal(un[3:], pred[3:])
    un_d = fm.unscale_prediction(torch.from_numpy(pred).to(hip_device), cfg)
    assert un_d.is_cuda and np.array_equal(un_d.cpu().numpy(), un)

    # map_pred_to_ply (+ world frame)
    span = (sc["max_bounds"] - sc["min_bounds"]) * (G - 1) / (D - 1)
    np.savez(os.path.join(out_dir, "grid.npz"), min_bounds=sc["min_bounds"], max_bounds=sc["min_bounds"] + span, grid_shape=np.array([G, G, G]))
    json.dump(sc["dataparser"], open(os.path.join(out_dir, "dataparser_transforms.json"), "w"))
    ply, wply = os.path.join(out_dir, "out.ply"), os.path.join(out_dir, "world.ply")
    fm.map_pred_to_ply(os.path.join(odir, "sample_0_pred.npy"), os.path.join(odir, "sample_0_mask.npy"), os.path.join(out_dir, "grid.npz"), ply,
                       "obj", world_output_path=wply, dataparser_path=os.path.join(out_dir, "dataparser_transforms.json"), cfg=cfg)
    v, _ = read_ply(ply)
    assert [f"{n}:{v.dtype.fields[n][0].str}" for n in v.dtype.names] == list(g["ply_dtype"])
    assert len(v) == len(g["ply_x"]) > 500
    for name in ("x", "y", "z", "red", "green", "blue", "alpha", "part_label", "material_id"):
        assert np.array_equal(v[name], g[f"ply_{name}"]), name                       # same points, same order, same ids
    for name in ("density", "E", "nu", "conf"):
        assert rel_l2(v[name], g[f"ply_{name}"]) < 2e-6, name
    w, _ = read_ply(wply)
    assert rel_l2(np.stack([w["x"], w["y"], w["z"]], 1), g["world_xyz"]) < 1e-6
    assert np.array_equal(w["material_id"], v["material_id"])

    # the reader side of gs_simulation.py's load_point_cloud
    pts = fm.load_material_points(ply)
    assert pts["pos"].is_cuda and pts["pos"].shape == (len(v), 3) and np.array_equal(pts["material_id"], v["material_id"])
    assert np.array_equal(pts["part_labels"], v["part_label"]) and pts["conf"].dtype == np.float32

    # shape errors are the reference's ValueErrors
    np.save(os.path.join(out_dir, "bad_mask.npy"), np.zeros((G, G, G - 1), np.float32))
    with pytest.raises(ValueError):
        fm.map_pred_to_ply(os.path.join(odir, "sample_0_pred.npy"), os.path.join(out_dir, "bad_mask.npy"), os.path.join(out_dir, "grid.npz"), ply, "obj", cfg=cfg)


def test_voxel_points_edge_cases(hip_device):
    from pixie_amd import field_mapping as fm
    rng = np.random.default_rng(0)
    pred = rng.normal(size=(11, 5, 6, 7)).astype(np.float32)
    empty = fm.voxel_points(pred, np.zeros((5, 6, 7), np.float32), [0, 0, 0], [1, 1, 1])
    assert empty["xyz"].shape == (0, 3) and empty["material_id"].numel() == 0
    full = fm.voxel_points(pred, np.ones((5, 6, 7), np.float32), [0, 0, 0], [1, 1, 1])
    assert full["xyz"].shape == (210, 3)
    assert np.array_equal(full["material_id"].cpu().numpy(), pred[3:].argmax(0).reshape(-1))
    assert np.allclose(full["conf"].cpu().numpy(), pred[3:].max(0).reshape(-1))
    # a single class channel is the class index itself (get_mat_id, map_pred_to_coords.py:122-126), conf = 1
    one = np.concatenate([pred[:3], rng.integers(0, 8, (1, 5, 6, 7)).astype(np.float32)], 0)
    r = fm.voxel_points(one, np.ones((5, 6, 7), np.float32), [0, 0, 0], [1, 1, 1])
    assert np.array_equal(r["material_id"].cpu().numpy(), one[3].reshape(-1).astype(np.int32)) and float(r["conf"].min()) == 1.0
    # 300 x 300 voxels span several scan chunks of the compaction
    big_mask = (rng.random((3, 300, 300)) < 0.3).astype(np.float32)
    big = fm.voxel_points(rng.normal(size=(11, 3, 300, 300)).astype(np.float32), big_mask, [0, 0, 0], [1, 1, 1])
    assert big["xyz"].shape[0] == int(big_mask.sum())
    lin = [np.linspace(0, 1, n).astype(np.float32) for n in (3, 300, 300)]
    idx = np.argwhere(big_mask > 0)
    assert np.array_equal(big["xyz"].cpu().numpy(), np.stack([lin[0][idx[:, 0]], lin[1][idx[:, 1]], lin[2][idx[:, 2]]], 1))
