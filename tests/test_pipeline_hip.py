"""BASELINE configs[2] as one scene: the device-resident route (pixie_amd/pipeline.py: U-Net -> un-scaling + K-NN field transfer ->
solver -> fused step loop, every hand-over a device tensor) against the STAGED route the reference's three programs take -- the
prediction through save_predictions' .npy files (inference_combined.py:173-217), the per-particle properties through host arrays
(material_field.py:303-363), one p2g2p() call per substep and an export per frame (gs_simulation.py:633-634).  Each stage has its
own parity tests against the oracle; what this file checks is that nothing is lost or reordered in between: bit for bit."""
import os
import types

import numpy as np
import pytest
import torch

from pixie_amd.synthetic import pipeline_scene
from pixie_amd.unet_plan import synthetic_state_dict

pytestmark = pytest.mark.gpu


def _networks(dev, D, C):
    from pixie_amd.unet import RegressionUNet, SegmentationUNet
    kw = dict(feature_channels=C, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4), attention_resolutions=(), grid_size=D)
    seg, cont = SegmentationUNet(num_classes=8, **kw), RegressionUNet(out_channels=3, **kw)
    seg.load_numpy_state(synthetic_state_dict(seg.cfg, 0)); cont.load_numpy_state(synthetic_state_dict(cont.cfg, 1000))
    return seg.to(dev).eval(), cont.to(dev).eval()


@pytest.mark.parametrize("D,n,substeps", [(32, 20_000, 60)])
def test_device_resident_route_equals_the_staged_route(hip_device, tmp_path, D, n, substeps):
    from pixie_amd import field_mapping as fm
    from pixie_amd.material_field import field_to_particles
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    from pixie_amd.pipeline import neural_scene_rollout
    from pixie_amd.unet import predict_material_field
    sc = pipeline_scene(D, 64, n, seed=0, n_grid=32)
    seg, cont = _networks(hip_device, D, 64)
    feat, mask = torch.from_numpy(sc["feat"]).to(hip_device), torch.from_numpy(sc["mask"]).to(hip_device)
    x0, vol = torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"])
    to_field = lambda x: (x - 1.0) * sc["field_scale"]
    configure = lambda s: s.add_bounding_box()

    # (A) device-resident
    timings = {}
    sa, pred_a, conf_a = neural_scene_rollout(seg, cont, feat, mask, x0, vol, n_grid=sc["n_grid"], grid_lim=sc["grid_lim"], dt=sc["dt"],
                                              n_substeps=substeps, params=sc["params"], min_bounds=sc["min_bounds"], max_bounds=sc["max_bounds"],
                                              to_field_frame=to_field, configure=configure, ranges=sc["ranges"], timings=timings)
    assert set(timings) == {"unet_ms", "solver_setup_ms", "field_to_particles_ms", "rollout_ms", "total_ms"} and timings["total_ms"] > 0

    # (B) staged: program 1 writes the prediction ...
    with torch.no_grad():
        _, seg_pred, _, cont_pred = predict_material_field(seg, cont, feat)
    cfg = types.SimpleNamespace(training=types.SimpleNamespace(num_material_classes=8, **sc["ranges"]))
    info = {"sample_id": [torch.tensor(0)], "data_path": ["d"], "feature_path": ["f"], "mask_path": ["m"]}
    fm.save_predictions(cfg, str(tmp_path), 0, "obj", info, seg_pred[0], cont_pred[0], torch.zeros((4, D, D, D)), None, mask, D)
    pred_file = np.load(os.path.join(tmp_path, "obj", "sample_0_pred.npy"))
    mask_file = np.load(os.path.join(tmp_path, "obj", "sample_0_mask.npy"))
    assert np.array_equal(pred_file, pred_a.cpu().numpy())
    # ... program 3 reads it, transfers the field to its particles through host arrays and steps the solver one p2g2p() at a time
    props = field_to_particles(torch.from_numpy(pred_file).to(hip_device), torch.from_numpy(mask_file).to(hip_device), sc["min_bounds"], sc["max_bounds"],
                               to_field(x0).to(hip_device), ranges=sc["ranges"])
    assert int(props["n_too_far"]) == 0
    host = {k: props[k].cpu().numpy() for k in ("E", "nu", "density", "material_id")}
    assert len(np.unique(host["material_id"])) >= 4            # the predicted field really is mixed-material
    sb = MPM_Simulator_WARP(10)
    sb.load_initial_data_from_torch(x0, vol, None, n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
    sb.set_parameters_dict(sc["params"])
    configure(sb)
    sb.set_per_particle(E=host["E"], nu=host["nu"], density=host["density"], material=host["material_id"])
    sb.finalize_mu_lam()
    for i in range(substeps):
        sb.p2g2p(i, sc["dt"])
    for f in ("material", "E", "mu", "lam", "mass"):
        assert torch.equal(sa.get_field(f), sb.get_field(f)), f
    assert torch.equal(sa.export_particle_x_to_torch(), sb.export_particle_x_to_torch())
    for f in ("v", "C", "F", "F_trial", "yield_stress"):
        assert torch.equal(sa.get_field(f), sb.get_field(f)), f
    assert torch.equal(conf_a, props["conf"])
    x = sa.get_field("x")
    assert bool(torch.isfinite(x).all()) and sa.out_of_bounds == 0
    assert float((x.cpu() - x0).norm()) > 0                    # it moved


def test_pipelined_batch_equals_scene_by_scene(hip_device):
    """neural_scene_batch overlaps the rollout of scene i (side stream, own host thread) with the networks of scene i + 1: every
    scene must end in the state the plain scene-by-scene route gives it, bit for bit."""
    from pixie_amd.pipeline import neural_scene_batch, neural_scene_rollout
    D, n, substeps = 32, 15_000, 80
    seg, cont = _networks(hip_device, D, 64)
    scs = [pipeline_scene(D, 64, n, seed=s, n_grid=32) for s in (0, 1, 2)]
    kw = dict(n_grid=32, grid_lim=scs[0]["grid_lim"], dt=scs[0]["dt"], n_substeps=substeps, params=scs[0]["params"], min_bounds=scs[0]["min_bounds"],
              max_bounds=scs[0]["max_bounds"], to_field_frame=lambda x: (x - 1.0) * scs[0]["field_scale"], configure=lambda s: s.add_bounding_box(),
              ranges=scs[0]["ranges"])
    dev_scene = lambda sc: (torch.from_numpy(sc["feat"]).to(hip_device), torch.from_numpy(sc["mask"]).to(hip_device),
                            torch.from_numpy(sc["x"]).to(hip_device), torch.from_numpy(sc["vol"]).to(hip_device))
    batch = neural_scene_batch(seg, cont, (dev_scene(sc) for sc in scs), **kw)
    xs = [b.export_particle_x_to_torch().clone() for b in batch]       # on the current stream: ordered after every rollout
    for sc, b, x in zip(scs, batch, xs):
        a, _, _ = neural_scene_rollout(seg, cont, *dev_scene(sc), **kw)
        assert torch.equal(x, a.export_particle_x_to_torch())
        for f in ("v", "C", "F_trial", "material", "E"):
            assert torch.equal(a.get_field(f), b.get_field(f)), f
        assert float((x.cpu() - torch.from_numpy(sc["x"])).norm()) > 0
