#!/usr/bin/env python
"""GPU box: runs the REFERENCE'S DRIVER CODE for the hot path against pixie_amd.

What runs is the reference's own text (scripts/reference_drivers/make_scratch.py put it into _scratch/ in the build
container): the only edits are the import swaps INTEGRATION.md section 1 documents, realised as module aliases in _stubs/.

  A. U-Net program   MaterialVoxelDataset -> DataLoader -> create_models -> load_checkpoint(strict=False) ->
                     process_batch -> save_predictions   (WG/trainer/inference_combined.py:81-217 and what it calls)
                     on synthetic files in the reference's on-disk formats; checks the four sample_0_*.npy files and the
                     logits against oracle/unet_oracle.py (pinned to the reference modules).
  B. field mapping   the reference's map_pred_to_ply (pixie/voxel/map_pred_to_coords.py:128-283, on the plyfile
                     stand-in) and pixie_amd.field_mapping.map_pred_to_ply on the same files: the PLYs must agree.
  C. MPM program     gs_simulation.py's solver set-up and frame loop (:483-502, :531, :558-634), material_field.py
                     (K-NN smoothing, ground slab, DBSCAN cluster BCs, the N-box material upload) and decode_param.py
                     unmodified, once on pixie_amd.mpm_solver and once on a thin adapter around the C oracle with the
                     same class surface: per-frame positions / covariances handed to the rasteriser must agree.
                     Also decides the `live_exports` default: does the driver ever read an exported tensor it held
                     across p2g2p calls without re-exporting?
  E. sharded U-Net program  run_inference_on_gpu (inference_combined.py:229-288) verbatim as `--world` processes (mp.spawn, gloo):
                     ddp_setup, DistributedSampler, DataLoader, gather_all_metrics, generate_metrics_report; against a 1-process run.
  D. particle pre-pass  gs_simulation.py:413-482 verbatim (rotation, transform2origin, shift2center111, fill_particles,
                     get_particle_volume, init_filled_particles) with the reference's custom_sand_config.json -- the config
                     that sets `"smooth": true` -- on pixie_amd.particle_filling; the result is checked against the chain
                     of oracle/filling_oracle.py (pinned to the reference's filling.py) on the same Gaussians.
Writes gpurun_out/reference_drivers.log (copied to profiles/ by the session script).  Exit code 0 = every check passed.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SCRATCH = os.path.join(HERE, "_scratch")
sys.path[:0] = [os.path.join(HERE, "_stubs"), SCRATCH, REPO]

LOG = []


def say(*a):
    line = " ".join(str(x) for x in a)
    LOG.append(line)
    print(line, flush=True)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


class Cfg(dict):
    """Attribute access over nested dicts -- the slice of omegaconf.DictConfig the cut code uses."""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) else v

    def get(self, k, d=None):
        return getattr(self, k) if k in self else d


def make_cfg(root, D, C):
    ranges = dict(density_min=1.70319, density_max=3.87143, E_min=3.01830, E_max=10.88168, nu_min=0.210276, nu_max=0.449269)  # normalization_ranges.yaml
    return Cfg(training=dict(to_normalize=True, feature_type="clip", target_obj_classes=None, default_grid_size=D, feature_channels=C,
                             in_material_channels=4, num_material_classes=8, background_id=7, enforce_mask_consistency=True, sample_id=0,
                             cond_dim=32, training=dict(unet_model_channels=64, unet_num_res_blocks=3, unet_channel_mult=[1, 1, 2, 4],
                                                        attention_resolutions=[]), **ranges),
               paths=dict(render_outputs_dir=os.path.join(root, "render_outputs"), normalization_stats_dir=os.path.join(root, "normalization_stats")))


def synth_objects(cfg, D, C, obj_ids):
    """render_outputs/<id>/{clip_features_features.npy (D,D,D,C) float16, clip_features_mask.npy, clip_features.npz,
    sample_0/material_grid.npy (D,D,D,4)}: the reference's on-disk input format."""
    for k, oid in enumerate(obj_ids):
        rng = np.random.default_rng(50 + k)
        d = os.path.join(cfg.paths.render_outputs_dir, oid)
        os.makedirs(os.path.join(d, "sample_0"), exist_ok=True)
        g = (np.arange(D) - (D - 1) / 2) / (D / 2)
        rr = np.sqrt(g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2)
        mask = rr < 0.7
        feat = (rng.standard_normal((D, D, D, C), dtype=np.float32) * mask[..., None]).astype(np.float16)
        np.save(os.path.join(d, "clip_features_features.npy"), feat)
        np.save(os.path.join(d, "clip_features_mask.npy"), mask)
        mat = np.zeros((D, D, D, 4), np.float32)
        mat[..., 0] = 10 ** rng.uniform(2.3, 3.3, (D, D, D)); mat[..., 1] = 10 ** rng.uniform(5.0, 6.3, (D, D, D)); mat[..., 2] = rng.uniform(0.22, 0.43, (D, D, D))
        mat[..., 3] = np.where(mask, rng.integers(0, 7, (D, D, D)), 7)
        np.save(os.path.join(d, "sample_0", "material_grid.npy"), mat)
        np.savez(os.path.join(d, "clip_features.npz"), min_bounds=np.array([-0.5, -0.45, -0.55]), max_bounds=np.array([0.5, 0.55, 0.45]),
                 voxel_size=1.0 / D, feature_dim=C, grid_shape=np.array([D, D, D]))


# ============================================================================================ A. the U-Net program
def part_a(root, D, C):
    import ref_unet_driver as R
    from oracle import unet_oracle
    from pixie_amd.unet_plan import synthetic_state_dict
    R.get_obj_class_for_id = lambda obj_id, cfg: "tree"
    R.load_json = lambda p: json.load(open(p))
    cfg = make_cfg(root, D, C)
    obj_ids = ["synthetic_a", "synthetic_b"]
    synth_objects(cfg, D, C, obj_ids)
    ds = R.MaterialVoxelDataset(cfg)                                                   # my_data.py:19-224
    assert sorted(ds.obj_ids) == obj_ids, ds.obj_ids
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)   # inference_combined.py:247-256 (one rank)
    seg, cont = R.create_models(cfg, 0)                                                # :81-105, the swapped classes, `.to(rank)`
    sds = {}
    for tag, net, seed in (("seg", seg, 0), ("cont", cont, 1000)):
        sd = synthetic_state_dict(net.cfg, seed)
        sds[tag] = sd
        path = os.path.join(root, f"{tag}_epoch_3.pth")
        torch.save({"epoch": 3, "model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer_state_dict": {},
                    "scheduler_state_dict": None}, path)                                 # training_discrete.py:257-264
        ep = R.load_checkpoint(path, net, rank=0)                                      # training_utils.py:191-225 (strict=False)
        assert ep == 4, ep
    seg.eval(); cont.eval()
    metrics = R.InferenceMetrics()
    out_dir = os.path.join(root, "inference_results")
    t0 = time.perf_counter()
    for batch in loader:
        R.process_batch(seg, cont, batch, cfg, 0, metrics, out_dir)                     # :108-170 -> save_predictions :173-217
    torch.cuda.synchronize()
    say(f"A. process_batch over {len(ds)} objects (batch of 2, {D}^3 x {C}): {time.perf_counter() - t0:.2f} s; seg_acc {metrics.seg_accuracies}, cont_mse {metrics.cont_mse_values}")
    assert len(metrics.seg_accuracies) == 1 and sorted(metrics.local_obj_ids) == obj_ids
    for k, oid in enumerate(ds.obj_ids):
        d = os.path.join(out_dir, oid)
        pred = np.load(os.path.join(d, "sample_0_pred.npy")); gt = np.load(os.path.join(d, "sample_0_gt.npy")); m = np.load(os.path.join(d, "sample_0_mask.npy"))
        info = np.load(os.path.join(d, "sample_0_info.npy"), allow_pickle=True).item()
        assert pred.shape == (11, D, D, D) and pred.dtype == np.float32 and gt.shape == (4, D, D, D) and m.shape == (D, D, D) and info["obj_id"] == oid
        assert np.array_equal(pred[3:].sum(0), np.ones((D, D, D), np.float32))         # one-hot
        feat = ds[k][0].unsqueeze(0).numpy()
        lo = unet_oracle.unet_forward(sds["seg"], seg.cfg, feat).numpy()[0]
        co = unet_oracle.unet_forward(sds["cont"], cont.cfg, feat).numpy()[0]
        e_c = rel(pred[:3], co)
        agree = float((pred[3:].argmax(0) == lo.argmax(0)).mean())
        say(f"   {oid}: sample_0_pred.npy continuous channels vs the pinned oracle {e_c:.2e}; argmax agreement {agree:.6f}")
        assert e_c < 1e-4 and agree > 0.999
    return cfg, out_dir, ds.obj_ids


# ============================================================================================ B. field mapping
def part_b(cfg, out_dir, obj_id, stable_field):
    import ref_unet_driver  # noqa: F401  (same scratch dir)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_map_pred", os.path.join(SCRATCH, "ref_map_pred.py"))
    M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
    from pixie_amd import field_mapping
    from pixie_amd.ply_io import read_ply
    d = os.path.join(out_dir, obj_id)
    npz = os.path.join(cfg.paths.render_outputs_dir, obj_id, "clip_features.npz")
    pred_path = os.path.join(d, "sample_0_pred.npy")
    if stable_field:      # random-init networks predict E over 1e3..1e11 Pa, unstable at dt 1e-4; the MPM part gets a prescribed field
        D = cfg.training.default_grid_size
        rng = np.random.default_rng(3)
        pred = np.zeros((11, D, D, D), np.float32)
        pred[0] = rng.uniform(-0.4, 0.1, (D, D, D)); pred[1] = rng.uniform(-0.48, -0.2, (D, D, D)); pred[2] = rng.uniform(-0.9, 0.6, (D, D, D))
        zz = np.arange(D)[None, None, :] * np.ones((D, D, 1))
        cls = np.where(zz < 0.36 * D, 6, 0)                                           # a "stationary" cap at the bottom, jelly above
        pred[3:] = (np.arange(8)[:, None, None, None] == cls[None]).astype(np.float32)
        pred_path = os.path.join(d, "sample_0_pred_prescribed.npy")
        np.save(pred_path, pred)
    mask_path = os.path.join(d, "sample_0_mask.npy")
    ours, theirs = os.path.join(d, "mapped_preds.ply"), os.path.join(d, "mapped_preds_reference_code.ply")
    M.map_pred_to_ply(pred_path, mask_path, npz, theirs, obj_id, cfg=cfg)                # the reference's function, verbatim
    field_mapping.map_pred_to_ply(pred_path, mask_path, npz, ours, obj_id, cfg=cfg)
    a, _ = read_ply(ours); b, _ = read_ply(theirs)
    assert a.dtype == b.dtype and len(a) == len(b), (a.dtype, b.dtype, len(a), len(b))
    worst = 0.0
    for name in a.dtype.names:
        if a[name].dtype.kind == "f":
            worst = max(worst, rel(a[name], b[name]))
        else:
            assert np.array_equal(a[name], b[name]), name
    say(f"B. map_pred_to_ply: {len(a)} vertices, fields {a.dtype.names}; pixie_amd vs the reference's function: integer fields equal, float fields {worst:.1e}")
    assert worst < 1e-6
    return ours


# ============================================================================================ C. the MPM program
class _ArrayView:
    def __init__(self, get):
        self._get = get

    def numpy(self):
        return self._get()


class OracleSolver:
    """oracle/mpm_oracle.c behind the class surface the reference's drivers use (test infrastructure)."""
    precision = "f64_omp"

    def __init__(self, n_particles, n_grid=100, grid_lim=1.0, device="cuda:0"):
        self.o = None

    def load_initial_data_from_torch(self, tensor_x, tensor_volume, tensor_cov=None, n_grid=100, grid_lim=1.0, device="cuda:0"):
        from oracle.mpm_oracle import OracleMPM
        self.n_particles = tensor_x.shape[0]
        self.o = OracleMPM(self.n_particles, n_grid, grid_lim, self.precision)
        self.o.load_initial_data(tensor_x.cpu().numpy(), tensor_volume.cpu().numpy(), None if tensor_cov is None else tensor_cov.cpu().numpy())
        self.mpm_state = types.SimpleNamespace(particle_x=_ArrayView(lambda: np.array(self.o.field("x"))))
        print("Particles initialized from torch data.")

    def set_parameters_dict(self, kwargs={}, device="cuda:0"):
        self.o.set_parameters_dict(kwargs)

    def finalize_mu_lam(self, device="cuda:0"):
        self.o.finalize_mu_lam()

    def p2g2p(self, step, dt, device="cuda:0"):
        self.o.p2g2p(step, dt)

    def export_particle_x_to_torch(self):
        return torch.from_numpy(np.array(self.o.field("x"), dtype=np.float32)).cuda()     # float32, as a Warp array is

    def export_particle_cov_to_torch(self, device="cuda:0"):
        return torch.from_numpy(np.array(self.o.export_cov(), dtype=np.float32)).cuda()

    def __getattr__(self, name):
        if name in ("set_velocity_on_cuboid", "add_surface_collider", "add_bounding_box", "add_impulse_on_particles",
                    "enforce_particle_velocity_translation", "enforce_particle_velocity_rotation", "release_particles_sequentially"):
            return getattr(self.o, name)
        raise AttributeError(name)


def part_c(root, ply_path, n_particles, frames):
    import material_field as MF      # verbatim file
    import ref_gs_main as G
    from pixie_amd import mpm_solver as product
    from pixie_amd.particle_filling import get_particle_volume
    from utils.decode_param import decode_param_json
    from utils.transformation_utils import apply_cov_rotations, apply_rotations, generate_rotation_matrices, shift2center111, transform2origin
    material_params, bc_params, time_params, preprocessing_params, camera_params = decode_param_json(os.path.join(SCRATCH, "custom_tree_config.json"))
    time_params["frame_num"] = frames
    # gs_simulation.py:401-470 on synthetic "Gaussians": a ball of kernels inside the voxel grid's bounds (the reference
    # loads them from a trained 3DGS model, which is out of scope)
    rng = np.random.default_rng(11)
    d = rng.normal(size=(n_particles, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    init_pos = torch.tensor((d * (0.33 * rng.random(n_particles) ** (1 / 3))[:, None] + np.array([0.0, 0.05, -0.05])).astype(np.float32), device="cuda")
    A = rng.normal(size=(n_particles, 3, 3)) * 4e-3
    S = A @ A.transpose(0, 2, 1) + 2e-5 * np.eye(3)
    init_cov = torch.tensor(np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32), device="cuda")
    rotation_matrices = generate_rotation_matrices(torch.tensor(preprocessing_params["rotation_degree"]), preprocessing_params["rotation_axis"])   # :413-416
    rotated_pos = apply_rotations(init_pos, rotation_matrices)                                       # :417
    transformed_pos, scale_origin, original_mean_pos = transform2origin(rotated_pos)                 # :440
    transformed_pos = shift2center111(transformed_pos, preprocessing_params["z_shift_value"])        # :441
    init_cov = apply_cov_rotations(init_cov, rotation_matrices) * (scale_origin ** 2)                # :442
    gs_num = transformed_pos.shape[0]
    mpm_init_pos = transformed_pos.to("cuda:0")
    mpm_init_vol = get_particle_volume(mpm_init_pos, material_params["n_grid"], material_params["grid_lim"] / material_params["n_grid"],
                                       unifrom=material_params["material"] == "sand").to("cuda:0")    # :466-471
    args = types.SimpleNamespace(point_cloud_path=ply_path, render_img=True, debug=False)
    runs = {}
    class OracleSolver32(OracleSolver):
        precision = "f32_omp"
    for tag, cls in (("pixie_amd", product.MPM_Simulator_WARP), ("oracle", OracleSolver), ("oracle_f32", OracleSolver32)):
        G.MPM_Simulator_WARP = cls
        frames_seen, held = [], {}

        def hook(frame, solver, pos, pos_render, cov3D_render):
            x_mpm = solver.get_field("x").cpu().numpy().astype(np.float64) if tag == "pixie_amd" else np.array(solver.o.field("x"), dtype=np.float64)
            frames_seen.append((pos_render.detach().cpu().numpy().copy(), cov3D_render.detach().cpu().numpy().copy(), x_mpm))
            if tag == "pixie_amd":
                if frame == 0:
                    held["x"] = solver.export_particle_x_to_torch()      # an export HELD across the next frame's p2g2p calls
                elif frame == 1:     # `pos` is what the driver itself exported for this frame (gs_simulation.py:591)
                    held["current_when_read"] = bool((held["x"][:pos.shape[0]] == pos).all().item())
        t0 = time.perf_counter()
        solver = G.simulate(args, dict(material_params), list(bc_params), dict(time_params), dict(preprocessing_params), mpm_init_pos, mpm_init_vol,
                            init_cov, gs_num, scale_origin, original_mean_pos, rotation_matrices, hook)
        torch.cuda.synchronize()
        runs[tag] = frames_seen
        steps = int(time_params["frame_dt"] / time_params["substep_dt"]) * frames
        say(f"C. {tag}: gs_simulation set-up + {frames} frames x {steps // frames} substeps on {n_particles} particles: {time.perf_counter() - t0:.1f} s")
        if tag == "pixie_amd":
            mats = np.unique(solver.mpm_state.particle_material.numpy(), return_counts=True)
            say(f"   materials after apply_material_field_to_simulation: {dict(zip(mats[0].tolist(), mats[1].tolist()))}; time {solver.time:.4f}")
            assert 6 in mats[0] and 0 in mats[0], mats
            say(f"   held export: a tensor from export_particle_x_to_torch() held across a frame of p2g2p calls "
                f"{'equals' if held['current_when_read'] else 'DIFFERS from'} the driver's own export of the next frame (one persistent tensor per "
                f"field, refreshed by every export call; the drivers export before every read: gs_simulation.py:591,594, material_field.py:244,322) "
                f"-> the reference's drivers never observe a stale tensor; live_exports stays opt-in")
            assert held["current_when_read"]
    ok = True
    x0 = runs["oracle"][0][2]
    for f in range(len(runs["oracle"])):
        (p1, c1, x1), (p2, c2, x2), (p3, c3, x3) = runs["pixie_amd"][f], runs["oracle"][f], runs["oracle_f32"][f]
        ex, ec = rel(p1, p2), rel(c1, c2)
        line = f"   frame {f}: pos_render vs the float64 oracle run {ex:.2e}; cov3D_render {ec:.2e}"
        ok = ok and ex < 1e-4 and ec < 1e-4
        if f:   # Displacement in the solver's frame.  This scene hardly moves (one impulse of 0.48 N for one substep: 1e-4 of the
            # coordinate in 400 substeps), so x += dt v adds ~1-4 float32 ulps per substep and its rounding is a SYSTEMATIC
            # per-particle loss -- in any float32 solver, the reference's Warp kernels included.  The product is therefore held
            # to the float32 oracle (same arithmetic, same rounding), with the float64 run shown for scale.
            ed64, drift, ed32 = rel(x1 - x0, x2 - x0), rel(x3 - x0, x2 - x0), rel(x1 - x0, x3 - x0)
            line += (f"; displacement |x - x0| = {np.linalg.norm(x2 - x0) / np.sqrt(len(x0)):.2e} rms: vs the float32 oracle {ed32:.2e}, vs the float64 oracle "
                     f"{ed64:.2e} (the float32 oracle's own distance from it: {drift:.2e})")
            ok = ok and ed32 < 1e-2 and ed64 < max(1e-4, 1.5 * drift)
        say(line)
    assert ok


# ============================================================================================ D. the particle pre-pass
def part_d(n_gaussians):
    import ref_gs_main as G
    from oracle import filling_oracle as fo
    from utils.decode_param import decode_param_json
    material_params, bc_params, time_params, preprocessing_params, camera_params = decode_param_json(os.path.join(SCRATCH, "custom_sand_config.json"))
    fp = preprocessing_params["particle_filling"]
    assert fp["smooth"] is True and fp["visualize"] is True and material_params["material"] == "sand"
    # synthetic "Gaussians" in the splat model's own frame: a closed shell (the reference loads them from a trained 3DGS model)
    rng = np.random.default_rng(5)
    d = rng.normal(size=(n_gaussians, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = (d * np.array([0.8, 0.7, 0.75]) + np.array([0.3, -0.2, 0.1])).astype(np.float32)
    sig = rng.uniform(0.020, 0.030, size=(n_gaussians, 3)) * np.array([1.0, 0.85, 1.15])          # x 1/1.6 after transform2origin: 1-1.8 filling cells
    Q = np.linalg.qr(rng.normal(size=(n_gaussians, 3, 3)))[0]
    S = Q @ (sig[:, :, None] ** 2 * np.eye(3)) @ Q.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    opa = rng.uniform(0.5, 1.0, size=(n_gaussians, 1)).astype(np.float32)
    shs = rng.normal(size=(n_gaussians, 16, 3)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    t0 = time.perf_counter()
    L = G.prepass(preprocessing_params, material_params, t(pos), t(cov), t(opa), t(shs))
    torch.cuda.synchronize()
    gs_num0 = n_gaussians
    n_all = L["mpm_init_pos"].shape[0]
    say(f"D. gs_simulation.py:413-482 on pixie_amd.particle_filling with custom_sand_config.json (filling grid {fp['n_grid']}^3 in the box {fp['boundary'][:2]}, "
        f"density_threshold {fp['density_threshold']}, search_threshold {fp['search_threshold']}, exclude {fp['search_exclude_direction']}, smooth=True, visualize=True): "
        f"{gs_num0} Gaussians -> {n_all} particles in {time.perf_counter() - t0:.1f} s")
    assert L["gs_num"] == n_all and L["mpm_init_vol"].shape == (n_all,) and L["mpm_init_cov"].shape == (n_all, 6)
    assert L["shs_render"].shape == (n_all, 16, 3) and L["opacity_render"].shape == (n_all, 1)
    # --- the same chain on the pinned oracle, from the arrays the reference code handed to fill_particles
    tp, tc = L["transformed_pos"].cpu().numpy(), L["init_cov"].cpu().numpy()
    bnd, n = fp["boundary"], fp["n_grid"]
    dx = np.float64(np.float32(max(bnd[1] - bnd[0], bnd[3] - bnd[2], bnd[5] - bnd[4]) / n))
    lo = np.array([bnd[0], bnd[2], bnd[4]], np.float32)
    assert ((tp > lo) & (tp < np.array([bnd[1], bnd[3], bnd[5]], np.float32))).all()
    count0, dens = fo.densify((tp - lo).astype(np.float64), opa, tc, n, dx)
    dense, per = fo.dense_cells(count0, dens, fp["density_threshold"], fp["max_partciels_per_cell"])
    count1 = np.where(dense, fp["max_partciels_per_cell"], count0)
    sm = fo.smooth_constrained(dens.astype(np.float32).astype(np.float64), max_iters=500).astype(np.float32)
    inside = fo.internal_cells(count1, sm, fp["search_threshold"], fp["search_exclude_direction"], fp["ray_cast_direction"])
    want = int(per.sum() + fp["max_partciels_per_cell"] * inside.sum())
    new = L["mpm_init_pos"][gs_num0:].cpu().numpy()
    cell = np.floor((new - lo).astype(np.float64) / dx).astype(int)
    hist = np.zeros((n,) * 3, int)
    np.add.at(hist, tuple(np.clip(cell, 0, n - 1).T), 1)
    expect = np.where(inside, fp["max_partciels_per_cell"], 0) + per
    differing = int((hist != expect).sum())
    # the smoothed field is compared against a threshold after 500 Jacobi sweeps in float64 on the device vs scipy: a handful of
    # cells within ~1e-6 of the threshold may fall either side (and flip the cells on their rays)
    say(f"   new particles: {len(new)} (oracle chain on the same Gaussians: {want}; dense cells {int(dense.sum())}, internal cells {int(inside.sum())}); "
        f"cells whose particle count differs from the oracle chain: {differing}")
    assert abs(len(new) - want) <= max(3, want // 1000) and differing <= max(6, want // 500)
    assert torch.equal(L["mpm_init_pos"][:gs_num0], L["transformed_pos"])
    # volumes: sand -> uniform (gs_simulation.py:470)
    vol = L["mpm_init_vol"].cpu().numpy()
    ref_vol = fo.particle_volume(L["mpm_init_pos"].cpu().numpy().astype(np.float64), material_params["n_grid"],
                                 np.float64(np.float32(material_params["grid_lim"] / material_params["n_grid"])))
    say(f"   get_particle_volume(unifrom=True): {vol[0]:.6e} everywhere (oracle mean {ref_vol.mean():.6e})")
    assert np.ptp(vol) == 0 and abs(vol[0] - ref_vol.mean()) < 1e-5 * ref_vol.mean()
    # attributes of the filled particles: the nearest original Gaussian's (a sample, brute force on the host)
    pick = rng.choice(len(new), size=min(400, len(new)), replace=False)
    idx = fo.nearest(tp, new[pick])
    assert np.array_equal(L["mpm_init_cov"][gs_num0:].cpu().numpy()[pick], tc[idx])
    assert np.array_equal(L["opacity_render"][gs_num0:, 0].cpu().numpy()[pick], opa[idx, 0])
    assert np.array_equal(L["shs_render"][gs_num0:].cpu().numpy()[pick], shs[idx])
    say(f"   init_filled_particles: covariance / opacity / SH rows of {len(pick)} sampled new particles equal their nearest Gaussian's (brute force)")
    return L


# ============================================================================================ E. the sharded U-Net program
class NCfg(dict):
    """Nested attribute-access config that can be assigned to (load_normalization_ranges writes cfg.training.E_min = ...)."""
    def __init__(self, d=()):
        super().__init__()
        for k, v in dict(d).items():
            self[k] = NCfg(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, d=None):
        return self[k] if k in self else d


def _part_e_worker(rank, world, cfg_dict, seg_ckpt, cont_ckpt, out_dir, port):
    """What mp.spawn runs per rank in the reference (main_worker -> run_inference_on_gpu, inference_combined.py:291-353)."""
    import ref_sharded_inference as S
    import ref_unet_driver as R
    R.get_obj_class_for_id = lambda obj_id, cfg: "tree"
    R.load_json = lambda p: json.load(open(p))
    S.DDP_BACKEND, S.DDP_PORT = "gloo", str(port)
    if torch.cuda.device_count() < world:
        # The reference uses the process rank as the device ordinal (`torch.cuda.set_device(rank)`, `.to(rank)`); this box shows fewer
        # devices than ranks, so THE RUNNER maps every integer device ordinal to device 0 -- the ranks share the one GPU.  Everything
        # else (rendezvous, DistributedSampler split, per-rank DataLoader, gather_object, rank 0's report) is the reference's own code.
        _set_device, _t_to, _m_to = torch.cuda.set_device, torch.Tensor.to, torch.nn.Module.to
        fix = lambda a: tuple(0 if (isinstance(x, int) and not isinstance(x, bool)) else x for x in a)
        torch.cuda.set_device = lambda d: _set_device(0)
        torch.Tensor.to = lambda self, *a, **k: _t_to(self, *fix(a), **k)
        torch.nn.Module.to = lambda self, *a, **k: _m_to(self, *fix(a), **k)
    S.run_inference_on_gpu(rank, world, NCfg(cfg_dict), seg_ckpt, cont_ckpt, None, out_dir, print_table=False)   # verbatim :229-288


def part_e(root, D, C, world):
    """run_inference_on_gpu (inference_combined.py:229-288) verbatim -- ddp_setup, load_normalization_ranges, load_test_dataset,
    DistributedSampler(shuffle=False), DataLoader(pin_memory), create_models, load_checkpoint, process_batch per batch,
    InferenceMetrics.gather_all_metrics (dist.gather_object), generate_metrics_report on rank 0 -- as `world` processes started with
    torch.multiprocessing.spawn, on gloo (the reference's "nccl" cannot form two ranks on this box's single GPU).  Its
    rank == device-ordinal convention needs `world` visible devices; with fewer, the runner maps integer device ordinals to
    device 0 inside the workers (the ranks share the GPU) -- the only thing emulated.
    Checks: every object's four files exist exactly once, they equal a single-process run over the same objects bit for bit,
    rank 0's report lists every object."""
    import torch.multiprocessing as mp
    from pixie_amd.unet_plan import UNetConfig, synthetic_state_dict
    cfg = make_cfg(root, D, C)
    cfg["training"]["inference"] = dict(batch_size=1, data_worker=0, use_saved_test_split=False)
    cfg["training"]["training"]["train_size"] = 0.0
    obj_ids = [f"sharded_{k}" for k in range(4)]
    for d in (cfg.paths.render_outputs_dir,):
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    synth_objects(cfg, D, C, obj_ids)
    os.makedirs(cfg.paths.normalization_stats_dir, exist_ok=True)
    import yaml
    yaml.safe_dump({"density_p1": 1.70319, "density_p99": 3.87143, "E_p1": 3.01830, "E_p99": 10.88168, "nu_p1": 0.210276, "nu_p99": 0.449269},
                   open(os.path.join(cfg.paths.normalization_stats_dir, "normalization_ranges.yaml"), "w"))
    ckpts = []
    for tag, oc, seed in (("seg", 8, 0), ("cont", 3, 1000)):
        sd = synthetic_state_dict(UNetConfig(feature_channels=C, grid_size=D, out_channels=oc), seed)
        path = os.path.join(root, f"{tag}_sharded_epoch_3.pth")
        torch.save({"epoch": 3, "model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer_state_dict": {}, "scheduler_state_dict": None}, path)
        ckpts.append(path)
    import socket
    outs = {}
    for w in sorted({1, world}):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        out_dir = os.path.join(root, f"sharded_results_w{w}")
        t0 = time.perf_counter()
        mp.spawn(_part_e_worker, args=(w, json.loads(json.dumps(cfg)), ckpts[0], ckpts[1], out_dir, port), nprocs=w, join=True)
        outs[w] = out_dir
        files = sorted(os.listdir(out_dir))
        shared = " -- the ranks SHARE the one GPU: the runner maps integer device ordinals to 0" if torch.cuda.device_count() < w else ""
        say(f"E. run_inference_on_gpu verbatim, world_size {w} (mp.spawn, gloo, {torch.cuda.device_count()} visible device(s){shared}): {time.perf_counter() - t0:.1f} s; {files}")
        ev = json.load(open(os.path.join(out_dir, "evaluated_obj_ids.json")))
        assert ev == obj_ids, ev
        assert any(f.startswith("metrics") or f.endswith("metrics.json") for f in files) or len(files) >= len(obj_ids) + 2, files
        for oid in obj_ids:
            assert sorted(os.listdir(os.path.join(out_dir, oid))) == ["sample_0_gt.npy", "sample_0_info.npy", "sample_0_mask.npy", "sample_0_pred.npy"], oid
    if world > 1:
        for oid in obj_ids:
            for f in ("sample_0_pred.npy", "sample_0_gt.npy", "sample_0_mask.npy"):
                a, b = np.load(os.path.join(outs[1], oid, f)), np.load(os.path.join(outs[world], oid, f))
                assert np.array_equal(a, b), (oid, f)
        say(f"   world_size {world} == world_size 1: the {len(obj_ids)} objects' pred / gt / mask files are bit-identical; rank 0's report lists all of them")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=64)        # map_pred_to_ply asserts a 64^3 mask (map_pred_to_coords.py:182)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--particles", type=int, default=20000)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--only", default="abcd")
    ap.add_argument("--world", type=int, default=2, help="part e: ranks of the sharded U-Net program (needs that many visible devices)")
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs the GPU box"
    root = tempfile.mkdtemp(prefix="pixie_ref_drivers_")
    if "a" in a.only or "b" in a.only or "c" in a.only:
        cfg, out_dir, obj_ids = part_a(root, a.grid, a.channels)
        ply = part_b(cfg, out_dir, obj_ids[0], stable_field=True)
    if "c" in a.only:
        part_c(root, ply, a.particles, a.frames)
    if "d" in a.only:
        part_d(a.gaussians)
    if "e" in a.only:
        part_e(root, min(a.grid, 32), a.channels, a.world)
    say("ALL CHECKS PASSED")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "reference_drivers.log"), "w").write("\n".join(LOG) + "\n")


if __name__ == "__main__":
    main()
