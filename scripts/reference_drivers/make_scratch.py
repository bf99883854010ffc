#!/usr/bin/env python
"""Build container only: puts the reference's DRIVER code for the hot path into scripts/reference_drivers/_scratch/
(git-ignored: reference sources never enter the history; the directory travels to the GPU box with the gpurun snapshot).

VERBATIM copies (byte for byte; their imports are satisfied by the alias / stand-in modules in _stubs/):
    third_party/PhysGaussian/material_field.py            perform_knn_smoothing, fix_to_ground, handle_stationary_clusters,
                                                           _apply_material_properties_to_solver, apply_material_field_to_simulation
    third_party/PhysGaussian/utils/decode_param.py         decode_param_json, set_boundary_conditions
    third_party/PhysGaussian/utils/transformation_utils.py
    third_party/PhysGaussian/config/objaverse/custom_tree_config.json   (+ the sport-balls config)
CUT with `ast` (their files cannot be imported whole: hydra / omegaconf / wandb / the Gaussian rasteriser are absent):
    ref_unet_driver.py   create_models, process_batch, save_predictions (WG/trainer/inference_combined.py:81-217);
                         masked_mean, compute_accuracy, load_checkpoint (pixie/training_utils.py); InferenceMetrics
                         (pixie/metrics.py:105-153); MaterialVoxelDataset (WG/data_utils/my_data.py:19-224)
    ref_sharded_inference.py   run_inference_on_gpu, load_test_dataset (WG/trainer/inference_combined.py:48-79,229-288);
                         load_normalization_ranges (pixie/training_utils.py:21-48); generate_metrics_report, save_metrics_file
                         (pixie/metrics.py); its header's ddp_setup takes the backend from the runner (the reference hard-codes "nccl")
    ref_map_pred.py      unscale_prediction, get_mat_id, map_pred_to_ply (pixie/voxel/map_pred_to_coords.py:41-75, 122-283)
    ref_gs_main.py       load_point_cloud (gs_simulation.py:108-202) and the statements of gs_simulation.py's __main__ block
                         that set up and drive the solver (:483-502, :531, :558-561, the frame loop :573-634 with the
                         camera / rasteriser / image statements removed), wrapped into `simulate(...)`; and the particle
                         pre-pass of the same block (:413-482: rotation, sim-area crop, transform2origin, shift2center111,
                         fill_particles, get_particle_volume, init_filled_particles), wrapped into `prepass(...)`.
Each cut function's source text is the reference's, unmodified; the only edits are the ones INTEGRATION.md section 1 documents
(the imports of SegmentationUNet / RegressionUNet / MPM_Simulator_WARP) -- made in the header this script writes, not in
the function bodies (`from particle_filling.filling import *` becomes `from pixie_amd.particle_filling import *` the same way)
-- and `frame_hook(...)`, a call appended to the frame loop so the runner can observe every frame.
"""
import ast
import os
import shutil

REF = "/root/reference"
PG = f"{REF}/third_party/PhysGaussian"
WG = f"{REF}/third_party/Wavelet-Generation"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_scratch")


def cut(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            first = min([node.lineno] + [d.lineno for d in node.decorator_list])
            found[node.name] = (f"# ---- {os.path.relpath(path, REF)}:{first}-{node.end_lineno} (verbatim)\n"
                                + "\n".join(src.splitlines()[first - 1:node.end_lineno]))
    missing = [n for n in names if n not in found]
    assert not missing, (path, missing)
    return "\n\n\n".join(found[n] for n in names)


def main_body_cut(path):
    """The solver set-up and the frame loop of gs_simulation.py's `if __name__ == "__main__":` block."""
    src = open(path).read()
    lines = src.splitlines()
    tree = ast.parse(src)
    main_if = next(n for n in tree.body if isinstance(n, ast.If) and "__name__" in ast.unparse(n.test))

    def text(node):
        return "\n".join(lines[node.lineno - 1:node.end_lineno])

    keep, loop = [], None
    for st in main_if.body:
        seg = ast.get_source_segment(src, st) or ""
        if isinstance(st, ast.For) and "frame_num" in ast.unparse(st.iter):
            loop = st
        elif any(key in seg for key in ("mpm_solver = MPM_Simulator_WARP", "mpm_solver.load_initial_data_from_torch", "mpm_solver.set_parameters_dict(material_params)",
                                        "set_boundary_conditions(mpm_solver", "pc_params = load_point_cloud", "apply_material_field_to_simulation(",
                                        "substep_dt = time_params", "frame_dt = time_params", "frame_num = time_params", "step_per_frame = int(")) \
                and not isinstance(st, (ast.If, ast.For)):
            keep.append(st)
        elif isinstance(st, ast.Expr) and seg.strip() == "mpm_solver.finalize_mu_lam()":
            keep.append(st)
    assert loop is not None and len(keep) == 11, [ast.unparse(k)[:50] for k in keep]
    body = [f"    # gs_simulation.py:{st.lineno}-{st.end_lineno} (verbatim)\n" + text(st) for st in keep]
    # frame loop: keep the solver-facing statements only
    out = [f"    # gs_simulation.py:{loop.lineno} (verbatim loop header; tqdm dropped)", "    for frame in range(frame_num):"]
    for st in loop.body:
        seg = ast.get_source_segment(src, st) or ""
        if isinstance(st, ast.Assign) and seg.startswith("pos = mpm_solver.export_particle_x_to_torch()"):
            out.append(f"        # :{st.lineno}\n" + text(st))
        elif isinstance(st, ast.If) and ast.unparse(st.test) == "args.render_img":
            out.append(f"        # :{st.lineno}\n        if args.render_img:")
            for sub in st.body:
                sseg = ast.get_source_segment(src, sub) or ""
                if isinstance(sub, ast.Assign) and sseg.split(" =")[0] in ("cov3D", "pos_render", "cov3D_render"):
                    out.append(f"            # :{sub.lineno}-{sub.end_lineno}\n" + text(sub))
            out.append("            frame_hook(frame, mpm_solver, pos, pos_render, cov3D_render)      # (added: observation point for the runner)")
        elif isinstance(st, ast.For) and "step_per_frame" in ast.unparse(st.iter):
            out.append(f"        # :{st.lineno}-{st.end_lineno}\n" + text(st))
    return "\n".join(body), "\n".join(out)


def prepass_cut(path):
    """gs_simulation.py:413-482 of the main block: from `rotation_matrices = generate_rotation_matrices(` to the
    `if filling_params and filling_params.get("visualize", False): ... else: ...` statement, every statement verbatim."""
    src = open(path).read()
    lines = src.splitlines()
    tree = ast.parse(src)
    main_if = next(n for n in tree.body if isinstance(n, ast.If) and "__name__" in ast.unparse(n.test))
    body = main_if.body
    first = next(i for i, st in enumerate(body) if (ast.get_source_segment(src, st) or "").startswith("rotation_matrices = generate_rotation_matrices("))
    last = next(i for i, st in enumerate(body) if isinstance(st, ast.If) and ast.unparse(st.test).startswith("filling_params and filling_params.get('visualize'"))
    assert 0 < first < last
    out = []
    for st in body[first:last + 1]:
        out.append(f"    # gs_simulation.py:{st.lineno}-{st.end_lineno} (verbatim)\n" + "\n".join(lines[st.lineno - 1:st.end_lineno]))
    return "\n".join(out)


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, "utils"))
    for src, dst in ((f"{PG}/material_field.py", "material_field.py"), (f"{PG}/utils/decode_param.py", "utils/decode_param.py"),
                     (f"{PG}/utils/transformation_utils.py", "utils/transformation_utils.py"),
                     (f"{PG}/config/objaverse/custom_tree_config.json", "custom_tree_config.json"),
                     (f"{PG}/config/objaverse/custom_sand_config.json", "custom_sand_config.json"),
                     (f"{PG}/config/objaverse/custom_sport_balls_config.json", "custom_sport_balls_config.json")):
        shutil.copyfile(src, os.path.join(OUT, dst))
    open(os.path.join(OUT, "utils", "__init__.py"), "w").close()
    open(os.path.join(OUT, "utils", "camera_view_utils.py"), "w").write("# out of scope (camera orbit helpers)\n")

    unet = "\n\n\n".join([
        '"""Cut from the reference by scripts/reference_drivers/make_scratch.py -- see there.  Header = the documented import swap."""',
        "import logging\nimport os\nfrom collections import defaultdict\nfrom pathlib import Path\n\nimport numpy as np\nimport torch\n\n"
        "from pixie_amd.unet import RegressionUNet, SegmentationUNet      # INTEGRATION.md section 1 (was: trainer.training_discrete / training_continuous_mse)\n\n"
        "get_obj_class_for_id = load_json = None        # pixie/utils.py helpers used by the dataset; the runner provides them",
        cut(f"{REF}/pixie/training_utils.py", ["masked_mean", "compute_accuracy", "load_checkpoint"]),
        cut(f"{REF}/pixie/metrics.py", ["InferenceMetrics"]),
        cut(f"{WG}/data_utils/my_data.py", ["MaterialVoxelDataset"]),
        cut(f"{WG}/trainer/inference_combined.py", ["create_models", "process_batch", "save_predictions"]),
    ])
    open(os.path.join(OUT, "ref_unet_driver.py"), "w").write(unet + "\n")

    # the multi-process entry (WG/trainer/inference_combined.py:229-288) with what it calls.  The header's ddp_setup is the ONE edit
    # beyond the documented import swap: the reference's (pixie/training_utils.py:50-55) hard-codes backend "nccl" and
    # torch.cuda.set_device(rank), which cannot form two ranks on a one-GPU box (or on CPU); the runner passes backend / port in.
    sharded = "\n\n\n".join([
        '"""Cut from the reference by scripts/reference_drivers/make_scratch.py -- see there.  The sharded inference loop."""',
        "import logging\nimport math\nimport os\nimport sys\nfrom collections import defaultdict\nfrom pathlib import Path\n\nimport numpy as np\nimport torch\n"
        "import torch.distributed as dist\nimport yaml\nfrom torch.utils.data import DataLoader, Subset\nfrom torch.utils.data.distributed import DistributedSampler\n\n"
        "from ref_unet_driver import InferenceMetrics, MaterialVoxelDataset, create_models, load_checkpoint, process_batch\n\n"
        "DictConfig = object\nDDP_BACKEND, DDP_PORT = 'gloo', '12355'\n\n\n"
        "def set_logger():\n    logging.basicConfig(level=logging.INFO)\n\n\n"
        "def tqdm(it, **kw):\n    return it\n\n\n"
        "def save_json(obj, path):\n    import json\n    json.dump(obj, open(path, 'w'), indent=2, default=float)\n\n\n"
        "def ddp_setup(rank, world_size):      # pixie/training_utils.py:50-55 with the backend / port of the runner (see make_scratch.py)\n"
        "    os.environ['MASTER_ADDR'] = '127.0.0.1'\n    os.environ['MASTER_PORT'] = DDP_PORT\n"
        "    dist.init_process_group(DDP_BACKEND, rank=rank, world_size=world_size)\n    torch.cuda.set_device(rank)",
        cut(f"{REF}/pixie/training_utils.py", ["load_normalization_ranges"]),
        cut(f"{REF}/pixie/metrics.py", ["save_metrics_file", "generate_metrics_report"]),
        cut(f"{WG}/trainer/inference_combined.py", ["load_test_dataset", "run_inference_on_gpu"]),
    ])
    open(os.path.join(OUT, "ref_sharded_inference.py"), "w").write(sharded + "\n")

    mp = "\n\n\n".join([
        '"""Cut from pixie/voxel/map_pred_to_coords.py by scripts/reference_drivers/make_scratch.py -- see there."""',
        "import logging\nimport os\nfrom pathlib import Path\n\nimport numpy as np\nfrom plyfile import PlyData, PlyElement      # the stand-in over pixie_amd.ply_io\n\nDictConfig = object",
        cut(f"{REF}/pixie/voxel/map_pred_to_coords.py", ["unscale_prediction", "get_mat_id", "map_pred_to_ply"]),
    ])
    open(os.path.join(OUT, "ref_map_pred.py"), "w").write(mp + "\n")

    setup, loop = main_body_cut(f"{PG}/gs_simulation.py")
    gs = "\n\n\n".join([
        '"""Cut from third_party/PhysGaussian/gs_simulation.py by scripts/reference_drivers/make_scratch.py -- see there."""',
        "import numpy as np\nimport torch\nfrom plyfile import PlyData, PlyElement\n\n"
        "from mpm_solver_warp.mpm_solver_warp import MPM_Simulator_WARP      # resolves to pixie_amd.mpm_solver (INTEGRATION.md section 1)\n"
        "from pixie_amd.particle_filling import *      # INTEGRATION.md section 1 (was: from particle_filling.filling import *)\n"
        "from material_field import apply_material_field_to_simulation, transform_to_original_coordinates\n"
        "from utils.decode_param import *\nfrom utils.transformation_utils import *",
        cut(f"{PG}/gs_simulation.py", ["load_point_cloud"]),
        "def simulate(args, material_params, bc_params, time_params, preprocessing_params, mpm_init_pos, mpm_init_vol, init_cov, gs_num,\n"
        "             scale_origin, original_mean_pos, rotation_matrices, frame_hook, device=\"cuda:0\"):\n"
        "    \"\"\"The solver-facing statements of gs_simulation.py's main block, in order.  Inputs = the variables that block has\n"
        "    computed by line 482 (from the Gaussian-splat model, which is out of scope); `frame_hook` is the only addition.\"\"\"\n"
        "    grid_lim = material_params[\"grid_lim\"]\n"
        "    unselected_pos = None\n"
        "    # gs_simulation.py:479-481 (verbatim)\n"
        "    mpm_init_cov = torch.zeros((mpm_init_pos.shape[0], 6), device=device)\n"
        "    mpm_init_cov[:gs_num] = init_cov\n"
        + setup + "\n" + loop + "\n    return mpm_solver",
        "def prepass(preprocessing_params, material_params, init_pos, init_cov, init_opacity, init_shs):\n"
        "    \"\"\"The particle pre-pass of gs_simulation.py's main block, in order.  Inputs = the variables that block holds at line 412\n"
        "    (the opacity-filtered Gaussians of the splat model, which is out of scope); returns its locals.\"\"\"\n"
        + prepass_cut(f"{PG}/gs_simulation.py") + "\n    return dict(locals())",
    ])
    open(os.path.join(OUT, "ref_gs_main.py"), "w").write(gs + "\n")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
