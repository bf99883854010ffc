#!/usr/bin/env python
"""Writes tests/golden/frame_export.npz: the per-frame export for the rasteriser computed by the
REFERENCE's own code.

gs_simulation.py:591-600 per frame:   pos = export_particle_x_to_torch()[:gs_num];  cov3D = export_particle_cov_to_torch()
    pos_render  = transform_to_original_coordinates(undoshift2center111(pos, z_shift), scale_origin, mean, rotations)
    cov3D_render = apply_inverse_cov_rotations(cov3D / scale_origin**2, rotations)
The functions are cut out of third_party/PhysGaussian/utils/transformation_utils.py and material_field.py:81-86 with `ast`
and executed unmodified except that the device string "cuda" becomes "cpu" (no GPU in the build container; torch CPU float32
and float64).  The covariances come from the reference's `compute_cov_from_F` kernel (mpm_utils.py:529-553) run on the Warp
interpreter: scene `jelly_apic` of tests/golden/mpm_ref_golden.npz (x after 6 substeps, cov_out), so the whole chain
F_trial -> covariance -> scene frame is the reference's.
"""
import ast
import os

import numpy as np
import torch

REF = "/root/reference/third_party/PhysGaussian"
HERE = os.path.dirname(os.path.abspath(__file__))


class CudaToCpu(ast.NodeTransformer):
    def visit_Constant(self, node):
        return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node

    def visit_Call(self, node):          # tensor.cuda() -> tensor
        self.generic_visit(node)
        if isinstance(node.func, ast.Attribute) and node.func.attr == "cuda" and not node.args:
            return node.func.value
        return node


def cut(path, names, ns):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (path, names)
    mod = ast.fix_missing_locations(CudaToCpu().visit(ast.Module(body=body, type_ignores=[])))
    exec(compile(mod, path, "exec"), ns)


def main():
    ns = {"torch": torch, "np": np}
    cut(f"{REF}/utils/transformation_utils.py",
        ["undotransform2origin", "generate_rotation_matrix", "generate_rotation_matrices", "apply_cov_rotation", "get_mat_from_upper",
         "get_uppder_from_mat", "undoshift2center111", "apply_inverse_rotation", "apply_inverse_rotations", "apply_inverse_cov_rotations"], ns)
    cut(f"{REF}/material_field.py", ["transform_to_original_coordinates"], ns)

    z = np.load(os.path.join(HERE, "mpm_ref_golden.npz"))
    x, cov = z["jelly_apic/k6/x"], z["jelly_apic/cov_out"]
    gs_num, z_shift = 150, 0.05
    degrees, axes = [30.0, -75.0, 12.0], [0, 2, 1]
    scale_origin, mean = 0.37, np.array([0.3, -1.2, 2.0])
    out = dict(x=x, cov=cov, gs_num=gs_num, z_shift=z_shift, degrees=np.array(degrees), axes=np.array(axes), scale_origin=scale_origin, mean=mean)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        torch.set_default_dtype(dt)                  # torch.zeros(..) inside get_mat_from_upper follows the default dtype
        rots = ns["generate_rotation_matrices"](torch.tensor(degrees, dtype=dt), axes)
        rots = [r.to(dt) for r in rots]
        pos = torch.tensor(x, dtype=dt)[:gs_num]
        c6 = torch.tensor(cov, dtype=dt).view(-1, 6)[:gs_num]
        s, m = torch.tensor(scale_origin, dtype=dt), torch.tensor(mean, dtype=dt)
        pos_render = ns["transform_to_original_coordinates"](ns["undoshift2center111"](pos, z_shift), s, m, rots)
        cov_render = ns["apply_inverse_cov_rotations"](c6 / (s ** 2), rots)
        out[f"pos_{tag}"], out[f"cov_{tag}"] = pos_render.numpy(), cov_render.numpy()
        out[f"rot_{tag}"] = torch.stack(rots).numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "frame_export.npz"), **out)
    print("wrote frame_export.npz; f32-vs-f64 of the reference chain: pos",
          np.abs(out["pos_f32"] - out["pos_f64"]).max() / np.abs(out["pos_f64"]).max(), "cov",
          np.abs(out["cov_f32"] - out["cov_f64"]).max() / np.abs(out["cov_f64"]).max())


if __name__ == "__main__":
    main()
