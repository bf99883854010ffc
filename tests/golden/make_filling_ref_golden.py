#!/usr/bin/env python
"""Writes tests/golden/filling_ref_golden.npz: the particle pre-pass computed by the REFERENCE'S OWN CODE.

third_party/PhysGaussian/particle_filling/filling.py is imported UNMODIFIED on top of tests/golden/ti_shim -- a NumPy
interpreter of the Taichi subset that file uses (Taichi is not installed here) -- and its `fill_particles`,
`get_particle_volume` and `init_filled_particles` run as written, every @ti.kernel / @ti.func statement by statement
(tests/_filling_ref_driver.py holds the scenes and the calls).  Stored per scene: the inputs; the reference's particle-count
grid after each of its three kernels and its density grid (float64 arithmetic on the float32 problem data = the reference
values; a second run in float32 arithmetic = how far the reference's own code drifts in single precision); the particles it
returned; the volumes and the nearest-Gaussian attributes of gs_simulation.py:466-482.

What the fixture can and cannot pin.  Deterministic in the reference and pinned here: both grids, which cells are filled and
with how many particles, volumes, nearest attributes.  NOT deterministic in the reference itself and therefore not pinned: the
order of the new particles (one atomic counter, filling.py:106, :225) and their ti.random() offsets inside the cell.
`ti.sym_eig` is the interpreter's one stand-in (LAPACK); the kernel uses only Q diag(1/sig) Q^T and max(sig), which no
eigenvector convention changes.  `smooth=True` calls PyMCubes, which is absent: not in this fixture.
Every threshold is checked to be >= 2e-4 (relative) away from every cell's density, and the float32 run must give the same
integer results, so no comparison against this fixture hinges on a rounding.

Run in the build container (needs /root/reference):   python tests/golden/make_filling_ref_golden.py      (~8 min)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import _filling_ref_driver as drv  # noqa: E402

INT_KEYS = ("count_after_densify_grids", "count_after_fill_dense_grids", "count_after_internal_filling")


def main():
    mod, ti = drv.load_reference()
    assert mod.__file__.startswith("/root/reference/"), mod.__file__
    out, meta = {}, {}
    for name, sc in drv.scenes().items():
        t0 = time.time()
        r64 = drv.run_reference(mod, ti, sc, "f64", seed=1)
        r32 = drv.run_reference(mod, ti, sc, "f32", seed=1)
        kw = sc["kw"]
        d = r64["density"]
        margin = min(float(np.abs(d - kw[k]).min() / kw[k]) for k in ("density_thres", "search_thres"))
        assert margin >= 2e-4, (name, margin)
        for k in INT_KEYS:
            assert np.array_equal(r64[k], r32[k]), (name, k)
        assert r64["launches"] == ["densify_grids", "fill_dense_grids", "internal_filling"]
        # the particles the reference returned sit in the cells its grid says it filled
        n0 = len(sc["pos"])
        bnd = kw.get("boundary")
        lo = np.array([bnd[0], bnd[2], bnd[4]]) if bnd else np.zeros(3)
        dx = max(bnd[1] - bnd[0], bnd[3] - bnd[2], bnd[5] - bnd[4]) / sc["grid_n"] if bnd else sc["grid_dx"]
        cell = np.floor((r64["out"][n0:] - lo) / np.float64(np.float32(dx))).astype(int)
        hist = np.zeros((sc["grid_n"],) * 3, int)
        np.add.at(hist, tuple(cell.T), 1)
        assert np.array_equal(hist, r64["count_after_internal_filling"] - r64["count_after_densify_grids"]), name
        assert np.array_equal(r64["out"][:n0], sc["pos"].astype(np.float64))
        pre = name + "/"
        for k in ("pos", "opacity", "cov"):
            out[pre + k] = sc[k]
        for k in INT_KEYS:
            out[pre + k] = r64[k]
        out[pre + "density"] = d
        out[pre + "density_f32"] = r32["density"].astype(np.float32)
        out[pre + "new_particles"] = r64["out"][n0:].astype(np.float32)
        for k in ("vol_pos", "volume", "volume_uniform", "attr_old_pos", "attr_new_pos", "attr_shs", "attr_out_shs", "attr_out_opacity", "attr_out_cov"):
            out[pre + k] = r64[k]
        out[pre + "volume_f32"] = r32["volume"].astype(np.float32)
        for k in ("attr_out_shs", "attr_out_opacity", "attr_out_cov"):     # the float32 run picks the same nearest Gaussians
            assert np.array_equal(r64[k][len(r64["attr_old_pos"]):], r32[k][len(r32["attr_old_pos"]):].astype(np.float64)), (name, k)
        drift = float(np.linalg.norm(r32["density"] - d) / np.linalg.norm(d))
        meta[name] = dict(grid_n=sc["grid_n"], grid_dx=sc["grid_dx"], kw=kw, margin=margin, density_drift_f32=drift,
                          n_dense=int((r64[INT_KEYS[1]] - r64[INT_KEYS[0]]).sum()), n_total=int(len(r64["out"]) - n0))
        print(f"{name}: {time.time() - t0:.0f} s, margin {margin:.2e}, f32 drift of the density {drift:.2e}, "
              f"{meta[name]['n_dense']} dense + {meta[name]['n_total'] - meta[name]['n_dense']} internal particles", flush=True)
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "filling_ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
