"""oracle/filling_oracle.py against tests/golden/filling_ref_golden.npz -- grids, filled cells, volumes and nearest-Gaussian
attributes computed by the REFERENCE'S OWN particle_filling/filling.py, executed unmodified on tests/golden/ti_shim (see
tests/golden/make_filling_ref_golden.py for what that pins and what the reference itself leaves undetermined).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import filling_oracle as fo
from tests import _filling_ref_driver as drv

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "filling_ref_golden.npz")


def load():
    z = np.load(FIXTURE)
    return z, json.loads(str(z["meta"]))


def problem(z, meta, name):
    """The arrays the reference's kernels saw: after the `boundary` crop / shift of filling.py:307-322 where one is given."""
    m = meta[name]
    pos, op, cov = z[name + "/pos"], z[name + "/opacity"], z[name + "/cov"]
    bnd = m["kw"].get("boundary")
    dx = m["grid_dx"]
    if bnd:
        keep = np.ones(len(pos), bool)
        for a in range(3):
            keep &= (pos[:, a] > bnd[2 * a]) & (pos[:, a] < bnd[2 * a + 1])
        pos = (pos - np.array([bnd[0], bnd[2], bnd[4]], np.float32))[keep]          # float32 subtraction, as torch does it
        op, cov = op[keep], cov[keep]
        dx = max(bnd[1] - bnd[0], bnd[3] - bnd[2], bnd[5] - bnd[4]) / m["grid_n"]
    return pos.astype(np.float64), op, cov, np.float64(np.float32(dx))


SCENES = sorted(json.loads(str(np.load(FIXTURE)["meta"])))


def test_fixture_covers_the_options():
    z, meta = load()
    kws = [meta[s]["kw"] for s in SCENES]
    assert {k["max_particles_per_cell"] for k in kws} >= {1, 2, 3}
    assert {k["search_exclude_dir"] for k in kws} >= {5, -1}
    assert {k["ray_cast_dir"] for k in kws} >= {0, 4, 7}                    # +x, +z and the "no parity test" branch (filling.py:155-156)
    assert any("boundary" in k for k in kws)
    assert all(meta[s]["margin"] >= 2e-4 for s in SCENES)
    for s in SCENES:                                                        # every scene fills dense AND internal cells
        assert meta[s]["n_dense"] > 500 and meta[s]["n_total"] - meta[s]["n_dense"] > 100
        assert z[s + "/count_after_densify_grids"].max() > max(k["max_particles_per_cell"] for k in kws)    # crowded cells stay as they are


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_the_reference_grids(name):
    z, meta = load()
    kw = meta[name]["kw"]
    pos, op, cov, dx = problem(z, meta, name)
    n = meta[name]["grid_n"]
    count0, dens = fo.densify(pos, op, cov, n, dx)
    assert np.array_equal(count0, z[name + "/count_after_densify_grids"])
    ref = z[name + "/density"]
    assert np.abs(dens - ref).max() <= 1e-12 * ref.max()                    # measured 2e-16 .. 4e-16
    dense, per = fo.dense_cells(count0, dens, kw["density_thres"], kw["max_particles_per_cell"])
    count1 = np.where(dense, kw["max_particles_per_cell"], count0)
    assert np.array_equal(count1, z[name + "/count_after_fill_dense_grids"])
    assert int(per.sum()) == meta[name]["n_dense"]
    inside = fo.internal_cells(count1, dens, kw["search_thres"], kw["search_exclude_dir"], kw["ray_cast_dir"])
    count2 = np.where(inside, kw["max_particles_per_cell"], count1)
    assert np.array_equal(count2, z[name + "/count_after_internal_filling"])
    assert int(per.sum() + kw["max_particles_per_cell"] * inside.sum()) == meta[name]["n_total"]
    # the float32 run of the reference's own code is this far from its float64 self: the yardstick for float32 implementations
    assert meta[name]["density_drift_f32"] < 1e-6


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_the_reference_volumes_and_attributes(name):
    z, meta = load()
    vol = fo.particle_volume(z[name + "/vol_pos"].astype(np.float64), 16, np.float64(np.float32(1.0 / 16)))
    assert np.abs(vol - z[name + "/volume"]).max() <= 1e-15
    assert np.abs(z[name + "/volume_uniform"] - z[name + "/volume"].mean()).max() < 1e-9 * vol.mean() and np.ptp(z[name + "/volume_uniform"]) == 0
    old, new = z[name + "/attr_old_pos"], z[name + "/attr_new_pos"]
    idx = fo.nearest(old, new)
    k = len(old)
    shs = z[name + "/attr_shs"].reshape(k, -1)
    assert np.array_equal(z[name + "/attr_out_shs"].reshape(-1, shs.shape[1])[k:], shs[idx].astype(np.float64))
    assert np.array_equal(z[name + "/attr_out_opacity"][k:, 0], z[name + "/opacity"][:k][idx].astype(np.float64))
    assert np.array_equal(z[name + "/attr_out_cov"][k:], z[name + "/cov"][:k][idx].astype(np.float64))
    assert np.array_equal(z[name + "/attr_out_shs"].reshape(-1, shs.shape[1])[:k], shs.astype(np.float64))


def test_live_reference_run_small_scene():
    """Re-computes a small scene with the reference's code where /root/reference exists (the build container)."""
    try:
        mod, ti = drv.load_reference()
    except FileNotFoundError:
        pytest.skip("needs /root/reference (build container only)")
    n, dx = 12, 1.0 / 12
    p, o, c = drv._shell(500, 9, (0.5, 0.5, 0.5), (0.38, 0.38, 0.38), (0.4 * dx, 0.55 * dx))
    kw = dict(density_thres=1.0, search_thres=0.6, max_particles_per_cell=2, search_exclude_dir=5, ray_cast_dir=4)
    sc = dict(pos=p, opacity=o, cov=c, grid_n=n, grid_dx=dx, kw=kw)
    r = drv.run_reference(mod, ti, sc, "f64", seed=3)
    count0, dens = fo.densify(p.astype(np.float64), o, c, n, np.float64(np.float32(dx)))
    assert np.array_equal(count0, r["count_after_densify_grids"])
    assert np.abs(dens - r["density"]).max() <= 1e-12 * dens.max()
    assert min(np.abs(dens - kw[k]).min() for k in ("density_thres", "search_thres")) > 1e-6
    dense, _ = fo.dense_cells(count0, dens, 1.0, 2)
    count1 = np.where(dense, 2, count0)
    assert np.array_equal(count1, r["count_after_fill_dense_grids"])
    inside = fo.internal_cells(count1, dens, 0.6, 5, 4)
    assert inside.sum() > 100 and np.array_equal(np.where(inside, 2, count1), r["count_after_internal_filling"])


def test_interpreter_details():
    """The two AST rewrites and the typing rules the fixture relies on (tests/golden/ti_shim/taichi/__init__.py)."""
    import sys
    shim = os.path.join(os.path.dirname(__file__), "golden", "ti_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    import taichi as ti
    assert ti.__file__.startswith(shim)
    g = ti.field(dtype=int, shape=(3, 3))
    assert g._atomic_add((1, 2), 4) == 0 and g._atomic_add(ti.Vector([1, 2]), 1) == 4 and g[1, 2] == 5 and isinstance(g[1, 2], int)
    v = ti.Vector([1, 2, 3]) * np.float32(0.5)
    assert all(isinstance(x, np.float32) for x in v) and list(v) == [0.5, 1.0, 1.5]           # i32 * f32 -> f32
    assert ti.floor(np.float32(-0.5), dtype=int) == -1 and isinstance(ti.ceil(np.float32(1.2), dtype=int), int)
    assert list(ti._ti_range(np.float32(-1.0), np.float32(2.0))) == [-1, 0, 1]              # range bounds are cast to int
    assert ti.math.mod(3, 2) == 1 and ti.math.mod(4, 2) == 0
    w, q = ti.sym_eig(ti.Matrix([[2.0, 0.0, 0.0], [0.0, 3.0, 0.0], [0.0, 0.0, 1.0]]))
    assert sorted(float(x) for x in w) == [1.0, 2.0, 3.0]
