"""Scenes and the driver for the particle-filling fixture computed by the REFERENCE'S OWN filling.py (test infrastructure).

`load_reference()` imports third_party/PhysGaussian/particle_filling/filling.py unmodified on top of tests/golden/ti_shim
(a NumPy interpreter of the Taichi subset it uses); `run_reference(mod, ti, scene, precision)` calls the reference's
`fill_particles`, `get_particle_volume` and `init_filled_particles` as gs_simulation.py:442-482 does and reads the reference's
intermediate grids out of the interpreter after every kernel.  Needs /root/reference: build container only."""
import contextlib
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/third_party/PhysGaussian/particle_filling/filling.py"


def _shell(n, seed, centre, radii, sigma, open_bottom=None):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    if open_bottom is not None:
        d = d[d[:, 2] > open_bottom]
    pos = np.asarray(centre) + d * np.asarray(radii)
    # rotated anisotropic covariances, all six entries non-zero
    A = rng.normal(size=(len(pos), 3, 3))
    Q = np.linalg.qr(A)[0]
    s = rng.uniform(sigma[0], sigma[1], size=(len(pos), 3)) * np.array([1.0, 0.8, 1.25])
    S = Q @ (s[:, :, None] ** 2 * np.eye(3)) @ Q.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
    op = rng.uniform(0.7, 1.0, len(pos))
    return pos.astype(np.float32), op.astype(np.float32), cov.astype(np.float32)


def _torus(n, seed, R, r, sigma):
    rng = np.random.default_rng(seed)
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    pos = 0.5 + np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], 1)
    s2 = rng.uniform(sigma[0], sigma[1], n) ** 2
    cov = np.zeros((n, 6)); cov[:, 0] = s2; cov[:, 3] = 1.2 * s2; cov[:, 5] = 0.9 * s2; cov[:, 1] = 0.15 * s2; cov[:, 4] = -0.1 * s2
    return pos.astype(np.float32), np.full(n, 0.9, np.float32), cov.astype(np.float32)


def scenes():
    """name -> dict(pos, opacity, cov, grid_n, grid_dx, and the keyword arguments of fill_particles)."""
    n = 24
    dx = 1.0 / n
    out = {}
    p, o, c = _shell(2200, 1, (0.48, 0.53, 0.5), (0.30, 0.26, 0.28), (0.6 * dx, 0.95 * dx))
    out["shell"] = dict(pos=p, opacity=o, cov=c, grid_n=n, grid_dx=dx, kw=dict(density_thres=2.0, search_thres=1.0, max_particles_per_cell=1,
                                                                             search_exclude_dir=5, ray_cast_dir=4))
    p, o, c = _shell(2200, 2, (0.5, 0.5, 0.52), (0.29, 0.29, 0.3), (0.6 * dx, 0.95 * dx), open_bottom=-0.6)
    out["open_bottom_ppc3"] = dict(pos=p, opacity=o, cov=c, grid_n=n, grid_dx=dx, kw=dict(density_thres=1.5, search_thres=0.8, max_particles_per_cell=3,
                                                                                        search_exclude_dir=5, ray_cast_dir=4))
    out["open_bottom_no_exclusion_parity_off"] = dict(pos=p, opacity=o, cov=c, grid_n=n, grid_dx=dx,
                                                      kw=dict(density_thres=2.0, search_thres=1.0, max_particles_per_cell=1, search_exclude_dir=-1, ray_cast_dir=7))
    p, o, c = _torus(3000, 3, 0.27, 0.12, (0.55 * dx, 0.8 * dx))
    out["torus_ray_x"] = dict(pos=p, opacity=o, cov=c, grid_n=n, grid_dx=dx, kw=dict(density_thres=2.0, search_thres=1.0, max_particles_per_cell=2,
                                                                                   search_exclude_dir=-1, ray_cast_dir=0))
    # the `boundary` route: the box is shifted and scaled; grid_dx passed in is ignored (filling.py:319-321)
    p, o, c = _shell(2200, 4, (0.5, 0.5, 0.5), (0.3, 0.27, 0.29), (0.6 * dx, 0.95 * dx))
    shift, scale = np.array([0.2, -0.1, 0.3], np.float32), np.float32(1.5)
    out["boundary_box"] = dict(pos=(p * scale + shift).astype(np.float32), opacity=o, cov=(c * scale * scale).astype(np.float32), grid_n=n, grid_dx=123.0,
                               kw=dict(density_thres=2.01, search_thres=1.0, max_particles_per_cell=1, search_exclude_dir=5, ray_cast_dir=4,
                                       boundary=[0.2, 1.7, -0.1, 1.4, 0.3, 1.8]))
    return out


def load_reference():
    """(reference module, interpreter module).  Raises FileNotFoundError without /root/reference."""
    if not os.path.exists(REF_FILE):
        raise FileNotFoundError(REF_FILE)
    shim = os.path.join(HERE, "golden", "ti_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    import taichi as ti
    assert ti.__file__.startswith(shim), ti.__file__
    spec = importlib.util.spec_from_file_location("pixie_reference_filling", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, ti


@contextlib.contextmanager
def _no_gpu():
    """filling.py moves tensors with `.cuda()` (filling.py:311, :322, :376, ...); the build container has no GPU."""
    import torch
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = saved


def run_reference(mod, ti, scene, precision="f64", seed=0):
    import torch
    ti.set_precision(precision)
    ti.seed_random(seed)
    del ti.FIELDS[:], ti.KERNEL_LOG[:]
    snaps = {}

    def hook(name, bound):
        if name in ("densify_grids", "fill_dense_grids", "internal_filling"):
            snaps[name] = (bound["grid"].to_numpy().copy(), bound["grid_density"].to_numpy().copy())
    ti.KERNEL_HOOK = hook
    try:
        with _no_gpu():
            pos, op, cov = (torch.from_numpy(scene[k]) for k in ("pos", "opacity", "cov"))
            out = mod.fill_particles(pos, op[:, None], cov, scene["grid_n"], 200_000, scene["grid_dx"], **scene["kw"])
            launches = list(ti.KERNEL_LOG)
            res = dict(out=out.numpy().astype(np.float64), launches=launches)
            for k, (g, d) in snaps.items():
                res["count_after_" + k] = g.astype(np.int32)
            res["density"] = snaps["densify_grids"][1].astype(np.float64)
            assert np.array_equal(snaps["internal_filling"][1], snaps["densify_grids"][1])
            # gs_simulation.py:466-474: the volume of every particle (grid of the MPM domain), then the filled particles' attributes
            new = out[len(pos):]
            bnd = scene["kw"].get("boundary")
            lo = np.array([bnd[0], bnd[2], bnd[4]], np.float32) if bnd else np.zeros(3, np.float32)
            width = np.float32(max(bnd[1] - bnd[0], bnd[3] - bnd[2], bnd[5] - bnd[4])) if bnd else np.float32(1.0)
            vol_pos = ((out - torch.from_numpy(lo)) / float(width)).to(torch.float32)           # into the unit box the volume grid covers
            res["vol_pos"] = vol_pos.numpy()
            res["volume"] = mod.get_particle_volume(vol_pos, 16, 1.0 / 16).numpy().astype(np.float64)
            res["volume_uniform"] = mod.get_particle_volume(vol_pos, 16, 1.0 / 16, unifrom=True).numpy().astype(np.float64)
            k_old, k_new = min(len(pos), 500), min(len(new), 150)
            rng = np.random.default_rng(7)
            shs = torch.from_numpy(rng.normal(size=(k_old, 4, 3)).astype(np.float32))
            res["attr_old_pos"], res["attr_new_pos"], res["attr_shs"] = pos[:k_old].numpy(), new[:k_new].numpy().astype(np.float32), shs.numpy()
            s2, o2, c2 = mod.init_filled_particles(pos[:k_old], shs, cov[:k_old], op[:k_old, None], torch.from_numpy(res["attr_new_pos"]))
            res["attr_out_shs"], res["attr_out_opacity"], res["attr_out_cov"] = s2.numpy().astype(np.float64), o2.numpy().astype(np.float64), c2.numpy().astype(np.float64)
    finally:
        ti.KERNEL_HOOK = None
        ti.set_precision("f32")
    return res
