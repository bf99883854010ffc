"""Particle pre-pass of the MPM program on the device.

Mirrors third_party/PhysGaussian/particle_filling/filling.py -- `fill_particles` (:291-380), `get_particle_volume`
(:273-288), `init_filled_particles` (:406-447) -- with the same signatures, so gs_simulation.py:442-482 runs with
    -from particle_filling.filling import *
    +from pixie_amd.particle_filling import *
and no Taichi.  The kernels are in csrc/particle_filling.hip; there is no CPU path.  Differences a caller can observe:
the random offsets of new particles inside their cells come from a counter-based hash instead of ti.random() and the new
particles are returned in a canonical order, so the whole output is reproducible for a given `seed=`; running out of
`max_samples` raises instead of writing past the buffer.  `smooth=True` (filling.py:351-358 hands the density grid to the
third-party `mcubes.smooth(..., method="constrained", max_iters=500)` on the host) runs `smooth_constrained` below on the
device: a restatement of PyMCubes' published algorithm (the package is not installed in this image, so that one step is
unpinned; the test suite holds a scipy restatement of the same algorithm that the device version is tested against).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check

__all__ = ["fill_particles", "get_particle_volume", "init_filled_particles", "smooth_constrained"]


def _dev(t: torch.Tensor) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.PixieHipError("pixie_amd.particle_filling needs a HIP device (no CPU fallback)")
    return t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())


def _p(t):
    return C.c_void_p(t.data_ptr())


def _canonical_order(new: torch.Tensor, n_dense: int) -> torch.Tensor:
    """The kernels hand out output slots with one atomicAdd per cell, so the ORDER of the new particles depends on the
    dispatch order (their positions do not: cell + counter-based hash).  Everything downstream indexes by particle id
    (nearest-Gaussian attributes, volumes), so each phase's particles are put in lexicographic (x, y, z) order: the output
    is reproducible row for row.  Dense-cell particles stay ahead of internal-fill ones, as in filling.py:355-375."""
    def lexsort(t):
        if t.shape[0] < 2:
            return t
        for axis in (2, 1, 0):
            t = t[torch.sort(t[:, axis], stable=True).indices]
        return t
    return torch.cat([lexsort(new[:n_dense]), lexsort(new[n_dense:])], dim=0)


def density_grids(pos, opacity, cov, grid_n: int, grid_dx: float):
    """densify_grids (:26-92): returns (count int32 (n,n,n), density float32 (n,n,n)) on the device."""
    dev = _dev(pos)
    pos = pos.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    opacity = opacity.detach().to(dev, torch.float32).reshape(-1).contiguous()
    cov = cov.detach().to(dev, torch.float32).reshape(-1, 6).contiguous()
    count = torch.zeros((grid_n,) * 3, dtype=torch.int32, device=dev)
    density = torch.zeros((grid_n,) * 3, dtype=torch.float32, device=dev)
    if pos.shape[0] == 0:      # e.g. a boundary box that excludes every Gaussian: empty grids
        return count, density
    check(_lib.load().pixie_fill_densify(_p(pos), _p(opacity), _p(cov), pos.shape[0], int(grid_n), float(grid_dx), _p(count), _p(density),
                                         _lib.current_stream_ptr()), "pixie_fill_densify")
    return count, density


def _edt_squared(inside: torch.Tensor) -> torch.Tensor:
    """Exact squared Euclidean distance of every True voxel to the nearest False voxel (0 on False voxels): the separable
    min-plus form, one axis at a time -- g(i) = min_j f(j) + (i - j)^2 -- as whole-array shifted minima on the device.
    Voxels beyond the array are not background (scipy.ndimage.distance_transform_edt's convention)."""
    big = float(3 * max(inside.shape) ** 2 + 1)
    f = torch.where(inside, torch.full((), big, device=inside.device), torch.zeros((), device=inside.device)).float()
    for axis in range(3):
        n = f.shape[axis]
        g = f.clone()
        top = float(f.max())
        d = 1
        while d < n and d * d < top:
            lo = f.narrow(axis, 0, n - d) + float(d * d)       # candidate from the voxel d steps before / after
            g.narrow(axis, d, n - d).copy_(torch.minimum(g.narrow(axis, d, n - d), lo))
            hi = f.narrow(axis, d, n - d) + float(d * d)
            g.narrow(axis, 0, n - d).copy_(torch.minimum(g.narrow(axis, 0, n - d), hi))
            d += 1
        f = g
    return f


def signed_distance(density: torch.Tensor) -> torch.Tensor:
    """PyMCubes `signed_distance_function` of the binary volume (density != 0): positive inside, the surface half a voxel
    outside the outermost inside voxel."""
    inside = density != 0
    return torch.where(inside, _edt_squared(inside).double().sqrt() - 0.5, -(_edt_squared(~inside).double().sqrt() - 0.5))


_SMOOTH_WARNED = False


def _warn_smooth_unpinned():
    """Once per process: `smooth=True` is the one step of this package whose reference (PyMCubes' mcubes.smooth, a third-party wheel
    absent from the reference tree and from this image) could not be run against it -- a restatement of its published algorithm,
    NOT parity-pinned ("Deviations").  custom_sand_config.json:39 and custom_rocks_config.json:40 set it."""
    global _SMOOTH_WARNED
    if not _SMOOTH_WARNED:
        _SMOOTH_WARNED = True
        import warnings
        warnings.warn("pixie_amd.particle_filling: smooth=True runs a restatement of PyMCubes' constrained smoothing (mcubes.smooth is not available "
                      "to compare against): this sub-step is not parity-pinned to the reference; every other step of fill_particles is.",
                      RuntimeWarning, stacklevel=3)


def smooth_constrained(density: torch.Tensor, max_iters: int = 500, rel_tol: float = 1e-6, band_radius: int = 4) -> torch.Tensor:
    """mcubes.smooth(df, method="constrained", max_iters=500) on the device (see the module docstring; the algorithm is
    stated step by step in the checker's `smooth_constrained` under tests' filling oracle).  Dense-grid form of the sparse system: with b the
    band mask and, per axis, y = F u the second differences whose out-of-band neighbours are replaced by the voxel itself,
    (Q u)_c = sum_axis [ F_cc y_c + b_{c-1} y_{c-1} + b_{c+1} y_{c+1} ],  Q_cc = sum_axis [ F_cc^2 + b_{c-1} + b_{c+1} ],
    F_cc = -2 + (number of out-of-band neighbours along the axis).  float64 throughout, as the reference's numpy code."""
    u0 = signed_distance(density)
    band = u0.abs() < band_radius
    if not bool(band.any()):
        return u0
    b = band.double()

    def shift(t, axis, step):      # t at (c + step) along axis, 0 beyond the array
        out = torch.zeros_like(t)
        n = t.shape[axis]
        if step == 1:
            out.narrow(axis, 0, n - 1).copy_(t.narrow(axis, 1, n - 1))
        else:
            out.narrow(axis, 1, n - 1).copy_(t.narrow(axis, 0, n - 1))
        return out
    nb = [(shift(b, a, -1), shift(b, a, 1)) for a in range(3)]
    fcc = [-2.0 + (1.0 - m) + (1.0 - p) for m, p in nb]
    qdiag = sum(fcc[a] ** 2 + nb[a][0] + nb[a][1] for a in range(3))
    inv_d = torch.where(band, 1.0 / qdiag, torch.zeros_like(qdiag))

    def apply_q(x):                # x is zero outside the band; returns (Q x, sum |F x|^2)
        qx = torch.zeros_like(x)
        energy = x.new_zeros(())
        for a in range(3):
            y = (fcc[a] * x + shift(x, a, -1) + shift(x, a, 1)) * b
            energy = energy + (y * y).sum()
            qx = qx + fcc[a] * y + shift(y, a, -1) + shift(y, a, 1)
        return qx * b, energy
    x = u0 * b
    inf = torch.full_like(x, float("inf"))
    upper = torch.where(x < 0, x, inf)
    lower = torch.where(x > 0, x, -inf)
    upper = torch.where(upper.abs() < 1, torch.zeros_like(x), upper)
    lower = torch.where(lower.abs() < 1, torch.zeros_like(x), lower)
    upper = torch.where(band, upper, torch.zeros_like(x))
    lower = torch.where(band, lower, torch.zeros_like(x))
    check_each, weight = 10, 0.5
    cum_rel_tol = 1 - (1 - rel_tol) ** check_each
    energy_now = float(apply_q(x)[1]) / 2
    for i in range(max_iters):
        qx, _ = apply_q(x)
        x = weight * (-(qx - qdiag * x) * inv_d) + (1 - weight) * x
        x = torch.minimum(torch.maximum(x, lower), upper)
        if (i + 1) % check_each == 0:
            energy_before, energy_now = energy_now, float(apply_q(x)[1]) / 2
            # zero energy: already flat (the NumPy form of this loop divides 0 by 0 into NaN and keeps iterating to no effect)
            if energy_before <= 0.0 or (energy_before - energy_now) / energy_before < cum_rel_tol:
                break
    return torch.where(band, x, u0)


def fill_particles(pos, opacity, cov, grid_n: int, max_samples: int, grid_dx: float, density_thres=2.0, search_thres=1.0,
                   max_particles_per_cell=1, search_exclude_dir=5, ray_cast_dir=4, boundary: list = None, smooth: bool = False,
                   seed: int = 0, return_grids: bool = False):
    """filling.py:291-380.  Returns cat([pos, new particles]); with return_grids also (count, density, n_dense, n_total)."""
    dev = _dev(pos)
    lib = _lib.load()
    pos_clone = pos.detach().to(dev, torch.float32).clone()
    pos, opacity, cov = pos_clone, opacity.detach().to(dev, torch.float32), cov.detach().to(dev, torch.float32)
    new_origin = None
    if boundary is not None:
        assert len(boundary) == 6
        mask = torch.ones(pos_clone.shape[0], dtype=torch.bool, device=dev)
        max_diff = 0.0
        for i in range(3):
            mask = torch.logical_and(mask, pos_clone[:, i] > boundary[2 * i])
            mask = torch.logical_and(mask, pos_clone[:, i] < boundary[2 * i + 1])
            max_diff = max(max_diff, boundary[2 * i + 1] - boundary[2 * i])
        pos, opacity, cov = pos[mask], opacity.reshape(-1)[mask], cov.reshape(-1, 6)[mask]
        grid_dx = max_diff / grid_n
        new_origin = torch.tensor([boundary[0], boundary[2], boundary[4]], dtype=torch.float32, device=dev)
        pos = pos - new_origin
    count, density = density_grids(pos, opacity, cov, grid_n, grid_dx)
    particles = torch.empty((int(max_samples), 3), dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    st = _lib.current_stream_ptr()
    check(lib.pixie_fill_dense_cells(_p(count), _p(density), int(grid_n), float(grid_dx), float(density_thres), int(max_particles_per_cell),
                                     _p(particles), int(max_samples), _p(counter), int(seed) & 0xFFFFFFFF, st), "pixie_fill_dense_cells")
    n_dense = int(counter.item())
    print("after dense grids: ", n_dense)
    search_field = density
    if smooth:    # filling.py:351-358: the internal filling reads the constrained-smoothed signed distance of the density's support
        _warn_smooth_unpinned()
        search_field = smooth_constrained(density, max_iters=500).to(torch.float32).contiguous()
        print("smooth finished")
    check(lib.pixie_fill_internal_cells(_p(count), _p(search_field), int(grid_n), float(grid_dx), int(max_particles_per_cell), int(search_exclude_dir),
                                        int(ray_cast_dir), float(search_thres), _p(particles), int(max_samples), _p(counter),
                                        int(seed) & 0xFFFFFFFF, st), "pixie_fill_internal_cells")
    fill_num = int(counter.item())
    print("after internal grids: ", fill_num)
    if fill_num > max_samples:
        raise RuntimeError(f"fill_particles: {fill_num} new particles do not fit max_samples = {max_samples} "
                           "(the reference would write past its buffer here)")
    new = _canonical_order(particles[:fill_num], n_dense)
    if new_origin is not None:
        new = new + new_origin
    out = torch.cat([pos_clone, new], dim=0)
    if return_grids:
        return out, count, density, n_dense, fill_num
    return out


def get_particle_volume(pos, grid_n: int, grid_dx: float, unifrom: bool = False):
    """filling.py:273-288 (the misspelt keyword is the reference's)."""
    dev = _dev(pos)
    p = pos.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    scratch = torch.empty((grid_n,) * 3, dtype=torch.int32, device=dev)
    vol = torch.empty(p.shape[0], dtype=torch.float32, device=dev)
    check(_lib.load().pixie_particle_volume(_p(p), p.shape[0], int(grid_n), float(grid_dx), _p(scratch), _p(vol), _lib.current_stream_ptr()),
          "pixie_particle_volume")
    if unifrom:
        return torch.mean(vol).repeat(p.shape[0])
    return vol


def init_filled_particles(pos, shs, cov, opacity, new_pos):
    """filling.py:406-447: every new particle takes the SH coefficients, opacity and covariance of its nearest original one."""
    dev = _dev(pos)
    shs2 = shs.reshape(pos.shape[0], -1).to(dev)
    p = pos.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    q = new_pos.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    nearest = torch.empty(q.shape[0], dtype=torch.int32, device=dev)
    if q.shape[0] > 0:         # nothing was filled: the attribute arrays come back unchanged
        check(_lib.load().pixie_nearest_particle(_p(p), p.shape[0], _p(q), q.shape[0], _p(nearest), _lib.current_stream_ptr()),
              "pixie_nearest_particle")
    idx = nearest.long()
    shs_tensor = torch.cat([shs2, shs2[idx]], dim=0)
    shs_tensor = shs_tensor.view(shs_tensor.shape[0], -1, 3)
    opacity_tensor = torch.cat([opacity.to(dev), opacity.to(dev).reshape(-1)[idx].reshape(-1, 1)], dim=0)
    cov_tensor = torch.cat([cov.to(dev), cov.to(dev).reshape(-1, 6)[idx]], dim=0)
    return shs_tensor, opacity_tensor, cov_tensor
