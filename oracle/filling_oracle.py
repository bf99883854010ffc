"""NumPy restatement of the reference's Taichi particle-filling kernels -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module, as the checker for pixie_amd/particle_filling.py; pixie_amd/ never does.
Restates third_party/PhysGaussian/particle_filling/filling.py: compute_density :13-23, densify_grids :26-92,
fill_dense_grids :95-121 (which cells, how many points -- the points themselves are ti.random()), collision_search :124-149,
collision_times :152-183, internal_filling :186-244, compute_particle_volume :257-266, get_attr_from_closest :383-403.
PARITY UNPINNED: Taichi is not installed here and the reference holds no vectors for these kernels; the restatement is
float64 and is anchored by closed-form and brute-force cases in tests/test_filling_oracle.py (a single isotropic Gaussian's
density; one rotated anisotropic Gaussian re-computed cell by cell from the definition; a hollow shell whose interior must
fill; a shell open on the excluded side; a torus, whose filled set must be the tube interior of the implicit equation).
"""
from __future__ import annotations

import numpy as np


def densify(pos, opacity, cov6, grid_n, dx):
    pos = np.asarray(pos, np.float64); cov6 = np.asarray(cov6, np.float64); opacity = np.asarray(opacity, np.float64).reshape(-1)
    count = np.zeros((grid_n,) * 3, np.int64)
    density = np.zeros((grid_n,) * 3, np.float64)
    corners = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], np.float64)
    for p, op, c in zip(pos, opacity, cov6):
        ci = np.floor(p / dx).astype(int)
        if (ci >= 0).all() and (ci < grid_n).all():
            count[tuple(ci)] += 1
        C = np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]])
        sig, Q = np.linalg.eigh(C)
        sig = np.maximum(sig, 1e-8)
        Cinv = Q @ np.diag(1.0 / sig) @ Q.T
        r = int(np.ceil(np.sqrt(sig.max()) / dx))
        lo, hi = np.maximum(ci - r, 0), np.minimum(ci + r, grid_n - 1)
        if (lo > hi).any():
            continue
        I, J, K = np.meshgrid(*(np.arange(lo[a], hi[a] + 1) for a in range(3)), indexing="ij")
        cells = np.stack([I, J, K], -1).reshape(-1, 3).astype(np.float64)
        d = p[None, None, :] - (cells[:, None, :] + corners[None]) * dx                     # (cells, 8, 3)
        q = np.einsum("cka,ab,ckb->ck", d, Cinv, d)
        dens = op * np.exp(-0.5 * q).sum(1) / 8.0
        np.add.at(density, (I.reshape(-1), J.reshape(-1), K.reshape(-1)), dens)
    return count, density


def dense_cells(count, density, thres, max_ppc):
    """fill_dense_grids: (mask of cells that receive points, points per such cell)"""
    mask = (density > thres) & (count < max_ppc)
    return mask, np.where(mask, max_ppc - count, 0)


def _hit_beyond(solid, axis, positive):
    """collision_search for every cell at once: is there a solid cell strictly beyond it along the direction?"""
    s = np.moveaxis(solid, axis, -1)
    if positive:
        after = np.flip(np.logical_or.accumulate(np.flip(s, -1), -1), -1)
        res = np.concatenate([after[..., 1:], np.zeros_like(after[..., :1])], -1)
    else:
        before = np.logical_or.accumulate(s, -1)
        res = np.concatenate([np.zeros_like(before[..., :1]), before[..., :-1]], -1)
    return np.moveaxis(res, -1, axis)


def _rising_edges_beyond(solid, axis, positive):
    """collision_times for an EMPTY start cell (state False): number of False -> True transitions met walking away."""
    s = np.moveaxis(solid, axis, -1)
    if not positive:
        s = np.flip(s, -1)
    n = s.shape[-1]
    out = np.zeros(s.shape, np.int64)
    for i in range(n):                       # small grids only (test infrastructure)
        state = np.zeros(s.shape[:-1], bool)
        times = np.zeros(s.shape[:-1], np.int64)
        for k in range(i + 1, n):
            new = s[..., k]
            times += (new & ~state)
            state = new
        out[..., i] = times
    if not positive:
        out = np.flip(out, -1)
    return np.moveaxis(out, -1, axis)


def internal_cells(count, density, threshold, exclude_dir, ray_cast_dir):
    """internal_filling: mask of empty cells that get filled.  dir: 0:+x 1:-x 2:+y 3:-y 4:+z 5:-z."""
    solid = density > threshold
    hit = np.ones(count.shape, bool)
    for d in range(6):
        if d != exclude_dir:
            hit &= _hit_beyond(solid, d // 2, d % 2 == 0)
    if 0 <= ray_cast_dir <= 5:
        times = _rising_edges_beyond(solid, ray_cast_dir // 2, ray_cast_dir % 2 == 0)
    else:
        times = np.ones(count.shape, np.int64)
    return (count == 0) & hit & (times % 2 == 1)


def particle_volume(pos, grid_n, dx):
    pos = np.asarray(pos, np.float64)
    idx = np.floor(pos / dx).astype(int)
    flat = (idx[:, 0] * grid_n + idx[:, 1]) * grid_n + idx[:, 2]
    cnt = np.bincount(flat, minlength=grid_n ** 3)
    return dx ** 3 / cnt[flat]


def nearest(pos, new_pos):
    pos = np.asarray(pos, np.float64); new_pos = np.asarray(new_pos, np.float64)
    d = np.linalg.norm(new_pos[:, None, :] - pos[None], axis=2)
    return d.argmin(1)       # first minimum, as the strict < of the reference loop
