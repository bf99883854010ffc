"""NumPy restatement of the reference's Taichi particle-filling kernels -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module, as the checker for pixie_amd/particle_filling.py; pixie_amd/ never does.
Restates third_party/PhysGaussian/particle_filling/filling.py: compute_density :13-23, densify_grids :26-92,
fill_dense_grids :95-121 (which cells, how many points -- the points themselves are ti.random()), collision_search :124-149,
collision_times :152-183, internal_filling :186-244, compute_particle_volume :257-266, get_attr_from_closest :383-403.
PINNED to the reference's own code (round 4): Taichi is not installed here, but filling.py is Python source -- it is imported
UNMODIFIED on tests/golden/ti_shim (a NumPy interpreter of the Taichi subset it uses) and its fill_particles /
get_particle_volume / init_filled_particles are run as written on five scenes (tests/golden/make_filling_ref_golden.py ->
tests/golden/filling_ref_golden.npz).  This restatement reproduces the reference's density grids to 4e-16, its count grids
after each of the three kernels, the filled cells, the volumes and the nearest-Gaussian attributes exactly
(tests/test_filling_ref_golden.py, which also re-runs the reference live where /root/reference exists).  Not pinnable, because
the reference itself leaves them undetermined: the order of the new particles and their ti.random() offsets.  The closed-form
and brute-force anchors of tests/test_filling_oracle.py stay as a second, independent check.  The smoothing step at the end
of this file (PyMCubes, absent) is the one part that remains unpinned; see its docstring.
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


# ----------------------------------------------------------------------------- mcubes.smooth(df, method="constrained")
def signed_distance(binary):
    """PyMCubes mcubes/smoothing.py `signed_distance_function`: Euclidean distance transform, positive inside, with the
    level set half a voxel outside the outermost inside voxel:  inside: edt(inside) - 0.5,  outside: -(edt(outside) - 0.5)."""
    from scipy import ndimage as ndi
    binary = np.asarray(binary) != 0
    return np.where(binary, ndi.distance_transform_edt(binary) - 0.5, -(ndi.distance_transform_edt(~binary) - 0.5))


def smooth_constrained(df, max_iters=500, rel_tol=1e-6, band_radius=4):
    """What filling.py:351-358 does to the density grid when `smooth` is set:
        smoothed = mcubes.smooth(df, method="constrained", max_iters=500).astype(np.float32)
    PyMCubes (third party, pinned nowhere in the reference: `import mcubes`, filling.py:8) is NOT installed here, so this is a
    restatement of its published algorithm -- mcubes/smoothing.py `smooth_constrained`, the method of V. Lempitsky, "Surface
    extraction from binary volumes with higher-order smoothness", CVPR 2010 -- PARITY UNPINNED, anchored by the analytic
    cases of tests/test_filling_oracle.py:
      * the input is read as a BINARY volume (non-zero density = inside);
      * u0 = the signed Euclidean distance above; the unknowns are the voxels of the band |u0| < band_radius (4);
      * minimise  1/2 |F u|^2  over the band, F = the three axis-wise second differences [1, -2, 1] (a neighbour outside the
        band is replaced by the voxel itself), subject to  u >= u0 where u0 > 0,  u <= u0 where u0 < 0, and the bound
        relaxed to 0 where |u0| < 1 (the voxels next to the surface may move up to the surface but not across it);
      * solver: projected weighted Jacobi (weight 1/2) on Q = F^T F, at most `max_iters` sweeps, stopping when the energy
        improved by less than 1 - (1 - rel_tol)^10 over the last 10 sweeps;
      * result: u0 with the band replaced by the solution (float64; the caller casts to float32)."""
    from scipy import sparse
    u0 = signed_distance(df)
    band = np.abs(u0) < band_radius
    nvar = int(band.sum())
    if nvar == 0:
        return u0
    idx = np.full(band.shape, -1, np.int64)
    idx[band] = np.arange(nvar)
    rows, cols, vals = [], [], []
    me = idx[band]
    for axis in range(3):
        diag = np.full(nvar, -2.0)
        for step in (-1, 1):
            nb = np.roll(idx, -step, axis=axis)
            edge = [slice(None)] * 3
            edge[axis] = -1 if step == 1 else 0
            nb[tuple(edge)] = -1                       # no wrap-around: the neighbour beyond the array is "outside the band"
            nb = nb[band]
            has = nb >= 0
            rows.append(3 * me[has] + axis); cols.append(nb[has]); vals.append(np.ones(int(has.sum())))
            diag[~has] += 1.0
        rows.append(3 * me + axis); cols.append(me); vals.append(diag)
    F = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * nvar, nvar))
    Q = (F.T @ F).tocsr()
    x = u0[band].astype(np.float64)
    upper = np.where(x < 0, x, np.inf)
    lower = np.where(x > 0, x, -np.inf)
    upper[np.abs(upper) < 1] = 0
    lower[np.abs(lower) < 1] = 0
    d_inv = 1.0 / Q.diagonal()
    R = Q - sparse.diags(Q.diagonal())
    check_each, weight = 10, 0.5
    cum_rel_tol = 1 - (1 - rel_tol) ** check_each
    energy_now = float(x @ (Q @ x)) / 2
    for i in range(max_iters):
        x = weight * (-d_inv * (R @ x)) + (1 - weight) * x
        x = np.minimum(np.maximum(x, lower), upper)
        if (i + 1) % check_each == 0:
            energy_before, energy_now = energy_now, float(x @ (Q @ x)) / 2
            if (energy_before - energy_now) / energy_before < cum_rel_tol:
                break
    out = u0.copy()
    out[band] = x
    return out
