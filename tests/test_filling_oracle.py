"""Anchors for oracle/filling_oracle.py (CPU): closed-form cases of the reference's particle-filling kernels."""
import numpy as np

from oracle import filling_oracle as fo


def shell_scene(n=5000, seed=0, radius=0.3, open_bottom=False):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    if open_bottom:
        d = d[d[:, 2] > -0.55]
    pos = 0.5 + radius * d
    s2 = rng.uniform(0.012, 0.02, len(pos)) ** 2
    cov = np.zeros((len(pos), 6)); cov[:, 0] = s2; cov[:, 3] = 1.3 * s2; cov[:, 5] = 0.8 * s2
    cov[:, 1] = 0.1 * s2
    return pos, np.full(len(pos), 0.9), cov


def test_single_isotropic_gaussian_density():
    dx, n = 0.1, 10
    pos = np.array([[0.47, 0.52, 0.55]]); s2 = 0.05 ** 2
    cov = np.array([[s2, 0, 0, s2, 0, s2]])
    count, dens = fo.densify(pos, [0.7], cov, n, dx)
    assert count.sum() == 1 and count[4, 5, 5] == 1
    # cell (4,5,5): mean over its 8 corners of 0.7 exp(-|p - corner|^2 / (2 s2))
    corners = np.array([[4 + i, 5 + j, 5 + k] for i in (0, 1) for j in (0, 1) for k in (0, 1)]) * dx
    want = 0.7 * np.exp(-0.5 * ((pos - corners) ** 2).sum(1) / s2).mean()
    assert abs(dens[4, 5, 5] - want) < 1e-12
    r = int(np.ceil(0.05 / dx))     # reach of the splat: ceil(sigma / dx) cells
    touched = np.argwhere(dens > 0)
    assert (np.abs(touched - np.array([4, 5, 5])) <= r).all() and len(touched) == (2 * r + 1) ** 3


def test_hollow_shell_fills_inside_only():
    n, dx = 32, 1.0 / 32
    pos, op, cov = shell_scene()
    count, dens = fo.densify(pos, op, cov, n, dx)
    dense, per = fo.dense_cells(count, dens, 2.0, 1)
    assert dense.sum() > 0 and (per[dense] == 1).all()
    count2 = np.where(dense, 1, count)
    inside = fo.internal_cells(count2, dens, 1.0, exclude_dir=5, ray_cast_dir=4)
    idx = np.argwhere(inside)
    rad = np.linalg.norm((idx + 0.5) * dx - 0.5, axis=1)
    assert len(idx) > 500 and rad.max() < 0.3 and (count2[inside] == 0).all()
    # nothing outside the shell is filled, and the centre is
    assert inside[16, 16, 16] and not inside[1, 1, 1]


def test_shell_open_on_the_excluded_side():
    """exclude_dir = 5 (-z): cells above a hole in the bottom still count as enclosed; with no exclusion they do not."""
    n, dx = 32, 1.0 / 32
    pos, op, cov = shell_scene(open_bottom=True)
    count, dens = fo.densify(pos, op, cov, n, dx)
    a = fo.internal_cells(count, dens, 1.0, exclude_dir=5, ray_cast_dir=4)
    b = fo.internal_cells(count, dens, 1.0, exclude_dir=-1, ray_cast_dir=4)
    assert a[16, 16, 16] and not b[16, 16, 16] and b.sum() < a.sum()


def test_volume_and_nearest():
    rng = np.random.default_rng(1)
    pos = rng.uniform(0.05, 0.95, size=(500, 3))
    vol = fo.particle_volume(pos, 8, 0.125)
    assert abs((vol).sum() - 0.125 ** 3 * len(np.unique(np.floor(pos / 0.125).astype(int), axis=0))) < 1e-12
    new = rng.uniform(0, 1, size=(50, 3))
    idx = fo.nearest(pos, new)
    assert (np.linalg.norm(new - pos[idx], axis=1) <= np.linalg.norm(new[:, None] - pos[None], axis=2).min(1) + 1e-15).all()
