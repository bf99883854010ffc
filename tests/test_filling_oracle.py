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


def test_anisotropic_gaussian_against_brute_force():
    """Second anchor for densify: ONE rotated, strongly anisotropic Gaussian, every touched cell re-computed from the
    definition (filling.py:13-23, :60-92) with nothing shared with the oracle's vectorised code: explicit inverse of the
    covariance, a Python loop over the cell's eight corners, the reach ceil(sqrt(largest eigenvalue) / dx) from numpy's
    eigvalsh."""
    dx, n = 0.05, 20
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]]) @ \
        np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.4), -np.sin(0.4)], [0.0, np.sin(0.4), np.cos(0.4)]])
    Cm = R @ np.diag([0.11 ** 2, 0.03 ** 2, 0.015 ** 2]) @ R.T
    pos = np.array([[0.512, 0.467, 0.533]])
    cov6 = np.array([[Cm[0, 0], Cm[0, 1], Cm[0, 2], Cm[1, 1], Cm[1, 2], Cm[2, 2]]])
    count, dens = fo.densify(pos, [0.6], cov6, n, dx)
    ci = np.floor(pos[0] / dx).astype(int)
    assert count.sum() == 1 and count[tuple(ci)] == 1
    Cinv = np.linalg.inv(Cm)
    reach = int(np.ceil(np.sqrt(np.linalg.eigvalsh(Cm).max()) / dx))
    assert reach == 3
    checked = 0
    for i in range(n):
        for j in range(n):
            for k in range(n):
                inside = max(abs(i - ci[0]), abs(j - ci[1]), abs(k - ci[2])) <= reach
                if not inside:
                    assert dens[i, j, k] == 0.0
                    continue
                acc = 0.0
                for a in (0, 1):
                    for b in (0, 1):
                        for c in (0, 1):
                            d = pos[0] - np.array([i + a, j + b, k + c]) * dx
                            acc += np.exp(-0.5 * d @ Cinv @ d)
                assert abs(dens[i, j, k] - 0.6 * acc / 8.0) < 1e-13
                checked += 1
    assert checked == (2 * reach + 1) ** 3


def torus_scene(n=40000, seed=3, R=0.28, r=0.11):
    """Points on a torus surface (axis z) centred in the unit box, small isotropic Gaussians."""
    rng = np.random.default_rng(seed)
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    pos = 0.5 + np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], 1)
    s2 = np.full(n, 0.012 ** 2)
    cov = np.zeros((n, 6)); cov[:, 0] = s2; cov[:, 3] = s2; cov[:, 5] = s2
    return pos, np.full(n, 0.9), cov


def test_torus_interior_by_ray_parity():
    """Second anchor for internal_cells: a torus (genus 1).  Cells in the central hole see the shell along +-x and +-y but
    not along +-z; cells of the tube see it in all six directions and cross it once along the parity ray.  The filled set
    must be the tube's interior as the implicit equation gives it -- every filled cell inside the tube, every cell
    comfortably inside the tube filled, nothing in the hole."""
    n, dx = 48, 1.0 / 48
    R, r = 0.28, 0.11
    pos, op, cov = torus_scene(R=R, r=r)
    count, dens = fo.densify(pos, op, cov, n, dx)
    filled = fo.internal_cells(count, dens, 1.0, exclude_dir=-1, ray_cast_dir=4)
    c = (np.indices((n, n, n)).reshape(3, -1).T + 0.5) * dx - 0.5
    rho = np.sqrt((np.sqrt(c[:, 0] ** 2 + c[:, 1] ** 2) - R) ** 2 + c[:, 2] ** 2).reshape(n, n, n)     # distance to the tube's centre circle
    assert filled.sum() > 2000
    # nothing outside the tube (the hole included) is filled -- up to the splat's own thickness: empty cells whose density
    # exceeds the threshold count as solid AND as fillable (internal_filling tests the particle count only, filling.py:204)
    assert (rho[filled] < r + 2.0 * dx).all()
    deep = rho < r - 3.5 * dx                               # cells well inside, clear of the splatted shell
    assert deep.sum() > 800 and filled[deep].mean() > 0.999
    assert not filled[n // 2, n // 2, n // 2]               # the centre of the hole


# ----------------------------------------------------------------------------- mcubes.smooth(df, method="constrained")
def _blob(n=36, seed=0):
    rng = np.random.default_rng(seed)
    g = np.arange(n) - (n - 1) / 2
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    inside = ((X / 12) ** 2 + (Y / 8) ** 2 + (Z / 10) ** 2 < 1) | ((X - 5) ** 2 + (Y - 4) ** 2 + (Z + 4) ** 2 < 30)
    inside &= ~((X + 3) ** 2 + Y ** 2 + Z ** 2 < 8)                      # with a cavity
    return inside * rng.uniform(0.5, 5.0, inside.shape)


def test_signed_distance_against_brute_force():
    df = _blob(18, 1)
    u0 = fo.signed_distance(df)
    inside = df != 0
    pts_in, pts_out = np.argwhere(inside), np.argwhere(~inside)
    for c in np.argwhere(np.ones_like(inside))[::7]:
        other = pts_out if inside[tuple(c)] else pts_in
        d = np.sqrt(((other - c) ** 2).sum(1).min()) - 0.5
        assert abs(u0[tuple(c)] - (d if inside[tuple(c)] else -d)) < 1e-12


def test_constrained_smoothing_properties():
    """PyMCubes is not installed (the step is unpinned): what the restatement must satisfy by construction of the method."""
    df = _blob()
    u0 = fo.signed_distance(df)
    s = fo.smooth_constrained(df, max_iters=500)
    band = np.abs(u0) < 4
    assert np.array_equal(s[~band], u0[~band])                            # only the band moves
    assert np.abs(s - u0).max() > 0.3                                     # and it does move
    inside = df != 0
    assert (s[inside] >= 0).all() and (s[~inside] <= 0).all()             # the surface never crosses a voxel centre
    far = np.abs(u0) >= 1
    assert (s[inside & far] >= u0[inside & far] - 1e-12).all()            # inside values only grow, outside values only fall
    assert (s[~inside & far] <= u0[~inside & far] + 1e-12).all()
    # the functional 1/2 |F u|^2 (second differences inside the band, replicated ends) went down
    def energy(u):
        e = 0.0
        for axis in range(3):
            um, up = np.roll(u, 1, axis), np.roll(u, -1, axis)
            bm, bp = np.roll(band, 1, axis), np.roll(band, -1, axis)
            y = np.where(bm, um, u) + np.where(bp, up, u) - 2 * u
            e += float((y[band] ** 2).sum())
        return e / 2
    assert energy(s) < 0.6 * energy(u0)
    # more sweeps never raise it; the result depends on the support of the density only
    assert energy(fo.smooth_constrained(df, max_iters=20)) >= energy(s) - 1e-9
    assert np.array_equal(fo.smooth_constrained((df != 0).astype(float)), s)
    # symmetric input -> symmetric output
    cube = np.zeros((24, 24, 24)); cube[6:18, 6:18, 6:18] = 1.0
    sc = fo.smooth_constrained(cube)
    assert np.allclose(sc, sc[::-1]) and np.allclose(sc, sc.transpose(1, 0, 2)) and np.allclose(sc, sc.transpose(2, 1, 0)[:, :, ::-1])
    # the corners of the cube are rounded: the field at a corner voxel is pulled onto the surface (its bound, 0), a face centre is not
    assert sc[6, 6, 6] < 0.25 and sc[6, 12, 12] >= 0.5


def test_device_implementation_of_the_smoothing_equals_the_oracle():
    """pixie_amd.particle_filling.smooth_constrained is whole-array torch arithmetic (the product runs it on the GPU inside
    fill_particles(smooth=True)); on CPU tensors it must reproduce the scipy restatement."""
    import torch
    from pixie_amd.particle_filling import signed_distance, smooth_constrained
    df = _blob(30, 2)
    assert np.abs(signed_distance(torch.from_numpy(df)).numpy() - fo.signed_distance(df)).max() < 1e-12
    assert np.abs(smooth_constrained(torch.from_numpy(df)).numpy() - fo.smooth_constrained(df)).max() < 1e-9
