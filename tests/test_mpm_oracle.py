"""CPU suite for the MPM half.

Analytic known-answer tests of the MPM oracle and the device arithmetic of
pixie_amd/csrc/mpm_math.h compared against the oracle on the host.  (The oracle's pin to the reference's own solver code --
mpm_solver_warp.py run unmodified on tests/golden/wp_shim -- is tests/test_mpm_ref_golden.py.)
"""
import numpy as np
import pytest

from oracle.mpm_oracle import OracleMPM, svd3
from pixie_amd.synthetic import apply_scene, mpm_ball_scene
from tests import _harness


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make_oracle(scene, precision="f32", n=None):
    n = n or scene["x"].shape[0]
    o = OracleMPM(n, scene["n_grid"], scene["grid_lim"], precision)
    o.load_initial_data(scene["x"][:n], scene["vol"][:n], scene["cov"][:n])
    return o


# ----------------------------------------------------------------------------- oracle KATs
def test_svd_convention_and_accuracy():
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        U, S, V = svd3(A, "f64")
        assert np.abs(U @ np.diag(S) @ V.T - A).max() < 1e-12
        assert abs(np.linalg.det(U) - 1) < 1e-12 and abs(np.linalg.det(V) - 1) < 1e-12
        assert S[0] >= S[1] >= abs(S[2]) and np.sign(S[2]) == np.sign(np.linalg.det(A))
        assert np.allclose(np.abs(S), np.linalg.svd(A, compute_uv=False), atol=1e-12)


def test_partition_of_unity_and_momentum_conservation():
    sc = mpm_ball_scene(4000, seed=3)
    for prec, tol in (("f32", 2e-5), ("f64", 1e-12)):
        o = make_oracle(sc, prec)
        apply_scene(o, sc)
        rng = np.random.default_rng(1)
        o.field("v")[:] = rng.normal(size=(4000, 3))
        o.field("C")[:] = rng.normal(size=(4000, 3, 3))
        o.phase("zero_grid")
        o.phase("compute_stress", sc["dt"])
        assert np.abs(o.field("stress")).max() == 0.0  # F_trial = I  =>  tau = 0
        o.phase("p2g", sc["dt"])
        m, v, C, x = o.field("mass").astype(np.float64), o.field("v").astype(np.float64), o.field("C").astype(np.float64), o.field("x").astype(np.float64)
        assert abs(o.field("grid_m").astype(np.float64).sum() / m.sum() - 1) < tol          # sum_i w_ip = 1
        mom = o.field("grid_v_in").astype(np.float64).reshape(-1, 3).sum(0)
        assert rel_l2(mom, (m[:, None] * v).sum(0)) < 20 * tol                                # sum_i w_ip (x_i - x_p) = 0
        assert o.out_of_bounds == 0


def test_rigid_translation_keeps_F_identity():
    sc = mpm_ball_scene(3000, seed=5)
    sc["params"] = dict(material="jelly", g=[0, 0, 0], E=1e5, nu=0.3, density=1000.0)
    sc["bcs"] = []; sc["fix_ground"] = None
    o = make_oracle(sc, "f64")
    apply_scene(o, sc, per_particle=False)
    o.field("v")[:] = np.array([0.3, -0.2, 0.1])
    x0 = o.field("x").copy()
    o.run(sc["dt"], 20)
    assert np.abs(o.field("F") - np.eye(3)).max() < 1e-10
    assert np.abs(o.field("stress")).max() < 1e-4
    assert np.abs(o.field("x") - x0 - 20 * sc["dt"] * np.array([0.3, -0.2, 0.1])).max() < 1e-10


def test_uniform_stretch_fcr_closed_form():
    """F = sI  =>  tau = (2 mu (s-1) s + lam s^3 (s^3-1)) I   (mpm_utils.py:10-17)"""
    sc = mpm_ball_scene(64, seed=2)
    o = make_oracle(sc, "f64")
    sc["bcs"] = []; sc["fix_ground"] = None
    apply_scene(o, sc)
    for s in (0.9, 1.0, 1.07):
        o.field("F_trial")[:] = s * np.eye(3)
        o.phase("compute_stress", sc["dt"])
        mu, lam = o.field("mu").astype(np.float64), o.field("lam").astype(np.float64)
        want = 2 * mu * (s - 1) * s + lam * s ** 3 * (s ** 3 - 1)
        got = o.field("stress")
        assert np.allclose(got[:, 0, 0], want, rtol=1e-9, atol=1e-6)
        assert np.abs(got[:, 0, 1]).max() < 1e-6


def test_cuboid_reset_and_impulse_windows():
    sc = mpm_ball_scene(2000, seed=7)
    o = make_oracle(sc, "f32")
    apply_scene(o, sc)
    m = o.field("mass").copy()
    o.p2g2p(0, sc["dt"])
    # after one substep of the tree scenario the mean velocity is the impulse f/m*dt, damped, except in the slab
    assert o.field("v")[:, 0].mean() < 0
    z = sc["x"][:, 2]
    bottom = z < z.min() + 0.01
    assert np.abs(o.field("v")[bottom]).max() < np.abs(o.field("v")[~bottom]).mean()
    v1 = o.field("v").copy()
    o.p2g2p(1, sc["dt"])  # impulse window [0, dt) is closed now
    assert np.abs(o.field("v")).mean() < 2 * np.abs(v1).mean()


# ----------------------------------------------------------------------------- device math on the host
def _rand_F(rng, n, scale):
    def rot(n):
        q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        w, x, y, z = q.T
        return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    s = 1 + scale * rng.uniform(-1, 1, size=(n, 3))
    return (rot(n) @ (s[:, :, None] * np.transpose(rot(n), (0, 2, 1)))).astype(np.float32)


def test_device_svd_matches_oracle_convention():
    lib = _harness.load()
    rng = np.random.default_rng(4)
    n = 5000
    for scale in (1e-6, 0.05, 0.5):
        A = np.ascontiguousarray(_rand_F(rng, n, scale))
        U = np.zeros((n, 3, 3), np.float32); S = np.zeros((n, 3), np.float32); V = np.zeros((n, 3, 3), np.float32)
        lib.hh_svd3(n, A.ctypes.data, U.ctypes.data, S.ctypes.data, V.ctypes.data)
        assert np.abs(U @ (S[:, :, None] * np.transpose(V, (0, 2, 1))) - A).max() < 3e-6
        assert np.abs(np.linalg.det(U.astype(np.float64)) - 1).max() < 1e-5
        assert np.abs(np.linalg.det(V.astype(np.float64)) - 1).max() < 1e-5
        sref = np.linalg.svd(A.astype(np.float64), compute_uv=False)
        assert np.abs(S - sref).max() < 3e-6
        Uo, _, Vto = np.linalg.svd(A.astype(np.float64))
        assert np.abs(U @ np.transpose(V, (0, 2, 1)) - Uo @ Vto).max() < 3e-6  # polar rotation, convention free
    # inverted element: the sign goes to the last singular value
    A = np.diag([1.2, 0.9, -0.7]).astype(np.float32)[None].copy()
    U = np.zeros((1, 3, 3), np.float32); S = np.zeros((1, 3), np.float32); V = np.zeros((1, 3, 3), np.float32)
    lib.hh_svd3(1, A.ctypes.data, U.ctypes.data, S.ctypes.data, V.ctypes.data)
    assert S[0, 0] > 0 and S[0, 1] > 0 and S[0, 2] < 0


def test_left_stretch_is_the_left_half_of_the_svd():
    """left_stretch (mpm_math.h, host build): F F^T = U diag(sig^2) U^T with U a proper rotation, |sig| the singular values,
    the sign of det F on the one of smallest magnitude (Warp's convention for wp.svd3, which is all the constitutive laws
    use of the ordering), and a lane's sweep count: 0 for the identity, 1-2 for the strains of a stable simulation, <= 4 for
    arbitrary matrices (the wave runs the maximum over its lanes)."""
    lib = _harness.load()
    rng = np.random.default_rng(14)
    n = 5000
    for scale, max_sweeps in ((0.0, 0), (1e-6, 2), (0.02, 3), (0.5, 4), ("wild", 5)):
        if scale == "wild":
            A = rng.normal(size=(n, 3, 3)).astype(np.float32)
        else:
            A = np.ascontiguousarray(_rand_F(rng, n, scale))
        U = np.zeros((n, 3, 3), np.float32); S = np.zeros((n, 3), np.float32); sw = np.zeros(n, np.int32)
        lib.hh_left_stretch(n, A.ctypes.data, U.ctypes.data, S.ctypes.data, sw.ctypes.data)
        A64, U64 = A.astype(np.float64), U.astype(np.float64)
        b = A64 @ np.transpose(A64, (0, 2, 1))
        rec = U64 @ ((S.astype(np.float64) ** 2)[:, :, None] * np.transpose(U64, (0, 2, 1)))
        print(f"left_stretch scale {scale}: sweeps histogram {np.bincount(sw, minlength=7).tolist()}, |U s^2 U^T - F F^T| {np.abs(rec - b).max() / np.abs(b).max():.1e}, "
              f"orthogonality {np.abs(U64 @ np.transpose(U64, (0, 2, 1)) - np.eye(3)).max():.1e}")
        assert sw.max() <= max_sweeps
        assert np.abs(rec - b).max() / np.abs(b).max() < 2e-6
        assert np.abs(U64 @ np.transpose(U64, (0, 2, 1)) - np.eye(3)).max() < 2e-6
        assert np.abs(np.linalg.det(U64) - 1).max() < 1e-5
        sref = np.linalg.svd(A64, compute_uv=False)
        # float32 resolution of a singular value: 3e-7 of the largest one (the entries of F are rounded at that level) + 4e-6 of itself
        assert (np.abs(np.sort(np.abs(S), axis=1)[:, ::-1] - sref) / (4e-6 * sref + 3e-7 * sref[:, :1])).max() < 1.0
        neg = (S < 0)
        det = np.linalg.det(A64)
        assert (neg.sum(1) == (det < 0)).all()                                        # one negative value iff det F < 0 ...
        assert (np.abs(S)[neg] == np.abs(S).min(1)[neg.any(1)]).all()                 # ... and it is the smallest in magnitude


@pytest.mark.parametrize("material,scale,ys,inverted", [(0, 0.05, 0.0, 0), (0, 1e-3, 0.0, 0), (0, 0.3, 0.0, 0), (0, 0.3, 0.0, 1),
                                                        (1, 0.08, 2.0e3, 0), (1, 0.01, 1.0e9, 0), (1, 0.002, 50.0, 0), (1, 0.08, 2.0e3, 1),
                                                        (2, 0.05, 0.0, 0), (2, 0.002, 0.0, 0), (3, 0.08, 1.0e3, 0), (3, 0.08, 1.0e3, 1),
                                                        (5, 0.08, 2.0e3, 0), (5, 0.002, 100.0, 0), (5, 0.01, 1.0e9, 0), (5, 0.08, 2.0e3, 1), (6, 0.05, 0.0, 0)])
def test_device_stress_matches_oracle(material, scale, ys, inverted):
    """return_map_and_stress of mpm_math.h (host build) -- ONE left-stretch decomposition, the return mapping on its singular
    values, tau = U diag(t) U^T, F = F_trial + (U diag(s'/s - 1) U^T) F_trial -- against the C oracle's compute_stress, which
    restates the reference's TWO wp.svd3 calls and its matrix-product stress formulas line by line (and is pinned to the
    reference's own code by tests/test_mpm_ref_golden.py).  The float64 oracle is the yardstick; the float32 oracle -- the
    reference's own arithmetic in its own precision -- is measured beside it and the product must be at least as close.
    `inverted`: a fifth of the particles start with det F < 0 (the materials whose reference code survives that: the
    Drucker-Prager and water laws are NaN there in the reference itself)."""
    lib = _harness.load()
    rng = np.random.default_rng(10 + material)
    n = 3000
    sc = mpm_ball_scene(n, seed=1)
    o, o32 = make_oracle(sc, "f64"), make_oracle(sc, "f32")
    sc["bcs"] = []; sc["fix_ground"] = None
    Ft = np.ascontiguousarray(_rand_F(rng, n, scale))
    if inverted:
        bad = rng.random(n) < 0.2
        Ft[bad] = Ft[bad] @ np.diag([1.0, 1.0, -0.6]).astype(np.float32)
        Ft[bad & (rng.random(n) < 0.3), :, 1] *= 0.02     # some crushed below the 0.01 clamp of the Hencky strain on top
        Ft = np.ascontiguousarray(Ft)
    for oo in (o, o32):
        apply_scene(oo, sc)
        oo.field("F_trial")[:] = Ft
        oo.field("material")[:] = material
        oo.field("yield_stress")[:] = ys
        oo._lib.mpm_set_scalar(oo._h, b"hardening", 1.0)
        oo._lib.mpm_set_scalar(oo._h, b"xi", 0.05)
        oo._lib.mpm_set_scalar(oo._h, b"plastic_viscosity", 10.0)
        oo.finalize_mu_lam_bulk()
    mu, lam, bulk, ysv = (o.field(f).astype(np.float32) for f in ("mu", "lam", "bulk", "yield_stress"))
    mat = np.full(n, material, np.int32)
    F = np.zeros((n, 3, 3), np.float32); tau = np.zeros((n, 3, 3), np.float32)
    alpha = float(np.sqrt(2 / 3) * 2 * np.sin(25 / 180 * 3.14159265) / (3 - np.sin(25 / 180 * 3.14159265)))
    lib.hh_stress(n, mat.ctypes.data, Ft.ctypes.data, mu.ctypes.data, lam.ctypes.data, bulk.ctypes.data, ysv.ctypes.data,
                  alpha, 1.0, 0.05, 0.1, 10.0, 1e-4, F.ctypes.data, tau.ctypes.data)
    o.phase("compute_stress", 1e-4); o32.phase("compute_stress", 1e-4)
    scale_tau = max(np.abs(o.field("stress")).max(), 1e-30)
    e_F, d_F = rel_l2(F, o.field("F")), rel_l2(o32.field("F"), o.field("F"))
    e_t, d_t = np.abs(tau - o.field("stress")).max() / scale_tau, np.abs(o32.field("stress") - o.field("stress")).max() / scale_tau
    moved = float((np.abs(o.field("F") - Ft).reshape(n, -1).max(1) > 1e-7).mean())
    print(f"material {material} strain {scale} ys {ys:g} inverted {inverted}: {100 * moved:.0f} % yield; F {e_F:.1e} (float32 oracle {d_F:.1e}), "
          f"stress {e_t:.1e} of max (float32 oracle {d_t:.1e})")
    assert np.isfinite(F).all() and np.isfinite(tau).all()
    assert np.abs(tau - np.transpose(tau, (0, 2, 1))).max() == 0.0 or material == 0       # symmetric by construction
    assert e_F < 5e-7
    if material != 6:
        # at a strain of 1e-3 the stress is a 1e-3 effect of matrices rounded at 6e-8: 1e-4 is float32's own floor there
        assert e_t < (1e-4 if scale < 5e-3 else 2e-5) and e_t < max(1.5 * d_t, 1e-6)
    assert rel_l2(ysv, o.field("yield_stress")) < 2e-5 if ys else True
    assert np.allclose(mu, o.field("mu"), rtol=1e-6) and np.allclose(lam, o.field("lam"), rtol=1e-6)


def _rot(rng, n):
    q, _ = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    q[np.linalg.det(q) < 0, :, 0] *= -1.0
    return q


@pytest.mark.parametrize("material,ys", [(1, 5e3), (5, 5e3), (3, 5e3), (2, 0.0)])
@pytest.mark.parametrize("case", ["two_small_axes", "rank_deficient"])
def test_crushed_and_rank_deficient_elements(case, material, ys):
    """ADVICE r5, two holes of the single-decomposition constitutive path (mpm_math.h), on F_trial = U diag(sigma) V^T with random rotations:
    `two_small_axes`: sigma = (1, 1e-2, 2e-2) x (1 +- 10 %).  The Jacobi iteration runs on F F^T, whose float32 entries resolve the frame of
      the two small axes only to 1e-7 / (4e-4 - 1e-4) = 3e-4 rad; refine_crushed_frame finishes it on F itself.  Stress and F must meet the
      bars of test_device_stress_matches_oracle's crushed cases against the float64 oracle (which runs the reference's two full SVDs).
    `rank_deficient`: the smallest singular value EXACTLY zero (F_trial = U diag(s0, s1, 0) V^T assembled in float64, then the null direction
      projected out in float32).  sigma'/sigma - 1 is infinite there: the correction form of the returned F gave NaN, which P2G spreads;
      rebuild_rank_deficient re-assembles U diag(sigma') V^T like the reference (its max(sigma, 0.01) clamp).  Finite, and equal to the
      float64 oracle's F where that is unique (one null axis)."""
    lib = _harness.load()
    rng = np.random.default_rng(77 + material)
    n = 2000
    U, V = _rot(rng, n), _rot(rng, n)
    if case == "two_small_axes":
        sig = np.array([1.0, 1e-2, 2e-2]) * rng.uniform(0.9, 1.1, (n, 3))
    else:
        sig = np.stack([rng.uniform(0.8, 1.2, n), rng.uniform(0.5, 0.9, n), np.zeros(n)], 1)
    Ft = np.einsum("nij,nj,nkj->nik", U, sig, V).astype(np.float32)
    if case == "rank_deficient":    # make the float32 matrix singular to rounding: remove what the cast left along the null direction
        Ft = (Ft.astype(np.float64) - np.einsum("ni,nj->nij", np.einsum("nij,nj->ni", Ft.astype(np.float64), V[:, :, 2]), V[:, :, 2])).astype(np.float32)
    Ft = np.ascontiguousarray(Ft)
    sc = mpm_ball_scene(n, seed=1)
    sc["bcs"] = []; sc["fix_ground"] = None
    o, o32 = make_oracle(sc, "f64"), make_oracle(sc, "f32")
    for oo in (o, o32):
        apply_scene(oo, sc)
        oo.field("F_trial")[:] = Ft
        oo.field("material")[:] = material
        oo.field("yield_stress")[:] = ys
        oo._lib.mpm_set_scalar(oo._h, b"hardening", 1.0)
        oo._lib.mpm_set_scalar(oo._h, b"xi", 0.05)
        oo._lib.mpm_set_scalar(oo._h, b"plastic_viscosity", 10.0)
        oo.finalize_mu_lam_bulk()
    mu, lam, bulk, ysv = (o.field(f).astype(np.float32) for f in ("mu", "lam", "bulk", "yield_stress"))
    mat = np.full(n, material, np.int32)
    F = np.zeros((n, 3, 3), np.float32); tau = np.zeros((n, 3, 3), np.float32)
    alpha = float(np.sqrt(2 / 3) * 2 * np.sin(25 / 180 * 3.14159265) / (3 - np.sin(25 / 180 * 3.14159265)))
    lib.hh_stress(n, mat.ctypes.data, Ft.ctypes.data, mu.ctypes.data, lam.ctypes.data, bulk.ctypes.data, ysv.ctypes.data,
                  alpha, 1.0, 0.05, 0.1, 10.0, 1e-4, F.ctypes.data, tau.ctypes.data)
    o.phase("compute_stress", 1e-4); o32.phase("compute_stress", 1e-4)
    ok = np.isfinite(o.field("F")).reshape(n, -1).all(1) & np.isfinite(o.field("stress")).reshape(n, -1).all(1)
    # (Drucker-Prager on an F that does NOT yield keeps sigma = 0 and takes its logarithm -- in the reference's own formula, mpm_utils.py:71-86;
    # every other law clamps.  The product must be finite wherever the reference's algorithm is.)
    # A singular F is outside that law's domain: whether its SVD returns 0 or 1e-9 for the null axis decides between -inf and a finite
    # garbage value, in the reference, in either oracle and in the product alike -- there only F must stay finite and the comparison runs
    # over the particles where all three stresses are finite.)
    assert np.isfinite(F).all()
    if material == 2:
        ok = ok & np.isfinite(tau).reshape(n, -1).all(1) & np.isfinite(o32.field("stress")).reshape(n, -1).all(1)
    assert np.isfinite(tau[ok]).all(), int((~np.isfinite(tau[ok])).sum())
    assert ok.all() or material == 2
    scale_tau = max(np.abs(o.field("stress")[ok]).max(), 1e-30)
    e_F, d_F = rel_l2(F[ok], o.field("F")[ok]), rel_l2(o32.field("F")[ok], o.field("F")[ok])
    e_t = np.abs(tau[ok] - o.field("stress")[ok]).max() / scale_tau
    d_t = np.abs(o32.field("stress")[ok] - o.field("stress")[ok]).max() / scale_tau
    moved = float((np.abs(o.field("F")[ok] - Ft[ok]).reshape(int(ok.sum()), -1).max(1) > 1e-7).mean())
    print(f"{case}, material {material}: {100 * moved:.0f} % moved, {int(ok.sum())} of {n} finite in the float64 oracle; "
          f"F {e_F:.1e} (float32 oracle {d_F:.1e}), stress {e_t:.1e} of max (float32 oracle {d_t:.1e})")
    assert ok.sum() > 0.5 * n or material == 2
    if ok.any():
        assert e_F < max(5e-6, 1.5 * d_F) and e_t < max(5e-5, 1.5 * d_t)


def test_polar_iteration_converges_with_and_without_scaling():
    """polar_rotation (mpm_math.h, host build) over the whole range it is used on: nearly rigid inputs, the strains of a stable
    simulation, singular values from 0.15 to 3.  The iteration must settle within its six steps and give the polar factor U V^T
    to float32 accuracy; det <= 0 must be refused (the caller then takes the SVD route).  (Written for a variant that applied
    the Frobenius scaling only far from a rotation and left after one step for nearly rigid particles -- `unscaled` below is the
    region where that variant ran the plain iteration; it was timed equal to this one on the GPU and not kept,
    profiles/r4x_mpm_polar_iteration_variants.txt.)"""
    lib = _harness.load()
    rng = np.random.default_rng(5)
    def rot(n):
        q = np.linalg.qr(rng.normal(size=(n, 3, 3)))[0]
        return q * np.sign(np.linalg.det(q))[:, None, None]
    cases = []
    for lo, hi, n in ((0.99995, 1.00005, 1500), (0.9997, 1.0003, 1500), (0.985, 1.015, 2000), (0.9, 1.1, 2000), (0.7, 1.4, 3000), (0.28, 1.6, 3000), (0.15, 3.0, 2000)):
        cases.append(rng.uniform(lo, hi, size=(n, 3)))
    corners = np.array([[1.32, 1.32, 0.29], [1.58, 0.9, 0.45], [1.45, 0.62, 0.97], [1.35, 0.74, 1.0], [1.0, 1.0, 1.0], [1.2, 1.2, 1.2], [0.8, 0.8, 0.8]])
    sv = np.concatenate(cases + [corners])
    n = len(sv)
    U, V = rot(n), rot(n)
    F = np.ascontiguousarray((U * sv[:, None, :]) @ V.transpose(0, 2, 1), dtype=np.float32)
    R = np.zeros_like(F); ok = np.zeros(n, np.int32)
    lib.hh_polar(n, F.ctypes.data, R.ctypes.data, ok.ctypes.data)
    want = U @ V.transpose(0, 2, 1)
    unscaled = (np.abs((sv ** 2).sum(1) - 3) <= 0.5) & (np.abs(sv.prod(1) - 1) <= 0.5)
    assert 6000 < unscaled.sum() < n - 1000                         # both regions are populated
    assert ok[unscaled].all()                                        # the plain iteration settles wherever it is chosen
    assert ok.mean() > 0.99
    err = np.abs(R - want).reshape(n, -1).max(1)
    print("polar: settled", ok.mean(), "max |R - U V^T|", err[ok == 1].max(), "unscaled region", err[unscaled].max())
    assert err[ok == 1].max() < 5e-6 and err[unscaled].max() < 2e-6
    # reflected input: refused
    Fm = np.ascontiguousarray(F[:50]).copy(); Fm[:, :, 0] *= -1
    okm, Rm = np.ones(50, np.int32), np.zeros_like(Fm)
    lib.hh_polar(50, Fm.ctypes.data, Rm.ctypes.data, okm.ctypes.data)
    assert not okm.any()


def test_fixed_corotated_stress_series_against_float64():
    """kirchhoff_stress (mpm_math.h, host build), material 0: at small strain (|F F^T - I|_F <= 0.12, det F > 0) the stress
    2 mu (F - R) F^T + lam J (J - 1) I is evaluated as 2 mu (b - sqrt b), b = F F^T, by a degree-5 polynomial in E = b - I
    (no rotation, a third of the instructions); above, by the Newton polar iteration as before.  Against the float64 formula
    with scipy's polar decomposition on the SAME float32 F: under arbitrary rotations the series is several times closer than
    the rotation route was (6e-8 against 2.8e-7 of 2 mu, absolute), without rotation it keeps full relative precision, and the
    two routes join without a step at the threshold."""
    from scipy.linalg import polar
    lib = _harness.load()
    rng = np.random.default_rng(0)

    def rot(n):
        q = np.linalg.qr(rng.normal(size=(n, 3, 3)))[0]
        return q * np.sign(np.linalg.det(q))[:, None, None]

    def device(F32, mu, lam):
        n = len(F32)
        mat = np.zeros(n, np.int32); F = np.zeros((n, 3, 3), np.float32); tau = np.zeros((n, 3, 3), np.float32)
        muv = np.full(n, mu, np.float32); lamv = np.full(n, lam, np.float32); z = np.zeros(n, np.float32); ys = np.zeros(n, np.float32)
        lib.hh_stress(n, mat.ctypes.data, F32.ctypes.data, muv.ctypes.data, lamv.ctypes.data, z.ctypes.data, ys.ctypes.data,
                      0.0, 0.0, 0.0, 0.0, 0.0, 1e-4, F.ctypes.data, tau.ctypes.data)
        assert np.array_equal(F, F32)
        return tau

    def truth(F32, mu, lam):
        out = np.zeros((len(F32), 3, 3))
        for p, F in enumerate(F32.astype(np.float64)):
            R, _ = polar(F); J = np.linalg.det(F)
            out[p] = 2 * mu * (F - R) @ F.T + lam * J * (J - 1) * np.eye(3)
        return out

    def sample(eps, rotated, n=1500):
        S = rng.normal(size=(n, 3, 3)); S = (S + S.transpose(0, 2, 1)) / 2
        S *= eps / np.abs(np.linalg.eigvalsh(S)).max(1)[:, None, None]
        R = rot(n) if rotated else np.tile(np.eye(3), (n, 1, 1))
        return np.ascontiguousarray((R @ (np.eye(3) + S)).astype(np.float32))

    mu, lam = 1.0, 0.0                    # the deviatoric part; lam J (J - 1) is common to both routes
    for eps in (1e-6, 1e-4, 1e-3, 1e-2, 0.03):
        F32 = sample(eps, True)
        err = np.abs(device(F32, mu, lam) - truth(F32, mu, lam)).max()
        assert err < 1.5e-7, (eps, err)                                   # measured 5.8e-8 .. 6.2e-8 (rotation route: 2.4e-7 .. 2.9e-7)
        F32 = sample(eps, False)
        tt = truth(F32, mu, lam)
        rel = np.abs(device(F32, mu, lam) - tt).reshape(len(F32), -1).max(1) / np.abs(tt).reshape(len(F32), -1).max(1)
        assert rel.max() < 1e-6, (eps, rel.max())                         # measured median 7e-8
    # across the threshold (|E|_F = 0.12 is a stretch of ~5 %): lanes on either side, same accuracy class, symmetric output
    F32 = sample(0.05, True, 3000)
    E = F32.astype(np.float64) @ F32.astype(np.float64).transpose(0, 2, 1) - np.eye(3)
    small = (E ** 2).sum((1, 2)) <= 0.0144
    assert 0.2 < small.mean() < 0.8
    t = device(F32, 1.0, 1.5); tt = truth(F32, 1.0, 1.5)
    err = np.abs(t - tt).reshape(len(F32), -1).max(1)
    assert err[small].max() < 6e-7 and err[~small].max() < 1.5e-6         # (the J (J - 1) term in float32 carries ~2e-7 by itself)
    assert np.array_equal(t, t.transpose(0, 2, 1))
    # a reflection of a nearly rigid F has b = F F^T ~ I too: it must NOT take the series (det F < 0: the SVD route's convention)
    Fm = sample(1e-3, True, 200); Fm[:, :, 0] *= -1
    Fm = np.ascontiguousarray(Fm)
    assert np.abs(device(Fm, 1.0, 0.0)).max() > 1.0                       # 2 mu (F - R) F^T with a proper R is O(1) there, not O(1e-3)


def test_device_stencil_matches_oracle():
    lib = _harness.load()
    rng = np.random.default_rng(3)
    n = 1000
    x = rng.uniform(0.3, 1.7, size=(n, 3)).astype(np.float32)
    base = np.zeros((n, 3), np.int32); w = np.zeros((n, 3, 3), np.float32); dw = np.zeros((n, 3, 3), np.float32)
    inv_dx = np.float32(50 / 2.0)
    lib.hh_stencil(n, x.ctypes.data, inv_dx, base.ctypes.data, w.ctypes.data, dw.ctypes.data)
    gp = x * inv_dx
    assert np.array_equal(base, (gp - np.float32(0.5)).astype(np.int32))
    assert np.abs(w.sum(2) - 1).max() < 1e-6 and np.abs(dw.sum(2)).max() < 1e-6


def test_config3_fixture_is_this_oracle():
    """tests/golden/mpm_config3.npz (the 100 000-particle, 1 000-substep trajectory the GPU test compares the HIP solver
    with) was produced by oracle/mpm_oracle.c: re-running its first checkpoint (20 substeps, ~6 s) with the oracle in
    the tree must reproduce it to the last bit (serial C, same compiler flags), so the fixture cannot drift from the
    oracle source unnoticed."""
    import os
    from pixie_amd.synthetic import apply_scene, mpm_ball_scene
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpm_config3.npz"))
    n, stride, cp = int(g["n"]), int(g["stride"]), int(g["checkpoints"][0])
    sc = mpm_ball_scene(n, seed=int(g["seed"]))
    o = OracleMPM(n, sc["n_grid"], sc["grid_lim"], "f64")
    o.load_initial_data(sc["x"], sc["vol"], sc["cov"])
    apply_scene(o, sc)
    o.run(sc["dt"], cp)
    for f in ("x", "v", "F_trial", "C"):
        got, want = np.asarray(o.field(f))[::stride], g[f"{f}_{cp}"]
        assert np.array_equal(got, want) or float(np.abs(got - want).max()) <= 1e-13 * max(float(np.abs(want).max()), 1e-300), f
