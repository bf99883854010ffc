"""Pins the MPM oracle to the REFERENCE'S OWN CODE.

tests/golden/mpm_ref_golden.npz was computed by the reference's `MPM_Simulator_WARP` itself -- mpm_solver_warp.py,
mpm_utils.py, warp_utils.py imported unmodified on a numpy interpreter of the Warp API (tests/golden/wp_shim,
tests/golden/make_mpm_ref_golden.py).  Here oracle/mpm_oracle.c (float64 build) runs the same scenes from the same
inputs and must land on the same numbers: every particle field at every checkpoint, the grid after the last substep,
the two per-frame exports.  `wp.svd3` is the one stand-in inside the fixture (see the interpreter's docstring);
`test_svd_convention` shows what depends on it.
"""
import os

import numpy as np
import pytest

from tests._mpm_ref_driver import OracleAdapter, STATE_FIELDS, load_fixture, run

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "mpm_ref_golden.npz")
SCENES = load_fixture(FIXTURE)
TOL = 1e-10     # float64 oracle vs float64 evaluation of the reference's kernels (measured: <= 5e-13)
LONG_FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "mpm_ref_long_golden.npz")
LONG = load_fixture(LONG_FIXTURE)     # two 150-substep rollouts by the same reference code (make_mpm_ref_golden.py --long)
TOL_LONG = 1e-6   # the same comparison after 150 substeps of contact and plastic flow: the yield decisions and the degenerate SVDs of
#                   nearly undeformed particles amplify the last bit -- measured: tree 1.9e-7, sand 8.8e-8, metal 2.4e-7 (grid_v_out), two to
#                   three orders below the distance of the reference's own float32 evaluation from its float64 one (1e-4 .. 1e-3)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def oracle_rollout(scene, arrays, precision="f64"):
    ad = OracleAdapter(scene, arrays, precision)
    snaps = {}
    run(ad, scene, arrays, lambda cp, st: snaps.__setitem__(cp, st))
    return ad, snaps


@pytest.mark.parametrize("name", [n for n in SCENES if not n.endswith("_lapack")])
def test_oracle_equals_reference_code(name):
    scene, arrays, ref = SCENES[name]
    ad, snaps = oracle_rollout(scene, arrays)
    worst = {}
    for cp in scene["checkpoints"]:
        for f in STATE_FIELDS:
            worst[f] = max(worst.get(f, 0.0), rel(snaps[cp][f], ref[f"k{cp}/{f}"]))
    for f in ("grid_m", "grid_v_in", "grid_v_out", "mass"):
        worst[f] = rel(ad.read(f), ref[f])
    assert np.array_equal(ad.read("material"), ref["material"])
    cov, R = ad.exports()
    worst["cov"], worst["R"] = rel(cov, ref["cov_out"]), rel(R, ref["R_out"])
    assert abs(ad.time - float(ref["time"])) < 1e-15
    print(name, {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < TOL, worst
    # the scene must have exercised what it is there for
    mat = ref["material"].astype(int)
    if name in ("metal", "snow", "mixed_materials"):
        k_last = scene["checkpoints"][-1]
        initial = arrays["yield_stress"] if "yield_stress" in arrays else np.float32(scene["params"]["yield_stress"])
        changed = np.abs(ref[f"k{k_last}/yield_stress"] - initial) > 0
        assert changed.any(), "no particle yielded"
    if name == "snow":
        assert (ref[f"k{scene['checkpoints'][-1]}/mu"] == 0).any(), "no particle lost its stiffness"
    if name == "mixed_materials":
        assert set(np.unique(mat)) == {0, 1, 2, 3, 5, 6}


ROLLOUTS = [n for n in sorted(LONG) if not LONG[n][0].get("store_f32")]


def test_knife_edge_collider_plane():
    """A collider plane on a node plane: `float(k) * dx - point < 0` is decided by the last bit (see make_mpm_ref_golden.py,
    long_scenes).  The reference's source evaluated with every float32 operation rounded and evaluated exactly disagree about
    a whole plane of nodes; the float64 oracle must follow the exact evaluation, the float32 oracle (gcc, no contraction) the
    rounded one -- each to its usual accuracy -- and the two fixtures must really be far apart."""
    scene, arrays, ref = LONG["knife_edge_floor"]
    _, s64 = oracle_rollout(scene, arrays, "f64")
    _, s32 = oracle_rollout(scene, arrays, "f32")
    for cp in scene["checkpoints"]:
        for f in ("x", "v", "C", "F", "stress"):
            # (1e-8, not TOL: the column starts from F = I, where the singular values are degenerate and the two SVDs -- LAPACK in
            # the fixture, Jacobi in the oracle -- pick different bases; the reconstruction amplifies 1e-16 by the inverse gap)
            assert rel(s64[cp][f], ref[f"k{cp}/{f}"]) < 1e-8, (cp, f)
        assert rel(s32[cp]["v"], ref[f"k{cp}_f32/v"]) < 1e-5 and rel(s32[cp]["F"], ref[f"k{cp}_f32/F"]) < 1e-5
        assert rel(ref[f"k{cp}_f32/v"], ref[f"k{cp}/v"]) > 0.05


@pytest.mark.parametrize("name", ROLLOUTS)
def test_oracle_tracks_the_reference_code_over_a_rollout(name):
    """150 substeps (checkpoints at 50 / 100 / 150): the tree scenario of custom_tree_config.json on a moving ball, and a sand
    and a metal column hitting a sticky floor (custom_sand_config.json's flags; von Mises with hardening).  The oracle must still be ON the reference's numbers,
    and its float32 build must have drifted from them by about as much as the reference's own float32 run has."""
    scene, arrays, ref = LONG[name]
    ad, snaps = oracle_rollout(scene, arrays)
    worst = {}
    for cp in scene["checkpoints"]:
        for f in STATE_FIELDS:
            worst[f] = max(worst.get(f, 0.0), rel(snaps[cp][f], ref[f"k{cp}/{f}"]))
    for f in ("grid_m", "grid_v_in", "grid_v_out"):
        worst[f] = rel(ad.read(f), ref[f])
    cov, R = ad.exports()
    worst["cov"], worst["R"] = rel(cov, ref["cov_out"]), rel(R, ref["R_out"])
    print(name, {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < TOL_LONG, worst
    assert abs(ad.time - float(ref["time"])) < 1e-12
    last = scene["checkpoints"][-1]
    moved = np.linalg.norm(ref[f"k{last}/x"] - arrays["x0"], axis=1)
    assert np.median(moved) > 5e-4                                   # a rollout, not a perturbation: > 0.5 % of a cell for most particles
    if name == "metal_rollout":
        assert (ref[f"k{last}/yield_stress"] > 2.0e3 * 1.5).sum() > 20  # hardened
    if name == "sand_rollout":                                       # the column reached the floor and flowed plastically
        F = ref[f"k{last}/F"].reshape(len(moved), 9)
        assert (ref[f"k{last}/x"][:, 2] < 0.69).sum() > 20 and np.abs(F - np.eye(3).reshape(9)).max() > 1e-3
    _, s32 = oracle_rollout(scene, arrays, "f32")
    ours = np.array([rel(s32[last][f], ref[f"k{last}/{f}"]) for f in STATE_FIELDS])
    print(name, "float32 oracle vs reference f64:", {f: f"{a:.1e}" for f, a in zip(STATE_FIELDS, ours)},
          "| the reference's own float32 evaluation:", {f: f"{b:.1e}" for f, b in zip(STATE_FIELDS, ref[f"drift/k{last}"])})
    for f, a, b in zip(STATE_FIELDS, ours, ref[f"drift/k{last}"]):
        # The fixture's float32 run takes its SVDs from LAPACK (sgesdd, accurate to an ulp); the oracle's float32 build runs the
        # Jacobi SVD of the restatement in float32, as Warp's svd3 does on the GPU -- noisier, and a plastic rollout feeds that
        # noise back through the yield surface every substep (metal: v 1.4e-4 against 8.8e-6, C 4.4e-4 against 4.1e-5).  Within 16x, or inside 3e-4.
        assert a < 16 * max(b, 1e-7) or a < 3e-4, (name, f, a, b)


def test_float32_oracle_drifts_like_the_reference_code_in_float32():
    """The float32 build of the oracle and the float32 evaluation of the reference's kernels are two single-precision
    runs of the same arithmetic: their distances from the float64 result must be of the same size (the tolerances of
    the GPU tests are scaled by this drift)."""
    for name in ("jelly_apic", "metal", "sand"):
        scene, arrays, ref = SCENES[name]
        _, s32 = oracle_rollout(scene, arrays, "f32")
        cp = scene["checkpoints"][-1]
        ours = np.array([rel(s32[cp][f], ref[f"k{cp}/{f}"]) for f in STATE_FIELDS])
        theirs = ref[f"drift/k{cp}"]
        for f, a, b in zip(STATE_FIELDS, ours, theirs):
            assert a < 8 * max(b, 1e-7), (name, f, a, b)


def test_svd_convention():
    """det F < 0: under LAPACK's convention (s >= 0, improper U or V) the reference's code gives DIFFERENT stresses for the
    inverted particles and the same for all others; the oracle follows Warp's convention (proper rotations)."""
    scene, arrays, warp = SCENES["inverted"]
    _, _, lapack = SCENES["inverted_lapack"]
    det = np.linalg.det(arrays["Ft0"].astype(np.float64).reshape(-1, 3, 3))
    inv = det < 0
    assert 60 < inv.sum() < 120
    d = np.abs(warp["k1/stress"] - lapack["k1/stress"]).reshape(len(det), -1).max(1)
    scale = np.abs(warp["k1/stress"]).max()
    assert (d[~inv] < 1e-9 * scale).all()
    assert (d[inv] > 1e-6 * scale).sum() > 0.5 * inv.sum()
    ad, snaps = oracle_rollout(scene, arrays)
    assert rel(snaps[1]["stress"], warp["k1/stress"]) < TOL
    assert rel(snaps[1]["stress"], lapack["k1/stress"]) > 1e-3


def test_launch_order_of_one_substep():
    """p2g2p's kernel sequence as the reference executed it (mpm_solver_warp.py:514-637)."""
    import json
    meta = json.loads(str(np.load(FIXTURE)["meta"]))
    order = meta["jelly_apic"]["substep_launch_order"]
    assert order == ["zero_grid", "apply_force", "compute_stress_from_F_trial", "p2g_apic_with_stress",
                     "grid_normalization_and_gravity", "add_damping_via_grid", "collide", "g2p"]
    order = meta["jelly_pic"]["substep_launch_order"]
    assert order == ["zero_grid", "compute_stress_from_F_trial", "p2g_apic_with_stress", "grid_normalization_and_gravity",
                     "collide", "collide", "collide", "collide", "g2p"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/third_party/PhysGaussian/mpm_solver_warp"), reason="needs the reference tree")
def test_fixture_is_what_the_reference_code_computes_today():
    """Build container only: imports the reference's solver on the interpreter and re-computes one scene live, so the
    committed fixture cannot drift away from the generator or from the reference tree."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.argv=['x']; sys.path.insert(0, 'tests/golden'); import make_mpm_ref_golden as g;"
        "sc, arr = g.scenes()['jelly_rpic']; snaps, extra, order = g.rollout(sc, arr, 'f64');"
        "z = np.load('tests/golden/mpm_ref_golden.npz');"
        "err = max(float(np.abs(snaps[cp][f] - z[f'jelly_rpic/k{cp}/{f}']).max()) for cp in sc['checkpoints'] for f in g.STATE_FIELDS);"
        "print('LIVE_MAX_ABS_DIFF', err)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("LIVE_MAX_ABS_DIFF")][0]
    assert float(line.split()[1]) == 0.0, line
