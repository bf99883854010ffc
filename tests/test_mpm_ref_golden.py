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
