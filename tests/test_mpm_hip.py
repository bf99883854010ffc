"""GPU parity tests for the MPM half: libpixie_hip.so (through pixie_amd.mpm_solver, i.e. the C ABI)
against oracle/mpm_oracle.c on identical seeded scenes.  fp32 arithmetic; tolerances stated per test.

Positions and deformation gradients are O(1) quantities: rel-L2 <= 1e-4 is required outright.
Velocities / APIC matrices in the near-static scenes are O(1e-3) signals riding on fp32 roundoff of
O(1) state, so their error is judged against the float64 oracle relative to how far the float32
oracle itself drifts from it (the reference's own Warp run would drift the same way).
"""
import numpy as np
import pytest
import torch

from oracle.mpm_oracle import OracleMPM
from pixie_amd.synthetic import apply_scene, mpm_ball_scene

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# The two accumulation modes of the P2G scatter (csrc/mpm.hip): 64 = exact 64-bit fixed point, 32 = packed pairs of 32-bit
# sums (half the LDS atomics).  Every rollout test holds BOTH to the same particle-level bars.
SCATTER_MODES = (64, 32)
# Where a quantity cannot meet 1e-4 against the float64 oracle for a reason that lies in float32 itself (x += dt v below ulp(x);
# a yield surface crossed on the other side), the yardstick is the distance of the float32 ORACLE -- the reference's algorithm
# in the reference's precision -- from the float64 one, and the bar is DRIFT_K times it.  Measured over every test of this file
# the product sits between 0.4 and 1.0 of that yardstick (profiles/r5a_pytest_mpm_numbers.txt; 4.0 was the bar until round 4).
DRIFT_K = 1.5


def make_hip(scene, per_particle=True, bits=None, diag=False):
    """diag=True: the handle lives in libpixie_hip_diag.so (same sources, -DPIXIE_DIAG), which adds the per-phase entry point."""
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    s = MPM_Simulator_WARP(10, diag=diag)
    s.load_initial_data_from_torch(torch.from_numpy(scene["x"]), torch.from_numpy(scene["vol"]), torch.from_numpy(scene["cov"]),
                                   n_grid=scene["n_grid"], grid_lim=scene["grid_lim"])
    apply_scene(s, scene, per_particle=per_particle)
    if bits is not None:
        s._set_scalar("scatter_bits", bits)
    return s


def make_oracle(scene, precision="f32", per_particle=True):
    o = OracleMPM(scene["x"].shape[0], scene["n_grid"], scene["grid_lim"], precision)
    o.load_initial_data(scene["x"], scene["vol"], scene["cov"])
    apply_scene(o, scene, per_particle=per_particle)
    return o


def get(s, name):
    return s.get_field(name).cpu().numpy()


def test_library_is_native(hip_device):
    from pixie_amd import _lib
    assert _lib.load().pixie_build_arch() == b"gfx950"


def test_phase_by_phase_parity(hip_device):
    """One substep, kernel by kernel, from an identical non-trivial state (random v, C, F_trial)."""
    sc = mpm_ball_scene(20000, seed=1)
    n = 20000
    rng = np.random.default_rng(0)
    v0 = (0.5 * rng.normal(size=(n, 3))).astype(np.float32)
    C0 = (2.0 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    Ft0 = (np.eye(3) + 0.03 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    h, o = make_hip(sc, bits=64, diag=True), make_oracle(sc, "f32")   # the exact accumulation mode; the packed one: test_packed_scatter_parity
    h.set_field("v", v0); h.set_field("C", C0.reshape(n, 9)); h.set_field("F_trial", Ft0.reshape(n, 9))
    o.field("v")[:] = v0; o.field("C")[:] = C0; o.field("F_trial")[:] = Ft0
    dt = sc["dt"]
    # phase 0: modifiers + stress + P2G
    o.phase("zero_grid"); o.phase("pre_p2g", dt); o.phase("compute_stress", dt); o.phase("p2g", dt)
    h.phase(0, dt)
    assert rel_l2(get(h, "F").reshape(n, 3, 3), o.field("F")) < 1e-6
    assert rel_l2(get(h, "stress").reshape(n, 3, 3), o.field("stress")) < 5e-4  # 2mu(F-R): cancellation-limited
    assert rel_l2(get(h, "v"), o.field("v")) < 1e-6  # impulse applied
    assert rel_l2(get(h, "grid_m"), o.field("grid_m")) < 1e-5
    assert rel_l2(get(h, "grid_v_in"), o.field("grid_v_in")) < 1e-4
    # phase 1: grid update + damping + BCs
    o.phase("grid_update", dt); o.phase("grid_damping"); o.phase("apply_bcs", dt)
    h.phase(1, dt)
    gv_h, gv_o = get(h, "grid_v_out"), o.field("grid_v_out")
    # Node sums are accumulated in 64-bit fixed point with an LSB of 2^-50 of the largest contribution in a tile
    # (csrc/mpm.hip).  A stencil-corner node at the free surface can have a mass 1e-8 ... 1e-13 of a particle's while
    # the force term (proportional to grad w ~ sqrt(w)) gives it a velocity of 1e2 ... 1e4 m/s: such nodes dominate a
    # plain rel-L2 over grid_v_out although they feed G2P with weights of the same 1e-8 ... 1e-13.  Parity is therefore
    # required at 1e-4 on the nodes that carry mass (> 1e-6 of one particle's), at 1 % on the vanishing-mass nodes
    # (whose relative rounding error is the LSB over their mass), and the support must agree exactly.
    m_o = o.field("grid_m")
    heavy = m_o > 1e-6 * float(o.field("mass").max())
    print(f"grid nodes with mass: {(m_o > 1e-15).sum()}, of which carrying > 1e-6 particle masses: {heavy.sum()}; "
          f"rel-L2 heavy {rel_l2(gv_h[heavy], gv_o[heavy]):.2e}, all {rel_l2(gv_h, gv_o):.2e}")
    assert heavy.sum() > 0.8 * (m_o > 1e-15).sum()
    assert rel_l2(gv_h[heavy], gv_o[heavy]) < 1e-4
    assert rel_l2(gv_h, gv_o) < 1e-2
    assert np.array_equal(gv_h == 0, gv_o == 0)  # same support: BC slab and empty cells
    assert float(np.abs(get(h, "grid_m")).max()) == 0.0  # grid kernel cleared (m*v, m) behind itself
    # phase 2: G2P
    o.phase("g2p", dt)
    h.phase(2, dt)
    assert rel_l2(get(h, "x"), o.field("x")) < 1e-6
    assert rel_l2(get(h, "v"), o.field("v")) < 1e-4
    assert rel_l2(get(h, "C").reshape(n, 3, 3), o.field("C")) < 1e-4
    assert rel_l2(get(h, "F_trial").reshape(n, 3, 3), o.field("F_trial")) < 1e-6
    assert h.out_of_bounds == 0


def _assert_rollout_parity(h, o32, o64, sc, tag):
    """Positions are judged by the DISPLACEMENT x - x0 (the signal): within max(1e-4, 4 x the float32 oracle's own distance
    from the float64 oracle), like v and C.  F_trial within 1e-4 of the float64 oracle outright.  (The rel-L2 of the O(1)
    coordinate itself is kept only as a sanity line: in the quiet scenes 1e-4 there is ~100x the displacement signal.)"""
    assert abs(h.time - o64.time) < 1e-12
    x_h, F_h, v_h, C_h = get(h, "x"), get(h, "F_trial").reshape(-1, 3, 3), get(h, "v"), get(h, "C").reshape(-1, 3, 3)
    assert np.isfinite(x_h).all()
    assert rel_l2(x_h, o64.field("x")) < 1e-4
    assert rel_l2(F_h, o64.field("F_trial")) < 1e-4
    # displacement (the signal itself, not the O(1) coordinate)
    disp_h, disp_o = x_h - sc["x"], o64.field("x") - sc["x"]
    drift_d = rel_l2(o32.field("x") - sc["x"], disp_o)
    assert rel_l2(disp_h, disp_o) < max(1e-4, DRIFT_K * drift_d)
    # v and C are compared on the scale of the velocity field: C is a velocity gradient, so its natural unit is
    # rms|v| / dx (in free fall C is ~0 and a norm-relative error would compare roundoff with roundoff).
    n = x_h.shape[0]
    v_rms = float(np.linalg.norm(o64.field("v")) / np.sqrt(n))
    inv_dx = sc["n_grid"] / sc["grid_lim"]
    for name, got, scale in (("v", v_h, v_rms), ("C", C_h, max(v_rms * inv_dx, float(np.linalg.norm(o64.field("C")) / np.sqrt(n))))):
        drift = float(np.linalg.norm(o32.field(name).astype(np.float64) - o64.field(name)) / np.sqrt(n)) / scale
        err = float(np.linalg.norm(got.astype(np.float64) - o64.field(name)) / np.sqrt(n)) / scale
        print(f"{tag} {name}: hip-vs-f64 {err:.3e}  oracle f32-vs-f64 {drift:.3e}  (scale {scale:.3e})")
        assert err < max(1e-4, DRIFT_K * drift)
    assert h.out_of_bounds == 0


@pytest.mark.parametrize("scenario,steps", [("tree", 200), ("ball", 200)])
def test_rollout_parity(hip_device, scenario, steps):
    sc = mpm_ball_scene(20000, seed=2, scenario=scenario)
    o32, o64 = make_oracle(sc, "f32"), make_oracle(sc, "f64")
    o32.run(sc["dt"], steps); o64.run(sc["dt"], steps)
    for bits in SCATTER_MODES:
        h = make_hip(sc, bits=bits)
        h.run(sc["dt"], steps)
        _assert_rollout_parity(h, o32, o64, sc, f"{scenario}/{bits}-bit")


def test_rollout_parity_at_the_1m_bench_size(hip_device):
    """The scene bench.py's `mpm_1m` leg times -- BASELINE configs[4]'s per-GPU MPM workload: 1 000 000 particles, n_grid 120,
    tree scenario -- for 20 substeps against the scalar C oracle in float32 and float64 (2 x ~25 s of one host core): the
    multi-item blocks, the 5555-item work list and the 120^3 block tables of the full-size run, not a scaled-down stand-in."""
    sc = mpm_ball_scene(1_000_000, seed=0, n_grid=120)
    o32, o64 = make_oracle(sc, "f32"), make_oracle(sc, "f64")
    o32.run(sc["dt"], 20); o64.run(sc["dt"], 20)
    for bits in SCATTER_MODES:
        h = make_hip(sc, bits=bits)
        h.run(sc["dt"], 20)
        _assert_rollout_parity(h, o32, o64, sc, f"1M/{bits}-bit")
        assert int(h._get_scalar("n_work_items")) > 4000
        del h


def test_fast_particles_drift_controller_and_slow_path(hip_device):
    """Particles crossing a cell every ~13 substeps (30 m/s, dx = 0.04, dt = 1e-4).  (a) Automatic cadence: the measured drift per interval sets the
    next re-binning interval, no particle leaves its workgroup's LDS tile.  (b) Cadence pinned far too long: stencils
    leave the tile, those particles go straight to HBM (slow path) -- results still match the oracle."""
    sc = mpm_ball_scene(20000, seed=12, scenario="ball")
    sc["params"] = dict(material="jelly", g=[0.0, 0.0, 0.0], E=2e5, nu=0.3, density=500.0)
    sc["bcs"] = []
    v0 = np.tile(np.array([[30.0, -12.0, 7.0]], np.float32), (20000, 1))
    o = make_oracle(sc, "f64")
    o.field("v")[:] = v0
    o.run(sc["dt"], 60)
    for pinned in (None, 20):   # 20 substeps = 1.5 cells of drift: past the one-cell margin of the tile, inside the active blocks
        h = make_hip(sc)
        h.set_field("v", v0)
        if pinned:
            h._set_scalar("resort_interval", pinned)
        h.run(sc["dt"], 60)
        slow, rebins = h._get_scalar("slow_path_particles"), h._get_scalar("n_rebins")
        print(f"resort_interval {pinned or 'auto'}: slow-path particle-substeps {slow:.0f}, rebins {rebins:.0f}, "
              f"dropped {h._get_scalar('dropped_particles'):.0f}, final interval {h._get_scalar('resort_interval'):.0f}")
        if pinned:
            assert slow > 0
        else:
            assert slow == 0 and rebins >= 6
        assert h._get_scalar("dropped_particles") == 0 and h.out_of_bounds == 0
        assert rel_l2(get(h, "x"), o.field("x")) < 1e-5
        assert rel_l2(get(h, "v"), o.field("v")) < 1e-4
        assert rel_l2(get(h, "F_trial").reshape(-1, 3, 3), o.field("F_trial")) < 1e-5


def test_single_step_api_equals_batched(hip_device):
    """One substep per pixie_mpm_step call (p2g2p + flush: P2G launch, G2P launch) == run(dt, n) (fused launches) up to
    the instruction schedule; the reference's own loop -- p2g2p() n times, then an export -- is DEFERRED by the shim and
    must be bit-identical to run(dt, n)."""
    sc = mpm_ball_scene(8000, seed=4, scenario="ball")
    a, b, d = make_hip(sc), make_hip(sc), make_hip(sc)
    for i in range(20):
        a.p2g2p(i, sc["dt"])
        a.flush()
    b.run(sc["dt"], 20)
    for i in range(20):               # gs_simulation.py:633-634, unmodified
        d.p2g2p(i, sc["dt"])
    assert d._pending == 20           # nothing has been enqueued yet
    assert abs(d.time - 20 * sc["dt"]) < 1e-12 and d._pending == 0   # observing the solver flushes the queue
    # Both are deterministic (fixed-point tile sums, fixed-order grid gather), but not the same arithmetic: the fused
    # kernel keeps v, C and F_trial in registers between G2P and the next P2G, the single-step API stores and reloads
    # them through the caller-visible arrays in between (bit-identical values) and runs the stress on its own launch
    # with a different instruction schedule (FMA contraction across the fused boundary differs).  Each path IS
    # reproducible: two runs of the same path must agree bit for bit.
    assert rel_l2(get(a, "x"), get(b, "x")) < 1e-6
    assert rel_l2(get(a, "v"), get(b, "v")) < 1e-4
    c = make_hip(sc)
    c.run(sc["dt"], 20)
    for f in ("x", "v", "C", "F_trial"):
        assert np.array_equal(get(b, f), get(c, f)), f"run() is not bit-reproducible in {f}"
        assert np.array_equal(get(b, f), get(d, f)), f"the deferred p2g2p loop differs from run() in {f}"
    # a change of dt inside the loop flushes what was queued with the old one
    e, f2 = make_hip(sc), make_hip(sc)
    for i in range(5):
        e.p2g2p(i, sc["dt"])
    for i in range(5):
        e.p2g2p(i, 0.5 * sc["dt"])
    f2.run(sc["dt"], 5); f2.run(0.5 * sc["dt"], 5)
    assert np.array_equal(get(e, "x"), get(f2, "x")) and abs(e.time - f2.time) < 1e-15


@pytest.mark.parametrize("bits", SCATTER_MODES)
def test_rollout_bit_reproducible_with_crowded_blocks(hip_device, bits):
    """Run-to-run bit-reproducibility at the density of BASELINE configs[3] (100 k particles in a 50^3 grid: ~800 particles
    per 4^3 block, i.e. every block is split into several 256-particle work items, each with its own fixed-point scale).
    The split must not depend on the order in which the re-binning's atomics happened to arrive: slots inside a block are
    ranked by (cell, previous slot).  Three re-binnings are forced inside the run."""
    sc = mpm_ball_scene(100000, seed=12)
    res = []
    for rep in range(3):
        h = make_hip(sc, bits=bits)
        h._set_scalar("resort_interval", 40)
        h.run(sc["dt"], 150)
        assert int(h._get_scalar("slow_path_particles")) == 0 and h.out_of_bounds == 0
        assert int(h._get_scalar("n_work_items")) > 400
        res.append({f: get(h, f) for f in ("x", "v", "C", "F_trial")})
    for rep in (1, 2):
        for f in res[0]:
            assert np.array_equal(res[rep][f], res[0][f]), (rep, f)


@pytest.mark.parametrize("bits", SCATTER_MODES)
def test_sparse_tile_publishing_bit_identical(hip_device, bits):
    """set_scalar "sparse_tiles": P2G stores only the non-zero nodes of a work item's 8^3 tile plus a 512-bit occupancy mask, the
    grid kernel reads only those (on automatically for scenes that fill the chip, together with the grid kernel's
    high-occupancy instantiation, "grid_rb" 1).  Skipping all-zero addends cannot change a
    sum: particle fields, the grid read back while a P2G is pending, and the updated grid must all be bit-identical."""
    sc = mpm_ball_scene(12000, seed=21)
    res = {}
    variants = [(0, 4), (1, 4), (1, 1), (0, 1), (1, 2)]   # (sparse tiles, grid kernel's loads in flight per candidate block)
    for sparse, rb in variants:
        h = make_hip(sc, bits=bits, diag=True)
        h._set_scalar("sparse_tiles", sparse)
        h._set_scalar("grid_rb", rb)
        h.run(sc["dt"], 40)
        out = {f: get(h, f) for f in ("x", "v", "C", "F_trial")}
        h.phase(0, sc["dt"])                                   # a P2G whose tiles are still staged
        out["grid_m"] = h.get_field("grid_m").cpu().numpy()
        out["grid_v_in"] = h.get_field("grid_v_in").cpu().numpy()
        h.phase(1, sc["dt"])
        out["grid_v_out"] = h.get_field("grid_v_out").cpu().numpy()
        assert h.out_of_bounds == 0
        res[sparse, rb] = out
    assert float(np.abs(res[0, 4]["grid_m"]).sum()) > 0
    for key in variants[1:]:
        for f in res[0, 4]:
            assert np.array_equal(res[key][f], res[0, 4][f]), (key, f)


def test_latency_optimised_variant_matches(hip_device):
    """Scenes too small to fill the chip (<= 2 work items per CU: the whole work list is resident at once and a launch lasts
    one work item's latency) run the kernel variant built without scheduling barriers and with the register budget of two
    waves per SIMD (set_scalar "wide": -1 auto, 0 / 1 forced).  Same arithmetic in another instruction order."""
    sc = mpm_ball_scene(20000, seed=6, scenario="ball")
    res = {}
    for wide in (0, 1):
        h = make_hip(sc)
        h._set_scalar("wide", wide)
        h.run(sc["dt"], 150)
        res[wide] = {f: get(h, f) for f in ("x", "v", "F_trial")}
        assert h.out_of_bounds == 0
    for f in ("x", "F_trial"):
        assert rel_l2(res[1][f], res[0][f]) < 1e-6
    assert rel_l2(res[1]["v"], res[0]["v"]) < 1e-4


def test_additional_material_params_batch_equals_sequential_launches(hip_device):
    """material_field.py:343-363 uploads the field as one 1 mm box per particle through set_parameters_dict(
    {"additional_material_params": [...]}): N sequential launches in the reference (a particle within 1 mm of a later one
    takes the LATER one's values).  The one-launch path must leave exactly what the sequential launches leave."""
    from pixie_amd import _lib
    from pixie_amd._lib import check, d3
    sc = mpm_ball_scene(3000, seed=21)
    rng = np.random.default_rng(2)
    x = sc["x"].copy()
    x[1500:] = x[:1500] + rng.uniform(-6e-4, 6e-4, size=(1500, 3)).astype(np.float32)     # pairs closer than 1 mm: boxes overlap
    sc["x"] = x
    plist = [dict(point=x[i].tolist(), size=[0.001, 0.001, 0.001], density=float(200 + i), E=float(1e5 + 7 * i), nu=float(0.2 + 1e-5 * i),
                  material=int(i % 7)) for i in range(3000)]
    plist.append(dict(point=[1.0, 1.0, 1.2], size=[0.2, 0.3, 0.1], density=50.0, E=3e4, nu=0.11, material="snow"))     # a big late box
    a, b = make_hip(sc), make_hip(sc)
    a.set_parameters_dict({"additional_material_params": [dict(p) for p in plist]})
    lib = _lib.load()
    from pixie_amd.mpm_solver import get_material_name
    for p in plist:
        mat = get_material_name(p["material"]) if isinstance(p["material"], str) else p["material"]
        check(lib.pixie_mpm_apply_additional_params(b._h, d3(p["point"]), d3(p["size"]), float(p["E"]), float(p["nu"]), float(p["density"]), int(mat),
                                                    b._stream), "apply_additional_params")
    b._update_mass()
    for f in ("E", "nu", "density", "material", "mass"):
        assert np.array_equal(get(a, f), get(b, f)), f
    later = get(a, "density")[:1500] != (200 + np.arange(1500))
    assert later.sum() > 100                      # many particles did take a later box's values


def test_deferred_substeps_keep_the_stream_they_were_queued_on(hip_device):
    """p2g2p() only queues; the batch must be enqueued on the stream that was current when it was queued, and a
    change of the current stream ends the batch."""
    sc = mpm_ball_scene(6000, seed=9)
    a, b = make_hip(sc), make_hip(sc)
    side = torch.cuda.Stream(hip_device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(5):
            a.p2g2p(i, sc["dt"])
        assert a._pending == 5 and a._pending_stream == side.cuda_stream
    a.p2g2p(5, sc["dt"])                        # queued under another stream: the five above were flushed onto `side` first
    assert a._pending == 1 and a._pending_stream == (torch.cuda.current_stream().cuda_stream or 0)
    torch.cuda.current_stream().wait_stream(side)
    b.run(sc["dt"], 6)
    assert np.array_equal(get(a, "x"), get(b, "x")) and np.array_equal(get(a, "F_trial"), get(b, "F_trial"))


def test_run_batch_equals_running_the_scenes_one_after_the_other(hip_device):
    """pixie_amd.mpm_solver.run_batch: several independent scenes on their own HIP streams and host threads (the batch
    configuration of BASELINE configs[3]) -- every scene must end in exactly the state it reaches alone, and the caller's stream
    must be ordered after all of them (the export below runs on the current stream with no synchronisation in between)."""
    from pixie_amd.mpm_solver import run_batch
    scs = [mpm_ball_scene(30_000, seed=40 + i, n_grid=40, scenario=("tree", "ball", "tree")[i]) for i in range(3)]
    batch, alone = [make_hip(sc) for sc in scs], [make_hip(sc) for sc in scs]
    for k in range(3):                       # three calls: the streams are kept and re-used, queued substeps are flushed first
        batch[1].p2g2p(0, scs[1]["dt"])      # a deferred substep on one of them
        alone[1].p2g2p(0, scs[1]["dt"])
        run_batch(batch, scs[0]["dt"], 40)
        xs = [b.export_particle_x_to_torch().clone() for b in batch]
        for a in alone:
            a.run(scs[0]["dt"], 40)
        for b, a, x in zip(batch, alone, xs):
            assert torch.equal(x, a.export_particle_x_to_torch())
            for f in ("v", "C", "F_trial"):
                assert np.array_equal(get(b, f), get(a, f)), f
            assert abs(b.time - a.time) < 1e-12
    run_batch([], 1e-4, 5)


def test_deferred_substeps_order_the_observing_stream_after_them(hip_device):
    """the caller synchronises as it would for the reference's eager kernels -- wait_stream(side) right after the
    p2g2p() loop, BEFORE anything flushes the queue -- so that wait sees an empty stream.  The flush (here: the next batch on the
    default stream, then an export) must itself order the current stream after the substeps it puts on `side`; a slow kernel is
    put in front of them on `side` so that a missing dependency shows as a race."""
    sc = mpm_ball_scene(200_000, seed=9, n_grid=64)
    a, b = make_hip(sc), make_hip(sc)
    side = torch.cuda.Stream(hip_device)
    side.wait_stream(torch.cuda.current_stream())
    ballast = torch.randn((4096, 4096), device=hip_device)
    with torch.cuda.stream(side):
        for _ in range(6):
            ballast = ballast @ ballast * 1e-3         # keeps `side` busy when the substeps arrive on it
        for i in range(5):
            a.p2g2p(i, sc["dt"])
    torch.cuda.current_stream().wait_stream(side)       # too early to see the substeps: they are still queued in the shim
    for i in range(5, 9):
        a.p2g2p(i, sc["dt"])                            # first call flushes the five onto `side`; these four run on the current stream
    xa = a.export_particle_x_to_torch().clone()
    b.run(sc["dt"], 9)
    assert torch.equal(xa, b.export_particle_x_to_torch())
    assert np.array_equal(get(a, "F_trial"), get(b, "F_trial"))


def test_mass_contrast_selects_the_exact_scatter(hip_device):
    """the packed scatter's quantum is 2^-30 of the SUM of a work item's bounds, so nodes fed only by particles much
    lighter than their tile-mates are quantised at visible weights.  A scene whose upper half is 1e4 times lighter than its
    lower half (both halves meet inside blocks): the default mode must come out as the exact one, its light-side grid
    velocities must meet the usual 1e-4 bar, and forcing the packed mode must be measurably worse there (the limit the
    `scatter_bits` property documents).  A uniform scene keeps the packed mode."""
    sc = mpm_ball_scene(20000, seed=4)
    n = 20000
    light = sc["x"][:, 2] > 1.0
    sc["density"] = np.where(light, 0.2, 2000.0).astype(np.float32)
    rng = np.random.default_rng(1)
    v0 = (0.5 * rng.normal(size=(n, 3))).astype(np.float32)
    Ft0 = (np.eye(3) + 0.02 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    dt = sc["dt"]
    o = make_oracle(sc, "f64")
    o.field("v")[:] = v0; o.field("F_trial")[:] = Ft0
    o.phase("zero_grid"); o.phase("pre_p2g", dt); o.phase("compute_stress", dt); o.phase("p2g", dt)
    o.phase("grid_update", dt); o.phase("grid_damping"); o.phase("apply_bcs", dt)
    gv_o, m_o = o.field("grid_v_out").astype(np.float64), o.field("grid_m").astype(np.float64)
    m_light = float(o.field("mass")[light].max())
    # nodes that carry mass on the scale of LIGHT particles only (the heavy half does not reach them)
    sel = (m_o > 1e-2 * m_light) & (m_o < 20 * m_light)
    assert sel.sum() > 500
    errs = {}
    for mode in (0, 32, 64):
        h = make_hip(sc, diag=True)
        h._set_scalar("scatter_bits", mode)
        h.set_field("v", v0); h.set_field("F_trial", Ft0.reshape(n, 9))
        h.phase(0, dt); h.phase(1, dt)
        errs[mode] = (rel_l2(get(h, "grid_v_out").astype(np.float64)[sel], gv_o[sel]), h.scatter_bits, h._get_scalar("mass_contrast"))
    print("light-side grid_v_out vs the float64 oracle:", {k: f"{v[0]:.2e} (mode {v[1]}, contrast {v[2]:.3g})" for k, v in errs.items()})
    assert errs[0][1] == 64 and errs[0][2] > 5e3                  # auto picked the exact mode
    assert errs[0][0] < 1e-4 and errs[64][0] < 1e-4
    assert errs[32][0] > 3 * errs[64][0]                          # the packed mode's documented limit
    uniform = make_hip(mpm_ball_scene(8000, seed=5))
    uniform.run(dt, 2)
    assert uniform.scatter_bits == 32 and uniform._get_scalar("mass_contrast") < 32


def test_packed_scatter_parity(hip_device):
    """set_scalar "scatter_bits" 32: two 32-bit fixed-point sums per LDS atomic (2 atomics per node instead of 4).  The sums
    stay exact integers (bit-reproducible), the quantum grows from 2^-42 to 2^-22 of the largest contribution bound in a
    tile.  Required: the same particle-level parity as the exact mode (x, F 1e-4 outright; v, C within the float32
    oracle's own drift), momentum-weighted grid parity, and the documented loss: nodes whose whole mass is below the
    quantum (stencil corners at a free surface) are dropped -- their share of the grid's momentum is reported."""
    sc = mpm_ball_scene(20000, seed=1)
    n = 20000
    rng = np.random.default_rng(0)
    v0 = (0.5 * rng.normal(size=(n, 3))).astype(np.float32)
    C0 = (2.0 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    Ft0 = (np.eye(3) + 0.03 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    h, o = make_hip(sc, diag=True), make_oracle(sc, "f32")
    h._set_scalar("scatter_bits", 32)
    h.set_field("v", v0); h.set_field("C", C0.reshape(n, 9)); h.set_field("F_trial", Ft0.reshape(n, 9))
    o.field("v")[:] = v0; o.field("C")[:] = C0; o.field("F_trial")[:] = Ft0
    dt = sc["dt"]
    o.phase("zero_grid"); o.phase("pre_p2g", dt); o.phase("compute_stress", dt); o.phase("p2g", dt)
    h.phase(0, dt)
    m_h, m_o = get(h, "grid_m").astype(np.float64), o.field("grid_m").astype(np.float64)
    p_h, p_o = get(h, "grid_v_in").astype(np.float64), o.field("grid_v_in").astype(np.float64)
    e_m, e_p = rel_l2(m_h, m_o), rel_l2(p_h, p_o)
    e_msum = abs(m_h.sum() - m_o.sum()) / m_o.sum()
    e_psum = np.abs(p_h.sum(axis=(0, 1, 2)) - p_o.sum(axis=(0, 1, 2))).max() / np.abs(p_o).sum()
    o.phase("grid_update", dt); o.phase("grid_damping"); o.phase("apply_bcs", dt)
    h.phase(1, dt)
    gv_h, gv_o = get(h, "grid_v_out").astype(np.float64), o.field("grid_v_out").astype(np.float64)
    m_p = float(o.field("mass").max())
    # momentum-weighted: what G2P hands back to the particles is sum_i w_ip v_i, and w_ip ~ m_i / m_p
    werr = np.linalg.norm((m_o[..., None] * (gv_h - gv_o))) / np.linalg.norm(m_o[..., None] * gv_o)
    lost = (m_o > 1e-15) & (np.abs(gv_h).sum(-1) == 0) & (np.abs(gv_o).sum(-1) > 0)
    by_mass = {thr: rel_l2(gv_h[m_o > thr * m_p], gv_o[m_o > thr * m_p]) for thr in (1.0, 0.1, 1e-2, 1e-3, 1e-4)}
    print(f"packed scatter, one substep from a rough state (|v| ~ 0.9, |C| ~ 6): grid_m {e_m:.2e}, grid_v_in {e_p:.2e}, total mass {e_msum:.1e}, total momentum {e_psum:.1e}; "
          f"grid_v_out rel-L2 over nodes heavier than k particle masses: " + ", ".join(f"k={k:g}: {v:.2e}" for k, v in by_mass.items())
          + f"; momentum-weighted over all nodes {werr:.2e}; nodes quantised to zero mass {lost.sum()} of {(m_o > 1e-15).sum()}, carrying "
          f"{m_o[lost].sum() / m_o.sum():.2e} of the mass")
    assert e_m < 1e-5 and e_p < 1e-4
    assert e_msum < 1e-6 and e_psum < 1e-6            # mass and momentum conserved through the rounding
    assert by_mass[0.1] < 1e-4                        # nodes inside the material: the exact mode's bar
    assert by_mass[1e-3] < 1e-3                       # light nodes: quantum / mass grows as the mass shrinks
    assert werr < 1e-5                                # what the particles get back
    assert m_o[lost].sum() / m_o.sum() < 1e-6         # the documented loss: corner nodes below the quantum
    o.phase("g2p", dt)
    h.phase(2, dt)
    assert rel_l2(get(h, "x"), o.field("x")) < 1e-6
    assert rel_l2(get(h, "v"), o.field("v")) < 1e-4
    assert rel_l2(get(h, "C").reshape(n, 3, 3), o.field("C")) < 1e-4
    assert rel_l2(get(h, "F_trial").reshape(n, 3, 3), o.field("F_trial")) < 1e-6
    # integer sums: bit-reproducible like the exact mode (the rollouts against the oracle are in test_rollout_parity & co.)
    sc2 = mpm_ball_scene(20000, seed=2, scenario="tree")
    runs = []
    for _ in range(2):
        h2 = make_hip(sc2, bits=32)
        h2.run(sc2["dt"], 100)
        runs.append((get(h2, "x"), get(h2, "v")))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_inverted_particles_take_the_svd_route(hip_device):
    """Jelly particles with det F <= 0 (inverted elements): the Newton polar iteration of the fast path does not give
    U V^T of the proper-rotation SVD there (mpm_math.h: polar_rotation returns false) and the kernel must fall back to
    svd3, as kirchoff_stress_FCR (mpm_utils.py:10-17) uses wp.svd3 for every particle.  A tenth of the particles start
    reflected / collapsed along one axis; compared with the float64 oracle after a few substeps (the inverted elements
    push back violently, so the horizon is short), plus the stress of the very first substep."""
    sc = mpm_ball_scene(20000, seed=21, scenario="ball")
    sc["params"] = dict(material="jelly", g=[0.0, 0.0, 0.0], E=2e4, nu=0.3, density=1000.0)
    n = 20000
    rng = np.random.default_rng(5)
    Ft0 = (np.eye(3) + 0.02 * rng.normal(size=(n, 3, 3)))
    bad = rng.random(n) < 0.1
    flip = np.diag([1.0, 1.0, -0.6])
    Ft0[bad] = Ft0[bad] @ flip
    Ft0[bad & (rng.random(n) < 0.3), :, 1] *= 1e-3          # some nearly rank-deficient on top
    Ft0 = Ft0.astype(np.float32)
    assert (np.linalg.det(Ft0.astype(np.float64)) < 0).sum() > 1500
    # stress of the first substep: the branch under test, before any dynamics amplify differences
    h0, o0 = make_hip(sc, per_particle=False, diag=True), make_oracle(sc, "f64", per_particle=False)
    h0.set_field("F_trial", Ft0.reshape(n, 9)); o0.field("F_trial")[:] = Ft0
    h0.phase(0, sc["dt"])
    o0.phase("zero_grid"); o0.phase("pre_p2g", sc["dt"]); o0.phase("compute_stress", sc["dt"])
    tau_h, tau_o = get(h0, "stress").reshape(n, 3, 3), o0.field("stress")
    assert np.isfinite(tau_h).all()
    assert rel_l2(tau_h[bad], tau_o[bad]) < 1e-4
    assert rel_l2(tau_h[~bad], tau_o[~bad]) < 5e-4          # 2 mu (F - R): cancellation-limited (as in the phase test)
    h0.phase(1, sc["dt"]); h0.phase(2, sc["dt"])            # (finish the substep: a pending phase-API P2G blocks the handle)
    h, o32, o64 = make_hip(sc, per_particle=False), make_oracle(sc, "f32", per_particle=False), make_oracle(sc, "f64", per_particle=False)
    h.set_field("F_trial", Ft0.reshape(n, 9))
    o32.field("F_trial")[:] = Ft0; o64.field("F_trial")[:] = Ft0
    h.run(sc["dt"], 10); o32.run(sc["dt"], 10); o64.run(sc["dt"], 10)
    x_h, F_h = get(h, "x"), get(h, "F_trial").reshape(n, 3, 3)
    assert np.isfinite(x_h).all() and np.isfinite(F_h).all()
    d_x = rel_l2(o32.field("x"), o64.field("x")); d_F = rel_l2(o32.field("F_trial"), o64.field("F_trial"))
    e_x = rel_l2(x_h, o64.field("x")); e_F = rel_l2(F_h, o64.field("F_trial"))
    print(f"inverted particles: x {e_x:.2e} (oracle f32 drift {d_x:.2e}), F_trial {e_F:.2e} (drift {d_F:.2e}); still inverted: "
          f"{int((np.linalg.det(F_h.astype(np.float64)) < 0).sum())}")
    assert e_x < 1e-4 and e_F < max(1e-4, DRIFT_K * d_F)
    assert h.out_of_bounds == 0


PLASTIC = [
    # (name, material id, set_parameters_dict entries) -- ids 3 (visco-plastic) and 6 (water EOS) have no entry in the
    # reference's name map (mpm_solver_warp.py:20-26 excludes "visplas"/"fluid"; "stationary" is id 6 with bulk = 0), so
    # they are assigned per particle, as material_field.py:343-363 does
    ("sand", 2, dict(material="sand", friction_angle=30.0)),
    ("metal", 1, dict(material="metal", yield_stress=3e3, hardening=1, xi=0.05)),
    ("snow", 5, dict(material="snow", yield_stress=3e3, hardening=0, softening=0.1)),
    ("visplas", 3, dict(material="jelly", yield_stress=2e3, plastic_viscosity=10.0)),
    ("water", 6, dict(material="stationary")),
]


@pytest.mark.parametrize("bits", SCATTER_MODES)
@pytest.mark.parametrize("name,mid,extra", PLASTIC, ids=[m[0] for m in PLASTIC])
def test_plastic_materials_rollout(hip_device, name, mid, extra, bits):
    """Every constitutive branch of compute_stress_from_F_trial (mpm_utils.py:467-526) on the device, 60 substeps from a
    deformed state, against the float64 oracle.  Tolerances: x, F at the north-star 1e-4; every quantity additionally
    gets DRIFT_K times the distance of the float32 ORACLE from the float64 oracle (the same restatement run in the
    reference's own precision), because the return mappings are discontinuous maps of F (yield / no yield, the sand
    cone's three cases): a particle sitting within rounding distance of the yield surface takes the other branch in
    float32, which is a property of the reference's arithmetic, not of this implementation.  The measured numbers are
    printed; where the float32 oracle's drift is below 2.5e-5 the bar is just 1e-4."""
    n = 8000
    sc = mpm_ball_scene(n, seed=6, scenario="ball")
    sc["params"] = dict(g=[0.0, 0.0, -9.8], E=1e5, nu=0.3, density=1000.0, **extra)
    h = make_hip(sc, per_particle=False, bits=bits)
    o, o32 = make_oracle(sc, "f64", per_particle=False), make_oracle(sc, "f32", per_particle=False)
    for s in (h, o, o32):
        s.set_per_particle(material=np.full(n, mid, np.int32))
        if mid == 6:
            s.finalize_mu_lam_bulk()
    # start deformed so the return mappings are exercised from step 1
    rng = np.random.default_rng(1)
    Ft = (np.eye(3) + 0.05 * rng.normal(size=(n, 3, 3))).astype(np.float32)
    h.set_field("F_trial", Ft.reshape(n, 9)); o.field("F_trial")[:] = Ft; o32.field("F_trial")[:] = Ft
    h.run(sc["dt"], 60); o.run(sc["dt"], 60); o32.run(sc["dt"], 60)
    v_rms = float(np.linalg.norm(o.field("v")) / np.sqrt(n))
    report = []
    for f, shape in (("x", (n, 3)), ("F", (n, 3, 3)), ("v", (n, 3)), ("yield_stress", (n,))):
        if f == "yield_stress" and mid not in (1, 3, 5):
            continue
        ref = o.field(f).astype(np.float64)
        scale = max(np.linalg.norm(ref), (v_rms * np.sqrt(n)) if f == "v" else 0.0, 1e-300)
        err = float(np.linalg.norm(get(h, f).reshape(shape).astype(np.float64) - ref) / scale)
        drift = float(np.linalg.norm(o32.field(f).astype(np.float64) - ref) / scale)
        report.append(f"{f}: hip {err:.2e} / f32-oracle {drift:.2e}")
        assert err < max(1e-4, DRIFT_K * drift), (name, f, err, drift)
    print(f"{name}: " + "; ".join(report))
    assert float(np.abs(get(h, "stress")).max()) > 0
    if mid != 6:
        assert rel_l2(get(h, "F"), get(h, "F_trial")) > 1e-6   # the return mapping moved F
    assert h.out_of_bounds == 0


@pytest.mark.parametrize("bits", SCATTER_MODES)
def test_rollout_parity_config3(hip_device, bits):
    """The north-star MPM configuration (BASELINE configs[2]: 100 000 particles, n_grid 50, the tree scenario -- the scene
    bench.py times) for 1 000 substeps against the float64 C oracle.  The oracle needs ~4 min per precision at this size,
    so its trajectory is a committed fixture (tests/golden/make_mpm_golden.py; tests/test_mpm_oracle.py re-runs its first
    checkpoint live): every 16th particle's x, v, C, F_trial at substeps 20 / 100 / 500 / 1000, whole-population norms,
    momentum and centre of mass, and the float32 oracle's own drift from the float64 one.
    Bar: x and F_trial <= 1e-4 outright; the displacement, v and C <= max(1e-4, DRIFT_K x the float32 oracle's drift), v and C
    measured on the scale of the velocity field as in test_rollout_parity."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpm_config3.npz"))
    n, stride = int(g["n"]), int(g["stride"])
    sc = mpm_ball_scene(n, seed=int(g["seed"]))
    assert sc["n_grid"] == int(g["n_grid"]) and sc["dt"] == float(g["dt"])
    h = make_hip(sc, bits=bits)
    x0 = sc["x"].astype(np.float64)
    mass = get(h, "mass").astype(np.float64)
    inv_dx = sc["n_grid"] / sc["grid_lim"]
    done = 0
    for cp in [int(c) for c in g["checkpoints"]]:
        h.run(sc["dt"], cp - done)
        done = cp
        x, v, C, F = (get(h, f).astype(np.float64) for f in ("x", "v", "C", "F_trial"))
        C, F = C.reshape(-1, 3, 3), F.reshape(-1, 3, 3)
        d_x, d_disp, d_v, d_C, d_F = (float(t) for t in g[f"drift_{cp}"])
        m = n // stride + (1 if n % stride else 0)
        ns = np.sqrt(m)
        v_rms = float(np.linalg.norm(g[f"v_{cp}"]) / ns)
        c_scale = max(v_rms * inv_dx, float(np.linalg.norm(g[f"C_{cp}"]) / ns))
        e_x = rel_l2(x[::stride], g[f"x_{cp}"])
        e_F = rel_l2(F[::stride], g[f"F_trial_{cp}"])
        e_disp = rel_l2((x - x0)[::stride], g[f"x_{cp}"] - x0[::stride])
        e_v = float(np.linalg.norm(v[::stride] - g[f"v_{cp}"]) / ns) / v_rms
        e_C = float(np.linalg.norm(C[::stride] - g[f"C_{cp}"]) / ns) / c_scale
        # the float32 oracle's drift on the same scales (the fixture stores norm-relative numbers over all particles)
        dv = d_v * float(g[f"norms_{cp}"][1]) / np.sqrt(n) / v_rms
        dC = d_C * float(g[f"norms_{cp}"][2]) / np.sqrt(n) / c_scale
        norms = np.array([np.linalg.norm(x - x0), np.linalg.norm(v), np.linalg.norm(C), np.linalg.norm(F - np.eye(3))])
        e_norm = np.abs(norms / g[f"norms_{cp}"] - 1.0)
        e_p = float(np.linalg.norm((mass[:, None] * v).sum(0) - g[f"momentum_{cp}"]) / (mass.sum() * float(g[f"norms_{cp}"][1]) / np.sqrt(n)))
        e_com = float(np.abs((mass[:, None] * x).sum(0) / mass.sum() - g[f"com_{cp}"]).max())
        agg = g[f"drift_agg_{cp}"]   # the float32 oracle's own |norm ratio - 1| x4, momentum error, centre-of-mass error
        print(f"config 3 @ substep {cp}: x {e_x:.2e}, F_trial {e_F:.2e}, displacement {e_disp:.2e} (f32 oracle {d_disp:.2e}), "
              f"v {e_v:.2e} (f32 oracle {dv:.2e}), C {e_C:.2e} (f32 oracle {dC:.2e}); whole population: norms off by "
              f"{e_norm.max():.2e} (f32 oracle {agg[:4].max():.2e}), momentum {e_p:.2e} ({agg[4]:.2e}), centre of mass {e_com:.2e} ({agg[5]:.2e})")
        # Round 6: the float32 oracle's own trajectory is in the fixture, so the product is ALSO compared with the
        # reference's algorithm in the reference's precision directly.  `order`: how far two float32 evaluations of that algorithm are
        # from each other when only the order of the P2G sums differs (scalar oracle vs its OpenMP build).
        x32, v32, C32, F32 = (g[f"{f}32_{cp}"].astype(np.float64) for f in ("x", "v", "C", "F_trial"))
        f_disp = rel_l2((x - x0)[::stride], x32 - x0[::stride])
        f_v = float(np.linalg.norm(v[::stride] - v32) / ns) / v_rms
        f_C = float(np.linalg.norm(C[::stride] - C32.reshape(-1, 3, 3)) / ns) / c_scale
        f_F = rel_l2(F[::stride], F32.reshape(-1, 3, 3))
        o_disp, o_v, o_C, o_F = (float(t) for t in g[f"order_{cp}"])
        print(f"config 3 @ substep {cp}: product vs the FLOAT32 oracle: displacement {f_disp:.2e}, v {f_v:.2e}, C {f_C:.2e}, F_trial {f_F:.2e}  "
              f"(two float32 summation orders of the oracle apart: {o_disp:.2e}, {o_v:.2e}, {o_C:.2e}, {o_F:.2e}; "
              f"float32 oracle vs float64: {d_disp:.2e}, {dv:.2e}, {dC:.2e})")
        # measured 0.03-0.74 of the float32 oracle's own distance from float64 (profiles/r6_config3_drift_attribution.txt: the drift is the
        # float32 accumulation of x += dt v and of F_trial = (I + dt grad v) F; two builds that round the 3x3 product differently part by a
        # fraction of it).  Bar: never further from the float32 oracle than that oracle is from float64.
        assert f_disp < max(1e-4, d_disp) and f_v < max(1e-4, dv) and f_C < max(1e-4, dC) and f_F < 1e-4
        assert np.isfinite(x).all() and np.isfinite(v).all()
        assert e_x < 1e-4 and e_F < 1e-4
        assert e_disp < max(1e-4, DRIFT_K * d_disp)
        assert e_v < max(1e-4, DRIFT_K * dv) and e_C < max(1e-4, DRIFT_K * dC)
        # aggregates over all 100 000 particles: systematic (not averaging-out) errors show here.  float32 positions lose
        # the part of dt * v below half an ulp of x (dt * v ~ 1e-7 ... 1e-6 against ulp(x) = 1.2e-7 in this quiet scene),
        # in the oracle's float32 build exactly as on the device, hence the float32 oracle's own numbers as the yardstick
        assert (e_norm < np.maximum(1e-4, DRIFT_K * agg[:4])).all(), (e_norm, agg[:4])
        assert e_p < max(1e-4, DRIFT_K * agg[4]) and e_com < max(1e-7, DRIFT_K * agg[5])
    assert h.out_of_bounds == 0 and int(g["oob"][0]) == 0
    assert abs(h.time - 1000 * sc["dt"]) < 1e-9


def test_compensated_positions_follow_the_float64_trajectory(hip_device):
    """set_scalar "compensated_x" (off by default: the reference's float32 `x += dt v`, mpm_utils.py:447, drops what lies below
    ulp(x)/2 of every increment -- most of the motion in the quiet north-star scene, which is why its displacement sits 1.6e-2 from
    the float64 oracle in the reference's precision exactly as here).  With the remainder carried in three more words per particle
    the stored x is the float32 rounding of the accumulated position: against the committed float64 trajectory of BASELINE
    configs[2] the displacement error must fall several-fold at every checkpoint (measured 7.5x / 7.9x / 4.2x at substeps 20 / 100 /
    500, profiles/r5c_compensated_x_experiment.txt; v and C do not move: they are not a position-rounding effect), x itself
    improves, and everything else is untouched -- v, C, F_trial stay within float32 roundoff of the default mode's."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpm_config3.npz"))
    n, stride = int(g["n"]), int(g["stride"])
    sc = mpm_ball_scene(n, seed=int(g["seed"]))
    x0 = sc["x"].astype(np.float64)
    plain, comp = make_hip(sc), make_hip(sc)
    comp._set_scalar("compensated_x", 1)
    done = 0
    for cp in (20, 100, 500):
        plain.run(sc["dt"], cp - done); comp.run(sc["dt"], cp - done); done = cp
        want = g[f"x_{cp}"] - x0[::stride]
        e_plain = rel_l2((get(plain, "x").astype(np.float64) - x0)[::stride], want)
        e_comp = rel_l2((get(comp, "x").astype(np.float64) - x0)[::stride], want)
        d_disp = float(g[f"drift_{cp}"][1])
        print(f"compensated x @ substep {cp}: displacement vs float64 {e_comp:.2e} (default mode {e_plain:.2e}, float32 oracle {d_disp:.2e}); "
              f"x {rel_l2(get(comp, 'x')[::stride], g[f'x_{cp}']):.2e} (default {rel_l2(get(plain, 'x')[::stride], g[f'x_{cp}']):.2e})")
        assert e_comp < 0.4 * e_plain and e_comp < 0.4 * d_disp
        assert rel_l2(get(comp, "x")[::stride], g[f"x_{cp}"]) < rel_l2(get(plain, "x")[::stride], g[f"x_{cp}"])
        for f in ("v", "C", "F_trial"):
            assert rel_l2(get(comp, f), get(plain, f)) < (1e-3 if f != "F_trial" else 1e-6), f      # same dynamics, different rounding of x
    assert comp.out_of_bounds == 0
    # new positions carry no remainder
    comp.import_particle_x_from_torch(torch.from_numpy(sc["x"]))
    plain.import_particle_x_from_torch(torch.from_numpy(sc["x"]))
    assert np.array_equal(get(comp, "x"), get(plain, "x"))


@pytest.mark.parametrize("material", ["sand", "snow", "metal", "mixed"])
@pytest.mark.parametrize("bits", SCATTER_MODES)
def test_plastic_reference_configs_100k(hip_device, material, bits):
    """The reference's own plastic configurations (PG/config/objaverse/custom_{sand,snow,metal}_config.json: parameters,
    n_grid 200 / 120, substep 2e-5 / 1e-5, gravity, damping, boundary conditions) with 100 000 particles for 200 substeps,
    against the float64 C oracle's committed trajectory (tests/golden/make_mpm_plastic_golden.py, which perturbs the initial
    F and v so that the return mappings work from the first substep).  "mixed": the mixed-material scene bench.py times -- ids
    0 / 1 / 2 / 5 drawn per particle under one set of solver scalars, so every wave holds all four constitutive branches.  Bar: x and F <= 1e-4 outright; v (on the scale of
    rms|v|) and the yield stress <= max(1e-4, DRIFT_K x the float32 oracle's own drift)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_mpm_plastic_golden import plastic_scene, start
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"mpm_plastic_{material}.npz"))
    n, stride = int(g["n"]), int(g["stride"])
    sc = plastic_scene(material)
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    h = MPM_Simulator_WARP(10)
    h.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]),
                                   n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
    start(h, sc, lambda f, a: h.set_field(f, a.reshape(n, -1)))
    h._set_scalar("scatter_bits", bits)
    done = 0
    for cp in [int(c) for c in g["checkpoints"]]:
        h.run(sc["dt"], cp - done); done = cp
        d_x, d_disp, d_v, d_F, d_ys = (float(t) for t in g[f"drift_{cp}"])
        x, v, F, ys = (get(h, f).astype(np.float64) for f in ("x", "v", "F", "yield_stress"))
        F = F.reshape(-1, 3, 3)
        e_x, e_F = rel_l2(x[::stride], g[f"x_{cp}"]), rel_l2(F[::stride], g[f"F_{cp}"])
        e_v = rel_l2(v[::stride], g[f"v_{cp}"])
        e_ys = rel_l2(ys[::stride], g[f"yield_stress_{cp}"]) if material != "sand" else 0.0
        print(f"{material} @ substep {cp}: x {e_x:.2e} (f32 oracle {d_x:.2e}), F {e_F:.2e} ({d_F:.2e}), v {e_v:.2e} ({d_v:.2e}), "
              f"yield stress {e_ys:.2e} ({d_ys:.2e})")
        assert np.isfinite(x).all() and np.isfinite(F).all()
        assert e_x < 1e-4 and e_F < 1e-4
        assert e_v < max(1e-4, DRIFT_K * d_v) and e_ys < max(1e-4, DRIFT_K * d_ys)
    print(f"{material}: {100 * float(g['yielded_fraction']):.0f} % of the particles yielded within the 200 substeps")
    assert float(g["yielded_fraction"]) > 0.05                    # the return mapping is really at work in this scene
    assert h.out_of_bounds == 0 and int(g["oob"]) == 0


def test_boundary_conditions_and_modifiers(hip_device):
    sc = mpm_ball_scene(8000, seed=8, scenario="ball")
    sc["params"] = dict(material="jelly", g=[0.0, 0.0, -2.0], E=5e4, nu=0.3, density=500.0, rpic_damping=0.1, grid_v_damping_scale=0.999)
    sc["bcs"] = [dict(type="bounding_box"),
                 dict(type="cuboid", point=[1.0, 1.0, 0.55], size=[0.3, 0.3, 0.05], velocity=[0.0, 0.2, 0.1], start_time=0.0, end_time=3e-3, reset=1),
                 dict(type="surface_collider", point=[1.0, 1.0, 0.52], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0, start_time=0.0, end_time=1e3),
                 dict(type="surface_collider", point=[0.6, 1.0, 1.0], normal=[1.0, 0.0, 0.0], surface="slip", friction=0.5, start_time=0.0, end_time=1e3),
                 dict(type="enforce_particle_translation", point=[1.0, 1.0, 1.4], size=[0.2, 0.2, 0.1], velocity=[0.1, 0.0, 0.0], start_time=0.0, end_time=2e-3),
                 dict(type="particle_impulse", force=[0.0, 0.02, 0.0], num_dt=3, start_time=1e-3)]
    h, o = make_hip(sc), make_oracle(sc, "f64")
    for s in (h, o):
        s.enforce_particle_velocity_rotation(point=[1.0, 1.0, 1.0], normal=[0.0, 0.0, 1.0], half_height_and_radius=[0.05, 0.2],
                                             rotation_scale=0.5, translation_scale=0.01, start_time=0.0, end_time=1.5e-3)
    o32 = make_oracle(sc, "f32")
    o32.enforce_particle_velocity_rotation(point=[1.0, 1.0, 1.0], normal=[0.0, 0.0, 1.0], half_height_and_radius=[0.05, 0.2],
                                           rotation_scale=0.5, translation_scale=0.01, start_time=0.0, end_time=1.5e-3)
    # 30 substeps: every modifier and BC has been active, and the state is still well conditioned
    h.run(sc["dt"], 30); o.run(sc["dt"], 30); o32.run(sc["dt"], 30)
    assert rel_l2(get(h, "x"), o.field("x")) < 1e-5
    drift = rel_l2(o32.field("v"), o.field("v"))
    err = rel_l2(get(h, "v"), o.field("v"))
    direct = rel_l2(get(h, "v"), o32.field("v"))
    print(f"bc test v @30: hip-vs-f64 {err:.3e}  oracle f32-vs-f64 {drift:.3e}  hip-vs-oracle-f32 {direct:.3e}")
    # the common bar of this file (this test alone allowed 1e-3).  The float32 and float64 oracles differ by
    # ~1e-2 here for a systematic reason -- the velocity pins and the impulse window act on float32 time / float32 state --
    # and the HIP solver follows the float32 arithmetic, so it is ALSO held to the float32 oracle directly.
    assert err < max(1e-4, DRIFT_K * drift)
    assert direct < 1e-4
    assert rel_l2(get(h, "F_trial").reshape(-1, 3, 3), o.field("F_trial")) < 1e-4
    # 20 more: through the cuboid's end_time and its 15-substep "reset" window, which zeroes the whole grid
    # (mpm_solver_warp.py:895-897); afterwards v restarts from roundoff-sized forces, so it is only required to be
    # as close to the float64 oracle as the float32 oracle is.
    h.run(sc["dt"], 20); o.run(sc["dt"], 20); o32.run(sc["dt"], 20)
    assert rel_l2(get(h, "x"), o.field("x")) < 1e-5
    assert rel_l2(get(h, "F_trial").reshape(-1, 3, 3), o.field("F_trial")) < 1e-4
    drift = rel_l2(o32.field("v"), o.field("v"))
    err = rel_l2(get(h, "v"), o.field("v"))
    print(f"bc test v @50: hip-vs-f64 {err:.3e}  oracle f32-vs-f64 {drift:.3e}  hip-vs-oracle-f32 {rel_l2(get(h, 'v'), o32.field('v')):.3e}")
    assert err < max(1e-4, DRIFT_K * drift)


def test_regrid_after_load_keeps_everything_but_the_grid(hip_device):
    """set_parameters_dict(n_grid=...) AFTER the particles were loaded (mpm_solver_warp.py:315-342 re-allocates the three grid
    arrays and recomputes dx, nothing else): material, gravity, model scalars, boundary conditions, particle modifiers and the
    time must survive (the first implementation re-created the handle and lost them)."""
    sc = mpm_ball_scene(6000, seed=3, scenario="ball", n_grid=40)
    sc["params"] = dict(material="sand", g=[0.0, 0.0, -9.8], E=1e5, nu=0.3, density=1000.0, friction_angle=35.0, rpic_damping=0.05,
                        grid_v_damping_scale=0.999)
    sc["bcs"] = [dict(type="bounding_box"),
                 dict(type="surface_collider", point=[1.0, 1.0, 0.6], normal=[0.0, 0.0, 1.0], surface="sticky", friction=0.0, start_time=0.0, end_time=1e3),
                 dict(type="particle_impulse", force=[0.0, 0.05, 0.0], num_dt=5, start_time=0.0)]
    h = make_hip(sc, per_particle=False)
    h.run(sc["dt"], 5)                                   # time advances, the impulse window is half used
    h.set_parameters_dict({"n_grid": 50})                # same material, scalars, BCs; finer grid
    assert h.n_grid == 50 and abs(h._get_scalar("dx") - np.float32(2.0 / 50)) < 1e-9 and abs(h.time - 5 * sc["dt"]) < 1e-12
    assert int(get(h, "material")[0]) == 2 and abs(h._get_scalar("rpic_damping") - np.float32(0.05)) < 1e-9
    # the oracle: the same 5 substeps on the 40-grid, its particle state moved onto a 50-grid solver set up the same way
    # (float64 as the yardstick, float32 for the rounding floor of this nearly stress-free sand ball in free fall)
    res = {}
    for prec in ("f64", "f32"):
        o40 = make_oracle(sc, prec, per_particle=False)
        o40.run(sc["dt"], 5)
        sc50 = dict(sc); sc50["n_grid"] = 50
        o50 = make_oracle(sc50, prec, per_particle=False)
        for f in ("x", "v", "C", "F", "F_trial"):
            o50.field(f)[:] = o40.field(f)
        o50._lib.mpm_set_scalar(o50._h, b"time", float(o40.time))
        o50.run(sc["dt"], 25)
        res[prec] = {f: np.array(o50.field(f), np.float64) for f in ("x", "v", "F")}
    h.run(sc["dt"], 25)
    assert rel_l2(get(h, "x"), res["f64"]["x"]) < 1e-5
    assert rel_l2(get(h, "F").reshape(-1, 3, 3), res["f64"]["F"]) < 1e-4
    v_rms = float(np.linalg.norm(res["f64"]["v"]) / np.sqrt(6000))
    err = float(np.linalg.norm(get(h, "v") - res["f64"]["v"]) / np.sqrt(6000)) / v_rms
    drift = float(np.linalg.norm(res["f32"]["v"] - res["f64"]["v"]) / np.sqrt(6000)) / v_rms
    print(f"regrid: v hip-vs-f64 {err:.2e}, oracle f32-vs-f64 {drift:.2e}")
    assert err < max(1e-4, DRIFT_K * drift)
    assert h.out_of_bounds == 0


def test_exports_cov_and_rotation(hip_device):
    sc = mpm_ball_scene(5000, seed=9, scenario="ball")
    h, o = make_hip(sc), make_oracle(sc, "f32")
    h.run(sc["dt"], 30); o.run(sc["dt"], 30)
    assert rel_l2(h.export_particle_cov_to_torch().cpu().numpy(), o.export_cov()) < 1e-5
    assert rel_l2(h.export_particle_R_to_torch().cpu().numpy(), o.export_R()) < 1e-5
    assert h.export_particle_F_to_torch().shape == (5000, 9)
    # exports are persistent per-field tensors refreshed in place (the reference returns aliases of solver memory): no
    # allocation per call, and a tensor the caller kept sees the next export
    xa = h.export_particle_x_to_torch()
    x_before = xa.clone()
    h.run(sc["dt"], 5)
    xb = h.export_particle_x_to_torch()
    assert xa.data_ptr() == xb.data_ptr() and not torch.equal(x_before, xb) and torch.equal(xb, h.get_field("x"))
    assert h.mpm_state.particle_x.numpy().shape == (5000, 3)
    E = torch.full((5000,), 3.0e5)
    h.mpm_model.E = E  # gs_simulation.py:528 style assignment
    assert np.allclose(h.mpm_model.E.numpy(), 3.0e5)


def test_live_exports_mode(hip_device):
    """mpm_solver_warp.py:659-741 hand out aliases of solver memory: a tensor kept across p2g2p reads current data.  Default
    here: a kept tensor is refreshed by the next export call (documented deviation, lets p2g2p be deferred).  Opt-in
    `live_exports = True`: every tensor handed out is refreshed after each p2g2p -- the reference's observable behaviour."""
    sc = mpm_ball_scene(4000, seed=3, scenario="ball")
    live, ref = make_hip(sc), make_hip(sc)
    live.live_exports = True
    x_held, v_held = live.export_particle_x_to_torch(), live.export_particle_v_to_torch()
    F_held = live.export_particle_F_to_torch()
    for i in range(7):
        live.p2g2p(i, sc["dt"])
        ref.p2g2p(i, sc["dt"])
        assert live._pending == 0            # nothing deferred in this mode
    torch.cuda.synchronize()
    for held, name, tol in ((x_held, "x", 1e-6), (v_held, "v", 1e-4), (F_held, "F", 1e-6)):
        assert torch.equal(held, live.get_field(name)), name    # the kept tensors ARE the current state
        # ... and that state is the deferred path's (single-substep launches: other schedule, same maths)
        assert rel_l2(held.cpu().numpy(), ref.get_field(name).cpu().numpy()) < tol, name
    # default mode: the kept tensor is stale until the next export call
    x_kept = ref.export_particle_x_to_torch()
    before = x_kept.clone()
    ref.p2g2p(7, sc["dt"])
    assert torch.equal(x_kept, before)
    ref.export_particle_x_to_torch()
    assert not torch.equal(x_kept, before)


def test_undefined_material_raises(hip_device):
    from pixie_amd.mpm_solver import MPM_Simulator_WARP, get_material_name
    s = MPM_Simulator_WARP(16, n_grid=8, grid_lim=1.0)
    with pytest.raises(TypeError):
        s.set_parameters_dict({"material": "fluid"})  # excluded from the name map, mpm_solver_warp.py:20-21
    assert get_material_name("jelly") == 0 and get_material_name("rigid") == 6 and get_material_name(3) == -1


def test_full_size_properties(hip_device):
    """BASELINE config 3 size (100k particles, n_grid 50, 500 substeps): properties that do not need the oracle."""
    sc = mpm_ball_scene(100_000, seed=0)
    sc["bcs"] = []; sc["fix_ground"] = None
    sc["params"] = dict(material="jelly", g=[0.0, 0.0, 0.0], E=2e6, nu=0.4, density=200.0)  # no damping, no BC, no gravity
    h = make_hip(sc)
    rng = np.random.default_rng(0)
    v0 = (0.2 * rng.normal(size=(100_000, 3))).astype(np.float32)
    h.set_field("v", v0)
    mass = get(h, "mass").astype(np.float64)
    p0 = (mass[:, None] * v0).sum(0)
    h.run(sc["dt"], 500)
    v = get(h, "v").astype(np.float64)
    assert np.isfinite(v).all()
    p1 = (mass[:, None] * v).sum(0)
    # APIC transfers conserve linear momentum; fp32 atomics leave ~1e-5 relative noise
    assert np.linalg.norm(p1 - p0) / np.linalg.norm(mass[:, None] * v0, axis=None) < 1e-4
    J = np.linalg.det(get(h, "F_trial").reshape(-1, 3, 3).astype(np.float64))
    assert J.min() > 0.5 and J.max() < 2.0
    assert h.out_of_bounds == 0
    assert abs(h.time - 500 * sc["dt"]) < 1e-9


def test_full_size_sand_config(hip_device):
    """The reference's plastic-material configuration (custom_sand_config.json: Drucker-Prager sand, n_grid 200,
    substep 2e-5, bounding box + sticky floor at z = 0.48) with 1M particles: size-independent properties."""
    n = 1_000_000
    sc = mpm_ball_scene(n, seed=0, n_grid=200, dt=2e-5, scenario="sand")
    h = make_hip(sc)
    z0 = get(h, "x")[:, 2].astype(np.float64)
    h.run(sc["dt"], 400)
    x = get(h, "x").astype(np.float64); v = get(h, "v").astype(np.float64)
    F = get(h, "F_trial").reshape(-1, 3, 3).astype(np.float64)
    assert np.isfinite(x).all() and np.isfinite(v).all() and np.isfinite(F).all()
    assert h.out_of_bounds == 0
    t = 400 * sc["dt"]
    # free fall until the floor is reached: the bulk has dropped by g t^2 / 2 (8 mm here), nothing has gone through the floor
    drop = z0 - x[:, 2]
    high = z0 > 1.0                               # particles that cannot have felt the floor yet
    assert abs(np.median(drop[high]) - 0.5 * 9.8 * t * t) < 0.15 * 0.5 * 9.8 * t * t
    assert x[:, 2].min() > 0.48 - 2.0 / 200
    # Drucker-Prager projection keeps the elastic deformation gradient near the identity for stiff sand
    J = np.linalg.det(F)
    assert J.min() > 0.8 and J.max() < 1.2
    assert int(h._get_scalar("dropped_particles")) == 0


def test_full_size_mixed_material_scene_properties(hip_device):
    """The mixed-material scene bench.py times (1 M particles, ids 0 / 1 / 2 / 5 drawn per particle: every work item holds all
    four constitutive branches, laid out class by class), at full size through properties that do not need an oracle:
    total momentum follows P <- (P + M g dt) * damping exactly as the grid update prescribes (stress forces are internal, the
    walls are out of reach in 150 substeps), nothing becomes non-finite or leaves the grid, the material ids survive the
    re-binning permutations, and two runs agree bit for bit."""
    from pixie_amd.synthetic import mpm_plastic_scene, start_plastic
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    n, steps = 1_000_000, 150
    sc = mpm_plastic_scene("mixed", n, seed=0)

    def make():
        h = MPM_Simulator_WARP(10)
        h.load_initial_data_from_torch(torch.from_numpy(sc["x"]), torch.from_numpy(sc["vol"]), torch.from_numpy(sc["cov"]),
                                       n_grid=sc["n_grid"], grid_lim=sc["grid_lim"])
        start_plastic(h, sc, lambda f, a: h.set_field(f, a.reshape(n, -1)))
        return h
    h = make()
    mass = get(h, "mass").astype(np.float64)
    P = (mass[:, None] * sc["v0"].astype(np.float64)).sum(0)
    M, g, d = mass.sum(), np.array(sc["params"]["g"]), sc["params"]["grid_v_damping_scale"]
    for _ in range(steps):
        P = (P + M * g * sc["dt"]) * d
    h.run(sc["dt"], steps)
    x, v, F = get(h, "x"), get(h, "v").astype(np.float64), get(h, "F_trial")
    assert np.isfinite(x).all() and np.isfinite(v).all() and np.isfinite(F).all() and h.out_of_bounds == 0
    P_h = (mass[:, None] * v).sum(0)
    err = np.linalg.norm(P_h - P) / np.linalg.norm(P)
    print(f"mixed 1 M: total momentum after {steps} substeps off the prescribed recurrence by {err:.2e}; rebins {int(h._get_scalar('n_rebins'))}")
    assert err < 2e-5
    assert np.array_equal(get(h, "material"), sc["material"])
    assert rel_l2(get(h, "F"), F) > 1e-6                         # the return mappings are at work
    ys = get(h, "yield_stress")
    assert (ys[sc["material"] == 1] != np.float32(sc["params"]["yield_stress"])).mean() > 0.05   # metal hardened where it yielded
    h2 = make()
    h2.run(sc["dt"], steps)
    assert np.array_equal(get(h2, "x"), x) and np.array_equal(get(h2, "F_trial"), F)


def test_tiny_and_degenerate_scenes(hip_device):
    """Edge cases of the particle set: one particle, two particles in one cell, a handful spread over distant blocks, every particle
    frozen (selection = 1: the reference's kernels skip it everywhere), zero substeps -- each against the float64 oracle (or the
    exact no-op it must be)."""
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    base = mpm_ball_scene(64, seed=12, scenario="ball")
    for tag, idx in (("one particle", [0]), ("two in one cell", [0, 0]), ("five far apart", [0, 9, 17, 33, 60])):
        sc = {k: (v[idx].copy() if isinstance(v, np.ndarray) and v.shape[:1] == (64,) else v) for k, v in base.items()}
        if tag == "two in one cell":
            sc["x"][1] = sc["x"][0] + np.float32(1e-3)
        h, o = make_hip(sc), make_oracle(sc, "f64")
        h.run(sc["dt"], 0)                                    # nothing to do
        assert np.array_equal(get(h, "x"), sc["x"]) and h.time == 0.0
        h.run(sc["dt"], 30); o.run(sc["dt"], 30)
        ex, ev = rel_l2(get(h, "x"), o.field("x")), rel_l2(get(h, "v"), o.field("v"))
        print(f"{tag}: x {ex:.1e} v {ev:.1e} after 30 substeps of free fall")
        assert ex < 1e-6 and ev < 1e-4 and h.out_of_bounds == 0
        assert abs(float(get(h, "v")[0, 2]) + 9.8 * 30 * sc["dt"]) < 1e-4      # g t (the walls are far away)
    sc = dict(base)
    h = make_hip(sc)
    h.set_field("selection", np.ones(64, np.int32))
    h.run(sc["dt"], 20)
    assert np.array_equal(get(h, "x"), sc["x"]) and float(np.abs(get(h, "v")).max()) == 0.0
    assert np.isfinite(h.get_field("grid_v_out").cpu().numpy()).all()


def test_particles_leaving_the_grid_freeze_with_a_defined_state(hip_device):
    """A particle whose stencil leaves the grid is undefined behaviour in the reference (out-of-range grid indices).  Here it is frozen where
    it is (selection 2), counted once, and -- since the fused kernel keeps v, C and F_trial of active particles in registers and a
    re-binning no longer moves those rows for them -- given a defined observable state: v = 0, C = 0, F_trial = its last F.  Everything
    stays finite, the rest of the scene goes on, and re-binnings in between (forced) do not scramble the frozen rows."""
    sc = mpm_ball_scene(20000, seed=5, scenario="ball")
    sc["bcs"] = []                                  # no bounding box: nothing stops them
    sc["params"] = dict(sc["params"], g=[0.0, 0.0, 0.0])
    fast = np.arange(20000) % 50 == 0               # 400 particles taken out of the ball and put in a sheet 0.2 from the +x wall, flying at it
    rng = np.random.default_rng(3)
    sc["x"] = sc["x"].copy()
    sc["x"][fast] = np.stack([np.full(400, 1.75), rng.uniform(0.6, 1.4, 400), rng.uniform(0.6, 1.4, 400)], 1).astype(np.float32)
    h = make_hip(sc)
    v0 = np.zeros((20000, 3), np.float32)
    v0[fast] = [25.0, 0.0, 0.0]                     # 0.06 cells per substep: the stencil leaves the grid at x = 1.94 after ~75 substeps
    h.set_field("v", v0)
    h._set_scalar("resort_interval", 4)
    h.run(sc["dt"], 300)
    sel = get(h, "selection").reshape(-1)
    gone = sel == 2
    print(f"{int(gone.sum())} particles left the grid, {int(h._get_scalar('n_rebins'))} re-binnings, dropped {h._get_scalar('dropped_particles'):.0f}")
    assert h.out_of_bounds == int(gone.sum()) + int(h._get_scalar("dropped_particles")) and int(gone.sum()) == 400 and int(h._get_scalar("n_rebins")) > 10
    assert np.array_equal(gone, fast)
    x, v, C, Ft, F = (get(h, f) for f in ("x", "v", "C", "F_trial", "F"))
    for a in (x, v, C, Ft, F):
        assert np.isfinite(a).all()
    assert np.abs(v[gone]).max() == 0.0 and np.abs(C[gone]).max() == 0.0
    assert np.array_equal(Ft[gone], F[gone]) and np.abs(np.linalg.det(Ft[gone].reshape(-1, 3, 3)) - 1.0).max() < 0.5
    assert (np.abs(x[gone] - 1.0).max(1) > 0.8).all()     # frozen at a wall, where they left
    h.run(sc["dt"], 50)                             # and stay exactly there
    assert np.array_equal(get(h, "x")[gone], x[gone]) and np.abs(get(h, "v")[gone]).max() == 0.0
    assert int((get(h, "selection").reshape(-1) == 2).sum()) >= int(gone.sum())


def test_work_item_capacity_follows_the_scene_density(hip_device):
    """item_cap "auto": 256-thread work items in dense scenes, 128-thread ones -- from the FIRST binning on: the choice is made from that
    binning's own block histogram -- where few blocks hold more than 128 particles (<= 15 % more work items; a 256-thread workgroup then
    runs two waves without a particle); a forced capacity stays; the choice changes work-item composition only -- the trajectory stays within
    packed-scatter quantisation of the other choice; and a handle re-used for a new scene decides afresh, like a new handle (the
    flag used to survive set_field("x")).  (Not the new handle's BITS: the order inside a block ranks particles by their previous slot, which
    a re-used handle inherits from its old scene -- the same sums in another order, held to the usual packed-scatter bars.)"""
    dense = make_hip(mpm_ball_scene(100_000, seed=2))                      # 100 k in 50^3: ~190 particles per occupied block
    dense.run(1e-4, 40)
    assert int(dense._get_scalar("item_cap")) == 256 and int(dense._get_scalar("n_rebins")) >= 2
    sc = mpm_ball_scene(100_000, seed=2, n_grid=160)                       # the same particles in 160^3: ~10 per occupied block
    sparse, forced = make_hip(sc), make_hip(sc)
    forced._set_scalar("item_cap", 256)
    sparse.run(sc["dt"], 1)
    assert int(sparse._get_scalar("item_cap")) == 128 and int(sparse._get_scalar("n_rebins")) == 1
    sparse.run(sc["dt"], 39); forced.run(sc["dt"], 40)
    assert int(sparse._get_scalar("item_cap")) == 128 and int(forced._get_scalar("item_cap")) == 256
    assert int(sparse._get_scalar("n_work_items")) <= 1.15 * int(forced._get_scalar("n_work_items"))
    for f in ("x", "F_trial"):
        assert rel_l2(get(sparse, f), get(forced, f)) < 1e-6, f
    assert rel_l2(get(sparse, "v"), get(forced, "v")) < 1e-4
    assert sparse.out_of_bounds == 0
    # the sparse handle re-loaded with the dense scene's positions == a fresh handle on the dense scene, bit for bit
    xd = (1.0 + 0.12 * (sc["x"] - 1.0)).astype(np.float32)          # the same ball squeezed to 10 cells across: thousands of particles per block
    fresh = make_hip(sc)                                              # (same BCs and modifier masks as `sparse`)
    for hnd in (sparse, fresh):
        hnd.set_field("x", xd); hnd.set_field("v", np.zeros_like(xd)); hnd.set_field("C", np.zeros((100_000, 9), np.float32))
        hnd.set_field("F_trial", np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (100_000, 1)))
        hnd._set_scalar("time", 0.0)
        hnd.run(sc["dt"], 12)
    assert int(sparse._get_scalar("item_cap")) == 256 == int(fresh._get_scalar("item_cap"))
    assert int(sparse._get_scalar("n_work_items")) == int(fresh._get_scalar("n_work_items"))
    assert rel_l2(get(sparse, "x"), get(fresh, "x")) < 1e-6 and rel_l2(get(sparse, "F_trial"), get(fresh, "F_trial")) < 1e-6
    assert rel_l2(get(sparse, "v"), get(fresh, "v")) < 1e-4


def test_export_frame_for_rendering(hip_device):
    """gs_simulation.py:591-600 in one launch, against tests/golden/frame_export.npz: the REFERENCE's own
    transformation_utils.py / material_field.py:81-86 functions (executed via `ast`, tests/golden/make_frame_export_golden.py)
    applied to the positions and the `compute_cov_from_F` covariances the reference's solver code produced for scene
    `jelly_apic` of mpm_ref_golden.npz.  The product solver is put into that scene's final state (x, F_trial, init_cov)."""
    import os
    from tests._mpm_ref_driver import load_fixture
    here = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(here, "frame_export.npz"))
    scene, arrays, ref = load_fixture(os.path.join(here, "mpm_ref_golden.npz"))["jelly_apic"]
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    h = MPM_Simulator_WARP(10)
    h.load_initial_data_from_torch(torch.from_numpy(arrays["x0"]), torch.from_numpy(arrays["vol"]), torch.from_numpy(arrays["cov"]),
                                   n_grid=scene["n_grid"], grid_lim=scene["grid_lim"])
    h.set_field("x", ref["k6/x"].astype(np.float32))
    h.set_field("F_trial", ref["k6/F_trial"].astype(np.float32).reshape(-1, 9))
    gs_num = int(g["gs_num"])
    c6 = h.export_particle_cov_to_torch().cpu().numpy().reshape(-1, 6)
    assert rel_l2(c6, g["cov"]) < 1e-6                                       # compute_cov_from_F, mpm_utils.py:529-553
    for rots in (g["rot_f64"], g["rot_f32"]):                                # the caller's matrices in either precision
        pos, cov = h.export_frame_for_rendering(gs_num, float(g["scale_origin"]), torch.tensor(g["mean"]), [torch.tensor(R) for R in rots],
                                                z_shift_value=float(g["z_shift"]))
        assert pos.shape == (gs_num, 3) and cov.shape == (gs_num, 6)
        e_pos, e_cov = rel_l2(pos.cpu().numpy(), g["pos_f64"]), rel_l2(cov.cpu().numpy(), g["cov_f64"])
        print(f"frame export vs the reference's code (f64): pos {e_pos:.2e} cov {e_cov:.2e}; the reference's own f32 run: "
              f"pos {rel_l2(g['pos_f32'], g['pos_f64']):.2e} cov {rel_l2(g['cov_f32'], g['cov_f64']):.2e}")
        assert e_pos < 1e-6 and e_cov < 2e-6
