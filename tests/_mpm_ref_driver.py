"""Drives ONE scene description through any of three solvers with the same script (test infrastructure):

  * the REFERENCE's `MPM_Simulator_WARP` itself (third_party/PhysGaussian/mpm_solver_warp/mpm_solver_warp.py, imported
    unmodified on the numpy interpreter of the Warp API in tests/golden/wp_shim) -- only in the build container, by
    tests/golden/make_mpm_ref_golden.py, which commits what it computes as tests/golden/mpm_ref_*.npz;
  * `oracle.mpm_oracle.OracleMPM` (the C restatement) -- tests/test_mpm_ref_golden.py holds it to the fixture;
  * `pixie_amd.mpm_solver.MPM_Simulator_WARP` (the HIP product) -- tests/test_mpm_hip.py, same fixture.

A scene is a dict: n_grid, grid_lim, dt, params (set_parameters_dict kwargs), calls (a list of [method, kwargs] with the
reference's method names, in the driver's registration order), bulk (finalize_mu_lam_bulk instead of finalize_mu_lam),
and arrays: x0, vol, cov, optional per-particle E / nu / density / material / yield_stress (assigned the way
gs_simulation.py:528 assigns E: a new array into the model struct) and optional initial v0 / C0 / Ft0.
"""
import numpy as np

STATE_FIELDS = ("x", "v", "C", "F_trial", "F", "stress", "yield_stress", "mu", "lam")


class Adapter:
    """What the three solvers do differently: construction, per-particle assignment, reading state."""

    def __init__(self, solver):
        self.s = solver

    def call(self, method, kwargs):
        getattr(self.s, method)(**kwargs)

    def step(self, dt, n):
        for k in range(n):
            self.s.p2g2p(k, dt)


class ReferenceAdapter(Adapter):
    """The reference class on the Warp interpreter (device strings are ignored there)."""

    def __init__(self, module, scene, arrays):
        import torch
        import warp as wp
        self.wp, self.torch, self.mod = wp, torch, module
        s = module.MPM_Simulator_WARP(10)                                   # gs_simulation.py:483
        s.load_initial_data_from_torch(torch.from_numpy(arrays["x0"]), torch.from_numpy(arrays["vol"]),
                                       torch.from_numpy(arrays["cov"]), n_grid=scene["n_grid"], grid_lim=scene["grid_lim"],
                                       device="cpu")                         # :484-487
        super().__init__(s)

    def set_parameters(self, params):
        self.s.set_parameters_dict(dict(params), device="cpu")

    def call(self, method, kwargs):
        kwargs = dict(kwargs)
        if method in ("add_impulse_on_particles", "enforce_particle_velocity_translation", "enforce_particle_velocity_rotation"):
            kwargs["device"] = "cpu"
        getattr(self.s, method)(**kwargs)

    def assign(self, name, arr):
        t, wp, s = self.torch.from_numpy(np.ascontiguousarray(arr)), self.wp, self.s
        if name in ("E", "nu", "yield_stress"):
            setattr(s.mpm_model, name, wp.from_torch(t))                     # gs_simulation.py:528
        elif name == "density":
            s.reset_densities_and_update_masses(t, device="cpu")             # mpm_solver_warp.py:639-656
        elif name == "material":
            s.mpm_state.particle_material = wp.from_torch(t, dtype=int)
        elif name == "v":
            s.import_particle_v_from_torch(t, device="cpu")
        elif name == "C":
            s.import_particle_C_from_torch(t, device="cpu")
        elif name == "F_trial":
            s.mpm_state.particle_F_trial = self.mod.torch2warp_mat33(t.reshape(-1, 3, 3).contiguous(), dvc="cpu")
        else:
            raise KeyError(name)

    def finalize(self, bulk):
        (self.s.finalize_mu_lam_bulk if bulk else self.s.finalize_mu_lam)(device="cpu")

    def step(self, dt, n):
        for k in range(n):
            self.s.p2g2p(k, dt, device="cpu")

    def read(self, name):
        st, md = self.s.mpm_state, self.s.mpm_model
        src = {"x": st.particle_x, "v": st.particle_v, "C": st.particle_C, "F_trial": st.particle_F_trial, "F": st.particle_F,
               "stress": st.particle_stress, "yield_stress": md.yield_stress, "mu": md.mu, "lam": md.lam, "mass": st.particle_mass,
               "grid_m": st.grid_m, "grid_v_in": st.grid_v_in, "grid_v_out": st.grid_v_out, "material": st.particle_material}[name]
        return np.array(src.numpy(), dtype=np.float64)

    def exports(self):
        cov = self.s.export_particle_cov_to_torch(device="cpu").numpy().astype(np.float64).reshape(-1, 6)
        R = self.s.export_particle_R_to_torch(device="cpu").numpy().astype(np.float64).reshape(-1, 9)
        return cov, R

    @property
    def time(self):
        return float(self.s.time)


class OracleAdapter(Adapter):
    def __init__(self, scene, arrays, precision="f64"):
        from oracle.mpm_oracle import OracleMPM
        s = OracleMPM(arrays["x0"].shape[0], scene["n_grid"], scene["grid_lim"], precision)
        s.load_initial_data(arrays["x0"], arrays["vol"], arrays["cov"])
        super().__init__(s)

    def set_parameters(self, params):
        self.s.set_parameters_dict(dict(params))

    def assign(self, name, arr):
        if name in ("E", "nu", "yield_stress", "material", "density"):
            self.s.set_per_particle(**{name: arr})
        else:
            self.s.field(name)[:] = np.asarray(arr).reshape(self.s.field(name).shape)

    def finalize(self, bulk):
        (self.s.finalize_mu_lam_bulk if bulk else self.s.finalize_mu_lam)()

    def read(self, name):
        return np.array(self.s.field(name), dtype=np.float64)

    def exports(self):
        return (np.array(self.s.export_cov(), np.float64).reshape(-1, 6), np.array(self.s.export_R(), np.float64).reshape(-1, 9))

    @property
    def time(self):
        return float(self.s.time)


class ProductAdapter(Adapter):
    """pixie_amd.mpm_solver.MPM_Simulator_WARP -- the reference's own signatures, so the calls equal ReferenceAdapter's."""

    def __init__(self, scene, arrays, scatter_bits=None):
        import torch
        from pixie_amd.mpm_solver import MPM_Simulator_WARP
        self.torch = torch
        s = MPM_Simulator_WARP(10)
        s.load_initial_data_from_torch(torch.from_numpy(arrays["x0"]).cuda(), torch.from_numpy(arrays["vol"]).cuda(),
                                       torch.from_numpy(arrays["cov"]).cuda(), n_grid=scene["n_grid"], grid_lim=scene["grid_lim"])
        if scatter_bits is not None:
            s._set_scalar("scatter_bits", scatter_bits)
        super().__init__(s)

    def set_parameters(self, params):
        self.s.set_parameters_dict(dict(params))

    def assign(self, name, arr):
        t, s = self.torch.from_numpy(np.ascontiguousarray(arr)).cuda(), self.s
        if name in ("E", "nu", "yield_stress"):
            setattr(s.mpm_model, name, t)
        elif name == "density":
            s.reset_densities_and_update_masses(t)
        elif name == "material":
            s.mpm_state.particle_material = t
        elif name == "v":
            s.import_particle_v_from_torch(t)
        elif name == "C":
            s.import_particle_C_from_torch(t)
        elif name == "F_trial":
            s.set_field("F_trial", t.reshape(-1, 9))
        else:
            raise KeyError(name)

    def finalize(self, bulk):
        (self.s.finalize_mu_lam_bulk if bulk else self.s.finalize_mu_lam)()

    def read(self, name):
        out = self.s.get_field(name).cpu().numpy().astype(np.float64)
        if name in ("C", "F_trial", "F", "stress"):
            out = out.reshape(-1, 3, 3)
        return out

    def exports(self):
        cov = self.s.export_particle_cov_to_torch().cpu().numpy().astype(np.float64).reshape(-1, 6)
        R = self.s.export_particle_R_to_torch().cpu().numpy().astype(np.float64).reshape(-1, 9)
        return cov, R

    @property
    def time(self):
        return float(self.s.time)


def set_up(adapter, scene, arrays):
    """The set-up order of gs_simulation.py:483-531: parameters, JSON boundary conditions, per-particle material field,
    finalize_mu_lam; then the initial state the scene prescribes."""
    adapter.set_parameters(scene["params"])
    for method, kwargs in scene["calls"]:
        adapter.call(method, kwargs)
    for name in ("E", "nu", "yield_stress", "material", "density"):
        if name in arrays:
            adapter.assign(name, arrays[name])
    adapter.finalize(scene.get("bulk", False))
    for name, key in (("v", "v0"), ("C", "C0"), ("F_trial", "Ft0")):
        if key in arrays:
            adapter.assign(name, arrays[key])


def run(adapter, scene, arrays, on_checkpoint):
    set_up(adapter, scene, arrays)
    done = 0
    for cp in scene["checkpoints"]:
        adapter.step(scene["dt"], cp - done)
        done = cp
        on_checkpoint(cp, {f: adapter.read(f) for f in STATE_FIELDS})


def load_fixture(path):
    """-> {scene name: (scene dict, arrays dict, results dict)} from a tests/golden/mpm_ref_*.npz."""
    import json
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    out = {}
    for name, scene in meta.items():
        pre = name + "/"
        arrays, results = {}, {}
        for key in z.files:
            if not key.startswith(pre):
                continue
            sub = key[len(pre):]
            (arrays if sub.startswith("in/") else results)[sub[3:] if sub.startswith("in/") else sub] = z[key]
        out[name] = (scene, arrays, results)
    return out
