"""Drop-in for the reference's MPM solver class on MI355X.

Mirrors `MPM_Simulator_WARP` and the module-level helpers of
third_party/PhysGaussian/mpm_solver_warp/mpm_solver_warp.py (lines cited per method): same
method names, argument meaning, defaults and error behaviour, so that gs_simulation.py and
material_field.py can drive it unchanged (INTEGRATION.md lists the two places that passed
Warp arrays and now pass torch tensors).  All compute is in libpixie_hip.so (csrc/mpm.hip);
this file only marshals arguments.  There is no CPU path.

Differences a caller can observe, by design:
  * export_*_to_torch return one persistent tensor per field in the caller's particle order, refreshed in
    place by each export call, instead of live zero-copy aliases of solver memory (the solver keeps
    particles block-sorted in SoA rows); get_field() returns fresh tensors;
  * `run(dt, n)` / `p2g2p_n` step n substeps in one call (two launches per substep -- block kernel + grid kernel --, no host sync);
  * `p2g2p(step, dt)` only QUEUES a substep.  The queue is flushed -- as ONE `run(dt, n)` -- by the next call that
    observes or changes the solver (any export / get_field / set_field / `.time` / BC or parameter call, `flush()`),
    so the reference's own loop (gs_simulation.py:633-634: `for step in range(step_per_frame): p2g2p(frame, dt)`,
    then the per-frame export) runs the fused step loop unmodified and gives bit-identical results to `run()`.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import BCDesc, PModDesc, check, d3

# mpm_solver_warp.py:10-26
MATERIAL_ID_TO_NAME = {0: "jelly", 1: "metal", 2: "sand", 3: "visplas", 4: "fluid", 5: "snow", 6: "stationary"}
EXCLUDED_MATERIAL_NAMES = ["visplas", "fluid"]
NAME_TO_MATERIAL_ID = {name: i for i, name in MATERIAL_ID_TO_NAME.items() if name not in EXCLUDED_MATERIAL_NAMES}
NAME_TO_MATERIAL_ID.update({"elastic": 0, "rigid": 6})


# The reference prints its progress lines (mpm_solver_warp.py:280-281, :290-292) and so does the shim; a caller that creates
# solvers in a loop (bench.py) switches them off with `pixie_amd.mpm_solver.VERBOSE = False`.
VERBOSE = True


def _say(*args):
    if VERBOSE:
        print(*args)


def get_material_name(material_id):
    """mpm_solver_warp.py:29-39 -- despite its name the reference maps NAME -> id (or -1)."""
    return NAME_TO_MATERIAL_ID.get(material_id, -1)


def get_material_id(material_name):
    """mpm_solver_warp.py:41-45"""
    return NAME_TO_MATERIAL_ID.get(material_name, -1)


_FLOAT_FIELDS = {"x": 3, "v": 3, "F": 9, "F_trial": 9, "C": 9, "stress": 9, "vol": 1, "mass": 1, "density": 1,
                 "E": 1, "nu": 1, "mu": 1, "lam": 1, "bulk": 1, "yield_stress": 1, "init_cov": 6, "cov": 6}
_INT_FIELDS = {"material": 1, "selection": 1}
_PERSISTENT = ["x", "v", "F", "F_trial", "C", "vol", "mass", "density", "E", "nu", "mu", "lam", "bulk",
               "yield_stress", "material", "selection", "init_cov"]


class _ArrayView:
    """What `solver.mpm_state.particle_x` returns: supports .numpy() like a Warp array."""

    def __init__(self, solver, name):
        self._solver, self._name = solver, name

    def torch(self):
        return self._solver.get_field(self._name)

    def numpy(self):
        return self.torch().cpu().numpy()

    def __len__(self):
        return self._solver.n_particles


class _StructView:
    """Attribute proxy standing in for MPMStateStruct / MPMModelStruct (warp_utils.py:6-74)."""

    def __init__(self, solver, prefix, scalars=()):
        object.__setattr__(self, "_solver", solver)
        object.__setattr__(self, "_prefix", prefix)
        object.__setattr__(self, "_scalars", dict(scalars))

    def _field(self, attr):
        name = attr[len(self._prefix):] if self._prefix and attr.startswith(self._prefix) else attr
        if name in _FLOAT_FIELDS or name in _INT_FIELDS or name in ("grid_m", "grid_v_in", "grid_v_out"):
            return name
        return None

    def __getattr__(self, attr):
        if attr in self._scalars:
            return self._scalars[attr]()
        name = self._field(attr)
        if name is None:
            raise AttributeError(attr)
        return _ArrayView(self._solver, name)

    def __setattr__(self, attr, value):
        name = self._field(attr)
        if name is None:
            self._solver._set_model_scalar(attr, value)
            return
        if isinstance(value, _ArrayView):
            value = value.torch()
        self._solver.set_field(name, value)


class MPM_Simulator_WARP:
    """mpm_solver_warp.py:47-1210"""

    def __init__(self, n_particles, n_grid=100, grid_lim=1.0, device="cuda:0", *, diag=False):
        """`diag=True` (tests and profilers only, no reference counterpart): the handle lives in libpixie_hip_diag.so, the
        -DPIXIE_DIAG build of the same sources, which adds phase() and kernel_times()."""
        self._h = None
        self._diag = bool(diag)
        self.initialize(n_particles, n_grid, grid_lim, device=device)
        self.time_profile = {}

    def _check(self, rc, what=""):
        check(rc, what, lib=self._L)

    # ------------------------------------------------------------------ lifetime
    def initialize(self, n_particles, n_grid=100, grid_lim=1.0, device="cuda:0"):
        """:52-180"""
        lib = self._L = _lib.load(diag=getattr(self, "_diag", False))
        if not torch.cuda.is_available():
            raise _lib.PixieHipError("MPM_Simulator_WARP needs a HIP device; pixie_amd has no CPU fallback")
        self._release()
        self._pending, self._pending_dt = 0, 0.0   # substeps queued by p2g2p() and their dt
        self._pending_stream = None                # the stream that was current when they were queued
        self._lost_reported = 0
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.n_particles = int(n_particles)
        self.n_grid = int(n_grid)
        self.grid_lim = float(grid_lim)
        h = C.c_void_p()
        self._check(lib.pixie_mpm_create(C.byref(h), self.n_particles, self.n_grid, self.grid_lim), "pixie_mpm_create")
        self._h = h
        self._material = 0
        self._gravity = [0.0, 0.0, 0.0]
        self.update_cov_with_F = False
        self.mpm_state = _StructView(self, "particle_")
        self.mpm_model = _StructView(self, "", scalars={
            "n_grid": lambda: self.n_grid, "grid_lim": lambda: self.grid_lim,
            "dx": lambda: self._get_scalar("dx"), "inv_dx": lambda: self._get_scalar("inv_dx"),
            "material": lambda: self._material, "n_particles": lambda: self.n_particles,
            "grid_v_damping_scale": lambda: self._get_scalar("grid_v_damping_scale"),
            "rpic_damping": lambda: self._get_scalar("rpic_damping"),
            "alpha": lambda: self._get_scalar("alpha"),
            "gravitational_accelaration": lambda: list(self._gravity),
            "update_cov_with_F": lambda: False,
        })

    def _release(self):
        self._pending = 0   # queued substeps of a solver that is going away are dropped
        if getattr(self, "_h", None):
            self._L.pixie_mpm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    @property
    def _stream(self):
        return _lib.current_stream_ptr()

    def _raw_stream(self) -> int:
        """torch's current stream on this solver's device as an integer handle (0 = the default stream), the cheap way: the
        deferred p2g2p() asks once per substep."""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index) or 0
        except AttributeError:
            return _lib.current_stream_ptr().value or 0

    @property
    def time(self):
        return self._get_scalar("time")

    @time.setter
    def time(self, value):
        self.flush()
        self._check(self._L.pixie_mpm_set_scalar(self._h, b"time", float(value)), "set time")

    def flush(self):
        """Enqueue the substeps queued by p2g2p() on the current stream (asynchronous, like run())."""
        n = getattr(self, "_pending", 0)
        if n:
            self._pending = 0
            queued = getattr(self, "_pending_stream", None)     # the stream the substeps were queued under (an integer handle)
            self._pending_stream = None
            self._check(self._L.pixie_mpm_step(self._h, self._pending_dt, n, self._stream if queued is None else C.c_void_p(queued)), "pixie_mpm_step")
            # The substeps reach their stream only NOW.  A caller synchronised for the reference's eager semantics
            # (`current.wait_stream(side)` right after its p2g2p() loop) waited on a stream that was still empty, so whatever
            # observes or continues the solver on the CURRENT stream has to be ordered after them here.
            if queued is not None and queued != self._raw_stream():
                qs = torch.cuda.ExternalStream(queued, device=self._dev_index) if queued else torch.cuda.default_stream(self._dev_index)
                ev = torch.cuda.Event()
                ev.record(qs)
                torch.cuda.current_stream(self._dev_index).wait_event(ev)
            self._warn_if_particles_lost()

    def _get_scalar(self, key):
        self.flush()
        out = C.c_double(0.0)
        self._check(self._L.pixie_mpm_get_scalar(self._h, key.encode(), C.byref(out)), f"get_scalar({key})")
        return out.value

    def _set_scalar(self, key, value):
        self.flush()
        self._check(self._L.pixie_mpm_set_scalar(self._h, key.encode(), float(value)), f"set_scalar({key})")

    def _set_model_scalar(self, attr, value):
        if attr == "gravitational_accelaration":
            self._gravity = [float(value[0]), float(value[1]), float(value[2])]
            for ax, nm in enumerate(("gx", "gy", "gz")):
                self._set_scalar(nm, self._gravity[ax])
        elif attr == "material":
            self._material = int(value)
        elif attr in ("rpic_damping", "grid_v_damping_scale", "hardening", "xi", "softening", "plastic_viscosity",
                      "friction_angle"):
            self._set_scalar(attr, value)
        else:
            raise AttributeError(f"cannot set mpm_model.{attr}")

    def _as_device_tensor(self, value, dtype):
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value))
        elif not torch.is_tensor(value):
            value = torch.as_tensor(value)
        return value.detach().to(device=self.device, dtype=dtype).contiguous()

    def set_field(self, name, value):
        dtype = torch.int32 if name in _INT_FIELDS else torch.float32
        t = self._as_device_tensor(value, dtype)
        self.flush()
        self._check(self._L.pixie_mpm_set_field(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.numel(), self._stream),
              f"set_field({name})")

    def get_field(self, name, out=None):
        n, g = self.n_particles, self.n_grid
        if out is not None:
            pass
        elif name in _INT_FIELDS:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        elif name in _FLOAT_FIELDS:
            k = _FLOAT_FIELDS[name]
            out = torch.empty((n, k) if k > 1 else (n,), dtype=torch.float32, device=self.device)
        elif name == "grid_m":
            out = torch.empty((g, g, g), dtype=torch.float32, device=self.device)
        elif name in ("grid_v_in", "grid_v_out"):
            out = torch.empty((g, g, g, 3), dtype=torch.float32, device=self.device)
        else:
            raise KeyError(name)
        self.flush()
        self._check(self._L.pixie_mpm_get_field(self._h, name.encode(), C.c_void_p(out.data_ptr()), out.numel(), self._stream),
              f"get_field({name})")
        return out

    def _fill(self, name, value):
        self.flush()
        self._check(self._L.pixie_mpm_fill_field(self._h, name.encode(), float(value), self._stream), f"fill({name})")

    def _update_mass(self):
        self.flush()
        self._check(self._L.pixie_mpm_update_mass(self._h, self._stream), "update_mass")

    # ------------------------------------------------------------------ initial data
    def load_initial_data_from_torch(self, tensor_x, tensor_volume, tensor_cov=None, n_grid=100, grid_lim=1.0,
                                     device="cuda:0"):
        """:234-281"""
        self.dim, n = tensor_x.shape[1], tensor_x.shape[0]
        assert tensor_x.shape[0] == tensor_volume.shape[0]
        self.initialize(n, n_grid, grid_lim, device=device)
        self.import_particle_x_from_torch(tensor_x, device=device)
        self.set_field("vol", tensor_volume)
        if tensor_cov is not None:
            self.set_field("init_cov", tensor_cov.reshape(-1))
        # v = 0 and F_trial = I are the create() defaults (:262-277)
        _say("Particles initialized from torch data.")
        _say("Total particles: ", self.n_particles)

    def set_parameters(self, device="cuda:0", **kwargs):
        """:284-285"""
        self.set_parameters_dict(kwargs, device)

    def set_parameters_dict(self, kwargs={}, device="cuda:0"):
        """:287-463"""
        if "material" in kwargs:
            _say("Setting material to ", kwargs["material"])
            self._material = get_material_name(kwargs["material"])
            _say("Material ID: ", self._material)
            if self._material == -1:
                raise TypeError("Undefined material type")
        new_lim = kwargs.get("grid_lim", self.grid_lim)
        new_ng = kwargs.get("n_grid", self.n_grid)
        if float(new_lim) != self.grid_lim or int(new_ng) != self.n_grid:
            self._regrid(int(new_ng), float(new_lim))
        self._fill("material", self._material)  # :345-354
        if "E" in kwargs:
            self._fill("E", kwargs["E"])
        if "nu" in kwargs:
            self._fill("nu", kwargs["nu"])
        if "bulk_modulus" in kwargs:
            self._fill("bulk", kwargs["bulk_modulus"])
        if "yield_stress" in kwargs:
            self._fill("yield_stress", kwargs["yield_stress"])
        for key in ("hardening", "xi", "friction_angle"):
            if key in kwargs:
                self._set_scalar(key, kwargs[key])
        if "g" in kwargs:
            self._set_model_scalar("gravitational_accelaration", kwargs["g"])
        if "spawn_offset" in kwargs:  # :400-406
            off = kwargs["spawn_offset"]
            pos = self.export_particle_x_to_torch()
            pos[:, 0] += off[0]; pos[:, 1] += off[1]; pos[:, 2] += off[2]
            self.import_particle_x_from_torch(pos)
        if "density" in kwargs:
            self._fill("density", kwargs["density"])
            self._update_mass()
        for key in ("rpic_damping", "plastic_viscosity", "softening", "grid_v_damping_scale"):
            if key in kwargs:
                self._set_scalar(key, kwargs[key])
        if "additional_material_params" in kwargs:  # :435-463
            lib = self._L
            plist = kwargs["additional_material_params"]
            for params in plist:
                if isinstance(params["material"], str):
                    params["material"] = get_material_name(params["material"])
            self.flush()
            if len(plist) > 4:
                # material_field.py:343-363 sends one 1 mm box PER PARTICLE (N launches of N threads in the reference): one
                # launch here, with the result of applying the boxes in list order (the last box containing a particle wins)
                boxes = np.array([[*p["point"], *p["size"]] for p in plist], dtype=np.float32).reshape(-1, 6)
                vals = np.array([[p["E"], p["nu"], p["density"]] for p in plist], dtype=np.float32).reshape(-1, 3)
                mats = np.array([int(p["material"]) for p in plist], dtype=np.int32)
                tb, tv, tm = (torch.from_numpy(a).to(self.device) for a in (boxes, vals, mats))
                self._check(lib.pixie_mpm_apply_additional_params_batch(self._h, len(plist), C.c_void_p(tb.data_ptr()), C.c_void_p(tv.data_ptr()),
                                                                  C.c_void_p(tm.data_ptr()), self._stream), "apply_additional_params_batch")
                torch.cuda.current_stream().synchronize()      # the three staging tensors die with this frame
            else:
                for params in plist:
                    self._check(lib.pixie_mpm_apply_additional_params(self._h, d3(params["point"]), d3(params["size"]),
                                                                float(params["E"]), float(params["nu"]),
                                                                float(params["density"]), int(params["material"]),
                                                                self._stream), "apply_additional_params")
            self._update_mass()

    def _regrid(self, n_grid, grid_lim):
        """set_parameters_dict may change n_grid / grid_lim after the particles were loaded (:315-342): like the
        reference, only the grid arrays are re-made and dx / inv_dx recomputed; particle fields, model scalars,
        boundary conditions, particle modifiers and the time survive (pixie_mpm_regrid, in place)."""
        self.flush()
        self._check(self._L.pixie_mpm_regrid(self._h, int(n_grid), float(grid_lim), self._stream), "pixie_mpm_regrid")
        self.n_grid, self.grid_lim = int(n_grid), float(grid_lim)

    def set_per_particle(self, E=None, nu=None, density=None, material=None, yield_stress=None):
        """Per-particle material assignment -- what material_field.py:343-363 does with N
        apply_additional_params launches, as plain array uploads."""
        if E is not None: self.set_field("E", E)
        if nu is not None: self.set_field("nu", nu)
        if material is not None: self.set_field("material", material)
        if yield_stress is not None: self.set_field("yield_stress", yield_stress)
        if density is not None:
            self.set_field("density", density)
            self._update_mass()

    def finalize_mu_lam(self, device="cuda:0"):
        """:465-471"""
        self.flush()
        self._check(self._L.pixie_mpm_finalize_mu_lam(self._h, 0, self._stream), "finalize_mu_lam")

    def finalize_mu_lam_bulk(self, device="cuda:0"):
        """:505-511"""
        self.flush()
        self._check(self._L.pixie_mpm_finalize_mu_lam(self._h, 1, self._stream), "finalize_mu_lam_bulk")

    def reset_densities_and_update_masses(self, all_particle_densities, device="cuda:0"):
        """:640-656"""
        self.set_field("density", all_particle_densities)
        self._update_mass()

    # ------------------------------------------------------------------ stepping
    def p2g2p(self, step, dt, device="cuda:0"):
        """:514-637 -- one substep.  Deferred: the substep is queued and runs, fused with its neighbours, when the
        solver is next observed or changed (module docstring); a change of dt flushes what was queued first."""
        dt = float(dt)
        stream = self._raw_stream()
        # a change of dt -- or of the CURRENT STREAM (a caller that wraps part of its loop in torch.cuda.stream(s)
        # must get its substeps on the stream they were issued under, not on whatever is current at flush time) -- ends a batch
        if self._pending and (dt != self._pending_dt or stream != self._pending_stream):
            self.flush()
        self._pending_dt = dt
        self._pending_stream = stream
        self._pending += 1
        if self.live_exports:
            self._refresh_views()

    def run(self, dt, n_substeps):
        """n substeps of p2g2p in one call (fused G2P->P2G->grid launches, no host synchronisation)."""
        dt = float(dt)
        stream = self._raw_stream()
        if self._pending and (dt != self._pending_dt or stream != self._pending_stream):
            self.flush()
        self._pending_dt = dt
        self._pending_stream = stream
        self._pending += int(n_substeps)
        self.flush()
        if self.live_exports:
            self._refresh_views()

    @property
    def scatter_bits(self):
        """Accumulation mode of the P2G scatter IN FORCE (csrc/mpm.hip).  32: pairs of 32-bit fixed-point sums per LDS atomic
        -- half the atomics, quantum 2^-30 of the summed contribution bounds of a 256-particle work item, the noise level of
        the reference's fp32 atomics; 64: exact 64-bit fixed point (quantum 2^-42), 20 % slower.  Both are integer sums:
        order-independent and bit-reproducible.
        Default (`scatter_bits = 0`): chosen at every re-binning from the particle-mass contrast max(m) / min(m) -- packed up
        to a contrast of 32, exact above.  The limit of the packed mode: a node fed only by particles R times lighter than
        their tile-mates is resolved to ~1e-6 R of its own mass, which is invisible for uniform scenes and visible from
        R ~ 1e2 (tests/test_mpm_hip.py::test_mass_contrast_selects_the_exact_scatter).  Assign 32 or 64 to force a mode."""
        return int(self._get_scalar("scatter_bits"))

    @scatter_bits.setter
    def scatter_bits(self, bits):
        self._set_scalar("scatter_bits", int(bits))

    def _warn_if_particles_lost(self):
        """Mass leaving the simulation must not be silent: the count read back at the last re-binning (no sync)."""
        out = C.c_double(0.0)
        self._check(self._L.pixie_mpm_get_scalar(self._h, b"lost_particles_seen", C.byref(out)), "get_scalar")
        lost = int(out.value)
        if lost > getattr(self, "_lost_reported", 0):
            import warnings
            warnings.warn(f"MPM_Simulator_WARP: {lost} particle(s) left the {self.n_grid}^3 grid (or every active block) and "
                          "were frozen / dropped from P2G; the reference writes out of bounds here", RuntimeWarning, stacklevel=3)
            self._lost_reported = lost

    p2g2p_n = run

    def phase(self, phase, dt):
        """Test hook (diag=True solvers only): 0 = modifiers+stress+P2G, 1 = grid update+damping+BCs, 2 = G2P."""
        self._need_diag("phase()")
        self.flush()
        self._check(self._L.pixie_mpm_phase(self._h, int(phase), float(dt), self._stream), "pixie_mpm_phase")

    @property
    def out_of_bounds(self):
        cnt = C.c_int64(0)
        self.flush()
        self._check(self._L.pixie_mpm_out_of_bounds(self._h, C.byref(cnt), self._stream), "out_of_bounds")
        return cnt.value

    def _need_diag(self, what):
        if not self._diag:
            raise _lib.PixieHipError(f"{what} is a diagnostic of the PIXIE_DIAG build: create the solver with MPM_Simulator_WARP(..., diag=True)")

    def set_profile(self, on=True):
        self._need_diag("set_profile()")
        self._set_scalar("profile", 1.0 if on else 0.0)

    def kernel_times(self):
        """(mean fused-particle-kernel ms, mean grid-kernel ms, launches) since the last call."""
        self._need_diag("kernel_times()")
        a, b, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        self.flush()
        self._check(self._L.pixie_mpm_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(n)), "kernel_times")
        return a.value, b.value, n.value

    # ------------------------------------------------------------------ import / export (:659-741)
    def import_particle_x_from_torch(self, tensor_x, clone=True, device="cuda:0"):
        if tensor_x is not None:
            self.set_field("x", tensor_x)

    def import_particle_v_from_torch(self, tensor_v, clone=True, device="cuda:0"):
        if tensor_v is not None:
            self.set_field("v", tensor_v)

    def import_particle_F_from_torch(self, tensor_F, clone=True, device="cuda:0"):
        if tensor_F is not None:
            self.set_field("F", torch.reshape(tensor_F, (-1, 9)))

    def import_particle_C_from_torch(self, tensor_C, clone=True, device="cuda:0"):
        if tensor_C is not None:
            self.set_field("C", torch.reshape(tensor_C, (-1, 9)))

    def _export_view(self, key, shape, dtype=torch.float32):
        """The reference's export_*_to_torch return zero-copy aliases of the Warp arrays (wp.to_torch, :659-741): no
        allocation per call and the same storage every time.  The solver keeps particles block-sorted in SoA rows, so a
        live alias in the caller's order cannot exist; the closest equivalent is ONE persistent tensor per field that
        every export call refreshes in place (one gather launch) and returns -- callers that hold on to the tensor see it
        updated by the next export call, never a new allocation per rendered frame (gs_simulation.py:591-600)."""
        views = self.__dict__.setdefault("_views", {})
        t = views.get(key)
        if t is None or t.device != self.device or tuple(t.shape) != tuple(shape):
            t = views[key] = torch.empty(shape, dtype=dtype, device=self.device)
        return t

    # Opt-in emulation of the reference's live aliases: with `live_exports = True` every tensor an export_* call has
    # handed out is refreshed after each p2g2p / run, so a caller that holds one across substeps reads current data as
    # it does from wp.to_torch.  It costs what it says -- no deferral, one gather launch per held field per substep --
    # which is why it is off by default: gs_simulation.py re-exports before every use (:591-600, :636-655).
    live_exports = False
    _EXPORTERS = {"x": "export_particle_x_to_torch", "v": "export_particle_v_to_torch", "stress": "export_particle_stress_to_torch",
                  "F": "export_particle_F_to_torch", "C": "export_particle_C_to_torch", "R": "export_particle_R_to_torch",
                  "cov": "export_particle_cov_to_torch"}

    def _refresh_views(self):
        for key in list(self.__dict__.get("_views", {})):
            getattr(self, self._EXPORTERS[key])()

    def export_particle_x_to_torch(self):
        return self.get_field("x", out=self._export_view("x", (self.n_particles, 3)))

    def export_particle_v_to_torch(self):
        return self.get_field("v", out=self._export_view("v", (self.n_particles, 3)))

    def export_particle_stress_to_torch(self):
        return self.get_field("stress", out=self._export_view("stress", (self.n_particles, 9))).reshape(-1, 3, 3)

    def export_particle_F_to_torch(self):
        return self.get_field("F", out=self._export_view("F", (self.n_particles, 9)))

    def export_particle_C_to_torch(self):
        return self.get_field("C", out=self._export_view("C", (self.n_particles, 9)))

    def export_particle_R_to_torch(self, device="cuda:0"):
        out = self._export_view("R", (self.n_particles, 9))
        self.flush()
        self._check(self._L.pixie_mpm_export_R(self._h, C.c_void_p(out.data_ptr()), self._stream), "export_R")
        return out

    def export_particle_cov_to_torch(self, device="cuda:0"):
        out = self._export_view("cov", (self.n_particles * 6,))
        self.flush()
        self._check(self._L.pixie_mpm_export_cov(self._h, C.c_void_p(out.data_ptr()), self._stream), "export_cov")
        return out

    def export_frame_for_rendering(self, gs_num, scale_origin, original_mean_pos, rotation_matrices, z_shift_value=0.0,
                                   with_cov=True):
        """The per-frame hand-off to the rasteriser of gs_simulation.py:591-600 in one launch: returns
        (pos_render (gs_num,3), cov3D_render (gs_num,6)) == (transform_to_original_coordinates(undoshift2center111(
        export_particle_x_to_torch()[:gs_num], z_shift), scale, mean, Rs), apply_inverse_cov_rotations(
        export_particle_cov_to_torch().view(-1,6)[:gs_num] / scale**2, Rs))."""
        M = np.eye(3)
        for R in reversed(list(rotation_matrices)):   # apply_inverse_rotations: p @ R_k, then @ R_{k-1}, ...
            M = M @ np.asarray(R.detach().cpu() if torch.is_tensor(R) else R, dtype=np.float64)
        mean = [float(v) for v in (original_mean_pos.detach().cpu() if torch.is_tensor(original_mean_pos) else original_mean_pos)]
        pos = torch.empty((int(gs_num), 3), dtype=torch.float32, device=self.device)
        cov = torch.empty((int(gs_num), 6), dtype=torch.float32, device=self.device) if with_cov else None
        self.flush()
        self._check(self._L.pixie_mpm_export_frame(self._h, int(gs_num), d3([1.0, 1.0, 1.0 + float(z_shift_value)]), float(scale_origin),
                                                 d3(mean), (C.c_double * 9)(*M.reshape(-1)), C.c_void_p(pos.data_ptr()),
                                                 C.c_void_p(cov.data_ptr()) if with_cov else None, self._stream), "export_frame")
        return pos, cov

    def print_time_profile(self):
        """:743-746"""
        print("MPM Time profile:")
        for key, value in self.time_profile.items():
            print(key, sum(value))

    # ------------------------------------------------------------------ boundary conditions
    def _add_bc(self, **kw):
        bc = BCDesc()
        bc.type = kw["type"]
        bc.surface_type = kw.get("surface_type", 0)
        bc.reset = kw.get("reset", 0)
        for nm in ("point", "size", "velocity", "normal"):
            vals = kw.get(nm, (0.0, 0.0, 0.0))
            for d in range(3):
                getattr(bc, nm)[d] = float(vals[d])
        bc.start_time = float(kw.get("start_time", 0.0))
        bc.end_time = float(kw.get("end_time", 999.0))
        bc.friction = float(kw.get("friction", 0.0))
        self.flush()
        self._check(self._L.pixie_mpm_add_bc(self._h, C.byref(bc)), "pixie_mpm_add_bc")

    def add_surface_collider(self, point, normal, surface="sticky", friction=0.0, start_time=0.0, end_time=999.0):
        """:749-843"""
        point = list(point)
        normal_scale = 1.0 / math.sqrt(float(sum(x ** 2 for x in normal)))
        normal = list(normal_scale * x for x in normal)
        if surface == "sticky" and friction != 0:
            raise ValueError("friction must be 0 on sticky surfaces.")
        surface_type = {"sticky": 0, "slip": 1, "cut": 11}.get(surface, 2)
        self._add_bc(type=0, point=point, normal=normal, surface_type=surface_type, friction=friction,
                     start_time=start_time, end_time=end_time)

    def set_velocity_on_cuboid(self, point, size, velocity, start_time=0.0, end_time=999.0, reset=0):
        """:853-908"""
        self._add_bc(type=1, point=list(point), size=size, velocity=velocity, start_time=start_time,
                     end_time=end_time, reset=reset)

    def add_bounding_box(self, start_time=0.0, end_time=999.0):
        """:910-977"""
        self._add_bc(type=2, start_time=start_time, end_time=end_time)

    def _add_pmod(self, **kw):
        pm = PModDesc()
        pm.type = kw["type"]
        for nm in ("point", "size", "force", "velocity", "normal", "h1", "h2"):
            vals = kw.get(nm, (0.0, 0.0, 0.0))
            for d in range(3):
                getattr(pm, nm)[d] = float(vals[d])
        for nm in ("half_height", "radius", "rotation_scale", "translation_scale", "start_time", "end_time"):
            setattr(pm, nm, float(kw.get(nm, 0.0)))
        self.flush()
        self._check(self._L.pixie_mpm_add_particle_modifier(self._h, C.byref(pm), self._stream), "add_particle_modifier")

    def add_impulse_on_particles(self, force, dt, point=[1, 1, 1], size=[1, 1, 1], num_dt=1, start_time=0.0,
                                 device="cuda:0"):
        """:982-1029"""
        self._add_pmod(type=0, force=force, point=point, size=size, start_time=start_time,
                       end_time=start_time + dt * num_dt)

    def enforce_particle_velocity_translation(self, point, size, velocity, start_time, end_time, device="cuda:0"):
        """:1031-1075"""
        self._add_pmod(type=1, point=point, size=size, velocity=velocity, start_time=start_time, end_time=end_time)

    def enforce_particle_velocity_rotation(self, point, normal, half_height_and_radius, rotation_scale,
                                           translation_scale, start_time, end_time, device="cuda:0"):
        """:1080-1181 (axis set-up :1092-1117 in float32, as wp.vec3 arithmetic)"""
        n = np.asarray(normal, dtype=np.float64)
        n = (n * (1.0 / math.sqrt(float(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)))).astype(np.float32)
        h1 = np.array([1.0, 1.0, 1.0], np.float32)
        if abs(float(np.dot(n, h1))) < 0.01:
            h1 = np.array([0.72, 0.37, -0.67], np.float32)
        h1 = h1 - np.float32(np.dot(h1, n)) * n
        h1 = h1 * np.float32(1.0 / np.linalg.norm(h1))
        h2 = np.cross(h1, n).astype(np.float32)
        self._add_pmod(type=2, point=point, normal=n, h1=h1, h2=h2, half_height=half_height_and_radius[0],
                       radius=half_height_and_radius[1], rotation_scale=rotation_scale,
                       translation_scale=translation_scale, start_time=start_time, end_time=end_time)

    def release_particles_sequentially(self, normal, start_position, end_position, num_layers, start_time, end_time):
        """:1185-1210 -- pins the particles in `num_layers` nested slabs along the axis `normal` names and lets go of the
        outermost slab first: slab k (k = 0 ... layers-1) spans (layers - k) * |start - end| / layers either side of
        `end_position` and is held from `start_time` until (k + 1) * end_time / layers.  The reference overrides the
        caller's `num_layers` with 50; kept."""
        layers = 50
        axis = next((i for i in (2, 1, 0) if normal[i] != 0), -1)   # the reference keeps the LAST non-zero component
        centre = [1.0, 1.0, 1.0]
        half = [1.0, 1.0, 1.0]
        for i in range(3):
            if normal[i] != 0:
                centre[i] = end_position
                half[i] = 0.0
        layer_thickness = abs(start_position - end_position) / layers
        for k in range(layers):
            half[axis] = layer_thickness * (layers - k)
            self.enforce_particle_velocity_translation(point=list(centre), size=list(half), velocity=[0.0, 0.0, 0.0],
                                                       start_time=start_time, end_time=(end_time / layers) * (k + 1))


def run_batch(solvers, dt, n_substeps, streams=None):
    """Advance several INDEPENDENT scenes by `n_substeps` substeps each, concurrently on one GPU (no reference counterpart: the
    reference runs one scene per process, gs_simulation.py:633-634; BASELINE configs[3] is a batch of scenes).

    One scene's substep is two dependent launches -- a VALU-bound block kernel and a latency-bound grid kernel -- so a single
    scene cannot fill the chip at 100 k particles and cannot overlap its own two kernels at 1 M.  Several scenes on their own HIP
    streams can: each solver's step loop is issued from its own host thread (the loop is one foreign call that releases the GIL) on
    its own stream.  Measured (bench.py): three 100 k scenes 2.2x the throughput of one, two 1 M scenes 1.2x.  Results are the
    solvers' own -- bit-identical to running them one after the other (tests/test_mpm_hip.py).

    `streams`: one torch.cuda.Stream per solver (created on first use and kept on the solvers otherwise).  The streams start
    after the work already queued on the current stream, and the current stream waits for all of them before this returns."""
    import threading
    solvers = list(solvers)
    if not solvers:
        return
    dev = solvers[0].device
    if any(s.device != dev for s in solvers):
        raise ValueError("run_batch: the solvers must live on one device (scenes are sharded across GPUs by process, pixie_amd/distributed.py)")
    cur = torch.cuda.current_stream(dev)
    if streams is None:
        streams = []
        for s in solvers:
            if getattr(s, "_batch_stream", None) is None:
                s._batch_stream = torch.cuda.Stream(dev)
            streams.append(s._batch_stream)
    else:
        streams = list(streams)
        if len(streams) != len(solvers):    # (zip would silently skip the solvers without a stream: ADVICE r5)
            raise ValueError(f"run_batch: {len(solvers)} solvers but {len(streams)} streams")
    errors = []

    def work(s, st):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(st):
                s.run(dt, n_substeps)
        except Exception as exc:      # surfaced in the caller's thread below
            errors.append(exc)

    for s, st in zip(solvers, streams):
        s.flush()
        st.wait_stream(cur)
    threads = [threading.Thread(target=work, args=(s, st)) for s, st in zip(solvers, streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for st in streams:
        cur.wait_stream(st)
    if errors:
        raise errors[0]
