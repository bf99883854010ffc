"""ctypes front-end of oracle/mpm_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this
module, and only as the checker.  pixie_amd/ never does.

`OracleMPM` mirrors the call surface of the reference's `MPM_Simulator_WARP`
(/root/reference/third_party/PhysGaussian/mpm_solver_warp/mpm_solver_warp.py:47-1210)
closely enough that a parity test drives the oracle and the HIP solver with the same
script.  Pinned to the reference's own solver code by tests/test_mpm_ref_golden.py (see mpm_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# material name -> id, mpm_solver_warp.py:10-26 (visplas/fluid are excluded from the name map)
NAME_TO_MATERIAL_ID = {"jelly": 0, "metal": 1, "sand": 2, "snow": 5, "stationary": 6, "elastic": 0, "rigid": 6}


def build(force: bool = False) -> None:
    """Compile both precisions of the C oracle with gcc (seconds)."""
    outs = [os.path.join(_HERE, "build", f"libmpm_oracle_{p}.so") for p in ("f32", "f64", "f32_omp", "f64_omp")]
    src = os.path.join(_HERE, "mpm_oracle.c")
    fresh = all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs)
    if force or not fresh:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])


def _load(precision: str):
    build()
    lib = C.CDLL(os.path.join(_HERE, "build", f"libmpm_oracle_{precision}.so"))
    dp = C.POINTER(C.c_double)
    lib.mpm_create.restype = C.c_void_p
    lib.mpm_create.argtypes = [C.c_int, C.c_int, C.c_double]
    lib.mpm_destroy.argtypes = [C.c_void_p]
    lib.mpm_field.restype = C.c_void_p
    lib.mpm_field.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_int)]
    lib.mpm_set_scalar.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.mpm_get_time.restype = C.c_double
    lib.mpm_get_time.argtypes = [C.c_void_p]
    lib.mpm_get_oob.restype = C.c_long
    lib.mpm_get_oob.argtypes = [C.c_void_p]
    for name in ("mpm_update_mass", "mpm_finalize_mu_lam", "mpm_compute_bulk", "mpm_zero_grid", "mpm_grid_damping"):
        getattr(lib, name).argtypes = [C.c_void_p]
    for name in ("mpm_pre_p2g", "mpm_compute_stress", "mpm_p2g", "mpm_grid_update", "mpm_apply_bcs", "mpm_g2p", "mpm_p2g2p"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_double]
    lib.mpm_run.argtypes = [C.c_void_p, C.c_double, C.c_int]
    lib.mpm_apply_additional_params.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double, C.c_double, C.c_int]
    lib.mpm_add_surface_collider.argtypes = [C.c_void_p, dp, dp, C.c_int, C.c_double, C.c_double, C.c_double]
    lib.mpm_set_velocity_on_cuboid.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_double, C.c_int]
    lib.mpm_add_bounding_box.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.mpm_add_impulse.argtypes = [C.c_void_p, dp, C.c_double, dp, dp, C.c_int, C.c_double]
    lib.mpm_enforce_translation.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_double]
    lib.mpm_enforce_rotation.argtypes = [C.c_void_p, dp, dp, dp, dp] + [C.c_double] * 6
    lib.mpm_compute_cov.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mpm_compute_R.argtypes = [C.c_void_p, C.c_void_p]
    lib.oracle_svd3.argtypes = [C.c_void_p] * 4
    return lib


_LIBS: dict = {}


def lib(precision: str = "f32"):
    if precision not in _LIBS:
        _LIBS[precision] = _load(precision)
    return _LIBS[precision]


def _d3(v):
    a = (C.c_double * 3)(*[float(x) for x in v])
    return C.cast(a, C.POINTER(C.c_double))


def svd3(A: np.ndarray, precision: str = "f32"):
    """Warp-convention SVD of one 3x3 (see mpm_oracle.c: svd3)."""
    dt = np.float32 if precision == "f32" else np.float64
    A = np.ascontiguousarray(A, dtype=dt).reshape(9)
    U = np.zeros(9, dt); S = np.zeros(3, dt); V = np.zeros(9, dt)
    lib(precision).oracle_svd3(A.ctypes.data, U.ctypes.data, S.ctypes.data, V.ctypes.data)
    return U.reshape(3, 3), S, V.reshape(3, 3)


class OracleMPM:
    """CPU oracle with the reference solver's method names (mpm_solver_warp.py:47-1210)."""

    def __init__(self, n_particles: int, n_grid: int = 100, grid_lim: float = 1.0, precision: str = "f32"):
        self.precision = precision
        self.dtype = np.float64 if precision.startswith("f64") else np.float32     # "*_omp": the multi-core builds (bench, driver runs)
        self._lib = lib(precision)
        self.n_particles = int(n_particles)
        self.n_grid = int(n_grid)
        self.grid_lim = float(grid_lim)
        self._h = C.c_void_p(self._lib.mpm_create(self.n_particles, self.n_grid, self.grid_lim))
        self.init_cov = None

    def __del__(self):
        try:
            if self._h:
                self._lib.mpm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- raw field views (numpy arrays aliasing the oracle's memory) --
    def field(self, name: str) -> np.ndarray:
        cnt = C.c_long(0); is_int = C.c_int(0)
        ptr = self._lib.mpm_field(self._h, name.encode(), C.byref(cnt), C.byref(is_int))
        if not ptr:
            raise KeyError(name)
        ctype = C.c_int if is_int.value else (C.c_double if self.precision.startswith("f64") else C.c_float)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt.value,))
        n = self.n_particles
        if name in ("x", "v"):
            return arr.reshape(n, 3)
        if name in ("F", "F_trial", "C", "stress"):
            return arr.reshape(n, 3, 3)
        if name == "grid_m":
            return arr.reshape(self.n_grid, self.n_grid, self.n_grid)
        if name in ("grid_v_in", "grid_v_out"):
            return arr.reshape(self.n_grid, self.n_grid, self.n_grid, 3)
        return arr

    @property
    def time(self) -> float:
        return self._lib.mpm_get_time(self._h)

    @property
    def out_of_bounds(self) -> int:
        return self._lib.mpm_get_oob(self._h)

    # -- load_initial_data_from_torch, mpm_solver_warp.py:234-281 --
    def load_initial_data(self, x, volume, cov=None):
        self.field("x")[:] = np.asarray(x, dtype=self.dtype)
        self.field("vol")[:] = np.asarray(volume, dtype=self.dtype)
        if cov is not None:
            self.init_cov = np.ascontiguousarray(np.asarray(cov, dtype=self.dtype).reshape(-1))

    # -- set_parameters_dict, mpm_solver_warp.py:287-463 --
    def set_parameters_dict(self, kwargs: dict):
        if "material" in kwargs:
            mid = NAME_TO_MATERIAL_ID.get(kwargs["material"], -1)
            if mid == -1:
                raise TypeError("Undefined material type")
            self.material = mid
        else:
            self.material = getattr(self, "material", 0)
        self.field("material")[:] = self.material
        for key, fld in (("E", "E"), ("nu", "nu"), ("bulk_modulus", "bulk"), ("yield_stress", "yield_stress")):
            if key in kwargs:
                self.field(fld)[:] = np.float32(kwargs[key])   # the reference fills a float32 array (warp_utils.py:222-230)
        for key in ("hardening", "xi", "friction_angle", "rpic_damping", "plastic_viscosity", "softening", "grid_v_damping_scale",
                    "alpha"):
            if key in kwargs:
                self._lib.mpm_set_scalar(self._h, key.encode(), float(kwargs[key]))
        if "g" in kwargs:
            for ax, nm in enumerate(("gx", "gy", "gz")):
                self._lib.mpm_set_scalar(self._h, nm.encode(), float(kwargs["g"][ax]))
        if "density" in kwargs:
            self.field("density")[:] = np.float32(kwargs["density"])
            self._lib.mpm_update_mass(self._h)
        if "additional_material_params" in kwargs:
            for prm in kwargs["additional_material_params"]:
                mat = prm["material"]
                if isinstance(mat, str):
                    mat = NAME_TO_MATERIAL_ID.get(mat, -1)
                self._lib.mpm_apply_additional_params(self._h, _d3(prm["point"]), _d3(prm["size"]), float(prm["E"]),
                                                      float(prm["nu"]), float(prm["density"]), int(mat))
            self._lib.mpm_update_mass(self._h)

    def set_per_particle(self, E=None, nu=None, density=None, material=None, yield_stress=None):
        """Direct per-particle assignment (what material_field.py:343-363 intends)."""
        if E is not None: self.field("E")[:] = E
        if nu is not None: self.field("nu")[:] = nu
        if material is not None: self.field("material")[:] = material
        if yield_stress is not None: self.field("yield_stress")[:] = yield_stress
        if density is not None:
            self.field("density")[:] = density
            self._lib.mpm_update_mass(self._h)

    def finalize_mu_lam(self):
        self._lib.mpm_finalize_mu_lam(self._h)

    def finalize_mu_lam_bulk(self):
        self._lib.mpm_finalize_mu_lam(self._h)
        self._lib.mpm_compute_bulk(self._h)

    # -- boundary conditions / modifiers --
    def add_surface_collider(self, point, normal, surface="sticky", friction=0.0, start_time=0.0, end_time=999.0):
        if surface == "sticky" and friction != 0:
            raise ValueError("friction must be 0 on sticky surfaces.")
        st = {"sticky": 0, "slip": 1, "cut": 11}.get(surface, 2)
        self._lib.mpm_add_surface_collider(self._h, _d3(point), _d3(normal), st, float(friction), float(start_time), float(end_time))

    def set_velocity_on_cuboid(self, point, size, velocity, start_time=0.0, end_time=999.0, reset=0):
        self._lib.mpm_set_velocity_on_cuboid(self._h, _d3(point), _d3(size), _d3(velocity), float(start_time), float(end_time), int(reset))

    def add_bounding_box(self, start_time=0.0, end_time=999.0):
        self._lib.mpm_add_bounding_box(self._h, float(start_time), float(end_time))

    def add_impulse_on_particles(self, force, dt, point=(1, 1, 1), size=(1, 1, 1), num_dt=1, start_time=0.0):
        self._lib.mpm_add_impulse(self._h, _d3(force), float(dt), _d3(point), _d3(size), int(num_dt), float(start_time))

    def enforce_particle_velocity_translation(self, point, size, velocity, start_time, end_time):
        self._lib.mpm_enforce_translation(self._h, _d3(point), _d3(size), _d3(velocity), float(start_time), float(end_time))

    def enforce_particle_velocity_rotation(self, point, normal, half_height_and_radius, rotation_scale, translation_scale, start_time, end_time):
        normal, h1, h2 = rotation_axes(normal)
        self._lib.mpm_enforce_rotation(self._h, _d3(point), _d3(normal), _d3(h1), _d3(h2), float(half_height_and_radius[0]),
                                       float(half_height_and_radius[1]), float(rotation_scale), float(translation_scale),
                                       float(start_time), float(end_time))

    def release_particles_sequentially(self, normal, start_position, end_position, num_layers, start_time, end_time):
        """mpm_solver_warp.py:1185-1210 (host logic only: 50 nested velocity pins with staggered end times; the
        reference overrides the caller's num_layers with 50)."""
        num_layers = 50
        point, size, axis = [0, 0, 0], [0, 0, 0], -1
        for i in range(3):
            if normal[i] == 0:
                point[i] = 1
                size[i] = 1
            else:
                axis = i
                point[i] = end_position
        half_length_portion = abs(start_position - end_position) / num_layers
        end_time_portion = end_time / num_layers
        for i in range(num_layers):
            size[axis] = half_length_portion * (num_layers - i)
            self.enforce_particle_velocity_translation(point=point, size=size, velocity=[0, 0, 0], start_time=start_time,
                                                       end_time=end_time_portion * (i + 1))

    # -- stepping --
    def p2g2p(self, step, dt):
        self._lib.mpm_p2g2p(self._h, float(dt))

    def run(self, dt, n_substeps):
        self._lib.mpm_run(self._h, float(dt), int(n_substeps))

    def phase(self, name: str, dt: float = 0.0):
        """Run one kernel of the substep by name (for per-kernel parity tests)."""
        if name in ("zero_grid", "grid_damping"):
            getattr(self._lib, "mpm_" + name)(self._h)
        else:
            getattr(self._lib, "mpm_" + name)(self._h, float(dt))

    # -- exports --
    def export_cov(self):
        cov = np.zeros(self.n_particles * 6, self.dtype)
        self._lib.mpm_compute_cov(self._h, self.init_cov.ctypes.data, cov.ctypes.data)
        return cov

    def export_R(self):
        Rm = np.zeros((self.n_particles, 9), self.dtype)
        self._lib.mpm_compute_R(self._h, Rm.ctypes.data)
        return Rm


def rotation_axes(normal):
    """Host-side axis set-up of enforce_particle_velocity_rotation, mpm_solver_warp.py:1092-1117
    (float32 vector maths, as wp.vec3 does)."""
    n = np.asarray(normal, dtype=np.float64)
    n = (n * (1.0 / np.sqrt(float(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)))).astype(np.float32)
    h1 = np.array([1.0, 1.0, 1.0], np.float32)
    if abs(float(np.dot(n, h1))) < 0.01:
        h1 = np.array([0.72, 0.37, -0.67], np.float32)
    h1 = h1 - np.float32(np.dot(h1, n)) * n
    h1 = h1 * np.float32(1.0 / np.linalg.norm(h1))
    h2 = np.cross(h1, n).astype(np.float32)
    return n, h1, h2
