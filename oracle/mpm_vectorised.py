"""Second, INDEPENDENT CPU restatement of the reference's MLS-MPM substep -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module, and only as the checker.  pixie_amd/ never does.

Why it exists.  Until round 4 nothing upstream could check oracle/mpm_oracle.c (the reference ships no vectors for the
solver and NVIDIA Warp cannot run here), so this file -- a second restatement written directly from the reference source in
a different form (array programming on torch CPU tensors, one batched expression per reference statement, LAPACK SVD
instead of a hand-written Jacobi), deliberately sharing no code with mpm_oracle.c -- was the anchor: tests/test_mpm_vectorised.py
requires the two to agree in float64 on every material model, every boundary condition and every particle modifier.  Since
round 4 the C oracle is PINNED to the reference's own solver code (tests/golden/wp_shim runs mpm_solver_warp.py unmodified;
tests/test_mpm_ref_golden.py); this file stays as an independent cross-check.  It used to be a multi-core CPU baseline
of bench.py (now: the C oracle under OpenMP).

Restated from (paths relative to /root/reference/third_party/PhysGaussian/mpm_solver_warp):
  mpm_utils.py:10-17    kirchoff_stress_FCR            mpm_utils.py:89-135   von_mises_return_mapping
  mpm_utils.py:20-28    kirchoff_stress_water          mpm_utils.py:138-191  ..._with_damage
  mpm_utils.py:52-68    kirchoff_stress_StVK           mpm_utils.py:195-239  viscoplasticity_return_mapping_with_StVK
  mpm_utils.py:71-86    kirchoff_stress_drucker_prager mpm_utils.py:242-279  sand_return_mapping
  mpm_utils.py:282-300  mu/lam, bulk, zero_grid        mpm_utils.py:303-312  compute_dweight
  mpm_utils.py:338-394  p2g_apic_with_stress           mpm_utils.py:398-409  grid_normalization_and_gravity
  mpm_utils.py:412-463  g2p                            mpm_utils.py:467-526  compute_stress_from_F_trial
  mpm_utils.py:583-588  add_damping_via_grid           mpm_utils.py:591-663  apply_additional_params, selections
  mpm_solver_warp.py:514-637  p2g2p (launch order)     mpm_solver_warp.py:785-840, 874-905, 917-974  BC kernels
  mpm_solver_warp.py:1015-1027, 1061-1073, 1137-1179   particle modifiers
wp.svd3 (warp-lang 0.10.1, not in /root/reference): U, V proper rotations, singular values ordered by magnitude, only
the last may be negative -- obtained here by canonicalising LAPACK's SVD.
Conventions carried over: wp.mat33(a, b, c) stores its vector arguments as COLUMNS (w[d, i] = weight of offset i along
axis d); wp.int truncates toward zero; `time`, `dt`, every struct scalar and every literal are float32 inside the
kernels (the float64 mode keeps those float32-rounded VALUES and does the arithmetic in double).
"""
from __future__ import annotations

import math

import numpy as np
import torch

NAME_TO_MATERIAL_ID = {"jelly": 0, "metal": 1, "sand": 2, "snow": 5, "stationary": 6, "elastic": 0, "rigid": 6}  # mpm_solver_warp.py:10-26


def _f32(x) -> float:
    """A Python float as the kernels see it (struct members and kernel arguments are 32-bit)."""
    return float(np.float32(x))


def svd3_warp(F: torch.Tensor):
    """Batched wp.svd3: F = U diag(s) V^T with det U = det V = +1, |s0| >= |s1| >= |s2|, only s2 may be negative."""
    U, s, Vh = torch.linalg.svd(F)
    V = Vh.transpose(-1, -2)
    flipU = torch.linalg.det(U) < 0
    flipV = torch.linalg.det(V) < 0
    U = U.clone(); V = V.clone(); s = s.clone()
    U[flipU, :, 2] = -U[flipU, :, 2]
    V[flipV, :, 2] = -V[flipV, :, 2]
    s[flipU ^ flipV, 2] = -s[flipU ^ flipV, 2]
    return U, s, V


def _udvt(U, d, V):
    return (U * d[:, None, :]) @ V.transpose(-1, -2)


class VectorisedMPM:
    """Call surface of MPM_Simulator_WARP (mpm_solver_warp.py:47-1210), so pixie_amd.synthetic.apply_scene drives it."""

    def __init__(self, n_particles: int, n_grid: int = 100, grid_lim: float = 1.0, precision: str = "f64"):
        self.dt_ = torch.float64 if precision == "f64" else torch.float32
        self.precision = precision
        n = self.n_particles = int(n_particles)
        self.n_grid, self.grid_lim = int(n_grid), float(grid_lim)
        self.dx, self.inv_dx = _f32(grid_lim / n_grid), _f32(float(n_grid / grid_lim))   # :62-66
        z = lambda *shape: torch.zeros(shape, dtype=self.dt_)
        self.x, self.v = z(n, 3), z(n, 3)
        self.F, self.C, self.stress = z(n, 3, 3), z(n, 3, 3), z(n, 3, 3)
        self.F_trial = torch.eye(3, dtype=self.dt_).repeat(n, 1, 1)                     # :272-277
        self.vol, self.mass, self.density = z(n), z(n), z(n)
        self.E, self.nu, self.mu, self.lam, self.bulk, self.yield_stress = z(n), z(n), z(n), z(n), z(n), z(n)
        self.material = torch.zeros(n, dtype=torch.int64)
        self.selection = torch.zeros(n, dtype=torch.int64)
        G = self.n_grid
        self.grid_m, self.grid_v_in, self.grid_v_out = z(G, G, G), z(G, G, G, 3), z(G, G, G, 3)
        # MPMModelStruct scalars, defaults of initialize() :74-92
        self.g = [0.0, 0.0, 0.0]
        self.rpic_damping, self.grid_v_damping_scale = 0.0, 1.1
        self.hardening, self.xi, self.softening, self.plastic_viscosity = 0.0, 0.0, 0.1, 0.0
        sin_phi = math.sin(25.0 / 180.0 * 3.14159265)
        self.alpha = _f32(math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi))
        self.time = 0.0
        self.mat_default = 0
        self.impulses, self.v_modifiers, self.colliders = [], [], []
        self.out_of_bounds = 0
        # node coordinates float(grid_x) * model.dx, as the BC kernels compute them
        idx = torch.arange(G, dtype=self.dt_)
        self._node = (idx * self.dx).to(self.dt_) if precision == "f64" else (idx.float() * np.float32(self.dx))

    # ------------------------------------------------------------------ set-up (mpm_solver_warp.py:234-471)
    def load_initial_data(self, x, volume, cov=None):
        self.x[:] = torch.as_tensor(np.asarray(x), dtype=self.dt_)
        self.vol[:] = torch.as_tensor(np.asarray(volume), dtype=self.dt_)
        self.init_cov = None if cov is None else torch.as_tensor(np.asarray(cov), dtype=self.dt_).reshape(-1, 6)

    def set_parameters_dict(self, kw: dict):
        if "material" in kw:
            mid = NAME_TO_MATERIAL_ID.get(kw["material"], -1)
            if mid == -1:
                raise TypeError("Undefined material type")
            self.mat_default = mid
        self.material[:] = self.mat_default                                              # :345-354
        for key, name in (("E", "E"), ("nu", "nu"), ("bulk_modulus", "bulk"), ("yield_stress", "yield_stress")):
            if key in kw:
                getattr(self, name)[:] = _f32(kw[key])
        for key in ("hardening", "xi", "rpic_damping", "plastic_viscosity", "softening", "grid_v_damping_scale"):
            if key in kw:
                setattr(self, key, _f32(kw[key]))
        if "friction_angle" in kw:                                                       # :390-393
            sin_phi = math.sin(kw["friction_angle"] / 180.0 * 3.14159265)
            self.alpha = _f32(math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi))
        if "g" in kw:
            self.g = [_f32(c) for c in kw["g"]]
        if "density" in kw:
            self.density[:] = _f32(kw["density"])
            self.mass[:] = self.density * self.vol
        for prm in kw.get("additional_material_params", []):                             # mpm_utils.py:591-610
            mat = prm["material"]
            mat = NAME_TO_MATERIAL_ID.get(mat, -1) if isinstance(mat, str) else int(mat)
            pt = torch.tensor([_f32(c) for c in prm["point"]], dtype=self.dt_)
            sz = torch.tensor([_f32(c) for c in prm["size"]], dtype=self.dt_)
            inside = ((self.x > pt - sz) & (self.x < pt + sz)).all(dim=1)
            self.E[inside] = _f32(prm["E"]); self.nu[inside] = _f32(prm["nu"])
            self.density[inside] = _f32(prm["density"]); self.material[inside] = mat
        if "additional_material_params" in kw:
            self.mass[:] = self.density * self.vol

    def set_per_particle(self, E=None, nu=None, density=None, material=None, yield_stress=None):
        for name, val in (("E", E), ("nu", nu), ("yield_stress", yield_stress)):
            if val is not None:
                getattr(self, name)[:] = torch.as_tensor(np.asarray(val), dtype=self.dt_)
        if material is not None:
            self.material[:] = torch.as_tensor(np.asarray(material), dtype=torch.int64)
        if density is not None:
            self.density[:] = torch.as_tensor(np.asarray(density), dtype=self.dt_)
            self.mass[:] = self.density * self.vol

    def finalize_mu_lam(self):                                                           # mpm_utils.py:282-288
        self.mu[:] = self.E / (2.0 * (1.0 + self.nu))
        self.lam[:] = self.E * self.nu / ((1.0 + self.nu) * (1.0 - 2.0 * self.nu))

    def finalize_mu_lam_bulk(self):                                                      # :290-293
        self.finalize_mu_lam()
        self.bulk[:] = self.lam + 2.0 / 3.0 * self.mu

    # ------------------------------------------------------------------ boundary conditions / modifiers
    def _vec(self, v):
        return torch.tensor([_f32(c) for c in v], dtype=self.dt_)

    def add_surface_collider(self, point, normal, surface="sticky", friction=0.0, start_time=0.0, end_time=999.0):
        if surface == "sticky" and friction != 0:
            raise ValueError("friction must be 0 on sticky surfaces.")
        scale = 1.0 / math.sqrt(float(sum(c ** 2 for c in normal)))
        self.colliders.append(dict(kind="surface", point=self._vec(point), normal=self._vec([scale * c for c in normal]),
                                   surface_type={"sticky": 0, "slip": 1, "cut": 11}.get(surface, 2),
                                   start=_f32(start_time), end=_f32(end_time)))

    def set_velocity_on_cuboid(self, point, size, velocity, start_time=0.0, end_time=999.0, reset=0):
        self.colliders.append(dict(kind="cuboid", point=self._vec(point), size=self._vec(size), velocity=self._vec(velocity),
                                   start=_f32(start_time), end=_f32(end_time), reset=int(reset)))

    def add_bounding_box(self, start_time=0.0, end_time=999.0):
        self.colliders.append(dict(kind="bbox", start=_f32(start_time), end=_f32(end_time)))

    def _box_mask(self, point, size):                                                    # mpm_utils.py:613-642
        off = self.x - self._vec(point)
        return (off.abs() < self._vec(size)).all(dim=1)

    def add_impulse_on_particles(self, force, dt, point=(1, 1, 1), size=(1, 1, 1), num_dt=1, start_time=0.0):
        self.impulses.append(dict(force=self._vec(force), mask=self._box_mask(point, size), start=_f32(start_time),
                                  end=_f32(start_time + dt * num_dt)))

    def enforce_particle_velocity_translation(self, point, size, velocity, start_time, end_time):
        self.v_modifiers.append(dict(kind="translation", velocity=self._vec(velocity), mask=self._box_mask(point, size),
                                     start=_f32(start_time), end=_f32(end_time)))

    def enforce_particle_velocity_rotation(self, point, normal, half_height_and_radius, rotation_scale, translation_scale,
                                           start_time, end_time):
        # axis set-up :1092-1117 is wp.vec3 (float32) arithmetic on the host
        n = np.asarray(normal, np.float64)
        n = (n * (1.0 / math.sqrt(float(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)))).astype(np.float32)
        h1 = np.ones(3, np.float32)
        if abs(float(n @ h1)) < 0.01:
            h1 = np.array([0.72, 0.37, -0.67], np.float32)
        h1 = h1 - np.float32(h1 @ n) * n
        h1 = h1 * np.float32(1.0 / np.sqrt(np.float32(h1 @ h1)))
        h2 = np.cross(h1, n).astype(np.float32)
        n_t, h1_t, h2_t = (torch.as_tensor(a.astype(np.float64), dtype=self.dt_) for a in (n, h1, h2))
        off = self.x - self._vec(point)                                                  # mpm_utils.py:645-663
        along = off @ n_t
        radial = torch.linalg.norm(off - along[:, None] * n_t, dim=1)
        mask = (along.abs() < _f32(half_height_and_radius[0])) & (radial < _f32(half_height_and_radius[1]))
        self.v_modifiers.append(dict(kind="rotation", point=self._vec(point), normal=n_t, h1=h1_t, h2=h2_t, mask=mask,
                                     rot=_f32(rotation_scale), trans=_f32(translation_scale), start=_f32(start_time),
                                     end=_f32(end_time)))

    # ------------------------------------------------------------------ the substep (mpm_solver_warp.py:514-637)
    def p2g2p(self, step, dt):
        dt_host = float(dt)          # `self.time = self.time + dt` is Python-float arithmetic on the host (:637)
        dt = _f32(dt)
        t = _f32(self.time)
        self.grid_m.zero_(); self.grid_v_in.zero_(); self.grid_v_out.zero_()             # zero_grid
        self._pre_p2g(t, dt)
        self._compute_stress(dt)
        self._p2g(dt)
        self._grid_update(t, dt, dt_host)
        self._g2p(dt)
        self.time = self.time + dt_host

    def run(self, dt, n_substeps):
        for i in range(int(n_substeps)):
            self.p2g2p(i, dt)

    def _pre_p2g(self, t, dt):
        for imp in self.impulses:                                                        # apply_force :1015-1027
            if t >= imp["start"] and t < imp["end"]:
                m = imp["mask"]
                self.v[m] = self.v[m] + (imp["force"][None, :] / self.mass[m, None]) * dt
        for mod in self.v_modifiers:
            if not (t >= mod["start"] and t < mod["end"]):
                continue
            m = mod["mask"]
            if mod["kind"] == "translation":                                             # :1061-1073
                self.v[m] = mod["velocity"]
            else:                                                                        # :1137-1179
                off = self.x[m] - mod["point"]
                n, h1, h2 = mod["normal"], mod["h1"], mod["h2"]
                hd = torch.linalg.norm(off - (off @ n)[:, None] * n, dim=1)
                theta = torch.acos((off @ h1) / hd)
                theta = torch.where((off @ h2) > 0, theta, -theta)
                a1 = -hd * torch.sin(theta) * mod["rot"]
                a2 = hd * torch.cos(theta) * mod["rot"]
                self.v[m] = a1[:, None] * h1 + a2[:, None] * h2 + mod["trans"] * n

    # -- return mappings: each takes the rows `idx` of F_trial and returns the elastic F for them
    def _rm_von_mises(self, idx, damage: bool):
        Ft = self.F_trial[idx]
        U, s_old, V = svd3_warp(Ft)
        sig = s_old.clamp_min(_f32(0.01))
        eps = sig.log()
        mu, lam, ys = self.mu[idx], self.lam[idx], self.yield_stress[idx]
        tr = eps.sum(1)
        tau = 2.0 * mu[:, None] * eps + (lam * tr)[:, None]
        cond = tau - tau.sum(1, keepdim=True) / 3.0
        yielding = torch.linalg.norm(cond, dim=1) > ys
        if damage:
            yielding = yielding & ~(ys <= 0)                                             # "return F_trial" :158-159
        eps_hat = eps - (tr / 3.0)[:, None]
        eps_hat_norm = torch.linalg.norm(eps_hat, dim=1) + _f32(1e-6)
        dgamma = eps_hat_norm - ys / (2.0 * mu)
        shift = (dgamma / eps_hat_norm)[:, None] * eps_hat
        Fe = _udvt(U, (eps - shift).exp(), V)
        new_ys, new_mu, new_lam = ys.clone(), mu.clone(), lam.clone()
        if damage:
            new_ys = torch.where(yielding, ys - self.softening * torch.linalg.norm(shift, dim=1), ys)
            broken = yielding & (new_ys <= 0)
            new_mu = torch.where(broken, torch.zeros_like(mu), mu)
            new_lam = torch.where(broken, torch.zeros_like(lam), lam)
        if self.hardening == 1:
            new_ys = torch.where(yielding, new_ys + 2.0 * new_mu * self.xi * dgamma, new_ys)
        self.yield_stress[idx], self.mu[idx], self.lam[idx] = new_ys, new_mu, new_lam
        return torch.where(yielding[:, None, None], Fe, Ft)

    def _rm_visco(self, idx, dt):
        Ft = self.F_trial[idx]
        U, s_old, V = svd3_warp(Ft)
        sig = s_old.clamp_min(_f32(0.01))
        b = sig * sig
        eps = sig.log()
        tr = eps.sum(1)
        eps_hat = eps - (tr / 3.0)[:, None]
        mu = self.mu[idx]
        s_trial = 2.0 * mu[:, None] * eps_hat
        s_norm = torch.linalg.norm(s_trial, dim=1)
        y = s_norm - math.sqrt(2.0 / 3.0) * self.yield_stress[idx]
        mu_hat = mu * b.sum(1) / 3.0
        s_new_norm = s_norm - y / (1.0 + self.plastic_viscosity / (2.0 * mu_hat * dt))
        s_new = (s_new_norm / s_norm)[:, None] * s_trial
        eps_new = 1.0 / (2.0 * mu[:, None]) * s_new + (tr / 3.0)[:, None]
        return torch.where((y > 0)[:, None, None], _udvt(U, eps_new.exp(), V), Ft)

    def _rm_sand(self, idx):
        Ft = self.F_trial[idx]
        U, sig, V = svd3_warp(Ft)
        eps = sig.abs().clamp_min(_f32(1e-14)).log()
        tr = eps.sum(1)
        eps_hat = eps - (tr / 3.0)[:, None]
        eps_hat_norm = torch.linalg.norm(eps_hat, dim=1)
        mu, lam = self.mu[idx], self.lam[idx]
        dgamma = eps_hat_norm + (3.0 * lam + 2.0 * mu) / (2.0 * mu) * tr * self.alpha
        out = Ft.clone()                                                                  # dgamma <= 0
        tip = (dgamma > 0) & (tr > 0)
        out[tip] = (U @ V.transpose(-1, -2))[tip]
        proj = (dgamma > 0) & (tr <= 0)
        H = eps - eps_hat * (dgamma / eps_hat_norm)[:, None]
        out[proj] = _udvt(U, H.exp(), V)[proj]
        return out

    def _compute_stress(self, dt):
        """compute_stress_from_F_trial, mpm_utils.py:467-526"""
        live = self.selection == 0
        F = self.F.clone()
        elastic = live.clone()
        for mat, fn in ((1, lambda i: self._rm_von_mises(i, False)), (2, self._rm_sand), (3, lambda i: self._rm_visco(i, dt)),
                        (5, lambda i: self._rm_von_mises(i, True))):
            idx = torch.nonzero(live & (self.material == mat)).squeeze(1)
            if idx.numel():
                F[idx] = fn(idx)
            elastic &= self.material != mat
        F[elastic] = self.F_trial[elastic]
        self.F = torch.where(live[:, None, None], F, self.F)
        F = self.F
        J = torch.linalg.det(F)
        U, sig, V = svd3_warp(F)
        mu, lam = self.mu, self.lam
        eye = torch.eye(3, dtype=self.dt_)
        stress = torch.zeros_like(F)
        Ftr = F.transpose(-1, -2)
        m = self.material
        fcr = (2.0 * mu)[:, None, None] * (F - U @ V.transpose(-1, -2)) @ Ftr + eye * (lam * J * (J - 1.0))[:, None, None]
        stress = torch.where(((m == 0) | (m == 5))[:, None, None], fcr, stress)
        sg = sig.clamp_min(_f32(0.01))
        e = sg.log()
        stvk = _udvt(U, 2.0 * mu[:, None] * e + (lam * e.sum(1))[:, None], V) @ Ftr
        stress = torch.where(((m == 1) | (m == 3))[:, None, None], stvk, stress)
        ls = sig.log()
        centre = 2.0 * mu[:, None] * ls * (1.0 / sig) + (lam * ls.sum(1))[:, None] * (1.0 / sig)
        dp = _udvt(U, centre, V) @ Ftr
        stress = torch.where((m == 2)[:, None, None], dp, stress)
        pressure = -self.bulk * (J.pow(-_f32(1.1)) - 1.0)
        water = eye * (J * pressure)[:, None, None]
        stress = torch.where((m == 6)[:, None, None], water, stress)
        stress = (stress + stress.transpose(-1, -2)) / 2.0
        self.stress = torch.where(live[:, None, None], stress, self.stress)

    def _stencil(self):
        gp = self.x * self.inv_dx
        base = (gp - 0.5).to(torch.int64)                                                 # wp.int: toward zero
        fx = gp - base.to(self.dt_)
        wa, wb, wc = 1.5 - fx, fx - 1.0, fx - 0.5
        w = torch.stack([wa * wa * 0.5, 0.75 - wb * wb, wc * wc * 0.5], dim=2)            # w[p, axis, offset]
        dw = torch.stack([fx - 1.5, -2.0 * (fx - 1.0), fx - 0.5], dim=2)
        return base, fx, w, dw

    def _inside(self, base):
        ok = ((base >= 0) & (base + 2 < self.n_grid)).all(dim=1)
        return ok

    def _p2g(self, dt):
        """p2g_apic_with_stress, mpm_utils.py:338-394"""
        base, fx, w, dw = self._stencil()
        live = (self.selection == 0) & self._inside(base)
        self.out_of_bounds += int(((self.selection == 0) & ~self._inside(base)).sum())
        C = self.C
        C = (1.0 - self.rpic_damping) * C + self.rpic_damping / 2.0 * (C - C.transpose(-1, -2))
        if self.rpic_damping < _f32(-0.001):
            C = torch.zeros_like(C)
        G = self.n_grid
        gm, gv = self.grid_m.view(-1), self.grid_v_in.view(-1, 3)
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    dpos = (torch.tensor([i, j, k], dtype=self.dt_) - fx) * self.dx
                    weight = w[:, 0, i] * w[:, 1, j] * w[:, 2, k]
                    dweight = torch.stack([dw[:, 0, i] * w[:, 1, j] * w[:, 2, k], w[:, 0, i] * dw[:, 1, j] * w[:, 2, k],
                                           w[:, 0, i] * w[:, 1, j] * dw[:, 2, k]], dim=1) * self.inv_dx
                    force = -self.vol[:, None] * (self.stress @ dweight[:, :, None]).squeeze(2)
                    add = (weight * self.mass)[:, None] * (self.v + (C @ dpos[:, :, None]).squeeze(2)) + dt * force
                    node = ((base[:, 0] + i) * G + (base[:, 1] + j)) * G + (base[:, 2] + k)
                    gv.index_add_(0, node[live], add[live])
                    gm.index_add_(0, node[live], (weight * self.mass)[live])

    def _grid_update(self, t, dt, dt_host):
        """grid_normalization_and_gravity (mpm_utils.py:398-409), add_damping_via_grid (:583-588), BC kernels"""
        has = self.grid_m > _f32(1e-15)
        g = torch.tensor(self.g, dtype=self.dt_)
        v = self.grid_v_in * (1.0 / torch.where(has, self.grid_m, torch.ones_like(self.grid_m)))[..., None] + dt * g
        self.grid_v_out = torch.where(has[..., None], v, self.grid_v_out)
        if self.grid_v_damping_scale < 1.0:                                               # mpm_solver_warp.py:595
            self.grid_v_out = self.grid_v_out * self.grid_v_damping_scale
        G = self.n_grid
        X, Y, Z = torch.meshgrid(self._node, self._node, self._node, indexing="ij")
        I = torch.arange(G)
        IX, IY, IZ = torch.meshgrid(I, I, I, indexing="ij")
        for c in self.colliders:
            vo = self.grid_v_out
            if c["kind"] == "surface":                                                    # :785-840
                if not (t >= c["start"] and t < c["end"]):
                    continue
                p, n = c["point"], c["normal"]
                below = ((X - p[0]) * n[0] + (Y - p[1]) * n[1] + (Z - p[2]) * n[2]) < 0.0
                if c["surface_type"] == 11:
                    outside = (Z < _f32(0.4)) | (Z > _f32(0.53))
                    cut = torch.stack([vo[..., 0], torch.zeros_like(Z), vo[..., 2]], dim=-1) * _f32(0.3)
                    new = torch.where(outside[..., None], torch.zeros_like(vo), cut)
                else:
                    # sticky -- and slip / frictional too: the reference computes the projected velocity and then
                    # stores zero (:821-840)
                    new = torch.zeros_like(vo)
                self.grid_v_out = torch.where(below[..., None], new, vo)
            elif c["kind"] == "cuboid":                                                   # :874-905
                if t >= c["start"] and t < c["end"]:
                    p, s = c["point"], c["size"]
                    inside = ((X - p[0]).abs() < s[0]) & ((Y - p[1]).abs() < s[1]) & ((Z - p[2]).abs() < s[2])
                    self.grid_v_out = torch.where(inside[..., None], c["velocity"].expand_as(vo), vo)
                elif c["reset"] == 1 and np.float32(t) < np.float32(c["end"]) + np.float32(15.0) * np.float32(dt):
                    self.grid_v_out = torch.zeros_like(vo)
                # host `modify` :899-905 (Python-float arithmetic, stored back into a float32 vec3)
                if self.time >= c["start"] and self.time < c["end"]:
                    c["point"] = self._vec([float(c["point"][d]) + dt_host * float(c["velocity"][d]) for d in range(3)])
            else:                                                                         # bounding box :917-974
                if not (t >= c["start"] and t < c["end"]):
                    continue
                pad = 3
                vo = vo.clone()
                for axis, idx in enumerate((IX, IY, IZ)):
                    comp = vo[..., axis]
                    comp[(idx < pad) & (comp < 0)] = 0.0
                    comp[(idx >= G - pad) & (comp > 0)] = 0.0
                self.grid_v_out = vo

    def _g2p(self, dt):
        """g2p, mpm_utils.py:412-463"""
        base, fx, w, dw = self._stencil()
        live = (self.selection == 0) & self._inside(base)
        G = self.n_grid
        gv = self.grid_v_out.view(-1, 3)
        new_v = torch.zeros_like(self.v)
        new_C = torch.zeros_like(self.C)
        new_F = torch.zeros_like(self.C)
        cb = base.clamp(0, G - 3)
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    dpos = torch.tensor([i, j, k], dtype=self.dt_) - fx
                    weight = w[:, 0, i] * w[:, 1, j] * w[:, 2, k]
                    node = ((cb[:, 0] + i) * G + (cb[:, 1] + j)) * G + (cb[:, 2] + k)
                    gvel = gv[node]
                    new_v = new_v + gvel * weight[:, None]
                    new_C = new_C + gvel[:, :, None] * dpos[:, None, :] * (weight * self.inv_dx * 4.0)[:, None, None]
                    dweight = torch.stack([dw[:, 0, i] * w[:, 1, j] * w[:, 2, k], w[:, 0, i] * dw[:, 1, j] * w[:, 2, k],
                                           w[:, 0, i] * w[:, 1, j] * dw[:, 2, k]], dim=1) * self.inv_dx
                    new_F = new_F + gvel[:, :, None] * dweight[:, None, :]
        eye = torch.eye(3, dtype=self.dt_)
        Ft = (eye + new_F * dt) @ self.F
        l1, l3 = live[:, None], live[:, None, None]
        self.x = torch.where(l1, self.x + dt * new_v, self.x)
        self.v = torch.where(l1, new_v, self.v)
        self.C = torch.where(l3, new_C, self.C)
        self.F_trial = torch.where(l3, Ft, self.F_trial)

    # ------------------------------------------------------------------ read-out
    def field(self, name: str) -> np.ndarray:
        t = {"x": self.x, "v": self.v, "F": self.F, "F_trial": self.F_trial, "C": self.C, "stress": self.stress, "vol": self.vol,
             "mass": self.mass, "density": self.density, "E": self.E, "nu": self.nu, "mu": self.mu, "lam": self.lam,
             "bulk": self.bulk, "yield_stress": self.yield_stress, "material": self.material, "selection": self.selection,
             "grid_m": self.grid_m, "grid_v_in": self.grid_v_in, "grid_v_out": self.grid_v_out}[name]
        return t.numpy()
