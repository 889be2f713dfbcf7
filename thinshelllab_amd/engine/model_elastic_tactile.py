"""Tactile pad FEM body: host-side counterpart of ``Elastic``
(/root/reference/code/engine/model_elastic_tactile.py:12-80, :214-230, :253-326).
One-time work only (rest matrices, lumped masses, surface orientation); the stable Neo-Hookean
energy / force / Hessian (:81-201) run in libtsl_hip.so (csrc/k_fem.hpp, kind 0).
"""
import numpy as np
import torch

from . import readfile
from .field import Field, ScalarField


class Elastic:
    kind = 0

    def __init__(self, dt, offset, ratio):
        self.E = 300000
        self.nu = 0.2
        mu, lam = self.E / (2 * (1 + self.nu)), self.E * self.nu / ((1 + self.nu) * (1 - 2 * self.nu))
        self.mu = ScalarField(mu, self._param("mu")); self.lam = ScalarField(lam, self._param("lam")); self.alpha = ScalarField(1 + mu / lam, self._param("alpha"))
        self.density = 2000.0
        self.dt = dt
        self.offset = offset
        self.gravity = ScalarField([0.0, 0.0, -9.8], self._gravity_written)
        self.ratio = ratio
        self._sys = None
        n_verts, ox = readfile.read_node()
        n_cells, tets = readfile.read_ele()
        self.n_surfaces, faces = readfile.read_smesh()
        self.F_ox_array = np.array(ox, dtype=np.float64)
        self.F_vertices_array = np.array(tets, dtype=np.int32)
        self.f2v_array = np.array(faces, dtype=np.int32)
        self.n_verts = n_verts
        self.n_cells = n_cells
        self.is_surface = np.zeros(n_verts, dtype=np.int32)
        self.count()
        z3 = lambda n: torch.zeros((n, 3), dtype=torch.float64)
        self.F_vertices = Field(torch.as_tensor(self.F_vertices_array.copy()))
        self.F_x = Field(z3(n_verts)); self.F_x_prev = Field(z3(n_verts)); self.F_v = Field(z3(n_verts))
        self.F_ox = Field(torch.as_tensor(self.F_ox_array.copy()))
        self.F_m = Field(torch.zeros(n_verts, dtype=torch.float64))
        self.F_B = Field(torch.zeros((n_cells, 3, 3), dtype=torch.float64))
        self.F_W = Field(torch.zeros(n_cells, dtype=torch.float64))
        self.ext_force = Field(z3(n_verts))
        self.f2v = Field(torch.zeros((self.n_surfaces, 3), dtype=torch.int32))
        self.offset_faces = 0
        self.body_idx = 0

    def _param(self, name):
        """writes after the engine context exists are forwarded (tsl_set_param "elastic<i>.mu|lam|alpha"), like the cloth's Kb / Kl / Ka"""
        def cb(field):
            sys = getattr(self, "_sys", None)
            if sys is not None and sys._ctx is not None:
                sys._ctx.set_param(f"elastic{sys.elastics.index(self)}.{name}", field.value)
        return cb

    def _gravity_written(self, field):
        if self._sys is not None:
            self._sys._refresh_gravity()

    # -- :293-300, :302-321
    def is_bottom_func(self, i):
        return self.F_ox_array[i][2] < 0.001

    def is_inner_circle_func(self, i):
        return np.linalg.norm(self.F_ox_array[i]) < 0.0076

    def is_surf_func(self, i):
        return np.linalg.norm(self.F_ox_array[i]) > 0.0148

    # -- :253-263 (vertex predicates used by the scenes' set_frozen kernels)
    def is_bottom(self, i):
        return bool(self.F_ox_array[i][2] < 0.001 and self.is_surface[i])

    def is_inner_circle(self, i):
        return bool(np.linalg.norm(self.F_ox_array[i]) < 0.0076 and self.is_surface[i])

    def is_surf(self, i):
        return bool(np.linalg.norm(self.F_ox_array[i]) > 0.0148 and self.is_surface[i])

    def bound_mask(self):
        """vertices with is_bottom or is_inner_circle (the frozen / gripper-driven set)."""
        r = np.linalg.norm(self.F_ox_array, axis=1)
        return ((self.F_ox_array[:, 2] < 0.001) | (r < 0.0076)) & (self.is_surface != 0)

    def surf_mask(self):
        r = np.linalg.norm(self.F_ox_array, axis=1)
        return (r > 0.0148) & (self.is_surface != 0) & ~self.bound_mask()

    def count(self):
        self.is_surface[:] = 0
        self.is_surface[self.f2v_array.ravel()] = 1
        self.frozen_cnt = int(self.bound_mask().sum())
        self.surf_point = int(self.surf_mask().sum())

    # -- :214-230 init_pos, :265-291 init_surface_indices
    def init(self, offsetx, offsety, offsetz, flip):
        self._init_args = (offsetx, offsety, offsetz, bool(flip))
        x = self.ratio * self.F_ox_array
        if flip:
            x = -x
        x = x + np.array([offsetx, offsety, offsetz])
        self.F_x.from_numpy(x)
        self.F_v.fill(0)
        t = self.F_vertices_array
        Ds = np.stack([x[t[:, 0]] - x[t[:, 3]], x[t[:, 1]] - x[t[:, 3]], x[t[:, 2]] - x[t[:, 3]]], axis=2)  # columns
        self.F_B.from_numpy(np.linalg.inv(Ds))
        W = np.abs(np.linalg.det(Ds)) / 6
        self.F_W.from_numpy(W)
        m = np.zeros(self.n_verts)
        np.add.at(m, t.ravel(), np.repeat(W / 4 * self.density, 4))
        self.F_m.from_numpy(m)
        f = self.f2v_array.copy()
        p1, p2, p3 = x[f[:, 0]], x[f[:, 1]], x[f[:, 2]]
        n = np.cross(p2 - p1, p3 - p1)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        inner = np.array([offsetx, offsety, offsetz + (-0.002 if flip else 0.002) * self.ratio])
        inward = np.einsum("ij,ij->i", n, inner - p1) > 0
        r = np.linalg.norm(self.F_ox_array, axis=1)
        ic = (r < 0.0076) & (self.is_surface != 0)
        all_inner = ic[f[:, 0]] & ic[f[:, 1]] & ic[f[:, 2]]
        swap = (inward & ~all_inner) | (~inward & all_inner)
        f[swap, 1], f[swap, 2] = f[swap, 2].copy(), f[swap, 1].copy()
        self.f2v.from_numpy(f)

    def _desc(self):
        return dict(kind=0, n_verts=self.n_verts, n_cells=self.n_cells, v_offset=self.offset, mu=self.mu.value, lam=self.lam.value,
                    alpha=self.alpha.value, tets=self.F_vertices_array, B=self.F_B.to_numpy().reshape(-1, 9), W=self.F_W.to_numpy())
