"""Box / loaded-mesh FEM body: host-side counterpart of ``Elastic``
(/root/reference/code/engine/model_elastic_offset.py:11-92, :232-250, :285-304, :333-405).
One-time work only; the Neo-Hookean (log J) energy / force / Hessian (:94-208, :314-331) run in
libtsl_hip.so (csrc/k_fem.hpp, kind 1).
"""
import numpy as np
import torch

from . import readfile
from .field import Field, ScalarField


class Elastic:
    kind = 1

    def __init__(self, dt, Len, offset, Nx, Ny, Nz, density=2000.0, load=False):
        self.E = 5e5
        self.nu = 0.0
        mu, lam = self.E / (2 * (1 + self.nu)), self.E * self.nu / ((1 + self.nu) * (1 - 2 * self.nu))
        self.mu = ScalarField(mu, self._param("mu")); self.lam = ScalarField(lam, self._param("lam"))
        self.density = density
        self.dt = dt
        self.offset = offset
        self.gravity = ScalarField([0.0, 0.0, -9.8], self._gravity_written)
        self._sys = None
        n_cube = np.array([int(Nx), int(Ny), int(Nz)])
        self.n_cube = n_cube
        self.n_verts = int(n_cube.prod())
        self.n_cells = int(5 * (n_cube - 1).prod())
        self.dx = Len / (n_cube.max() - 1)
        su = sum((n_cube[i] - 1) * (n_cube[(i + 1) % 3] - 1) for i in range(3))
        self.n_surfaces = int(2 * su * 2)
        self.load = False
        if load:
            self.n_verts, v = readfile.read_node("../data/ball.node")
            self.n_cells, t = readfile.read_ele("../data/ball.ele")
            self.n_surfaces, s = readfile.read_smesh("../data/ball.face")
            self.vertex = np.array(v, dtype=np.float64); self.tet_mesh = np.array(t, dtype=np.int32); self.surface_mesh = np.array(s, dtype=np.int32)
            self.load = True
        nv, nc = self.n_verts, self.n_cells
        z3 = lambda n: torch.zeros((n, 3), dtype=torch.float64)
        self.F_vertices = Field(torch.zeros((nc, 4), dtype=torch.int32))
        self.F_x = Field(z3(nv)); self.F_x_prev = Field(z3(nv)); self.F_v = Field(z3(nv)); self.F_ox = Field(z3(nv))
        self.F_m = Field(torch.zeros(nv, dtype=torch.float64))
        self.F_B = Field(torch.zeros((nc, 3, 3), dtype=torch.float64))
        self.F_W = Field(torch.zeros(nc, dtype=torch.float64))
        self.ext_force = Field(z3(nv))
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

    def i2p(self, I):
        return (I[..., 0] * self.n_cube[1] + I[..., 1]) * self.n_cube[2] + I[..., 2]

    # -- :285-304: five tets per cube, corner codes XOR-ed with the cube parity
    def get_vertices(self):
        nx, ny, nz = self.n_cube
        I = np.stack(np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), np.arange(nz - 1), indexing="ij"), -1).reshape(-1, 3)
        e = ((I[:, 0] * (ny - 1) + I[:, 1]) * (nz - 1) + I[:, 2]) * 5
        tets = np.zeros((self.n_cells, 4), np.int32)
        codes = [(j, j ^ 1, j ^ 2, j ^ 4) for j in (0, 3, 5, 6)] + [(1, 2, 4, 7)]
        for slot, vs in enumerate(codes):
            for c, v in enumerate(vs):
                bits = np.array([(v >> k) & 1 for k in range(3)])
                tets[e + slot, c] = self.i2p(I + ((bits[None, :] ^ I) & 1))
        self.F_vertices.from_numpy(tets)
        G = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
        ox = np.zeros((self.n_verts, 3))
        ox[self.i2p(G)] = G * self.dx
        self.F_ox.from_numpy(ox)

    # -- :232-250
    def init_pos(self, offsetx, offsety, offsetz):
        x = self.F_ox.to_numpy()
        t = self.F_vertices.to_numpy()
        Ds = np.stack([x[t[:, 0]] - x[t[:, 3]], x[t[:, 1]] - x[t[:, 3]], x[t[:, 2]] - x[t[:, 3]]], axis=2)
        self.F_B.from_numpy(np.linalg.inv(Ds))
        W = np.abs(np.linalg.det(Ds)) / 6
        self.F_W.from_numpy(W)
        m = np.zeros(self.n_verts)
        np.add.at(m, t.ravel(), np.repeat(W / 4 * self.density, 4))
        self.F_m.from_numpy(m)
        self.F_x.from_numpy(x + np.array([offsetx, offsety, offsetz]))
        self.F_v.fill(0)

    # -- :333-376 (the reference appends with an atomic counter: face order is arbitrary there; cell order here)
    def get_surface_indices(self):
        x = self.F_x.to_numpy(); t = self.F_vertices.to_numpy()
        nc = self.n_cube

        def check(u):
            ans = np.zeros_like(u); rest = u.copy()
            for i in range(3):
                k = rest % nc[2 - i]; rest = rest // nc[2 - i]
                ans |= np.where(k == 0, 1 << (i * 2), 0)
                ans |= np.where(k == nc[2 - i] - 1, 1 << (i * 2 + 1), 0)
            return ans

        out = []
        for c in range(self.n_cells):
            if c % 5 == 4:
                continue
            for i in (0, 2, 3):
                verts = [int(t[c][(i + j) % 4]) for j in range(3)]
                cv = check(np.array(verts))
                if cv[0] & cv[1] & cv[2]:
                    v3 = int(t[c][(i + 3) % 4])
                    normal = np.cross(x[verts[1]] - x[verts[0]], x[verts[2]] - x[verts[0]])
                    if normal.dot(x[v3] - x[verts[0]]) > 0:
                        verts[1], verts[2] = verts[2], verts[1]
                    out.append(verts)
        f = np.zeros((self.n_surfaces, 3), np.int32)
        f[: len(out)] = np.array(out, np.int32)[: self.n_surfaces]
        self.f2v.from_numpy(f)

    # -- :378-393
    def init_normal(self, offset_x, offset_y, offset_z):
        x = self.F_x.to_numpy(); f = self.f2v.to_numpy()
        p1, p2, p3 = x[f[:, 0]], x[f[:, 1]], x[f[:, 2]]
        n = np.cross(p2 - p1, p3 - p1)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        swap = np.einsum("ij,ij->i", n, np.array([offset_x, offset_y, offset_z]) - p1) > 0
        f[swap, 1], f[swap, 2] = f[swap, 2].copy(), f[swap, 1].copy()
        self.f2v.from_numpy(f)

    # -- :395-405
    def init(self, offsetx, offsety, offsetz):
        self._init_args = (offsetx, offsety, offsetz, False)
        if not self.load:
            self.get_vertices()
            self.init_pos(offsetx, offsety, offsetz)
            self.get_surface_indices()
        else:
            self.F_vertices.from_numpy(self.tet_mesh)
            self.f2v.from_numpy(self.surface_mesh)
            self.F_ox.from_numpy(self.vertex)
            self.init_pos(offsetx, offsety, offsetz)
            self.init_normal(offsetx, offsety, offsetz)

    # -- :406-409 init_arch with init_pos_arch (:253-270): the rest configuration is an arch, z += arch sin(pi x / (Nx - 1))
    def init_arch(self, offsetx, offsety, offsetz, arch):
        self._init_args = (offsetx, offsety, offsetz, False)
        self._arch = float(arch)
        self.get_vertices()
        ox = self.F_ox.to_numpy().copy()
        I = np.arange(self.n_verts) // (self.n_cube[1] * self.n_cube[2])
        ox[:, 2] += arch * np.sin(I.astype(np.float64) / float(self.n_cube[0] - 1) * 3.1415926)
        flat = self.F_ox.to_numpy().copy()
        self.F_ox.from_numpy(ox)          # init_pos computes the rest matrices and masses from F_ox
        self.init_pos(offsetx, offsety, offsetz)
        self.F_ox.from_numpy(flat)        # the reference leaves F_ox flat (only F_x carries the arch)
        self.get_surface_indices()

    def _desc(self):
        return dict(kind=1, n_verts=self.n_verts, n_cells=self.n_cells, v_offset=self.offset, mu=self.mu.value, lam=self.lam.value, alpha=0.0,
                    tets=self.F_vertices.to_numpy(), B=self.F_B.to_numpy().reshape(-1, 9), W=self.F_W.to_numpy())
