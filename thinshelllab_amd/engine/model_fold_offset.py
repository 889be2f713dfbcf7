"""Thin-shell body: host-side counterpart of ``Cloth``
(/root/reference/code/engine/model_fold_offset.py:10-107).

Only one-time work lives here (index tables, rest data, initial poses, initial plastic rest angles);
energies / forces / Hessians run in ``libtsl_hip.so`` (csrc/k_cloth.hpp).  ``pos``/``vel``/``prev_pos``
become views into the scene's global node arrays once the scene binds the body.
"""
import math

import numpy as np
import torch

from .field import Field, ScalarField


class Cloth:
    def __init__(self, N, dt, Len, tot_NV, rho, offset, is_square=True, M=0):
        # model_fold_offset.py:11-33
        self.is_square = is_square
        self.N = N
        self.M = N if is_square else M
        self.dt = dt
        self.dx = Len / N
        self.NF = 2 * self.N * self.M
        self.NV = (self.N + 1) * (self.M + 1)
        self.offset = offset
        self.rho = rho
        self.base_area = self.dx ** 2 * 0.5
        self.grid_len = self.dx
        self.mass = rho * (self.dx ** 2)
        self._sys = None
        self._idx = 0
        self.Kl = ScalarField(1000.0, self._param("Kl"))
        self.Ka = ScalarField(1000.0, self._param("Ka"))
        self.Kb = ScalarField(100.0, self._param("Kb"))
        self.k_angle = ScalarField(3.14, self._param("k_angle"))
        self.gravity = ScalarField([0.0, 0.0, -9.8], self._gravity_written)
        z3 = lambda n: torch.zeros((n, 3), dtype=torch.float64)
        self.pos = Field(z3(self.NV))
        self.prev_pos = Field(z3(self.NV))
        self.vel = Field(z3(self.NV))
        self.manipulate_force = Field(z3(self.NV))
        self.ref_angle = Field(z3(self.NF))
        self.f2v = Field(torch.zeros((self.NF, 3), dtype=torch.int32))
        self.counter_face = Field(torch.zeros((self.NF, 3), dtype=torch.int32))
        self.counter_point = Field(torch.zeros((self.NF, 3), dtype=torch.int32))
        self.V = Field(torch.zeros(self.NF, dtype=torch.float64))
        self.l_i = Field(torch.zeros((self.NF, 3), dtype=torch.float64))
        self.offset_faces = 0
        self.body_idx = 0
        self.x32 = None
        self.f_vis = None

    # -- parameter plumbing
    def _param(self, name):
        def cb(field):
            if self._sys is not None and self._sys._ctx is not None:
                self._sys._ctx.set_param(f"cloth{self._idx}.{name}", field.value)
        return cb

    def _gravity_written(self, field):
        if self._sys is not None:
            self._sys._refresh_gravity()

    # -- model_fold_offset.py:928-1018 (fields are zero-initialised; unwritten entries stay 0)
    def init_mesh(self):
        N, M = self.N, self.M
        i, j = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
        i = i.ravel(); j = j.ravel()
        k = (i * M + j) * 2
        a = i * (M + 1) + j
        b = a + 1
        c = a + M + 2
        d = a + M + 1
        even = (i + j) % 2 == 0
        f2v = np.zeros((self.NF, 3), np.int32)
        cf = np.zeros((self.NF, 3), np.int32)
        cp = np.zeros((self.NF, 3), np.int32)
        f2v[k] = np.where(even[:, None], np.stack([c, b, a], 1), np.stack([b, a, d], 1))
        f2v[k + 1] = np.where(even[:, None], np.stack([a, d, c], 1), np.stack([d, c, b], 1))
        up = ((i - 1) * M + j) * 2 + 1      # second triangle of cell (i-1, j)
        down = ((i + 1) * M + j) * 2        # first triangle of cell (i+1, j)
        e, o = even, ~even
        # even cells (:943-971)
        cf[k[e], 0] = np.where(i[e] > 0, up[e], -1);            cp[k[e], 0] = np.where(i[e] > 0, 2, 0)
        cf[k[e], 2] = np.where(j[e] < M - 1, k[e] + 2, -1);     cp[k[e], 2] = 0
        cf[k[e] + 1, 0] = np.where(i[e] < N - 1, down[e], -1);  cp[k[e] + 1, 0] = np.where(i[e] < N - 1, 2, 0)
        cf[k[e] + 1, 2] = np.where(j[e] > 0, k[e] - 2, -1);     cp[k[e] + 1, 2] = 0
        # odd cells (:981-1009): slot 2 of the first triangle is written twice, slot 0 never
        cf[k[o], 2] = np.where(i[o] > 0, up[o], -1);            cp[k[o], 2] = 0
        cf[k[o] + 1, 0] = np.where(j[o] < M - 1, k[o] + 3, -1); cp[k[o] + 1, 0] = np.where(j[o] < M - 1, 2, 0)
        cf[k[o] + 1, 2] = np.where(i[o] < N - 1, down[o], -1);  cp[k[o] + 1, 2] = 0
        cf[k[o], 2] = np.where(j[o] > 0, k[o] - 2, -1);         cp[k[o], 2] = np.where(j[o] > 0, 2, cp[k[o], 2])
        cf[k, 1] = k + 1; cp[k, 1] = 1
        cf[k + 1, 1] = k; cp[k + 1, 1] = 1
        self.f2v.from_numpy(f2v); self.counter_face.from_numpy(cf); self.counter_point.from_numpy(cp)

    def _rest(self):
        self.V.fill(self.grid_len ** 2 * 0.5)
        li = np.empty((self.NF, 3)); li[:, 0] = self.grid_len; li[:, 1] = self.grid_len; li[:, 2] = self.grid_len * math.sqrt(2.0)
        self.l_i.from_numpy(li)

    # -- model_fold_offset.py:825-838
    def init_pos_offset(self, offsetx, offsety, offsetz):
        self.ref_angle.fill(0)
        i, j = np.meshgrid(np.arange(self.N + 1), np.arange(self.M + 1), indexing="ij")
        p = np.stack([i * self.grid_len + offsetx, j * self.grid_len + offsety, np.full(i.shape, float(offsetz))], -1).reshape(-1, 3)
        self.pos.from_numpy(p)
        self.vel.fill(0)
        self._rest()

    # -- model_fold_offset.py:840-868.  ``rows`` generalises the hard-coded 15-row layout (top layer rows <= L,
    # arc rows L+1..R-1, bottom rows >= R, x mirrored about row ``rows``) to finer grids for the scaled
    # BASELINE configs; rows=15 with half_curv_num=2 is the reference pose.
    def init_pos_offset_fold(self, offsetx, offsety, offsetz, half_curv_num, rows=15, L=None, R=None, r=None):
        self.ref_angle.fill(0)
        g = self.grid_len
        if r is None:
            r = g
            if half_curv_num != 2:
                r = g * (half_curv_num * 2 - 1) / 3.1415
        if L is None:
            L = 7 - half_curv_num + 1
        if R is None:
            R = 7 + half_curv_num
        n_arc = R - L  # reference: half_curv_num * 2 - 1
        p = np.zeros((self.N + 1, self.M + 1, 3))
        j = np.arange(self.M + 1)
        for i in range(self.N + 1):
            if i <= L:
                p[i, :, 0] = (rows - i) * g + offsetx; p[i, :, 1] = j * g + offsety; p[i, :, 2] = offsetz + 2 * r
            if L + 1 <= i <= R - 1:
                x = (rows - L) * g
                angle = (i - L) / n_arc * 3.1415
                p[i, :, 0] = x - r * math.sin(angle) + offsetx; p[i, :, 1] = j * g + offsety; p[i, :, 2] = offsetz + r * (1 + math.cos(angle))
            if i >= R:
                p[i, :, 0] = i * g + offsetx; p[i, :, 1] = j * g + offsety; p[i, :, 2] = offsetz
        self.pos.from_numpy(p.reshape(-1, 3))
        self.vel.fill(0)
        self._rest()

    # -- host versions of compute_normal_dir (:169-174) / compute_angle (:126-138), used only at init
    def _hinge_angles(self):
        pos = self.pos.to_numpy(); f2v = self.f2v.to_numpy(); cf = self.counter_face.to_numpy()
        a, b, c = pos[f2v[:, 0]], pos[f2v[:, 1]], pos[f2v[:, 2]]
        n = np.cross(b - a, c - b)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        fi, l = np.nonzero(cf > np.arange(self.NF)[:, None])
        f2 = cf[fi, l]
        cos = np.einsum("ij,ij->i", n[fi], n[f2])
        with np.errstate(invalid="ignore"):
            theta = np.where(cos < 0.999999, np.arccos(np.clip(cos, -1, 1)), 2 * np.sqrt(np.abs(1.0 - cos)) / np.sqrt(1 + cos))
        e = pos[f2v[fi, (l + 1) % 2]] - pos[f2v[fi, l]]
        theta = np.where(np.einsum("ij,ij->i", n[f2], e) < -1e-10 * np.linalg.norm(e, axis=1), -theta, theta)
        return fi, l, theta

    # -- model_fold_offset.py:787-797
    def init_ref_angle(self):
        fi, l, theta = self._hinge_angles()
        ra = self.ref_angle.to_numpy()
        dis = theta - ra[fi, l]
        ad = np.abs(dis)
        ka = self.k_angle.value
        m = ad > ka
        ra[fi[m], l[m]] += (ad[m] - ka) * dis[m] / ad[m]
        self.ref_angle.from_numpy(ra)

    # :812-822: rest angle 1.7 rad on the hinges whose wings sit on grid rows (4, 6) and (9, 11)
    def init_ref_angle_bridge(self):
        f2v = self.f2v.to_numpy(); cf = self.counter_face.to_numpy(); cp = self.counter_point.to_numpy()
        fi, l = np.nonzero(cf > np.arange(self.NF)[:, None])
        r1 = f2v[fi, l] // (self.M + 1)
        r2 = f2v[cf[fi, l], cp[fi, l]] // (self.M + 1)
        m = ((r1 == 4) & (r2 == 6)) | ((r1 == 9) & (r2 == 11))
        ra = self.ref_angle.to_numpy()
        ra[fi[m], l[m]] = 1.7
        self.ref_angle.from_numpy(ra)
        self._init_args = ("bridge",) + tuple(self._init_args[1:])

    def init(self, offsetx, offsety, offsetz):
        self._init_args = ("flat", offsetx, offsety, offsetz, 0)
        self.init_mesh()
        self.init_pos_offset(offsetx, offsety, offsetz)
        self.ref_angle.fill(0)

    def init_fold(self, offsetx, offsety, offsetz, curv_num, **kw):
        self._init_args = ("fold" if not kw else "fold_scaled", offsetx, offsety, offsetz, curv_num)
        self.init_mesh()
        self.init_pos_offset_fold(offsetx, offsety, offsetz, curv_num, **kw)
        self.init_ref_angle()

    def init_load(self, ref_pos):
        self.init_mesh()
        self.ref_angle.fill(0)
        self.pos.from_numpy(np.asarray(ref_pos)[: self.NV])
        self.vel.fill(0)
        self._rest()

    def clear_manipulation(self):
        self.manipulate_force.fill(0)

    def get_vert_mass(self, i):
        return self.mass

    # -- description handed to tsl_ctx_create
    def _desc(self):
        return dict(N=self.N, M=self.M, NV=self.NV, NF=self.NF, v_offset=self.offset, dx=self.dx, mass=self.mass,
                    Kl=self.Kl.value, Ka=self.Ka.value, Kb=self.Kb.value, k_angle=self.k_angle.value,
                    f2v=self.f2v.to_numpy(), counter_face=self.counter_face.to_numpy(), counter_point=self.counter_point.to_numpy(),
                    rest_area=self.V.to_numpy(), rest_len=self.l_i.to_numpy())
