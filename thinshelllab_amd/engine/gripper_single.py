"""Single-pad rigid drive: counterpart of ``gripper`` (/root/reference/code/engine/gripper_single.py:28-162).
A few hundred vertices per part, touched once per step: host numpy; only ``update_bound`` writes into
the scene's HBM node array (frozen pad vertices, :152-156)."""
import numpy as np
import torch

from .field import Field


def quat_to_rotmat(q):
    s, x, y, z = q
    return np.array([[s * s + x * x - y * y - z * z, 2 * (x * y - s * z), 2 * (x * z + s * y)],
                     [2 * (x * y + s * z), s * s - x * x + y * y - z * z, 2 * (y * z - s * x)],
                     [2 * (x * z - s * y), 2 * (y * z + s * x), s * s - x * x - y * y + z * z]])


def _vec(field_or_array, j):
    a = field_or_array.to_numpy() if isinstance(field_or_array, Field) else np.asarray(field_or_array)
    return np.asarray(a[j], dtype=np.float64)


class gripper:
    paired = False

    def __init__(self, dt, n_verts, n_bound, n_surf, cnt):
        self.n_verts = n_verts
        self.dt = dt
        self.n_bound = n_bound
        self.n_surf = n_surf
        self.n_part = cnt
        self.F_x = Field(torch.zeros((cnt, n_verts, 3), dtype=torch.float64))
        self.F_x_world = Field(torch.zeros((cnt, n_verts, 3), dtype=torch.float64))
        self.bound_idx = Field(torch.zeros(n_bound, dtype=torch.int32))
        self.surface_idx = Field(torch.zeros(n_surf, dtype=torch.int32))
        self.pos = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.rot = Field(torch.zeros((cnt, 4), dtype=torch.float64))
        self.d_pos = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.d_angle = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.rotmat = Field(torch.zeros((cnt, 3, 3), dtype=torch.float32))  # f32 like the reference (:48)

    def _pads(self, sys, j):
        return [sys.elastics[j + 1]]

    # :50-74
    def init(self, sys, pos_array):
        pos_array = np.asarray(pos_array, dtype=np.float64)
        self.pos.from_numpy(pos_array[: self.n_part])
        r = np.zeros((self.n_part, 4)); r[:, 0] = 1.0
        self.rot.from_numpy(r)
        fx = np.zeros((self.n_part, self.n_verts, 3))
        for j in range(self.n_part):
            fx[j] = self._pads(sys, j)[0].F_x.to_numpy() - pos_array[j]
        self.F_x.from_numpy(fx)
        e1 = sys.elastics[1]
        self.bound_idx.from_numpy(np.nonzero(e1.bound_mask())[0].astype(np.int32)[: self.n_bound])
        self.surface_idx.from_numpy(np.nonzero(e1.surf_mask())[0].astype(np.int32)[: self.n_surf])
        self.get_rotmat()

    # :76-79
    def set(self, pos, rot, step):
        self.pos.from_numpy(pos.to_numpy()[step]); self.rot.from_numpy(rot.to_numpy()[step])

    # :87-95
    def get_rotmat(self):
        q = self.rot.to_numpy()
        self.rotmat.from_numpy(np.stack([quat_to_rotmat(q[j]) for j in range(self.n_part)]).astype(np.float32))

    def _world(self, local):
        R = self.rotmat.to_numpy().astype(np.float64)  # values already rounded to f32
        return self.pos.to_numpy()[:, None, :] + np.einsum("jab,jnb->jna", R, local)

    # :81-85
    def get_vert_pos(self):
        self.F_x_world.from_numpy(self._world(self.F_x.to_numpy()))

    # :115-131
    def step_simple(self, delta_pos, delta_rot):
        pos = self.pos.to_numpy(); rot = self.rot.to_numpy()
        for j in range(self.n_part):
            dp = _vec(delta_pos, j); dr = _vec(delta_rot, j)
            pos[j] += dp
            v2 = rot[j, 1:4].copy()
            real = -dr.dot(v2)
            res = rot[j, 0] * dr + np.cross(dr, v2)
            rot[j, 0] += real
            rot[j, 1:4] += res
            rot[j] /= np.linalg.norm(rot[j])
        self.pos.from_numpy(pos); self.rot.from_numpy(rot)
        self.get_rotmat()
        self.get_vert_pos()

    # :152-156
    def update_bound(self, sys):
        b = self.bound_idx.to_numpy().astype(np.int64)
        w = self.F_x_world.to_numpy()
        for j in range(self.n_part):
            e = self._pads(sys, j)[0]
            e.F_x.t[torch.as_tensor(b, device=e.F_x.t.device)] = torch.as_tensor(w[j, b], device=e.F_x.t.device)

    # :158-162
    def update_all(self, sys):
        w = self.F_x_world.to_numpy()
        for j in range(self.n_part):
            e = self._pads(sys, j)[0]
            e.F_x.from_numpy(w[j])

    # :133-150
    def gather_grad(self, grad, sys):
        g = (grad.to_numpy() if isinstance(grad, Field) else np.asarray(grad)).reshape(-1, 3)
        b = self.bound_idx.to_numpy().astype(np.int64)
        R = self.rotmat.to_numpy().astype(np.float64)
        fx = self.F_x.to_numpy()
        dpos = np.zeros((self.n_part, 3)); dang = np.zeros((self.n_part, 3))
        for j in range(self.n_part):
            gj = g[self._pads(sys, j)[0].offset + b]
            dpos[j] = gj.sum(0)
            dang[j] = np.cross(fx[j, b] @ R[j].T, gj).sum(0)
        dpos /= 1.0 * self.n_bound; dang /= 1.0 * self.n_bound
        self.d_pos.from_numpy(np.clip(dpos, -10, 10)); self.d_angle.from_numpy(np.clip(dang, -100, 100))
