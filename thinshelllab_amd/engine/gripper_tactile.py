"""Paired (upper/lower pad) rigid drive: counterpart of ``gripper``
(/root/reference/code/engine/gripper_tactile.py:10-292).  Host numpy; see gripper_single.py."""
import os

import numpy as np
import torch

from .field import Field
from .gripper_single import _vec, quat_to_rotmat


class gripper:
    paired = True

    def __init__(self, dt, n_verts, n_bound, n_surf, cnt):
        self.n_verts = n_verts
        self.dt = dt
        self.n_bound = n_bound
        self.n_surf = n_surf
        self.n_part = cnt
        z = lambda: Field(torch.zeros((cnt, n_verts, 3), dtype=torch.float64))
        self.F_x_upper, self.F_x_upper_world, self.F_x_lower, self.F_x_lower_world = z(), z(), z(), z()
        self.bound_idx = Field(torch.zeros(n_bound, dtype=torch.int32))
        self.surface_idx = Field(torch.zeros(n_surf, dtype=torch.int32))
        self.pos = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.rot = Field(torch.zeros((cnt, 4), dtype=torch.float64))
        self.d_pos = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.d_angle = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.d_dist = Field(torch.zeros(cnt, dtype=torch.float64))
        self.rotmat = Field(torch.zeros((cnt, 3, 3), dtype=torch.float32))
        self.half_gripper_dist = Field(torch.zeros(cnt, dtype=torch.float64))

    # :103-133
    def init(self, sys, pos_array):
        pos_array = np.asarray(pos_array, dtype=np.float64)
        self.pos.from_numpy(pos_array[: self.n_part])
        r = np.zeros((self.n_part, 4)); r[:, 0] = 1.0
        self.rot.from_numpy(r)
        self.half_gripper_dist.fill(0)
        up = np.zeros((self.n_part, self.n_verts, 3)); lo = np.zeros_like(up)
        for j in range(self.n_part):
            up[j] = sys.elastics[j * 2 + 1].F_x.to_numpy() - pos_array[j]
            lo[j] = sys.elastics[j * 2 + 2].F_x.to_numpy() - pos_array[j]
        self.F_x_upper.from_numpy(up); self.F_x_lower.from_numpy(lo)
        e1 = sys.elastics[1]
        self.bound_idx.from_numpy(np.nonzero(e1.bound_mask())[0].astype(np.int32)[: self.n_bound])
        self.surface_idx.from_numpy(np.nonzero(e1.surf_mask())[0].astype(np.int32)[: self.n_surf])
        self.get_rotmat()

    def set(self, pos, rot, step):
        self.pos.from_numpy(pos.to_numpy()[step]); self.rot.from_numpy(rot.to_numpy()[step])

    def get_rotmat(self):
        q = self.rot.to_numpy()
        self.rotmat.from_numpy(np.stack([quat_to_rotmat(q[j]) for j in range(self.n_part)]).astype(np.float32))

    def get_vert_pos(self):
        R = self.rotmat.to_numpy().astype(np.float64)
        p = self.pos.to_numpy()[:, None, :]
        self.F_x_upper_world.from_numpy(p + np.einsum("jab,jnb->jna", R, self.F_x_upper.to_numpy()))
        self.F_x_lower_world.from_numpy(p + np.einsum("jab,jnb->jna", R, self.F_x_lower.to_numpy()))

    # :178-194
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

    # :196-218 step + open_gripper: rigid motion plus a change of the pad distance (local z of the upper / lower pad)
    def step(self, delta_pos, delta_rot, delta_dis):
        dd = np.asarray(delta_dis.to_numpy() if isinstance(delta_dis, Field) else delta_dis, dtype=np.float64).reshape(-1)
        up = self.F_x_upper.to_numpy(); lo = self.F_x_lower.to_numpy(); hd = self.half_gripper_dist.to_numpy()
        for j in range(self.n_part):
            hd[j] += dd[j]
            up[j, :, 2] += dd[j]
            lo[j, :, 2] -= dd[j]
        self.F_x_upper.from_numpy(up); self.F_x_lower.from_numpy(lo); self.half_gripper_dist.from_numpy(hd)
        self.step_simple(delta_pos, delta_rot)

    # :244-249
    def update_bound(self, sys):
        b = self.bound_idx.to_numpy().astype(np.int64)
        up = self.F_x_upper_world.to_numpy(); lo = self.F_x_lower_world.to_numpy()
        for j in range(self.n_part):
            for e, w in ((sys.elastics[j * 2 + 1], up), (sys.elastics[j * 2 + 2], lo)):
                e.F_x.t[torch.as_tensor(b, device=e.F_x.t.device)] = torch.as_tensor(w[j, b], device=e.F_x.t.device)

    # :220-242
    def gather_grad(self, grad, sys):
        g = (grad.to_numpy() if isinstance(grad, Field) else np.asarray(grad)).reshape(-1, 3)
        b = self.bound_idx.to_numpy().astype(np.int64)
        R = self.rotmat.to_numpy().astype(np.float64)
        up = self.F_x_upper.to_numpy(); lo = self.F_x_lower.to_numpy()
        dpos = np.zeros((self.n_part, 3)); dang = np.zeros((self.n_part, 3))
        for j in range(self.n_part):
            for e, fx in ((sys.elastics[j * 2 + 1], up), (sys.elastics[j * 2 + 2], lo)):
                gj = g[e.offset + b]
                dpos[j] += gj.sum(0)
                dang[j] += np.cross(fx[j, b] @ R[j].T, gj).sum(0)
        dpos /= 2.0 * self.n_bound; dang /= 2.0 * self.n_bound
        self.d_pos.from_numpy(np.clip(dpos, -10, 10)); self.d_angle.from_numpy(np.clip(dang, -10, 10))

    # :258-292
    _FILES = ["F_x_upper", "F_x_upper_world", "F_x_lower", "F_x_lower_world", "pos", "rot", "rotmat", "half_gripper_dist"]

    def save_all(self, path):
        for n in self._FILES:
            np.save(os.path.join(path, n + ".npy"), getattr(self, n).to_numpy())

    def load_all(self, path):
        for n in self._FILES:
            getattr(self, n).from_numpy(np.load(os.path.join(path, n + ".npy")))
