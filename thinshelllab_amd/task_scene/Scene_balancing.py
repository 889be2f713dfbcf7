"""Balancing task: counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_balancing.py
(15x7 cloth held by two paired tactile grippers, a heavy ball resting on it).  ``cloth_N``/``cloth_M`` scale
the grid for the 100k-triangle BASELINE config; ``geom_scale`` enlarges the whole scene
(bodies, offsets, contact shell, broad-phase cell) by one factor so that a refined cloth keeps the native 4 mm spacing."""
import os

import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50  # Scene_balancing.py:251
    _plastic = 0

    def __init__(self, cloth_size=0.06, device="cuda:0", cloth_N=15, cloth_M=7, geom_scale=1.0):
        self._cN, self._cM = cloth_N, cloth_M
        self._gs = float(geom_scale)
        super().__init__(cloth_size=cloth_size, enable_gripper=True, device=device)
        self.cloths[0].k_angle[None] = 3.14

    def init_scene_parameters(self):
        # Scene_balancing.py:31-48
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 5
        self.elastic_size = [0.007 * self._gs] + [0.015 * self._gs] * 4
        self.elastic_Nx = 5
        self.elastic_Ny = 5
        self.elastic_Nz = 5
        self.cloth_N = self._cN
        self.cloth_M = self._cM
        self.k_contact = 10000
        self.eps_contact = 0.00041 * self._gs
        self.grid_h = 0.003 * self._gs
        self.grid_extent = 0.2 * self._gs
        self.eps_v = 0.01
        self.max_n_constraints = 10000 if self._cN <= 15 else 400000
        self.damping = 1.0

    def init_objects(self):
        # Scene_balancing.py:50-67
        rho = 4e1
        self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, 0, is_square=False, M=self.cloth_M))
        self.elastic_offset = (self.cloth_N + 1) * (self.cloth_M + 1)
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz, 10000.0 / self._gs, load=True))
        if self._gs != 1.0:
            self.elastics[0].vertex = self.elastics[0].vertex * self._gs  # the loaded ball mesh is in absolute coordinates
        tmp_tot += self.elastics[0].n_verts
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_balancing.py:78-86 (cloth centred like the native 0.06 x 0.028 sheet)
        c = self.cloths[0]
        self.cloths[0].init(-0.5 * c.dx * c.N, -0.5 * c.dx * c.M if (self._cN, self._cM) != (15, 7) else -0.015, 0.)
        g = self._gs
        self.elastics[0].init(0., 0., 0.0039 * g)
        self.elastics[1].init(0.023 * g, 0., 0.0079 * g, True)
        self.elastics[2].init(0.023 * g, 0., -0.0079 * g, False)
        self.elastics[3].init(-0.023 * g, 0, 0.0079 * g, True)
        self.elastics[4].init(-0.023 * g, 0, -0.0079 * g, False)
        self.gripper.init(self, np.array([[0.023 * g, 0., 0.0], [-0.023 * g, 0., 0.0]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_balancing.py:98-109
        pairs = []
        for c in self.cloths:
            for j, e in enumerate(self.elastics):
                mu = 0.2 if j == 0 else None
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, mu))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, mu))
        return pairs

    def set_frozen_kernel(self):
        # Scene_balancing.py:111-136
        fr = self.frozen.t.view(-1, 3)
        for e in self.elastics[1:]:
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    def _centre(self):
        return (self.cloth_N + 1) // 2 * (self.cloth_M + 1) + (self.cloth_M + 1) // 2

    def compute_reward(self):
        # Scene_balancing.py:138-145
        e = self.elastics[0]
        x = e.F_x.to_numpy(); ctr = self.cloths[0].pos.to_numpy()[self._centre()]
        return float(-(((x[:, 0] - ctr[0]) ** 2) + ((x[:, 1] - ctr[1]) ** 2)).sum())

    def compute_reward_all(self, analy_grad):
        # Scene_balancing.py:147-154
        e = self.elastics[0]
        pb = analy_grad.pos_buffer.t
        tt = self.cloths[0].offset + self._centre()
        d = pb[:, e.offset:e.offset + e.n_verts, 0:2] - pb[:, tt:tt + 1, 0:2]
        return float(-(d ** 2).sum().item())

    def action(self, step, delta_pos, delta_rot):
        # Scene_balancing.py:182-199
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)

    # Scene_balancing.py:202-222
    def save_all(self, path):
        self.gripper.save_all(path)
        self.save_state(os.path.join(path, "state"))
        flag, dr, _, _ = self._ensure_ctx().proj_export()
        np.save(os.path.join(path, "proj_flag.npy"), flag); np.save(os.path.join(path, "proj_dir.npy"), dr)
        np.save(os.path.join(path, "border_flag.npy"), self.border_flag.to_numpy())

    def load_all(self, path):
        self.gripper.load_all(path)
        self.load_state(os.path.join(path, "state"))
        self._ensure_ctx().proj_import(np.load(os.path.join(path, "proj_flag.npy")), np.load(os.path.join(path, "proj_dir.npy")))
        self.border_flag.from_numpy(np.load(os.path.join(path, "border_flag.npy")))
