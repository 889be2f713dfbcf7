"""Forming task: counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_forming.py -- a 15x7 sheet folded with a wide
arc (half_curve_num = 3) over the frozen table, one tactile pad pressing it into a target shape; plastic hinges."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 1000  # base time_step (BaseScene.py:1327-1370)
    _plastic = 1        # timestep_finish calls update_ref_angle (Scene_forming.py:141-145)

    def __init__(self, cloth_size=0.06, device="cuda:0"):
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., 0.]

    def init_scene_parameters(self):
        # Scene_forming.py:30-46
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 2
        self.elastic_size = [0.07, 0.015]
        self.elastic_Nx = 9
        self.elastic_Ny = 9
        self.elastic_Nz = 2
        self.cloth_N = 15
        self.cloth_M = 7
        self.k_contact = 20000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # Scene_forming.py:57-72
        rho = 4e1
        self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, 0, False, self.cloth_M))
        self.elastic_offset = (self.cloth_N + 1) * (self.cloth_M + 1)
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_forming.py:74-83
        half_curve_num = 3
        c = self.cloths[0]
        c.init_fold(-0.07, -0.02, 0.00035, half_curve_num)
        self.elastics[0].init(-0.035, -0.035, -0.00875)
        r = c.grid_len * (half_curve_num * 2 - 1) / 3.1415
        x = -0.07 + (7 + half_curve_num) / 16 * 0.1 - r * 0.86 + 0.01
        self.elastics[1].init(x, 0.0, 2 * r + 0.00785, True)
        self.gripper.init(self, np.array([[x, 0.0, 2 * r + 0.00785]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_forming.py:95-105
        pairs = []
        for c in self.cloths:
            for e in self.elastics:
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, None))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, None))
        return pairs

    def set_frozen_kernel(self):
        # Scene_forming.py:107-123
        fr = self.frozen.t.view(-1, 3)
        e0, e1, c = self.elastics[0], self.elastics[1], self.cloths[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        fr[e1.offset:e1.offset + e1.n_verts][torch.as_tensor(e1.bound_mask())] = 1
        fr[c.offset + c.N * (c.M + 1): c.offset + (c.N + 1) * (c.M + 1)] = 1

    def compute_reward(self, target_pos):
        # Scene_forming.py:125-132
        d = self.cloths[0].pos.to_numpy() - np.asarray(target_pos)
        return float(-(d ** 2).sum())

    def action(self, step, delta_pos, delta_rot):
        # Scene_forming.py:134-139
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
