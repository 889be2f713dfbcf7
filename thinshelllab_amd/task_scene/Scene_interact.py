"""Interaction task ("separating"): counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_interact.py -- a 15x15 sheet
on a frozen 16x16x2 table held by one paired tactile gripper, with a free 6x6x4 block lying on the sheet (cloth-body, body-table
contact); the gripper closes during the first steps (``gripper.step`` with a pad distance), then moves along the trajectory."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50   # Scene_interact.py:213-236
    _plastic = 1       # timestep_finish calls update_ref_angle (:189-193)

    def __init__(self, cloth_size=0.06, device="cuda:0", soft=False, dense=10000.0):
        self.dense = dense
        self.soft = soft
        super().__init__(cloth_size=cloth_size, enable_gripper=True, device=device)
        self.gravity[None] = [0., 0., -9.8]
        self.cloths[0].k_angle[None] = 3.14

    def init_scene_parameters(self):
        # Scene_interact.py:37-56
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 4
        self.elastic_size = [0.06, 0.015, 0.015, 0.012]
        self.elastic_Nx = 16
        self.elastic_Ny = 16
        self.elastic_Nz = 2
        self.cloth_N = 15
        self.cloth_M = 15
        self.extra_obj = True
        self.effector_cnt = 3
        self.k_contact = 30000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # Scene_interact.py:58-80 (the soft / rigid branches build the same block)
        rho = 4e1
        nv = (self.cloth_N + 1) ** 2
        for i in range(self.cloth_cnt):
            self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, i * nv))
        self.elastic_offset = nv * self.cloth_cnt
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt - 1):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.elastics.append(Elastic(self.dt, self.elastic_size[3], tmp_tot, 6, 6, 4, density=self.dense))
        tmp_tot += 6 * 6 * 4
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_interact.py:89-96
        self.cloths[0].init(-0.045, -0.03, 0.0004)
        self.elastics[0].init(-0.03, -0.03, -0.004)
        self.elastics[1].init(-0.04, 0., 0.0083, True)
        self.elastics[2].init(-0.04, 0., -0.0075, False)
        self.elastics[3].init(0.001, -0.006, 0.0008)
        self.gripper.init(self, np.array([[-0.04, 0., 0.0004]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_interact.py:107-124: cloth against every body (0.2 for the table and the block), block against the table (0.1)
        pairs = []
        for c in self.cloths:
            for j, e in enumerate(self.elastics):
                mu = 0.2 if j in (0, 3) else None
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, mu))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, mu))
        e0, e3 = self.elastics[0], self.elastics[3]
        pairs.append((e0.body_idx, e3.offset, e3.offset + e3.n_verts, 0.1))
        pairs.append((e3.body_idx, e0.offset, e0.offset + e0.n_verts, 0.1))
        return pairs

    def set_frozen_kernel(self):
        # Scene_interact.py:127-146
        fr = self.frozen.t.view(-1, 3)
        e0 = self.elastics[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        for e in self.elastics[1:3]:
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    def compute_reward(self):
        # Scene_interact.py:148-155
        return float(-self.cloths[0].pos.to_numpy()[:, 0].sum() + self.elastics[3].F_x.to_numpy()[:, 0].sum() * 256.0 / 144.0)

    def compute_reward_1(self):
        # Scene_interact.py:157-162
        return float(-self.elastics[3].F_x.to_numpy()[:, 0].sum())

    def action(self, step, delta_pos, delta_rot):
        # Scene_interact.py:164-172: the gripper closes by 0.6 mm per step during the first four steps
        if step < 5:
            self.gripper.step(delta_pos, delta_rot, np.array([-0.0006] * self.gripper.n_part))
        else:
            self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
