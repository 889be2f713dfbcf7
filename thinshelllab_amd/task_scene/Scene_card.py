"""Card task (system identification of the bending stiffness): counterpart of ``Scene`` in
/root/reference/code/task_scene/Scene_card.py -- three stacked 12x8 cards on a frozen table, two tactile pads rotated by
+-90 degrees about y pinching the stack from the sides and a third one above it, driven by a three-part single gripper."""
import math

import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 1000  # base time_step (BaseScene.py:1327-1370)
    _plastic = 1        # timestep_finish calls update_ref_angle (Scene_card.py:177-181)

    def __init__(self, cloth_size=0.06, device="cuda:0"):
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., 0.]
        self.cloths[0].k_angle[None] = 3.14

    def init_scene_parameters(self):
        # Scene_card.py:36-53
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 3
        self.elastic_cnt = 4
        self.elastic_size = [0.07, 0.015, 0.015, 0.015]
        self.elastic_Nx = 9
        self.elastic_Ny = 9
        self.elastic_Nz = 2
        self.cloth_N = 12
        self.cloth_M = 8
        self.k_contact = 20000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 0.95

    def init_objects(self):
        # Scene_card.py:64-79
        rho = 4e1
        nv = (self.cloth_N + 1) * (self.cloth_M + 1)
        for i in range(self.cloth_cnt):
            self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, i * nv, False, self.cloth_M))
        self.elastic_offset = nv * self.cloth_cnt
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    GRIPPER_POS = np.array([[-0.0285, 0.0, 0.01], [0.0485, 0.0, 0.01], [0.01, 0.0, 0.0185]])

    def init(self):
        # Scene_card.py:81-96
        self.cloths[0].init(-0.02, -0.02, 0.01)
        self.cloths[1].init(-0.02, -0.02, 0.0104)
        self.cloths[2].init(-0.02, -0.02, 0.0108)
        self.elastics[0].init(-0.025, -0.025, -0.00875)
        self.elastics[1].init(-0.0285, 0.0, 0.01, False)
        self.elastics[2].init(0.0485, 0.0, 0.01, False)
        self.elastics[3].init(0.01, 0.0, 0.0185, True)
        self.gripper.init(self, self.GRIPPER_POS)
        h = math.sqrt(2.0) * 0.5
        rot = self.gripper.rot.to_numpy()
        rot[0] = [h, 0, h, 0]
        rot[1] = [h, 0, -h, 0]
        self.gripper.rot.from_numpy(rot)
        self.gripper.get_rotmat()
        self.gripper.get_vert_pos()
        self.gripper.update_all(self)

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_card.py:113-129: neighbouring cards (mu 0.1, both directions), every elastic body against every card
        # (mu_cloth_elastic, ten times that for the two upper cards)
        pairs = []
        cl = self.cloths
        for i in range(self.cloth_cnt):
            for j in range(self.cloth_cnt):
                if abs(i - j) == 1:
                    pairs.append((cl[i].body_idx, cl[j].offset, cl[j].offset + cl[j].NV, 0.1))
                    pairs.append((cl[j].body_idx, cl[i].offset, cl[i].offset + cl[i].NV, 0.1))
        for i in range(self.cloth_cnt):
            for e in self.elastics:
                pairs.append((e.body_idx, cl[i].offset, cl[i].offset + cl[i].NV, None, 0.0 if i == 0 else 10.0))
        return pairs

    def set_frozen_kernel(self):
        # Scene_card.py:131-155
        fr = self.frozen.t.view(-1, 3)
        e0 = self.elastics[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        for e in self.elastics[1:]:
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    def compute_reward(self):
        # Scene_card.py:157-162
        return float(-self.cloths[0].pos.to_numpy()[:, 0].sum())

    def action(self, step, delta_pos, delta_rot):
        # Scene_card.py:164-175 (the pads are part of the global node array here: no push-up needed)
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
