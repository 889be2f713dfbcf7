"""Sliding task (system identification of the cloth-cloth friction coefficient): counterpart of ``Scene`` in
/root/reference/code/task_scene/Scene_sliding.py -- three stacked 15x15 sheets on a frozen 16x16x2 table, one tactile pad on
top dragging the stack; the cloth-cloth pairs use a second live friction parameter ``mu_cloth_cloth``."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.field import ScalarField
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50   # Scene_sliding.py:245-269
    _plastic = 1       # timestep_finish calls update_ref_angle (:133-137)

    def __init__(self, cloth_size=0.06, device="cuda:0"):
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., 0.]
        self.cloths[0].k_angle[None] = 3.14
        self.mu_cloth_cloth = ScalarField(1.0, lambda f: self._set_param("mu_cloth_cloth", f.value))
        # Scene_sliding.py:27-32: stiffer pad
        e = self.elastics[1]
        e.E = 500000
        e.nu = 0.2
        mu, lam = e.E / (2 * (1 + e.nu)), e.E * e.nu / ((1 + e.nu) * (1 - 2 * e.nu))
        e.mu[None] = mu
        e.lam[None] = lam
        e.alpha[None] = 1 + mu / lam

    def init_scene_parameters(self):
        # Scene_sliding.py:34-50
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 3
        self.elastic_cnt = 2
        self.elastic_size = [0.1, 0.015]
        self.elastic_Nx = 16
        self.elastic_Ny = 16
        self.elastic_Nz = 2
        self.cloth_N = 15
        self.cloth_M = 15
        self.k_contact = 10000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # BaseScene.init_objects with three square sheets (the scene does not override it; cloth_cnt = 3)
        rho = 4e1
        nv = (self.cloth_N + 1) * (self.cloth_N + 1)
        for i in range(self.cloth_cnt):
            self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, i * nv))
        self.elastic_offset = nv * self.cloth_cnt
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_sliding.py:61-68
        self.cloths[0].init(-0.03, -0.03, 0.0004)
        self.cloths[1].init(-0.03, -0.03, 0.0008)
        self.cloths[2].init(-0.03, -0.03, 0.0012)
        self.elastics[0].init(-0.05, -0.05, -0.00666)
        self.elastics[1].init(0.0, 0., 0.0105, True)
        self.gripper.init(self, np.array([[0.0, 0., 0.0105]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_sliding.py:78-97: neighbouring sheets first (mu_cloth_cloth, both directions; their constraints are the "nc1"
        # block of the friction gradient), then every sheet against the table (0.4) and the pad (mu_cloth_elastic)
        pairs = []
        cl = self.cloths
        for i in range(self.cloth_cnt):
            for j in range(self.cloth_cnt):
                if abs(i - j) == 1:
                    pairs.append((cl[i].body_idx, cl[j].offset, cl[j].offset + cl[j].NV, "cloth_cloth"))
                    pairs.append((cl[j].body_idx, cl[i].offset, cl[i].offset + cl[i].NV, "cloth_cloth"))
        for i in range(self.cloth_cnt):
            for j, e in enumerate(self.elastics):
                mu = 0.4 if j == 0 else None
                pairs.append((cl[i].body_idx, e.offset, e.offset + e.n_verts, mu))
                pairs.append((e.body_idx, cl[i].offset, cl[i].offset + cl[i].NV, mu))
        return pairs

    def set_frozen_kernel(self):
        # Scene_sliding.py:99-111
        fr = self.frozen.t.view(-1, 3)
        e0, e1 = self.elastics[0], self.elastics[1]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        fr[e1.offset:e1.offset + e1.n_verts][torch.as_tensor(e1.bound_mask())] = 1

    def compute_reward(self):
        # Scene_sliding.py:113-118
        return float(-self.cloths[0].pos.to_numpy()[:, 0].sum())

    def action(self, step, delta_pos, delta_rot):
        # Scene_sliding.py:120-131
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
