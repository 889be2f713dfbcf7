"""Synthetic cloth-only scene (no reference counterpart): the square-cloth free-drape workload of
BASELINE.json configs[1] / SURVEY.md section 8d cfg2 -- N x M grid, one pinned row (i = N, like
Scene_folding.py:123-127), gravity, no contact bodies.  It reuses the BaseScene driver unchanged."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50

    def __init__(self, cloth_size=0.1 / 15 * 71, N=71, M=None, Kb=100.0, k_angle=3.14, pin_row=True, perturb=1e-4, device="cuda:0", newton_cap=50):
        self._N = N
        self._M = N if M is None else M
        self._Kb = Kb
        self._k_angle = k_angle
        self._pin = pin_row
        self._perturb = perturb
        self._newton_cap = newton_cap
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.cloths[0].Kb[None] = Kb
        self.cloths[0].k_angle[None] = k_angle

    def init_scene_parameters(self):
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 0
        self.cloth_N = self._N
        self.cloth_M = self._M
        self.k_contact = 10000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 16
        self.damping = 1.0

    def init_objects(self):
        self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, 0, 4e1, 0, False, self.cloth_M))
        self.tot_NV = self.cloths[0].NV

    def init(self):
        c = self.cloths[0]
        c.init(0.0, 0.0, 0.0)
        if self._perturb:
            i, j = np.meshgrid(np.arange(c.N + 1), np.arange(c.M + 1), indexing="ij")
            p = c.pos.to_numpy()
            p[:, 2] = (self._perturb * np.sin(7.0 * i) * np.cos(5.0 * j)).reshape(-1)
            c.pos.from_numpy(p)

    def contact_pairs(self):
        return []

    def set_frozen_kernel(self):
        if self._pin:
            c = self.cloths[0]
            fr = self.frozen.t.view(-1, 3)
            fr[c.offset + c.N * (c.M + 1): c.offset + (c.N + 1) * (c.M + 1)] = 1

    def action(self, step, delta_pos=None, delta_rot=None, delta_dis=None):
        pass
