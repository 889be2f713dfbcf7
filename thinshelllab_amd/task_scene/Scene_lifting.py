"""Lifting task: counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_lifting.py
(flat 15x15 cloth, a small free cube lying on it, three tactile pads driven by a 3-part gripper)."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 15  # Scene_lifting.py:203
    _plastic = 0

    def __init__(self, cloth_size=0.06, device="cuda:0", cloth_N=15):
        self._cN = cloth_N
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.cloths[0].k_angle[None] = 3.14

    def init_scene_parameters(self):
        # Scene_lifting.py:33-49
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 4
        self.elastic_size = [0.007, 0.015, 0.015, 0.015]
        self.elastic_Nx = 5
        self.elastic_Ny = 5
        self.elastic_Nz = 5
        self.cloth_N = self._cN
        self.cloth_M = self._cN
        self.k_contact = 500
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # Scene_lifting.py:51-66
        rho = 4e1
        for i in range(self.cloth_cnt):
            self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, i * ((self.cloth_N + 1) ** 2)))
        self.elastic_offset = ((self.cloth_N + 1) ** 2) * self.cloth_cnt
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz, 20000.0))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_lifting.py:77-85
        self.cloths[0].init(-0.03, -0.03, 0.)
        self.elastics[0].init(-0.025, -0.005, 0.0003)
        self.elastics[1].init(0.01, 0., 0.0079, True)
        self.elastics[2].init(0., -0.015, -0.0079, False)
        self.elastics[3].init(0., 0.015, -0.0079, False)
        self.gripper.init(self, np.array([[0.01, 0., 0.0079], [0., -0.015, -0.0079], [0., 0.015, -0.0079]]))

    def reset_pos(self):
        self.init()

    def init_property(self):
        # Scene_lifting.py:87-103: cloth and pads weightless, the cube feels gravity
        super().init_property()
        for c in self.cloths:
            c.gravity.t.zero_()
        self.elastics[0].gravity.t.copy_(torch.as_tensor(np.asarray(self.gravity[None], dtype=np.float64)))
        for e in self.elastics[1:]:
            e.gravity.t.zero_()

    def contact_pairs(self):
        # Scene_lifting.py:114-130
        pairs = []
        for c in self.cloths:
            for e in self.elastics:
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, None))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, None))
        return pairs

    def set_frozen_kernel(self):
        # Scene_lifting.py:131-150
        fr = self.frozen.t.view(-1, 3)
        for e in self.elastics[1:]:
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    def compute_reward(self):
        # Scene_lifting.py:152-159
        e = self.elastics[0]
        d = e.F_x.to_numpy() - e.F_ox.to_numpy()
        return float(-(((d[:, 0] + 0.025 + 0.012) ** 2) + ((d[:, 1] + 0.005 + 0.012) ** 2) + ((d[:, 2] - 0.0003) ** 2)).sum())

    def action(self, step, delta_pos, delta_rot):
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
