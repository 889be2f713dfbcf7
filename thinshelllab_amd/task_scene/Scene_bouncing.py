"""Bouncing task (system identification of the bending stiffness): counterpart of ``Scene`` in
/root/reference/code/task_scene/Scene_bouncing.py -- a 15x15 sheet with two pre-bent hinge rows ("bridge") dropped onto the frozen
table under gravity, dt = 2 ms, plastic hinges."""
from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 1000  # base time_step (BaseScene.py:1327-1370)
    _plastic = 1        # timestep_finish calls update_ref_angle (Scene_bouncing.py:121-125)

    def __init__(self, cloth_size=0.06, device="cuda:0"):
        self._first_init = True
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., -9.8]
        self.cloths[0].k_angle[None] = 3.14

    def init_scene_parameters(self):
        # Scene_bouncing.py:38-55
        self.dt = 2e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 1
        self.elastic_size = [0.07]
        self.elastic_Nx = 9
        self.elastic_Ny = 9
        self.elastic_Nz = 2
        self.cloth_N = 15
        self.cloth_M = 15
        self.k_contact = 40000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # Scene_bouncing.py:66-81
        rho = 4e1
        self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, 0))
        self.elastic_offset = (self.cloth_N + 1) * (self.cloth_M + 1)
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_bouncing.py:83-86 (z = 0.39 mm: inside the contact shell of the table)
        self.cloths[0].init(-0.03, -0.03, 0.00039)
        self.elastics[0].init(-0.035, -0.035, -0.00875)
        self.cloths[0].init_ref_angle_bridge()

    def reset_pos(self):
        # Scene_bouncing.py:88-91 (z = 3.9 mm: ten times higher than init -- kept as in the reference)
        self.cloths[0].init(-0.03, -0.03, 0.0039)
        self.elastics[0].init(-0.035, -0.035, -0.00875)
        self.cloths[0].init_ref_angle_bridge()

    def contact_pairs(self):
        # Scene_bouncing.py:93-99
        pairs = []
        for c in self.cloths:
            for e in self.elastics:
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, None))
        return pairs

    def set_frozen_kernel(self):
        # Scene_bouncing.py:101-107
        fr = self.frozen.t.view(-1, 3)
        e0 = self.elastics[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1

    def compute_reward(self):
        # Scene_bouncing.py:109-117
        c = self.cloths[0]
        z = c.pos.to_numpy()[:, 2]
        row = (self._np_arange(c.NV) // (c.M + 1))
        return float(z[(row == 5) | (row == 10)].sum())

    @staticmethod
    def _np_arange(n):
        import numpy as np
        return np.arange(n)
