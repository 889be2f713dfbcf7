"""Folding task: counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_folding.py
(cloth 15x3 pre-folded over itself, frozen table, one tactile pad driven by a single-part gripper;
plastic hinge rest angles).  ``cloth_N`` / ``cloth_M`` scale the grid for the BASELINE configs
(SURVEY.md section 8d cfg3): rows keep their proportions (fold rows, reward rows scale with N/15)."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401  (Body re-exported like the reference module)
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50   # Scene_folding.py:294
    _plastic = 1       # timestep_finish calls update_ref_angle (Scene_folding.py:227-231)

    def __init__(self, cloth_size=0.06, device="cuda:0", cloth_N=15, cloth_M=3):
        self._cN, self._cM = cloth_N, cloth_M
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., 0.]
        self.cloths[0].k_angle[None] = 0.5

    @property
    def row_scale(self):
        return self.cloth_N // 15

    def init_scene_parameters(self):
        # Scene_folding.py:34-50
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 2
        self.elastic_size = [0.07, 0.015]
        self.elastic_Nx = 9
        self.elastic_Ny = 9
        self.elastic_Nz = 2
        self.cloth_N = self._cN
        self.cloth_M = self._cM
        self.k_contact = 10000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000 if self._cN <= 15 else 200000
        self.damping = 1.0

    def init_objects(self):
        # Scene_folding.py:61-76
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
        # Scene_folding.py:78-87.  For scaled grids the fold keeps the native physical radius r0 = 0.1/15 m.
        half_curve_num = 2
        c = self.cloths[0]
        s = self.row_scale
        if s == 1:
            c.init_fold(-0.07, -0.01, 0.0004, half_curve_num)
            r = c.grid_len * (half_curve_num * 2 - 1) / 3.1415
        else:
            r0 = self.cloth_size / 15.0
            n_arc = max(int(round(3.1415 * r0 / c.grid_len)), 2)
            L = 6 * s
            c.init_fold(-0.07, -0.01, 0.0004, half_curve_num, rows=15 * s, L=L, R=L + n_arc, r=r0)
            r = r0 * 3 / 3.1415
        self.elastics[0].init(-0.035, -0.035, -0.00875)
        x = -0.07 + (7 + half_curve_num) / 16 * 0.1 - r * 0.86 + 0.005
        self.elastics[1].init(x, 0.0, 2 * r + 0.0079, True)
        self.gripper.init(self, np.array([[x, 0.0, 2 * r + 0.0079]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_folding.py:99-108
        pairs = []
        for c in self.cloths:
            for e in self.elastics:
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, None))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, None))
        return pairs

    def set_frozen_kernel(self):
        # Scene_folding.py:110-127
        fr = self.frozen.t.view(-1, 3)
        e0, e1, c = self.elastics[0], self.elastics[1], self.cloths[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        fr[e1.offset:e1.offset + e1.n_verts][torch.as_tensor(e1.bound_mask())] = 1
        fr[c.offset + c.N * (c.M + 1): c.offset + (c.N + 1) * (c.M + 1)] = 1

    def _reward_rows(self, row_a, row_b):
        c = self.cloths[0]
        f2v = c.f2v.to_numpy(); cf = c.counter_face.to_numpy(); cp = c.counter_point.to_numpy()
        fi, l = np.nonzero(cf > np.arange(c.NF)[:, None])
        p1 = f2v[fi, l]; p2 = f2v[cf[fi, l], cp[fi, l]]
        m = (p1 // (c.M + 1) == row_a) & (p2 // (c.M + 1) == row_b)
        return fi[m], l[m]

    def fold_rows(self):
        s = self.row_scale
        return ((6 * s, 6 * s + 2), (6 * s + 1, 6 * s + 3)) if s > 1 else ((6, 8), (7, 9))

    def compute_reward(self, curve7, curve8):
        # Scene_folding.py:129-147
        ra = self.cloths[0].ref_angle.to_numpy()
        ret = 0.0
        (a7, b7), (a8, b8) = self.fold_rows()
        fi, l = self._reward_rows(a7, b7); ret += float((-ra[fi, l] * curve7).sum())
        fi, l = self._reward_rows(a8, b8); ret += float((-ra[fi, l] * curve8).sum())
        return ret

    def compute_reward_8(self):
        return self.compute_reward(-1, 1)

    def compute_reward_7(self):
        return self.compute_reward(1, -1)

    def action(self, step, delta_pos, delta_rot):
        # Scene_folding.py:215-225
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
