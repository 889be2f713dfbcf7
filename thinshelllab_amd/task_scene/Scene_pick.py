"""Pick-and-fold task: counterpart of ``Scene`` in /root/reference/code/task_scene/Scene_pick.py -- a 16x16 sheet lying on an arched,
frozen 16x16x2 table under gravity, two tactile pads of a two-part single gripper above it; plastic hinges, Newton cap 50."""
import numpy as np
import torch

from ..engine.BaseScene import BaseScene, Body  # noqa: F401
from ..engine.model_elastic_offset import Elastic
from ..engine.model_elastic_tactile import Elastic as tactile
from ..engine.model_fold_offset import Cloth


class Scene(BaseScene):
    _newton_cap = 50   # Scene_pick.py:238-283
    _plastic = 1       # timestep_finish calls update_ref_angle (:187-191)

    def __init__(self, cloth_size=0.06, device="cuda:0"):
        super().__init__(cloth_size=cloth_size, enable_gripper=False, device=device)
        self.gravity[None] = [0., 0., -9.8]
        self.cloths[0].k_angle[None] = 0.5

    def init_scene_parameters(self):
        # Scene_pick.py:29-45
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 3
        self.elastic_size = [0.06, 0.015, 0.015]
        self.elastic_Nx = 16
        self.elastic_Ny = 16
        self.elastic_Nz = 2
        self.cloth_N = 16
        self.cloth_M = 16
        self.k_contact = 10000
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_objects(self):
        # BaseScene.init_objects (BaseScene.py:196-211): square sheet, box table, tactile pads
        rho = 4e1
        self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, 0))
        self.elastic_offset = (self.cloth_N + 1) * (self.cloth_N + 1)
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init(self):
        # Scene_pick.py:56-62
        self.cloths[0].init(-0.03, -0.03, 0.0004)
        self.elastics[0].init_arch(-0.03, -0.03, -0.008, 0.004)
        self.elastics[1].init(-0.025, 0., 0.0079, True)
        self.elastics[2].init(0.025, 0., 0.0079, True)
        self.gripper.init(self, np.array([[-0.025, 0., 0.0079], [0.025, 0., 0.0079]]))

    def reset_pos(self):
        self.init()

    def contact_pairs(self):
        # Scene_pick.py:72-89 (table: mu = 0.1)
        pairs = []
        for c in self.cloths:
            for j, e in enumerate(self.elastics):
                mu = 0.1 if j == 0 else None
                pairs.append((c.body_idx, e.offset, e.offset + e.n_verts, mu))
                pairs.append((e.body_idx, c.offset, c.offset + c.NV, mu))
        return pairs

    def static_friction_loss(self, analy_grad, step, constraints=None, pos=None):
        """Scene_pick.py:193-236 (call site commented out in the reference, analytic_grad_single.py:231): for the sliding
        constraints recorded after the table pair (i >= nc1, i.e. every constraint that does not touch elastics[0]) the friction
        force per unit pressure, pushed onto the contact normals of the PREVIOUS step:
        pos_grad[step - 1, idx[i2]] += -dfdp w1[i2] n k_contact f_loss_ratio, dfdp = w1[i1] (T^T (u k f1(r)))[j1] / (k / mu) summed
        over (i1, j1), w1 = (w0, w1, w2, -1).  The engine appends constraints in no fixed order, so the nc1 split is taken from the
        vertices: a constraint of the table pair has a vertex inside elastics[0]."""
        import numpy as np
        c, T, u, r, sl = self._friction_slip(constraints, pos)
        e0 = self.elastics[0]
        table = ((c["idx"] >= e0.offset) & (c["idx"] < e0.offset + e0.n_verts)).any(1)
        sl = sl & ~table
        if not sl.any():
            return
        h = self.eps_v * self.dt
        f1 = np.where(r > h, 1.0 / np.maximum(r, 1e-300), -r / h ** 2 + 2.0 / h)   # BaseScene.f1 (:463-469)
        g1 = np.einsum("nij,ni->nj", T, u * (c["k"] * f1)[:, None])
        w1 = np.concatenate([c["w"], -np.ones((len(r), 1))], 1)
        pressure = c["k"] / c["mu"]
        dfdp = (w1.sum(1) * g1.sum(1)) / pressure          # the reference accumulates over every (i1, j1)
        g = analy_grad.pos_grad.to_numpy()
        add = -(dfdp * self.k_contact * analy_grad.f_loss_ratio)[:, None, None] * w1[:, :, None] * c["n"][:, None, :]
        for i2 in range(4):
            np.add.at(g[step - 1], c["idx"][sl, i2], add[sl, i2])
        analy_grad.pos_grad.from_numpy(g)

    def set_frozen_kernel(self):
        # Scene_pick.py:91-109
        fr = self.frozen.t.view(-1, 3)
        e0 = self.elastics[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        for e in self.elastics[1:]:
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    # ---- rewards (Scene_pick.py:119-172)
    def _row(self, k):
        c = self.cloths[0]
        return np.arange(c.NV) // (c.M + 1) == k

    def compute_reward(self):
        return float(self.cloths[0].pos.to_numpy()[self._row(8), 2].sum())

    def compute_reward_deliver(self, analy_grad):
        c = self.cloths[0]
        d = c.pos.to_numpy() - analy_grad.pos_buffer.to_numpy()[69, c.offset:c.offset + c.NV] - 0.01
        return float(-(d ** 2).sum())

    def compute_reward_pick_fold(self):
        c = self.cloths[0]
        fi, l, theta = c._hinge_angles()
        f2v = c.f2v.to_numpy(); cf = c.counter_face.to_numpy(); cp = c.counter_point.to_numpy()
        r1 = f2v[fi, l] // (c.M + 1)
        r2 = f2v[cf[fi, l], cp[fi, l]] // (c.M + 1)
        m = (r1 == 7) & (r2 == 9)
        ra = c.ref_angle.to_numpy()
        return float(ra[fi[m], l[m]].sum() + 0.01 * theta[m].sum())

    def compute_reward_pick_and_fold(self):
        return self.compute_reward_pick_fold() + self.compute_reward()

    def action(self, step, delta_pos, delta_rot):
        # Scene_pick.py:174-185
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)
