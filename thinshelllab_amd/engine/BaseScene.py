"""Scene state + implicit-Euler driver: host-side counterpart of ``BaseScene``
(/root/reference/code/engine/BaseScene.py).

Python keeps what the task scenes / Grad / trajopt scripts touch (fields, bodies, gripper, frozen sets) and
hands the hot path to ``libtsl_hip.so`` through ``TslContext``:
  time_step            -> tsl_step            (BaseScene.py:1327-1370)
  compute_energy       -> tsl_energy          (:427-451)
  compute_residual_and_Hessian / compute_Hessian -> tsl_assemble (:976-1052)
  H.solve              -> tsl_solve           (sparse_solver.py:85-105)
State lives once, in global HBM tensors; per-body fields are views (no pushup/pushdown copies).
"""
from dataclasses import dataclass

import numpy as np
import torch

from . import gripper_single
from .field import Field, ScalarField
from .gripper_tactile import gripper
from .model_elastic_offset import Elastic
from .model_elastic_tactile import Elastic as tactile
from .model_fold_offset import Cloth


@dataclass
class Body:
    v_start: int
    v_end: int
    f_start: int
    f_end: int


class _SystemMatrix:
    """Stand-in for ``SparseMatrix`` (sparse_solver.py): the matrix itself lives inside the context."""

    def __init__(self, sys):
        self._sys = sys
        self.n = sys.tot_NV * 3

    def clear_all(self):
        pass

    def solve(self, b):
        x, _ = self._sys._ensure_ctx().solve(b.contiguous().to(self._sys.device, torch.float64))
        return x

    def to_csr(self):
        return self._sys._ensure_ctx().matrix_csr()


class BaseScene:
    _newton_cap = 1000   # BaseScene.py:1342
    _plastic = 0         # base timestep_finish does not update ref angles (:1321-1325)

    def __init__(self, cloth_size=0.1, dt=5e-3, enable_gripper=True, device="cuda:0"):
        # BaseScene.py:31-60
        self.dt = dt
        self.h = self.dt
        self.cloth_cnt = 2
        self.elastic_cnt = 3
        self.cloth_size = cloth_size
        self.elastic_size = [0.06, 0.015, 0.015]
        self.cloth_N = 31
        self.elastic_Nx = 16
        self.elastic_Ny = 16
        self.elastic_Nz = 2
        self.enable_gripper = enable_gripper
        self.k_contact = 1000
        self.eps_contact = 0.001
        self.eps_v = 0.01
        self.max_n_constraints = 100000
        self.damping = 1.0
        self.extra_obj = False
        self.effector_cnt = -1
        self.grid_h = 0.003
        if isinstance(device, str) and device.startswith("cuda") and not torch.cuda.is_available():
            device = "cpu"  # host-only use (tables, poses); any engine call will raise (no CPU path)
        self.device = torch.device(device)
        self._ctx = None
        self._dirty = set()
        self.last_stats = {}

        self.init_scene_parameters()
        if self.effector_cnt == -1:
            self.effector_cnt = self.elastic_cnt
        self.gravity = ScalarField([0.0, 0.0, -9.8], lambda f: self._refresh_gravity())
        self.mu_cloth_elastic = ScalarField(1.0, lambda f: self._set_param("mu_cloth_elastic", f.value))
        self.cloths = []
        self.elastics = []
        self.tot_NV = ((self.cloth_N + 1) ** 2) * self.cloth_cnt
        self.init_objects()
        NV = self.tot_NV
        dev = self.device
        # BaseScene.py:69-88
        self.pos = Field(torch.zeros((NV, 3), dtype=torch.float64, device=dev))
        self.vel = Field(torch.zeros((NV, 3), dtype=torch.float64, device=dev))
        self.prev_pos = Field(torch.zeros((NV, 3), dtype=torch.float64, device=dev))
        self.x1 = Field(torch.zeros((NV, 3), dtype=torch.float64, device=dev))
        self.mass = Field(torch.zeros(NV, dtype=torch.float64))
        self.tot_NF = 0
        for c in self.cloths:
            c.offset_faces = self.tot_NF
            self.tot_NF += c.NF
        for e in self.elastics:
            e.offset_faces = self.tot_NF
            self.tot_NF += e.n_surfaces
        self.frozen = Field(torch.zeros(NV * 3, dtype=torch.int32), lambda f: self._dirty.add("frozen"))
        self.faces = Field(torch.zeros((self.tot_NF, 3), dtype=torch.int32))
        self.border_flag = Field(torch.zeros(NV, dtype=torch.int32), lambda f: self._dirty.add("border"))
        self.ext_force = Field(torch.zeros((NV, 3), dtype=torch.float64), lambda f: self._dirty.add("ext_force"))
        self.x32 = Field(torch.zeros((NV, 3), dtype=torch.float32))
        self.f_vis = Field(torch.zeros(self.tot_NF * 3, dtype=torch.int32))
        # BaseScene.py:91-99
        self.body_list = []
        for c in self.cloths:
            self.body_list.append(Body(c.offset, c.offset + c.NV, c.offset_faces, c.offset_faces + c.NF))
        for e in self.elastics:
            self.body_list.append(Body(e.offset, e.offset + e.n_verts, e.offset_faces, e.offset_faces + e.n_surfaces))
        for i, c in enumerate(self.cloths):
            c.body_idx = i; c._idx = i; c._sys = self
        for i, e in enumerate(self.elastics):
            e.body_idx = i + self.cloth_cnt; e._sys = self
        self.nc = ScalarField(0, dtype=torch.int32)
        self.E = ScalarField(0.0)
        self.F = Field(torch.zeros(NV * 3, dtype=torch.float64, device=dev))
        self.H = _SystemMatrix(self)
        self.tmp_z_frozen = Field(torch.zeros(NV * 3, dtype=torch.float64, device=dev))
        # plastic rest angles of all cloths, one tensor; each cloth's field is a view
        n_cf = sum(c.NF for c in self.cloths)
        self._ref_angle = torch.zeros((max(n_cf, 1), 3), dtype=torch.float64, device=dev)
        self._bind_bodies()
        # gripper (BaseScene.py:166-172)
        if enable_gripper:
            self.gripper = gripper(self.dt, self.elastics[1].n_verts, self.elastics[1].frozen_cnt, self.elastics[1].surf_point,
                                   int((self.effector_cnt - 1) // 2))
        elif self.elastic_cnt > 1:
            self.gripper = gripper_single.gripper(self.dt, self.elastics[1].n_verts, self.elastics[1].frozen_cnt, self.elastics[1].surf_point,
                                                  self.effector_cnt - 1)
        self.action_dim = 3 * (self.effector_cnt - 1)
        if not enable_gripper:
            self.action_dim = int(6 * (self.effector_cnt - 1))
        if self.effector_cnt - 1 > 0:
            self.n_obs_cloth = 4
            self.n_obs_elastic = 16
            self.n_sample_cloth = self.cloths[0].N // 4   # BaseScene.py:187-191
            self.m_sample_cloth = self.cloths[0].M // 4
            self.obs_dim = (self.n_obs_cloth * self.n_obs_cloth * self.cloth_cnt + self.n_obs_elastic * self.elastic_cnt) * 6 + 7 * self.gripper.n_part
            self.observation = Field(torch.zeros(self.obs_dim, dtype=torch.float64))
            self.tot_force = Field(torch.zeros((self.effector_cnt - 1, 3), dtype=torch.float64))

    # ------------------------------------------------------------------ construction helpers
    def _bind_bodies(self):
        """make per-body state fields views of the global arrays (replaces pushup/pushdown_property)."""
        fs = 0
        for c in self.cloths:
            sl = slice(c.offset, c.offset + c.NV)
            for name, glob in (("pos", self.pos), ("vel", self.vel), ("prev_pos", self.prev_pos)):
                old = getattr(c, name).t
                glob.t[sl].copy_(old.to(glob.t.device))
                getattr(c, name).t = glob.t[sl]
            old = c.ref_angle.t
            self._ref_angle[fs:fs + c.NF].copy_(old.to(self._ref_angle.device))
            c.ref_angle.t = self._ref_angle[fs:fs + c.NF]
            fs += c.NF
        for e in self.elastics:
            sl = slice(e.offset, e.offset + e.n_verts)
            for name, glob in (("F_x", self.pos), ("F_v", self.vel), ("F_x_prev", self.prev_pos)):
                old = getattr(e, name).t
                glob.t[sl].copy_(old.to(glob.t.device))
                getattr(e, name).t = glob.t[sl]

    def init_objects(self):
        # BaseScene.py:196-211
        rho = 4e1
        for i in range(self.cloth_cnt):
            self.cloths.append(Cloth(self.cloth_N, self.dt, self.cloth_size, self.tot_NV, rho, i * ((self.cloth_N + 1) ** 2)))
        self.elastic_offset = ((self.cloth_N + 1) ** 2) * self.cloth_cnt
        tmp_tot = self.elastic_offset
        self.elastics.append(Elastic(self.dt, self.elastic_size[0], tmp_tot, self.elastic_Nx, self.elastic_Ny, self.elastic_Nz))
        tmp_tot += self.elastic_Nx * self.elastic_Ny * self.elastic_Nz
        for i in range(1, self.elastic_cnt):
            self.elastics.append(tactile(self.dt, tmp_tot, self.elastic_size[i] / 0.03))
            tmp_tot += self.elastics[i].n_verts
        self.tot_NV = tmp_tot

    def init_scene_parameters(self):
        # BaseScene.py:213-225
        self.dt = 5e-3
        self.h = self.dt
        self.cloth_cnt = 1
        self.elastic_cnt = 3
        self.elastic_size = [0.06, 0.015, 0.015]
        self.cloth_N = 15
        self.k_contact = 500
        self.eps_contact = 0.0004
        self.eps_v = 0.01
        self.max_n_constraints = 10000
        self.damping = 1.0

    def init_all(self):
        self.init()
        self.init_property()
        self.set_frozen()
        self.set_ext_force()
        self.update_visual()

    def init(self):
        # BaseScene.py:235-242
        self.cloths[0].init(-0.03, -0.03, 0.000399)
        self.elastics[0].init(-0.03, -0.03, -0.004)
        self.elastics[1].init(-0.02, 0., 0.0105, True)
        self.elastics[2].init(-0.02, 0., -0.0105, False)
        self.gripper.init(self, np.array([[-0.02, 0., 0.0]]))

    def reset_pos(self):
        self.init()

    def reset(self):
        # BaseScene.py:252-268
        self.reset_pos()
        self.set_ext_force()
        self.set_frozen()
        self.update_visual()
        if self._ctx is not None:
            self._ctx.contact_reset()

    def init_property(self):
        # BaseScene.py:361-383: gravity per body, masses, faces; (re)creates the engine context
        g = np.asarray(self.gravity[None], dtype=np.float64)
        for c in self.cloths:
            c.gravity.t.copy_(torch.as_tensor(g))
        if self.elastic_cnt > 0:
            self.elastics[0].gravity.t.copy_(torch.as_tensor(g))
        for i in range(1, self.effector_cnt):
            self.elastics[i].gravity.t.zero_()
        for i in range(self.effector_cnt, self.elastic_cnt):
            self.elastics[i].gravity.t.copy_(torch.as_tensor(g))
        m = np.zeros(self.tot_NV)
        for c in self.cloths:
            m[c.offset:c.offset + c.NV] = c.mass
        for e in self.elastics:
            m[e.offset:e.offset + e.n_verts] = e.F_m.to_numpy()
        self.mass.from_numpy(m)
        f = np.zeros((self.tot_NF, 3), np.int32)
        for c in self.cloths:
            f[c.offset_faces:c.offset_faces + c.NF] = c.f2v.to_numpy() + c.offset
        for e in self.elastics:
            f[e.offset_faces:e.offset_faces + e.n_surfaces] = e.f2v.to_numpy() + e.offset
        self.faces.from_numpy(f)
        self.f_vis.from_numpy(f.reshape(-1))
        self._close_ctx()

    def _gravity_array(self):
        g = np.zeros((self.tot_NV, 3))
        for c in self.cloths:
            g[c.offset:c.offset + c.NV] = c.gravity.to_numpy()
        for e in self.elastics:
            g[e.offset:e.offset + e.n_verts] = e.gravity.to_numpy()
        return g

    def _refresh_gravity(self):
        self._dirty.add("gravity")

    def _ext_force_array(self):
        f = self.ext_force.to_numpy().copy()
        for c in self.cloths:
            f[c.offset:c.offset + c.NV] += c.manipulate_force.to_numpy()
        for e in self.elastics:
            f[e.offset:e.offset + e.n_verts] += e.ext_force.to_numpy()
        return f

    # ------------------------------------------------------------------ static-friction loss (BaseScene.py:732-775)
    def _friction_slip(self, constraints=None, pos=None):
        """tangential slip of every constraint of the current step: (constraints, u (nc, 2), r (nc,), sliding mask r > 0.9 dt eps_v)
        -- the quantities both static_friction_loss variants start from (BaseScene.py:745-750, Scene_pick.py:205-211).  Host side:
        the reference never calls these kernels (analytic_grad_single.py:231 is commented out), they complete the named surface."""
        import numpy as np
        c = self._ensure_ctx().constraints() if constraints is None else constraints
        x = self.pos.to_numpy() if pos is None else pos
        idx, w = c["idx"], c["w"]
        T = c["T"].reshape(-1, 2, 3)
        x_c = (x[idx[:, :3]] * w[:, :, None]).sum(1)
        dx = x[idx[:, 3]] - x_c - c["dx0"]
        u = np.einsum("nij,nj->ni", T, dx)
        r = np.linalg.norm(u, axis=1)
        return c, T, u, r, r > self.dt * self.eps_v * 0.9

    def static_friction_loss(self, analy_grad, step, constraints=None, pos=None):
        """BaseScene.py:732-775: pos_grad[step, idx[i1]] += u_3d w1[i1] f_loss_ratio k for every sliding constraint,
        u_3d = T^T u, w1 = (-w0, -w1, -w2, 1)"""
        import numpy as np
        c, T, u, r, sl = self._friction_slip(constraints, pos)
        if not sl.any():
            return
        u3 = np.einsum("nij,ni->nj", T, u)
        w1 = np.concatenate([-c["w"], np.ones((len(r), 1))], 1)
        g = analy_grad.pos_grad.to_numpy()
        add = u3[:, None, :] * w1[:, :, None] * (analy_grad.f_loss_ratio * c["k"])[:, None, None]
        for i1 in range(4):
            np.add.at(g[step], c["idx"][sl, i1], add[sl, i1])
        analy_grad.pos_grad.from_numpy(g)

    # contact relationship of contact_analysis (BaseScene.py:818-835); scenes override
    def contact_pairs(self):
        pairs = []
        for i in range(self.cloth_cnt):
            for j in range(self.cloth_cnt):
                if abs(i - j) == 1:
                    pairs.append((self.cloths[i].body_idx, self.cloths[j].offset, self.cloths[j].offset + self.cloths[j].NV, 0.1))
                    pairs.append((self.cloths[j].body_idx, self.cloths[i].offset, self.cloths[i].offset + self.cloths[i].NV, 0.1))
        for i in range(self.cloth_cnt):
            for j in range(self.elastic_cnt):
                mu = None if j != 0 else 0.2
                pairs.append((self.cloths[i].body_idx, self.elastics[j].offset, self.elastics[j].offset + self.elastics[j].n_verts, mu))
                pairs.append((self.elastics[j].body_idx, self.cloths[i].offset, self.cloths[i].offset + self.cloths[i].NV, mu))
        return pairs

    # ------------------------------------------------------------------ engine context
    def _close_ctx(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def _ensure_ctx(self):
        if self._ctx is None:
            from ..context import TslContext
            self._ctx = TslContext(
                tot_NV=self.tot_NV, dt=self.dt, mass=self.mass.to_numpy(), gravity=self._gravity_array(), frozen=self.frozen.to_numpy(),
                cloths=[c._desc() for c in self.cloths], elastics=[e._desc() for e in self.elastics], faces=self.faces.to_numpy(),
                bodies=[(b.v_start, b.v_end, b.f_start, b.f_end) for b in self.body_list], pairs=self.contact_pairs(),
                k_contact=self.k_contact, eps_contact=self.eps_contact, eps_v=self.eps_v, damping=self.damping,
                max_n_constraints=self.max_n_constraints, grid_h=self.grid_h, device=str(self.device))
            self._ctx.set_param("mu_cloth_elastic", self.mu_cloth_elastic.value)
            if hasattr(self, "mu_cloth_cloth"):
                self._ctx.set_param("mu_cloth_cloth", self.mu_cloth_cloth.value)
            if getattr(self, "grid_extent", None):
                self._ctx.set_param("grid_extent", self.grid_extent)
            self._ctx.set_param("newton_cap", self._newton_cap)
            self._ctx.set_param("plastic", self._plastic)
            self._ctx.set_ext_force(self._ext_force_array())
            self._dirty.clear()
        if self._dirty:
            if "frozen" in self._dirty:
                self._ctx.set_frozen(self.frozen.to_numpy())
            if "border" in self._dirty:
                self._ctx.set_border(self.border_flag.to_numpy())
            if "ext_force" in self._dirty:
                self._ctx.set_ext_force(self._ext_force_array())
            if "gravity" in self._dirty:
                self._ctx.set_gravity(self._gravity_array())
            self._dirty.clear()
        return self._ctx

    def _set_param(self, key, value):
        if self._ctx is not None:
            self._ctx.set_param(key, value)

    # ------------------------------------------------------------------ frozen / external force
    def set_frozen_kernel(self):
        # BaseScene.py:1445-1463
        fr = self.frozen.t.view(-1, 3)
        e0 = self.elastics[0]
        fr[e0.offset:e0.offset + e0.n_verts] = 1
        for j in (1, 2):
            e = self.elastics[j]
            fr[e.offset:e.offset + e.n_verts][torch.as_tensor(e.bound_mask())] = 1

    def set_frozen(self):
        self.frozen.t.zero_()
        self.set_frozen_kernel()
        self._dirty.add("frozen")

    def set_ext_force(self):
        self.ext_force.t.zero_()
        for c in self.cloths:
            c.clear_manipulation()
        self._dirty.add("ext_force")

    def update_visual(self):
        self.x32.t.copy_(self.pos.t.detach().to("cpu", torch.float32))

    # ------------------------------------------------------------------ stepping
    def push_down_pos(self):
        pass  # per-body arrays are views of self.pos

    def push_down_vel(self):
        pass

    def pushup_property(self, dest, src, offset):
        if dest.t.data_ptr() != src.t.data_ptr():  # already aliased when src is a bound body field
            dest.t[offset:offset + src.t.shape[0]].copy_(src.t.to(dest.t.device))

    def _state(self):
        return self.pos.t, self.prev_pos.t, self.vel.t, self._ref_angle

    def compute_energy(self):
        e = self._ensure_ctx().energy(*self._state())
        self.E[None] = e
        return e

    def compute_residual_and_Hessian(self, check_PD=False, iter=0, spd=True):
        self._ensure_ctx().assemble(*self._state(), spd=spd, grad=self.F.t)
        return True

    def compute_Hessian(self, spd=True):
        self._ensure_ctx().assemble(*self._state(), spd=spd, grad=None)

    def calc_vn(self):
        pass  # part of contact detection inside the context

    def contact_analysis(self):
        pass

    def update_ref_angle(self):
        self._ensure_ctx().update_ref_angle(self.pos.t, self._ref_angle)

    def time_step(self, f_contact, frame_idx, force_stick=True):
        """BaseScene.time_step (BaseScene.py:1327-1370).  ``f_contact`` is ``geometry.projection_query``; the
        detection it stands for runs inside tsl_step (None disables contact)."""
        ctx = self._ensure_ctx()
        ctx.set_param("contact", 0.0 if f_contact is None else 1.0)
        st = ctx.step(*self._state())
        self.last_stats = st
        self.nc[None] = st["nc"]
        self.E[None] = st["energy"]
        return st

    def action(self, step, delta_pos, delta_rot, delta_dis=None):
        # BaseScene.py:1489-1500
        self.gripper.step_simple(delta_pos, delta_rot)
        self.gripper.update_bound(self)

    # ------------------------------------------------------------------ adjoint helpers (BaseScene.py:270-315)
    def copy_pos_kernel(self, target_pos, step):
        self.pos.t.copy_(target_pos.t[step])

    def copy_prev_pos_kernel(self, target_pos, step):
        self.prev_pos.t.copy_(target_pos.t[step - 1])

    def copy_pos_only(self, target_pos, step):
        self.copy_pos_kernel(target_pos, step)
        self.copy_prev_pos_kernel(target_pos, step + 1)

    def copy_pos_and_refangle(self, analy_grad, step):
        self.copy_pos_kernel(analy_grad.pos_buffer, step)
        self.copy_prev_pos_kernel(analy_grad.pos_buffer, step)
        rb = analy_grad.ref_angle_buffer.t[step - 1].reshape(-1, 3)
        self._ref_angle[: rb.shape[0]].copy_(rb)

    def copy_refangle(self, analy_grad, step):
        rb = analy_grad.ref_angle_buffer.t[step].reshape(-1, 3)
        self._ref_angle[: rb.shape[0]].copy_(rb)

    # ------------------------------------------------------------------ misc surface kept for the scripts
    def compute_reward(self):
        c = self.cloths[0]
        return float(c.pos.t[:, 2].sum().item())

    def check_pos_nan(self):
        return bool(torch.isnan(self.pos.t).any().item())

    def gather_force(self):
        """BaseScene.py:1541-1549: sum of Elastic.get_force over the gripper-driven vertices of every effector pad"""
        f = self._ensure_ctx().elastic_force(self.pos.t, torch.empty_like(self.pos.t)).cpu().numpy()
        tot = np.zeros((self.effector_cnt - 1, 3))
        for j in range(1, self.effector_cnt):
            e = self.elastics[j]
            tot[j - 1] = f[e.offset:e.offset + e.n_verts][e.bound_mask()].sum(0)
        self.tot_force.from_numpy(tot)
        return tot

    def check_early_stop(self, frame, ifprint=False, RL=False):
        # BaseScene.py:1559-1584
        if self.check_pos_nan():
            if ifprint:
                print("exist nan")
            return True
        if self.effector_cnt - 1 <= 0:
            return False
        tot = self.gather_force()
        for i in range(self.effector_cnt - 1):
            if (np.abs(tot[i]) > 10).any():
                if ifprint:
                    print("too much force")
                return True
            if np.sqrt((tot[i] ** 2).sum()) < 0.2 and frame > 10 and not RL:
                if ifprint:
                    print("no contact")
                return True
        return False

    def get_observation_kernel(self):
        # BaseScene.py:1586-1619 (cloth samples index pos[jj * cloth_N + kk] -- cloth_N, not M + 1 -- and the elastic sample
        # index (n_verts // n_obs_elastic) * j - 1 is -1 for j = 0, i.e. the last vertex: both kept; on non-square cloths such as
        # the 15x7 balancing sheet the cloth index runs past NV -- an unchecked out-of-range read in the reference, zeros here)
        obs = np.zeros(self.obs_dim)
        no = self.n_obs_cloth
        for i, c in enumerate(self.cloths):
            x = c.pos.to_numpy(); v = c.vel.to_numpy()
            for j in range(no):
                for k in range(no):
                    xx = i * no * no + j * no + k
                    jj = self.n_sample_cloth // 2 + j * self.n_sample_cloth
                    kk = self.m_sample_cloth // 2 + k * self.m_sample_cloth
                    q = jj * self.cloth_N + kk
                    if q < c.NV:
                        obs[xx * 6:xx * 6 + 3] = x[q]
                        obs[xx * 6 + 3:xx * 6 + 6] = v[q]
        for i, e in enumerate(self.elastics):
            x = e.F_x.to_numpy(); v = e.F_v.to_numpy()
            for j in range(self.n_obs_elastic):
                xx = no * no * self.cloth_cnt + i * self.n_obs_elastic + j
                ii = (e.n_verts // self.n_obs_elastic) * j - 1
                obs[xx * 6:xx * 6 + 3] = x[ii]
                obs[xx * 6 + 3:xx * 6 + 6] = v[ii]
        base = (no * no * self.cloth_cnt + self.elastic_cnt * self.n_obs_elastic) * 6
        gp = self.gripper.pos.to_numpy(); gr = self.gripper.rot.to_numpy()
        for j in range(self.gripper.n_part):
            obs[base + j * 7: base + j * 7 + 3] = gp[j]
            obs[base + j * 7 + 3: base + j * 7 + 7] = gr[j]
        self.observation.from_numpy(obs)
        return obs

    def get_observation(self):
        return self.get_observation_kernel()

    def save_state(self, save_path):
        torch.save({'pos': self.pos.to_torch('cpu'), 'vel': self.vel.to_torch('cpu')}, save_path)

    def load_state(self, save_path):
        data = torch.load(save_path)
        self.pos.t.copy_(data['pos'].to(self.pos.t.device))
        self.vel.t.copy_(data['vel'].to(self.vel.t.device))
        self.update_ref_angle()   # BaseScene.py:1376-1384: the loaded pose also moves the plastic rest angles
        self.update_visual()
