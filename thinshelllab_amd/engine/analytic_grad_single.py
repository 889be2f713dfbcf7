"""Trajectory adjoint: host-side counterpart of ``Grad``
(/root/reference/code/engine/analytic_grad_single.py).

The tape (``pos_buffer``, ``ref_angle_buffer``) and the gradient buffers live in HBM; one reverse step
(``transfer_grad``, :217-257) is a single ``tsl_adjoint_step`` call -- contact re-detection, the un-projected
Hessian, one linear solve, the frozen-dof coupling and the three back-propagation kernels -- followed by the
(tiny) gripper reduction on the host (``get_gripper_grad``, :118-139).
"""
import numpy as np
import torch

from .field import Field


class Grad:
    def __init__(self, sys, tot_timestep, n_parts, friction_loss=False, f_loss_ratio=0.001, vertical_only=False):
        # analytic_grad_single.py:5-26
        self.n_part = n_parts
        self.tot_NV = sys.tot_NV
        dev = sys.device
        T = tot_timestep
        z = lambda *shape: Field(torch.zeros(shape, dtype=torch.float64, device=dev))
        self.pos_buffer = z(T, sys.tot_NV, 3)
        self.gripper_pos_buffer = Field(torch.zeros((T, max(n_parts, 1), 3), dtype=torch.float64))
        self.gripper_rot_buffer = Field(torch.zeros((T, max(n_parts, 1), 4), dtype=torch.float64))
        self.cloth_cnt = sys.cloth_cnt
        self.NF = sys.cloths[0].NF
        self.ref_angle_buffer = z(T, sys.cloth_cnt, self.NF, 3)
        self.dt = sys.dt
        self.pos_grad = z(T, sys.tot_NV, 3)
        self.x_hat_grad = z(sys.tot_NV * 3)
        self.gripper_grad = Field(torch.zeros((T, max(n_parts, 1), 6), dtype=torch.float64))
        self.angleref_grad = z(T, sys.cloth_cnt, self.NF, 3)
        self.mass = Field(torch.zeros(sys.tot_NV, dtype=torch.float64))
        self.tot_timestep = T
        self.damping = 1.0
        self.friction_loss = friction_loss
        self.f_loss_ratio = f_loss_ratio
        self.vertical_only = vertical_only
        self.last_stats = {}

    def reset(self):
        self.pos_buffer.fill(0)
        self.pos_grad.fill(0)
        self.angleref_grad.fill(0)

    def init_mass(self, sys):
        self.mass.copy_from(sys.mass)

    # :37-51
    def copy_pos(self, sys, step):
        self.pos_buffer.t[step].copy_(sys.pos.t)
        self.ref_angle_buffer.t[step].view(-1, 3).copy_(sys._ref_angle[: self.cloth_cnt * self.NF])
        if self.n_part > 0 and hasattr(sys, "gripper"):
            self.gripper_pos_buffer.t[step].copy_(sys.gripper.pos.t)
            self.gripper_rot_buffer.t[step].copy_(sys.gripper.rot.t)

    # :176-185
    def clamp_grad(self, step):
        self.pos_grad.t[step].clamp_(-1000, 1000)
        self.angleref_grad.t[step].clamp_(-1000, 1000)

    # :118-139
    def get_gripper_grad(self, step, sys):
        sys.gripper.get_rotmat()
        sys.gripper.gather_grad(sys.tmp_z_frozen, sys)
        dp = sys.gripper.d_pos.to_numpy(); da = sys.gripper.d_angle.to_numpy()
        g = self.gripper_grad.t
        for j in range(self.n_part):
            if self.vertical_only:
                g[step, j, 2] = float(dp[j][2])
            else:
                g[step, j, 0:3] = torch.as_tensor(dp[j]); g[step, j, 3:6] = torch.as_tensor(da[j])

    # :217-257
    def transfer_grad(self, step, sys, f_contact):
        ctx = sys._ensure_ctx()
        ctx.set_param("contact", 0.0 if f_contact is None else 1.0)
        self.last_stats = ctx.adjoint_step(step, self.tot_timestep, self.pos_buffer.t, self.pos_grad.t, self.ref_angle_buffer.t, self.angleref_grad.t,
                                           sys.tmp_z_frozen.t, self.damping)
        self.check_solve(step)
        # leave the scene in the state the reference leaves it in (copy_pos_and_refangle + gripper.set)
        sys.copy_pos_and_refangle(self, step)
        if self.n_part > 0 and hasattr(sys, "gripper"):
            sys.gripper.set(self.gripper_pos_buffer, self.gripper_rot_buffer, step)
            if step > 0:
                self.get_gripper_grad(step, sys)

    def check_solve(self, step):
        """The reference's H.solve is an exact sparse solve (sparse_solver.py:85-105); an adjoint solve that did NOT converge would
        silently corrupt pos_grad / gripper_grad of every earlier step, so it raises (set ``allow_unconverged`` to collect instead)."""
        st = self.last_stats
        self.worst_rel_residual = max(getattr(self, "worst_rel_residual", 0.0), st["rel_residual"])
        if st["flag"] == 3:
            self.unconverged = getattr(self, "unconverged", 0) + 1
            if not getattr(self, "allow_unconverged", False):
                from .._lib import TslError
                raise TslError(f"transfer_grad(step {step}): linear solve not converged (rel_residual {st['rel_residual']:.3e} after {st['iters']} iterations, "
                               f"method {st['method']}); the gradients of earlier steps would be wrong")

    # ---- loss seeds
    def get_loss(self, sys):  # :259-263
        self.pos_grad.t[:, : sys.cloths[0].NV, 0] = -1

    def get_loss_sheet(self, sys):  # :265-269
        self.pos_grad.t[1:, : sys.cloths[0].NV, 0] = 1

    def get_loss_book(self, sys):  # :274-278
        self.pos_grad.t[1:, : sys.cloths[0].NV, 0] = -1

    def _fold_rows(self, sys, row_a, row_b):
        """hinges (f, l) of cloth 0 whose own wing lies on grid row ``row_a`` and the opposite wing on ``row_b``"""
        c = sys.cloths[0]
        f2v = c.f2v.to_numpy(); cf = c.counter_face.to_numpy(); cp = c.counter_point.to_numpy()
        fi, l = np.nonzero(cf > np.arange(c.NF)[:, None])
        p1 = f2v[fi, l]
        p2 = f2v[cf[fi, l], cp[fi, l]]
        m = (p1 // (c.M + 1) == row_a) & (p2 // (c.M + 1) == row_b)
        return fi[m], l[m]

    def get_loss_fold(self, sys, curve7, curve8, rows=((6, 8), (7, 9))):  # :280-294
        ag = self.angleref_grad.t
        for (ra, rb), val in zip(rows, (curve7, curve8)):
            fi, l = self._fold_rows(sys, ra, rb)
            ag[self.tot_timestep - 1, 0, torch.as_tensor(fi), torch.as_tensor(l)] = val

    def get_loss_interact(self, sys):  # :408-420
        c = sys.cloths[0]; e = sys.elastics[3]
        j = self.tot_timestep - 1
        self.pos_grad.t[j, c.offset:c.offset + c.NV, 0] = 1
        self.pos_grad.t[j, e.offset:e.offset + e.n_verts, 0] = -1 * 256.0 / 144.0

    def get_loss_interact_1(self, sys):  # :422-426
        e = sys.elastics[3]
        self.pos_grad.t[self.tot_timestep - 1, e.offset:e.offset + e.n_verts, 0] = 1

    def get_loss_pick(self, sys):  # :323-327
        c = sys.cloths[0]
        sel = c.offset + np.nonzero(np.arange(c.NV) // (c.M + 1) == 8)[0]
        self.pos_grad.t[:, torch.as_tensor(sel), 2] = -1

    def get_loss_pick_fold(self, sys):  # :373-382
        fi, l = self._fold_rows(sys, 7, 9)
        self.angleref_grad.t[:, 0, torch.as_tensor(fi), torch.as_tensor(l)] = -1

    def get_loss_push(self, sys, target_pos):  # :296-300
        c = sys.cloths[0]
        j = self.tot_timestep - 1
        t = torch.as_tensor(np.asarray(target_pos), dtype=torch.float64, device=self.pos_grad.t.device)
        self.pos_grad.t[j, c.offset:c.offset + c.NV] = 2 * (self.pos_buffer.t[j, c.offset:c.offset + c.NV] - t)

    def get_loss_lift(self, sys):  # :302-312
        e = sys.elastics[0]
        j = self.tot_timestep - 1
        sl = slice(e.offset, e.offset + e.n_verts)
        d = self.pos_buffer.t[j, sl] - self.pos_buffer.t[0, sl]
        d[:, 0] += 0.012; d[:, 1] += 0.012
        self.pos_grad.t[j, sl] = d

    def get_loss_balance(self, sys):  # :428-443 (the cloth-centre entry keeps the value of the LAST ball vertex, like the kernel)
        e = sys.elastics[0]
        tt = sys.cloths[0].offset + (sys.cloth_N + 1) // 2 * (sys.cloth_M + 1) + (sys.cloth_M + 1) // 2
        pb = self.pos_buffer.t
        sl = slice(e.offset, e.offset + e.n_verts)
        d = 2 * (pb[1:, sl, 0:2] - pb[1:, tt:tt + 1, 0:2])
        self.pos_grad.t[1:, sl, 0:2] = d
        self.pos_grad.t[1:, tt, 0:2] = -d[:, -1, :]

    def get_loss_throwing(self, sys):  # :462-471
        e = sys.elastics[0]; c = sys.cloths[0]
        self.pos_grad.t[1:, e.offset:e.offset + e.n_verts, 2] = -1
        pb = self.pos_buffer.t
        i = torch.arange(sys.cloth_M)
        self.pos_grad.t[1:, c.offset + i, 2] = 20 * pb[1:, c.offset + i, 2]
        k = c.offset + i + sys.cloth_N * (sys.cloth_M + 1)
        self.pos_grad.t[1:, k, 2] = 20 * pb[1:, k, 2]

    # ---- seeds no driver of the reference calls (kept for the surface: analytic_grad_single.py:314-321, 329-371, 384-406, 445-460)
    def get_loss_sep(self, sys):  # :314-321: pull two sheets apart along x, seeds on every tape step
        c0, c1 = sys.cloths[0], sys.cloths[1]
        self.pos_grad.t[:, c0.offset:c0.offset + c0.NV, 0] = 1
        self.pos_grad.t[:, c1.offset:c1.offset + c1.NV, 0] = -1

    def get_loss_bounce(self, sys):  # :329-371: seed on the first grid row at the apex after step 40 (and on the higher of its neighbours)
        c = sys.cloths[0]
        T = self.tot_timestep
        row = slice(c.offset, c.offset + c.M + 1)
        zsum = self.pos_buffer.t[:, row, 2].sum(dim=1).cpu().numpy()
        tt = T - 1
        max_z = -1.0
        for j in range(40, T):
            if zsum[j] > max_z:
                max_z = zsum[j]; tt = j
        pb, pg = self.pos_buffer.t, self.pos_grad.t
        if tt < T - 1:
            k = tt - 1 if zsum[tt - 1] > zsum[tt + 1] else tt + 1
            pg[k, row, 2] = 2 * (pb[k, row, 2] - sys.target)
        pg[tt, row, 2] = 2 * (pb[tt, row, 2] - sys.target)
        return tt

    def get_loss_card(self, sys):  # :384-388 (the same rows as get_loss_pick)
        self.get_loss_pick(sys)

    def get_loss_slide_simple(self, sys):  # :390-393
        c = sys.cloths[0]
        self.pos_grad.t[self.tot_timestep - 1, c.offset:c.offset + c.NV, 0] = 1

    def get_loss_deliver(self, sys):  # :395-406: final cloth pose 1 cm (every axis) beyond the pose of tape step 69
        c = sys.cloths[0]
        j = self.tot_timestep - 1
        sl = slice(c.offset, c.offset + c.NV)
        self.pos_grad.t[j, sl] = 2 * (self.pos_buffer.t[j, sl] - self.pos_buffer.t[69, sl] - 0.01)

    def get_loss_side(self, sys):  # :445-460: get_loss_balance towards the vertex a quarter of the way along the cloth
        e = sys.elastics[0]
        tt = sys.cloths[0].offset + (sys.cloth_N + 1) // 4 * (sys.cloth_M + 1) + (sys.cloth_M + 1) // 2
        pb = self.pos_buffer.t
        sl = slice(e.offset, e.offset + e.n_verts)
        d = 2 * (pb[1:, sl, 0:2] - pb[1:, tt:tt + 1, 0:2])
        self.pos_grad.t[1:, sl, 0:2] = d
        self.pos_grad.t[1:, tt, 0:2] = -d[:, -1, :]

    # :492-516
    def accumulate_gripper_grad(self, traj, max_dist):
        g = self.gripper_grad.t
        for step in range(self.tot_timestep - 2, 1, -1):
            for j in range(self.n_part):
                if traj.calculate_dist(step + 1, max_dist, j) > traj.max_moving_dist - 0.00005:
                    g[step, j] += g[step + 1, j]

    def apply_action_limit_grad(self, traj, max_dist):
        g = self.gripper_grad.t
        tr = traj.traj.t
        for step in range(1, self.tot_timestep):
            for j in range(self.n_part):
                dist = traj.calculate_dist(step, max_dist, j)
                if dist > traj.max_moving_dist:
                    d = (tr[step, j] - tr[step - 1, j]).to(g.dtype)
                    g[step, j, 0:3] += d[0:3] * (dist - traj.max_moving_dist) * 10000000
                    g[step, j, 3:6] += d[3:6] * (dist - traj.max_moving_dist) * 100000
