"""System-identification adjoint: host-side counterpart of ``Grad``
(/root/reference/code/engine/analytic_grad_system.py).

Same tape and reverse step as ``analytic_grad_single.Grad`` (one ``tsl_adjoint_step`` per step) with the differences of the
reference class: ``pos_grad`` is clamped to +-1 and nothing else (:104-109), there is no gripper gradient, and every step
accumulates the parameter gradients ``grad_kb`` / ``grad_mu`` / ``grad_lam`` = sum over the free dofs of
``p . d(force)/d(parameter)`` (:69-80, ``tsl_param_grad``) or, with ``count_friction_grad``, the friction-coefficient gradient of
``Scene_sliding`` (``contact_energy_backprop_friction``, Scene_sliding.py:139-176, ``tsl_friction_grad``).
"""
import torch

from .field import Field, ScalarField


class Grad:
    def __init__(self, sys, tot_timestep, n_parts):
        # analytic_grad_system.py:5-31
        self.tot_NV = sys.tot_NV
        self.n_part = n_parts
        dev = sys.device
        T = tot_timestep
        z = lambda *shape: Field(torch.zeros(shape, dtype=torch.float64, device=dev))
        self.pos_buffer = z(T, sys.tot_NV, 3)
        if n_parts > 0:
            self.gripper_pos_buffer = Field(torch.zeros((T, n_parts, 3), dtype=torch.float64))
            self.gripper_rot_buffer = Field(torch.zeros((T, n_parts, 4), dtype=torch.float64))
        self.cloth_cnt = sys.cloth_cnt
        self.NF = sys.cloths[0].NF
        self.ref_angle_buffer = z(T, sys.cloth_cnt, self.NF, 3)
        self.dt = sys.dt
        self.pos_grad = z(T, sys.tot_NV, 3)
        self.x_hat_grad = z(sys.tot_NV * 3)
        self.gripper_grad = Field(torch.zeros((T, 7), dtype=torch.float64))
        self.mass = Field(torch.zeros(sys.tot_NV, dtype=torch.float64))
        self.tot_timestep = T
        self.grad_lam = ScalarField(0.0)
        self.grad_mu = ScalarField(0.0)
        self.grad_friction_coef = ScalarField(0.0)
        self.grad_kb = ScalarField(0.0)
        self.angleref_grad = z(T, sys.cloth_cnt, self.NF, 3)
        self.damping = 1.0
        self.count_friction_grad = False
        self.count_mu_lam_grad = False
        self.count_kb_grad = True
        self.last_stats = {}

    def reset(self):  # :33-39
        self.pos_buffer.fill(0)
        self.pos_grad.fill(0)
        self.grad_mu[None] = 0
        self.grad_lam[None] = 0
        self.grad_friction_coef[None] = 0
        self.grad_kb[None] = 0

    def init_mass(self, sys):  # :41-44
        self.mass.copy_from(sys.mass)

    def copy_pos(self, sys, step):  # :46-59
        self.pos_buffer.t[step].copy_(sys.pos.t)
        self.ref_angle_buffer.t[step].view(-1, 3).copy_(sys._ref_angle[: self.cloth_cnt * self.NF])
        if self.n_part > 0 and hasattr(sys, "gripper"):
            self.gripper_pos_buffer.t[step].copy_(sys.gripper.pos.t)
            self.gripper_rot_buffer.t[step].copy_(sys.gripper.rot.t)

    def check_solve(self, step):
        from .analytic_grad_single import Grad as _G
        _G.check_solve(self, step)

    def clamp_grad(self, step):  # :104-109
        self.pos_grad.t[step].clamp_(-1, 1)

    def transfer_grad(self, step, sys, f_contact):  # :112-160
        ctx = sys._ensure_ctx()
        ctx.set_param("contact", 0.0 if f_contact is None else 1.0)
        ctx.set_param("adj_clamp", 1.0); ctx.set_param("adj_clamp_angleref", 0.0)
        try:
            self.last_stats = ctx.adjoint_step(step, self.tot_timestep, self.pos_buffer.t, self.pos_grad.t, self.ref_angle_buffer.t, self.angleref_grad.t,
                                               sys.tmp_z_frozen.t, self.damping)
            self.check_solve(step)
            if self.count_friction_grad:   # :150-153: either the friction coefficient or the stiffness parameters
                self.grad_friction_coef[None] = self.grad_friction_coef[None] + ctx.friction_grad(self.pos_buffer.t[step])
                g = dict(kb=0.0, mu=0.0, lam=0.0)
            else:
                g = ctx.param_grad(self.pos_buffer.t[step], self.ref_angle_buffer.t[step - 1])
        finally:
            ctx.set_param("adj_clamp", 1000.0); ctx.set_param("adj_clamp_angleref", 1.0)
        if not self.count_friction_grad:
            if self.count_mu_lam_grad:
                self.grad_mu[None] = self.grad_mu[None] + g["mu"]
                self.grad_lam[None] = self.grad_lam[None] + g["lam"]
            if self.count_kb_grad:
                self.grad_kb[None] = self.grad_kb[None] + g["kb"]
        sys.copy_pos_and_refangle(self, step)

    # ---- loss seeds
    def get_loss_slide(self, sys, pos_grad=False):  # :171-173
        c = sys.cloths[0]
        self.pos_grad.t[1:, c.offset:c.offset + c.NV, 0] = 1

    def get_loss_card(self, sys):  # :175-177
        c = sys.cloths[0]
        self.pos_grad.t[self.tot_timestep - 1, c.offset:c.offset + c.NV, 0] = 1

    def get_loss_table(self, sys):  # :179-183
        c = sys.cloths[0]
        i = torch.arange(c.NV, device=self.pos_grad.t.device)
        row = torch.div(i, c.N + 1, rounding_mode="floor")
        sel = c.offset + i[(row == 5) | (row == 10)]
        self.pos_grad.t[1:, sel, 2] = -1
