"""First-order optimisers over the trajectory tensor: counterparts of ``Adam``, ``Adam_single`` and ``SGD_single``
(/root/reference/code/optimizer/optim.py:3-104).  Element-wise host updates (a few hundred numbers)."""
import math

import torch

from ..engine.field import Field


def _t(x):
    return x.t if isinstance(x, Field) else x


class Adam:
    def __init__(self, parameters_shape, lr, beta_1, beta_2, eps):
        self.tot_timestep = parameters_shape[0]
        self.action_dim = parameters_shape[1]
        self.lr = lr
        self.beta_1 = float(beta_1)
        self.beta_2 = beta_2
        self.eps = eps
        self.momentum_buffer = torch.zeros(tuple(parameters_shape[:2]), dtype=torch.float64)
        self.v_buffer = torch.zeros(tuple(parameters_shape[:2]), dtype=torch.float64)
        self.iter = 0.0

    # optim.py:16-29
    def step(self, parameters, grads):
        p, g = _t(parameters), _t(grads).to(torch.float64).cpu()
        self.momentum_buffer = self.beta_1 * self.momentum_buffer + (1 - self.beta_1) * g
        self.v_buffer = self.beta_2 * self.v_buffer + (1 - self.beta_2) * (g * g)
        m_cap = self.momentum_buffer / (1 - self.beta_1 ** (self.iter + 1))
        v_cap = self.v_buffer / (1 - self.beta_2 ** (self.iter + 1))
        p -= ((self.lr * m_cap) / torch.sqrt(v_cap + self.eps)).to(p.device)
        self.iter += 1.0

    def reset(self):
        self.iter = 0.0
        self.momentum_buffer.zero_()
        self.v_buffer.zero_()


class Adam_single:
    def __init__(self, parameters_shape, lr, beta_1, beta_2, eps, discount=0.9):
        self.tot_timestep, self.action_dim1, self.action_dim2 = parameters_shape
        self.beta_1 = float(beta_1)
        self.beta_2 = beta_2
        self.eps = eps
        self.momentum_buffer = torch.zeros(tuple(parameters_shape), dtype=torch.float64)
        self.v_buffer = torch.zeros(tuple(parameters_shape), dtype=torch.float64)
        self.iter = 0.0
        self.lr = lr
        self.ori_lr = lr
        self.discount = discount

    # optim.py:53-75 (note: sqrt(v_cap + eps), and lr *= discount every 10th iteration)
    def step(self, parameters, grads):
        p, g = _t(parameters), _t(grads).to(torch.float64).cpu()
        if torch.isnan(g).any():
            print("nan in gripper grid!!")
        self.momentum_buffer = self.beta_1 * self.momentum_buffer + (1 - self.beta_1) * g
        self.v_buffer = self.beta_2 * self.v_buffer + (1 - self.beta_2) * (g * g)
        m_cap = self.momentum_buffer / (1 - self.beta_1 ** (self.iter + 1))
        v_cap = self.v_buffer / (1 - self.beta_2 ** (self.iter + 1))
        p -= ((self.lr * m_cap) / torch.sqrt(v_cap + self.eps)).to(p.device)
        self.iter += 1.0
        if int(self.iter) % 10 == 0:
            self.lr *= self.discount

    def reset(self):
        self.iter = 0.0
        self.lr = self.ori_lr
        self.momentum_buffer.zero_()
        self.v_buffer.zero_()


class SGD_single:
    def __init__(self, parameters_shape, lr, beta_1, beta_2, eps):
        self.tot_timestep, self.action_dim1, self.action_dim2 = parameters_shape
        self.lr = lr

    def step(self, parameters, grads):
        p = _t(parameters)
        p -= (self.lr * _t(grads).to(torch.float64)).to(p.device)

    def reset(self):
        pass
