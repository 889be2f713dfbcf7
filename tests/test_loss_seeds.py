"""Host logic of the adjoint's loss seeds and action-limit penalties: ``Grad.get_loss_lift`` /
``get_loss_balance`` / ``get_loss_throwing`` / ``apply_action_limit_grad`` against literal loop restatements of the reference kernels
(/root/reference/code/engine/analytic_grad_single.py:302-312, 428-443, 462-471, 504-516).  The classes only need tensors, so the
test runs them on the CPU with a stand-in scene object."""
from types import SimpleNamespace

import numpy as np
import torch

from thinshelllab_amd.agent.traj_opt_single import agent_trajopt
from thinshelllab_amd.engine.analytic_grad_single import Grad


def _fake_scene(N=6, M=4, n_ball=7):
    NVc = (N + 1) * (M + 1)
    cloth = SimpleNamespace(NV=NVc, NF=2 * N * M, offset=0, N=N, M=M)
    ball = SimpleNamespace(offset=NVc, n_verts=n_ball)
    return SimpleNamespace(tot_NV=NVc + n_ball, device=torch.device("cpu"), cloth_cnt=1, cloths=[cloth], elastics=[ball], dt=5e-3,
                           cloth_N=N, cloth_M=M)


def _grad(T=5, n_part=2, seed=0):
    s = _fake_scene()
    g = Grad(s, T, n_part)
    rng = np.random.default_rng(seed)
    g.pos_buffer.from_numpy(rng.normal(size=(T, s.tot_NV, 3)))
    return s, g


def test_get_loss_lift_matches_reference_loop():
    s, g = _grad()
    g.get_loss_lift(s)
    pb = g.pos_buffer.to_numpy(); ref = np.zeros_like(pb)
    j = g.tot_timestep - 1; e = s.elastics[0]
    for i in range(e.n_verts):   # analytic_grad_single.py:302-312
        ref[j, e.offset + i, 0] = pb[j, e.offset + i, 0] - pb[0, e.offset + i, 0] + 0.012
        ref[j, e.offset + i, 1] = pb[j, e.offset + i, 1] - pb[0, e.offset + i, 1] + 0.012
        ref[j, e.offset + i, 2] = pb[j, e.offset + i, 2] - pb[0, e.offset + i, 2]
    assert np.array_equal(g.pos_grad.to_numpy(), ref)


def test_get_loss_balance_matches_reference_loop():
    s, g = _grad()
    g.get_loss_balance(s)
    pb = g.pos_buffer.to_numpy(); ref = np.zeros_like(pb)
    e = s.elastics[0]; c = s.cloths[0]
    tt = (s.cloth_N + 1) // 2 * (s.cloth_M + 1) + (s.cloth_M + 1) // 2
    for i in range(e.n_verts):   # analytic_grad_single.py:428-443, serial order: the cloth-centre entry keeps the last ball vertex
        for j in range(g.tot_timestep - 1):
            for k in (0, 1):
                d = 2 * (pb[j + 1, e.offset + i, k] - pb[j + 1, c.offset + tt, k])
                ref[j + 1, e.offset + i, k] = d
                ref[j + 1, c.offset + tt, k] = -d
    assert np.array_equal(g.pos_grad.to_numpy(), ref)


def test_get_loss_throwing_matches_reference_loop():
    s, g = _grad()
    g.get_loss_throwing(s)
    pb = g.pos_buffer.to_numpy(); ref = np.zeros_like(pb)
    e = s.elastics[0]; c = s.cloths[0]
    for j in range(g.tot_timestep - 1):   # analytic_grad_single.py:462-471
        for i in range(e.n_verts):
            ref[j + 1, e.offset + i, 2] = -1
        for i in range(s.cloth_M):
            ref[j + 1, c.offset + i, 2] = 20 * pb[j + 1, c.offset + i, 2]
            k = c.offset + i + s.cloth_N * (s.cloth_M + 1)
            ref[j + 1, k, 2] = 20 * pb[j + 1, k, 2]
    assert np.array_equal(g.pos_grad.to_numpy(), ref)


def test_apply_action_limit_grad_matches_reference_loop():
    s, g = _grad(T=6, n_part=2, seed=3)
    rng = np.random.default_rng(1)
    agent = agent_trajopt(6, 2, max_moving_dist=0.001)
    tr = np.cumsum(rng.normal(0, 8e-4, (6, 2, 6)), axis=0)   # some steps beyond the limit, some within
    tr[:, :, 3:] *= 10
    agent.traj.from_numpy(tr)
    g0 = rng.normal(size=(6, 2, 6))
    g.gripper_grad.from_numpy(g0)
    g.apply_action_limit_grad(agent, 0.015)
    ref = g0.copy()
    hit = 0
    for step in range(1, 6):   # analytic_grad_single.py:504-516 with traj_opt_single.py:29-40
        for j in range(2):
            dp = tr[step, j, 0:3] - tr[step - 1, j, 0:3]; dr = tr[step, j, 3:6] - tr[step - 1, j, 3:6]
            dist = np.sqrt(dp @ dp) + np.sqrt(dr @ dr) * 0.015
            if dist > agent.max_moving_dist:
                hit += 1
                ref[step, j, 0:3] += dp * (dist - agent.max_moving_dist) * 10000000
                ref[step, j, 3:6] += dr * (dist - agent.max_moving_dist) * 100000
    assert 0 < hit < 10
    assert np.allclose(g.gripper_grad.to_numpy(), ref, rtol=1e-13, atol=0)
    # accumulate_gripper_grad (:492-502)
    g.gripper_grad.from_numpy(g0)
    g.accumulate_gripper_grad(agent, 0.015)
    ref = g0.copy()
    for step in range(6 - 2, 1, -1):
        for j in range(2):
            dp = tr[step + 1, j, 0:3] - tr[step, j, 0:3]; dr = tr[step + 1, j, 3:6] - tr[step, j, 3:6]
            if np.sqrt(dp @ dp) + np.sqrt(dr @ dr) * 0.015 > agent.max_moving_dist - 0.00005:
                ref[step, j] += ref[step + 1, j]
    assert np.allclose(g.gripper_grad.to_numpy(), ref, rtol=1e-13, atol=0)
