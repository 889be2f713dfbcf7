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


def _fake_constraints(s, n=40, seed=3):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, s.tot_NV, size=(n, 4)).astype(np.int32)
    w = rng.random((n, 3))   # NOT normalised: with barycentric weights the pick variant's sum over w1 = (w, -1) cancels to rounding noise
    T = rng.normal(size=(n, 2, 3))
    return dict(idx=idx, w=w, k=rng.random(n) * 50, dx0=rng.normal(size=(n, 3)) * 1e-5, T=T.reshape(n, 6), n=rng.normal(size=(n, 3)),
                mu=0.1 + rng.random(n))


def _slip_scene(s):
    s.eps_v = 0.01; s.k_contact = 500.0; s.h = s.dt
    return s


def test_static_friction_loss_matches_reference_loops():
    """BaseScene.static_friction_loss (BaseScene.py:732-775) and the Scene_pick override (Scene_pick.py:193-236) against literal
    restatements of the two kernels on synthetic constraints: sliding and sticking contacts, constraints on the table body"""
    from thinshelllab_amd.engine.BaseScene import BaseScene
    from thinshelllab_amd.task_scene.Scene_pick import Scene as PickScene
    s, g = _grad(T=4)
    _slip_scene(s)
    s._friction_slip = lambda c=None, p=None: BaseScene._friction_slip(s, c, p)
    c = _fake_constraints(s)
    rng = np.random.default_rng(9)
    pos = rng.normal(size=(s.tot_NV, 3)) * 3e-5      # slips around dt eps_v = 5e-5: both branches of the threshold and of f1
    g.f_loss_ratio = 0.37
    step = 2
    thr = s.dt * s.eps_v * 0.9

    def slip(i):
        idx, w = c["idx"][i], c["w"][i]
        x_c = pos[idx[0]] * w[0] + pos[idx[1]] * w[1] + pos[idx[2]] * w[2]
        T = c["T"][i].reshape(2, 3)
        u = T @ (pos[idx[3]] - x_c - c["dx0"][i])
        return idx, w, T, u, np.linalg.norm(u)

    # BaseScene variant
    g.pos_grad.fill(0)
    BaseScene.static_friction_loss(s, g, step, constraints=c, pos=pos)
    ref = np.zeros((4, s.tot_NV, 3)); n_sl = 0
    for i in range(len(c["k"])):
        idx, w, T, u, r = slip(i)
        if r > thr:
            n_sl += 1
            w1 = np.array([-w[0], -w[1], -w[2], 1.0])
            u3 = np.array([u[0] * T[0, j] + u[1] * T[1, j] for j in range(3)])
            for i1 in range(4):
                for j1 in range(3):
                    ref[step, idx[i1], j1] += u3[j1] * w1[i1] * g.f_loss_ratio * c["k"][i]
    assert 0 < n_sl < len(c["k"])
    assert np.allclose(g.pos_grad.to_numpy(), ref, rtol=1e-13, atol=1e-18)

    # Scene_pick variant: constraints that touch elastics[0] (the table pair, i < nc1 in the reference) are skipped
    e0 = s.elastics[0]
    g.pos_grad.fill(0)
    PickScene.static_friction_loss(s, g, step, constraints=c, pos=pos)
    ref = np.zeros((4, s.tot_NV, 3)); n_used = 0
    h = s.eps_v * s.dt
    for i in range(len(c["k"])):
        idx, w, T, u, r = slip(i)
        if any(e0.offset <= v < e0.offset + e0.n_verts for v in idx):
            continue
        if r > thr:
            n_used += 1
            f1 = 1.0 / r if r > h else -r / h ** 2 + 2.0 / h
            pressure = c["k"][i] / c["mu"][i]
            g1 = (u * c["k"][i] * f1) @ T
            w1 = np.array([w[0], w[1], w[2], -1.0])
            for i1 in range(4):
                for j1 in range(3):
                    dfdp = w1[i1] * g1[j1] / pressure
                    for i2 in range(4):
                        for j2 in range(3):
                            ref[step - 1, idx[i2], j2] += -dfdp * w1[i2] * c["n"][i][j2] * s.k_contact * g.f_loss_ratio
    assert n_used > 0
    assert np.allclose(g.pos_grad.to_numpy(), ref, rtol=1e-11, atol=1e-16)
