"""Generates tests/golden/*.npz from the CPU oracle (oracle/).  The reference cannot be imported or run here (taichi /
cupy absent), so these are regression vectors of the restatement, not reference outputs:
inputs (scene name, action sequence, loss seed) and expected outputs (final positions, plastic angles, gradients).
    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def folding_case():
    from helpers import oracle_from_scene
    from oracle import pyoracle as po
    from thinshelllab_amd.task_scene.Scene_folding import Scene
    po.set_threads(1)
    s = Scene(cloth_size=0.1)
    s.cloths[0].Kb[None] = 400.0
    s.init_all()
    s.mu_cloth_elastic[None] = 5.0
    s.prev_pos.copy_from(s.pos)
    o = oracle_from_scene(po, s)
    c = s.cloths[0]
    x = s.pos.to_numpy()
    x[: c.NV, 2] += 2e-6 * np.sin(0.7 * np.arange(c.NV) + 0.3)   # off the exact contact threshold (see tests/test_gpu_scenes.py)
    o.pos[:] = x; o.prev_pos[:] = x; o.push_down_all()
    T = 5
    o.set_solver(1e-12)
    o.grad_new(T, 1)
    o.grad_copy_pos(0)
    actions = np.zeros((T, 1, 6))
    actions[1:, 0, 2] = -1.5e-4
    actions[1:, 0, 0] = 5e-5
    actions[1:, 0, 4] = 2e-3
    for f in range(1, T):
        o.action(actions[f, :, 0:3], actions[f, :, 3:6])
        o.time_step()
        o.grad_copy_pos(f)
    NV = o.tot_NV
    NF = o.int("cloth0.NF")
    ag = o.arr("grad.angleref_grad", (T, 1, NF, 3))
    f2v = o.arr("cloth0.f2v", (-1, 3)); cf = o.arr("cloth0.counter_face", (-1, 3)); cp = o.arr("cloth0.counter_point", (-1, 3))
    for f in range(NF):
        for l in range(3):
            if cf[f][l] > f:
                p1 = f2v[f][l] // 4; p2 = f2v[cf[f][l]][cp[f][l]] // 4
                if (p1, p2) == (6, 8): ag[T - 1, 0, f, l] = 1.0
                if (p1, p2) == (7, 9): ag[T - 1, 0, f, l] = -1.0
    for st in range(T - 1, 0, -1):
        o.grad_transfer(st)
    return dict(x0=x, actions=actions, pos_buffer=o.arr("grad.pos_buffer", (T, NV, 3)).copy(),
                ref_angle_buffer=o.arr("grad.ref_angle_buffer", (T, 1, NF, 3)).copy(),
                gripper_grad=o.arr("grad.gripper_grad", (T, 1, 6)).copy(), pos_grad=o.arr("grad.pos_grad", (T, NV, 3)).copy(),
                angleref_grad=o.arr("grad.angleref_grad", (T, 1, NF, 3)).copy())


if __name__ == "__main__":
    d = folding_case()
    np.savez_compressed(os.path.join(HERE, "folding_rollout.npz"), **d)
    print({k: v.shape for k, v in d.items()}, "gripper_grad", d["gripper_grad"][1:, 0])
