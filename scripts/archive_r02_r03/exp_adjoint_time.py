"""Probe (not part of the product): where the time of a reverse step goes (tsl_adjoint_step against the host work around it)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.analytic_grad_single import Grad
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
K = 12
n_part = s.gripper.n_part
g = Grad(s, K + 1, n_part); g.init_mass(s); g.copy_pos(s, 0)
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
tf = []
for f in range(1, K + 1):
    s.action(f, dpos, drot)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.time_step(projection_query, f)
    torch.cuda.synchronize(); tf.append(time.perf_counter() - t0)
    g.copy_pos(s, f)
g.get_loss_balance(s)
ctx = s._ctx
ta, tc = [], []
for st in range(K, 0, -1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.set_param("contact", 1.0)
    ls = ctx.adjoint_step(st, g.tot_timestep, g.pos_buffer.t, g.pos_grad.t, g.ref_angle_buffer.t, g.angleref_grad.t, s.tmp_z_frozen.t, g.damping)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.copy_pos_and_refangle(g, st); s.gripper.set(g.gripper_pos_buffer, g.gripper_rot_buffer, st); g.get_gripper_grad(st, s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ta.append(t1 - t0); tc.append(t2 - t1)
    print(f"adjoint step {st}: tsl_adjoint_step {1e3 * (t1 - t0):.2f} ms (iters {ls['iters']}, method {ls['method']}), host tail {1e3 * (t2 - t1):.2f} ms", flush=True)
info = ctx.direct_info()
print("forward ms per step", [round(1e3 * t, 1) for t in tf])
print("mean adjoint C call", 1e3 * np.mean(ta), "ms, host tail", 1e3 * np.mean(tc), "ms; plans", info["plans"], "plan seconds", info["plan_seconds"])
