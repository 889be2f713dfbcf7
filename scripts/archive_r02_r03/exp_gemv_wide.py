"""Probe (not part of the product): the sweep launches of the upper levels with four narrow workgroups per 16-row chunk ("direct_gemv_wide_below")."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
for f in range(1, 5):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
x0 = None
for below in (0, 150, 300, 600, 1200, 2400, 1 << 30, 0, 600):
    ctx.set_param("direct_gemv_wide_below", below)
    s.compute_residual_and_Hessian(spd=True)
    x, ss = ctx.solve(b.clone())
    if x0 is None: x0 = x.clone()
    r = ctx.bench_direct(4, 20)
    print("direct_gemv_wide_below", below, "iters", ss["iters"], "rel_residual %.2e" % ss["rel_residual"], " |x - x0| / |x0| = %.1e" % float((x - x0).abs().max() / x0.abs().max()),
          f" one application {r['us_per_launch'] * r['launches']:7.1f} us", flush=True)
