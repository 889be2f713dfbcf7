"""Probe (not part of the product): per-batch time of the G = W F12 launches with parts of the K loop removed ("ds_dbg": 0 full, 6 every workgroup streams the
same operand tiles, 7 no global loads after the first slab, 10 no barriers either, 11 no LDS refill either, 12 one slab only: prologue + epilogue, 13 F22 not read, 3 no extend-add)."""
import sys, os
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
x, ss = ctx.solve(s.F.to_torch().clone())
nb = int(ctx.direct_info()["batches"])
for cls, name in ((2, "G"), (1, "schur")):
    for b in range(nb):
        ctx.set_param("ds_bench_batch", b)
        line = f"{name} batch {b:2d}:"
        for dbg in (0, 6, 7, 10, 11, 12, 13, 3):
            ctx.set_param("ds_dbg", dbg)
            r = ctx.bench_direct(cls, 10)
            t = r["us_per_launch"] * r["launches"]
            if dbg == 0:
                line += f" {r['flops_per_launch'] * r['launches'] * 1e-9:5.2f} GF "
            line += f"  dbg {dbg}: {t:6.1f} us"
        ctx.set_param("ds_dbg", 0)
        print(line, flush=True)
