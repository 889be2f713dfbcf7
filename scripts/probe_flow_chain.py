"""Probe (not part of the product): per-batch time of the inversions on the cfg4 plan and the device-clock trace of the root's dataflow chain
("ds_dbg" 30: when every pivot inverse was published, when two far workgroups finished every step)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
for f in range(1, 4):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
ctx.set_param("verbose", 3)
s.action(4, dpos, drot); st = s.time_step(projection_query, 4)
ctx.set_param("verbose", 0)
nb = int(ctx.direct_info()["batches"])
for b in range(nb):
    ctx.set_param("ds_bench_batch", b)
    line = f"batch {b:2d}:"
    for cls, name in ((0, "gj_step"), (3, "inv_small"), (5, "gj_flow")):
        r = ctx.bench_direct(cls, 10)
        if r["launches"]:
            line += f"  {name} {r['us_per_launch'] * r['launches']:8.1f} us ({r['launches']} launches)"
    print(line, flush=True)
ctx.set_param("ds_bench_batch", -1)
r = ctx.bench_direct(5, 10)
print("gj_flow total", r["us_per_launch"] * r["launches"], "us,", r["launches"], "launches")
ctx.set_param("ds_dbg", 30)
for rep in range(3):
    s.compute_residual_and_Hessian(spd=True)
    x, ss = ctx.solve(s.F.to_torch().clone())
ctx.set_param("ds_dbg", 0)
