"""Probe (not part of the product): per-batch time of the factorisation kernels on the cfg4 plan (tsl_bench_direct with "ds_bench_batch").
(The timing switches that removed the gather / all but one K slab from the Schur kernel -- profiles/r04c_cfg4_schur_per_batch.txt -- left with round 5's prune.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for f in range(1, steps + 1):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
print("nc", st["nc"])
ctx.set_param("verbose", 3)
s.action(steps + 1, dpos, drot); st = s.time_step(projection_query, steps + 1)
ctx.set_param("verbose", 0)
s.compute_residual_and_Hessian(spd=True)
x, ss = ctx.solve(s.F.to_torch().clone())   # prints the plan (verbose 3) if it is rebuilt
ctx.set_param("verbose", 0)
nb = int(ctx.direct_info()["batches"])
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for b in range(nb):
    ctx.set_param("ds_bench_batch", b)
    line = f"batch {b:2d}:"
    for cls, name in ((0, "inv"), (2, "G"), (1, "schur")):
        r = ctx.bench_direct(cls, 10)
        t = r["us_per_launch"] * r["launches"]
        tot[cls] += t
        tf = r["flops_per_launch"] * r["launches"] / max(t, 1e-9) * 1e-6
        line += f"  {name} {t:8.1f} us ({r['launches']:3d} launches, {tf:5.1f} TF/s)"
        if cls == 1:
            line += f" {r['bytes_per_launch'] * r['launches'] / max(t, 1e-9) * 1e-3:6.0f} GB/s"
    print(line, flush=True)
ctx.set_param("ds_bench_batch", -1)
print("totals us:", tot, "sum", sum(tot.values()))
