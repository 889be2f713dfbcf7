"""Probe (not part of the product): kernel classes of the sparse direct solve on the cfg4 plan, per batch and in total
(tsl_bench_direct with "ds_bench_batch").  usage: probe_direct.py [steps] ; TSL_PARAMS=key=value,... sets engine parameters."""
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for f in range(1, steps + 1):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
print("nc", st["nc"], "newton", st["newton_iters"], "unconverged", st["unconverged"])
s.compute_residual_and_Hessian(spd=True)
x, ss = ctx.solve(s.F.to_torch().clone())
print("solve:", ss)
info = ctx.direct_info(); cnt = ctx.direct_counters()
print({k: info[k] for k in ("supernodes", "levels", "batches", "flops_per_factorization", "front_bytes")}, cnt)
CLS = ((6, "extend"), (0, "gj_step"), (3, "inv_small"), (5, "gj_flow"), (2, "G"), (1, "schur"))
nb = int(info["batches"])
tot = {c: 0.0 for c, _ in CLS}
if "--batches" in sys.argv:
    for b in range(nb):
        ctx.set_param("ds_bench_batch", b)
        line = f"batch {b:2d}:"
        for cls, name in CLS:
            r = ctx.bench_direct(cls, 10)
            t = r["us_per_launch"] * r["launches"]
            if r["launches"] == 0:
                continue
            tot[cls] += t
            line += f"  {name} {t:7.1f} us ({r['flops_per_launch'] * r['launches'] / max(t, 1e-9) * 1e-6:5.1f} TF/s, {r['bytes_per_launch'] * r['launches'] / max(t, 1e-9) * 1e-3:6.0f} GB/s)"
        print(line, flush=True)
    ctx.set_param("ds_bench_batch", -1)
for cls, name in CLS + ((4, "gemv"),):
    r = ctx.bench_direct(cls, 10)
    t = r["us_per_launch"] * r["launches"]
    print(f"{name:10s} {t:8.1f} us per factorisation / application, {r['launches']:3d} launches, {r['flops_per_launch'] * r['launches'] / max(t, 1e-9) * 1e-6:5.1f} TF/s, "
          f"{r['bytes_per_launch'] * r['launches'] / max(t, 1e-9) * 1e-3:6.0f} GB/s algorithmic, {r['bytes_per_launch'] * r['launches'] * 1e-6:7.1f} MB")
for dbg, what in ((13, "schur without the gather of the children"), (12, "schur, ONE K slab (prologue + epilogue)")):
    ctx.set_param("ds_dbg", dbg)
    r = ctx.bench_direct(1, 10)
    print(f"{what}: {r['us_per_launch'] * r['launches']:8.1f} us")
ctx.set_param("ds_dbg", 0)
if "--s32" in sys.argv:
    for thr in (0, 1100, 3000, 8000, 1 << 30):
        ctx.set_param("direct_s32_below", thr)
        line = f"s32_below {thr}: "
        tot = 0
        for b in range(nb):
            ctx.set_param("ds_bench_batch", b)
            r = ctx.bench_direct(1, 10)
            t = r["us_per_launch"] * r["launches"]; tot += t
            line += f"{t:6.1f} "
        ctx.set_param("ds_bench_batch", -1)
        r = ctx.bench_direct(1, 10)
        print(line, f"| sum {tot:7.1f} | all together {r['us_per_launch'] * r['launches']:7.1f} us")
    ctx.set_param("direct_s32_below", 0)
