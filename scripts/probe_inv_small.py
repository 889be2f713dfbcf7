"""Probe (not part of the product): the batches of the LDS inversion kernel (k_ds_inv_small) replayed alone (tsl_bench_direct class 3, one batch at a time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
for f in range(1, 4):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
ctx.set_param("verbose", 3)
s.action(4, dpos, drot); s.time_step(projection_query, 4)
ctx.set_param("verbose", 0)
nb = int(ctx.direct_info()["batches"])
for b in range(nb):
    ctx.set_param("ds_bench_batch", b)
    for cls, name in ((3, "LDS kernel"), (5, "dataflow"), (0, "block steps")):
        r = ctx.bench_direct(cls, 20)
        if r["launches"] > 0:
            print(f"batch {b:2d}: {name:12s} {r['us_per_launch'] * r['launches']:8.1f} us in {int(r['launches'])} launch(es), {r['flops_per_launch'] * r['launches'] / max(r['us_per_launch'] * r['launches'], 1e-9) * 1e-6:6.2f} TFLOP/s", flush=True)
ctx.set_param("ds_bench_batch", -1)
