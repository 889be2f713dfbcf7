"""Probe (not part of the product): does the dataflow inversion of a batch overlap with the GEMMs (G + Schur) of the batch before it when the two are issued
on two streams?  tsl_bench_direct class 7 against the classes alone (5; 2 + 1), per batch of the cfg4 plan."""
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
nb = int(ctx.direct_info()["batches"])
for b in range(1, nb):
    ctx.set_param("ds_bench_batch", b)
    fl = ctx.bench_direct(5, 10)
    tf = fl["us_per_launch"] * fl["launches"]
    ctx.set_param("ds_bench_batch", b - 1)
    g = ctx.bench_direct(2, 10); sc = ctx.bench_direct(1, 10)
    tg = g["us_per_launch"] * g["launches"] + sc["us_per_launch"] * sc["launches"]
    ctx.set_param("ds_bench_batch", b)
    both = ctx.bench_direct(7, 10)
    tb = both["us_per_launch"] * both["launches"]
    if tf > 0 and tb > 0:
        print(f"batch {b:2d}: dataflow inversion {tf:7.1f} us | G + Schur of batch {b - 1:2d}: {tg:7.1f} us | side by side {tb:7.1f} us = {tb / (tf + tg):.2f} of the sum, {tb / max(tf, tg):.2f} of the longer", flush=True)
ctx.set_param("ds_bench_batch", -1)
