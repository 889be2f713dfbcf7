import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
for f in range(1, 4):
    s.action(f, dpos, drot); s.time_step(projection_query, f)
print("normal  ", ctx.bench_direct(0, 10))
try:
    ctx.set_param("ds_dbg", 1)
    print("no inv  ", ctx.bench_direct(0, 10))
except Exception as e:
    print("no ds_dbg in this build")
