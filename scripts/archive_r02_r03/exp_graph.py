"""Probe (not part of the product): kernel classes of the direct solve replayed as stream launches against hipGraph replays, and with
the pivot-tile inversion switched off (ds_dbg 1): where the time of a dependent launch goes.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
for f in range(1, 4):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
ctx = s._ctx
for dbg, name in ((0, "stream launches"), (4, "hipGraph replays"), (1, "stream launches, no pivot-tile inversion"), (9, "update tiles without their own read / write (half the tile traffic)")):
    ctx.set_param("ds_dbg", dbg)
    out = {k: ctx.bench_direct(k, 10) for k in (0, 1, 2, 4)}
    print(name, {k: (round(v["us_per_launch"], 2), v["launches"]) for k, v in out.items()}, flush=True)
ctx.set_param("ds_dbg", 0)
