"""Probe (not part of the product): wall time of every cfg4 step and the dataflow counters, to catch a k_ds_gj_flow launch that runs into its poll limit."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query

N = 224
s = Scene(cloth_size=0.12, cloth_N=N, cloth_M=N); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = 1e-4 * np.where(np.arange(n_part) % 2 == 0, 1.0, -1.0)
for f in range(1, int(os.environ.get("STEPS", "12")) + 1):
    torch.cuda.synchronize(); t0 = time.time()
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
    torch.cuda.synchronize(); dt = time.time() - t0
    k = ctx.direct_counters()
    print(f"step {f}: {1e3 * dt:8.1f} ms  newton {st['newton_iters']}  flow launches {k['flow_launches']} aborts {k['flow_aborts']}", flush=True)
