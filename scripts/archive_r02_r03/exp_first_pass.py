"""Probe (not part of the product): distribution of the relative residual after the FIRST application of the factors over the Newton iterations of a few cfg4 time steps
(verbose 5 prints every refinement iteration to stderr; run as  python scripts/exp_first_pass.py 2> log; the script re-reads its own stderr file if given)."""
import sys, os, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "parse":
    first, second = [], []
    for ln in open(sys.argv[2]):
        m = re.search(r"refinement (\d+): rel_residual ([0-9.e+-]+)", ln)
        if m:
            (first if m.group(1) == "1" else second if m.group(1) == "2" else []).append(float(m.group(2)))
    import numpy as np
    f = np.array(first)
    print("solves", len(f), " first pass <= 1e-10:", int((f <= 1e-10).sum()), " median", float(np.median(f)))
    for lo, hi in ((0, 1e-11), (1e-11, 3e-11), (3e-11, 1e-10), (1e-10, 2e-10), (2e-10, 5e-10), (5e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1)):
        print(f"  {lo:8.0e} .. {hi:8.0e}: {int(((f > lo) & (f <= hi)).sum())}")
    s2 = np.array(second)
    print("second pass: median", float(np.median(s2)) if len(s2) else None, " max", float(s2.max()) if len(s2) else None)
    sys.exit(0)
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
    if f == 3: ctx.set_param("verbose", 5)
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
