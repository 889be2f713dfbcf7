"""Probe: PCG iterations of the first Newton system of the stiff drape (compare with a scipy multigrid prototype) and after steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_drape import Scene
N = int(sys.argv[1]); size = float(sys.argv[2])
s = Scene(cloth_size=size, N=N); s.init_all()
ctx = s._ensure_ctx(); ctx.set_param("cg_maxit", 100000)
def probe(tag):
    s.compute_residual_and_Hessian(spd=True)
    b = s.F.to_torch().clone()
    x, st = ctx.solve(b)
    print(tag, "iters", st["iters"], "flag", st["flag"], "restarts", st["restarts"], flush=True)
probe("initial")
for k in range(3):
    st = s.time_step(None, k + 1)
    print("step", k, st["newton_iters"], st["cg_iters"], st["last_delta"], flush=True)
    probe(f"after step {k}")
