"""Probe (not part of the product): the same operator factorised and solved N times on the dataflow path; every solve must take the same number of
refinement iterations and return the same answer as the launch-per-block-step path (a race in the flag protocol would show as a solve that needs more)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
N = int(os.environ.get("GRID", "224"))
s = Scene(cloth_size=0.12, cloth_N=N, cloth_M=N); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for f in range(1, steps + 1):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
ctx.set_param("direct_flow", 0)
x0, ss0 = ctx.solve(b.clone())
print("reference (direct_flow 0):", ss0["iters"], ss0["rel_residual"], flush=True)
for flow in (1, 3):
    ctx.set_param("direct_flow", flow)
    worst = 0.0; hist = {}
    for r in range(reps):
        s.compute_residual_and_Hessian(spd=True)     # invalidates the factors: the next solve factorises again
        x, ss = ctx.solve(b.clone())
        hist[ss["iters"]] = hist.get(ss["iters"], 0) + 1
        worst = max(worst, float((x - x0).abs().max() / x0.abs().max()))
    print("direct_flow", flow, "solves", reps, "refinement iterations -> count", hist, " worst |x - x0| / |x0|", worst, flush=True)
