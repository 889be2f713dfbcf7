"""Probe (not part of the product): the level sweeps of one application of the factors as one launch for the levels >= L0 ("direct_sweep_flow")
against one launch per level and mode; same solve, same answer."""
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
for f in range(1, steps + 1):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
print("nc", st["nc"], flush=True)
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
xs = {}
for L0 in (0, 1, 2, 3, 4, 5):
    ctx.set_param("direct_sweep_flow", L0)
    s.compute_residual_and_Hessian(spd=True)
    x, ss = ctx.solve(b.clone())
    xs[L0] = x.clone()
    r = ctx.bench_direct(4, 20)
    print("direct_sweep_flow", L0, {k: ss[k] for k in ("iters", "rel_residual", "flag") if k in ss}, " |x - x0| / |x0| =", float((x - xs[0]).abs().max() / xs[0].abs().max()),
          f" one application {r['us_per_launch'] * r['launches']:7.1f} us in {r['launches']} launches", flush=True)
for L0 in (0, 2, 0, 2):
    ctx.set_param("direct_sweep_flow", L0)
    torch.cuda.synchronize(); t0 = time.time()
    for f in range(steps + 1, steps + 4):
        s.action(f, dpos, drot); st = s.time_step(projection_query, f)
    torch.cuda.synchronize()
    print("direct_sweep_flow", L0, "3 forward steps", round((time.time() - t0) * 1e3 / 3, 1), "ms per step", "unconverged", st.get("unconverged"), flush=True)
    steps += 3
