"""Probe (not part of the product): the block steps of the upper levels as one persistent dataflow launch ("direct_flow", k_ds_gj_flow) against
the launch-per-block-step path, batch by batch on the cfg4 plan, and the solve with either."""
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
for flow in (0, 1):
    ctx.set_param("direct_flow", flow)
    s.compute_residual_and_Hessian(spd=True)
    x, ss = ctx.solve(b.clone())
    xs[flow] = x.clone()
    print("flow", flow, {k: ss[k] for k in ("iters", "rel_residual", "flag", "method") if k in ss}, flush=True)
print("max |x1 - x0| / max |x0| =", float((xs[1] - xs[0]).abs().max() / xs[0].abs().max()), flush=True)
nb = int(ctx.direct_info()["batches"])
tot = {0: 0.0, 1: 0.0, 3: 0.0}
def t_of(r): return r["us_per_launch"] * r["launches"]
for bt in range(nb):
    ctx.set_param("ds_bench_batch", bt)
    ctx.set_param("direct_flow", 0)
    r0 = ctx.bench_direct(0, 10); r3 = ctx.bench_direct(3, 10)
    base = t_of(r0) + t_of(r3)
    line = f"batch {bt:2d}: " + (f"launch per block step {t_of(r0):7.1f} us ({r0['launches']:3d} launches)" if r0["launches"] else f"LDS kernel            {t_of(r3):7.1f} us (  1 launch  )")
    for flow in (1, 3):
        ctx.set_param("direct_flow", flow)
        r5 = ctx.bench_direct(5, 10)
        t = t_of(r5) if r5["launches"] else base
        tot[flow] += t
        line += f"   direct_flow {flow}: " + (f"dataflow {t_of(r5):7.1f} us" if r5["launches"] else "unchanged        ")
    tot[0] += base
    print(line, flush=True)
print("inversions per factorisation, us:", tot)
ctx.set_param("ds_bench_batch", -1)
# whole time steps with either
for flow in (0, 1, 3, 0, 1, 3):
    ctx.set_param("direct_flow", flow)
    torch.cuda.synchronize(); t0 = time.time()
    for f in range(steps + 1, steps + 4):
        s.action(f, dpos, drot); st = s.time_step(projection_query, f)
    torch.cuda.synchronize()
    print("flow", flow, "3 forward steps", round((time.time() - t0) * 1e3 / 3, 1), "ms per step", "unconverged", st.get("unconverged"), flush=True)
    steps += 3
