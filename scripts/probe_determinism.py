"""Probe (not part of the product): which quantity of a step differs between two fresh runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.engine.geometry import projection_query
from thinshelllab_amd.task_scene.Scene_balancing import Scene

def run(nsteps):
    s = Scene(cloth_size=0.06, cloth_N=48, cloth_M=48); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
    ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
    for kv in os.environ.get("TSL_PARAMS", "").split(","):
        if "=" in kv: ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    n_part = s.gripper.n_part
    dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
    out = {}
    for f in range(1, nsteps + 1):
        s.action(f, dpos, drot)
        if f == nsteps:
            nc = projection_query(s)
            idx = ctx.constraints_export()[0] if hasattr(ctx, "constraints_export") else None
            out["nc"] = np.array([nc])
            out["E"] = np.array([s.compute_energy()])
            s.compute_residual_and_Hessian(spd=True)
            out["F"] = s.F.to_numpy().copy()
            H = ctx.operator_csr(); H.sort_indices()
            out["H"] = H.data.copy()
            x, st = ctx.solve(s.F.to_torch().clone())
            out["x"] = x.cpu().numpy().copy()
            fl, dr, pi, pw = ctx.proj_export()
            out["proj_flag"] = fl.copy(); out["proj_dir"] = dr.copy(); out["proj_w"] = pw.copy()
        st = s.time_step(projection_query, f)
        out[f"pos{f}"] = s.pos.to_numpy().copy()
        out[f"st{f}"] = np.array([st["nc"], st["newton_iters"], st["ls_evals"], st["energy"]])
    return out

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a = run(n); b = run(n)
for k in a:
    d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() if a[k].shape == b[k].shape else "shape"
    print(f"{k:10s} equal {np.array_equal(a[k], b[k])}  max|d| {d}")
