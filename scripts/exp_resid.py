"""Probe (not part of the product): where does the PCG residual live after the fast modes are gone?  cfg4 scene, state after a few steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
N = int(sys.argv[1]) if len(sys.argv) > 1 else 224
lit = len(sys.argv) > 2 and sys.argv[2] == "literal"
gs = 1.0 if lit else N * 0.004 / 0.06
s = Scene(cloth_size=0.12 * N / 224 if lit else N * 0.004, cloth_N=N, cloth_M=N, geom_scale=gs)
s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx()
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = 5e-5 * gs; drot[:, 1] = 2e-3
if lit:
    drot[:] = 0; dpos[:, 2] = 1e-4 * np.where(np.arange(n_part) % 2 == 0, 1.0, -1.0)
for f in range(1, int(os.environ.get('STEPS', '3')) + 1):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
    print(f, st["nc"], st["newton_iters"], st["cg_iters"], flush=True)
projection_query(s)
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
H = ctx.operator_csr()
bn = b.cpu().numpy()
cons = ctx.constraints()["idx"]
c = s.cloths[0]
grp = np.zeros(s.tot_NV, int)  # 0 cloth, 1 cloth contact, 2.. bodies
grp[np.unique(cons[cons < c.NV])] = 1
for i, e in enumerate(s.elastics):
    grp[e.offset:e.offset + e.n_verts] = 2 + i
for tol in (1e-2, 1e-4, 1e-6, 1e-8):
    ctx.set_param("cg_tol", tol)
    x, st = ctx.solve(b)
    r = (bn - H @ x.cpu().numpy()).reshape(-1, 3)
    rn = np.linalg.norm(r, axis=1)
    tot = np.linalg.norm(rn)
    print(f"tol {tol:g}: iters {st['iters']} |r|/|b| {tot/np.linalg.norm(bn):.2e} share by group:", " ".join(f"{g}:{np.linalg.norm(rn[grp==g])/tot:.2f}" for g in range(2 + len(s.elastics))), flush=True)
import time
ctx.set_param("cg_tol", 1e-10)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
s.compute_residual_and_Hessian(spd=True)
x, st = ctx.solve(b)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    x, st = ctx.solve(b)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"solve: {dt*1e3:.2f} ms, {st['iters']} iterations, {dt*1e6/st['iters']:.1f} us per iteration", flush=True)
torch.cuda.synchronize(); t0 = time.time()
for rep in range(5):
    s.compute_residual_and_Hessian(spd=True)
torch.cuda.synchronize(); print(f"assemble: {(time.time()-t0)/5*1e3:.2f} ms", flush=True)
torch.cuda.synchronize(); t0 = time.time()
for rep in range(5):
    s.compute_energy()
torch.cuda.synchronize(); print(f"energy: {(time.time()-t0)/5*1e3:.2f} ms", flush=True)
