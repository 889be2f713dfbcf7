"""Probe (not part of the product): WHERE the residual of the first application of the factors sits (cfg4 operator after a few driven
steps): per cloth grid line, to tell the top separators (grid mid-lines) from the leaves."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query

N = 224
s = Scene(cloth_size=0.12, cloth_N=N, cloth_M=N); s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4]
for f in range(1, 7):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
H = ctx.operator_csr()
ctx.set_param("cg_tol", 1.0)            # accept the first application
x1, st1 = ctx.solve(b.clone())
ctx.set_param("cg_tol", 1e-10)
bn = b.cpu().numpy(); r = bn - H @ x1.cpu().numpy()
print("first pass: rel residual", np.linalg.norm(r) / np.linalg.norm(bn), st1)
try:
    print("tiles through the guarded form in that factorisation:", ctx.direct_counters().get("tiles_guarded"), " perturbed pivots:", ctx.direct_info()["perturbed_pivots"])
except Exception as e:
    print(e)
rv = (r.reshape(-1, 3) ** 2).sum(1)
c = s.cloths[0]
g = rv[c.offset:c.offset + c.NV].reshape(N + 1, N + 1)
tot = rv.sum()
print("share of |r|^2 on the cloth:", g.sum() / tot, " on the bodies:", 1 - g.sum() / tot)
rows = g.sum(1) / tot; cols = g.sum(0) / tot
top = np.argsort(-rows)[:12]; print("grid rows with the largest share:", [(int(i), round(float(rows[i]), 4)) for i in top])
top = np.argsort(-cols)[:12]; print("grid cols with the largest share:", [(int(i), round(float(cols[i]), 4)) for i in top])
mid = set(range(N // 2 - 2, N // 2 + 3))
print("share within 2 lines of the two mid-lines (root separator):", float(g[list(mid), :].sum() + g[:, list(mid)].sum()) / tot)
q = [N // 4, 3 * N // 4]
ql = set(i + d for i in q for d in range(-2, 3))
print("share within 2 lines of the quarter lines (level 8 / 7 separators):", float(g[list(ql), :].sum() + g[:, list(ql)].sum()) / tot)
flat = np.sort(rv)[::-1]
print("share of the 100 / 1000 / 10000 largest vertices:", flat[:100].sum() / tot, flat[:1000].sum() / tot, flat[:10000].sum() / tot)
