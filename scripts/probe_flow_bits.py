"""Probe (not part of the product): are repeated factorisations of ONE operator on the dataflow path bit-identical?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_direct import _drape
s = _drape(160, 96, 5e-5, seed=5)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1); ctx.set_param("direct_leaf", 16); ctx.set_param("cg_tol", 1.0)
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv: ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
s.compute_residual_and_Hessian(spd=True)
b = s.F.to_torch().clone()
xs = []
for rep in range(8):
    s.compute_residual_and_Hessian(spd=True)
    x, st = ctx.solve(b.clone())
    xs.append(x.cpu().numpy().copy())
print("flow launches", ctx.direct_counters()["flow_launches"], "first-pass residual", st["rel_residual"])
for k in range(1, 8):
    d = np.abs(xs[k] - xs[0])
    print(k, "equal", np.array_equal(xs[k], xs[0]), "max|d|", d.max(), "n differing", int((d > 0).sum()), "of", d.size)
