"""Probe (not part of the product): Schur launches with single-writer entries stored (default) against all-atomic extend-add (ds_dbg 5)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thinshelllab_amd.task_scene.Scene_balancing import Scene
from thinshelllab_amd.engine.geometry import projection_query
s = Scene(cloth_size=0.12, cloth_N=224, cloth_M=224)
s.init_all(); s.mu_cloth_elastic[None] = 5.0; s.prev_pos.copy_from(s.pos)
n_part = s.gripper.n_part
dpos = np.zeros((n_part, 3)); drot = np.zeros((n_part, 3)); dpos[:, 2] = [1e-4, -1e-4][:n_part]
for f in range(1, 9):
    s.action(f, dpos, drot); st = s.time_step(projection_query, f)
print("nc", st["nc"], "unconverged", st["unconverged"], "max_rel_residual", st["max_rel_residual"])
ctx = s._ctx
for rep in range(2):
    for dbg, name in ((0, "stores + atomics"), (5, "atomics only"), (3, "no extend-add")):
        ctx.set_param("ds_dbg", dbg)
        r = ctx.bench_direct(1, 10)
        print(name, round(r["us_per_launch"] * r["launches"], 1), "us per factorisation", flush=True)
ctx.set_param("ds_dbg", 0)
# product alone (G = W F12 launches: no extend-add): real operands / every workgroup on the same operand tiles / no global loads after the first slab
for dbg, name in ((0, "G product"), (6, "G product, cache-resident operands"), (7, "G product, no global loads in the K loop"), (10, "G product, no global loads and no barriers in the K loop"), (11, "G product, no loads, no barriers, no LDS refill: LDS reads + matrix cores only")):
    ctx.set_param("ds_dbg", dbg)
    r = ctx.bench_direct(2, 10)
    t = r["us_per_launch"] * r["launches"]
    print(name, round(t, 1), "us per factorisation,", round(r["flops_per_launch"] * r["launches"] / t * 1e-6, 1), "TFLOP/s", flush=True)
ctx.set_param("ds_dbg", 0)
# persistent variant: at most `cap` workgroups walk the tiles (does desynchronising the workgroups of a CU raise the matrix-core utilisation?)
for cap in (0, 1024, 2048, 768):
    ctx.set_param("direct_gemm_persist", cap)
    r1 = ctx.bench_direct(1, 10); r2 = ctx.bench_direct(2, 10)
    print("gemm_persist", cap, "Schur", round(r1["us_per_launch"] * r1["launches"], 1), "us  G", round(r2["us_per_launch"] * r2["launches"], 1), "us", flush=True)
ctx.set_param("direct_gemm_persist", 0)
