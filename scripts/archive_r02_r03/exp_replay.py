"""Probe (not part of the product): roll the cfg4 scene of bench.py until a time step reports unconverged solves, then replay that
step from the saved pre-step state under different static-pivoting tolerances ("direct_piv_tol").
usage: exp_replay.py [steps] [tol,tol,...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from argparse import Namespace
import bench
from thinshelllab_amd.engine.geometry import projection_query

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
tols = [float(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "1e-13,1e-10,1e-8,1e-6").split(",")]
args = Namespace(workload="cfg4", grid=224, cloth_size=0.12)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx()
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
os.makedirs("/tmp/st", exist_ok=True)
n_replayed = 0
for f in range(1, T + 1):
    d = bench._drive(s.gripper.n_part, 1.0, 0, f, int(os.environ.get('IDLE', '0')))
    s.save_all("/tmp/st")
    s.action(f, *d)
    t0 = time.time()
    st = s.time_step(projection_query, f)
    print(f"step {f}: nc {st['nc']} its {st['cg_iters']} unconverged {st['unconverged']} max_res {st['max_rel_residual']:.1e} delta {st['last_delta']:.1e} {time.time() - t0:.2f} s", flush=True)
    if st["unconverged"] > 0 and n_replayed < 3:
        n_replayed += 1
        s.save_all("/tmp/st_after") if os.makedirs("/tmp/st_after", exist_ok=True) is None else None
        for tol in tols:
            s.load_all("/tmp/st")
            ctx.set_param("direct_piv_tol", tol)
            s.action(f, *d)
            t0 = time.time()
            r = s.time_step(projection_query, f)
            print(f"   replay tol {tol:g}: nc {r['nc']} its {r['cg_iters']} unconverged {r['unconverged']} max_res {r['max_rel_residual']:.1e} delta {r['last_delta']:.1e} {time.time() - t0:.2f} s", flush=True)
        s.load_all("/tmp/st_after")
        ctx.set_param("direct_piv_tol", 1e-8)
