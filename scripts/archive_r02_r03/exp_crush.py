"""Probe (not part of the product): the cfg4 rollout of bench.py with a per-step report of the FEM bodies -- range of det F over the
tetrahedra of every body, bounding box, largest |x - x_prev| -- to see what state the late steps with stalled solves are in.
usage: exp_crush.py [steps] [idle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from argparse import Namespace
import bench
from thinshelllab_amd.engine.geometry import projection_query

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
idle = int(sys.argv[2]) if len(sys.argv) > 2 else 0
args = Namespace(workload="cfg4", grid=224, cloth_size=0.12)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx()
for kv in os.environ.get("TSL_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
for f in range(1, T + 1):
    d = bench._drive(s.gripper.n_part, 1.0, 0, f, int(os.environ.get('IDLE', '0')))
    prev = s.pos.to_numpy().copy()
    s.action(f, *d)
    st = s.time_step(projection_query, f)
    x = s.pos.to_numpy()
    line = f"step {f}: nc {st['nc']} newton {st['newton_iters']} its {st['cg_iters']} unconverged {st['unconverged']} max_res {st['max_rel_residual']:.1e} delta {st['last_delta']:.1e} |dx|max {np.abs(x - prev).max():.2e}"
    for e in s.elastics:
        tets = np.asarray(e.F_vertices.to_numpy()).reshape(-1, 4) + e.offset
        B = e.F_B.to_numpy().reshape(-1, 3, 3)
        Ds = np.stack([x[tets[:, k]] - x[tets[:, 3]] for k in range(3)], axis=2)
        J = np.linalg.det(Ds @ B)
        line += f" | J [{J.min():.3g}, {J.max():.3g}]"
    print(line, flush=True)
