"""Probe (not part of the product): how the constraint SET of the bench scene changes from step to step -- which share of the steps
would find their set inside the union of the sets of earlier steps (a plan made for a superset stays valid)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from thinshelllab_amd.engine.geometry import projection_query
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
args = types.SimpleNamespace(workload=wl, grid={"cfg4": 224, "cfg3": 200}[wl], cloth_size=None, idle=0)
s = bench.build_scene(args, 0)
ctx = s._ensure_ctx(); ctx.set_param("direct", 1)
hist = []
for f in range(1, T + 1):
    s.action(f, *bench._drive(s.gripper.n_part, s._bench_gs, 0, f, 0))
    st = s.time_step(projection_query, f)
    cur = set(map(tuple, ctx.constraints()["idx"].tolist()))
    prev = hist[-1] if hist else set()
    uni = set().union(*hist) if hist else set()
    u3 = set().union(*hist[-3:]) if hist else set()
    print(f"step {f:3d}: nc {len(cur):4d}  new vs previous {len(cur - prev):3d}  dropped {len(prev - cur):3d}  new vs union of last 3 {len(cur - u3):3d}  new vs union of all {len(cur - uni):3d} (union {len(uni)})  plans {st['plans']}")
    hist.append(cur)
